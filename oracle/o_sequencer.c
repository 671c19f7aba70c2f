/* oracle/o_sequencer.c -- CPU restatement of the reference's Sequencer for the per-voice event path.
 * TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Follows src/sequencer.rs (ReplayMode::None, no loop point, no backend, no edits): push :355-398, ready_to_active
 * :584-605, end_of_event :685-703, tick :769-836, process :838-951, fade_in / fade_out :122-216, Fade::at :51-56,
 * smooth5 / sine_ease math.rs:418-420,453-458, delerp :218-220.
 * One generalisation, needed to check a voice BANK: every event may read its own input stream (the reference feeds
 * all events the sequencer's one input; with identical streams the arithmetic is the same).  Besides the mix the
 * renderer returns every event's own faded contribution, which is what the device writes per voice. */
#include "fundsp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAXB 64
#define O_MAX_CH 64 /* as in fundsp_oracle.c */

typedef struct {
    onode *unit;
    double start_time, end_time, fade_in, fade_out;
    int ease; /* 0 = Fade::Power, 1 = Fade::Smooth */
    int index; /* position in push order = voice number */
    int state; /* 0 ready, 1 active, 2 past */
} oevent;

struct oseq {
    int inputs, outputs;
    double sample_rate, sample_duration, time, active_threshold;
    oevent *ev;
    int n, cap;
    int *active; /* indices into ev, in the reference's `active` vector order */
    int n_active;
};

static float smooth5f(float x) { return ((x * 6.0f - 15.0f) * x + 10.0f) * x * x * x; } /* math.rs:418-420 */
static float sine_easef(float x) {                                                       /* math.rs:453-458 */
    const float PI_F = (float)3.14159265358979323846, HALF_PI_F = (float)(3.14159265358979323846 * 0.5);
    const float D = (float)(5.0 * 3.14159265358979323846 * 3.14159265358979323846);
    x = x * HALF_PI_F;
    return 16.0f * x * (PI_F - x) / (D - 4.0f * x * (PI_F - x));
}
static float ease_at(int ease, float x) { return ease == 0 ? sine_easef(x) : smooth5f(x); }
static double delerp(double a, double b, double x) { return (x - a) / (b - a); }
static size_t round_usize(double x) { /* `round(x) as usize`: round half away from zero, saturating cast */
    double r = round(x);
    if (!(r > 0.0)) return 0;
    if (r >= 1.8446744073709552e19) return (size_t)-1;
    return (size_t)r;
}

oseq *o_seq_new(int inputs, int outputs, double sample_rate) { /* Sequencer::new :312-341 + set_sample_rate */
    oseq *s = (oseq *)calloc(1, sizeof(oseq));
    s->inputs = inputs;
    s->outputs = outputs;
    s->sample_rate = sample_rate;
    s->sample_duration = 1.0 / sample_rate;
    return s;
}
void o_seq_free(oseq *s) {
    if (!s) return;
    for (int i = 0; i < s->n; i++) o_free(s->ev[i].unit);
    free(s->ev);
    free(s->active);
    free(s);
}
int o_seq_push(oseq *s, double start, double end, int ease, double fade_in, double fade_out, onode *unit) { /* :355-398 */
    if (o_inputs(unit) != s->inputs || o_outputs(unit) != s->outputs) return -1;
    if (fade_in > end - start || fade_out > end - start) return -1;
    if (s->n == s->cap) {
        s->cap = s->cap ? s->cap * 2 : 16;
        s->ev = (oevent *)realloc(s->ev, (size_t)s->cap * sizeof(oevent));
        s->active = (int *)realloc(s->active, (size_t)s->cap * sizeof(int));
    }
    o_set_sample_rate(unit, s->sample_rate);
    oevent e = {unit, start, end, fade_in, fade_out, ease, s->n, 0};
    if (start < s->active_threshold) { /* push_event :392-394 */
        e.state = 1;
        s->ev[s->n] = e;
        s->active[s->n_active++] = s->n;
    } else {
        s->ev[s->n] = e;
    }
    return s->n++;
}
static void ready_to_active(oseq *s, double next_end_time) { /* :584-605; the ready heap pops by ascending start time */
    s->active_threshold = next_end_time - s->sample_duration * 0.5;
    for (;;) {
        int best = -1;
        for (int i = 0; i < s->n; i++)
            if (s->ev[i].state == 0 && (best < 0 || s->ev[i].start_time < s->ev[best].start_time)) best = i;
        if (best < 0 || !(s->ev[best].start_time < s->active_threshold)) break;
        s->ev[best].state = 1;
        s->active[s->n_active++] = best;
    }
}
static void end_of_event(oseq *s, int i) { /* :685-703: swap_remove */
    s->ev[s->active[i]].state = 2;
    s->active[i] = s->active[s->n_active - 1];
    s->n_active--;
}

/* One Sequencer::process call (:838-951).  in: [n events][inputs][MAXB] (per-event input block), mix: [outputs][MAXB],
 * per_event: [n events][outputs][MAXB] or NULL (each event's faded contribution at its block position, 0 elsewhere). */
void o_seq_process(oseq *s, int size, const float *in, float *mix, float *per_event) {
    if (size == 0) return;
    memset(mix, 0, (size_t)s->outputs * MAXB * sizeof(float));
    if (per_event) memset(per_event, 0, (size_t)s->n * s->outputs * MAXB * sizeof(float));
    const double sd = s->sample_duration, time = s->time;
    const double end_time = time + sd * (double)size; /* loop_point = infinity */
    ready_to_active(s, end_time);
    float bi[O_MAX_CH * MAXB], bo[O_MAX_CH * MAXB];
    int i = 0;
    while (i < s->n_active) {
        oevent *e = &s->ev[s->active[i]];
        if (e->end_time <= time + 0.5 * sd) {
            end_of_event(s, i);
            continue;
        }
        size_t start_index = e->start_time <= time ? 0 : round_usize((e->start_time - time) * s->sample_rate);
        size_t end_index = (size_t)size;
        if (!(e->end_time >= end_time)) {
            size_t r = round_usize((e->end_time - time) * s->sample_rate);
            end_index = r < (size_t)size ? r : (size_t)size;
        }
        if (end_index > start_index) {
            const int n = (int)(end_index - start_index);
            const float *ein = in + (size_t)e->index * s->inputs * MAXB;
            for (int c = 0; c < s->inputs; c++) /* input.span(start_index, n) :866-871 */
                for (int j = 0; j < n; j++) bi[c * MAXB + j] = ein[c * MAXB + (int)start_index + j];
            memset(bo, 0, sizeof bo);
            o_process(e->unit, n, bi, bo);
            /* fade_in :122-167 (indices are those of the event's own buffer, which starts at start_index) */
            {
                double fade_start = e->start_time, fade_end = fade_start + e->fade_in;
                if (e->fade_in > 0.0 && fade_end > time) {
                    size_t fade_end_i = fade_end >= end_time ? end_index : round_usize((fade_end - time) / sd);
                    float phase = (float)delerp(fade_start, fade_end, time + (double)start_index * sd);
                    float d = (float)(sd / e->fade_in);
                    if (fade_end_i > MAXB) fade_end_i = MAXB;
                    for (int c = 0; c < s->outputs; c++) {
                        float fade = phase;
                        for (size_t j = 0; j < fade_end_i; j++) {
                            bo[c * MAXB + j] *= ease_at(e->ease, fade);
                            fade += d;
                        }
                    }
                }
            }
            /* fade_out :169-216 */
            {
                double fade_end = e->end_time, fade_start = fade_end - e->fade_out;
                if (e->fade_out > 0.0 && fade_start < end_time) {
                    size_t fade_i = fade_start <= time ? 0 : round_usize((fade_start - time) / sd);
                    float phase = (float)delerp(fade_start, fade_end, time + (double)fade_i * sd);
                    float d = (float)(sd / e->fade_out);
                    for (int c = 0; c < s->outputs; c++) {
                        float fade = phase;
                        for (size_t j = fade_i; j < end_index; j++) {
                            bo[c * MAXB + j] *= ease_at(e->ease, 1.0f - fade);
                            fade += d;
                        }
                    }
                }
            }
            for (int c = 0; c < s->outputs; c++) /* :911-927 (the f32x8 and scalar adds are the same sums) */
                for (size_t j = start_index; j < end_index; j++) {
                    mix[c * MAXB + j] += bo[c * MAXB + j - start_index];
                    if (per_event) per_event[((size_t)e->index * s->outputs + c) * MAXB + j] = bo[c * MAXB + j - start_index];
                }
        }
        i++;
    }
    s->time = end_time;
}

/* One Sequencer::tick call (:769-836).  in: [n events][inputs], mix: [outputs], per_event: [n events][outputs] or NULL */
void o_seq_tick(oseq *s, const float *in, float *mix, float *per_event) {
    const double sd = s->sample_duration;
    for (int c = 0; c < s->outputs; c++) mix[c] = 0.0f;
    if (per_event) memset(per_event, 0, (size_t)s->n * s->outputs * sizeof(float));
    const double end_time = s->time + sd;
    ready_to_active(s, end_time);
    float tb[O_MAX_CH];
    int i = 0;
    while (i < s->n_active) {
        oevent *e = &s->ev[s->active[i]];
        if (e->end_time <= s->time + 0.5 * sd) {
            end_of_event(s, i);
            continue;
        }
        o_tick(e->unit, in + (size_t)e->index * s->inputs, tb);
        if (e->fade_in > 0.0) {
            float f = (float)delerp(e->start_time, e->start_time + e->fade_in, s->time);
            if (f < 1.0f)
                for (int c = 0; c < s->outputs; c++) tb[c] *= ease_at(e->ease, f);
        }
        if (e->fade_out > 0.0) {
            float f = (float)delerp(e->end_time - e->fade_out, e->end_time, s->time);
            if (f > 0.0f)
                for (int c = 0; c < s->outputs; c++) tb[c] *= ease_at(e->ease, 1.0f - f);
        }
        for (int c = 0; c < s->outputs; c++) {
            mix[c] += tb[c];
            if (per_event) per_event[(size_t)e->index * s->outputs + c] = tb[c];
        }
        i++;
    }
    s->time = end_time;
}

/* Render `length` frames in blocks of <= 64 (process = 1) or sample by sample (process = 0).
 * in: [n events][inputs][length], mix: [outputs][length], per_event: [n events][outputs][length] or NULL. */
void o_seq_render(oseq *s, size_t length, int process, const float *in, float *mix, float *per_event) {
    const int ne = s->n, ni = s->inputs, no = s->outputs;
    float *bi = (float *)calloc((size_t)(ne ? ne : 1) * (ni ? ni : 1) * MAXB, sizeof(float));
    float *bm = (float *)calloc((size_t)no * MAXB, sizeof(float));
    float *bp = per_event ? (float *)calloc((size_t)(ne ? ne : 1) * no * MAXB, sizeof(float)) : NULL;
    for (size_t t0 = 0; t0 < length; t0 += MAXB) {
        int nn = (int)(length - t0 < MAXB ? length - t0 : MAXB);
        if (process) {
            for (int e = 0; e < ne; e++)
                for (int c = 0; c < ni; c++)
                    for (int j = 0; j < nn; j++) bi[((size_t)e * ni + c) * MAXB + j] = in[((size_t)e * ni + c) * length + t0 + j];
            o_seq_process(s, nn, bi, bm, bp);
            for (int c = 0; c < no; c++)
                for (int j = 0; j < nn; j++) mix[(size_t)c * length + t0 + j] = bm[c * MAXB + j];
            if (bp)
                for (int e = 0; e < ne; e++)
                    for (int c = 0; c < no; c++)
                        for (int j = 0; j < nn; j++) per_event[((size_t)e * no + c) * length + t0 + j] = bp[((size_t)e * no + c) * MAXB + j];
        } else {
            for (int j = 0; j < nn; j++) {
                float *ti = (float *)bi; /* [ne][ni] */
                for (int e = 0; e < ne; e++)
                    for (int c = 0; c < ni; c++) ti[(size_t)e * ni + c] = in[((size_t)e * ni + c) * length + t0 + j];
                float tm[O_MAX_CH];
                o_seq_tick(s, ti, tm, bp);
                for (int c = 0; c < no; c++) mix[(size_t)c * length + t0 + j] = tm[c];
                if (bp)
                    for (int e = 0; e < ne; e++)
                        for (int c = 0; c < no; c++) per_event[((size_t)e * no + c) * length + t0 + j] = bp[(size_t)e * no + c];
            }
        }
    }
    free(bi);
    free(bm);
    free(bp);
}
double o_seq_time(const oseq *s) { return s->time; }

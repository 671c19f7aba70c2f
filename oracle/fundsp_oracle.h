/*
 * oracle/fundsp_oracle.h -- TEST INFRASTRUCTURE ONLY (CPU oracle for the FunDSP hot path).
 * See fundsp_oracle.c for the reference citations.  Loaded with ctypes from tests/oracle.py.
 */
#ifndef FUNDSP_ORACLE_H
#define FUNDSP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct onode onode;

enum {
    O_CONSTANT = 0, O_PASS, O_SINE, O_NOISE, O_SVF, O_FIXED_SVF, O_BIQUAD, O_BUTTER_LOWPASS, O_RESONATOR,
    O_BIQUAD_BANK, O_MOOG, O_FIR, O_TICK, O_DELAY, O_PIPE, O_STACK, O_BINOP, O_UNOP,
    O_WAVESYNTH, O_ADSR_LIVE, O_PANNER, O_REVERB_STEREO, O_SHAPER, O_PHASE_OSC, O_CHAOS, O_NLBIQUAD, O_TAP, O_ALLNEST,
    O_ONEPOLE, O_PINKPASS, O_MORPH, O_REZ, O_FOLLOW, O_AFOLLOW, O_MLS, O_OVERSAMPLE, O_DSF, O_PLUCK, O_ENVELOPE, O_RESAMPLE, O_ENVELOPE_IN,
    O_MULTIPASS, O_SINK, O_SPLIT, O_JOIN, O_REVERSE, O_IMPULSE, O_MAP, O_BRANCH, O_BUS, O_THRU, O_MULTI, O_DECLICK, O_FEEDBACK, O_PHASESYNTH, O_WRAP, O_METER, O_VAR, O_LIMITER, O_REVERB3, O_MIXER, O_HOLD, O_WAVEPLAYER
};
enum { O_OP_LOWPOLE = 0, O_OP_HIGHPOLE, O_OP_DCBLOCK, O_OP_ALLPOLE };
enum { O_SH_CLIP = 0, O_SH_CLIPTO, O_SH_TANH, O_SH_ATAN, O_SH_SOFTSIGN, O_SH_CRUSH, O_SH_SOFTCRUSH, O_SH_ADAPTIVE_TANH,
       O_SH_ADAPTIVE /* + inner shape: Adaptive<S> for any S (shape.rs:162-201) */ };
enum { O_OSC_RAMP = 0, O_OSC_POLYSAW, O_OSC_POLYSQUARE, O_OSC_POLYPULSE };
/* SvfMode order follows src/svf.rs:281-742 */
enum {
    O_SVF_LOWPASS = 0, O_SVF_HIGHPASS, O_SVF_BANDPASS, O_SVF_NOTCH, O_SVF_PEAK, O_SVF_ALLPASS,
    O_SVF_BELL, O_SVF_LOWSHELF, O_SVF_HIGHSHELF
};
enum { O_BQ_BUTTER = 0, O_BQ_RESONATOR, O_BQ_LOWPASS, O_BQ_HIGHPASS, O_BQ_BELL };
enum { O_ADD = 0, O_SUB, O_MUL };
enum { O_NEG = 0, O_ID, O_ADD_SCALAR, O_NEG_ADD_SCALAR, O_MUL_SCALAR };

/* leaves */
typedef void (*o_map_fn)(const float *in, float *out, void *ctx); /* closure callback of Map / ShapeFn / VarFn */
onode *o_constant(int n, const float *v);
onode *o_pass(void);
onode *o_sine(void);
onode *o_noise(void);
onode *o_fixed_svf(int mode, float cutoff, float q, float gain);
onode *o_svf(int mode, float cutoff, float q, float gain);
onode *o_biquad(float a1, float a2, float b0, float b1, float b2);
onode *o_butter_lowpass(int inputs, float cutoff);
onode *o_resonator(int inputs, float center, float q);
onode *o_biquad_bank(void);
void o_biquad_bank_set(onode *n, int index, float a1, float a2, float b0, float b1, float b2);
onode *o_moog(int inputs, float cutoff, float q);
onode *o_fir(int n_taps, const float *w);
onode *o_tick_node(int channels);
onode *o_delay(double time);
/* wavetables are shared data (Arc<Wavetable> in the reference): created once, referenced by synth nodes */
typedef struct owavetable owavetable;
owavetable *o_wavetable_create(int n_tables, const float *pitches, const int *lengths, const float *data);
void o_wavetable_free(owavetable *t);
onode *o_wavesynth(const owavetable *table, int outputs);
/* WavePlayer (wave.rs:739, ID 65) over caller-owned wave data [channels][length]; loop_point < 0 = None */
onode *o_waveplayer(const float *data, int channels, size_t length, int channel, size_t start_point, size_t end_point,
                   long loop_point);
onode *o_phasesynth(const owavetable *table);   /* PhaseSynth wavetable.rs:358 (ID 35): input 0 = phase */
/* A node that IS a fixed inner graph under its own ID, pinging the inner graph first and hashing its ID last:
 * PulseWave (wavetable.rs:437-491, ID 44).  Takes ownership of x. */
onode *o_wrap(onode *x, uint64_t id);
void o_wavesynth_set_phase(onode *n, float phase);
onode *o_adsr_live(float attack, float decay, float sustain, float release);
onode *o_panner(int inputs, float pan);
/* Shaper<S> (shape.rs:205); for O_SH_ADAPTIVE_TANH p0 = hardness, p1 = timescale */
onode *o_shaper_adaptive(int inner, float p0, float p1, float timescale);  /* Shaper<Adaptive<S>>, S = O_SH_CLIP .. O_SH_SOFTCRUSH */
onode *o_tap(int linear, float min_delay, float max_delay);   /* Tap<U1> / TapLinear<U1> (delay.rs:148,386) */
onode *o_multitap(int linear, int taps, float min_delay, float max_delay); /* Tap<N> / TapLinear<N> */
onode *o_allnest2(onode *x);                                  /* AllNest<U2, X>: coefficient on input 1 */
onode *o_allnest(float coefficient, onode *x);                /* AllNest<U1, X> (delay.rs:294), takes ownership of x */
onode *o_onepole(int kind, int inputs, float cutoff_or_delay); /* Lowpole/Highpole/DCBlock/Allpole (filter.rs) */
onode *o_pinkpass(void);
/* Envelope<f32, E, R> (envelope.rs:17): `fn` plays the Rust closure E(t) -> R, writing `outputs` (<= 8) values */
typedef void (*o_env_fn)(float t, float *out, void *ctx);
onode *o_envelope(float interval, int outputs, o_env_fn fn, void *ctx);
void o_envfn_criterion_envelope(float t, float *out, void *ctx);   /* the closures of benches/benchmark.rs:57,94 (criterion) */
void o_envfn_criterion_phaser(float t, float *out, void *ctx);
/* EnvelopeIn<f32, E, I, R> (envelope.rs:185) with a stateless closure E(t, &inputs) -> R */
typedef void (*o_envin_fn)(float t, const float *in, float *out, void *ctx);
onode *o_envelope_in(float interval, int inputs, int outputs, o_envin_fn fn, void *ctx);
/* Pluck (oscillator.rs:215); excitation = the funutd Rnd stream the reference draws in initialize_line, given by the caller */
onode *o_pluck(float frequency, float gain_per_second, float high_frequency_damping, const float *excitation, size_t n_exc);
onode *o_dsf(int inputs, float harmonic_spacing, float roughness); /* Dsf<U1/U2> (oscillator.rs:121) */
onode *o_resample(onode *x);                                     /* Resample<X> (resample.rs:210): x a generator; input 0 = speed */
onode *o_oversample(onode *x);                                   /* Oversampler<X> (oversample.rs:66), takes ownership of x */
onode *o_rez(int inputs, float bandpass, float cutoff, float q); /* Rez<f32, U1/U3> (rez.rs): inputs 1 or 3 */
enum { O_METER_SAMPLE = 0, O_METER_PEAK, O_METER_RMS };
onode *o_meter(int mode, double timescale, int monitor);        /* MeterNode (dynamics.rs:398, ID 61) / Monitor (:441, ID 56) */
float o_meter_level(const onode *n);                             /* what Monitor stores in its Shared */
onode *o_mixer(int inputs, int outputs, const float *matrix);   /* Mixer<M, N> (pan.rs:95, ID 84); matrix[out][in] */
onode *o_var_fn(float value, int outputs, o_map_fn fn, void *ctx); /* VarFn (shared.rs:136, ID 70): fn(&value, out) */
onode *o_var(float value);                                       /* Var (shared.rs:85, ID 68) */
void o_var_set(onode *n, float value);
onode *o_limiter(int channels, float attack_time, float release_time); /* Limiter<N> (dynamics.rs:125, ID 25) */
onode *o_follow(float response_time);                           /* Follow<f32> (follow.rs:31) */
onode *o_afollow(float attack_time, float release_time);         /* AFollow<f32> (follow.rs:137) */
/* Hold (noise.rs:242, ID 76): `draws` is the stream Rnd::from_u64(hash).f64() of funutd (crate source absent: supplied
 * by the caller, as for o_pluck); reset restarts it */
onode *o_hold(float variability, const double *draws, size_t n_draws);
onode *o_mls(unsigned bits);                                     /* Mls (noise.rs:103), 1 <= bits <= 31 */
void o_mls_set_seed(onode *n, uint64_t seed);
uint64_t o_mls_period(unsigned bits);                            /* test helper: cycle length from the all-ones state */
onode *o_morph(float cutoff, float q, float morph);           /* Morph (svf.rs:1040) */
onode *o_shaper(int shape, float p0, float p1);
onode *o_phase_osc(int kind);                       /* Ramp / PolySaw / PolySquare / PolyPulse (oscillator.rs:441-760) */
void o_osc_set_phase(onode *n, float phase);
onode *o_chaos(int lorenz);                         /* Rossler / Lorenz (oscillator.rs:323-435) */
/* FbBiquad / DirtyBiquad and their Fixed variants (biquad.rs:494-920): mode = O_BQ_RESONATOR..O_BQ_BELL, inputs 1/3/4 */
onode *o_nlbiquad(int dirty, int inputs, int mode, int shape, float p0, float p1, float center, float q, float gain);
float o_math_atanf(float x);
float o_math_wide_atanf(float x);
double o_adaptive_smoothing(float timescale, double sample_rate);
/* reverb_stereo(room_size, time, damping): 32-line FDN (prelude.rs:1732-1762). */
onode *o_reverb_stereo(double room_size, double time, double damping);
/* Reverb<F> (reverb.rs:152, ID 85) = reverb3_stereo(time, diffusion, filter): `filters` are the 16 clones of the loop
 * filter in block order (filter0, filter1 of block 0, then block 1, ...); takes ownership */
onode *o_reverb3(double time, double diffusion, onode **filters);
/* derived constants of reverb_stereo at `sample_rate`: FIR weights (3), delay lengths in samples (32), pan weights (32+32) */
void o_reverb_stereo_params(double room_size, double time, double damping, double sample_rate, float *w3, int *delays32,
                            float *wl32, float *wr32);
/* combinators (take ownership of children) */
/* routing leaves and the remaining combinators (audionode.rs) */
onode *o_multipass(int n);                 /* MultiPass<N> :373 */
onode *o_sink(int n);                      /* Sink<N> :437 */
onode *o_split(int m, int n);              /* Split<N> (m == 1, :527) / MultiSplit<M, N> (:571) */
onode *o_join(int m, int n);               /* Join<N> (m == 1, :617) / MultiJoin<M, N> (:668) */
onode *o_reverse(int n);                   /* Reverse<N> :2808 */
onode *o_impulse(int n);                   /* Impulse<N> :2841 */
onode *o_map(int inputs, int outputs, o_map_fn fn, void *ctx); /* Map<M, I, O> :1330 */
onode *o_shape_fn(o_map_fn fn, void *ctx);  /* Shaper<ShapeFn<S>> shape.rs:35,205 (ID 42): fn maps in[0] -> out[0] */
onode *o_declick(float duration);          /* Declick<f32> dynamics.rs:245 */
onode *o_branch(onode *x, onode *y);       /* Branch :1653 */
onode *o_bus(onode *x, onode *y);          /* Bus :1796 */
/* Feedback<N, X, U> (feedback.rs:71, y == NULL) / Feedback2<N, X, Y, U> (:193); hadamard: U = FrameHadamard (fdn,
 * fdn2) instead of FrameId (feedback, feedback2).  Takes ownership. */
onode *o_feedback(onode *x, onode *y, int hadamard);
onode *o_thru(onode *x);                   /* Thru :1951 */
enum { O_MULTI_BUS = 0, O_MULTI_STACK, O_MULTI_BRANCH, O_MULTI_REDUCE, O_MULTI_CHAIN };
/* MultiBus :2065 / MultiStack :2211 / MultiBranch :2532 / Reduce :2366 (op = O_ADD..O_MUL) / Chain :2673; takes
 * ownership of the n nodes, which must have equal arities */
onode *o_multi(int kind, int n, onode **nodes, int op);
onode *o_pipe(onode *x, onode *y);
onode *o_stack(onode *x, onode *y);
onode *o_binop(int op, onode *x, onode *y);
onode *o_unop(int op, onode *x, float scalar);
void o_free(onode *n);

int o_inputs(const onode *n);
int o_outputs(const onode *n);
void o_reset(onode *n);
void o_set_sample_rate(onode *n, double sr);
void o_set_seed(onode *n, uint64_t seed);
void o_sine_set_phase(onode *n, float phase);
void o_noise_set_seed(onode *n, uint64_t seed);
uint64_t o_sine_hash(const onode *n);
float o_sine_phase(const onode *n);
uint32_t o_noise_state(const onode *n);

void o_tick(onode *n, const float *in, float *out);
void o_process(onode *n, int size, const float *in, float *out);
size_t o_wave_render(onode *n, double sample_rate, double duration, float *out, size_t capacity);
void o_render_blocks(onode *n, size_t length, int block, const float *in, float *out);
void o_render_ticks(onode *n, size_t length, const float *in, float *out);

/* coefficient helpers + scalar math (unit tests) */
void o_svf_coefs(int mode, float sr, float cutoff, float q, float gain, float *out6);
void o_biquad_coefs(int kind, float sr, float f, float q, float gain, float *out5);
void o_moog_coefs(float sr, float cutoff, float q, float *out3);
/* Wavetable::new for a built-in table (o_wavetable.c): 0 saw, 1 square, 2 triangle, 4 organ, 5 soft saw, 6 hammond */
int o_make_wavetable(int kind, int max_tables, float *pitches, int *lengths, size_t data_cap, float *data);
float o_math_sinf(float x);
float o_math_cosf(float x);
float o_math_tanf(float x);
float o_math_tanhf(float x);
float o_math_expf(float x);
float o_math_expm1f(float x);
float o_math_wide_sinf(float x);
double o_math_rnd1(uint64_t x);
uint64_t o_math_hash1(uint64_t x);
uint64_t o_math_atto(uint64_t state, uint64_t data);
uint32_t o_math_hash32x(uint32_t x);

/* voice-bank driver (oracle/o_bank.c): V independent voices of one BASELINE config, used as the parity
 * checker at bank scale and as bench.py's cpu_baseline ("port").  Per-voice parameters are inputs (the same
 * arrays are handed to the HIP engine), so nothing about the workload definition is duplicated here.
 *   config 2: voice = noise().seed(seed[v]) >> biquad(BiquadCoefs::lowpass(sr, p0[v]=fc, p1[v]=q))   (one BiquadBank lane)
 *   config 3: voice = sine_hz(p0=f) * f * (p1=m) + f >> sine() >> lowpass_hz(p2=fc, p3=q), then set_seed(seed[v]) */
typedef struct {
    int config;
    int process_mode;   /* 1 = AudioNode::process in <=64-sample blocks (Wave::render chunking), 0 = per-sample tick */
    int out_layout;     /* 0 = [voice][frame] (CPU-natural), 1 = [frame][voice] (device-native), 2 = no store */
    int threads;
    double sample_rate;
    size_t voices, frames;
    const float *p0, *p1, *p2, *p3;
    const uint64_t *seed;
} o_bank_job;
double o_bank_render(const o_bank_job *job, float *out);
/* The same config-3 render through a MONOMORPHISED process() (oracle/o_fast.c): what rustc makes of the reference's
 * statically typed graph -- no node tree, no switch per node per block, both sines 8 frames per vector exactly as
 * Sine::process (oscillator.rs:74-86) with `wide`'s f32x8 arithmetic on the host's SIMD unit, the SVF per sample.
 * Bit-identical to o_bank_render (tests/test_oracle_fast.py); bench.py's cpu_baseline.value.  config 3, process mode. */
typedef struct {
    float f_const, mul_f, mul_m, add_f, mod_phase, mod_sd, car_phase, car_sd, a1, a2, a3, m0, m1, m2, ic1eq, ic2eq;
} o_fm_svf_regs;
int o_fm_svf_state(const onode *g, o_fm_svf_regs *r);
onode *o_bank_build_voice(const o_bank_job *job, size_t v);
/* timing hygiene of the cpu_baseline legs: pin worker t of a bank job to the t-th CPU of the process's affinity mask; how many
 * CPUs that mask holds; pin the calling thread likewise (threads that bench.py starts itself). */
void o_bank_pin_threads(int on);
int o_bank_allowed_cpus(void);
void o_bank_pin_self(int t);
double o_bank_render_fast(const o_bank_job *job, float *out);
/* Configs 4 and 5 the same way (fundsp_oracle.c, bottom): the statically dispatched, inlined process() of the voice graph, made of the
 * tree walk's own node functions -> bit-identical to o_process by construction.  o_c4_open matches
 * `((dc(f) >> saw() | dc(fc) | dc(q)) >> moog()) * ENV >> pan(p)`, ENV = adsr_live (gate = the graph's input) or var(g) >> adsr_live. */
typedef struct {
    onode *g, *saw, *moog, *env, *var, *pan;
    float f, fc, q;
} o_c4_voice;
int o_c4_open(onode *g, o_c4_voice *v);                                       /* 0, or -1 if the tree has another shape */
void o_c4_block(o_c4_voice *v, int size, const float *gate, float *out);     /* gate [64] or NULL (Var shape); out [2][64] */
void o_reverb_stereo_block(onode *n, int size, const float *in, float *out); /* in, out [2][64] */
/* round 6: the monomorphised block forms of reverb3_stereo(.., lowpole_hz(..)) and of the prelude's fdn example (CPU legs; bit-equal to the tree walk) */
int o_reverb3_block_ok(const onode *n);
void o_reverb3_block(onode *n, int size, const float *in, float *out); /* in, out [2][64] */
void o_fdn16_block(float **ring, const size_t *len, size_t *pos, float (*v)[3], float *value, const float *w, int size, const float *in, float *out);
int o_is_reverb_stereo(const onode *n);
/* Threaded drivers of the cpu_baseline legs of configs 4 / 5 (o_fast.c): voices [v0, v1) of a thread's slice rendered one after the other,
 * many per thread, `frames` frames each in 64-sample blocks; returns seconds.  Config 4: p0..p3 = f, fc, q, pan; adsr = a, d, s, r;
 * gate_var != 0: the Var shape, the variable follows plan[] = (value, frames) pairs; else the stream shape, gate[] = [frames].
 * `fast` = 0 renders through the generic tree walk instead (the same job, for the bit-equality test and the tree-walk figure);
 * out (or NULL) = [voice][2][frames]. */
typedef struct {
    int threads, fast, gate_var, n_plan;
    double sample_rate;
    size_t voices, frames;
    const float *p0, *p1, *p2, *p3;
    const uint64_t *seed;
    float adsr[4];
    const float *gate;      /* [frames] (stream shape) */
    const float *plan;      /* n_plan x (value, frames) (Var shape) */
} o_c4_job;
double o_c4_bank_render(const o_c4_job *job, float *out);
/* Config 2 as the reference runs it: BiquadBank<f32x8> (biquad_bank.rs:73-84), eight voices per SIMD instruction, fed by 8 Noise nodes;
 * lane k of bank j = voice 8 j + k of o_bank_render's config 2, bit for bit.  Seconds, or < 0 for another config / tick mode. */
double o_biquad_bank8_render(const o_bank_job *job, float *out);
/* Config 5: `instances` x reverb_stereo(room, time, damping) on the SAME stereo input x [2][frames]; out (or NULL) = [instance][2][frames] */
double o_reverb_bank_render(int threads, int fast, double sample_rate, size_t instances, size_t frames, double room, double time, double damping,
                            const float *x, float *out);
/* round 6: CPU legs (fast = 1: the monomorphised block forms, 0: the generic tree walk) of reverb3_stereo (which = 0: p = time, diffusion, lowpole cutoff) and of the prelude's fdn example (which = 1: p = 16
 * delays + 3 FIR weights); x [inputs][frames] shared by the instances, out [instances][outputs][frames] or NULL; returns seconds */
double o_graph_bank_render(int threads, int which, int fast, const double *p, double sample_rate, size_t instances, size_t frames, const float *x, float *out);
const char *o_fast_simd_flavour(void);

#ifdef __cplusplus
}
#endif

/* ---- Sequencer (sequencer.rs), oracle/o_sequencer.c: one event per voice, per-event inputs allowed --------------- */
typedef struct oseq oseq;
oseq *o_seq_new(int inputs, int outputs, double sample_rate);
void o_seq_free(oseq *s);
/* ease: 0 = Fade::Power, 1 = Fade::Smooth; takes ownership of unit; returns the event's index or -1 */
int o_seq_push(oseq *s, double start, double end, int ease, double fade_in, double fade_out, onode *unit);
void o_seq_process(oseq *s, int size, const float *in, float *mix, float *per_event);
void o_seq_tick(oseq *s, const float *in, float *mix, float *per_event);
void o_seq_render(oseq *s, size_t length, int process, const float *in, float *mix, float *per_event);
double o_seq_time(const oseq *s);

#endif

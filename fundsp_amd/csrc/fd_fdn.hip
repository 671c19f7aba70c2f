// fd_fdn.hip -- FDN reverb kernel (see fd_fdn.hpp).  Compiled with -fgpu-flush-denormals-to-zero: FunDSP's
// Feedback::new calls prevent_denormals() (feedback.rs:96, denormal.rs:18: MXCSR FTZ+DAZ), so the reference renders
// feedback graphs with flushed denormals; this translation unit matches that mode, the rest of the engine keeps IEEE
// denormals.
#include <cmath>

#include "fd_fdn.hpp"
#include "fd_opts.hpp"
#include "fd_math.hpp"

namespace fd {
int simd_count();          // fd_capi.hip
}

namespace fd {

static const double RV_DELAYS[32] = {  // prelude.rs:1739-1744
    0.073904, 0.052918, 0.066238, 0.066387, 0.037783, 0.080073, 0.050961, 0.075900, 0.043646,
    0.072095, 0.056194, 0.045961, 0.058934, 0.068016, 0.047529, 0.058156, 0.072972, 0.036084,
    0.062715, 0.076377, 0.044339, 0.076725, 0.077884, 0.046126, 0.067741, 0.049800, 0.051709,
    0.082923, 0.070121, 0.079315, 0.055039, 0.081859,
};

void fdn_make_const(double room_size, double time, double damping, double sample_rate, FdnConst* c) {
    // a = pow(db_amp(-60.0), 0.03 * room_size / 10.0 / time) as f32, db_amp(x) = exp((x / 20) * LN_10)  (math.rs:76-78,294)
    const double db_amp = std::exp((-60.0 / 20.0) * 2.302585092994046);
    const float a = (float)std::pow(db_amp, 0.03 * room_size / 10.0 / time);
    const float gain = 1.0f - (float)damping;  // fir3(1.0 - damping as f32)  prelude.rs:863-867
    const float alpha = (gain + 1.0f) / 2.0f;
    const float beta = (1.0f - alpha) / 2.0f;
    c->w[0] = beta * a;
    c->w[1] = alpha * a;
    c->w[2] = beta * a;
    int maxlen = 0;
    for (int i = 0; i < 32; i++) {
        const int delay = (int)std::round(RV_DELAYS[i] * room_size / 10.0 * sample_rate);
        c->len[i] = delay + 1;
        maxlen = c->len[i] > maxlen ? c->len[i] : maxlen;
        const float x = (float)((double)i / 31.0);  // sumf: F::from_f64(i / (N-1))  prelude.rs:1613-1616
        const float x2 = x * x;                     // smooth9  math.rs:431-437
        const float t = ((((70.0f * x - 315.0f) * x + 540.0f) * x - 420.0f) * x + 126.0f) * x2 * x2 * x;
        float p = -1.0f * (1.0f - t) + 1.0f * t;    // lerp(-1.0, 1.0, t)
        p = p > -1.0f ? p : -1.0f;                  // pan_weights: clamp11  pan.rs:13-17
        p = p < 1.0f ? p : 1.0f;
        const float angle = (p + 1.0f) * (F32_PI * 0.25f);
        c->wl[i] = cosf_musl(angle);
        c->wr[i] = sinf_musl(angle);
    }
    int cap = 256;
    while (cap < maxlen) cap <<= 1;
    c->cap = cap;
    c->ring_stride = (size_t)32 * ((size_t)cap + 64);
    c->sections = 1;
    c->generic = 0;
    c->lines = 32;
    c->taps = 3;
    c->nin = c->nout = 2;
    c->had_scale = (float)(1.0 / std::sqrt(32.0));
    c->out_scale = (float)(1.0 / 16.0);
}

static const float RV4_DELAYS[32] = {  // prelude.rs:1875-1908 ("Optimized delay times from `optimize.rs` example. Fitness -4546.")
    0.059326634f, 0.04778291f, 0.06995449f, 0.0393001f, 0.041604012f, 0.06215825f, 0.052269846f, 0.043227978f, 0.06966107f, 0.031615064f, 0.068442f,
    0.037332155f, 0.032944717f, 0.034493037f, 0.06787566f, 0.038824916f, 0.068260126f, 0.068044715f, 0.0688076f, 0.066724524f, 0.051293883f, 0.06023173f,
    0.040897705f, 0.031507637f, 0.060309593f, 0.049584292f, 0.04532072f, 0.056379095f, 0.035180368f, 0.041291796f, 0.046129026f, 0.05504605f,
};

void fdn_make_const_reverb4(double room_size, double time, double sample_rate, FdnConst* c) {
    // reverb4_stereo :1909-1913: every delay (f32) *= max(room_size as f32, 15.0) / 10.0; reverb4_stereo_delays :1921-1922: room_size = 10.0,
    // a = pow(db_amp(-60.0), 0.03 * room_size / 10.0 / time) as f32; lines delay(d as f64) >> fir((-a / 4, -a / 2, -a / 4)) :1924-1930
    const double db_amp = std::exp((-60.0 / 20.0) * 2.302585092994046);
    const float a = (float)std::pow(db_amp, 0.03 * 10.0 / 10.0 / time);
    c->w[0] = -a / 4.0f;
    c->w[1] = -a / 2.0f;
    c->w[2] = -a / 4.0f;
    const float rs = (float)room_size, scale = (rs > 15.0f ? rs : 15.0f) / 10.0f;
    int maxlen = 0;
    for (int i = 0; i < 32; i++) {
        const float d = RV4_DELAYS[i] * scale;
        const int delay = (int)std::round((double)d * sample_rate);  // Delay::new(d as f64), set_sample_rate delay.rs:108
        c->len[i] = delay + 1;
        maxlen = c->len[i] > maxlen ? c->len[i] : maxlen;
        c->wl[i] = c->wr[i] = 0.0f;
    }
    for (int i = 0; i < 16; i++) {  // sumf::<U16>(|x| pan(lerp(-1.0, 1.0, smooth9(x)))) over the SECOND network's lines: x = i / 15
        const float x = (float)((double)i / 15.0);
        const float x2 = x * x;
        const float t = ((((70.0f * x - 315.0f) * x + 540.0f) * x - 420.0f) * x + 126.0f) * x2 * x2 * x;
        float p = -1.0f * (1.0f - t) + 1.0f * t;
        p = p > -1.0f ? p : -1.0f;
        p = p < 1.0f ? p : 1.0f;
        const float angle = (p + 1.0f) * (F32_PI * 0.25f);
        c->wl[16 + i] = cosf_musl(angle);
        c->wr[16 + i] = sinf_musl(angle);
    }
    int cap = 256;
    while (cap < maxlen) cap <<= 1;
    c->cap = cap;
    c->ring_stride = (size_t)32 * ((size_t)cap + 64);
    c->sections = 2;
    c->generic = 0;
    c->lines = 32;
    c->taps = 3;
    c->nin = c->nout = 2;
    c->had_scale = (float)(1.0 / std::sqrt(16.0));
    c->out_scale = (float)(1.0 / 4.0);
}

void fdn_make_const_generic(const FdnDesc& d, double sample_rate, FdnConst* c) {
    *c = FdnConst{};
    int maxlen = 0;
    for (int i = 0; i < 32; i++) {
        // Delay::new(t): length = round(t * sample_rate) samples, ring of length + 1 (delay.rs:104-112); lines past the network: unused
        c->len[i] = i < d.lines ? (int)std::round(d.delay[i] * sample_rate) + 1 : 0;
        maxlen = c->len[i] > maxlen ? c->len[i] : maxlen;
    }
    for (int j = 0; j < 3; j++) c->w[j] = j < d.taps ? d.w[j] : 0.0f;
    int cap = 256;
    while (cap < maxlen) cap <<= 1;
    c->cap = cap;
    c->sections = 1;
    c->generic = 1;
    c->lines = d.lines;
    c->taps = d.taps;
    c->nin = d.nin;
    c->nout = d.nout;
    c->ring_stride = (size_t)d.lines * ((size_t)cap + 64);
    c->had_scale = (float)(1.0 / std::sqrt((double)d.lines));  // (1.0 / sqrt(N as f64)) as f32  feedback.rs:57
    c->out_scale = 1.0f;
}

constexpr int TS = 65;  // LDS row stride (floats): lane-per-row access is bank-conflict free

__global__ __launch_bounds__(256) void k_fdn_reset(FdnConst c, FdnState s, size_t instances) {
    // zero the rings and the per-line state (Feedback::reset feedback.rs:123-126 -> Delay::reset, Fir::reset)
    const size_t total = instances * c.ring_stride;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) s.rings[i] = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < instances * 32; i += (size_t)gridDim.x * 256) {
        if (i < instances) s.wpos[i] = 0;
        s.v1[i] = 0.0f;
        s.v2[i] = 0.0f;
        s.fb[i] = 0.0f;
    }
}

// Value of lane (l ^ h) inside each 32-lane instance.  Strides 1..8 are pure DPP register moves (no LDS round trip on
// the per-sample critical path): xor 1/2 = quad_perm; xor 4 = quad reversal then row_half_mirror (i -> 7-i);
// xor 8 = row_half_mirror then row_mirror (i -> 15-i).  Stride 16 crosses DPP rows: gfx950's v_permlane16_swap.
template <int H>
__device__ __forceinline__ float xor_lane(float v, bool upper16) {
    int x = __builtin_bit_cast(int, v), r;
    if (H == 1) r = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false);       // quad_perm [1,0,3,2]
    else if (H == 2) r = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
    else if (H == 4) {
        int t = __builtin_amdgcn_update_dpp(x, x, 0x1B, 0xF, 0xF, false);            // quad_perm [3,2,1,0]
        r = __builtin_amdgcn_update_dpp(t, t, 0x141, 0xF, 0xF, false);               // row_half_mirror
    } else if (H == 8) {
        int t = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false);           // row_half_mirror
        r = __builtin_amdgcn_update_dpp(t, t, 0x140, 0xF, 0xF, false);               // row_mirror
    } else {
        // permlane16_swap(a, b): a.row1 <-> b.row0 and a.row3 <-> b.row2.  With a = b = x: a's odd rows receive the
        // even rows' values and b's even rows receive the odd rows' values.
        auto sw = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
        r = upper16 ? (int)sw[0] : (int)sw[1];
    }
    return __builtin_bit_cast(float, r);
}

__device__ __forceinline__ void fdn_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// layout 0: voice-minor [ch][frame][instance]; layout 1: planar [instance][ch][fstride]
// IPW = instances per wave.  2 fills all 64 lanes of the recurrence; 1 leaves lanes 32-63 idle there but halves the LDS
// tiles (two workgroups per CU) and doubles the number of waves -- the per-sample recurrence is a chain of dependent
// VALU / DPP operations (a lone wave issues one every ~9 cycles), so a second resident wave per SIMD is worth more than
// full lanes whenever the bank has fewer than two 2-instance waves per SIMD (BASELINE config 5: 2048 per GPU).
template <int IPW>
__global__ __launch_bounds__(256) void k_fdn_render(FdnConst c, FdnState s, size_t V, const float* __restrict__ in,
                                                    float* __restrict__ out, size_t T, size_t fstride, int layout, FdnBus bus) {
    constexpr int R = 32 * IPW;                // ring rows (delay lines) per wave
    __shared__ float tile_all[4][R * TS];      // ring samples in / new ring samples out, one row per (instance, line)
    __shared__ float tileo_all[4][R * TS];     // line outputs of the block, for the ordered pan sum
    __shared__ float tin_all[4][2 * IPW * 64]; // [instance in wave][channel][frame]
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    float* tile = tile_all[wib];
    float* tileo = tileo_all[wib];
    float* tin = tin_all[wib];
    const size_t inst0 = ((size_t)blockIdx.x * 4 + wib) * IPW;
    if (inst0 >= V) return;
    const int j = IPW == 2 ? lane >> 5 : 0, k = lane & 31;
    const bool line = lane < R;                // this lane owns a delay line in the recurrence
    const size_t inst = inst0 + j;
    const bool valid = line && inst < V;
    const size_t sidx = (valid ? inst : inst0) * 32 + k;
    int wp = s.wpos[valid ? inst : inst0];  // the instance's write position (same value in its 32 lanes)
    float v1 = s.v1[sidx], v2 = s.v2[sidx], fb = s.fb[sidx];
    const int cmask = c.cap - 1;
    const size_t cp = (size_t)c.cap + 64;
    const float w0 = c.w[0], w1 = c.w[1], w2 = c.w[2];
    const float scale = c.had_scale;  // (1.0 / sqrt(N as f64)) as f32  feedback.rs:57
    uint32_t negmask[5];
#pragma unroll
    for (int st = 0; st < 5; st++) negmask[st] = (k & (1 << st)) ? 0x80000000u : 0u;

    // Ring rows and inputs of a block are fetched one block AHEAD into registers (issued before the recurrence of the
    // current block, consumed after it): every delay exceeds 128 samples, so the rows block b+1 reads are not touched
    // by block b's writes, and the HBM latency hides behind phase 2.  In these phases lane = frame (all 64 lanes).
    float xr[R], xin[2 * IPW];
    auto fetch = [&](size_t t0n, int wp_now) {
        const int sizen = (int)((T - t0n) < 64 ? (T - t0n) : 64);
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int kk = r & 31;
            const size_t ri = inst0 + (r >> 5);
            const int w0 = __builtin_amdgcn_readlane(wp_now, r);
            const int pos = (w0 + lane - (c.len[kk] - 1)) & cmask;  // frame `lane` reads the slot written len - 1 frames earlier
            xr[r] = (ri < V && lane < sizen) ? s.rings[ri * c.ring_stride + (size_t)kk * cp + (size_t)pos] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 2 * IPW; q++) {  // q = instance-in-wave * 2 + channel
            const size_t ri = inst0 + (q >> 1);
            const int ch = q & 1;
            xin[q] = (ri < V && lane < sizen)
                         ? (layout == 0 ? in[((size_t)ch * T + t0n + lane) * V + ri] : in[(ri * 2 + ch) * fstride + t0n + lane])
                         : 0.0f;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int r = 0; r < R; r++) tile[r * TS + lane] = xr[r];
#pragma unroll
        for (int q = 0; q < 2 * IPW; q++) tin[q * 64 + lane] = xin[q];
    };
    fetch(0, wp);
    stage();
    for (size_t t0 = 0; t0 < T; t0 += 64) {
        const int size = (int)((T - t0) < 64 ? (T - t0) : 64);
        const bool more = t0 + 64 < T;
        if (more) {  // ---- phase 1 of the NEXT block: loads in flight during this block's recurrence
            fetch(t0 + 64, (wp + 64) & cmask);
        }
        fdn_wave_sync();
        // ---- phase 2: 64 samples of the recirculating network, one lane per delay line.  Ring reads and inputs of 8
        // frames are fetched from LDS ahead of the serial recurrence (they do not depend on it); the only cross-lane
        // traffic on the per-sample critical path is the 5-stage Hadamard.
#ifndef FD_FDN_SKIP_P2
        if (line) {
            const float* trow = tile + lane * TS;
            const float* irow = tin + (j * 2 + (k & 1)) * 64;  // MultiSplit<U2,U16>: channel i takes input i % 2 (audionode.rs:600)
            const bool upper16 = (k & 16) != 0;
            for (int n0 = 0; n0 < size; n0 += 8) {
                float dd[8], xi[8], xo[8], oo[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    dd[u] = trow[n0 + u];
                    xi[u] = irow[n0 + u];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const float x = xi[u] + fb;   // Feedback::tick: input + value
                    xo[u] = x;                    // Delay::tick: the new sample takes the slot read this step
                    const float v0 = v1;          // Fir<U3>::tick fir.rs:57-70
                    v1 = v2;
                    v2 = dd[u];
                    float o = 0.0f;
                    o += w0 * v0;
                    o += w1 * v1;
                    o += w2 * v2;
                    oo[u] = o;
                    float h = o;                  // FrameHadamard feedback.rs:35-57, stages h = 1, 2, 4, 8, 16
                    // lower lane of a pair: x + y (own + partner); upper lane: x - y (partner - own)
                    h = xor_lane<1>(h, upper16) + u2f(f2u(h) ^ negmask[0]);
                    h = xor_lane<2>(h, upper16) + u2f(f2u(h) ^ negmask[1]);
                    h = xor_lane<4>(h, upper16) + u2f(f2u(h) ^ negmask[2]);
                    h = xor_lane<8>(h, upper16) + u2f(f2u(h) ^ negmask[3]);
                    h = xor_lane<16>(h, upper16) + u2f(f2u(h) ^ negmask[4]);
                    const float fbn = h * scale;
                    fb = (n0 + u < size) ? fbn : fb;  // ragged tail of the last block: frames past `size` change nothing
                    if (n0 + u >= size) { v2 = v1; v1 = v0; }  // (undo the shift)
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    tile[lane * TS + n0 + u] = xo[u];
                    tileo[lane * TS + n0 + u] = oo[u];
                }
            }
        }
#endif
        fdn_wave_sync();
        // ---- phase 3: write the 64 new ring samples per line back (coalesced), ordered pan sum with lane = frame
        for (int r0 = 0; r0 < R; r0 += 32) {
            float xw[32];
#pragma unroll
            for (int u = 0; u < 32; u++) xw[u] = tile[(r0 + u) * TS + lane];
#pragma unroll
            for (int u = 0; u < 32; u++) {
                const int r = r0 + u, kk = u;
                const size_t ri = inst0 + (r0 >> 5);
                const int w0 = __builtin_amdgcn_readlane(wp, r);
                const int pos = (w0 + lane) & cmask;
#ifndef FD_FDN_SKIP_STORE
                if (ri < V && lane < size) {
                    float* ring = s.rings + ri * c.ring_stride + (size_t)kk * cp;
                    ring[pos] = xw[u];
                    if (pos < 64) ring[c.cap + pos] = xw[u];  // the mirror of the first 64 slots
                }
#endif
            }
        }
#pragma unroll
        for (int jj = 0; jj < IPW; jj++) {
            const size_t ri = inst0 + jj;
            float l = 0.0f, rr = 0.0f;
            for (int kk = 0; kk < 32; kk++) {  // Reduce::tick left fold (audionode.rs:2427-2439) of Panner outputs
                const float o = tileo[(jj * 32 + kk) * TS + lane];
                const float pl = c.wl[kk] * o, pr = c.wr[kk] * o;
                l = kk == 0 ? pl : l + pl;
                rr = kk == 0 ? pr : rr + pr;
            }
            l *= (float)(1.0 / 16.0);  // * dc((1/16, 1/16))
            rr *= (float)(1.0 / 16.0);
            if (bus.mode) {  // wet * reverb [& dry * multipass()] (fd_fdn.hpp FdnBus): the block's input frames are still in their tile
                l = fdn_bus(bus, l, tin[(jj * 2 + 0) * 64 + lane]);
                rr = fdn_bus(bus, rr, tin[(jj * 2 + 1) * 64 + lane]);
            }
            if (ri < V && lane < size) {
                if (layout == 0) {
                    out[((size_t)0 * T + t0 + lane) * V + ri] = l;
                    out[((size_t)1 * T + t0 + lane) * V + ri] = rr;
                } else {
                    out[(ri * 2 + 0) * fstride + t0 + lane] = l;
                    out[(ri * 2 + 1) * fstride + t0 + lane] = rr;
                }
            }
        }
        wp = (wp + size) & cmask;
        fdn_wave_sync();
        if (more) stage();  // the tiles are free again: land the prefetched rows of the next block
    }
    if (valid) {
        if (k == 0) s.wpos[inst] = wp;
        s.v1[sidx] = v1;
        s.v2[sidx] = v2;
        s.fb[sidx] = fb;
    }
}

// ---- lane = FRAME formulation ------------------------------------------------------------------------------------
// Inside one 64-frame block nothing depends on the feedback: every delay is longer than 128 samples, so the 64 ring
// reads of a line -- and with them the FIR outputs, the Hadamard and the new feedback values -- are known up front; the
// only frame-to-frame coupling is `x[n] = in[n] + fb[n-1]` (the ring write) and the FIR's two-sample history.  So one
// wave renders one instance with lane = frame: the 32 lines live in 32 registers, the Hadamard is 5 x 32 register
// butterflies (no DPP, no permlane), ring rows are loaded and stored directly in the lane = frame order HBM wants
// (no LDS transposes), the pan sum is a register fold, and the one-frame shifts go through two small LDS rows per line.
// ~610 VALU + ~160 LDS instructions per instance-block instead of ~1660 + ~350; no LDS-bound occupancy limit.
// (non-temporal ring loads / stores measured 10.2 ms vs 6.4-6.9 ms: the L2 write-combining matters)
constexpr int HS = 68;  // floats per history row: [0..1] carry-in, [2..65] this block (also used with offset 1 for fb)

// CAP_LOG2: the ring capacity C = 1 << CAP_LOG2 is a compile-time constant here, so the 32 ring bases inside an instance
// are immediates and a block's ring addresses come out of a handful of scalar instructions:
//   read  of line k, frames 0..63:  rings[k * (C + 64) + ((w - D_k) & (C - 1)) + lane]      (the mirror zone: never wraps)
//   write of line k, frames 0..63:  rings[k * (C + 64) + w + lane]                           (when w + 64 <= C)
// Blocks whose write window wraps (one in C / 64), or touches the first 64 slots (mirror copy), or is ragged (the last
// block of a launch that is not a multiple of 64 frames) take the general per-lane path.
// NSEC = 2 (reverb4_stereo): the 32 lines are TWO 16-line networks in series -- lines 0-15 take the input, their FIR outputs are averaged to
// two channels (MultiJoin<U2, U8>) and split again (MultiSplit<U2, U8>) into lines 16-31, whose outputs are panned.  Nothing else changes: the
// ring reads, FIR outputs and Hadamard feedback of BOTH networks are known at the head of the block (every delay is longer than 128
// samples), the butterflies simply stop at stride 8, and the second network's ring write of frame n adds the join of the first one's FIR
// outputs of the same frame.
template <int CAP_LOG2, int NSEC>
__global__ __launch_bounds__(256) void k_fdn_render_frames(FdnConst c, FdnState s, size_t V, const float* __restrict__ in,
                                                           float* __restrict__ out, size_t T, size_t fstride, int layout, int tick_mode, FdnBus bus) {
    constexpr int C = 1 << CAP_LOG2, CMASK = C - 1, CP = C + 64;
    constexpr int NL = 32 / NSEC;  // lines per network
    __shared__ float hist_all[4][32 * HS];  // per line: delay outputs d[n-2], d[n-1] | d[0..63]
    __shared__ float fbr_all[4][32 * HS];   // per line: fb[-1] | fb[0..63]
    __shared__ float sc_wl[32], sc_wr[32];  // pan weights: an LDS broadcast read per use (128 kernel-argument scalars
    if (threadIdx.x < 32) {                 // next to the 32 delays do not fit the SGPR file)
        sc_wl[threadIdx.x] = c.wl[threadIdx.x];
        sc_wr[threadIdx.x] = c.wr[threadIdx.x];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* hist = hist_all[wib];
    float* fbr = fbr_all[wib];
    const size_t inst = (size_t)blockIdx.x * 4 + wib;
    if (inst >= V) return;
    const float w0 = c.w[0], w1 = c.w[1], w2 = c.w[2];
    const float scale = c.had_scale;  // (1.0 / sqrt(N as f64)) as f32  feedback.rs:57
    // The instance's rings as a buffer resource: buffer_load / buffer_store take a VGPR offset (lane * 4, the same for
    // every access), an SGPR offset (the line's base + the block's slot, scalar arithmetic) and no 64-bit VALU address math.
    const __amdgpu_buffer_rsrc_t rings = __builtin_amdgcn_make_buffer_rsrc(s.rings + inst * c.ring_stride, 0, (int)(c.ring_stride * sizeof(float)), 0x00020000);
    const int lane4 = lane * 4;
    int wp = __builtin_amdgcn_readfirstlane(s.wpos[inst]);  // write position of the block's first frame: wave-uniform
    if (lane < 32) {  // carry-in: Fir::v[1], v[2] and Feedback::value of every line
        hist[lane * HS + 0] = s.v1[inst * 32 + lane];
        hist[lane * HS + 1] = s.v2[inst * 32 + lane];
        fbr[lane * HS + 0] = s.fb[inst * 32 + lane];
    }
    float dn[32], xin[2];  // prefetched ring reads / inputs of the NEXT block (lane = frame)
    auto fetch = [&](size_t t0n, int wpn) {
        const int sizen = (int)((T - t0n) < 64 ? (T - t0n) : 64);
        // frame 0 of the block reads the slot written len - 1 frames ago; the 64 slots from there on are contiguous (mirror
        // zone), in bounds for every lane, and lanes past a ragged end read values nobody uses
#pragma unroll
        for (int k = 0; k < 32; k++) {
            const int r = (wpn - (c.len[k] - 1)) & CMASK;
            dn[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rings, lane4, (k * CP + r) * 4, 0));
        }
        if (sizen == 64) {
#pragma unroll
            for (int ch = 0; ch < 2; ch++)
                xin[ch] = layout == 0 ? in[((size_t)ch * T + t0n + lane) * V + inst] : (in + (inst * 2 + ch) * fstride + t0n)[lane];
        } else {
#pragma unroll
            for (int ch = 0; ch < 2; ch++)
                xin[ch] = lane < sizen ? (layout == 0 ? in[((size_t)ch * T + t0n + lane) * V + inst] : in[(inst * 2 + ch) * fstride + t0n + lane]) : 0.0f;
        }
    };
    fetch(0, wp);
    for (size_t t0 = 0; t0 < T; t0 += 64) {
        const int size = (int)((T - t0) < 64 ? (T - t0) : 64);
        float d[32], xi0 = xin[0], xi1 = xin[1];
#pragma unroll
        for (int k = 0; k < 32; k++) d[k] = dn[k];
        if (t0 + 64 < T) fetch(t0 + 64, (wp + 64) & CMASK);  // loads of the next block fly during this block's arithmetic
        // delay outputs -> history rows (lane n writes slot n + 2), then the FIR reads slots n, n + 1 (fir.rs:57-70)
#pragma unroll
        for (int k = 0; k < 32; k++) hist[k * HS + 2 + lane] = d[k];
        fdn_wave_sync();
        float o[32], h[32];
#pragma unroll
        for (int k = 0; k < 32; k++) {
            const float v0 = hist[k * HS + lane], v1 = hist[k * HS + lane + 1];
            float acc = 0.0f;
            acc += w0 * v0;
            acc += w1 * v1;
            acc += w2 * d[k];
            o[k] = acc;
            h[k] = acc;
        }
        // FrameHadamard feedback.rs:35-57: in-place butterflies h = 1, 2, 4, 8 (, 16); (x, y) -> (x + y, x - y), within each network
#pragma unroll
        for (int st = 1; st < NL; st <<= 1)
#pragma unroll
            for (int i = 0; i < 32; i++)
                if ((i & st) == 0) {
                    const float x = h[i], y = h[i + st];
                    h[i] = x + y;
                    h[i + st] = x - y;
                }
        // feedback of frame n -> row slot n + 1; the ring write of frame n needs slot n (Feedback::tick: input + value)
#pragma unroll
        for (int k = 0; k < 32; k++) fbr[k * HS + 1 + lane] = h[k] * scale;
        fdn_wave_sync();
        float xw[32];  // MultiSplit<U2, N/2>: line k takes input channel k % 2 (audionode.rs:600); Feedback::tick: input + value
#pragma unroll
        for (int k = 0; k < NL; k++) xw[k] = ((k & 1) ? xi1 : xi0) + fbr[k * HS + lane];
        if constexpr (NSEC == 2) {
            // MultiJoin<U2, U8> of the first network's 16 outputs -> 2 channels -> MultiSplit<U2, U8> into the second network.
            // process(): every term scaled by z = 1/8, then added (audionode.rs:706-720); tick(): the sum, divided by 8 (:697-705)
            float j0, j1;
            if (tick_mode) {
                j0 = o[0]; j1 = o[1];
#pragma unroll
                for (int i = 1; i < 8; i++) { j0 += o[2 * i]; j1 += o[2 * i + 1]; }
                j0 = j0 / 8.0f; j1 = j1 / 8.0f;
            } else {
                const float z = 1.0f / 8.0f;
                j0 = o[0] * z; j1 = o[1] * z;
#pragma unroll
                for (int i = 1; i < 8; i++) { j0 += o[2 * i] * z; j1 += o[2 * i + 1] * z; }
            }
#pragma unroll
            for (int k = NL; k < 32; k++) xw[k] = ((k & 1) ? j1 : j0) + fbr[k * HS + lane];
        }
        if (size == 64 && wp >= 64 && wp + 64 <= C) {  // the common block: one scalar offset per line
#pragma unroll
            for (int k = 0; k < 32; k++)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, xw[k]), rings, lane4, (k * CP + wp) * 4, 0);
        } else if (lane < size) {  // wrap, mirror zone or ragged tail: per-lane slots, Delay::tick slot by slot
            const int pos = (wp + lane) & CMASK;
#pragma unroll
            for (int k = 0; k < 32; k++) {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, xw[k]), rings, pos * 4, k * CP * 4, 0);
                if (pos < 64)  // keep the mirror of the first 64 slots
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, xw[k]), rings, (C + pos) * 4, k * CP * 4, 0);
            }
        }
        float l = 0.0f, rr = 0.0f;  // Reduce::tick left fold (audionode.rs:2427-2439) of the Panner outputs of the LAST network's lines
#pragma unroll
        for (int k = 32 - NL; k < 32; k++) {
            const float pl = sc_wl[k] * o[k], pr = sc_wr[k] * o[k];
            l = k == 32 - NL ? pl : l + pl;
            rr = k == 32 - NL ? pr : rr + pr;
        }
        l *= c.out_scale;  // * dc((1/16, 1/16)) | * dc((1/4, 1/4))
        rr *= c.out_scale;
        if (bus.mode) {  // wet * reverb [& dry * multipass()] (fd_fdn.hpp FdnBus)
            l = fdn_bus(bus, l, xi0);
            rr = fdn_bus(bus, rr, xi1);
        }
        if (lane < size) {
            if (layout == 0) {
                out[((size_t)0 * T + t0 + lane) * V + inst] = l;
                out[((size_t)1 * T + t0 + lane) * V + inst] = rr;
            } else {
                (out + (inst * 2 + 0) * fstride + t0)[lane] = l;
                (out + (inst * 2 + 1) * fstride + t0)[lane] = rr;
            }
        }
        fdn_wave_sync();
        // carry the last two delay outputs and the last feedback value of this block into slots 0, 1 / 0
        if (lane < 32) {
            const float a = hist[lane * HS + size], b = hist[lane * HS + size + 1], f = fbr[lane * HS + size];
            hist[lane * HS + 0] = a;
            hist[lane * HS + 1] = b;
            fbr[lane * HS + 0] = f;
        }
        wp = (wp + size) & CMASK;
        fdn_wave_sync();
    }
    if (lane == 0) s.wpos[inst] = wp;
    if (lane < 32) {
        s.v1[inst * 32 + lane] = hist[lane * HS + 0];
        s.v2[inst * 32 + lane] = hist[lane * HS + 1];
        s.fb[inst * 32 + lane] = fbr[lane * HS + 0];
    }
}


// ---- the generic network, lane = FRAME -----------------------------------------------------------------------------------------------
// `split::<N>() / multisplit::<M, N/M>() >> fdn::<N, _>(stacki(|i| delay(t_i) >> fir(w))) >> join::<N>() / multijoin::<M, N/M>()`: the Hadamard
// feedback delay network as the prelude documents it (prelude.rs:1323-1345, the "Mono Reverb" example :1334), N = NL lines of any delays
// longer than two blocks, FIR order K.  The same formulation as k_fdn_render_frames -- one wave per instance, lane = frame, the lines in
// registers, ring rows loaded one block ahead and stored as 256-byte runs -- with the ring capacity a run-time value (a handful of scalar
// instructions per line and block instead of immediates), no pan fold, and the splitter / joiner the graph names:
//   in : line k takes input channel k % nin   (Split :559-562, MultiSplit :600-606: output i = input i % M)
//   out: channel j = average of lines j, j + nout, ..  (Join / MultiJoin: process scales every term by 1 / n and adds :649-659, :710-724;
//        tick adds and divides :643-648, :700-708)
// Run-time compiled graphs of this shape render lane-per-voice at ~25 GB/s (every lane walks its N lines one after the other, 32 waves on
// the chip for 2 048 instances); this kernel is what Bank.from_graph / fdsp_fdn_create give them instead.
template <int NL, int K>
__global__ __launch_bounds__(256) void k_fdn_frames_generic(FdnConst c, FdnState s, size_t V, const float* __restrict__ in,
                                                            float* __restrict__ out, size_t T, size_t fstride, int layout, int tick_mode, FdnBus bus) {
    static_assert(K >= 1 && K <= 3 && NL >= 2 && NL <= 32 && (NL & (NL - 1)) == 0, "generic FDN: 2..32 lines (a power of two), FIR order 1..3");
    constexpr int H0 = K - 1;               // carried delay outputs per line: Fir::v[1 .. K-1]
    __shared__ float hist_all[4][NL * HS];  // per line: the H0 carried delay outputs | d[0..63]
    __shared__ float fbr_all[4][NL * HS];   // per line: fb[-1] | fb[0..63]
    const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* hist = hist_all[wib];
    float* fbr = fbr_all[wib];
    const size_t inst = (size_t)blockIdx.x * 4 + wib;
    if (inst >= V) return;
    const float w0 = c.w[0], w1 = c.w[1], w2 = c.w[2];
    const float scale = c.had_scale;
    const int nin = c.nin, nout = c.nout;   // 1 or 2
    const int CMASK = c.cap - 1, CP = c.cap + 64;
    const __amdgpu_buffer_rsrc_t rings = __builtin_amdgcn_make_buffer_rsrc(s.rings + inst * c.ring_stride, 0, (int)(c.ring_stride * sizeof(float)), 0x00020000);
    const int lane4 = lane * 4;
    int wp = __builtin_amdgcn_readfirstlane(s.wpos[inst]);
    if (lane < NL) {  // carry-in: Fir::v[1..K-1] (v1 = the older, v2 = the newer of a Fir<U3>; a Fir<U2> keeps its one sample in v2), Feedback::value
        if (K == 3) hist[lane * HS + 0] = s.v1[inst * 32 + lane];
        if (K >= 2) hist[lane * HS + H0 - 1] = s.v2[inst * 32 + lane];
        fbr[lane * HS + 0] = s.fb[inst * 32 + lane];
    }
    float dn[NL], xin[2];
    auto fetch = [&](size_t t0n, int wpn) {
        const int sizen = (int)((T - t0n) < 64 ? (T - t0n) : 64);
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int r = (wpn - (c.len[k] - 1)) & CMASK;   // frame 0 reads the slot written len - 1 frames ago; 64 contiguous slots from there (mirror zone)
            dn[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rings, lane4, (k * CP + r) * 4, 0));
        }
#pragma unroll
        for (int ch = 0; ch < 2; ch++)
            xin[ch] = (ch < nin && lane < sizen) ? (layout == 0 ? in[((size_t)ch * T + t0n + lane) * V + inst] : in[(inst * nin + ch) * fstride + t0n + lane]) : 0.0f;
    };
    fetch(0, wp);
    for (size_t t0 = 0; t0 < T; t0 += 64) {
        const int size = (int)((T - t0) < 64 ? (T - t0) : 64);
        float d[NL];
        const float xi0 = xin[0], xi1 = nin == 2 ? xin[1] : xin[0];
        const float xin_now[2] = {xin[0], xin[1]};   // (this block's input frames, by channel: the bus reads them at the end of the block)
#pragma unroll
        for (int k = 0; k < NL; k++) d[k] = dn[k];
        if (t0 + 64 < T) fetch(t0 + 64, (wp + 64) & CMASK);
        if (K > 1) {
#pragma unroll
            for (int k = 0; k < NL; k++) hist[k * HS + H0 + lane] = d[k];
            fdn_wave_sync();
        }
        float o[NL], h[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) {  // Fir::tick fir.rs:57-70: the sum starts at 0.0 and takes the taps oldest first
            float acc = 0.0f;
            if (K == 3) {
                acc += w0 * hist[k * HS + lane];
                acc += w1 * hist[k * HS + lane + 1];
                acc += w2 * d[k];
            } else if (K == 2) {
                acc += w0 * hist[k * HS + lane];
                acc += w1 * d[k];
            } else {
                acc += w0 * d[k];
            }
            o[k] = acc;
            h[k] = acc;
        }
#pragma unroll
        for (int st = 1; st < NL; st <<= 1)  // FrameHadamard feedback.rs:35-57
#pragma unroll
            for (int i = 0; i < NL; i++)
                if ((i & st) == 0) {
                    const float x = h[i], y = h[i + st];
                    h[i] = x + y;
                    h[i + st] = x - y;
                }
#pragma unroll
        for (int k = 0; k < NL; k++) fbr[k * HS + 1 + lane] = h[k] * scale;
        fdn_wave_sync();
        float xw[NL];  // Feedback::tick: input + value (feedback.rs:130-134)
#pragma unroll
        for (int k = 0; k < NL; k++) xw[k] = ((k & 1) ? xi1 : xi0) + fbr[k * HS + lane];
        if (size == 64 && wp >= 64 && wp + 64 <= c.cap) {
#pragma unroll
            for (int k = 0; k < NL; k++)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, xw[k]), rings, lane4, (k * CP + wp) * 4, 0);
        } else if (lane < size) {  // wrap, mirror zone or ragged tail
            const int pos = (wp + lane) & CMASK;
#pragma unroll
            for (int k = 0; k < NL; k++) {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, xw[k]), rings, pos * 4, k * CP * 4, 0);
                if (pos < 64) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, xw[k]), rings, (c.cap + pos) * 4, k * CP * 4, 0);
            }
        }
        // Join<N> / MultiJoin<M, N/M>: channel j over lines j, j + nout, ..
        float y0, y1 = 0.0f;
        if (nout == 1) {
            if (tick_mode) {
                y0 = o[0];
#pragma unroll
                for (int i = 1; i < NL; i++) y0 += o[i];
                y0 = y0 / (float)NL;
            } else {
                const float z = 1.0f / (float)NL;
                y0 = o[0] * z;
#pragma unroll
                for (int i = 1; i < NL; i++) y0 += o[i] * z;
            }
        } else {
            if (tick_mode) {
                y0 = o[0]; y1 = o[1];
#pragma unroll
                for (int i = 1; i < NL / 2; i++) { y0 += o[2 * i]; y1 += o[2 * i + 1]; }
                y0 = y0 / (float)(NL / 2); y1 = y1 / (float)(NL / 2);
            } else {
                const float z = 1.0f / (float)(NL / 2);
                y0 = o[0] * z; y1 = o[1] * z;
#pragma unroll
                for (int i = 1; i < NL / 2; i++) { y0 += o[2 * i] * z; y1 += o[2 * i + 1] * z; }
            }
        }
        if (bus.mode) {  // wet * network [& dry * multipass()] (fd_fdn.hpp FdnBus; mode 2: nin == nout)
            y0 = fdn_bus(bus, y0, xin_now[0]);
            y1 = fdn_bus(bus, y1, xin_now[1]);
        }
        if (lane < size) {
            if (layout == 0) {
                out[((size_t)0 * T + t0 + lane) * V + inst] = y0;
                if (nout == 2) out[((size_t)1 * T + t0 + lane) * V + inst] = y1;
            } else {
                out[(inst * nout + 0) * fstride + t0 + lane] = y0;
                if (nout == 2) out[(inst * nout + 1) * fstride + t0 + lane] = y1;
            }
        }
        fdn_wave_sync();
        if (lane < NL) {  // the block's last delay outputs and feedback value become the next block's carry-in
            float a = 0.0f, b = 0.0f;
            if (K == 3) { a = hist[lane * HS + size]; b = hist[lane * HS + size + 1]; }
            if (K == 2) b = hist[lane * HS + size];
            const float f = fbr[lane * HS + size];
            if (K == 3) hist[lane * HS + 0] = a;
            if (K >= 2) hist[lane * HS + H0 - 1] = b;
            fbr[lane * HS + 0] = f;
        }
        wp = (wp + size) & CMASK;
        fdn_wave_sync();
    }
    if (lane == 0) s.wpos[inst] = wp;
    if (lane < NL) {
        if (K == 3) s.v1[inst * 32 + lane] = hist[lane * HS + 0];
        if (K >= 2) s.v2[inst * 32 + lane] = hist[lane * HS + H0 - 1];
        s.fb[inst * 32 + lane] = fbr[lane * HS + 0];
    }
}

template <int NL>
static void fdn_launch_generic(const FdnConst& c, const FdnState& s, size_t instances, const float* in, float* out, size_t T, size_t fstride, int layout,
                               int tick_mode, hipStream_t stream, const FdnBus& bus) {
    const dim3 grid((unsigned)((instances + 3) / 4)), block(256);
    if (c.taps == 3) hipLaunchKernelGGL((k_fdn_frames_generic<NL, 3>), grid, block, 0, stream, c, s, instances, in, out, T, fstride, layout, tick_mode, bus);
    else if (c.taps == 2) hipLaunchKernelGGL((k_fdn_frames_generic<NL, 2>), grid, block, 0, stream, c, s, instances, in, out, T, fstride, layout, tick_mode, bus);
    else hipLaunchKernelGGL((k_fdn_frames_generic<NL, 1>), grid, block, 0, stream, c, s, instances, in, out, T, fstride, layout, tick_mode, bus);
}

// ---- voice-minor I/O for the lane = frame kernels -----------------------------------------------------------------------------------
// A lane = frame kernel reads its inputs and writes its outputs 64 consecutive FRAMES of one instance at a time: 256-byte runs in the planar
// layout [instance][channel][frame], but 64 separate lines in the engine's default voice-minor layout [channel][frame][instance] (a 32-line
// reverb on 2 048 instances: 10.3 ms against 5.3).  Banks of 64 instances or more therefore take voice-minor buffers through a staging copy:
// one tiled transpose in, the planar render, one tiled transpose out -- (inputs + outputs) x 8 bytes per instance-frame of extra traffic at
// copy speed instead of a gather per frame.
__global__ __launch_bounds__(256) void k_fdn_transpose(const float* __restrict__ src, float* __restrict__ dst, size_t V, size_t T, int C, int to_planar) {
    // to_planar: src [C][T][V] -> dst [V][C][T]; else src [V][C][T] -> dst [C][T][V].  One 64 x 64 (frame x instance) tile per workgroup.
    __shared__ float tile[64][65];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t v0 = (size_t)blockIdx.x * 64, t0 = (size_t)blockIdx.y * 64;
    const int c = blockIdx.z;
    if (to_planar) {
#pragma unroll 4
        for (int r = w; r < 64; r += 4)
            tile[r][lane] = (t0 + r < T && v0 + lane < V) ? src[((size_t)c * T + t0 + r) * V + v0 + lane] : 0.0f;
        __syncthreads();
#pragma unroll 4
        for (int r = w; r < 64; r += 4)
            if (v0 + r < V && t0 + lane < T) dst[((v0 + r) * C + c) * T + t0 + lane] = tile[lane][r];
    } else {
#pragma unroll 4
        for (int r = w; r < 64; r += 4)
            tile[r][lane] = (v0 + r < V && t0 + lane < T) ? src[((v0 + r) * C + c) * T + t0 + lane] : 0.0f;
        __syncthreads();
#pragma unroll 4
        for (int r = w; r < 64; r += 4)
            if (t0 + r < T && v0 + lane < V) dst[((size_t)c * T + t0 + r) * V + v0 + lane] = tile[lane][r];
    }
}

void fdn_launch_transpose(const float* src, float* dst, size_t V, size_t T, int channels, bool to_planar, hipStream_t stream) {
    if (V == 0 || T == 0 || channels == 0) return;
    hipLaunchKernelGGL(k_fdn_transpose, dim3((unsigned)((V + 63) / 64), (unsigned)((T + 63) / 64), (unsigned)channels), dim3(256), 0, stream, src, dst, V, T,
                       channels, to_planar ? 1 : 0);
}

void fdn_launch_reset(const FdnConst& c, const FdnState& s, size_t instances, hipStream_t stream) {
    hipLaunchKernelGGL(k_fdn_reset, dim3(2048), dim3(256), 0, stream, c, s, instances);
}

void fdn_launch_render(const FdnConst& c, const FdnState& s, size_t instances, const float* in, float* out, size_t T,
                       size_t fstride, int layout, int tick_mode, hipStream_t stream, const FdnBus& bus) {
    if (instances == 0 || T == 0) return;
    if (c.generic) {
        tl_opts.last_kernel = LK_FDN_FRAMES;
        switch (c.lines) {
        case 2: return fdn_launch_generic<2>(c, s, instances, in, out, T, fstride, layout, tick_mode, stream, bus);
        case 4: return fdn_launch_generic<4>(c, s, instances, in, out, T, fstride, layout, tick_mode, stream, bus);
        case 8: return fdn_launch_generic<8>(c, s, instances, in, out, T, fstride, layout, tick_mode, stream, bus);
        case 16: return fdn_launch_generic<16>(c, s, instances, in, out, T, fstride, layout, tick_mode, stream, bus);
        default: return fdn_launch_generic<32>(c, s, instances, in, out, T, fstride, layout, tick_mode, stream, bus);
        }
    }
    const bool frames = tl_opts.fdn_kernel == 0 || c.sections == 2;  // (two networks in series: the lane = frame formulation only)
    tl_opts.last_kernel = frames ? LK_FDN_FRAMES : LK_FDN_LINES;
    if (frames) {
        const dim3 grid((unsigned)((instances + 3) / 4)), block(256);
        switch (c.cap) {  // the ring capacity is a template parameter of the lane = frame kernel
#define FD_FDN_CASE(L)                                                                                                                                     \
    case 1 << L:                                                                                                                                           \
        if (c.sections == 2) hipLaunchKernelGGL((k_fdn_render_frames<L, 2>), grid, block, 0, stream, c, s, instances, in, out, T, fstride, layout, tick_mode, bus); \
        else hipLaunchKernelGGL((k_fdn_render_frames<L, 1>), grid, block, 0, stream, c, s, instances, in, out, T, fstride, layout, tick_mode, bus);                 \
        break;
            FD_FDN_CASE(8) FD_FDN_CASE(9) FD_FDN_CASE(10) FD_FDN_CASE(11) FD_FDN_CASE(12) FD_FDN_CASE(13) FD_FDN_CASE(14)
            FD_FDN_CASE(15) FD_FDN_CASE(16) FD_FDN_CASE(17) FD_FDN_CASE(18)
#undef FD_FDN_CASE
        default:  // longer than 2^18 slots (5.4 s at 48 kHz): the lane = line kernel takes any capacity (reverb_stereo; fdn_configure
                  // refuses such a reverb4_stereo)
            hipLaunchKernelGGL(k_fdn_render<1>, grid, block, 0, stream, c, s, instances, in, out, T, fstride, layout, bus);
        }
    } else if ((instances + 1) / 2 < 2 * (size_t)simd_count()) {
        // lane = line kernel: one instance per wave while that is what it takes to have two waves per SIMD
        const unsigned grid = (unsigned)((instances + 3) / 4);
        hipLaunchKernelGGL(k_fdn_render<1>, dim3(grid), dim3(256), 0, stream, c, s, instances, in, out, T, fstride, layout, bus);
    } else {
        const unsigned grid = (unsigned)((instances + 7) / 8);
        hipLaunchKernelGGL(k_fdn_render<2>, dim3(grid), dim3(256), 0, stream, c, s, instances, in, out, T, fstride, layout, bus);
    }
}

}  // namespace fd

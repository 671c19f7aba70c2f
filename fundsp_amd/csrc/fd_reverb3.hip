// fd_reverb3.hip -- reverb3_stereo's allpass loop, lane = frame (see fd_reverb3.hpp).  Compiled WITHOUT flush-to-zero: the reference's
// Reverb<F> is no Feedback node and keeps IEEE denormals.
#include <cmath>
#include <type_traits>

#include "fd_reverb3.hpp"
#include "fd_opts.hpp"
#include "fd_math.hpp"
#include "fd_nodes.hpp"   // svf_coefs, SvfCore (host + device)

namespace fd {

static const int RV3_LDELAYS[32] = {401, 421, 443, 463, 487, 503, 523, 547, 563, 587, 607, 619, 643, 661, 683, 701,   // reverb.rs:163-166
                                    727, 743, 761, 787, 809, 823, 839, 863, 883, 907, 929, 947, 967, 983, 1009, 1021};
static const int RV3_RDELAYS[32] = {419, 433, 457, 479, 491, 509, 541, 557, 577, 593, 613, 631, 653, 673, 691, 719,   // :167-170
                                    733, 757, 773, 797, 811, 829, 853, 877, 887, 911, 937, 953, 977, 997, 1013, 1033};
static const int RV3_BDELAYS[8] = {1087, 1091, 1093, 1097, 1103, 1109, 1117, 1123};                                   // :171
static const int RV3_PDELAYS[4] = {245, 367, 263, 349};                                                                // :198

bool rv3_make_const(double time, double diffusion, const Rv3Filter& filter, double sample_rate, Rv3Const* c) {
    *c = Rv3Const{};
    // Delay::new(t) at DEFAULT_SR, then set_sample_rate: time_in_samples = round(t * sample_rate) (delay.rs:105-112)
    auto samples = [&](int at_default) { return (int)std::round(((double)at_default / 44100.0) * sample_rate); };
    int longest = 0, shortest = 1 << 30;
    for (int i = 0; i < 4; i++) c->dpre[i] = RV3_PDELAYS[i];  // (n - 1) samples + the allpass's own tick; never re-sized (reverb.rs:226-238)
    for (int b = 0; b < 8; b++) {
        for (int j = 0; j < 4; j++) {
            c->dap0[b][j] = samples(RV3_LDELAYS[b + j * 8] - 1) + 1;   // :175-186
            c->dap1[b][j] = samples(RV3_RDELAYS[b + j * 8] - 1) + 1;
            longest = std::max(longest, std::max(c->dap0[b][j], c->dap1[b][j]));
            shortest = std::min(shortest, std::min(c->dap0[b][j], c->dap1[b][j]));
        }
        c->dblk[b] = samples(RV3_BDELAYS[7 - b]) + (b == 0 ? 1 : 0);    // :187
        longest = std::max(longest, c->dblk[b]);
        shortest = std::min(shortest, c->dblk[b]);
    }
    if (shortest <= 128 || longest > (1 << 20)) return false;
    int cap = 256;
    while (cap < longest + 64) cap <<= 1;
    c->cap = cap;
    c->ring_stride = (size_t)72 * ((size_t)cap + 64);
    // lerp(0.5, 0.9, diffusion) as f32 (:173; Lerp: a * (1 - t) + b * t in f64, math.rs:169-178)
    c->eta = (float)(0.5 * (1.0 - diffusion) + 0.9 * diffusion);
    // pow(db_amp(-60.0), 0.035 / time) as f32 (:196); db_amp(x) = exp((x / 20) * LN_10) (math.rs:76-78, 294)
    c->a = (float)std::pow(std::exp((-60.0 / 20.0) * 2.302585092994046), 0.035 / time);
    const float sr = (float)sample_rate;
    c->fkind = filter.kind;
    if (filter.kind == 0) {
        c->c = expf_musl(-F32_TAU * filter.cutoff / sr);   // Lowpole::set_cutoff, F = f32 (filter.rs:35-38)
        c->omc = 1.0f - c->c;
    } else {
        const SvfCoefs k = svf_coefs(filter.mode, sr, filter.cutoff, filter.q, filter.gain);   // FixedSvf::set_sample_rate -> update (svf.rs:989-992, 236-239)
        c->sa1 = k.a1; c->sa2 = k.a2; c->sa3 = k.a3; c->sm0 = k.m0; c->sm1 = k.m1; c->sm2 = k.m2;
    }
    return true;
}

__global__ __launch_bounds__(256) void k_rv3_zero(Rv3Const c, Rv3State s, size_t instances, int with_pre) {
    // Reverb::reset (reverb.rs:211-224): the loop blocks' lines, allpass z, filters and the feedback sample -- not `pre`
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    for (size_t i = gid; i < instances * c.ring_stride; i += step) s.rings[i] = 0.0f;
    for (size_t i = gid; i < instances * 32; i += step) s.fval[i] = 0.0f;
    for (size_t i = gid; i < instances; i += step) s.wpos[i] = 0;
    if (with_pre) {
        for (size_t i = gid; i < instances * 4 * (RV3_PRE_CAP + 64); i += step) s.pre[i] = 0.0f;
        for (size_t i = gid; i < instances; i += step) s.wpre[i] = 0;
    }
}

__global__ __launch_bounds__(64) void k_rv3_migrate(Rv3Const from, Rv3State sf, Rv3Const to, Rv3State st, size_t instances) {
    // one wave per instance, lane = allpass line (0 .. 63) | lane 0 also the feedback sample
    const size_t inst = blockIdx.x;
    if (inst >= instances) return;
    const int lane = threadIdx.x;
    const int b = lane >> 3, j = lane & 7, ring = b * 9 + j;
    const int dold = j < 4 ? from.dap0[b][j] : from.dap1[b][j - 4], dnew = j < 4 ? to.dap0[b][j] : to.dap1[b][j - 4];
    const int wold = sf.wpos[inst];
    const float* ro = sf.rings + inst * from.ring_stride;
    float* rn = st.rings + inst * to.ring_stride;
    const size_t cpo = (size_t)from.cap + 64, cpn = (size_t)to.cap + 64;
    // AllNest::z is not reset by set_sample_rate (delay.rs:312-319): the next tick still reads it, from a line that is empty otherwise
    rn[(size_t)ring * cpn + (size_t)((0 - dnew) & (to.cap - 1))] = ro[(size_t)ring * cpo + (size_t)((wold - dold) & (from.cap - 1))];
    if (lane == 0) {  // Reverb::feedback survives as well: it enters block 0's (empty) delay line on the next tick
        const float fb = ro[(size_t)8 * cpo + (size_t)((wold - 1) & (from.cap - 1))];
        const int slot = (0 - 1) & (to.cap - 1);
        rn[(size_t)8 * cpn + (size_t)slot] = fb;
    }
}

constexpr int RS = 68;  // floats per row of the filter hand-over tiles (16-byte aligned rows for the b128 reads of the serial lanes)

// One Schroeder allpass over the 64 frames of a block: z = the line's 64 reads of this block (prefetched), x in, y out, v stored to the line
// (AllNest::tick delay.rs:322-330: v = x - eta * z; y = eta * v + z).
#define RV3_AP(ZREG, RSRC, BASE, WIDE, POS, NEXT)                                                                                \
    {                                                                                                                            \
        const float z_ = ZREG;                                                                                                   \
        const float v_ = x - eta * z_;                                                                                           \
        x = eta * v_ + z_;                                                                                                       \
        if (WIDE) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v_), RSRC, lane4, ((BASE) + wpw) * 4, 0); \
        else if (lane < size) {                                                                                                  \
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v_), RSRC, (POS) * 4, (BASE) * 4, 0);            \
            if ((POS) < 64) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v_), RSRC, (capw + (POS)) * 4, (BASE) * 4, 0); \
        }                                                                                                                        \
        ZREG = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(RSRC, lane4, ((BASE) + (NEXT)) * 4, 0));           \
    }

template <int FK>
__global__ __launch_bounds__(256) void k_rv3_render(Rv3Const c, Rv3State s, size_t V, const float* __restrict__ in, float* __restrict__ out,
                                                    size_t T, size_t fstride, int layout, FdnBus bus) {
    __shared__ float tile_all[4][8 * RS];
    const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* tile = tile_all[wib];
    const size_t inst = (size_t)blockIdx.x * 4 + wib;
    if (inst >= V) return;
    const float eta = c.eta, a = c.a, fc = c.c, omc = c.omc;
    const int CMASK = c.cap - 1, CP = c.cap + 64;
    constexpr int PMASK = RV3_PRE_CAP - 1, PCP = RV3_PRE_CAP + 64;
    const __amdgpu_buffer_rsrc_t rings = __builtin_amdgcn_make_buffer_rsrc(s.rings + inst * c.ring_stride, 0, (int)(c.ring_stride * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t pres = __builtin_amdgcn_make_buffer_rsrc(s.pre + inst * 4 * PCP, 0, (int)(4 * PCP * sizeof(float)), 0x00020000);
    const int lane4 = lane * 4;
    int wp = __builtin_amdgcn_readfirstlane(s.wpos[inst]), wq = __builtin_amdgcn_readfirstlane(s.wpre[inst]);
    // filter state of the serial lanes 0-7 (lane = loop block): layer 0 = filter0, layer 1 = filter1; a FixedSvf has two words per filter
    float val0 = lane < 8 ? s.fval[inst * 32 + lane] : 0.0f, val1 = lane < 8 ? s.fval[inst * 32 + 8 + lane] : 0.0f;
    float vb0 = (FK == 1 && lane < 8) ? s.fval[inst * 32 + 16 + lane] : 0.0f, vb1 = (FK == 1 && lane < 8) ? s.fval[inst * 32 + 24 + lane] : 0.0f;
    SvfCore svf;
    svf.a1 = c.sa1; svf.a2 = c.sa2; svf.a3 = c.sa3; svf.m0 = c.sm0; svf.m1 = c.sm1; svf.m2 = c.sm2; svf.ic1eq = 0.0f; svf.ic2eq = 0.0f;
    // The 76 lines' reads of a block (lane = frame) sit in 76 registers; each is consumed in place and the NEXT block's read of the same line is
    // issued into the same register right behind it (its slots lie before this block's write window: every distance exceeds 128), so a load has a
    // whole block of arithmetic to land
    float zp[4], z0[8][4], z1[8][4], zd[8], xin[2];
#pragma unroll
    for (int i = 0; i < 4; i++)
        zp[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pres, lane4, (i * PCP + ((wq - c.dpre[i]) & PMASK)) * 4, 0));
#pragma unroll
    for (int b = 0; b < 8; b++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            z0[b][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rings, lane4, ((b * 9 + j) * CP + ((wp - c.dap0[b][j]) & CMASK)) * 4, 0));
            z1[b][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rings, lane4, ((b * 9 + 4 + j) * CP + ((wp - c.dap1[b][j]) & CMASK)) * 4, 0));
        }
        zd[b] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rings, lane4, ((b * 9 + 8) * CP + ((wp - c.dblk[b]) & CMASK)) * 4, 0));
    }
    auto fetch_in = [&](size_t t0n) {
        const int sizen = (int)((T - t0n) < 64 ? (T - t0n) : 64);
#pragma unroll
        for (int ch = 0; ch < 2; ch++)
            xin[ch] = lane < sizen ? (layout == 0 ? in[((size_t)ch * T + t0n + lane) * V + inst] : in[(inst * 2 + ch) * fstride + t0n + lane]) : 0.0f;
    };
    fetch_in(0);
    for (size_t t0 = 0; t0 < T; t0 += 64) {
        const int size = (int)((T - t0) < 64 ? (T - t0) : 64);
        const float in0 = xin[0], in1 = xin[1];
        if (t0 + 64 < T) fetch_in(t0 + 64);
        const int wn = (wp + 64) & CMASK, wqn = (wq + 64) & PMASK;   // the next block's write positions (a ragged block is the last: its reloads go unused)
        // write windows: the common block stores 64 contiguous slots per line with one scalar offset; a window that wraps, touches the mirrored
        // first 64 slots, or a ragged last block goes slot by slot -- two compile-time variants of the block's body, chosen once per block
        auto block_body = [&](auto WIDE) {
        const int pos = (wp + lane) & CMASK, posp = (wq + lane) & PMASK;
            constexpr bool wide = decltype(WIDE)::value, widep = wide;
            float x, input0, input1;
            {   // the input diffusers: input0 = pre1(pre0(in0 * 0.5)), input1 = pre3(pre2(in1 * 0.5))   reverb.rs:245-248
                const int wpw = wq, capw = RV3_PRE_CAP;
                x = in0 * 0.5f;
                RV3_AP(zp[0], pres, 0 * PCP, widep, posp, (wqn - c.dpre[0]) & PMASK)
                RV3_AP(zp[1], pres, 1 * PCP, widep, posp, (wqn - c.dpre[1]) & PMASK)
                input0 = x;
                x = in1 * 0.5f;
                RV3_AP(zp[2], pres, 2 * PCP, widep, posp, (wqn - c.dpre[2]) & PMASK)
                RV3_AP(zp[3], pres, 3 * PCP, widep, posp, (wqn - c.dpre[3]) & PMASK)
                input1 = x;
            }
            const int wpw = wp, capw = c.cap;
            // layer 0 of every loop block: delay -> a * v + input0 -> allpass0[0..3], up to the filter's feed-forward product
#pragma unroll
            for (int b = 0; b < 8; b++) {
                x = a * zd[b] + input0;
                zd[b] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rings, lane4, ((b * 9 + 8) * CP + ((wn - c.dblk[b]) & CMASK)) * 4, 0));
                RV3_AP(z0[b][0], rings, (b * 9 + 0) * CP, wide, pos, (wn - c.dap0[b][0]) & CMASK)
                RV3_AP(z0[b][1], rings, (b * 9 + 1) * CP, wide, pos, (wn - c.dap0[b][1]) & CMASK)
                RV3_AP(z0[b][2], rings, (b * 9 + 2) * CP, wide, pos, (wn - c.dap0[b][2]) & CMASK)
                RV3_AP(z0[b][3], rings, (b * 9 + 3) * CP, wide, pos, (wn - c.dap0[b][3]) & CMASK)
                tile[b * RS + lane] = FK == 0 ? omc * x : x;   // Lowpole::tick: (1 - coeff) * x ...  | FixedSvf: the sample itself
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < 8) {   // ... + coeff * value, 64 steps in registers (filter.rs:64-66)
                float r[64];
#pragma unroll
                for (int n = 0; n < 64; n++) r[n] = tile[lane * RS + n];
                if (FK == 1) {   // FixedSvf::tick, the reference's operations in the reference's order (svf.rs:995-1006)
                    svf.ic1eq = val0;
                    svf.ic2eq = vb0;
                    if (size == 64) {
#pragma unroll
                        for (int n = 0; n < 64; n++) r[n] = svf.tick(r[n]);
                    } else {
#pragma unroll
                        for (int n = 0; n < 64; n++) {
                            const float i1 = svf.ic1eq, i2 = svf.ic2eq;
                            r[n] = svf.tick(r[n]);
                            if (n >= size) { svf.ic1eq = i1; svf.ic2eq = i2; }
                        }
                    }
                    val0 = svf.ic1eq;
                    vb0 = svf.ic2eq;
                } else if (size == 64) {
#pragma unroll
                    for (int n = 0; n < 64; n++) {
                        val0 = r[n] + fc * val0;
                        r[n] = val0;
                    }
                } else {   // a ragged last block: the filter stops where the launch does
#pragma unroll
                    for (int n = 0; n < 64; n++) {
                        const float nv = r[n] + fc * val0;
                        val0 = n < size ? nv : val0;
                        r[n] = nv;
                    }
                }
#pragma unroll
                for (int n = 0; n < 64; n++) tile[lane * RS + n] = r[n];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float f0[8];
#pragma unroll
            for (int b = 0; b < 8; b++) f0[b] = tile[b * RS + lane];
            float out0 = f0[7];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // layer 1: a * v + input1 -> allpass1[0..3] -> filter1
#pragma unroll
            for (int b = 0; b < 8; b++) {
                x = a * f0[b] + input1;
                RV3_AP(z1[b][0], rings, (b * 9 + 4) * CP, wide, pos, (wn - c.dap1[b][0]) & CMASK)
                RV3_AP(z1[b][1], rings, (b * 9 + 5) * CP, wide, pos, (wn - c.dap1[b][1]) & CMASK)
                RV3_AP(z1[b][2], rings, (b * 9 + 6) * CP, wide, pos, (wn - c.dap1[b][2]) & CMASK)
                RV3_AP(z1[b][3], rings, (b * 9 + 7) * CP, wide, pos, (wn - c.dap1[b][3]) & CMASK)
                tile[b * RS + lane] = FK == 0 ? omc * x : x;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < 8) {
                float r[64];
#pragma unroll
                for (int n = 0; n < 64; n++) r[n] = tile[lane * RS + n];
                if (FK == 1) {   // FixedSvf::tick, the reference's operations in the reference's order (svf.rs:995-1006)
                    svf.ic1eq = val1;
                    svf.ic2eq = vb1;
                    if (size == 64) {
#pragma unroll
                        for (int n = 0; n < 64; n++) r[n] = svf.tick(r[n]);
                    } else {
#pragma unroll
                        for (int n = 0; n < 64; n++) {
                            const float i1 = svf.ic1eq, i2 = svf.ic2eq;
                            r[n] = svf.tick(r[n]);
                            if (n >= size) { svf.ic1eq = i1; svf.ic2eq = i2; }
                        }
                    }
                    val1 = svf.ic1eq;
                    vb1 = svf.ic2eq;
                } else if (size == 64) {
#pragma unroll
                    for (int n = 0; n < 64; n++) {
                        val1 = r[n] + fc * val1;
                        r[n] = val1;
                    }
                } else {   // a ragged last block: the filter stops where the launch does
#pragma unroll
                    for (int n = 0; n < 64; n++) {
                        const float nv = r[n] + fc * val1;
                        val1 = n < size ? nv : val1;
                        r[n] = nv;
                    }
                }
#pragma unroll
                for (int n = 0; n < 64; n++) tile[lane * RS + n] = r[n];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // every block's output enters the NEXT block's delay line (the last block's the first's: Reverb::feedback)
            float out1 = 0.0f;
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const float f1 = tile[b * RS + lane];
                if (b == 7) out1 = f1;
                const int base = (((b + 1) & 7) * 9 + 8) * CP;
                if (wide) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, f1), rings, lane4, (base + wp) * 4, 0);
                else if (lane < size) {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, f1), rings, pos * 4, base * 4, 0);
                    if (pos < 64) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, f1), rings, (c.cap + pos) * 4, base * 4, 0);
                }
            }
            if (bus.mode) {  // wet * reverb3_stereo(..) [& dry * multipass()] (fd_fdn.hpp FdnBus)
                out0 = fdn_bus(bus, out0, in0);
                out1 = fdn_bus(bus, out1, in1);
            }
            if (lane < size) {
                if (layout == 0) {
                    out[((size_t)0 * T + t0 + lane) * V + inst] = out0;
                    out[((size_t)1 * T + t0 + lane) * V + inst] = out1;
                } else {
                    out[(inst * 2 + 0) * fstride + t0 + lane] = out0;
                    out[(inst * 2 + 1) * fstride + t0 + lane] = out1;
                }
            }
        };
        if (size == 64 && wp >= 64 && wp + 64 <= c.cap && wq >= 64 && wq + 64 <= RV3_PRE_CAP) block_body(std::true_type{});
        else block_body(std::false_type{});
        wp = (wp + size) & CMASK;
        wq = (wq + size) & PMASK;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (lane == 0) {
        s.wpos[inst] = wp;
        s.wpre[inst] = wq;
    }
    if (lane < 8) {
        s.fval[inst * 32 + lane] = val0;
        s.fval[inst * 32 + 8 + lane] = val1;
        if (FK == 1) {
            s.fval[inst * 32 + 16 + lane] = vb0;
            s.fval[inst * 32 + 24 + lane] = vb1;
        }
    }
}

void rv3_launch_init(const Rv3Const& c, const Rv3State& s, size_t instances, hipStream_t stream) {
    hipLaunchKernelGGL(k_rv3_zero, dim3(2048), dim3(256), 0, stream, c, s, instances, 1);
}
void rv3_launch_reset(const Rv3Const& c, const Rv3State& s, size_t instances, hipStream_t stream) {
    hipLaunchKernelGGL(k_rv3_zero, dim3(2048), dim3(256), 0, stream, c, s, instances, 0);
}
void rv3_launch_migrate(const Rv3Const& from, const Rv3State& sfrom, const Rv3Const& to, const Rv3State& sto, size_t instances, hipStream_t stream) {
    if (instances == 0) return;
    hipLaunchKernelGGL(k_rv3_migrate, dim3((unsigned)instances), dim3(64), 0, stream, from, sfrom, to, sto, instances);
}
void rv3_launch_render(const Rv3Const& c, const Rv3State& s, size_t instances, const float* in, float* out, size_t T, size_t fstride,
                       int layout, hipStream_t stream, const FdnBus& bus) {
    if (instances == 0 || T == 0) return;
    tl_opts.last_kernel = LK_FDN_FRAMES;
    if (c.fkind == 1) hipLaunchKernelGGL(k_rv3_render<1>, dim3((unsigned)((instances + 3) / 4)), dim3(256), 0, stream, c, s, instances, in, out, T, fstride, layout, bus);
    else hipLaunchKernelGGL(k_rv3_render<0>, dim3((unsigned)((instances + 3) / 4)), dim3(256), 0, stream, c, s, instances, in, out, T, fstride, layout, bus);
}

}  // namespace fd

// fd_fdn.hpp -- the Hadamard feedback delay networks of FunDSP's reverbs: reverb_stereo (BASELINE config 5: one 32-line network) and
// reverb4_stereo (two 16-line networks in series), both `fdn::<N>(stacki(|i| delay(d_i) >> fir(w)))` shapes (prelude.rs:1336).
//
// Reference: prelude.rs:1732-1762 builds
//     multisplit::<U2,U16>() >> fdn::<U32>(stacki(|i| delay(DELAYS[i]*room/10) >> fir(weights)))
//                            >> sumf::<U32>(|x| pan(lerp(-1, 1, smooth9(x)))) * dc((1/16, 1/16))
// with fdn = Feedback<U32, _, FrameHadamard> (feedback.rs:18-146), Delay (delay.rs:72-139), Fir<U3> (fir.rs:14-89).
//
// Two formulations, identical samples (fdsp_set_option("fdn_kernel", ..)):
//  * lane = FRAME (default, k_fdn_render_frames): one wave renders one instance.  Every delay is longer than two blocks,
//    so inside a 64-frame block the ring reads -- and with them the FIR outputs, the Hadamard and the feedback values --
//    do not depend on the block's own writes: the 32 lines sit in 32 registers, the Hadamard is 5 x 32 register
//    butterflies in the reference's stage order, ring rows are loaded (one block ahead) and stored in the lane = frame
//    order HBM wants, the pan sum is a register fold; two 8.7-KB LDS rows per wave carry the one- and two-frame shifts.
//  * lane = DELAY LINE (k_fdn_render<IPW>): 32 lanes per instance, cross-lane Hadamard (DPP quad_perm / row mirrors,
//    v_permlane16_swap), ring rows staged through LDS tiles.
// Delay rings live in HBM (32 x (4096 + 64) floats = 520 KiB per instance at 48 kHz, room size 10); both kernels move
// the algorithmic minimum of 272 B per instance-frame (32 ring reads + 32 ring writes + 2 in + 2 out).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace fd {

// Ring memory of one instance: 32 rings of CP = C + 64 floats, C a power of two >= the longest delay + 1 (and >= 256).
// ALL lines share one write position w in [0, C): the sample of frame n goes to slot n mod C of every ring, and line k
// reads slot (w - D_k) mod C, D_k = len_k - 1 -- the same delay as the reference's Delay (write at i, advance, read at the
// new i: the sample written len - 1 ticks ago, delay.rs:116-124; slots never written read as the zeros of reset()).
// The last 64 floats of a ring MIRROR its first 64 (every store to a slot p < 64 also goes to C + p), so a block's 64
// consecutive reads starting anywhere in [0, C) never wrap; a block's 64 consecutive writes wrap once in C / 64 blocks.
// That makes the ring addresses of a block SCALAR (base + one wave-uniform offset + lane): no per-lane index arithmetic.
struct FdnConst {            // uniform over the bank (all instances share room / time / damping)
    int sections;            // 1: reverb_stereo (one 32-line FDN); 2: reverb4_stereo (two 16-line FDNs in series, lines 0-15 | 16-31)
    // the generic network `split / multisplit >> fdn::<N>(stacki(|i| delay(t_i) >> fir(w))) >> join / multijoin` (prelude.rs:1323-1345 and its
    // doc example :1334): generic != 0, `lines` = N (2, 4, 8, 16, 32), `taps` = the FIR's order (1..3), line k takes input channel k % nin
    // (Split<N> :527-568, MultiSplit<M, N/M> :571-613), output channel j averages lines j, j + nout, .. (Join<N> :617-660, MultiJoin<M, N/M>
    // :668-730).  The reverbs: generic = 0, lines = 32 (all of them, both sections), taps = 3, nin = nout = 2.
    int generic, lines, taps, nin, nout;
    float had_scale;         // (1.0 / sqrt(lines per section as f64)) as f32   feedback.rs:57
    float out_scale;         // the `* dc((s, s))` behind the pan fold: 1/16 (reverb_stereo), 1/4 (reverb4_stereo)
    int len[32];             // Delay ring length of the reference = delay in samples + 1  (delay.rs:108-110)
    int cap;                 // C: slots per ring (power of two); rings are cap + 64 floats apart
    float w[3];              // FIR weights: fir3(1 - damping).weights() * a  (prelude.rs:1746-1747)
    float wl[32], wr[32];    // pan weights of the output panners (prelude.rs:1759, pan.rs:13-17), indexed by the LINE they pan: all 32
                             // (reverb_stereo) or lines 16-31, the second network's (reverb4_stereo: sumf::<U16>)
    size_t ring_stride;      // floats per instance = lines * (cap + 64)
};

// what a generic network is made of (host side; uniform over the bank like the reverbs' room / time / damping)
struct FdnDesc {
    int lines = 0, taps = 0, nin = 0, nout = 0;
    double delay[32] = {};   // seconds, Delay::new(t) per line (delay.rs:93-113)
    float w[3] = {};         // Fir::new(weights), the same for every line
};

struct FdnState {
    float* rings;            // [instances][32][cap + 64]
    int* wpos;               // [instances]       shared write position (the reference's 32 Delay::i are all w mod len_k)
    float* v1;               // [instances][32]   Fir::v[1]
    float* v2;               // [instances][32]   Fir::v[2]
    float* fb;               // [instances][32]   Feedback::value
};

// A gain and a dry bus around the network -- the way the reference's documentation puts its reverbs to use: `multipass() & 0.2 * reverb_stereo(20.0,
// 2.0, 1.0)` (README.md:436), `0.2 * reverb_stereo(10.0, 1.0, 0.5) & multipass()` (wave.rs:514), `wet * reverb_stereo(10.0, time) & (1.0 - wet) *
// multipass()` (CHANGES.md:203).  `wet * node` is Unop<X, FrameMulScalar> (combinator.rs:477-488; every output sample times the scalar,
// audionode.rs:1190-1228), `x & y` is Bus (audionode.rs:1842-1877: both sides see the node's input, the outputs are added -- tick :1862-1866 and
// process :1868-1877 the same one addition per sample), MultiPass hands its input on (:373-403).  Folded into the kernels' epilogue (the block's
// input frames are still in registers there): out = wet * y (mode 1), out = dry * in + wet * y (mode 2), one rounding per operation like the
// reference's three nodes; a factor of 1.0 is the node the host left out (x * 1.0 == x).  Mode 2 needs as many outputs as inputs.
struct FdnBus {
    int mode = 0;            // 0: the network alone | 1: wet * network | 2: dry * multipass() & wet * network
    float wet = 1.0f, dry = 1.0f;
};
// (the bus of one output sample; `x` = the input sample of the same channel and frame)
__device__ __forceinline__ float fdn_bus(const FdnBus& b, float y, float x) {
    if (b.mode == 0) return y;
    const float w = b.wet * y;
    if (b.mode == 1) return w;
    const float d = b.dry * x;
    return d + w;
}

// host: constants of reverb_stereo(room_size, time, damping) at `sample_rate` (prelude.rs:1739-1759)
void fdn_make_const(double room_size, double time, double damping, double sample_rate, FdnConst* c);
void fdn_launch_reset(const FdnConst& c, const FdnState& s, size_t instances, hipStream_t stream);
// host: constants of reverb4_stereo(room_size, time) at `sample_rate` (prelude.rs:1873-1941): two fdn::<U16> of delay >> fir3 lines in
// series, `multijoin::<U2, U8>() >> multisplit::<U2, U8>()` between them, sumf::<U16>(pan) * dc((1/4, 1/4)) behind the second
void fdn_make_const_reverb4(double room_size, double time, double sample_rate, FdnConst* c);
// tick_mode: MultiJoin::tick sums and divides, MultiJoin::process scales every term first (audionode.rs:697-720) -- the only place where
// the two executors of these graphs differ in arithmetic (reverb4_stereo; reverb_stereo has no join)
// host: constants of the generic network at `sample_rate`
void fdn_make_const_generic(const FdnDesc& d, double sample_rate, FdnConst* c);
void fdn_launch_render(const FdnConst& c, const FdnState& s, size_t instances, const float* in, float* out, size_t T,
                       size_t fstride, int layout, int tick_mode, hipStream_t stream, const FdnBus& bus = FdnBus());

// [channels][T][V] (voice-minor) <-> [V][channels][T] (planar, frame stride T): the staging copies of voice-minor launches (fd_fdn.hip)
void fdn_launch_transpose(const float* src, float* dst, size_t V, size_t T, int channels, bool to_planar, hipStream_t stream);

}  // namespace fd

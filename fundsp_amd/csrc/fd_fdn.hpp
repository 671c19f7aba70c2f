// fd_fdn.hpp -- the 32-line feedback delay network of FunDSP's reverb_stereo (BASELINE config 5).
//
// Reference: prelude.rs:1732-1762 builds
//     multisplit::<U2,U16>() >> fdn::<U32>(stacki(|i| delay(DELAYS[i]*room/10) >> fir(weights)))
//                            >> sumf::<U32>(|x| pan(lerp(-1, 1, smooth9(x)))) * dc((1/16, 1/16))
// with fdn = Feedback<U32, _, FrameHadamard> (feedback.rs:18-146), Delay (delay.rs:72-139), Fir<U3> (fir.rs:14-89).
//
// Mapping (differs from the lane-per-voice kernels on purpose): one lane per DELAY LINE, 32 lanes per reverb
// instance, two instances per wave64.  The 32-point Hadamard of every sample is 5 cross-lane butterfly stages
// (DPP quad_perm for strides 1, 2; ds_swizzle bit-mode for 4, 8, 16) in the reference's stage order.  Delay rings live
// in HBM (372.6 KiB per instance at 48 kHz); because every delay is longer than a 64-sample block, a block's 64
// ring reads per line do not depend on its writes, so each block stages 64 contiguous samples per line through LDS
// with coalesced 256-B reads and writes -- 272 B of HBM traffic per instance-frame, the algorithmic minimum.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace fd {

struct FdnConst {            // uniform over the bank (all instances share room / time / damping)
    int off[32];             // ring offset of each line inside an instance's ring block (floats)
    int len[32];             // ring length = delay in samples + 1  (delay.rs:108-110)
    float w[3];              // FIR weights: fir3(1 - damping).weights() * a  (prelude.rs:1746-1747)
    float wl[32], wr[32];    // pan weights of the 32 output panners (prelude.rs:1759, pan.rs:13-17)
    size_t ring_stride;      // floats per instance
};

struct FdnState {
    float* rings;            // [instances][ring_stride]
    int* idx;                // [instances][32]   Delay::i
    float* v1;               // [instances][32]   Fir::v[1]
    float* v2;               // [instances][32]   Fir::v[2]
    float* fb;               // [instances][32]   Feedback::value
};

// host: constants of reverb_stereo(room_size, time, damping) at `sample_rate` (prelude.rs:1739-1759)
void fdn_make_const(double room_size, double time, double damping, double sample_rate, FdnConst* c);
void fdn_launch_reset(const FdnConst& c, const FdnState& s, size_t instances, hipStream_t stream);
void fdn_launch_render(const FdnConst& c, const FdnState& s, size_t instances, const float* in, float* out, size_t T,
                       size_t fstride, int layout, hipStream_t stream);

}  // namespace fd

// fd_reverb3.hpp -- reverb3_stereo(time, diffusion, lowpole_hz(cutoff)) (prelude.rs:1858-1871), the allpass-loop reverb `Reverb<F>` of
// reverb.rs:152-279, in the LANE = FRAME formulation of the FDN kernels (fd_fdn.hpp).
//
// Reference, per sample (Reverb::tick, reverb.rs:241-272): four Schroeder allpasses diffuse the inputs (`pre`), then the signal walks
// eight blocks -- Delay -> 4 allpasses (`a * v + input0` in front) -> loop filter -> 4 allpasses (`a * v + input1`) -> loop filter -- and the
// last block's value is the first block's input on the next sample.  A Schroeder allpass is AllNest<U1, Delay> (delay.rs:294-358):
// v = x - eta * z, y = eta * v + z, z = the delay line's output M ticks after v went in -- read on the NEXT tick, so v[n - (M + 1)].
//
// What makes a block of 64 frames parallel over its frames: every delay in the structure -- the 64 + 4 allpass lines (245 .. 1033
// samples at 44.1 kHz) and the eight block delays (1087 .. 1123) -- is longer than two blocks, so all 76 ring reads of a block are known at
// its head, each allpass is feed-forward inside the block, and the eight blocks do not see each other's output of the same block.  The one
// thing that IS serial in time is the loop filter (the documented one-pole lowpass, filter.rs:19-66: value = (1 - c) * x + c * value; or a
// FixedSvf such as the highshelf_hz the reference's examples put there, svf.rs:995-1006): 16 of them per instance, two per loop block.  So a wave renders one instance with lane = frame for everything but those, and hands each of the two
// filter layers over through a small LDS tile to lanes 0-7 (lane = loop block), which run the 64-step recurrence in registers.
//
// Rings: as in fd_fdn.hpp -- one write position per instance, ring k read at slot (w - dist_k), a 64-float mirror behind every ring so that
// a block's reads are one 256-byte run.  72 main rings per instance (per loop block: 4 + 4 allpass lines and the block delay's line, which
// holds the PREVIOUS block's output -- block 0's holds the last block's, read one slot further back: the reference's `feedback` sample);
// the four `pre` lines live apart with a write position of their own, because Reverb::reset and Reverb::set_sample_rate leave `pre` alone
// (reverb.rs:211-238: neither cleared nor re-sized -- their delays stay at (n - 1) samples whatever the rate).
// Denormals are kept (a graph without a Feedback node never calls prevent_denormals): this translation unit is NOT compiled with
// flush-to-zero, unlike fd_fdn.hip.  624 B per instance-frame: 76 ring reads + 76 ring writes + 2 in + 2 out.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "fd_fdn.hpp"   // FdnBus: the gain and dry bus the lane-per-frame kernels fold into their epilogue

namespace fd {

constexpr int RV3_PRE_CAP = 512;   // slots per `pre` ring (their delays are 245 .. 367 samples at every rate)

struct Rv3Const {                  // uniform over the bank
    int dpre[4];                   // read distance of the `pre` allpasses: (predelay - 1) + 1 samples
    int dap0[8][4], dap1[8][4];    // ... of the loop blocks' allpasses: round((ldelay / rdelay - 1) / 44100 * sr) + 1
    int dblk[8];                   // ... of the block delays: round(delays[7 - b] / 44100 * sr), + 1 for block 0 (the `feedback` sample)
    int cap;                       // slots per main ring (power of two > the longest distance + 64); rings are cap + 64 floats apart
    float eta;                     // lerp(0.5, 0.9, diffusion) as f32   reverb.rs:173
    float a;                       // pow(db_amp(-60.0), 0.035 / time) as f32   :196
    int fkind;                     // the loop filter: 0 = lowpole_hz(cutoff) (filter.rs:19-66), 1 = a FixedSvf (lowpass_hz .. highshelf_hz, svf.rs:861-1031)
    float c, omc;                  // Lowpole: coeff = exp(-TAU * cutoff / sr) (filter.rs:35-38), and 1 - coeff as tick computes it (:65)
    float sa1, sa2, sa3, sm0, sm1, sm2;   // FixedSvf: SvfCoefs of (mode, sr, cutoff, q, gain) (svf.rs:28-221)
    size_t ring_stride;            // floats per instance = 72 * (cap + 64)
};

struct Rv3Filter {                 // host side: which loop filter, with what parameters
    int kind = 0;                  // 0 lowpole_hz(cutoff) | 1 FixedSvf
    int mode = 0;                  // SVF_LOWPASS .. SVF_HIGHSHELF
    float cutoff = 0.0f, q = 1.0f, gain = 1.0f;
};

struct Rv3State {
    float* rings;                  // [instances][72][cap + 64]
    float* pre;                    // [instances][4][RV3_PRE_CAP + 64]
    int* wpos;                     // [instances] write position of the main rings
    int* wpre;                     // [instances] ... of the pre rings (never reset)
    float* fval;                   // [instances][32]: [0..15] Lowpole::value / FixedSvf::ic1eq of filter0 of blocks 0-7, filter1 of blocks 0-7; [16..31] ic2eq
};

// host: constants at `sample_rate`; false when a delay is not longer than two blocks there (the kernel's rule)
bool rv3_make_const(double time, double diffusion, const Rv3Filter& filter, double sample_rate, Rv3Const* c);
void rv3_launch_init(const Rv3Const& c, const Rv3State& s, size_t instances, hipStream_t stream);   // construction: everything zero
void rv3_launch_reset(const Rv3Const& c, const Rv3State& s, size_t instances, hipStream_t stream);  // Reverb::reset: all but `pre`
// Reverb::set_sample_rate to a NEW rate: the lines of `to` are zero (Delay resizes and resets, delay.rs:105-113) but what the reference
// does not reset survives -- every allpass's z (the sample it reads next), the feedback sample, the filters' values, all of `pre`
void rv3_launch_migrate(const Rv3Const& from, const Rv3State& sfrom, const Rv3Const& to, const Rv3State& sto, size_t instances, hipStream_t stream);
void rv3_launch_render(const Rv3Const& c, const Rv3State& s, size_t instances, const float* in, float* out, size_t T, size_t fstride,
                       int layout, hipStream_t stream, const FdnBus& bus = FdnBus());

}  // namespace fd

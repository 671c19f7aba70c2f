// fd_kinds_fm.hpp -- the graph types of BASELINE configs 1 and 3, shared by fd_kinds_fm.hip (their kinds, built with the ILP
// scheduling strategy) and fd_kinds_fm_ts.hip (their three-way time-split kernels, built with -amdgpu-sched-strategy=max-ilp: ROCm
// 7.2's clang crashes in the register allocator when it compiles k_render_ts3 under -amdgpu-sched-strategy=iterative-ilp).
// The types spell out exactly what the reference's operator overloads build (combinator.rs:289-488; Rust precedence
// `*` > `+` > `>>`).
#pragma once

#include "fd_engine.hpp"

namespace fd {
// sine_hz(f) = constant(f) >> sine()                       prelude.rs:349
using SineHz = Pipe<Constant<1>, Sine>;
// config 1: sine_hz(440) >> lowpass_hz(1000, 1)
using SineHzLowpass = Pipe<SineHz, FixedSvf>;
// config 3: sine_hz(f) * f * m + f >> sine() >> lowpass_hz(fc, q)     (README.md:98-103)
using FmMod = Unop<Unop<Unop<SineHz, UMulScalar>, UMulScalar>, UAddScalar>;
using FmSvf = Pipe<Pipe<FmMod, Sine>, FixedSvf>;
#ifndef FD_PIPE_MIN_T
static_assert(PipeMinT<FmSvf>::v == 64 && PipeMinT<SineHzLowpass>::v == 256, "launch lengths that leave the single-wave kernel: config 3 from one block on (measured)");
#endif

#define FD_FM_TS3_KERNELS(X)                                   \
    X(SineHzLowpass, 1) X(SineHzLowpass, 2)                    \
    X(typename FastOf<SineHzLowpass>::type, 1) X(typename FastOf<SineHzLowpass>::type, 2) \
    X(FmSvf, 1) X(FmSvf, 2)                                    \
    X(typename FastOf<FmSvf>::type, 1) X(typename FastOf<FmSvf>::type, 2)
}  // namespace fd

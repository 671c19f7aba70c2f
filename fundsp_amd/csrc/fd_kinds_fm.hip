// fd_kinds_fm.hip -- the oscillator -> filter chains of BASELINE configs 1 and 3 (the headline kernel lives here).
// The types spell out exactly what the reference's operator overloads build (combinator.rs:289-488; Rust precedence
// `*` > `+` > `>>`).
//
// A translation unit of its own because it is built with `-mllvm -amdgpu-sched-strategy=iterative-ilp` (Makefile): the
// stage that holds the carrier sine AND the serial SVF recurrence is one ~230-instruction basic block per SIMD item, and
// the default scheduler flips -- on unrelated edits -- between interleaving the four frame pairs' packed sine work and
// emitting it pair by pair with an `s_nop` after every dependent packed op (38-44 per item, 6-14 % slower; profiles/
// r03_ab1_knockout_prio.txt, r03_isa_fm_svf.txt).  The ILP strategy produces the interleaved form every time.  It cannot
// be a whole-library flag: ROCm 7.2's clang crashes with it in the register allocator on other kinds (Oversampler).
#include "fd_kinds_fm.hpp"

namespace fd {
// the three-way time-split kernels of these types are compiled in fd_kinds_fm_ts.hip (-amdgpu-sched-strategy=max-ilp, see the Makefile)
#define FD_X(G, GPW) extern template __global__ void k_render_ts3<G, GPW>(float* __restrict__, size_t, size_t, float* __restrict__, size_t, const void*);
FD_FM_TS3_KERNELS(FD_X)
#undef FD_X

void register_fm_kinds(std::vector<KindOps>& out) {
    out.push_back(make_kind<SineHz>("sine_hz"));
    out.push_back(make_kind<SineHzLowpass>("sine_hz_lowpass_hz"));
    out.push_back(make_kind<FmSvf>("fm_svf"));
}
}  // namespace fd

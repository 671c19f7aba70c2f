// fd_kinds_fm_mix.hip -- the kernels with the fused mix-down of the oscillator -> filter chains (BASELINE configs 1 and 3): the
// pipeline kernel here, built like fd_kinds_fm.hip with the ILP scheduling strategy (same hot loops); the three-way time-split
// kernels (the 2-, 4-, 8-GPU shards) are instantiated in fd_kinds_fm_ts.hip and declared `extern template` here.
#include "fd_kinds_fm.hpp"

namespace fd {
#define FD_X(G, GPW)                                                                                                                          \
    extern template __global__ void k_render_ts3_mix<G, GPW, MIX_SUM>(float* __restrict__, size_t, size_t, float* __restrict__, size_t, const void*, const float* __restrict__); \
    extern template __global__ void k_render_ts3_mix<G, GPW, MIX_PAN>(float* __restrict__, size_t, size_t, float* __restrict__, size_t, const void*, const float* __restrict__);
FD_FM_TS3_KERNELS(FD_X)
#undef FD_X

void attach_fm_mix(std::vector<KindOps>& out) {
    attach_mix<SineHzLowpass>(out, "sine_hz_lowpass_hz");
    attach_mix<FmSvf>(out, "fm_svf");
}
}  // namespace fd

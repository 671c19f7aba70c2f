// fd_kinds_fm_ts.hip -- explicit instantiations of the three-way time-split kernels (fd_device.hpp k_render_ts3) of the
// config-1 / config-3 graph types; fd_kinds_fm.hip declares them `extern template` (see fd_kinds_fm.hpp for why).
#include "fd_kinds_fm.hpp"

namespace fd {
#define FD_X(G, GPW) template __global__ void k_render_ts3<G, GPW>(float* __restrict__, size_t, size_t, float* __restrict__, size_t, const void*);
FD_FM_TS3_KERNELS(FD_X)
#undef FD_X
// ... and with the fused mix-down (fd_kinds_fm_mix.hip)
#define FD_X(G, GPW)                                                                                                                   \
    template __global__ void k_render_ts3_mix<G, GPW, MIX_SUM>(float* __restrict__, size_t, size_t, float* __restrict__, size_t, const void*, const float* __restrict__); \
    template __global__ void k_render_ts3_mix<G, GPW, MIX_PAN>(float* __restrict__, size_t, size_t, float* __restrict__, size_t, const void*, const float* __restrict__);
FD_FM_TS3_KERNELS(FD_X)
#undef FD_X
}  // namespace fd

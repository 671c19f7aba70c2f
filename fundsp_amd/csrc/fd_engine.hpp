// fd_engine.hpp -- host side of the bank kernels: per-kind dispatch table for the ahead-of-time compiled voice graphs.
#pragma once

#include <functional>
#include <string>
#include <vector>

#include "fd_device.hpp"
#include "fd_opts.hpp"

namespace fd {

struct SlotInfo {
    std::string name;
    int kind;
};

struct VDescribe {  // host only
    std::vector<SlotInfo>* out;
    std::vector<int> path;
    std::string prefix() const {
        std::string s;
        for (size_t i = 0; i < path.size(); i++) {
            if (i) s += ".";
            s += std::to_string(path[i]);
        }
        return s;
    }
    void add(const std::string& field, int kind) { out->push_back({prefix() + ":" + field, kind}); }
    void f(float&, FieldKind k, const char* name) { add(name, k); }
    void fi(float&, FieldKind k, const char* name, int index) {
        add(std::string(name) + "[" + std::to_string(index) + "]", k);
    }
    void u32(uint32_t&, FieldKind k, const char* name) { add(name, k); }
    void u64(uint64_t&, FieldKind k, const char* name) {
        add(std::string(name) + ".lo", k);
        add(std::string(name) + ".hi", k);
    }
    void enter(int i) { path.push_back(i); }
    void leave() { path.pop_back(); }
};

int simd_count();  // SIMDs of the current device (CUs x 4), fd_capi.hip

// ---- per-kind dispatch table ---------------------------------------------------------------------------------
struct KindOps {
    std::string name;
    int nin, nout, nrings;
    std::vector<SlotInfo> slots;
    // parameters the graph's Rust TYPE carries (filter modes, shape kinds ...: fdsp_graph_compile_rust): applied to every
    // voice of a new bank right after construction
    std::vector<std::pair<std::string, float>> presets;
    std::function<void(float* slots, size_t stride, size_t first, size_t count, int op, double sr,
                       const uint64_t* d_seeds, const void* aux, float* ring, uint32_t ring_cap, hipStream_t s)>
        lifecycle;
    std::function<void(float* slots, size_t stride, size_t V, const float* in, float* out, size_t T, size_t fstride,
                       int layout, int mode, const void* aux, float* ring, uint32_t ring_cap, hipStream_t s)>
        render;
    // the same launch for the tolerance-mode variant FastOf<G> of the graph (fdsp_set_option("math", 1)); empty when the
    // graph has no node with a tolerance-mode form (then `render` is used)
    std::function<void(float* slots, size_t stride, size_t V, const float* in, float* out, size_t T, size_t fstride,
                       int layout, int mode, const void* aux, float* ring, uint32_t ring_cap, hipStream_t s)>
        render_fast;
    // Render + mix-down in one launch (fd_device.hpp "fused mix-down"): part = device [voice groups][mix channels][T] partial
    // mixes, mix = MIX_SUM / MIX_PAN, panw = device [2][stride] pan weights (MIX_PAN).  Voice-minor inputs.  false = this graph /
    // launch has no fused kernel.  Empty for kinds built without one (attach_mix<G>).
    std::function<bool(float* slots, size_t stride, size_t V, const float* in, float* part, size_t T, int mix, int mode,
                       const void* aux, float* ring, uint32_t ring_cap, const float* panw, hipStream_t s)>
        render_mix, render_mix_fast;
    // Whatever the fused mix-down of this kind still has to do before its first launch (run-time compiled graphs: compile and load the
    // mix kernels) -- called by fdsp_bank_mix_reserve AND fdsp_bank_set_pan (either may be the last call a host makes BEFORE its real-time
    // loop or stream capture), and by fdsp_bank_set_option("math") for a bank that already holds a partial-mix buffer; `fast`: the tolerance-mode variant.  Empty for ahead-of-time kinds.
    std::function<void(bool fast)> prepare_mix;
    // ... and the render path of a bank of `voices` voices (run-time compiled three-stage generator chains: the time-split kernels that
    // small banks take live in the kind's second module; `fast`: the tolerance-mode twin FastOf<G>, a module of its own) -- called when a
    // bank is created and when fdsp_bank_set_option switches its arithmetic, so that no render compiles anything.
    std::function<void(size_t voices, bool fast)> prepare_render;
    // ... and the Sequencer's mixed output in one launch (render_events_body MIXE): part as above, [groups][outputs][T]; graphs of
    // at most two outputs (the block tiles of four waves must fit the CU's LDS)
    std::function<bool(float* slots, size_t stride, size_t V, const float* in, float* part, size_t T, const double* ev,
                       const int* fade, double time0, double sr, int mode, const void* aux, float* ring, uint32_t ring_cap, hipStream_t s)>
        render_events_mix;
    // Sequencer-style rendering (fd_device.hpp render_events_body): ev = device [4][stride] f64, fade = device [V] or null
    std::function<void(float* slots, size_t stride, size_t V, const float* in, float* out, size_t T, const double* ev,
                       const int* fade, double time0, double sr, int mode, const void* aux, float* ring,
                       uint32_t ring_cap, hipStream_t s)>
        render_events;
};

template <class G>
void launch_lifecycle(float* slots, size_t stride, size_t first, size_t count, int op, double sr,
                      const uint64_t* d_seeds, const void* aux, float* ring, uint32_t ring_cap, hipStream_t s) {
    if (count == 0) return;
    unsigned grid = (unsigned)((count + 63) / 64);
    hipLaunchKernelGGL((k_lifecycle<G>), dim3(grid), dim3(64), 0, s, slots, stride, first, count, op, sr, d_seeds, aux,
                       ring, ring_cap);
}

template <class G, int MODE, int LAYOUT>
void launch_render_cfg(float* slots, size_t stride, size_t V, const float* in, float* out, size_t T, size_t fstride,
                       const void* aux, float* ring, uint32_t ring_cap, hipStream_t s) {
    // 4-wave workgroups unless the per-wave LDS tiles of the planar path would not fit 4x in the CU's LDS
    constexpr int WPB = RenderGeom<G, LAYOUT>::WPB;
    const int vpw = LAYOUT == LAYOUT_VOICE_MINOR ? voices_per_wave(V, simd_count()) : 64;
    if (LAYOUT == LAYOUT_VOICE_MINOR) fstride = (size_t)vpw;  // see render_body: fstride carries voices-per-wave here
    const size_t waves = (V + vpw - 1) / vpw;
    unsigned grid = (unsigned)((waves + WPB - 1) / WPB);
    hipLaunchKernelGGL((k_render<G, MODE, LAYOUT, WPB>), dim3(grid), dim3(64 * WPB), 0, s, slots, stride, V, in, out, T,
                       fstride, aux, ring, ring_cap);
}

template <class G, int MODE, int WANT>
bool launch_render_pipe(float* slots, size_t stride, size_t V, const float* in, float* out, size_t T, const void* aux,
                        float* ring, uint32_t ring_cap, hipStream_t s) {
    constexpr PipePlan P = pipe_plan<G>(WANT);
    if constexpr (P.S >= 1) {
        constexpr int WAVES = PipeGeom<G::IN, P.S>::WAVES;  // for 4 voice groups
        const size_t cus = (size_t)simd_count() / 4;
        // light graphs keep 4 groups per workgroup; heavy ones (latency-bound waves) are spread so that every CU gets one
        const size_t groups = (V + 63) / 64;
        bool done = false;
        // (a workgroup of 1 or 2 groups spends the LDS of 4 on longer tiles -- PipeTiles -- so it must be alone on its CU)
        if constexpr (Cost<G>::v >= 150) {
            if (!done && groups <= cus) {
                hipLaunchKernelGGL((k_render_pipe<G, MODE, P.S, P.K1, P.K2, 1>), dim3((unsigned)groups), dim3(16 * WAVES), 0, s, slots,
                                   stride, V, in, out, T, aux, ring, ring_cap);
                done = true;
            } else if (!done && groups <= 2 * cus) {
                hipLaunchKernelGGL((k_render_pipe<G, MODE, P.S, P.K1, P.K2, 2>), dim3((unsigned)((groups + 1) / 2)), dim3(16 * 2 * WAVES), 0,
                                   s, slots, stride, V, in, out, T, aux, ring, ring_cap);
                done = true;
            }
        }
        if (!done)
            hipLaunchKernelGGL((k_render_pipe<G, MODE, P.S, P.K1, P.K2, 4>), dim3((unsigned)((groups + 3) / 4)), dim3(16 * 4 * WAVES), 0, s,
                               slots, stride, V, in, out, T, aux, ring, ring_cap);
        return true;
    } else {
        return false;
    }
}
template <class G, int MODE>
bool launch_render_split(float* slots, size_t stride, size_t V, const float* in, float* out, size_t T, const void* aux,
                         float* ring, uint32_t ring_cap, hipStream_t s) {
    if (tl_opts.pipe_split == 4) return launch_render_pipe<G, MODE, 1>(slots, stride, V, in, out, T, aux, ring, ring_cap, s);
    if (tl_opts.pipe_split == 2) return launch_render_pipe<G, MODE, 2>(slots, stride, V, in, out, T, aux, ring, ring_cap, s);
    if (tl_opts.pipe_split == 3) return launch_render_pipe<G, MODE, 3>(slots, stride, V, in, out, T, aux, ring, ring_cap, s);
    return launch_render_pipe<G, MODE, 0>(slots, stride, V, in, out, T, aux, ring, ring_cap, s);
}

// planar layout through the pipeline kernel (loader / compute stages / storer); false = not applicable
template <class G, int MODE>
bool launch_render_pipe_planar(float* slots, size_t stride, size_t V, const float* in, float* out, size_t T, size_t fstride,
                               const void* aux, float* ring, uint32_t ring_cap, hipStream_t s) {
    using PP = PlanarPlan<G>;
    if constexpr (PP::S >= 1) {
        constexpr int WAVES = PP::T::WAVES;
        const size_t groups = (V + 63) / 64;
        const size_t cus = (size_t)simd_count() / 4;
        // heavy graphs (latency-bound waves) are spread so that every CU gets a workgroup, as in launch_render_pipe
        bool done = false;
        if constexpr (Cost<G>::v >= 150) {
            if (groups < 2 * cus) {
                hipLaunchKernelGGL((k_render_pipe_planar<G, MODE, PP::S, PP::K1, 1>), dim3((unsigned)groups), dim3(64 * WAVES), 0, s, slots,
                                   stride, V, in, out, T, fstride, aux, ring, ring_cap);
                done = true;
            } else if (groups < 4 * cus) {
                hipLaunchKernelGGL((k_render_pipe_planar<G, MODE, PP::S, PP::K1, 2>), dim3((unsigned)((groups + 1) / 2)), dim3(128 * WAVES), 0,
                                   s, slots, stride, V, in, out, T, fstride, aux, ring, ring_cap);
                done = true;
            }
        }
        if (!done)
            hipLaunchKernelGGL((k_render_pipe_planar<G, MODE, PP::S, PP::K1, 4>), dim3((unsigned)((groups + 3) / 4)), dim3(256 * WAVES), 0, s,
                               slots, stride, V, in, out, T, fstride, aux, ring, ring_cap);
        return true;
    } else {
        return false;
    }
}


// Launch lengths from which a launch leaves the single-wave kernel: PipeMinT<G> (fd_device.hpp) for the stage pipeline, and one whole
// 64-frame block for the time-split kernels of small banks (config 3's shards: 9.2-9.8 us against 13.5-13.9 at T = 64, 12.5-13.4 against
// 22 at 128, 15-16 against 30 at 192; profiles/r04_small_t_kernels.txt).  "pipe_split" 2 / 3 force the pipeline at any length, 0 the
// single-wave kernel.
#ifndef FD_TS_MIN_T
#define FD_TS_MIN_T 64
#endif
// The planar pipeline (loader waves transpose 16-byte runs of the per-voice rows through LDS, a storer wave transposes back) beats the
// single-wave kernel's strided row accesses at every measured length and for every graph -- config 3: 9.8 vs 10.9 us at T = 16, 16.2 vs
// 19.1 at 64; config 4: 18.8 vs 28.0, 30.8 vs 51.9; config 2: 6.5 vs 6.9, 8.5 vs 9.0 (profiles/r04_small_t_kernels.txt, planar table).
#ifndef FD_PLANAR_PIPE_MIN_T
#define FD_PLANAR_PIPE_MIN_T 16
#endif
template <class G>
void launch_render(float* slots, size_t stride, size_t V, const float* in, float* out, size_t T, size_t fstride,
                   int layout, int mode, const void* aux, float* ring, uint32_t ring_cap, hipStream_t s) {
    if (V == 0 || T == 0) return;
    // banks that leave most SIMDs idle (<= 2 voice groups per CU): split the oscillator stages over time as well
    if constexpr (TsPlan<G>::ok) {
        const size_t groups = (V + 63) / 64, cus = (size_t)simd_count() / 4;
        if (tl_opts.time_split && tl_opts.pipe_split == 1 && layout == LAYOUT_VOICE_MINOR && mode == MODE_PROCESS && T % 64 == 0 && T >= FD_TS_MIN_T &&
            groups <= 2 * cus) {
            if (tl_opts.time_split == 1) {  // round 3: both oscillator stages split three ways, the filter wave (nearly) alone on a SIMD
                if (groups <= cus)
                    hipLaunchKernelGGL((k_render_ts3<G, 1>), dim3((unsigned)groups), dim3(64 * Ts3Roles<1>::WAVES), 0, s, slots, stride, V, out, T, aux);
                else
                    hipLaunchKernelGGL((k_render_ts3<G, 2>), dim3((unsigned)((groups + 1) / 2)), dim3(64 * Ts3Roles<2>::WAVES), 0, s, slots, stride, V, out, T, aux);
            } else if (groups <= cus)  // time_split = 2: round 2's layouts.  One workgroup per CU: 2 + 2 + 1 waves
                hipLaunchKernelGGL((k_render_ts<G, 2, 2>), dim3((unsigned)groups), dim3(64 * 5), 0, s, slots, stride, V, out, T, aux);
            else                // two workgroups per CU: 2 + 1 + 1 waves each, roles rotated between neighbours
                hipLaunchKernelGGL((k_render_ts<G, 2, 1>), dim3((unsigned)groups), dim3(64 * 4), 0, s, slots, stride, V, out, T, aux);
            tl_opts.last_kernel = LK_TIME_SPLIT;
            return;
        }
    }
    // planar rows that allow 16-byte runs go through the planar pipeline (same launch-size rule as below)
    if (layout == LAYOUT_PLANAR && tl_opts.pipe_split && (T >= FD_PLANAR_PIPE_MIN_T || tl_opts.pipe_split > 1) && fstride % 4 == 0 && ((uintptr_t)in & 15) == 0 &&
        ((uintptr_t)out & 15) == 0) {
        const bool done = mode == MODE_PROCESS ? launch_render_pipe_planar<G, MODE_PROCESS>(slots, stride, V, in, out, T, fstride, aux, ring, ring_cap, s)
                                               : launch_render_pipe_planar<G, MODE_TICK>(slots, stride, V, in, out, T, fstride, aux, ring, ring_cap, s);
        if (done) { tl_opts.last_kernel = LK_PIPELINE_PLANAR; return; }
    }
    // the stage pipeline from PipeMinT<G> frames on (one 64-frame block for chains worth cutting, four for light graphs)
    if (layout == LAYOUT_VOICE_MINOR && tl_opts.pipe_split && (T >= (size_t)PipeMinT<G>::v || tl_opts.pipe_split > 1)) {
        const bool done = mode == MODE_PROCESS ? launch_render_split<G, MODE_PROCESS>(slots, stride, V, in, out, T, aux, ring, ring_cap, s)
                                               : launch_render_split<G, MODE_TICK>(slots, stride, V, in, out, T, aux, ring, ring_cap, s);
        if (done) { tl_opts.last_kernel = LK_PIPELINE; return; }
    }
    tl_opts.last_kernel = LK_SINGLE_WAVE;
    if (layout == LAYOUT_VOICE_MINOR) {
        if (mode == MODE_PROCESS)
            launch_render_cfg<G, MODE_PROCESS, LAYOUT_VOICE_MINOR>(slots, stride, V, in, out, T, fstride, aux, ring, ring_cap, s);
        else
            launch_render_cfg<G, MODE_TICK, LAYOUT_VOICE_MINOR>(slots, stride, V, in, out, T, fstride, aux, ring, ring_cap, s);
    } else {
        if (mode == MODE_PROCESS)
            launch_render_cfg<G, MODE_PROCESS, LAYOUT_PLANAR>(slots, stride, V, in, out, T, fstride, aux, ring, ring_cap, s);
        else
            launch_render_cfg<G, MODE_TICK, LAYOUT_PLANAR>(slots, stride, V, in, out, T, fstride, aux, ring, ring_cap, s);
    }
}

// ---- render + mix-down in one launch ------------------------------------------------------------------------------
template <class G, int MODE, int MIX>
bool launch_render_pipe_mix(float* slots, size_t stride, size_t V, const float* in, float* part, size_t T, const void* aux,
                            float* ring, uint32_t ring_cap, const float* panw, hipStream_t s) {
    constexpr PipePlan P = pipe_plan<G>(0);
    if constexpr (P.S >= 1) {
        constexpr int WAVES = PipeGeom<G::IN, P.S>::WAVES;  // for 4 voice groups
        const size_t cus = (size_t)simd_count() / 4, groups = (V + 63) / 64;
        // the same workgroup widths as launch_render_pipe: heavy graphs on small banks are spread over the CUs
        if constexpr (Cost<G>::v >= 150) {
            if (groups <= cus) {
                hipLaunchKernelGGL((k_render_pipe_mix<G, MODE, P.S, P.K1, P.K2, 1, MIX>), dim3((unsigned)groups), dim3(16 * WAVES), 0, s, slots,
                                   stride, V, in, part, T, aux, ring, ring_cap, panw);
                return true;
            }
            if (groups <= 2 * cus) {
                hipLaunchKernelGGL((k_render_pipe_mix<G, MODE, P.S, P.K1, P.K2, 2, MIX>), dim3((unsigned)((groups + 1) / 2)), dim3(16 * 2 * WAVES), 0,
                                   s, slots, stride, V, in, part, T, aux, ring, ring_cap, panw);
                return true;
            }
        }
        hipLaunchKernelGGL((k_render_pipe_mix<G, MODE, P.S, P.K1, P.K2, 4, MIX>), dim3((unsigned)((groups + 3) / 4)), dim3(16 * 4 * WAVES), 0, s,
                           slots, stride, V, in, part, T, aux, ring, ring_cap, panw);
        return true;
    } else {
        return false;
    }
}
template <class G, int MIX>
bool launch_render_mix_m(float* slots, size_t stride, size_t V, const float* in, float* part, size_t T, int mode, const void* aux,
                         float* ring, uint32_t ring_cap, const float* panw, hipStream_t s) {
    if constexpr (TsPlan<G>::ok) {  // small banks of oscillator chains: the three-way time split, as in launch_render
        const size_t groups = (V + 63) / 64, cus = (size_t)simd_count() / 4;
        if (tl_opts.time_split == 1 && tl_opts.pipe_split == 1 && mode == MODE_PROCESS && T % 64 == 0 && T >= FD_TS_MIN_T && groups <= 2 * cus) {
            if (groups <= cus)
                hipLaunchKernelGGL((k_render_ts3_mix<G, 1, MIX>), dim3((unsigned)groups), dim3(64 * Ts3Roles<1>::WAVES), 0, s, slots, stride, V, part, T, aux, panw);
            else
                hipLaunchKernelGGL((k_render_ts3_mix<G, 2, MIX>), dim3((unsigned)((groups + 1) / 2)), dim3(64 * Ts3Roles<2>::WAVES), 0, s, slots, stride, V, part, T, aux, panw);
            tl_opts.last_kernel = LK_TIME_SPLIT;
            return true;
        }
    }
    const bool done = mode == MODE_PROCESS ? launch_render_pipe_mix<G, MODE_PROCESS, MIX>(slots, stride, V, in, part, T, aux, ring, ring_cap, panw, s)
                                           : launch_render_pipe_mix<G, MODE_TICK, MIX>(slots, stride, V, in, part, T, aux, ring, ring_cap, panw, s);
    if (done) tl_opts.last_kernel = LK_PIPELINE;
    return done;
}
template <class G>
bool launch_render_mix(float* slots, size_t stride, size_t V, const float* in, float* part, size_t T, int mix, int mode,
                       const void* aux, float* ring, uint32_t ring_cap, const float* panw, hipStream_t s) {
    if (V == 0 || T == 0) return true;
    if (mix == MIX_PAN) {
        if constexpr (G::OUT == 1) return launch_render_mix_m<G, MIX_PAN>(slots, stride, V, in, part, T, mode, aux, ring, ring_cap, panw, s);
        else return false;
    }
    return launch_render_mix_m<G, MIX_SUM>(slots, stride, V, in, part, T, mode, aux, ring, ring_cap, panw, s);
}
template <class G>
bool launch_render_events_mix(float* slots, size_t stride, size_t V, const float* in, float* part, size_t T, const double* ev,
                              const int* fade, double time0, double sr, int mode, const void* aux, float* ring,
                              uint32_t ring_cap, hipStream_t s) {
    if (V == 0 || T == 0) return true;
    if constexpr (G::OUT <= 2) {
        tl_opts.last_kernel = LK_EVENTS;
        const unsigned grid = (unsigned)(((V + 63) / 64 + 3) / 4);
        if (mode == MODE_PROCESS)
            hipLaunchKernelGGL((k_render_events_mix<G, MODE_PROCESS>), dim3(grid), dim3(256), 0, s, slots, stride, V, in, part, T, ev,
                               fade, time0, sr, aux, ring, ring_cap);
        else
            hipLaunchKernelGGL((k_render_events_mix<G, MODE_TICK>), dim3(grid), dim3(256), 0, s, slots, stride, V, in, part, T, ev,
                               fade, time0, sr, aux, ring, ring_cap);
        return true;
    } else {
        return false;
    }
}
// gives a kind its fused mix-down kernels (opt-in per kind: every instantiation is compile time and code size)
template <class G>
void attach_mix(std::vector<KindOps>& kinds, const char* name) {
    for (KindOps& k : kinds)
        if (k.name == name) {
            k.render_mix = &launch_render_mix<G>;
            k.render_events_mix = &launch_render_events_mix<G>;
            using GF = typename FastOf<G>::type;
            if constexpr (!SameType<GF, G>::v) k.render_mix_fast = &launch_render_mix<GF>;
        }
}

template <class G>
void launch_render_events(float* slots, size_t stride, size_t V, const float* in, float* out, size_t T, const double* ev,
                          const int* fade, double time0, double sr, int mode, const void* aux, float* ring,
                          uint32_t ring_cap, hipStream_t s) {
    if (V == 0 || T == 0) return;
    tl_opts.last_kernel = LK_EVENTS;
    const unsigned grid = (unsigned)(((V + 63) / 64 + 3) / 4);
    if (mode == MODE_PROCESS)
        hipLaunchKernelGGL((k_render_events<G, MODE_PROCESS>), dim3(grid), dim3(256), 0, s, slots, stride, V, in, out, T, ev,
                           fade, time0, sr, aux, ring, ring_cap);
    else
        hipLaunchKernelGGL((k_render_events<G, MODE_TICK>), dim3(grid), dim3(256), 0, s, slots, stride, V, in, out, T, ev,
                           fade, time0, sr, aux, ring, ring_cap);
}

template <class G>
KindOps make_kind(const char* name) {
    KindOps k;
    k.name = name;
    k.nin = G::IN;
    k.nout = G::OUT;
    k.nrings = G::RINGS;
    G g{};
    VDescribe d{&k.slots, {}};
    g.visit(d);
    k.lifecycle = &launch_lifecycle<G>;
    k.render = &launch_render<G>;
    using GF = typename FastOf<G>::type;
    if constexpr (!SameType<GF, G>::v) k.render_fast = &launch_render<GF>;
    k.render_events = &launch_render_events<G>;
    return k;
}

int jit_compile_code(const std::string& type_expr, const std::string& prelude, std::vector<char>* code, std::string* log);
int jit_make_kind(const std::string& name, const std::string& type_expr, const std::string& prelude, KindOps* out,
                  std::string* err);
int rust_translate(const char* rust_type_name, const char* hints, std::string* expr, std::string* presets);  // fd_rust.hip
void register_leaf_kinds(std::vector<KindOps>& out);
void register_graph_kinds(std::vector<KindOps>& out);

}  // namespace fd

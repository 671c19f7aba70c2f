// fd_jit.hip -- run-time compilation of arbitrary voice graphs (graph -> kernel compiler).
//
// FunDSP builds a graph as a statically typed combinator tree (An<Pipe<Pipe<Constant<U1>,Sine<f32>>,FixedSvf<..>>>);
// fd_nodes.hpp mirrors those types as device templates.  The ahead-of-time "kinds" instantiate a fixed set of them;
// this file instantiates ANY type expression over the same templates at run time: hiprtc compiles the very headers the
// library was built from (fd_math.hpp, fd_nodes.hpp, fd_device.hpp, read from fundsp_amd/csrc next to the .so) with the
// same flags (-O3, -ffp-contract=off), so a JIT kind is sample-identical to the same graph compiled ahead of time.
// No tracing, no IR: the "compiler" is the C++ template instantiation itself.
#ifndef _GNU_SOURCE
#define _GNU_SOURCE 1
#endif
#include <dlfcn.h>
#include <link.h>
#include <limits.h>
#include <stdlib.h>
#include <hip/hiprtc.h>

#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <cstdio>
#include <sstream>

#include "fd_engine.hpp"

namespace fd {

int jit_compile_src(const std::string& src, const std::string& type_expr, std::vector<char>* code, std::string* log);
std::string jit_source_mix(const std::string& type_expr, const std::string& prelude);

namespace {

std::string lib_dir() {
    Dl_info info;
    if (dladdr((void*)&lib_dir, &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        size_t k = p.find_last_of('/');
        return k == std::string::npos ? "." : p.substr(0, k);
    }
    return ".";
}

bool read_file(const std::string& path, std::string* out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::stringstream ss;
    ss << f.rdbuf();
    *out = ss.str();
    return true;
}

// ---- which compiler compiles the graphs ------------------------------------------------------------------------------------------
// "hiprtc compiles the very headers the library was built from, with the same flags" is only half of the promise: it must also be the COMPILER the
// library was built with.  Plain linkage does not guarantee that.  A host process may already hold another ROCm's libhiprtc / libamd_comgr under the
// same sonames -- a PyTorch wheel bundles its own (ROCm 7.0's, next to this image's 7.2) and loads them first --, the dynamic loader then satisfies
// this library's `libhiprtc.so.7` with THAT copy, and run-time compiled kinds come out of a different LLVM than the ahead-of-time ones.  Found the
// hard way: the bundled 7.0 compiler MISCOMPILES the planar single-wave kernel of `(mls() ^ impulse()) + c` (it allocates the Impulse's value and the
// already-dead `bits` of the Mls to one register and drops the copy that separates them: the first frame of the impulse comes out as bits + c);
// this ROCm's compiler, given the same source and flags, does not (tests/test_gpu_jit_compiler.py; the generated code of both is in
// profiles/r06_jit_compiler_miscompile.txt).
// So: when the process's hiprtc (or an already-loaded comgr) is not the file in the ROCm directory this library was built against, that ROCm's
// libhiprtc is loaded into a link-map namespace of its own (dlmopen: its dlopen("libamd_comgr.so.3") then resolves inside that namespace, to its own
// neighbour) and the compile calls go there.  FDSP_HIPRTC=linked keeps the process's copy, FDSP_HIPRTC=<path to a libhiprtc> names another one.
#ifndef FD_ROCM_LIB
#define FD_ROCM_LIB "/opt/rocm/lib"
#endif
struct Rtc {
    decltype(&hiprtcCreateProgram) create = &hiprtcCreateProgram;
    decltype(&hiprtcCompileProgram) compile = &hiprtcCompileProgram;
    decltype(&hiprtcGetCodeSize) code_size = &hiprtcGetCodeSize;
    decltype(&hiprtcGetCode) get_code = &hiprtcGetCode;
    decltype(&hiprtcGetProgramLogSize) log_size = &hiprtcGetProgramLogSize;
    decltype(&hiprtcGetProgramLog) get_log = &hiprtcGetProgramLog;
    decltype(&hiprtcGetErrorString) error_string = &hiprtcGetErrorString;
    decltype(&hiprtcDestroyProgram) destroy = &hiprtcDestroyProgram;
    std::string origin;   // what fdsp_jit_compiler() reports
};

std::string real_path(const std::string& p) {
    char buf[PATH_MAX];
    return realpath(p.c_str(), buf) ? std::string(buf) : p;
}

std::string loaded_path_of(void* symbol) {
    Dl_info info;
    return (dladdr(symbol, &info) && info.dli_fname) ? real_path(info.dli_fname) : std::string();
}

const Rtc& rtc() {
    static const Rtc instance = [] {
        Rtc r;
        const std::string linked = loaded_path_of((void*)&hiprtcCreateProgram);
        r.origin = "linked: " + linked;
        const char* env = getenv("FDSP_HIPRTC");
        if (env && !strcmp(env, "linked")) return r;
        const std::string want = real_path(env && *env ? std::string(env) : std::string(FD_ROCM_LIB) + "/libhiprtc.so.7");
        bool foreign = linked != want;
        if (!foreign) {  // the right hiprtc; but it looks its comgr up by soname, and a comgr of another ROCm may already answer to it
            if (void* c = dlopen("libamd_comgr.so.3", RTLD_NOLOAD | RTLD_LAZY)) {
                struct link_map* lm = nullptr;
                if (dlinfo(c, RTLD_DI_LINKMAP, &lm) == 0 && lm && lm->l_name && *lm->l_name) {
                    const std::string have = real_path(lm->l_name), dir = want.substr(0, want.find_last_of('/'));
                    foreign = have.compare(0, dir.size(), dir) != 0;
                }
                dlclose(c);
            }
        }
        if (!foreign) return r;
        void* h = dlmopen(LM_ID_NEWLM, want.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            r.origin += std::string(" (another ROCm's; isolating ") + want + " failed: " + dlerror() + ")";
            return r;
        }
        Rtc iso;
        bool ok = true;
        auto sym = [&](const char* name) { void* f = dlsym(h, name); ok = ok && f; return f; };
        iso.create = (decltype(iso.create))sym("hiprtcCreateProgram");
        iso.compile = (decltype(iso.compile))sym("hiprtcCompileProgram");
        iso.code_size = (decltype(iso.code_size))sym("hiprtcGetCodeSize");
        iso.get_code = (decltype(iso.get_code))sym("hiprtcGetCode");
        iso.log_size = (decltype(iso.log_size))sym("hiprtcGetProgramLogSize");
        iso.get_log = (decltype(iso.get_log))sym("hiprtcGetProgramLog");
        iso.error_string = (decltype(iso.error_string))sym("hiprtcGetErrorString");
        iso.destroy = (decltype(iso.destroy))sym("hiprtcDestroyProgram");
        if (!ok) {
            r.origin += " (another ROCm's; " + want + " lacks a hiprtc entry point)";
            return r;
        }
        iso.origin = "isolated: " + want + " (the process's own is " + linked + ")";
        return iso;
    }();
    return instance;
}

struct JitMix;

// One compiled graph: the code object plus, per HIP device, the module loaded on it (hipModule_t belongs to a device;
// a process may keep banks of the same kind on several GPUs, fdsp_bank_create_on).  Loaded lazily under a mutex.
struct JitFuncs {
    hipModule_t mod = nullptr;
    hipFunction_t lifecycle = nullptr, describe = nullptr;
    hipFunction_t render[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [mode][layout]
    hipFunction_t events[2] = {nullptr, nullptr};                            // [mode]
    hipFunction_t pipe[2] = {nullptr, nullptr};                              // [mode], pipeline kernel
    hipFunction_t pipe_small[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [groups per workgroup - 1][mode], heavy graphs only
    hipFunction_t pipe_planar[2] = {nullptr, nullptr};                       // [mode], planar-layout pipeline kernel
    hipFunction_t wide[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};      // [mode][layout], wide sums of generators on small banks (a chain of waves per voice group)
};
struct JitModule {
    static constexpr int MAXD = 64;
    std::vector<char> code;
    std::mutex mu;
    JitFuncs dev[MAXD];
    bool loaded[MAXD] = {false};
    int pipe_stages = 0, pipe_threads = 0, pipe_planar_threads = 0, pipe_min_t = 256;
    bool pipe_small = false;                                                 // heavy graph: workgroups of 1 / 2 voice groups for small banks
    bool ts_ok = false;                                                      // three-stage generator chain: small banks take the time-split kernels
    int wide_waves = 0;                                                      // a wide sum of generators: waves per voice group of the chain kernel (0 = none)
    int wpb[2] = {4, 4};                                                     // per layout
    // the tolerance-mode twin of this graph (FastOf<G>), compiled on first use
    std::string type_expr, prelude;
    bool has_fast = false, fast_failed = false;
    std::shared_ptr<JitModule> fast;
    // the fused mix-down kernels (of G, and of FastOf<G>), compiled on first use
    std::shared_ptr<JitMix> mix[2];
    bool mix_failed[2] = {false, false};
    std::mutex mix_mu;
    int nout = 0;
    ~JitModule() {  // a module is unloaded with ITS device current (it was loaded on that device's context)
        int prev = -1;
        const bool have_prev = hipGetDevice(&prev) == hipSuccess;
        for (int d = 0; d < MAXD; d++)
            if (loaded[d] && dev[d].mod && hipSetDevice(d) == hipSuccess) hipModuleUnload(dev[d].mod);
        if (have_prev) hipSetDevice(prev);
    }
    // functions of the CURRENT device (nullptr + *err on failure)
    const JitFuncs* get(std::string* err = nullptr) {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MAXD) {
            if (err) *err = "no current HIP device";
            return nullptr;
        }
        std::lock_guard<std::mutex> lock(mu);
        if (loaded[d]) return dev[d].mod ? &dev[d] : nullptr;
        loaded[d] = true;
        JitFuncs& f = dev[d];
        if (hipModuleLoadData(&f.mod, code.data()) != hipSuccess) {
            f.mod = nullptr;
            if (err) *err = "hipModuleLoadData failed for the compiled graph";
            return nullptr;
        }
        bool ok = hipModuleGetFunction(&f.lifecycle, f.mod, "jit_lifecycle") == hipSuccess &&
                  hipModuleGetFunction(&f.describe, f.mod, "jit_describe") == hipSuccess;
        for (int m = 0; m < 2 && ok; m++)
            for (int l = 0; l < 2 && ok; l++) {
                std::string fn = "jit_render_" + std::to_string(m) + std::to_string(l);
                ok = hipModuleGetFunction(&f.render[m][l], f.mod, fn.c_str()) == hipSuccess;
                fn = "jit_wide_" + std::to_string(m) + std::to_string(l);
                ok = ok && hipModuleGetFunction(&f.wide[m][l], f.mod, fn.c_str()) == hipSuccess;
            }
        for (int m = 0; m < 2 && ok; m++) {
            std::string fn = "jit_events_" + std::to_string(m);
            ok = hipModuleGetFunction(&f.events[m], f.mod, fn.c_str()) == hipSuccess;
            fn = "jit_pipe_" + std::to_string(m);
            ok = ok && hipModuleGetFunction(&f.pipe[m], f.mod, fn.c_str()) == hipSuccess;
            fn = "jit_pipe_planar_" + std::to_string(m);
            ok = ok && hipModuleGetFunction(&f.pipe_planar[m], f.mod, fn.c_str()) == hipSuccess;
            for (int g = 0; g < 2 && ok; g++) {
                fn = "jit_pipe_g" + std::to_string(g + 1) + "_" + std::to_string(m);
                ok = hipModuleGetFunction(&f.pipe_small[g][m], f.mod, fn.c_str()) == hipSuccess;
            }
        }
        if (!ok) {
            hipModuleUnload(f.mod);
            f.mod = nullptr;
            if (err) *err = "compiled graph is missing an entry point";
            return nullptr;
        }
        return &f;
    }
};

// The mix-down kernels of a compiled graph: code object + per-device module, like JitModule; fn[mix - 1][mode]
struct JitMix {
    std::vector<char> code;
    std::mutex mu;
    struct Dev { hipModule_t mod = nullptr; hipFunction_t fn[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}; hipFunction_t ev[2] = {nullptr, nullptr}; hipFunction_t ts[2] = {nullptr, nullptr}; hipFunction_t tsm[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}; bool loaded = false; } dev[JitModule::MAXD];
    ~JitMix() {
        int prev = -1;
        const bool have_prev = hipGetDevice(&prev) == hipSuccess;
        for (int d = 0; d < JitModule::MAXD; d++)
            if (dev[d].loaded && dev[d].mod && hipSetDevice(d) == hipSuccess) hipModuleUnload(dev[d].mod);
        if (have_prev) hipSetDevice(prev);
    }
    const Dev* get() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= JitModule::MAXD) return nullptr;
        std::lock_guard<std::mutex> lock(mu);
        Dev& f = dev[d];
        if (f.loaded) return f.mod ? &f : nullptr;
        f.loaded = true;
        if (hipModuleLoadData(&f.mod, code.data()) != hipSuccess) { f.mod = nullptr; return nullptr; }
        bool ok = true;
        for (int x = 0; x < 2 && ok; x++)
            for (int m = 0; m < 2 && ok; m++) {
                const std::string fn = "jit_pipe_mix_" + std::to_string(x + 1) + "_" + std::to_string(m);
                ok = hipModuleGetFunction(&f.fn[x][m], f.mod, fn.c_str()) == hipSuccess;
            }
        for (int m = 0; m < 2 && ok; m++) {
            const std::string fn = "jit_events_mix_" + std::to_string(m);
            ok = hipModuleGetFunction(&f.ev[m], f.mod, fn.c_str()) == hipSuccess;
        }
        for (int g = 0; g < 2 && ok; g++) {
            const std::string fn = "jit_ts3_g" + std::to_string(g + 1);
            ok = hipModuleGetFunction(&f.ts[g], f.mod, fn.c_str()) == hipSuccess;
            for (int x = 0; x < 2 && ok; x++) {
                const std::string fm = "jit_ts3_mix_g" + std::to_string(g + 1) + "_" + std::to_string(x + 1);
                ok = hipModuleGetFunction(&f.tsm[g][x], f.mod, fm.c_str()) == hipSuccess;
            }
        }
        if (!ok) { hipModuleUnload(f.mod); f.mod = nullptr; return nullptr; }
        return &f;
    }
};

// a launch that cannot happen (module not loadable on this device) must not look like success to the caller's
// hipGetLastError(): provoke an invalid-handle error
void jit_launch_failed() { hipModuleLaunchKernel(nullptr, 1, 1, 1, 1, 1, 1, 0, nullptr, nullptr, nullptr); }

// The second module of a compiled graph -- the fused mix-down kernels and the time-split kernels of G (which = 0) / FastOf<G> (1) --,
// compiled the first time a bank of the kind needs one of them.  Under a mutex of its own: a render of the kind on another thread does
// not wait for the compile.
JitMix* jit_extra_module(JitModule* jm, int which) {
    std::lock_guard<std::mutex> lock(jm->mix_mu);
    if (!jm->mix[which] && !jm->mix_failed[which]) {
        auto mm = std::make_shared<JitMix>();
        std::string log;
        const std::string type = which ? "typename fd::FastOf<" + jm->type_expr + ">::type" : jm->type_expr;
        if (jit_compile_src(jit_source_mix(type, jm->prelude), jm->type_expr, &mm->code, &log) == 0) jm->mix[which] = mm;
        else {
            jm->mix_failed[which] = true;
            fprintf(stderr, "fundsp_hip: the mix-down / time-split kernels of this graph failed to compile: %s\n", log.c_str());
        }
    }
    return jm->mix[which].get();
}

// `base` / `which`: the kind's module and which of its variants `jm` is (0 = G itself, 1 = its tolerance-mode twin) -- where the second module lives
void jit_render(JitModule* jm, float* slots, size_t stride, size_t V, const float* in, float* outp, size_t T, size_t fstride,
                int layout, int mode, const void* aux, float* ring, uint32_t ring_cap, hipStream_t s, JitModule* base = nullptr, int which = 0) {
    if (V == 0 || T == 0) return;
    const JitFuncs* f = jm->get();
    if (!f) return jit_launch_failed();
    // banks that leave most SIMDs idle (<= 2 voice groups per CU) of three-stage generator chains: the three-way time split, as launch_render
    // does for the ahead-of-time kinds (its kernels are compiled the first time such a launch happens; no such kernels -> the pipeline below)
    if (base && base->ts_ok && tl_opts.time_split == 1 && tl_opts.pipe_split == 1 && layout == LAYOUT_VOICE_MINOR && mode == MODE_PROCESS &&
        T % 64 == 0 && T >= FD_TS_MIN_T) {
        const size_t groups = (V + 63) / 64, cus = (size_t)simd_count() / 4;
        if (groups <= 2 * cus) {
            JitMix* mm = jit_extra_module(base, which);
            const JitMix::Dev* tf = mm ? mm->get() : nullptr;
            if (tf) {
                const int gpw = groups <= cus ? 1 : 2;
                void* targs[] = {&slots, &stride, &V, &outp, &T, &aux};
                hipModuleLaunchKernel(tf->ts[gpw - 1], (unsigned)((groups + gpw - 1) / gpw), 1, 1, 64u * (gpw == 1 ? Ts3Roles<1>::WAVES : Ts3Roles<2>::WAVES), 1, 1, 0, s,
                                      targs, nullptr);
                tl_opts.last_kernel = LK_TIME_SPLIT;
                return;
            }
        }
    }
    // A wide sum of generators at the root (fd_device.hpp WideSum): its kernels are the branch-major ones, whatever the layout -- the chain of
    // waves (render_body_wide_chain) for every launch of more than one block: it fills the chip from 256 voice groups on where one wave per
    // voice group needs 1 024, and at 1 024 groups it still wins by its two waves per SIMD (the reference's 100-sine bench, ms per rendered
    // second at 64 / 1 024 / 16 384 / 32 768 / 49 152 / 65 536 instances: 41 / 41 / 43 / 85 / 128 / 171 against 169 / 233 / 181 / 180 / 313 / 181,
    // profiles/r06_wide_chain_probe.txt); one-block launches and "pipe_split" 0: render_body_wide through the single-wave entry point below.
    // The stage pipelines walk such a graph frame-major with every branch in registers: never.
    const bool wide = jm->wide_waves > 0;
    if (wide && tl_opts.pipe_split && T > 64) {
        void* wargs[] = {&slots, &stride, &V, &in, &outp, &T, &fstride, &aux, &ring, &ring_cap};
        hipModuleLaunchKernel(f->wide[mode][layout], (unsigned)((V + 63) / 64), 1, 1, 64u * (unsigned)jm->wide_waves, 1, 1, 0, s, wargs, nullptr);
        tl_opts.last_kernel = LK_WIDE_CHAIN;
        return;
    }
    // loader wave / stage split; short launches (real-time blocks) are faster through the single-wave kernel, as for
    // the ahead-of-time kinds (launch_render)
    if (!wide && layout == LAYOUT_VOICE_MINOR && tl_opts.pipe_split && jm->pipe_stages >= 1 && (T >= (size_t)jm->pipe_min_t || tl_opts.pipe_split > 1)) {
        void* pargs[] = {&slots, &stride, &V, &in, &outp, &T, &aux, &ring, &ring_cap};
        const size_t groups = (V + 63) / 64, cus = (size_t)simd_count() / 4;
        if (jm->pipe_small && groups <= 2 * cus) {  // heavy graph, small bank: as launch_render_pipe
            const unsigned gpw = groups <= cus ? 1 : 2;
            hipModuleLaunchKernel(f->pipe_small[gpw - 1][mode], (unsigned)((groups + gpw - 1) / gpw), 1, 1,
                                  (unsigned)jm->pipe_threads / 4 * gpw, 1, 1, 0, s, pargs, nullptr);
            tl_opts.last_kernel = LK_PIPELINE;
            return;
        }
        hipModuleLaunchKernel(f->pipe[mode], (unsigned)((groups + 3) / 4), 1, 1, (unsigned)jm->pipe_threads, 1, 1, 0, s,
                              pargs, nullptr);
        tl_opts.last_kernel = LK_PIPELINE;
        return;
    }
    if (!wide && layout == LAYOUT_PLANAR && tl_opts.pipe_split && jm->pipe_planar_threads > 0 && (T >= FD_PLANAR_PIPE_MIN_T || tl_opts.pipe_split > 1) && fstride % 4 == 0 &&
        ((uintptr_t)in & 15) == 0 && ((uintptr_t)outp & 15) == 0) {  // loader / stages / storer (see launch_render)
        void* pargs[] = {&slots, &stride, &V, &in, &outp, &T, &fstride, &aux, &ring, &ring_cap};
        hipModuleLaunchKernel(f->pipe_planar[mode], (unsigned)(((V + 63) / 64 + 3) / 4), 1, 1, (unsigned)jm->pipe_planar_threads, 1, 1,
                              0, s, pargs, nullptr);
        tl_opts.last_kernel = LK_PIPELINE_PLANAR;
        return;
    }
    tl_opts.last_kernel = LK_SINGLE_WAVE;
    const int wpb = jm->wpb[layout];
    const int vpw = layout == LAYOUT_VOICE_MINOR ? voices_per_wave(V, simd_count()) : 64;
    if (layout == LAYOUT_VOICE_MINOR) fstride = (size_t)vpw;
    const size_t waves = (V + vpw - 1) / vpw;
    void* args[] = {&slots, &stride, &V, &in, &outp, &T, &fstride, &aux, &ring, &ring_cap};
    hipModuleLaunchKernel(f->render[mode][layout], (unsigned)((waves + wpb - 1) / wpb), 1, 1, 64 * wpb, 1, 1, 0, s, args, nullptr);
}

}  // namespace

std::string jit_source(const std::string& type_expr, const std::string& prelude) {
    std::string s;
    s += "#include \"fd_device.hpp\"\n";
    // user-supplied node / functor definitions (e.g. the closure of an Envelope) live in namespace fd like the built-ins
    if (!prelude.empty()) s += "namespace fd {\n" + prelude + "\n}\n";
    s += "namespace fd { using JitG = " + type_expr + "; }\n";
    s += "using fd::JitG;\n";
    s += "constexpr int JIT_WPB0 = fd::RenderGeom<JitG, 0>::WPB;\nconstexpr int JIT_WPB1 = fd::RenderGeom<JitG, 1>::WPB;\n";
    s += "extern \"C\" __global__ __launch_bounds__(64) void jit_lifecycle(float* slots, size_t stride, size_t first, "
         "size_t count, int op, double sr, const uint64_t* seeds, const void* aux, float* ring, uint32_t cap) {\n"
         "  fd::lifecycle_body<JitG>(slots, stride, first, count, op, sr, seeds, aux, ring, cap); }\n";
    for (int mode = 0; mode < 2; mode++)
        for (int layout = 0; layout < 2; layout++) {
            std::string m = std::to_string(mode), l = std::to_string(layout);
            s += "extern \"C\" __global__ __launch_bounds__(64 * JIT_WPB" + l + ") void jit_render_" + m + l +
                 "(float* __restrict__ slots, size_t stride, size_t V, const float* __restrict__ in, float* __restrict__ out, "
                 "size_t T, size_t fstride, const void* aux, float* ring, uint32_t cap) {\n"
                 "  fd::render_body<JitG, " + m + ", " + l + ", JIT_WPB" + l +
                 ">(slots, stride, V, in, out, T, fstride, aux, ring, cap); }\n";
        }
    for (int mode = 0; mode < 2; mode++)   // wide sums of generators on small banks: a chain of waves per voice group (empty bodies for every other graph)
        for (int layout = 0; layout < 2; layout++) {
            std::string m = std::to_string(mode), l = std::to_string(layout);
            s += "extern \"C\" __global__ __launch_bounds__(64 * fd::WideChain<JitG>::W) void jit_wide_" + m + l +
                 "(float* __restrict__ slots, size_t stride, size_t V, const float* __restrict__ in, float* __restrict__ out, size_t T, size_t fstride, "
                 "const void* aux, float* ring, uint32_t cap) {\n  fd::render_body_wide_chain<JitG, " + m + ", " + l + ">(slots, stride, V, in, out, T, fstride, aux, ring, cap); }\n";
        }
    s += "constexpr int JIT_PIPE_THREADS = fd::JitPipeThreads<JitG>::v;\n";
    for (int mode = 0; mode < 2; mode++) {
        std::string m = std::to_string(mode);
        s += "extern \"C\" __global__ __launch_bounds__(JIT_PIPE_THREADS) void jit_pipe_" + m +
             "(float* __restrict__ slots, size_t stride, size_t V, const float* __restrict__ in, float* __restrict__ out, "
             "size_t T, const void* aux, float* ring, uint32_t cap) {\n"
             "  fd::jit_pipe_body<JitG, " + m + ">(slots, stride, V, in, out, T, aux, ring, cap); }\n";
    }
    for (int g = 1; g <= 2; g++)
        for (int mode = 0; mode < 2; mode++) {
            std::string m = std::to_string(mode), gs = std::to_string(g);
            s += "extern \"C\" __global__ __launch_bounds__(fd::JitPipeSmall<JitG>::threads<" + gs + ">()) void jit_pipe_g" + gs + "_" + m +
                 "(float* __restrict__ slots, size_t stride, size_t V, const float* __restrict__ in, float* __restrict__ out, "
                 "size_t T, const void* aux, float* ring, uint32_t cap) {\n"
                 "  fd::jit_pipe_small_body<JitG, " + m + ", " + gs + ">(slots, stride, V, in, out, T, aux, ring, cap); }\n";
        }
    s += "constexpr int JIT_PIPE_PLANAR_THREADS = fd::JitPipePlanarThreads<JitG>::v;\n";
    for (int mode = 0; mode < 2; mode++) {
        std::string m = std::to_string(mode);
        s += "extern \"C\" __global__ __launch_bounds__(JIT_PIPE_PLANAR_THREADS) void jit_pipe_planar_" + m +
             "(float* __restrict__ slots, size_t stride, size_t V, const float* __restrict__ in, float* __restrict__ out, "
             "size_t T, size_t fstride, const void* aux, float* ring, uint32_t cap) {\n"
             "  fd::jit_pipe_planar_body<JitG, " + m + ">(slots, stride, V, in, out, T, fstride, aux, ring, cap); }\n";
    }
    for (int mode = 0; mode < 2; mode++) {
        std::string m = std::to_string(mode);
        s += "extern \"C\" __global__ __launch_bounds__(256) void jit_events_" + m +
             "(float* __restrict__ slots, size_t stride, size_t V, const float* __restrict__ in, float* __restrict__ out, "
             "size_t T, const double* __restrict__ ev, const int* __restrict__ fade, double time0, double sr, const void* aux, "
             "float* ring, uint32_t cap) {\n"
             "  fd::render_events_body<JitG, " + m + ">(slots, stride, V, in, out, T, ev, fade, time0, sr, aux, ring, cap); }\n";
    }
    s += "extern \"C\" __global__ void jit_describe(char* out, int cap, int* meta) { fd::describe_body<JitG>(out, cap, meta); }\n";
    return s;
}

// the fused mix-down kernels of a graph (fdsp_bank_process_mix): a module of their own, compiled the first time a bank of the kind mixes
std::string jit_source_mix(const std::string& type_expr, const std::string& prelude) {
    std::string s;
    s += "#include \"fd_device.hpp\"\n";
    if (!prelude.empty()) s += "namespace fd {\n" + prelude + "\n}\n";
    s += "namespace fd { using JitG = " + type_expr + "; }\nusing fd::JitG;\n";
    s += "constexpr int JIT_PIPE_THREADS = fd::JitPipeThreads<JitG>::v;\n";
    for (int mix = 1; mix <= 2; mix++)
        for (int mode = 0; mode < 2; mode++) {
            std::string m = std::to_string(mode), x = std::to_string(mix);
            s += "extern \"C\" __global__ __launch_bounds__(JIT_PIPE_THREADS) void jit_pipe_mix_" + x + "_" + m +
                 "(float* __restrict__ slots, size_t stride, size_t V, const float* __restrict__ in, float* __restrict__ part, "
                 "size_t T, const void* aux, float* ring, uint32_t cap, const float* __restrict__ panw) {\n"
                 "  fd::jit_pipe_mix_body<JitG, " + m + ", " + x + ">(slots, stride, V, in, part, T, aux, ring, cap, panw); }\n";
        }
    for (int g = 1; g <= 2; g++) {  // the three-way time-split kernels (voice-out, process mode): small banks of three-stage generator chains
        std::string gs = std::to_string(g);
        s += "extern \"C\" __global__ __launch_bounds__(64 * fd::Ts3Roles<" + gs + ">::WAVES) void jit_ts3_g" + gs +
             "(float* __restrict__ slots, size_t stride, size_t V, float* __restrict__ out, size_t T, const void* aux) {\n"
             "  fd::jit_ts3_body<JitG, " + gs + ">(slots, stride, V, out, T, aux); }\n";
        for (int mix = 1; mix <= 2; mix++) {
            std::string x = std::to_string(mix);
            s += "extern \"C\" __global__ __launch_bounds__(64 * fd::Ts3Roles<" + gs + ">::WAVES) void jit_ts3_mix_g" + gs + "_" + x +
                 "(float* __restrict__ slots, size_t stride, size_t V, float* __restrict__ part, size_t T, const void* aux, const float* __restrict__ panw) {\n"
                 "  fd::jit_ts3_mix_body<JitG, " + gs + ", " + x + ">(slots, stride, V, part, T, aux, panw); }\n";
        }
    }
    for (int mode = 0; mode < 2; mode++) {
        std::string m = std::to_string(mode);
        s += "extern \"C\" __global__ __launch_bounds__(256) void jit_events_mix_" + m +
             "(float* __restrict__ slots, size_t stride, size_t V, const float* __restrict__ in, float* __restrict__ part, "
             "size_t T, const double* __restrict__ ev, const int* __restrict__ fade, double time0, double sr, const void* aux, "
             "float* ring, uint32_t cap) {\n"
             "  fd::jit_events_mix_body<JitG, " + m + ">(slots, stride, V, in, part, T, ev, fade, time0, sr, aux, ring, cap); }\n";
    }
    return s;
}

// Compile only (no device needed): returns the code object or an error log.
int jit_compile_code(const std::string& type_expr, const std::string& prelude, std::vector<char>* code, std::string* log) {
    return jit_compile_src(jit_source(type_expr, prelude), type_expr, code, log);
}

const char* jit_compiler_origin() { return rtc().origin.c_str(); }

int jit_compile_src(const std::string& src, const std::string& type_expr, std::vector<char>* code, std::string* log) {
    const std::string dir = lib_dir() + "/csrc/";
    const char* names[3] = {"fd_math.hpp", "fd_nodes.hpp", "fd_device.hpp"};
    std::string hdr[3];
    for (int i = 0; i < 3; i++)
        if (!read_file(dir + names[i], &hdr[i])) {
            *log = "cannot read " + dir + names[i] + " (the JIT compiles the engine's own headers)";
            return -1;
        }
    const char* hsrc[3] = {hdr[0].c_str(), hdr[1].c_str(), hdr[2].c_str()};
    hiprtcProgram prog;
    if (rtc().create(&prog, src.c_str(), "fdsp_jit_graph.hip", 3, hsrc, names) != HIPRTC_SUCCESS) {
        *log = "hiprtcCreateProgram failed";
        return -1;
    }
    // same code-generation flags as the ahead-of-time kinds (see Makefile for -fno-slp-vectorize)
    // a graph with a Feedback node renders with f32 denormals flushed, like the reference after Feedback::new's
    // prevent_denormals() (feedback.rs:96, denormal.rs:18)
    const bool ftz = type_expr.find("Feedback") != std::string::npos;
    // A wide sum of plain oscillators / filters at the head of the graph (fd_device.hpp render_body_wide: 32 independent frame pairs per branch and
    // block): the default machine scheduler emits the packed sine polynomials pair by pair with an s_nop behind every dependent packed instruction
    // (328 per branch-block of the reference's 100-sine bench); the max-ILP strategy interleaves the pairs (4 s_nop, 160 -> 200 VGPRs).  Only for
    // graphs made ENTIRELY of the plain nodes it was tried on -- other strategies have crashed this compiler on other kinds.
    bool wide_ilp = type_expr.find("Reduce<") != std::string::npos || type_expr.find("MultiBus<") != std::string::npos;
    if (wide_ilp) {
        static const char* plain[] = {"Reduce", "MultiBus", "Pipe", "Unop", "Binop", "Stack", "Constant", "Sine", "Pass", "MultiPass", "Panner", "FixedSvf",
                                      "Noise", "Resonator", "OpAdd", "OpSub", "OpMul", "UNeg", "UAddScalar", "UNegAddScalar", "UMulScalar"};
        for (size_t i = 0; i < type_expr.size() && wide_ilp;) {
            if (isalpha((unsigned char)type_expr[i]) || type_expr[i] == '_') {
                size_t j = i;
                while (j < type_expr.size() && (isalnum((unsigned char)type_expr[j]) || type_expr[j] == '_')) j++;
                const std::string tok = type_expr.substr(i, j - i);
                bool known = false;
                for (const char* p : plain) known = known || tok == p;
                wide_ilp = known;
                i = j;
            } else {
                i++;
            }
        }
    }
    std::vector<const char*> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize"};
    if (ftz) {
        opts.push_back("-fgpu-flush-denormals-to-zero");
        opts.push_back("-DFD_FTZ=1");  // FD_FTZ: fd_math.hpp's flush-only selects
    }
    if (wide_ilp) {
        opts.push_back("-mllvm");
        opts.push_back("-amdgpu-sched-strategy=max-ilp");
    }
    hiprtcResult r = rtc().compile(prog, (int)opts.size(), opts.data());
    if (const char* dump = getenv("FDSP_JIT_DUMP")) {  // debugging aid: the generated source and code object of every compiled module, numbered
        static int dump_n = 0;
        const std::string base = std::string(dump) + "/jit_" + std::to_string(dump_n++);
        if (FILE* f = fopen((base + ".hip").c_str(), "w")) { fputs(("// " + type_expr + "\n" + src).c_str(), f); fclose(f); }
        size_t n = 0;
        if (r == HIPRTC_SUCCESS && rtc().code_size(prog, &n) == HIPRTC_SUCCESS) {
            std::vector<char> co(n);
            rtc().get_code(prog, co.data());
            if (FILE* f = fopen((base + ".co").c_str(), "wb")) { fwrite(co.data(), 1, n, f); fclose(f); }
        }
    }
    size_t ls = 0;
    rtc().log_size(prog, &ls);
    if (ls > 1) {
        log->resize(ls);
        rtc().get_log(prog, &(*log)[0]);
    }
    if (r != HIPRTC_SUCCESS) {
        *log = "hiprtc: " + std::string(rtc().error_string(r)) + " while compiling graph type `" + type_expr + "`:\n" + *log;
        rtc().destroy(&prog);
        return -1;
    }
    size_t cs = 0;
    rtc().code_size(prog, &cs);
    code->resize(cs);
    rtc().get_code(prog, code->data());
    rtc().destroy(&prog);
    return 0;
}

int jit_make_kind(const std::string& name, const std::string& type_expr, const std::string& prelude, KindOps* out,
                  std::string* err) {
    std::vector<char> code;
    if (jit_compile_code(type_expr, prelude, &code, err) != 0) return -1;
    auto jm = std::make_shared<JitModule>();
    jm->code = std::move(code);
    jm->type_expr = type_expr;
    jm->prelude = prelude;
    const JitFuncs* f0 = jm->get(err);
    if (!f0) return -1;
    // slot introspection on the device (the AOT kinds run the same visit() on the host)
    const int cap = 1 << 16;
    char* d_txt = nullptr;
    int* d_meta = nullptr;
    if (hipMalloc((void**)&d_txt, cap) != hipSuccess || hipMalloc((void**)&d_meta, 16 * sizeof(int)) != hipSuccess) {
        *err = "hipMalloc failed";
        return -1;
    }
    int cap_arg = cap;
    void* dargs[] = {&d_txt, &cap_arg, &d_meta};
    hipError_t e = hipModuleLaunchKernel(f0->describe, 1, 1, 1, 1, 1, 1, 0, nullptr, dargs, nullptr);
    std::vector<char> txt(cap);
    int meta[16] = {0};
    if (e == hipSuccess) e = hipMemcpy(txt.data(), d_txt, cap, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(meta, d_meta, sizeof meta, hipMemcpyDeviceToHost);
    hipFree(d_txt);
    hipFree(d_meta);
    if (e != hipSuccess) {
        *err = std::string("describe kernel failed: ") + hipGetErrorString(e);
        return -1;
    }
    out->name = name;
    out->nin = meta[0];
    out->nout = meta[1];
    out->nrings = meta[2];
    jm->wpb[LAYOUT_VOICE_MINOR] = 4;
    jm->wpb[LAYOUT_PLANAR] = meta[4];
    jm->pipe_stages = meta[5];
    jm->pipe_threads = meta[6];
    jm->pipe_planar_threads = meta[7];
    jm->has_fast = meta[8] != 0;
    jm->pipe_small = meta[9] != 0;
    jm->pipe_min_t = meta[10] > 0 ? meta[10] : 256;
    jm->ts_ok = meta[11] != 0;
    jm->wide_waves = meta[12];
    out->slots.clear();
    std::istringstream lines(std::string(txt.data(), (size_t)meta[3]));
    std::string line;
    while (std::getline(lines, line)) {
        size_t sp = line.find_last_of(' ');
        if (sp == std::string::npos) continue;
        out->slots.push_back({line.substr(0, sp), std::atoi(line.c_str() + sp + 1)});
    }
    out->lifecycle = [jm](float* slots, size_t stride, size_t first, size_t count, int op, double sr,
                          const uint64_t* d_seeds, const void* aux, float* ring, uint32_t ring_cap, hipStream_t s) {
        if (count == 0) return;
        const JitFuncs* f = jm->get();
        if (!f) return jit_launch_failed();
        void* args[] = {&slots, &stride, &first, &count, &op, &sr, &d_seeds, &aux, &ring, &ring_cap};
        hipModuleLaunchKernel(f->lifecycle, (unsigned)((count + 63) / 64), 1, 1, 64, 1, 1, 0, s, args, nullptr);
    };
    out->render = [jm](float* slots, size_t stride, size_t V, const float* in, float* outp, size_t T, size_t fstride,
                       int layout, int mode, const void* aux, float* ring, uint32_t ring_cap, hipStream_t s) {
        jit_render(jm.get(), slots, stride, V, in, outp, T, fstride, layout, mode, aux, ring, ring_cap, s, jm.get(), 0);
    };
    // tolerance mode: the same source with JitG = FastOf<G>, a module of its own -- compiled when a FAST bank of the kind is created or a
    // bank is switched to FAST (prepare_render), at the latest the first time a FAST bank renders
    auto fast_module = [jm]() -> JitModule* {
        std::lock_guard<std::mutex> lock(jm->mu);
        if (!jm->fast && !jm->fast_failed) {
            auto fm = std::make_shared<JitModule>();
            std::string log;
            if (jit_compile_code("typename fd::FastOf<" + jm->type_expr + ">::type", jm->prelude, &fm->code, &log) == 0) {
                fm->pipe_stages = jm->pipe_stages;   // FastOf keeps arities, chain shape and tile plan
                fm->pipe_min_t = jm->pipe_min_t;
                fm->pipe_threads = jm->pipe_threads;
                fm->pipe_planar_threads = jm->pipe_planar_threads;
                fm->pipe_small = jm->pipe_small;
                fm->wide_waves = jm->wide_waves;
                fm->wpb[0] = jm->wpb[0];
                fm->wpb[1] = jm->wpb[1];
                jm->fast = fm;
            } else {
                jm->fast_failed = true;
                fprintf(stderr, "fundsp_hip: tolerance-mode variant failed to compile, rendering exactly: %s\n", log.c_str());
            }
        }
        return jm->fast.get();
    };
    // Whatever a bank of `voices` voices still has to compile and load before its first render: the tolerance-mode module of a FAST bank, and
    // for small banks of a three-stage generator chain the second module with the time-split kernels (of G, or of FastOf<G> when that twin
    // exists -- the `which` jit_render will ask for).  Called when the bank is created / switched, not in its first render.
    if (jm->ts_ok || jm->has_fast)
        out->prepare_render = [jm, fast_module](size_t voices, bool fast) {
            JitModule* fm = (fast && jm->has_fast) ? fast_module() : nullptr;
            if (fm) fm->get();
            if (jm->ts_ok && (voices + 63) / 64 <= 2 * (size_t)simd_count() / 4)
                if (JitMix* mm = jit_extra_module(jm.get(), fm ? 1 : 0)) mm->get();
        };
    if (jm->has_fast)
        out->render_fast = [jm, fast_module](float* slots, size_t stride, size_t V, const float* in, float* outp, size_t T, size_t fstride,
                                             int layout, int mode, const void* aux, float* ring, uint32_t ring_cap, hipStream_t s) {
            JitModule* fm = fast_module();
            jit_render(fm ? fm : jm.get(), slots, stride, V, in, outp, T, fstride, layout, mode, aux, ring, ring_cap, s, jm.get(), fm ? 1 : 0);
        };
    // render + mix-down in one launch (fdsp_bank_process_mix): graphs with a pipeline plan; the kernels are compiled on first use
    jm->nout = meta[1];
    auto mix_module = [jm](int which) -> JitMix* { return jit_extra_module(jm.get(), which); };
    auto mix_launch = [jm, mix_module](int which, float* slots, size_t stride, size_t V, const float* in, float* part, size_t T, int mix, int mode,
                           const void* aux, float* ring, uint32_t ring_cap, const float* panw, hipStream_t s) -> bool {
        if (V == 0 || T == 0) return true;
        if (jm->pipe_stages < 1 || (mix == MIX_PAN && jm->nout != 1) || (mix == MIX_SUM && !jit_mix_channels_ok(jm->nout))) return false;
        JitMix* mm = mix_module(which);
        if (!mm) return false;
        const JitMix::Dev* f = mm->get();
        if (!f) { jit_launch_failed(); return true; }
        // small banks of three-stage generator chains: the time-split kernels with the fused mix-down, as launch_render_mix_m
        if (jm->ts_ok && jm->nout <= 2 && tl_opts.time_split == 1 && tl_opts.pipe_split == 1 && mode == MODE_PROCESS && T % 64 == 0 && T >= FD_TS_MIN_T) {
            const size_t groups = (V + 63) / 64, cus = (size_t)simd_count() / 4;
            if (groups <= 2 * cus) {
                const int gpw = groups <= cus ? 1 : 2;
                void* targs[] = {&slots, &stride, &V, &part, &T, &aux, &panw};
                hipModuleLaunchKernel(f->tsm[gpw - 1][mix - 1], (unsigned)((groups + gpw - 1) / gpw), 1, 1, 64u * (gpw == 1 ? Ts3Roles<1>::WAVES : Ts3Roles<2>::WAVES), 1, 1, 0, s,
                                      targs, nullptr);
                tl_opts.last_kernel = LK_TIME_SPLIT;
                return true;
            }
        }
        void* pargs[] = {&slots, &stride, &V, &in, &part, &T, &aux, &ring, &ring_cap, &panw};
        hipModuleLaunchKernel(f->fn[mix - 1][mode], (unsigned)(((V + 63) / 64 + 3) / 4), 1, 1, (unsigned)jm->pipe_threads, 1, 1, 0, s, pargs, nullptr);
        tl_opts.last_kernel = LK_PIPELINE;
        return true;
    };
    if (jm->pipe_stages >= 1 && (jm->nout == 1 || jit_mix_channels_ok(jm->nout))) {   // ("has_fused_mix" follows: no kernels, no entry)
        out->prepare_mix = [mix_module](bool fast) {   // compile + load ahead of the real-time loop / stream capture
            if (JitMix* mm = mix_module(fast ? 1 : 0)) mm->get();
        };
        out->render_mix = [mix_launch](float* slots, size_t stride, size_t V, const float* in, float* part, size_t T, int mix, int mode,
                                       const void* aux, float* ring, uint32_t ring_cap, const float* panw, hipStream_t s) {
            return mix_launch(0, slots, stride, V, in, part, T, mix, mode, aux, ring, ring_cap, panw, s);
        };
        if (jm->has_fast)
            out->render_mix_fast = [mix_launch](float* slots, size_t stride, size_t V, const float* in, float* part, size_t T, int mix, int mode,
                                                const void* aux, float* ring, uint32_t ring_cap, const float* panw, hipStream_t s) {
                return mix_launch(1, slots, stride, V, in, part, T, mix, mode, aux, ring, ring_cap, panw, s);
            };
    }
    if (jm->nout <= 2)  // the Sequencer's mixed output in one launch (fdsp_bank_process_events_mix)
        out->render_events_mix = [mix_module](float* slots, size_t stride, size_t V, const float* in, float* part, size_t T, const double* ev,
                                              const int* fade, double time0, double sr, int mode, const void* aux, float* ring,
                                              uint32_t ring_cap, hipStream_t s) -> bool {
            if (V == 0 || T == 0) return true;
            JitMix* mm = mix_module(0);
            if (!mm) return false;
            const JitMix::Dev* f = mm->get();
            if (!f) { jit_launch_failed(); return true; }
            void* args[] = {&slots, &stride, &V, &in, &part, &T, &ev, &fade, &time0, &sr, &aux, &ring, &ring_cap};
            hipModuleLaunchKernel(f->ev[mode], (unsigned)(((V + 63) / 64 + 3) / 4), 1, 1, 256, 1, 1, 0, s, args, nullptr);
            tl_opts.last_kernel = LK_EVENTS;
            return true;
        };
    out->render_events = [jm](float* slots, size_t stride, size_t V, const float* in, float* outp, size_t T,
                              const double* ev, const int* fade, double time0, double sr, int mode, const void* aux,
                              float* ring, uint32_t ring_cap, hipStream_t s) {
        if (V == 0 || T == 0) return;
        const JitFuncs* f = jm->get();
        if (!f) return jit_launch_failed();
        void* args[] = {&slots, &stride, &V, &in, &outp, &T, &ev, &fade, &time0, &sr, &aux, &ring, &ring_cap};
        hipModuleLaunchKernel(f->events[mode], (unsigned)(((V + 63) / 64 + 3) / 4), 1, 1, 256, 1, 1, 0, s, args, nullptr);
    };
    return 0;
}

}  // namespace fd

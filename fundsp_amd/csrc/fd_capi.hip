// fd_capi.hip -- C ABI (include/fundsp_hip.h) of the MI355X voice-bank engine.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <unordered_map>
#include <limits>
#include <vector>

#include "../../include/fundsp_hip.h"
#include "fd_engine.hpp"
#include "fd_fdn.hpp"
#include "fd_reverb3.hpp"
#include "fd_opts.hpp"

namespace {

thread_local std::string g_err;
}
namespace fd {
std::string jit_source_mix(const std::string& type_expr, const std::string& prelude);  // fd_jit.hip: the second module of a run-time compiled graph
int jit_compile_src(const std::string& src, const std::string& type_expr, std::vector<char>* code, std::string* log);
const char* jit_compiler_origin();
}  // namespace fd
namespace fd {
// process-wide DEFAULTS (fdsp_set_option); a bank's own value (fdsp_bank_set_option) overrides them for that bank.  Atomics:
// hosts drive banks from several threads.  The launch code never reads these: it reads the thread-local block below.
std::atomic<int> g_pipe_split{1}, g_fdn_kernel{0}, g_time_split{1};
std::atomic<int> g_math{FDSP_MATH_EXACT};  // default arithmetic of banks created from now on (fdsp_set_option("math", ..))
std::atomic<int> g_timing{1};              // default of the per-launch HIP event pair (fdsp_bank_last_kernel_ms)
std::atomic<long> g_zero_copy_max{1 << 18};  // floats; fdsp_bank_process_host reads/writes pinned host memory directly below this
thread_local LaunchOpts tl_opts;           // fd_opts.hpp: what the launch code reads, resolved per bank by every render entry point
int simd_count() {  // SIMDs (CUs x 4) of the CURRENT device, cached per device
    static int cache[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 1024;
    if (cache[dev] == 0) {
        int cus = 0;
        cache[dev] = (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) ? cus * 4 : 1024;
    }
    return cache[dev];
}
}  // namespace fd
namespace {

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
}  // namespace
namespace fd {
int api_fail(int code, const std::string& msg) { return fail(code, msg); }  // for the other translation units of the C ABI
}
namespace {
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(e_ == hipErrorOutOfMemory ? FDSP_ENOMEM : FDSP_EDEVICE,                        \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                            \
    } while (0)

// Kinds are never removed and banks keep pointers into the table, so entries live in a deque (stable addresses);
// run-time compiled graphs (fd_jit.cpp) append to it.
std::deque<fd::KindOps>& registry() {
    static std::deque<fd::KindOps> kinds;
    static std::once_flag once;
    std::call_once(once, [] {
        std::vector<fd::KindOps> tmp;
        fd::register_leaf_kinds(tmp);
        fd::register_graph_kinds(tmp);
        for (auto& k : tmp) kinds.push_back(std::move(k));
    });
    return kinds;
}
std::mutex g_registry_mutex;
// element `kind` of the registry, or nullptr; entries never move (deque) and are never removed, so the pointer stays
// valid after the lock is released
const fd::KindOps* kind_at(int kind) {
    auto& r = registry();
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    return (kind >= 0 && kind < (int)r.size()) ? &r[kind] : nullptr;
}

// ---- shared data (wavetables, sample buffers): the counterpart of FunDSP's static Arc<Wavetable> / Arc<Wave> singletons.
// The host keeps ONE copy per set / slot; every device that has banks gets its own fd::Aux block with device copies,
// brought up to date when a bank is created on it and whenever a set / slot is (re)installed.  Nothing here is bound to
// "the first device": one process drives any number of GPUs (fdsp_bank_create_on).
constexpr int MAX_DEVICES = 64;
struct HostTableSet { int n = 0; std::vector<float> pitch; std::vector<int> len; std::vector<float> data; uint64_t ver = 0; };
struct HostWave { int channels = 0; size_t length = 0; std::vector<float> data; uint64_t ver = 0; };
struct DeviceCtx {
    bool init = false;
    fd::Aux host_aux;            // host mirror of this device's Aux (data pointers are device pointers)
    fd::Aux* dev_aux = nullptr;
    uint64_t table_ver[fd::WT_SETS] = {0}, wave_ver[fd::WAVE_SLOTS] = {0};
    bool aux_dirty = false;            // host_aux is newer than *dev_aux (an upload of the block itself failed earlier)
    std::vector<void*> retired;        // device buffers *dev_aux may still point to: freed only after the block was rewritten
};
HostTableSet g_tables[fd::WT_SETS];
HostWave g_waves[fd::WAVE_SLOTS];
DeviceCtx g_devctx[MAX_DEVICES];
std::mutex g_aux_mutex;

struct DeviceGuard {  // make `dev` current for the scope (hipSetDevice is per host thread)
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (dev >= 0 && hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) hipSetDevice(prev);
    }
};
int current_device() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess ? dev : -1;
}
int device_of(const void* p) {  // the device a device pointer belongs to (-1: unknown, leave the current device alone)
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return a.device;
}

// bring device `dev`'s copies of the shared tables / waves up to date (g_aux_mutex held, `dev` current).
// Order: render launches do not take g_aux_mutex and other host threads may be launching on this device, so
// a buffer the device-side Aux block still points to is never freed first.  New buffers are allocated and filled, the
// Aux block is rewritten (stream-ordered: launches enqueued after it see the new pointers), the device is drained --
// every launch that could have read the old pointers has finished -- and only then are the replaced buffers freed.  A
// failure at any step leaves *dev_aux pointing at live memory; host_aux / the version numbers only advance for what
// was uploaded, and `aux_dirty` makes the next call retry the block itself.
hipError_t sync_shared_locked(int dev) {
    DeviceCtx& c = g_devctx[dev];
    if (!c.init) {
        std::memset(&c.host_aux, 0, sizeof c.host_aux);
        hipError_t e = hipMalloc((void**)&c.dev_aux, sizeof(fd::Aux));
        if (e != hipSuccess) return e;
        c.init = true;
        c.aux_dirty = true;
    }
    hipError_t err = hipSuccess;
    for (int set = 0; set < fd::WT_SETS && err == hipSuccess; set++) {
        const HostTableSet& t = g_tables[set];
        if (t.ver == c.table_ver[set]) continue;
        // device layout: every table circularly padded [t[len-1], t[0..len-1], t[0], t[1]] (fd_nodes.hpp wt_tap)
        std::vector<float> padded;
        padded.reserve(t.data.size() + 3 * (size_t)t.n);
        fd::WtSet nw = c.host_aux.wt[set];
        size_t src = 0;
        for (int i = 0; i < t.n; i++) {
            const size_t len = (size_t)t.len[i];
            nw.pitch[i] = t.pitch[i];
            nw.off[i] = (int)padded.size();
            nw.len[i] = t.len[i];
            padded.push_back(t.data[src + len - 1]);
            padded.insert(padded.end(), t.data.begin() + src, t.data.begin() + src + len);
            padded.push_back(t.data[src]);
            padded.push_back(t.data[src + 1]);
            src += len;
        }
        float* d = nullptr;
        err = hipMalloc((void**)&d, padded.size() * sizeof(float));
        if (err == hipSuccess) err = hipMemcpy(d, padded.data(), padded.size() * sizeof(float), hipMemcpyHostToDevice);
        if (err != hipSuccess) {
            if (d) hipFree(d);
            break;
        }
        fd::WtSet& w = c.host_aux.wt[set];
        if (w.data) c.retired.push_back(const_cast<float*>(w.data));
        w = nw;
        w.n = t.n;
        w.data = d;
        c.table_ver[set] = t.ver;
        c.aux_dirty = true;
    }
    for (int slot = 0; slot < fd::WAVE_SLOTS && err == hipSuccess; slot++) {
        const HostWave& hw = g_waves[slot];
        if (hw.ver == c.wave_ver[slot]) continue;
        float* d = nullptr;
        err = hipMalloc((void**)&d, hw.data.size() * sizeof(float));
        if (err == hipSuccess) err = hipMemcpy(d, hw.data.data(), hw.data.size() * sizeof(float), hipMemcpyHostToDevice);
        if (err != hipSuccess) {
            if (d) hipFree(d);
            break;
        }
        fd::WaveBuf& w = c.host_aux.wave[slot];
        if (w.data) c.retired.push_back(const_cast<float*>(w.data));
        w.data = d;
        w.channels = (uint32_t)hw.channels;
        w.length = (uint32_t)hw.length;
        c.wave_ver[slot] = hw.ver;
        c.aux_dirty = true;
    }
    // whatever was uploaded goes live even if a later set failed: the block, then the drain, then the frees
    if (c.aux_dirty) {
        const hipError_t e = hipMemcpy(c.dev_aux, &c.host_aux, sizeof(fd::Aux), hipMemcpyHostToDevice);
        if (e != hipSuccess) return err != hipSuccess ? err : e;  // *dev_aux unchanged: the retired buffers stay alive
        c.aux_dirty = false;
    }
    if (!c.retired.empty()) {
        const hipError_t e = hipDeviceSynchronize();  // no launch that read the old block can still be running
        if (e != hipSuccess) return err != hipSuccess ? err : e;
        for (void* p : c.retired) hipFree(p);
        c.retired.clear();
    }
    return err;
}

// the Aux block of device `dev` (created and synchronised on first use); nullptr on failure
const void* device_aux(int dev) {
    if (dev < 0 || dev >= MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lock(g_aux_mutex);
    DeviceGuard g(dev);
    if (sync_shared_locked(dev) != hipSuccess) return nullptr;
    return g_devctx[dev].dev_aux;
}

// after a set / slot changed on the host: refresh every device that already has an Aux block, and the current one
int sync_all_devices() {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(FDSP_EDEVICE, "no HIP device available for shared wavetable / wave data");
    const int cur = current_device();
    std::lock_guard<std::mutex> lock(g_aux_mutex);
    for (int dev = 0; dev < ndev && dev < MAX_DEVICES; dev++) {
        if (!g_devctx[dev].init && dev != cur) continue;
        DeviceGuard g(dev);
        hipError_t e = sync_shared_locked(dev);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? FDSP_ENOMEM : FDSP_EDEVICE, std::string("shared data upload: ") + hipGetErrorString(e));
    }
    return FDSP_OK;
}

int upload_table_set(int set, int n, const float* pitches, const int* lengths, const float* data) {
    if (set < 0 || set >= fd::WT_SETS || n < 3 || n > fd::WT_MAX_TABLES) return fail(FDSP_EINVAL, "bad wavetable set or table count");
    size_t total = 0;
    for (int i = 0; i < n; i++) {
        if (lengths[i] < 4 || (lengths[i] & (lengths[i] - 1))) return fail(FDSP_EINVAL, "table lengths must be powers of two >= 4");
        total += (size_t)lengths[i];
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(FDSP_EDEVICE, "no HIP device available for wavetables");
    {
        std::lock_guard<std::mutex> lock(g_aux_mutex);
        HostTableSet& t = g_tables[set];
        t.n = n;
        t.pitch.assign(pitches, pitches + n);
        t.len.assign(lengths, lengths + n);
        t.data.assign(data, data + total);
        t.ver++;
    }
    return sync_all_devices();
}

int upload_wave(int slot, int channels, size_t length, const float* data) {
    if (slot < 0 || slot >= fd::WAVE_SLOTS) return fail(FDSP_EINVAL, "wave slot out of range");
    if (channels < 1 || length == 0 || length > 0xFFFFFFF0ull || !data) return fail(FDSP_EINVAL, "bad wave shape or NULL data");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(FDSP_EDEVICE, "no HIP device available for waves");
    {
        std::lock_guard<std::mutex> lock(g_aux_mutex);
        HostWave& w = g_waves[slot];
        w.channels = channels;
        w.length = length;
        w.data.assign(data, data + (size_t)channels * length);
        w.ver++;
    }
    return sync_all_devices();
}

// microfft 0.6.0's inverse FFT as the reference's make_wave calls it (fft.rs:51-100, wavetable.rs:75), restated
// (the crate source is not in the reference tree): in-place f32 radix-2 decimation-in-time complex FFT -- bit-reversal
// reorder, butterfly stages of span 2, 4, .. N, twiddles exp(-2 pi i k / span) as correctly rounded f32 cosine / sine --
// with the inverse obtained by reversing elements 1..N, transforming forward and dividing by N.
struct Cf { float re, im; };
void cfft_inplace(std::vector<Cf>& x) {
    const size_t n = x.size();
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(x[i], x[j]);
    }
    for (size_t span = 2; span <= n; span <<= 1) {
        const size_t half = span / 2;
        for (size_t k = 0; k < half; k++) {
            const double ang = 6.283185307179586476925286766559 * (double)k / (double)span;
            const float wr = (float)std::cos(ang), wi = (float)-std::sin(ang);
            for (size_t i = k; i < n; i += span) {
                Cf& p = x[i];
                Cf& q = x[i + half];
                const float yr = wr * q.re - wi * q.im, yi = wr * q.im + wi * q.re;  // Complex32 * Complex32
                const float ur = p.re, ui = p.im;
                q.re = ur - yr;
                q.im = ui - yi;
                p.re = ur + yr;
                p.im = ui + yi;
            }
        }
    }
}
void ifft_inplace(std::vector<Cf>& x) {
    std::reverse(x.begin() + 1, x.end());
    cfft_inplace(x);
    const float fn = (float)x.size();
    for (Cf& c : x) {
        c.re = c.re / fn;
        c.im = c.im / fn;
    }
}

// Wavetable::new + make_wave (wavetable.rs:44-123) for the built-in shapes (saw_table :493, square_table :510,
// triangle_table :523, organ_table :546, soft_saw_table :574, hammond_table :598).  Every step in the reference's
// precision and order (f64 partial weights, f32 polar insert, f32 FFT, f32 peak normalisation): the tables come out
// bit-identical to the CPU oracle's restatement (tests/test_gpu_config4.py builds them here and compares).
int compute_default_table_set(int set, std::vector<float>& pitches, std::vector<int>& lengths, std::vector<float>& data) {
    auto phase = [set](uint32_t i) -> double {
        if (set == 0) return (i & 1) == 1 ? 0.0 : 0.5;
        if (set == 1 || set == 6) return 0.0;
        if (set == 2) return (i & 3) == 3 ? 0.5 : 0.0;
        return (i & 3) == 3 ? 0.5 : ((i & 1) == 1 ? 0.0 : 0.5);  // organ, soft saw
    };
    auto amplitude = [set](uint32_t i) -> double {  // the closures compute in u32 before the cast to f64
        if (set == 0) return 1.0 / (double)i;
        if (set == 1) return (i & 1) == 1 ? 1.0 / (double)i : 0.0;
        if (set == 2) return (i & 1) == 1 ? 1.0 / (double)(uint32_t)(i * i) : 0.0;
        if (set == 5) return 1.0 / (double)(uint32_t)(i * i);
        const uint32_t z = (uint32_t)__builtin_ctz(i), j = i >> z;
        if (set == 4) return 1.0 / (double)(uint32_t)(i + j * j * j);
        // hammond
        const double f = 1.0 / (double)(uint32_t)((z + 1) * (z + 1));
        if (i <= 3) return 1.0;
        return j == 1 || j == 3 ? f : (j == 9 ? 0.2 * f : 0.0);
    };
    if (!(set >= 0 && set <= 2) && !(set >= 4 && set <= 6))
        return fail(FDSP_EINVAL, "built-in table sets: 0 saw, 1 square, 2 triangle, 4 organ, 5 soft saw, 6 hammond");
    pitches.clear();
    lengths.clear();
    data.clear();
    const double p_factor = std::pow(2.0, 1.0 / 4.0);
    float max_amplitude = 0.0f;
    for (double pitch = 20.0; pitch <= 20000.0; pitch *= p_factor) {
        const size_t harmonics = (size_t)std::floor(22000.0 / pitch);
        size_t length = 1;
        while (length < 4 * harmonics) length <<= 1;
        length = length < 32 ? 32 : (length > 8192 ? 8192 : length);
        std::vector<Cf> a(length, Cf{0.0f, 0.0f});
        for (size_t i = 1; i <= harmonics; i++) {
            const double f = pitch * (double)i;
            double x = (f - 22000.0) / (20000.0 - 22000.0);            // delerp(MAX_F, FADE_F, f)
            x = std::fmin(std::fmax(x, 0.0), 1.0);                     // clamp01
            const double w = amplitude((uint32_t)i) * (((x * 6.0 - 15.0) * x + 10.0) * x * x * x);  // smooth5
            if (w > 0.0) {  // Complex32::from_polar(w as f32, (TAU * phase) as f32)
                const float r = (float)w, theta = (float)(6.283185307179586476925286766559 * phase((uint32_t)i));
                a[i] = Cf{r * fd::cosf_musl(theta), r * fd::sinf_musl(theta)};
            }
        }
        ifft_inplace(a);
        const float z = (float)length;
        for (size_t k = 0; k < length; k++) {
            const float v = a[k].im * z;
            max_amplitude = std::fmax(max_amplitude, std::fabs(v));
            data.push_back(v);
        }
        pitches.push_back((float)pitch);
        lengths.push_back((int)length);
    }
    if (max_amplitude > 0.0f) {
        const float z = 1.0f / max_amplitude;
        for (float& x : data) x *= z;
    }
    return FDSP_OK;
}
int build_default_table_set(int set) {
    std::vector<float> pitches, data;
    std::vector<int> lengths;
    const int rc = compute_default_table_set(set, pitches, lengths, data);
    if (rc != FDSP_OK) return rc;
    return upload_table_set(set, (int)pitches.size(), pitches.data(), lengths.data(), data.data());
}

}  // namespace

struct FdnBank {  // reverb_stereo / reverb4_stereo banks (fd_fdn.hip): rings + per-line state instead of the slot SoA
    int kind = 0;  // 0 = reverb_stereo(room, time, damping), 1 = reverb4_stereo(room, time), 2 = the generic network (fdsp_fdn_create: desc),
                   // 3 = reverb3_stereo(time, diffusion = `damping`, lowpole_hz(cutoff)): the allpass loop of fd_reverb3.hip (c3 / st3; `c` only carries nin / nout)
    double room = 0.0, time = 0.0, damping = 0.0;
    fd::Rv3Filter flt;      // kind 3: the loop filter
    fd::Rv3Const c3;
    fd::Rv3State st3{};
    fd::FdnDesc desc;
    fd::FdnConst c;
    fd::FdnState st;
    fd::FdnBus bus;           // fdsp_bank_set_bus: wet * node [& dry * multipass()] folded into the render kernels' epilogue
    float* stage = nullptr;   // planar staging of voice-minor launches: [V][inputs][frames] | [V][outputs][frames] (fd_fdn.hip "voice-minor I/O")
    size_t stage_n = 0;
};

struct fdsp_bank {
    FdnBank* fdn = nullptr;
    float* ring = nullptr;       // delay-ring memory [ring node][position][voice] for kinds with Delay / Tap nodes
    uint32_t ring_cap = 0;       // positions per ring node
    const fd::KindOps* ops = nullptr;
    size_t V = 0, stride = 0;
    int nslots = 0;
    float* slots = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool timed = false;
    bool ext_pending = false;  // the last render ran on a caller's stream: bank-stream work must wait for its e1
    // fdsp_bank_set_param_all queued device-side work on the bank's stream and did not wait for it.  A plain flag, also cleared from const
    // paths (sync_bank_stream): safe under the handle's threading rule -- ONE host thread at a time per bank (fundsp_hip.h "Threading"), the
    // rule `process(&mut self)` gives the reference's nodes; it is not an atomic and must not be read from a second thread.
    mutable bool async_param_pending = false;
    int math = FDSP_MATH_EXACT;  // FDSP_MATH_FAST: renders take the kind's tolerance-mode variant when it has one
    // per-bank launch options (fdsp_bank_set_option); -1 = follow the process-wide default at every launch
    int opt_pipe_split = -1, opt_time_split = -1, opt_fdn_kernel = -1, opt_timing = -1;
    size_t ring_frames = 0;      // as given at creation (fdsp_bank_clone)
    int last_kernel = 0;         // fd::LastKernel of the most recent render launch (fdsp_bank_get_option "last_kernel")
    bool ring_check_pending = false;  // a lifecycle launch may have changed a delay length: verify capacity before rendering
    uint32_t ring_short = 0;          // > 0: positions a node wanted and did not get (renders fail until a later update fits)
    int device = 0;              // the HIP device the bank lives on; every entry point makes it current for its own duration
    const void* aux = nullptr;   // that device's shared-data block (fd::Aux)
    double sr;
    std::unordered_map<std::string, int> index;
    // voice scheduler (fdsp_bank_set_events / fdsp_bank_process_events): device [4][stride] f64 + [stride] int, clock
    double* ev = nullptr;
    int* ev_fade = nullptr;
    double seq_time = 0.0;
    // host mirror of the events [V][4] and its aggregates: a launch that every voice sustains (inside its event, no fade
    // running) is a plain render and takes the pipeline kernel
    std::vector<double> ev_host;
    bool ev_dirty = true;
    double ev_max_start = 0.0, ev_min_end = 0.0, ev_max_fade_in_end = 0.0, ev_min_fade_out_start = 0.0;
    // fdsp_bank_process_host staging, grown on demand and kept: a real-time host calls once per 64-frame block
    float *st_in = nullptr, *st_out = nullptr;     // device
    float *pin_in = nullptr, *pin_out = nullptr;   // pinned host (small transfers only)
    size_t st_in_n = 0, st_out_n = 0, pin_in_n = 0, pin_out_n = 0;
    // fused mix-down (fdsp_bank_process_mix): the voice groups' partial mixes [groups][channels][frames], grown on demand and
    // kept (fdsp_bank_mix_reserve sizes it ahead of a real-time loop or a graph capture); pan weights [2][stride] (FDSP_MIX_PAN)
    float* mix_part = nullptr;
    size_t mix_part_n = 0;
    float* panw = nullptr;
};

namespace {

// A render on a caller stream is ordered after whatever the bank's own stream still has pending (parameter uploads).
// While the caller's stream is being CAPTURED into a HIP graph a host-side synchronize would invalidate the capture; the
// bank's stream is idle by then (every setter waits for the copies AND the coefficient-update launch it queued before it returns), so
// the wait is simply skipped.
// (The one setter that does NOT wait -- fdsp_bank_set_param_all, the device-side fill -- marks the bank; a capture that would start
// behind such work is refused instead of racing with it: the host calls fdsp_bank_synchronize first.)
// every host-side wait for the bank's stream goes through here: whatever fdsp_bank_set_param_all queued has landed afterwards
hipError_t sync_bank_stream(const fdsp_bank* b) {
    const hipError_t e = hipStreamSynchronize(b->stream);
    if (e == hipSuccess) b->async_param_pending = false;
    return e;
}
int order_after_bank_stream(const fdsp_bank* b, hipStream_t s) {
    if (s == b->stream) return FDSP_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
        if (b->async_param_pending)
            return fail(FDSP_EDEVICE, "a device-side fdsp_bank_set_param_all is still queued on the bank's stream: call fdsp_bank_synchronize before capturing "
                                      "(the capture is refused rather than left to race with the fill)");
        return FDSP_OK;
    }
    HIPCHK(sync_bank_stream(b));
    return FDSP_OK;
}

// ... and the other way round: lifecycle / parameter work on the bank's stream waits for the last render that ran on a
// caller's stream (its completion event e1), so that a reset or a parameter upload cannot overtake it.
hipError_t await_last_render(fdsp_bank* b) {
    if (!b->ext_pending) return hipSuccess;
    b->ext_pending = false;
    return hipStreamWaitEvent(b->stream, b->e1, 0);
}

// Kinds with delay lines keep their rings at the capacity fixed at bank creation.  A node whose length no longer fits
// (the reference would resize its buffer) reports the positions it wanted in one word behind the ring memory
// (Ctx::want_positions, written by the lifecycle kernels).  Parameters arrive one slot at a time, so a complaint is only
// final once the host starts rendering: every lifecycle launch that re-derives ALL voices clears the word first, marks
// the bank, and the next render call reads it back -- and fails, loudly, as long as the rings are too small.
uint32_t* ring_need_word(const fdsp_bank* b) {
    return reinterpret_cast<uint32_t*>(b->ring + (size_t)b->ops->nrings * b->ring_cap * b->stride);
}
void before_update_launch(fdsp_bank* b, size_t first, size_t count) {
    (void)first; (void)count;
    if (!b->ring || !b->ops || b->ops->nrings == 0) return;
    b->ring_check_pending = true;
}
// Parameters arrive one slot and one voice RANGE at a time, so the word a partial update leaves behind says nothing
// final: a too-long delay corrected range by range would keep its stale complaint, and a complaint of voices outside
// the last range could be lost.  So the check itself asks every voice again: clear the word, re-derive all
// voices (lifecycle op 1 = update(): recomputes the same coefficients from the same parameters, idempotent), read it.
int check_ring_need(fdsp_bank* b, bool capturing) {
    if (!b->ring || !b->ops || b->ops->nrings == 0) return FDSP_OK;
    if (b->ring_check_pending && !capturing) {
        uint32_t need = 0;
        HIPCHK(hipMemsetAsync(ring_need_word(b), 0, sizeof(uint32_t), b->stream));
        b->ops->lifecycle(b->slots, b->stride, 0, b->stride, 1, b->sr, nullptr, b->aux, b->ring, b->ring_cap, b->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&need, ring_need_word(b), sizeof need, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(sync_bank_stream(b));
        b->ring_check_pending = false;
        b->ring_short = need > b->ring_cap ? need : 0;
    }
    if (b->ring_short)
        return fail(FDSP_EINVAL, "ring capacity too small: a delay / tap / limiter node needs " + std::to_string(b->ring_short) +
                                     " positions with the current parameters and sample rate, the bank was created with ring_frames = " +
                                     std::to_string(b->ring_cap));
    return FDSP_OK;
}

int find_slot(const fdsp_bank* b, const char* name) {
    if (!name) return -1;
    auto it = b->index.find(name);
    return it == b->index.end() ? -1 : it->second;
}

int check_range(const fdsp_bank* b, size_t first, size_t count) {
    if (first > b->V || count > b->V - first) return fail(FDSP_EINVAL, "voice range out of bounds");
    return FDSP_OK;
}

// pan weights of Panner (pan.rs:13-17) for the mix-down
__global__ void k_pan_weights(const float* pan, float* wl, float* wr, size_t V, size_t live = ~(size_t)0) {
    size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    if (v >= live) {  // padding behind a bank's last voice: weight 0, 0 (a padded voice contributes +0.0 to the mix)
        wl[v] = 0.0f;
        wr[v] = 0.0f;
        return;
    }
    float value = pan ? pan[v] : 0.0f;
    float c = value < -1.0f ? -1.0f : (value > 1.0f ? 1.0f : value);  // clamp11
    float angle = (c + 1.0f) * (fd::F32_PI * 0.25f);
    wl[v] = fd::cosf_musl(angle);
    wr[v] = fd::sinf_musl(angle);
}

// ---- the mix-down's summation order (fd_device.hpp "fused mix-down"), for voice-out buffers --------------------------
// partial(group of 64 voices) = (S0 + S1) + (S2 + S3), Sq = the quarter's 16 voices added one after the other (voices past the
// end count as +0.0); then the groups' partials in an aligned binary tree.  The render kernels with the fused mix-down produce
// the same partials without the voice-out buffer ever existing.
FD_D float quad_swap(float x, int ctrl) {
    return fd::u2f((uint32_t)(ctrl == 0 ? __builtin_amdgcn_update_dpp((int)fd::f2u(x), (int)fd::f2u(x), 0xB1, 0xF, 0xF, false)
                                        : __builtin_amdgcn_update_dpp((int)fd::f2u(x), (int)fd::f2u(x), 0x4E, 0xF, 0xF, false)));
}
// x: [rows][V] voice-minor.  PAN = false: part[group][row] = partial sum of the row.  PAN = true (rows = frames of a mono render):
// part[group][c][row], c = left / right, every sample weighted by its voice's pan weights first.
// One workgroup = one voice group x 256 rows: thread (row, quarter) reads its quarter's 16 consecutive floats.
template <bool PAN, bool VEC4>
__global__ __launch_bounds__(256) void k_group_partials(const float* __restrict__ x, const float* __restrict__ wl,
                                                        const float* __restrict__ wr, float* __restrict__ part, size_t rows, size_t V) {
    const size_t g = blockIdx.x, r0 = (size_t)blockIdx.y * 256;
    const int q = threadIdx.x & 3, rl = threadIdx.x >> 2;
    const size_t vq = g * 64 + (size_t)q * 16;
    float a[PAN ? 16 : 1], b[PAN ? 16 : 1];
    if constexpr (PAN) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            a[j] = vq + j < V ? wl[vq + j] : 0.0f;
            b[j] = vq + j < V ? wr[vq + j] : 0.0f;
        }
    }
#pragma unroll 1
    for (int p = 0; p < 4; p++) {
        const size_t r = r0 + (size_t)p * 64 + rl;
        const bool on = r < rows;
        const float* src = x + (on ? r : 0) * V + vq;
        float xs[16];
        if (VEC4 && on && vq + 16 <= V) {  // aligned rows: four 16-byte loads per thread instead of sixteen dwords
            const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float4 q4 = s4[j];
                xs[4 * j] = q4.x; xs[4 * j + 1] = q4.y; xs[4 * j + 2] = q4.z; xs[4 * j + 3] = q4.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) xs[j] = (on && vq + j < V) ? src[j] : 0.0f;
        }
        float sl, sr = 0.0f;
        if constexpr (PAN) {
            sl = vq < V ? xs[0] * a[0] : 0.0f;
            sr = vq < V ? xs[0] * b[0] : 0.0f;
#pragma unroll
            for (int j = 1; j < 16; j++) {
                sl += vq + j < V ? xs[j] * a[j] : 0.0f;
                sr += vq + j < V ? xs[j] * b[j] : 0.0f;
            }
        } else {
            sl = xs[0];
#pragma unroll
            for (int j = 1; j < 16; j++) sl += xs[j];
        }
        const float tl = sl + quad_swap(sl, 0), ul = tl + quad_swap(tl, 1);
        if constexpr (PAN) {
            const float tr = sr + quad_swap(sr, 0), ur = tr + quad_swap(tr, 1);
            if (on && q == 0) {
                part[(g * 2 + 0) * rows + r] = ul;
                part[(g * 2 + 1) * rows + r] = ur;
            }
        } else if (on && q == 0) part[g * rows + r] = ul;
    }
}

// part: [G][R] -> mix[R]: the aligned binary tree over the G groups (a node without a right sibling passes through).
// A workgroup takes RB rows; its 256 / RB thread slices each reduce an ALIGNED run of C groups (C a power of two >= 16: a subtree of
// the same tree) -- sixteen groups loaded at a time (independent loads), reduced in registers, the blocks of sixteen merged by a binary
// counter (`stack[k]` holds the sum of an aligned run of 16 * 2^k groups) -- and the slices' sums meet in LDS, again in tree order.
// RB = 64 for long launches (coalesced 256-byte rows); RB = 16 for short ones (a 64-frame block has 128 rows: sixteen slices per row
// keep the chain of dependent loads short -- one thread per row took 60 us for 1 024 groups).
template <int RB>
__global__ __launch_bounds__(256) void k_mix_tree(const float* __restrict__ part, float* __restrict__ mix, size_t R, size_t G, size_t C) {
    constexpr int GS = 256 / RB;
    __shared__ float red[GS][RB];
    const int rl = threadIdx.x % RB, sl = threadIdx.x / RB;
    const size_t r = (size_t)blockIdx.x * RB + rl;
    const size_t g0 = (size_t)sl * C, g1 = g0 + C < G ? g0 + C : G;
    float acc = 0.0f;
    if (r < R && g0 < G) {
        float stack[28];
        const size_t nb = (g1 - g0 + 15) / 16;
        for (size_t b = 0; b < nb; b++) {
            const size_t gb = g0 + b * 16;
            const int m = (int)(g1 - gb < 16 ? g1 - gb : 16);  // groups of this block (uniform over the slice)
            float x[16];
#pragma unroll
            for (int j = 0; j < 16; j++) x[j] = j < m ? part[(gb + j) * R + r] : 0.0f;
#pragma unroll
            for (int span = 1; span < 16; span <<= 1)
#pragma unroll
                for (int j = 0; j + span < 16; j += 2 * span)
                    if (j + span < m) x[j] = x[j] + x[j + span];
            float carry = x[0];
            bool placed = false;
#pragma unroll
            for (int k = 0; k < 28; k++) {
                if (!placed) {
                    if (((b >> k) & 1) == 0) { stack[k] = carry; placed = true; }
                    else carry = stack[k] + carry;
                }
            }
        }
        bool have = false;
#pragma unroll
        for (int k = 0; k < 28; k++)
            if ((nb >> k) & 1) {
                acc = have ? stack[k] + acc : stack[k];
                have = true;
            }
    }
    red[sl][rl] = acc;
    __syncthreads();
    if (sl == 0 && r < R) {
        const int nvalid = (int)((G + C - 1) / C);  // slices that hold groups (<= GS)
        float x[GS];
#pragma unroll
        for (int j = 0; j < GS; j++) x[j] = j < nvalid ? red[j][rl] : 0.0f;
#pragma unroll
        for (int span = 1; span < GS; span <<= 1)
#pragma unroll
            for (int j = 0; j + span < GS; j += 2 * span)
                if (j + span < nvalid) x[j] = x[j] + x[j + span];
        mix[r] = x[0];
    }
}
// rows-per-workgroup policy and the slice width (a power of two >= 16 that covers the groups with the slices at hand)
void launch_mix_tree(const float* part, float* mix, size_t R, size_t G, hipStream_t s) {
    auto slice = [&](size_t slices) {
        size_t c = 16;
        while (c * slices < G) c <<= 1;
        return c;
    };
    if (R >= 2048) hipLaunchKernelGGL((k_mix_tree<64>), dim3((unsigned)((R + 63) / 64)), dim3(256), 0, s, part, mix, R, G, slice(4));
    else hipLaunchKernelGGL((k_mix_tree<16>), dim3((unsigned)((R + 15) / 16)), dim3(256), 0, s, part, mix, R, G, slice(16));
}

// the launch pair behind fdsp_sum_voices / fdsp_mix_stereo: group partials of a voice-out buffer, then the tree
template <bool PAN>
hipError_t launch_mix_rows(const float* x, const float* wl, const float* wr, float* part, float* mix, size_t rows, size_t V, hipStream_t s) {
    const size_t G = (V + 63) / 64, R = PAN ? 2 * rows : rows;
    if (V % 4 == 0 && ((uintptr_t)x & 15) == 0)
        hipLaunchKernelGGL((k_group_partials<PAN, true>), dim3((unsigned)G, (unsigned)((rows + 255) / 256)), dim3(256), 0, s, x, wl, wr, part, rows, V);
    else
        hipLaunchKernelGGL((k_group_partials<PAN, false>), dim3((unsigned)G, (unsigned)((rows + 255) / 256)), dim3(256), 0, s, x, wl, wr, part, rows, V);
    launch_mix_tree(part, mix, R, G, s);
    return hipGetLastError();
}

}  // namespace

extern "C" {

const char* fdsp_last_error(void) { return g_err.c_str(); }

int fdsp_kind_count(void) {
    auto& r = registry();
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    return (int)r.size();
}
const char* fdsp_kind_name(int kind) {
    const fd::KindOps* k = kind_at(kind);
    return k ? k->name.c_str() : nullptr;
}
int fdsp_kind_by_name(const char* name) {
    auto& r = registry();
    if (!name) return -1;
    std::lock_guard<std::mutex> lock(g_registry_mutex);  // fdsp_graph_compile_src appends concurrently
    for (size_t i = 0; i < r.size(); i++)
        if (r[i].name == name) return (int)i;
    return -1;
}

namespace {
// the launch options by name: range check shared by the process-wide and the per-bank setter
// ... each entry names its process-wide default and its per-bank override itself: the setters and the getter index by the table
struct OptSpec { const char* name; int lo, hi; const char* what; std::atomic<int>* global; int fdsp_bank::* field; };
const OptSpec LAUNCH_OPTS[] = {
    {"pipe_split", 0, 4, "pipe_split takes 0 (off), 1 (auto), 2 or 3 (stages), 4 (loader only)", &fd::g_pipe_split, &fdsp_bank::opt_pipe_split},
    {"time_split", 0, 2, "time_split takes 0 (off), 1 (small banks of eligible graphs: 3 + 3 + 1 waves per group) or 2 (round 2's 2 + 2 + 1 / 2 + 1 + 1 layouts)", &fd::g_time_split, &fdsp_bank::opt_time_split},
    {"fdn_kernel", 0, 1, "fdn_kernel takes 0 (lane per frame) or 1 (lane per delay line)", &fd::g_fdn_kernel, &fdsp_bank::opt_fdn_kernel},
    {"timing", 0, 1, "timing takes 0 (no per-launch event pair) or 1 (fdsp_bank_last_kernel_ms available)", &fd::g_timing, &fdsp_bank::opt_timing},
};
const OptSpec* launch_opt(const char* name) {
    if (!name) return nullptr;
    for (const OptSpec& o : LAUNCH_OPTS)
        if (std::strcmp(name, o.name) == 0) return &o;
    return nullptr;
}
// resolve a bank's launch options into the calling thread's block (fd_opts.hpp) -- every render entry point, right
// before it launches
void resolve_opts(const fdsp_bank* b) {
    fd::tl_opts.pipe_split = b->opt_pipe_split >= 0 ? b->opt_pipe_split : fd::g_pipe_split.load(std::memory_order_relaxed);
    fd::tl_opts.time_split = b->opt_time_split >= 0 ? b->opt_time_split : fd::g_time_split.load(std::memory_order_relaxed);
    fd::tl_opts.fdn_kernel = b->opt_fdn_kernel >= 0 ? b->opt_fdn_kernel : fd::g_fdn_kernel.load(std::memory_order_relaxed);
    fd::tl_opts.last_kernel = fd::LK_NONE;
}
bool timing_on(const fdsp_bank* b) { return (b->opt_timing >= 0 ? b->opt_timing : fd::g_timing.load(std::memory_order_relaxed)) != 0; }
}  // namespace

int fdsp_set_option(const char* name, int value) {
    if (const OptSpec* o = launch_opt(name)) {
        if (value < o->lo || value > o->hi) return fail(FDSP_EINVAL, o->what);
        o->global->store(value);
        return FDSP_OK;
    }
    if (name && std::strcmp(name, "host_zero_copy_max") == 0) {
        if (value < 0) return fail(FDSP_EINVAL, "host_zero_copy_max takes a float count >= 0");
        fd::g_zero_copy_max.store(value);
        return FDSP_OK;
    }
    if (name && std::strcmp(name, "math") == 0) {
        if (value != FDSP_MATH_EXACT && value != FDSP_MATH_FAST) return fail(FDSP_EINVAL, "math takes FDSP_MATH_EXACT (0) or FDSP_MATH_FAST (1)");
        fd::g_math.store(value);
        return FDSP_OK;
    }
    return fail(FDSP_EINVAL, "unknown option");
}

int fdsp_bank_set_option(fdsp_bank* b, const char* name, int value) {
    if (!b) return fail(FDSP_EINVAL, "bank is NULL");
    if (name && std::strcmp(name, "math") == 0) {
        if (value != FDSP_MATH_EXACT && value != FDSP_MATH_FAST) return fail(FDSP_EINVAL, "math takes FDSP_MATH_EXACT (0) or FDSP_MATH_FAST (1)");
        b->math = value;
        // run-time compiled kinds: the tolerance-mode modules are built HERE, not inside the bank's next render (which may be captured)
        if (value == FDSP_MATH_FAST && b->ops) {
            DeviceGuard guard(b->device);
            if (b->ops->prepare_render) b->ops->prepare_render(b->V, true);
            if (b->ops->prepare_mix && b->ops->render_mix_fast && (b->mix_part || b->panw)) b->ops->prepare_mix(true);
        }
        return FDSP_OK;
    }
    if (const OptSpec* o = launch_opt(name)) {  // -1 = back to the process-wide default
        if (value != -1 && (value < o->lo || value > o->hi)) return fail(FDSP_EINVAL, std::string(o->what) + "; -1 = follow the process-wide default");
        b->*(o->field) = value;
        return FDSP_OK;
    }
    return fail(FDSP_EINVAL, "unknown bank option");
}
int fdsp_bank_get_option(const fdsp_bank* b, const char* name) {
    if (!b) return fail(FDSP_EINVAL, "bank is NULL");
    if (name && std::strcmp(name, "math") == 0) return b->math;
    if (name && std::strcmp(name, "math_has_fast_variant") == 0) return (b->ops && b->ops->render_fast) ? 1 : 0;
    if (const OptSpec* o = launch_opt(name)) return b->*(o->field) >= 0 ? b->*(o->field) : o->global->load();
    if (name && std::strcmp(name, "has_fused_mix") == 0) return (b->ops && b->ops->render_mix) ? 1 : 0;
    if (name && std::strcmp(name, "last_kernel") == 0) return b->last_kernel;
    return fail(FDSP_EINVAL, "unknown bank option");
}

const char* fdsp_jit_compiler(void) { return fd::jit_compiler_origin(); }

int fdsp_graph_compile(const char* name, const char* type_expr) { return fdsp_graph_compile_src(name, type_expr, nullptr); }

int fdsp_graph_compile_src(const char* name, const char* type_expr, const char* source) {
    if (!name || !type_expr || !*name || !*type_expr) return fail(FDSP_EINVAL, "name or type expression missing");
    int existing = fdsp_kind_by_name(name);
    if (existing >= 0) return existing;  // kinds are immutable: a second compile of the same name is a lookup
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(FDSP_EDEVICE, "no HIP device available: compiled graphs are loaded onto the device");
    fd::KindOps k;
    std::string err;
    if (fd::jit_make_kind(name, type_expr, source ? source : "", &k, &err) != 0) return fail(FDSP_EINVAL, err);
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    for (size_t i = 0; i < registry().size(); i++)  // another thread may have compiled the same name meanwhile
        if (registry()[i].name == name) return (int)i;
    registry().push_back(std::move(k));
    return (int)registry().size() - 1;
}

int fdsp_graph_compile_rust(const char* name, const char* rust_type_name, const char* hints, const char* source) {
    if (!name || !*name) return fail(FDSP_EINVAL, "name missing");
    std::string expr, presets;
    if (int rc = fd::rust_translate(rust_type_name, hints, &expr, &presets)) return rc;
    const int before = fdsp_kind_by_name(name);
    const int k = fdsp_graph_compile_src(name, expr.c_str(), source);
    if (k < 0 || before >= 0) return k;  // error, or the name was compiled before (its presets are already in place)
    std::vector<std::pair<std::string, float>> ps;
    size_t pos = 0;
    while (pos < presets.size()) {
        const size_t nl = presets.find('\n', pos), eq = presets.find('=', pos);
        if (nl == std::string::npos || eq == std::string::npos || eq > nl) break;
        ps.push_back({presets.substr(pos, eq - pos), std::strtof(presets.c_str() + eq + 1, nullptr)});
        pos = nl + 1;
    }
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    registry()[k].presets = std::move(ps);
    return k;
}

int fdsp_graph_check(const char* type_expr) {
    if (!type_expr) return fail(FDSP_EINVAL, "type expression missing");
    std::vector<char> code;
    std::string log;
    if (fd::jit_compile_code(type_expr, "", &code, &log) != 0) return fail(FDSP_EINVAL, log);
    // ... and the kind's second module (fused mix-down and time-split kernels, compiled lazily at run time): a graph whose main module builds
    // must not fail there either
    code.clear();
    if (fd::jit_compile_src(fd::jit_source_mix(type_expr, ""), type_expr, &code, &log) != 0) return fail(FDSP_EINVAL, "second module (mix-down / time-split kernels): " + log);
    return FDSP_OK;
}

int fdsp_kind_inputs(int kind) {
    const fd::KindOps* k = kind_at(kind);
    return k ? k->nin : FDSP_EINVAL;
}
int fdsp_kind_outputs(int kind) {
    const fd::KindOps* k = kind_at(kind);
    return k ? k->nout : FDSP_EINVAL;
}
int fdsp_kind_slot_count(int kind) {
    const fd::KindOps* k = kind_at(kind);
    return k ? (int)k->slots.size() : FDSP_EINVAL;
}
const char* fdsp_kind_slot_name(int kind, int slot) {
    const fd::KindOps* k = kind_at(kind);
    if (!k || slot < 0 || slot >= (int)k->slots.size()) return nullptr;
    return k->slots[slot].name.c_str();
}
int fdsp_kind_slot_kind(int kind, int slot) {
    const fd::KindOps* k = kind_at(kind);
    if (!k || slot < 0 || slot >= (int)k->slots.size()) return FDSP_EINVAL;
    return k->slots[slot].kind;
}

int fdsp_device_count(void) {
    int ndev = 0;
    return hipGetDeviceCount(&ndev) == hipSuccess ? ndev : 0;
}

int fdsp_bank_device(const fdsp_bank* b) { return b ? b->device : FDSP_EINVAL; }

int fdsp_bank_create(const char* kind, size_t voices, fdsp_bank** out) {
    return fdsp_bank_create_on(-1, kind, voices, 0, out);
}

int fdsp_bank_create_ring(const char* kind, size_t voices, size_t ring_frames, fdsp_bank** out) {
    return fdsp_bank_create_on(-1, kind, voices, ring_frames, out);
}

int fdsp_bank_create_on(int device, const char* kind, size_t voices, size_t ring_frames, fdsp_bank** out) {
    if (!out) return fail(FDSP_EINVAL, "out is NULL");
    *out = nullptr;
    int k = fdsp_kind_by_name(kind);
    if (k < 0) return fail(FDSP_EINVAL, std::string("unknown voice-graph kind: ") + (kind ? kind : "(null)"));
    if (voices == 0) return fail(FDSP_EINVAL, "voices must be > 0");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(FDSP_EDEVICE, "no HIP device available: the fundsp_hip engine has no CPU fallback");
    if (device < 0) device = current_device();
    if (device < 0 || device >= ndev || device >= MAX_DEVICES) return fail(FDSP_EINVAL, "device index out of range");
    DeviceGuard guard(device);
    const void* aux = device_aux(device);
    if (!aux) return fail(FDSP_EDEVICE, "cannot set up the shared-data block on the device");
    fdsp_bank* b = new fdsp_bank();
    b->device = device;
    b->aux = aux;
    b->ops = kind_at(k);
    b->V = voices;
    b->stride = (voices + 63) / 64 * 64;
    b->nslots = (int)b->ops->slots.size();
    b->slots = nullptr;
    b->stream = nullptr;
    b->timed = false;
    b->sr = FDSP_DEFAULT_SR;
    b->math = fd::g_math.load();
    for (int i = 0; i < b->nslots; i++) b->index[b->ops->slots[i].name] = i;
    size_t bytes = (size_t)(b->nslots > 0 ? b->nslots : 1) * b->stride * sizeof(float);
    hipError_t e = hipMalloc((void**)&b->slots, bytes);
    if (e != hipSuccess) {
        delete b;
        return fail(FDSP_ENOMEM, std::string("hipMalloc(slots): ") + hipGetErrorString(e));
    }
    if (hipStreamCreate(&b->stream) != hipSuccess ||
        hipEventCreate(&b->e0) != hipSuccess || hipEventCreate(&b->e1) != hipSuccess) {
        fdsp_bank_destroy(b);  // frees whatever of {slots, stream, e0, e1} exists
        return fail(FDSP_EDEVICE, "stream/event creation failed");
    }
    if (b->ops->prepare_render) b->ops->prepare_render(voices, b->math == FDSP_MATH_FAST);  // (run-time compiled kinds: whatever a bank of this size and arithmetic still has to compile)
    if (b->ops->nrings > 0) {
        if (ring_frames == 0 || ring_frames > 0x7fffffffu) {
            fdsp_bank_destroy(b);
            return fail(FDSP_EINVAL, "this kind contains delay lines: create it with fdsp_bank_create_ring(kind, voices, ring_frames)");
        }
        b->ring_cap = (uint32_t)ring_frames;
        b->ring_frames = ring_frames;
        // + one 64-byte line behind the rings: the word in which a node reports that it wanted more positions
        const size_t rbytes = (size_t)b->ops->nrings * ring_frames * b->stride * sizeof(float) + 64;
        e = hipMalloc((void**)&b->ring, rbytes);
        if (e != hipSuccess) {
            fdsp_bank_destroy(b);
            return fail(FDSP_ENOMEM, std::string("hipMalloc(ring): ") + hipGetErrorString(e));
        }
        hipMemsetAsync(b->ring, 0, rbytes, b->stream);
    }
    hipMemsetAsync(b->slots, 0, bytes, b->stream);
    b->ops->lifecycle(b->slots, b->stride, 0, b->stride, 0, b->sr, nullptr, b->aux, b->ring, b->ring_cap, b->stream);
    e = sync_bank_stream(b);
    if (e != hipSuccess) {
        fdsp_bank_destroy(b);
        return fail(FDSP_EDEVICE, std::string("bank construction kernel failed: ") + hipGetErrorString(e));
    }
    if (!b->ops->presets.empty()) {  // parameters carried by the Rust type of the graph (fdsp_graph_compile_rust)
        std::vector<float> col(b->stride);
        for (const auto& kv : b->ops->presets) {
            auto it = b->index.find(kv.first);
            if (it == b->index.end()) {
                fdsp_bank_destroy(b);
                return fail(FDSP_EINVAL, "preset names an unknown slot: " + kv.first);
            }
            std::fill(col.begin(), col.end(), kv.second);
            hipMemcpyAsync(b->slots + (size_t)it->second * b->stride, col.data(), b->stride * sizeof(float), hipMemcpyHostToDevice, b->stream);
            sync_bank_stream(b);
        }
        b->ops->lifecycle(b->slots, b->stride, 0, b->stride, 1, b->sr, nullptr, b->aux, b->ring, b->ring_cap, b->stream);
        if (sync_bank_stream(b) != hipSuccess) {
            fdsp_bank_destroy(b);
            return fail(FDSP_EDEVICE, "applying the kind's presets failed");
        }
    }
    // Construction runs every node's update() with its DEFAULT parameters (e.g. the limiter's 5 ms attack), which the
    // caller is about to replace: a capacity complaint raised by defaults is discarded.
    if (b->ring) {
        const uint32_t line[2] = {0u, (uint32_t)(b->V > 0xFFFFFFFFull ? 0xFFFFFFFFull : b->V)};  // [need, real voices]
        hipMemcpyAsync(ring_need_word(b), line, sizeof line, hipMemcpyHostToDevice, b->stream);
        sync_bank_stream(b);
    }
    *out = b;
    return FDSP_OK;
}

static void fdn_free(FdnBank* f) {
    if (!f) return;
    if (f->st.rings) hipFree(f->st.rings);
    if (f->st.wpos) hipFree(f->st.wpos);
    if (f->st.v1) hipFree(f->st.v1);
    if (f->st.v2) hipFree(f->st.v2);
    if (f->st.fb) hipFree(f->st.fb);
    f->st = fd::FdnState{};
    if (f->st3.rings) hipFree(f->st3.rings);
    if (f->st3.wpos) hipFree(f->st3.wpos);
    f->st3.rings = nullptr;   // (pre, wpre, fval outlive a re-configuration: rv3_free_persistent)
    f->st3.wpos = nullptr;
}
static void rv3_free_persistent(FdnBank* f) {
    if (f->st3.pre) hipFree(f->st3.pre);
    if (f->st3.wpre) hipFree(f->st3.wpre);
    if (f->st3.fval) hipFree(f->st3.fval);
    f->st3.pre = nullptr; f->st3.wpre = nullptr; f->st3.fval = nullptr;
}
// reverb3_stereo banks: (re)allocate the loop's lines for a sample rate.  Transactional like fdn_configure.  What the reference's
// Reverb::set_sample_rate leaves alone survives the move (the `pre` diffusers entirely; every allpass's z, the feedback sample and the
// filters' values: fd_reverb3.hpp rv3_launch_migrate); the first configuration zeroes everything.
static int rv3_configure(fdsp_bank* b, double sr) {
    FdnBank* f = b->fdn;
    fd::Rv3Const c;
    if (!fd::rv3_make_const(f->time, f->damping, f->flt, sr, &c))
        return fail(FDSP_EINVAL, "reverb3_stereo: every delay must exceed 128 samples at the bank's sample rate (two blocks: the lane-per-frame kernel's rule; >= 14.2 kHz)");
    const size_t n = b->V;
    const bool first = f->st3.pre == nullptr;
    fd::Rv3State st = f->st3;
    st.rings = nullptr;
    st.wpos = nullptr;
    hipError_t e = hipMalloc((void**)&st.rings, n * c.ring_stride * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&st.wpos, n * sizeof(int));
    if (e == hipSuccess && first) e = hipMalloc((void**)&st.pre, n * 4 * (fd::RV3_PRE_CAP + 64) * sizeof(float));
    if (e == hipSuccess && first) e = hipMalloc((void**)&st.wpre, n * sizeof(int));
    if (e == hipSuccess && first) e = hipMalloc((void**)&st.fval, n * 32 * sizeof(float));
    if (e != hipSuccess) {
        if (st.rings) hipFree(st.rings);
        if (st.wpos) hipFree(st.wpos);
        if (first) { if (st.pre) hipFree(st.pre); if (st.wpre) hipFree(st.wpre); if (st.fval) hipFree(st.fval); }
        return fail(e == hipErrorOutOfMemory ? FDSP_ENOMEM : FDSP_EDEVICE, std::string("reverb3_stereo buffers: ") + hipGetErrorString(e));
    }
    if (first) fd::rv3_launch_init(c, st, n, b->stream);
    else {
        // the new lines start empty; pre / filter values stay where they are; z and the feedback sample move over
        fd::Rv3State fresh = st;
        hipMemsetAsync(st.rings, 0, n * c.ring_stride * sizeof(float), b->stream);
        hipMemsetAsync(st.wpos, 0, n * sizeof(int), b->stream);
        fd::rv3_launch_migrate(f->c3, f->st3, c, fresh, n, b->stream);
        hipStreamSynchronize(b->stream);
        hipFree(f->st3.rings);
        hipFree(f->st3.wpos);
    }
    f->c3 = c;
    f->st3 = st;
    f->c.nin = f->c.nout = 2;
    b->sr = sr;
    HIPCHK(hipGetLastError());
    return FDSP_OK;
}
static void fdn_free_stage(FdnBank* f) {
    if (f && f->stage) hipFree(f->stage);
    if (f) { f->stage = nullptr; f->stage_n = 0; }
}

// (re)allocate rings for the current sample rate and zero everything: Delay::set_sample_rate resizes + resets
// when the rate changes (delay.rs:105-113)
// Transactional: the new constants are validated and the new buffers allocated BEFORE anything of the bank changes; on
// any failure the bank keeps its old constants, rings and rate.
static int fdn_configure(fdsp_bank* b, double sr) {
    FdnBank* f = b->fdn;
    if (f->kind == 3) return rv3_configure(b, sr);
    fd::FdnConst c;
    if (f->kind == 2) fd::fdn_make_const_generic(f->desc, sr, &c);
    else if (f->kind == 1) fd::fdn_make_const_reverb4(f->room, f->time, sr, &c);
    else fd::fdn_make_const(f->room, f->time, f->damping, sr, &c);
    for (int i = 0; i < c.lines; i++)
        if (c.len[i] <= 128)
            return fail(FDSP_EINVAL, f->kind == 2 ? "fdsp_fdn_create: every delay must exceed 128 samples at the bank's sample rate (two blocks: the lane-per-frame kernel's rule)"
                                                  : "reverb_stereo / reverb4_stereo: every delay must exceed 128 samples (room_size * sample_rate too small)");
    if (f->kind != 0 && c.cap > (1 << 18))
        return fail(FDSP_EINVAL, "reverb4_stereo / fdsp_fdn_create: delays of more than 2^18 samples (too long for the lane-per-frame kernel at this sample rate)");
    const size_t n = b->V;
    fd::FdnState st{};
    hipError_t e = hipMalloc((void**)&st.rings, n * c.ring_stride * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&st.wpos, n * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&st.v1, n * 32 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&st.v2, n * 32 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&st.fb, n * 32 * sizeof(float));
    if (e != hipSuccess) {
        FdnBank tmp;
        tmp.st = st;
        fdn_free(&tmp);
        return fail(e == hipErrorOutOfMemory ? FDSP_ENOMEM : FDSP_EDEVICE, std::string("reverb_stereo buffers: ") + hipGetErrorString(e));
    }
    // The new lines start empty (Delay::set_sample_rate resizes and resets, delay.rs:105-113) -- but a change of rate resets nothing else: the
    // FIRs keep their two samples of history (Fir::set_sample_rate, fir.rs:52-54) and Feedback its value (feedback.rs:125-127), so a tail that
    // is sounding when the rate changes goes on from those, exactly like the reference's
    fd::fdn_launch_reset(c, st, n, b->stream);
    if (f->st.v1) {
        hipMemcpyAsync(st.v1, f->st.v1, n * 32 * sizeof(float), hipMemcpyDeviceToDevice, b->stream);
        hipMemcpyAsync(st.v2, f->st.v2, n * 32 * sizeof(float), hipMemcpyDeviceToDevice, b->stream);
        hipMemcpyAsync(st.fb, f->st.fb, n * 32 * sizeof(float), hipMemcpyDeviceToDevice, b->stream);
        hipStreamSynchronize(b->stream);
    }
    fdn_free(f);  // the old buffers
    f->c = c;
    f->st = st;
    b->sr = sr;
    HIPCHK(hipGetLastError());
    return FDSP_OK;
}

int fdsp_reverb_stereo_create(size_t instances, double room_size, double time, double damping, fdsp_bank** out) {
    return fdsp_reverb_stereo_create_on(-1, instances, room_size, time, damping, out);
}

static int fdn_bank_create_on(int kind, int device, size_t instances, double room_size, double time, double damping, fdsp_bank** out, const fd::FdnDesc* desc = nullptr,
                              const fd::Rv3Filter* flt = nullptr);
int fdsp_reverb_stereo_create_on(int device, size_t instances, double room_size, double time, double damping, fdsp_bank** out) {
    return fdn_bank_create_on(0, device, instances, room_size, time, damping, out);
}
int fdsp_reverb4_stereo_create(size_t instances, double room_size, double time, fdsp_bank** out) {
    return fdn_bank_create_on(1, -1, instances, room_size, time, 0.0, out);
}
int fdsp_reverb4_stereo_create_on(int device, size_t instances, double room_size, double time, fdsp_bank** out) {
    return fdn_bank_create_on(1, device, instances, room_size, time, 0.0, out);
}

int fdsp_fdn_create_on(int device, size_t instances, int lines, const double* delays, int taps, const float* weights, int inputs, int outputs, fdsp_bank** out) {
    if (out) *out = nullptr;
    if (lines != 2 && lines != 4 && lines != 8 && lines != 16 && lines != 32) return fail(FDSP_EINVAL, "fdsp_fdn_create: lines takes 2, 4, 8, 16 or 32 (FrameHadamard needs a power of two; the kernel holds at most 32 lines in registers)");
    if (taps < 1 || taps > 3) return fail(FDSP_EINVAL, "fdsp_fdn_create: taps takes 1..3 (Fir<U1> .. Fir<U3>)");
    if ((inputs != 1 && inputs != 2) || (outputs != 1 && outputs != 2)) return fail(FDSP_EINVAL, "fdsp_fdn_create: inputs / outputs take 1 (split / join) or 2 (multisplit::<U2, _> / multijoin::<U2, _>)");
    if (!delays || !weights) return fail(FDSP_EINVAL, "fdsp_fdn_create: delays or weights NULL");
    fd::FdnDesc d;
    d.lines = lines;
    d.taps = taps;
    d.nin = inputs;
    d.nout = outputs;
    for (int i = 0; i < lines; i++) {
        if (!(delays[i] >= 0.0) || !(delays[i] < 1e6)) return fail(FDSP_EINVAL, "fdsp_fdn_create: a delay is negative or not a number (Delay::new asserts time >= 0)");
        d.delay[i] = delays[i];
    }
    for (int j = 0; j < taps; j++) d.w[j] = weights[j];
    return fdn_bank_create_on(2, device, instances, 1.0, 1.0, 0.0, out, &d);
}
int fdsp_fdn_create(size_t instances, int lines, const double* delays, int taps, const float* weights, int inputs, int outputs, fdsp_bank** out) {
    return fdsp_fdn_create_on(-1, instances, lines, delays, taps, weights, inputs, outputs, out);
}
static int rv3_create(int device, size_t instances, double time, double diffusion, const fd::Rv3Filter& flt, fdsp_bank** out) {
    if (out) *out = nullptr;
    if (!(time > 0.0) || !(diffusion >= 0.0 && diffusion <= 1.0) || !(flt.cutoff > 0.0f))
        return fail(FDSP_EINVAL, "fdsp_reverb3_stereo_create: time > 0, diffusion in 0..1, the loop filter's cutoff > 0 Hz");
    return fdn_bank_create_on(3, device, instances, 1.0, time, diffusion, out, nullptr, &flt);
}
int fdsp_reverb3_stereo_create_on(int device, size_t instances, double time, double diffusion, float lowpole_cutoff, fdsp_bank** out) {
    fd::Rv3Filter f;
    f.kind = 0;
    f.cutoff = lowpole_cutoff;
    return rv3_create(device, instances, time, diffusion, f, out);
}
int fdsp_reverb3_stereo_create(size_t instances, double time, double diffusion, float lowpole_cutoff, fdsp_bank** out) {
    return fdsp_reverb3_stereo_create_on(-1, instances, time, diffusion, lowpole_cutoff, out);
}
int fdsp_reverb3_stereo_svf_create_on(int device, size_t instances, double time, double diffusion, int svf_mode, float cutoff, float q, float gain, fdsp_bank** out) {
    if (out) *out = nullptr;
    if (svf_mode < 0 || svf_mode > 8 || !(q > 0.0f) || !(gain > 0.0f)) return fail(FDSP_EINVAL, "fdsp_reverb3_stereo_svf_create: svf_mode 0..8 (FDSP_SVF_*), q > 0, gain > 0 (an amplitude, used by bell / lowshelf / highshelf)");
    fd::Rv3Filter f;
    f.kind = 1;
    f.mode = svf_mode;
    f.cutoff = cutoff; f.q = q; f.gain = gain;
    return rv3_create(device, instances, time, diffusion, f, out);
}
int fdsp_reverb3_stereo_svf_create(size_t instances, double time, double diffusion, int svf_mode, float cutoff, float q, float gain, fdsp_bank** out) {
    return fdsp_reverb3_stereo_svf_create_on(-1, instances, time, diffusion, svf_mode, cutoff, q, gain, out);
}

static int fdn_bank_create_on(int kind, int device, size_t instances, double room_size, double time, double damping, fdsp_bank** out, const fd::FdnDesc* desc,
                              const fd::Rv3Filter* flt) {
    if (!out) return fail(FDSP_EINVAL, "out is NULL");
    *out = nullptr;
    if (instances == 0 || !(room_size > 0.0) || !(time > 0.0)) return fail(FDSP_EINVAL, "bad reverb_stereo / reverb4_stereo arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(FDSP_EDEVICE, "no HIP device available: the fundsp_hip engine has no CPU fallback");
    if (device < 0) device = current_device();
    if (device < 0 || device >= ndev || device >= MAX_DEVICES) return fail(FDSP_EINVAL, "device index out of range");
    DeviceGuard guard(device);
    fdsp_bank* b = new fdsp_bank();
    b->device = device;
    b->fdn = new FdnBank();
    b->fdn->kind = kind;
    b->fdn->room = room_size;
    b->fdn->time = time;
    b->fdn->damping = damping;
    if (desc) b->fdn->desc = *desc;
    if (flt) b->fdn->flt = *flt;
    b->fdn->st = fd::FdnState{};
    b->ops = nullptr;
    b->V = instances;
    b->stride = instances;
    b->nslots = 0;
    b->slots = nullptr;
    b->timed = false;
    b->sr = FDSP_DEFAULT_SR;
    if (hipStreamCreate(&b->stream) != hipSuccess || hipEventCreate(&b->e0) != hipSuccess ||
        hipEventCreate(&b->e1) != hipSuccess) {
        fdsp_bank_destroy(b);
        return fail(FDSP_EDEVICE, "stream/event creation failed");
    }
    int rc = fdn_configure(b, b->sr);
    if (rc != FDSP_OK) {
        fdsp_bank_destroy(b);
        return rc;
    }
    hipError_t e = sync_bank_stream(b);
    if (e != hipSuccess) {
        fdsp_bank_destroy(b);
        return fail(FDSP_EDEVICE, hipGetErrorString(e));
    }
    *out = b;
    return FDSP_OK;
}

void fdsp_bank_destroy(fdsp_bank* b) {
    if (!b) return;
    DeviceGuard guard(b->device);
    // a render that ran on a caller's stream may still be reading the slots: wait for its completion event first
    if (b->ext_pending && b->e1) hipEventSynchronize(b->e1);
    if (b->stream) sync_bank_stream(b);
    if (b->fdn) {
        fdn_free(b->fdn);
        rv3_free_persistent(b->fdn);
        fdn_free_stage(b->fdn);
        delete b->fdn;
        b->fdn = nullptr;
    }
    if (b->slots) hipFree(b->slots);
    if (b->ring) hipFree(b->ring);
    if (b->ev) hipFree(b->ev);
    if (b->ev_fade) hipFree(b->ev_fade);
    if (b->mix_part) hipFree(b->mix_part);
    if (b->panw) hipFree(b->panw);
    if (b->st_in) hipFree(b->st_in);
    if (b->st_out) hipFree(b->st_out);
    if (b->pin_in) hipHostFree(b->pin_in);
    if (b->pin_out) hipHostFree(b->pin_out);
    if (b->e0) hipEventDestroy(b->e0);
    if (b->e1) hipEventDestroy(b->e1);
    if (b->stream) hipStreamDestroy(b->stream);
    delete b;
}

// `Clone` of a FunDSP node (AudioNode: Clone, audionode.rs:35; Net / Sequencer clone their units routinely): a new
// bank of the same kind on the same device that continues exactly where `src` stands -- slots (parameters, coefficients,
// state), delay rings, sample rate, arithmetic mode, launch options, scheduler events and clock, reverb line state.
int fdsp_bank_clone(const fdsp_bank* src, fdsp_bank** out) {
    if (!src || !out) return fail(FDSP_EINVAL, "bank or out is NULL");
    *out = nullptr;
    DeviceGuard guard(src->device);
    // everything queued on the source's stream (and a render on a caller's stream) lands before the copy reads it
    if (src->ext_pending && src->e1) HIPCHK(hipEventSynchronize(src->e1));
    HIPCHK(hipStreamSynchronize(src->stream));
    fdsp_bank* b = nullptr;
    int rc;
    if (src->fdn)
        rc = fdn_bank_create_on(src->fdn->kind, src->device, src->V, src->fdn->room, src->fdn->time, src->fdn->damping, &b, &src->fdn->desc, &src->fdn->flt);
    else
        rc = fdsp_bank_create_on(src->device, src->ops->name.c_str(), src->V, src->ring_frames, &b);
    if (rc != FDSP_OK) return rc;
    auto bail = [&](hipError_t e, const char* what) {
        fdsp_bank_destroy(b);
        return fail(e == hipErrorOutOfMemory ? FDSP_ENOMEM : FDSP_EDEVICE, std::string("fdsp_bank_clone: ") + what + ": " + hipGetErrorString(e));
    };
    hipError_t e = hipSuccess;
    if (src->fdn) {
        if (src->sr != b->sr) {  // the rings' capacity and lengths follow the sample rate
            e = sync_bank_stream(b);
            if (e != hipSuccess) return bail(e, "sync");
            rc = fdn_configure(b, src->sr);
            if (rc != FDSP_OK) {
                fdsp_bank_destroy(b);
                return rc;
            }
        }
        const size_t n = src->V;
        if (src->fdn->kind == 3) {
            const fd::Rv3State &a3 = src->fdn->st3, &d3 = b->fdn->st3;
            e = hipMemcpyAsync(d3.rings, a3.rings, n * src->fdn->c3.ring_stride * sizeof(float), hipMemcpyDeviceToDevice, b->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(d3.pre, a3.pre, n * 4 * (fd::RV3_PRE_CAP + 64) * sizeof(float), hipMemcpyDeviceToDevice, b->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(d3.wpos, a3.wpos, n * sizeof(int), hipMemcpyDeviceToDevice, b->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(d3.wpre, a3.wpre, n * sizeof(int), hipMemcpyDeviceToDevice, b->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(d3.fval, a3.fval, n * 32 * sizeof(float), hipMemcpyDeviceToDevice, b->stream);
            if (e != hipSuccess) return bail(e, "reverb3 state");
        } else {
        const fd::FdnState &a = src->fdn->st, &d = b->fdn->st;
        e = hipMemcpyAsync(d.rings, a.rings, n * src->fdn->c.ring_stride * sizeof(float), hipMemcpyDeviceToDevice, b->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d.wpos, a.wpos, n * sizeof(int), hipMemcpyDeviceToDevice, b->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d.v1, a.v1, n * 32 * sizeof(float), hipMemcpyDeviceToDevice, b->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d.v2, a.v2, n * 32 * sizeof(float), hipMemcpyDeviceToDevice, b->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d.fb, a.fb, n * 32 * sizeof(float), hipMemcpyDeviceToDevice, b->stream);
        if (e != hipSuccess) return bail(e, "reverb state");
        }
    } else {
        b->sr = src->sr;
        e = hipMemcpyAsync(b->slots, src->slots, (size_t)(src->nslots > 0 ? src->nslots : 1) * src->stride * sizeof(float), hipMemcpyDeviceToDevice, b->stream);
        if (e != hipSuccess) return bail(e, "slots");
        if (src->ring) {
            e = hipMemcpyAsync(b->ring, src->ring, (size_t)src->ops->nrings * src->ring_cap * src->stride * sizeof(float) + 64, hipMemcpyDeviceToDevice, b->stream);
            if (e != hipSuccess) return bail(e, "delay rings");
        }
        b->ring_check_pending = src->ring_check_pending;
        b->ring_short = src->ring_short;
        if (src->ev) {
            e = hipMalloc((void**)&b->ev, 4 * src->stride * sizeof(double));
            if (e == hipSuccess) e = hipMalloc((void**)&b->ev_fade, src->stride * sizeof(int));
            if (e == hipSuccess) e = hipMemcpyAsync(b->ev, src->ev, 4 * src->stride * sizeof(double), hipMemcpyDeviceToDevice, b->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(b->ev_fade, src->ev_fade, src->stride * sizeof(int), hipMemcpyDeviceToDevice, b->stream);
            if (e != hipSuccess) return bail(e, "events");
        }
        b->seq_time = src->seq_time;
        b->ev_host = src->ev_host;
        b->ev_dirty = src->ev_dirty;
        b->ev_max_start = src->ev_max_start;
        b->ev_min_end = src->ev_min_end;
        b->ev_max_fade_in_end = src->ev_max_fade_in_end;
        b->ev_min_fade_out_start = src->ev_min_fade_out_start;
    }
    if (src->panw) {
        e = hipMalloc((void**)&b->panw, 2 * src->stride * sizeof(float));
        if (e == hipSuccess) e = hipMemcpyAsync(b->panw, src->panw, 2 * src->stride * sizeof(float), hipMemcpyDeviceToDevice, b->stream);
        if (e != hipSuccess) return bail(e, "pan weights");
    }
    if (src->fdn) b->fdn->bus = src->fdn->bus;
    b->math = src->math;
    b->opt_pipe_split = src->opt_pipe_split;
    b->opt_time_split = src->opt_time_split;
    b->opt_fdn_kernel = src->opt_fdn_kernel;
    b->opt_timing = src->opt_timing;
    e = sync_bank_stream(b);
    if (e != hipSuccess) return bail(e, "copy");
    *out = b;
    return FDSP_OK;
}

// `wet * node` / `dry * multipass() & wet * node` around a reverb / network bank (fd_fdn.hpp FdnBus): a host-side setting, read at every launch
int fdsp_bank_set_bus(fdsp_bank* b, int mode, float wet, float dry) {
    if (!b) return fail(FDSP_EINVAL, "fdsp_bank_set_bus: bank NULL");
    if (!b->fdn) return fail(FDSP_ENOTSUP, "fdsp_bank_set_bus: reverb / network banks only (a run-time compiled graph carries its bus in the graph: fdsp_graph_compile)");
    if (mode < FDSP_BUS_NONE || mode > FDSP_BUS_DRY_WET) return fail(FDSP_EINVAL, "fdsp_bank_set_bus: mode takes FDSP_BUS_NONE, FDSP_BUS_WET or FDSP_BUS_DRY_WET");
    if (mode == FDSP_BUS_DRY_WET && b->fdn->c.nin != b->fdn->c.nout)
        return fail(FDSP_EINVAL, "fdsp_bank_set_bus: a Bus needs as many outputs as inputs (Bus<X, Y>: Y::Inputs = X::Inputs, Y::Outputs = X::Outputs; multipass::<N>() is N -> N)");
    b->fdn->bus.mode = mode;
    b->fdn->bus.wet = wet;
    b->fdn->bus.dry = dry;
    return FDSP_OK;
}

int fdsp_bank_get_bus(const fdsp_bank* b, int* mode, float* wet, float* dry) {
    if (!b) return fail(FDSP_EINVAL, "fdsp_bank_get_bus: bank NULL");
    if (!b->fdn) return fail(FDSP_ENOTSUP, "fdsp_bank_get_bus: reverb / network banks only");
    if (mode) *mode = b->fdn->bus.mode;
    if (wet) *wet = b->fdn->bus.wet;
    if (dry) *dry = b->fdn->bus.dry;
    return FDSP_OK;
}

int fdsp_bank_inputs(const fdsp_bank* b) { return b ? (b->fdn ? b->fdn->c.nin : b->ops->nin) : FDSP_EINVAL; }
int fdsp_bank_outputs(const fdsp_bank* b) { return b ? (b->fdn ? b->fdn->c.nout : b->ops->nout) : FDSP_EINVAL; }
size_t fdsp_bank_voices(const fdsp_bank* b) { return b ? b->V : 0; }

int fdsp_bank_set_sample_rate(fdsp_bank* b, double sr) {
    if (!b || !(sr > 0.0)) return fail(FDSP_EINVAL, "bad bank or sample rate");
    DeviceGuard guard(b->device);
    if (b->fdn) {
        if (sr == b->sr) return FDSP_OK;  // Delay::set_sample_rate: nothing happens unless the rate changes
        HIPCHK(await_last_render(b));
        HIPCHK(sync_bank_stream(b));
        return fdn_configure(b, sr);  // sets b->sr on success only
    }
    b->sr = sr;
    HIPCHK(await_last_render(b));
    before_update_launch(b, 0, b->V);
    b->ops->lifecycle(b->slots, b->stride, 0, b->stride, 1, sr, nullptr, b->aux, b->ring, b->ring_cap, b->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(sync_bank_stream(b));
    return FDSP_OK;
}

int fdsp_bank_reset(fdsp_bank* b) {
    if (!b) return fail(FDSP_EINVAL, "bank is NULL");
    DeviceGuard guard(b->device);
    if (b->fdn) {
        HIPCHK(await_last_render(b));
        if (b->fdn->kind == 3) fd::rv3_launch_reset(b->fdn->c3, b->fdn->st3, b->V, b->stream);
        else fd::fdn_launch_reset(b->fdn->c, b->fdn->st, b->V, b->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(sync_bank_stream(b));
        return FDSP_OK;
    }
    HIPCHK(await_last_render(b));
    b->ops->lifecycle(b->slots, b->stride, 0, b->stride, 2, b->sr, nullptr, b->aux, b->ring, b->ring_cap, b->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(sync_bank_stream(b));
    return FDSP_OK;
}

int fdsp_bank_set_seed(fdsp_bank* b, const uint64_t* h_seeds, size_t first, size_t count) {
    if (!b) return fail(FDSP_EINVAL, "bank is NULL");
    DeviceGuard guard(b->device);
    if (b->fdn) return FDSP_OK;  // no node of reverb_stereo uses its hash (Delay, Fir, Panner: default set_hash)
    if (int rc = check_range(b, first, count)) return rc;
    if (count == 0) return FDSP_OK;
    uint64_t* d = nullptr;
    if (h_seeds) {
        HIPCHK(hipMalloc((void**)&d, count * sizeof(uint64_t)));
        hipError_t e = hipMemcpyAsync(d, h_seeds, count * sizeof(uint64_t), hipMemcpyHostToDevice, b->stream);
        if (e != hipSuccess) {
            hipFree(d);
            return fail(FDSP_EDEVICE, hipGetErrorString(e));
        }
    }
    HIPCHK(await_last_render(b));
    b->ops->lifecycle(b->slots, b->stride, first, count, 3, b->sr, d, b->aux, b->ring, b->ring_cap, b->stream);
    hipError_t e = sync_bank_stream(b);
    if (d) hipFree(d);
    if (e != hipSuccess) return fail(FDSP_EDEVICE, hipGetErrorString(e));
    return FDSP_OK;
}

int fdsp_bank_slot_count(const fdsp_bank* b) { return b ? b->nslots : FDSP_EINVAL; }
const char* fdsp_bank_slot_name(const fdsp_bank* b, int slot) {
    return (b && b->ops && slot >= 0 && slot < b->nslots) ? b->ops->slots[slot].name.c_str() : nullptr;
}
int fdsp_bank_slot_kind(const fdsp_bank* b, int slot) {
    return (b && b->ops && slot >= 0 && slot < b->nslots) ? b->ops->slots[slot].kind : FDSP_EINVAL;
}

__global__ void k_fill_slot(float* __restrict__ row, float value, size_t V) {
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < V) row[v] = value;
}

static int set_words(fdsp_bank* b, int slot, const void* h_words, size_t first, size_t count) {
    HIPCHK(await_last_render(b));
    HIPCHK(hipMemcpyAsync(b->slots + (size_t)slot * b->stride + first, h_words, count * sizeof(float),
                          hipMemcpyHostToDevice, b->stream));
    HIPCHK(sync_bank_stream(b));  // h_words is borrowed for the call only
    return FDSP_OK;
}

int fdsp_bank_set_param(fdsp_bank* b, const char* name, const float* h_values, size_t first, size_t count) {
    if (b && b->fdn) return fail(FDSP_EINVAL, "reverb_stereo banks have no named slots (parameters are fixed at creation)");
    if (!b || !h_values) return fail(FDSP_EINVAL, "bank or values NULL");
    DeviceGuard guard(b->device);
    int s = find_slot(b, name);
    if (s < 0) return fail(FDSP_EINVAL, std::string("unknown slot: ") + (name ? name : "(null)"));
    if (int rc = check_range(b, first, count)) return rc;
    if (count == 0) return FDSP_OK;
    if (int rc = set_words(b, s, h_values, first, count)) return rc;
    // re-derive coefficients like the reference setters do (idempotent for untouched voices)
    HIPCHK(await_last_render(b));
    before_update_launch(b, first, count);
    b->ops->lifecycle(b->slots, b->stride, first, count, 1, b->sr, nullptr, b->aux, b->ring, b->ring_cap, b->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(sync_bank_stream(b));  // (the bank's stream is idle when a setter returns: order_after_bank_stream relies on it under capture)
    return FDSP_OK;
}

int fdsp_bank_set_param_all(fdsp_bank* b, const char* name, float value) {
    if (b && b->fdn) return fail(FDSP_EINVAL, "reverb_stereo banks have no named slots (parameters are fixed at creation)");
    if (!b) return fail(FDSP_EINVAL, "bank is NULL");
    // One value for every voice -- Shared::set_value of a control all voices watch (shared.rs:98-101), e.g. the gate of config 4's
    // `var(gate) >> adsr_live` between two launches: filled on the device, in stream order behind the last render, no host copy and
    // no host wait (the per-voice form above borrows a host array and has to wait for its copy).
    DeviceGuard guard(b->device);
    int s = find_slot(b, name);
    if (s < 0) return fail(FDSP_EINVAL, std::string("unknown slot: ") + (name ? name : "(null)"));
    if (b->V == 0) return FDSP_OK;
    HIPCHK(await_last_render(b));
    hipLaunchKernelGGL(k_fill_slot, dim3((unsigned)((b->V + 255) / 256)), dim3(256), 0, b->stream, b->slots + (size_t)s * b->stride, value, b->V);
    b->async_param_pending = true;
    before_update_launch(b, 0, b->V);
    b->ops->lifecycle(b->slots, b->stride, 0, b->V, 1, b->sr, nullptr, b->aux, b->ring, b->ring_cap, b->stream);
    HIPCHK(hipGetLastError());
    return FDSP_OK;
}

int fdsp_bank_set_param_u64(fdsp_bank* b, const char* name, const uint64_t* h_values, size_t first, size_t count) {
    if (b && b->fdn) return fail(FDSP_EINVAL, "reverb_stereo banks have no named slots (parameters are fixed at creation)");
    if (!b || !h_values || !name) return fail(FDSP_EINVAL, "bank, name or values NULL");
    DeviceGuard guard(b->device);
    int lo = find_slot(b, (std::string(name) + ".lo").c_str());
    int hi = find_slot(b, (std::string(name) + ".hi").c_str());
    if (lo < 0 || hi < 0) return fail(FDSP_EINVAL, std::string("unknown u64 slot: ") + name);
    if (int rc = check_range(b, first, count)) return rc;
    if (count == 0) return FDSP_OK;
    std::vector<uint32_t> wl(count), wh(count);
    for (size_t i = 0; i < count; i++) {
        wl[i] = (uint32_t)h_values[i];
        wh[i] = (uint32_t)(h_values[i] >> 32);
    }
    if (int rc = set_words(b, lo, wl.data(), first, count)) return rc;
    if (int rc = set_words(b, hi, wh.data(), first, count)) return rc;
    HIPCHK(await_last_render(b));
    before_update_launch(b, first, count);
    b->ops->lifecycle(b->slots, b->stride, first, count, 1, b->sr, nullptr, b->aux, b->ring, b->ring_cap, b->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(sync_bank_stream(b));
    return FDSP_OK;
}

int fdsp_bank_get_slot(fdsp_bank* b, const char* name, float* h_values, size_t first, size_t count) {
    if (b && b->fdn) return fail(FDSP_EINVAL, "reverb_stereo banks have no named slots (parameters are fixed at creation)");
    if (!b || !h_values) return fail(FDSP_EINVAL, "bank or values NULL");
    DeviceGuard guard(b->device);
    int s = find_slot(b, name);
    if (s < 0) return fail(FDSP_EINVAL, std::string("unknown slot: ") + (name ? name : "(null)"));
    if (int rc = check_range(b, first, count)) return rc;
    HIPCHK(await_last_render(b));
    HIPCHK(sync_bank_stream(b));
    HIPCHK(hipMemcpy(h_values, b->slots + (size_t)s * b->stride + first, count * sizeof(float), hipMemcpyDeviceToHost));
    return FDSP_OK;
}

int fdsp_bank_get_state(fdsp_bank* b, float* h_slots) {
    if (b && b->fdn) return fail(FDSP_EINVAL, "reverb_stereo banks have no named slots (parameters are fixed at creation)");
    if (!b || !h_slots) return fail(FDSP_EINVAL, "bank or buffer NULL");
    DeviceGuard guard(b->device);
    HIPCHK(await_last_render(b));
    HIPCHK(sync_bank_stream(b));
    if (b->nslots == 0) return FDSP_OK;
    HIPCHK(hipMemcpy2D(h_slots, b->V * sizeof(float), b->slots, b->stride * sizeof(float), b->V * sizeof(float),
                       (size_t)b->nslots, hipMemcpyDeviceToHost));
    return FDSP_OK;
}

int fdsp_bank_set_state(fdsp_bank* b, const float* h_slots) {
    if (b && b->fdn) return fail(FDSP_EINVAL, "reverb_stereo banks have no named slots (parameters are fixed at creation)");
    if (!b || !h_slots) return fail(FDSP_EINVAL, "bank or buffer NULL");
    DeviceGuard guard(b->device);
    HIPCHK(await_last_render(b));
    HIPCHK(sync_bank_stream(b));
    if (b->nslots == 0) return FDSP_OK;
    HIPCHK(hipMemcpy2D(b->slots, b->stride * sizeof(float), h_slots, b->V * sizeof(float), b->V * sizeof(float),
                       (size_t)b->nslots, hipMemcpyHostToDevice));
    return FDSP_OK;
}

int fdsp_bank_process(fdsp_bank* b, size_t frames, const float* d_in, float* d_out, int layout, size_t frame_stride,
                      int mode, void* stream) {
    if (!b) return fail(FDSP_EINVAL, "bank is NULL");
    DeviceGuard guard(b->device);
    if (frames == 0) return FDSP_OK;  // size == 0 is a legal no-op (audionode.rs:82)
    if (!d_out) return fail(FDSP_EINVAL, "d_out is NULL");
    if (fdsp_bank_inputs(b) > 0 && !d_in) return fail(FDSP_EINVAL, "d_in is NULL but the graph has inputs");
    if (layout != FDSP_LAYOUT_VOICE_MINOR && layout != FDSP_LAYOUT_PLANAR) return fail(FDSP_EINVAL, "bad layout");
    if (mode != FDSP_MODE_PROCESS && mode != FDSP_MODE_TICK) return fail(FDSP_EINVAL, "bad mode");
    if (layout == FDSP_LAYOUT_PLANAR && frame_stride < frames) return fail(FDSP_EINVAL, "frame_stride < frames");
    hipStream_t s = stream ? (hipStream_t)stream : b->stream;
    if (int rc = order_after_bank_stream(b, s)) return rc;  // order after pending parameter updates
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (s != b->stream) hipStreamIsCapturing(s, &cap);
    const bool capturing = cap != hipStreamCaptureStatusNone;  // a captured launch leaves the timing events alone
    // a previous render may still be running on ANOTHER stream (caller streams differ from call to call, or the last one
    // ran on a caller stream and this one runs on the bank's): it reads and writes the same voice state, so order behind it
    if (int rc = check_ring_need(b, capturing)) return rc;
    if (b->ext_pending && !capturing) HIPCHK(hipStreamWaitEvent(s, b->e1, 0));
    // the per-launch event pair is optional ("timing" = 0): a real-time host that renders one 64-frame block per call
    // saves two event records per launch; the completion event is then recorded only where ordering needs it (a
    // caller's stream)
    const bool timing = timing_on(b);
    if (!capturing && timing) HIPCHK(hipEventRecord(b->e0, s));
    resolve_opts(b);
    if (b->fdn) {  // Feedback::process is the per-sample tick (feedback.rs:136-146): both modes are the same arithmetic but for the joins (MultiJoin / Join)
        FdnBank* f = b->fdn;
        const int tick = mode == FDSP_MODE_TICK ? 1 : 0, nin = f->c.nin, nout = f->c.nout;
        // voice-minor buffers of banks with at least a tile of instances go through the planar staging copy (the lane = frame kernels read
        // 256-byte runs of it instead of gathering a line per frame); the staging buffer grows outside captures only, like the partial mixes
        bool staged = layout == FDSP_LAYOUT_VOICE_MINOR && b->V >= 64 && (f->kind == 3 || f->c.generic || fd::tl_opts.fdn_kernel == 0 || f->c.sections == 2);
        auto render = [&](const float* pi, float* po, size_t fs, int lay) {
            if (f->kind == 3) fd::rv3_launch_render(f->c3, f->st3, b->V, pi, po, frames, fs, lay, s, f->bus);   // (Reverb has no process override: one arithmetic)
            else fd::fdn_launch_render(f->c, f->st, b->V, pi, po, frames, fs, lay, tick, s, f->bus);
        };
        const size_t need = b->V * (size_t)(nin + nout) * frames;
        if (staged && need > f->stage_n) {
            if (capturing) staged = false;
            else {
                fdn_free_stage(f);  // (hipFree waits for launches that still use the old buffer)
                if (hipMalloc((void**)&f->stage, need * sizeof(float)) == hipSuccess) f->stage_n = need;
                else { (void)hipGetLastError(); f->stage = nullptr; staged = false; }   // no room: the kernels' own voice-minor path
            }
        }
        if (staged) {
            float* pin = f->stage;
            float* pout = f->stage + b->V * (size_t)nin * frames;
            fd::fdn_launch_transpose(d_in, pin, b->V, frames, nin, true, s);
            render(pin, pout, frames, FDSP_LAYOUT_PLANAR);
            fd::fdn_launch_transpose(pout, d_out, b->V, frames, nout, false, s);
        } else {
            render(d_in, d_out, frame_stride, layout);
        }
    } else if (b->math == FDSP_MATH_FAST && b->ops->render_fast)
        b->ops->render_fast(b->slots, b->stride, b->V, d_in, d_out, frames, frame_stride, layout, mode, b->aux, b->ring, b->ring_cap, s);
    else
        b->ops->render(b->slots, b->stride, b->V, d_in, d_out, frames, frame_stride, layout, mode, b->aux, b->ring, b->ring_cap, s);
    b->last_kernel = fd::tl_opts.last_kernel;
    HIPCHK(hipGetLastError());
    if (!capturing) {
        if (timing || s != b->stream) HIPCHK(hipEventRecord(b->e1, s));
        b->timed = timing;
        b->ext_pending = s != b->stream;
    }
    return FDSP_OK;
}

// ---- render + mix-down in one launch -------------------------------------------------------------------------------
namespace {
int mix_channels(const fdsp_bank* b, int mix) { return mix == FDSP_MIX_PAN ? 2 : fdsp_bank_outputs(b); }
// the pan weights exist from the first use on; every voice starts in the centre (pan = 0: Panner::new, pan.rs:41-45)
int ensure_panw(fdsp_bank* b) {
    if (b->panw) return FDSP_OK;
    hipError_t e = hipMalloc((void**)&b->panw, 2 * b->stride * sizeof(float));
    if (e != hipSuccess) return fail(FDSP_ENOMEM, std::string("hipMalloc(pan weights): ") + hipGetErrorString(e));
    hipLaunchKernelGGL(k_pan_weights, dim3((unsigned)((b->stride + 255) / 256)), dim3(256), 0, b->stream, (const float*)nullptr, b->panw,
                       b->panw + b->stride, b->stride, b->V);
    HIPCHK(hipGetLastError());
    HIPCHK(sync_bank_stream(b));
    return FDSP_OK;
}
int mix_reserve(fdsp_bank* b, size_t floats) {
    if (floats <= b->mix_part_n) return FDSP_OK;
    HIPCHK(sync_bank_stream(b));
    if (b->ext_pending) HIPCHK(hipEventSynchronize(b->e1));  // a render on a caller's stream may still write the old buffer
    if (b->mix_part) hipFree(b->mix_part);
    b->mix_part = nullptr;
    b->mix_part_n = 0;
    hipError_t e = hipMalloc((void**)&b->mix_part, floats * sizeof(float));
    if (e != hipSuccess) return fail(FDSP_ENOMEM, std::string("hipMalloc(partial mixes): ") + hipGetErrorString(e));
    b->mix_part_n = floats;
    return FDSP_OK;
}
}  // namespace

int fdsp_bank_set_pan(fdsp_bank* b, const float* h_pan, size_t first, size_t count) {
    if (!b || !h_pan) return fail(FDSP_EINVAL, "bank or h_pan NULL");
    DeviceGuard guard(b->device);
    if (int rc = check_range(b, first, count)) return rc;
    if (count == 0) return FDSP_OK;
    if (int rc = ensure_panw(b)) return rc;
    // a host that pans and then mixes (the FDSP_MIX_PAN flow) may never call fdsp_bank_mix_reserve: run-time compiled graphs build their
    // mix kernels here as well, so that the first fdsp_bank_process_mix compiles nothing
    if (b->ops && b->ops->prepare_mix) b->ops->prepare_mix(b->math == FDSP_MATH_FAST && (bool)b->ops->render_mix_fast);
    HIPCHK(await_last_render(b));
    float* d = nullptr;
    HIPCHK(hipMallocAsync((void**)&d, count * sizeof(float), b->stream));
    hipError_t e = hipMemcpyAsync(d, h_pan, count * sizeof(float), hipMemcpyHostToDevice, b->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_pan_weights, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, b->stream, (const float*)d, b->panw + first,
                           b->panw + b->stride + first, count, ~(size_t)0);
        e = hipGetLastError();
    }
    hipFreeAsync(d, b->stream);
    HIPCHK(e);
    HIPCHK(sync_bank_stream(b));
    return FDSP_OK;
}

int fdsp_bank_mix_reserve(fdsp_bank* b, size_t frames) {
    if (!b) return fail(FDSP_EINVAL, "bank is NULL");
    DeviceGuard guard(b->device);
    const size_t nm = (size_t)(fdsp_bank_outputs(b) > 2 ? fdsp_bank_outputs(b) : 2);
    if (b->ops && b->ops->prepare_mix) b->ops->prepare_mix(b->math == FDSP_MATH_FAST && (bool)b->ops->render_mix_fast);  // run-time compiled graphs: build the mix kernels NOW
    return mix_reserve(b, (b->stride / 64) * nm * frames);
}

int fdsp_bank_process_mix(fdsp_bank* b, size_t frames, const float* d_in, float* d_mix, int mix, int mode, void* stream) {
    if (!b) return fail(FDSP_EINVAL, "bank is NULL");
    DeviceGuard guard(b->device);
    if (frames == 0) return FDSP_OK;
    if (!d_mix) return fail(FDSP_EINVAL, "d_mix is NULL");
    if (fdsp_bank_inputs(b) > 0 && !d_in) return fail(FDSP_EINVAL, "d_in is NULL but the graph has inputs");
    if (mode != FDSP_MODE_PROCESS && mode != FDSP_MODE_TICK) return fail(FDSP_EINVAL, "bad mode");
    if (mix != FDSP_MIX_SUM && mix != FDSP_MIX_PAN) return fail(FDSP_EINVAL, "mix takes FDSP_MIX_SUM or FDSP_MIX_PAN");
    if (mix == FDSP_MIX_PAN && fdsp_bank_outputs(b) != 1) return fail(FDSP_EINVAL, "FDSP_MIX_PAN pans a mono graph; this one has " + std::to_string(fdsp_bank_outputs(b)) + " outputs (use FDSP_MIX_SUM)");
    if (b->fdn) return fail(FDSP_ENOTSUP, "reverb_stereo banks have no fused mix-down: render the instances and call fdsp_sum_voices");
    const bool fast = b->math == FDSP_MATH_FAST && b->ops->render_mix_fast;
    const auto& launch = fast ? b->ops->render_mix_fast : b->ops->render_mix;
    if (!launch) return fail(FDSP_ENOTSUP, "kind '" + b->ops->name + "' was built without a fused mix-down kernel: render voice-out and call fdsp_sum_voices / fdsp_mix_stereo");
    hipStream_t s = stream ? (hipStream_t)stream : b->stream;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (s != b->stream) hipStreamIsCapturing(s, &cap);
    const bool capturing = cap != hipStreamCaptureStatusNone;
    const size_t nm = (size_t)mix_channels(b, mix), groups = b->stride / 64, R = nm * frames;
    if (groups * R > b->mix_part_n || (mix == FDSP_MIX_PAN && !b->panw)) {
        if (capturing) return fail(FDSP_EINVAL, "fdsp_bank_process_mix during a stream capture: call fdsp_bank_mix_reserve (and fdsp_bank_set_pan) before capturing");
        if (int rc = mix_reserve(b, groups * R)) return rc;
        if (mix == FDSP_MIX_PAN) if (int rc = ensure_panw(b)) return rc;
    }
    if (int rc = order_after_bank_stream(b, s)) return rc;
    if (int rc = check_ring_need(b, capturing)) return rc;
    if (b->ext_pending && !capturing) HIPCHK(hipStreamWaitEvent(s, b->e1, 0));
    const bool timing = timing_on(b);
    if (!capturing && timing) HIPCHK(hipEventRecord(b->e0, s));
    resolve_opts(b);
    const bool done = launch(b->slots, b->stride, b->V, d_in, b->mix_part, frames, mix, mode, b->aux, b->ring, b->ring_cap, b->panw, s);
    if (!done) {
        b->timed = false;  // e0 was re-recorded for a launch that did not happen: no event pair to read
        return fail(FDSP_ENOTSUP, "kind '" + b->ops->name + "': no fused mix-down kernel for this graph shape (single-stage graphs without inputs, 3+ outputs with FDSP_MIX_PAN)");
    }
    b->last_kernel = fd::tl_opts.last_kernel;
    // the groups' partials -> d_mix, aligned binary tree (k_mix_tree); the time it takes is part of the render's event pair
    launch_mix_tree(b->mix_part, d_mix, R, (b->V + 63) / 64, s);
    HIPCHK(hipGetLastError());
    if (!capturing) {
        if (timing || s != b->stream) HIPCHK(hipEventRecord(b->e1, s));
        b->timed = timing;
        b->ext_pending = s != b->stream;
    }
    return FDSP_OK;
}

int fdsp_bank_set_ring(fdsp_bank* b, int ring_index, const float* data, size_t frames, size_t first, size_t count) {
    if (!b) return fail(FDSP_EINVAL, "bank is NULL");
    DeviceGuard guard(b->device);
    if (b->fdn || !b->ring) return fail(FDSP_EINVAL, "this bank has no ring memory");
    if (!data) return fail(FDSP_EINVAL, "data is NULL");
    if (ring_index < 0 || ring_index >= b->ops->nrings) return fail(FDSP_EINVAL, "ring index out of range");
    if (frames > b->ring_cap) return fail(FDSP_EINVAL, "more frames than the ring capacity given to fdsp_bank_create_ring");
    if (int rc = check_range(b, first, count)) return rc;
    if (frames == 0 || count == 0) return FDSP_OK;
    // device layout [ring node][position][voice]: transpose the caller's [voice][frame] rows on the host
    std::vector<float> t(frames * count);
    for (size_t v = 0; v < count; v++)
        for (size_t i = 0; i < frames; i++) t[i * count + v] = data[v * frames + i];
    HIPCHK(await_last_render(b));
    float* dst = b->ring + (size_t)ring_index * b->ring_cap * b->stride + first;
    HIPCHK(hipMemcpy2DAsync(dst, b->stride * sizeof(float), t.data(), count * sizeof(float), count * sizeof(float), frames,
                            hipMemcpyHostToDevice, b->stream));
    HIPCHK(sync_bank_stream(b));
    return FDSP_OK;
}

int fdsp_bank_set_events(fdsp_bank* b, const double* events, const int* fade, size_t first, size_t count) {
    if (!b) return fail(FDSP_EINVAL, "bank is NULL");
    DeviceGuard guard(b->device);
    if (b->fdn) return fail(FDSP_EINVAL, "reverb banks have no event scheduler");
    if (!events) return fail(FDSP_EINVAL, "events is NULL");
    if (int rc = check_range(b, first, count)) return rc;
    for (size_t i = 0; i < count; i++) {  // Sequencer::push asserts (sequencer.rs:366-367)
        const double* e = events + 4 * i;
        const double duration = e[1] - e[0];
        if (!(e[2] <= duration && e[3] <= duration) || e[2] < 0.0 || e[3] < 0.0)
            return fail(FDSP_EINVAL, "event fade times must be >= 0 and may not exceed the event's duration");
        if (fade && fade[i] != FDSP_FADE_POWER && fade[i] != FDSP_FADE_SMOOTH) return fail(FDSP_EINVAL, "bad fade curve");
    }
    HIPCHK(await_last_render(b));
    if (!b->ev) {
        HIPCHK(hipMalloc((void**)&b->ev, 4 * b->stride * sizeof(double)));
        HIPCHK(hipMalloc((void**)&b->ev_fade, b->stride * sizeof(int)));
        // voices without an event never play: start = end = +inf
        std::vector<double> init(4 * b->stride, 0.0);
        for (size_t v = 0; v < b->stride; v++) init[v] = init[b->stride + v] = std::numeric_limits<double>::infinity();
        std::vector<int> fi(b->stride, FDSP_FADE_SMOOTH);
        HIPCHK(hipMemcpyAsync(b->ev, init.data(), init.size() * sizeof(double), hipMemcpyHostToDevice, b->stream));
        HIPCHK(hipMemcpyAsync(b->ev_fade, fi.data(), fi.size() * sizeof(int), hipMemcpyHostToDevice, b->stream));
        HIPCHK(sync_bank_stream(b));
    }
    if (b->ev_host.empty()) {
        b->ev_host.assign(4 * b->V, 0.0);
        for (size_t v = 0; v < b->V; v++) b->ev_host[4 * v] = b->ev_host[4 * v + 1] = std::numeric_limits<double>::infinity();
    }
    std::memcpy(b->ev_host.data() + 4 * first, events, 4 * count * sizeof(double));
    b->ev_dirty = true;
    std::vector<double> col(count);
    for (int k = 0; k < 4; k++) {
        for (size_t i = 0; i < count; i++) col[i] = events[4 * i + k];
        HIPCHK(hipMemcpyAsync(b->ev + (size_t)k * b->stride + first, col.data(), count * sizeof(double), hipMemcpyHostToDevice, b->stream));
        HIPCHK(sync_bank_stream(b));
    }
    std::vector<int> fv(count, FDSP_FADE_SMOOTH);
    if (fade) fv.assign(fade, fade + count);
    HIPCHK(hipMemcpyAsync(b->ev_fade + first, fv.data(), count * sizeof(int), hipMemcpyHostToDevice, b->stream));
    HIPCHK(sync_bank_stream(b));
    return FDSP_OK;
}

int fdsp_bank_events_rewind(fdsp_bank* b, double time) {
    if (!b) return fail(FDSP_EINVAL, "bank is NULL");
    b->seq_time = time;
    return FDSP_OK;
}

double fdsp_bank_events_time(const fdsp_bank* b) { return b ? b->seq_time : 0.0; }

namespace {
// fdsp_bank_process_events (d_mix == nullptr: every event's faded samples to d_out) and fdsp_bank_process_events_mix (d_out == nullptr: the
// Sequencer's output, the sum of the events, to d_mix [outputs][frames]) share everything but the launch
int events_render(fdsp_bank* b, size_t frames, const float* d_in, float* d_out, float* d_mix, int mode, void* stream) {
    if (!b) return fail(FDSP_EINVAL, "bank is NULL");
    DeviceGuard guard(b->device);
    if (frames == 0) return FDSP_OK;
    const bool mixing = d_out == nullptr;
    if (!b->ev) return fail(FDSP_EINVAL, "no events set (fdsp_bank_set_events)");
    if (!d_out && !d_mix) return fail(FDSP_EINVAL, mixing ? "d_mix is NULL" : "d_out is NULL");
    if (fdsp_bank_inputs(b) > 0 && !d_in) return fail(FDSP_EINVAL, "d_in is NULL but the graph has inputs");
    if (mode != FDSP_MODE_PROCESS && mode != FDSP_MODE_TICK) return fail(FDSP_EINVAL, "bad mode");
    if (mixing && !b->ops->render_events_mix)
        return fail(FDSP_ENOTSUP, "kind '" + b->ops->name + "' has no fused Sequencer mix: call fdsp_bank_process_events and fdsp_sum_voices");
    hipStream_t s = stream ? (hipStream_t)stream : b->stream;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (s != b->stream) hipStreamIsCapturing(s, &cap);
    const bool capturing = cap != hipStreamCaptureStatusNone;  // a captured launch leaves the timing events alone
    const size_t nm = (size_t)fdsp_bank_outputs(b), groups = b->stride / 64, R = nm * frames;
    if (mixing && groups * R > b->mix_part_n) {
        if (capturing) return fail(FDSP_EINVAL, "fdsp_bank_process_events_mix during a stream capture: call fdsp_bank_mix_reserve before capturing");
        if (int rc = mix_reserve(b, groups * R)) return rc;
    }
    if (int rc = order_after_bank_stream(b, s)) return rc;
    if (int rc = check_ring_need(b, capturing)) return rc;
    if (b->ext_pending && !capturing) HIPCHK(hipStreamWaitEvent(s, b->e1, 0));
    const bool timing = timing_on(b);  // the per-launch event pair is optional, as in fdsp_bank_process
    if (!capturing && timing) HIPCHK(hipEventRecord(b->e0, s));
    // the clock after this launch, advanced exactly as the reference does: one f64 addition per block / per sample
    const double sd = 1.0 / b->sr, t_begin = b->seq_time;
    double t_end = t_begin;
    if (mode == FDSP_MODE_PROCESS)
        for (size_t t0 = 0; t0 < frames; t0 += 64) t_end += sd * (double)(frames - t0 < 64 ? frames - t0 : 64);
    else
        for (size_t t = 0; t < frames; t++) t_end += sd;
    if (b->ev_dirty) {
        const double inf = std::numeric_limits<double>::infinity();
        b->ev_max_start = -inf; b->ev_min_end = inf; b->ev_max_fade_in_end = -inf; b->ev_min_fade_out_start = inf;
        for (size_t v = 0; v < b->V; v++) {
            const double* e = b->ev_host.data() + 4 * v;
            b->ev_max_start = std::max(b->ev_max_start, e[0]);
            b->ev_min_end = std::min(b->ev_min_end, e[1]);
            if (e[2] > 0.0) b->ev_max_fade_in_end = std::max(b->ev_max_fade_in_end, e[0] + e[2]);
            if (e[3] > 0.0) b->ev_min_fade_out_start = std::min(b->ev_min_fade_out_start, e[1] - e[3]);
        }
        b->ev_dirty = false;
    }
    // Every voice inside its event for the whole launch and no fade running in any of its blocks (the kernel's own
    // per-block conditions, evaluated for the first and the last block): the scheduler's output is the units' plain
    // process / tick -- same arithmetic -- so the launch takes the (pipeline) render kernel.
    const bool sustained = !b->ev_host.empty() && b->ev_max_start <= t_begin && b->ev_min_end >= t_end &&
                           b->ev_max_fade_in_end <= t_begin && b->ev_min_fade_out_start >= t_end;
    resolve_opts(b);
    if (mixing) {
        bool done = false;
        if (sustained && b->ops->render_mix)  // (the exact type: the voice scheduler renders exactly, like fdsp_bank_process_events)
            done = b->ops->render_mix(b->slots, b->stride, b->V, d_in, b->mix_part, frames, FDSP_MIX_SUM, mode, b->aux, b->ring, b->ring_cap, b->panw, s);
        if (!done)
            done = b->ops->render_events_mix(b->slots, b->stride, b->V, d_in, b->mix_part, frames, b->ev, b->ev_fade, b->seq_time, b->sr, mode,
                                             b->aux, b->ring, b->ring_cap, s);
        if (!done) {
            b->timed = false;
            return fail(FDSP_ENOTSUP, "kind '" + b->ops->name + "': more than two outputs -- no fused Sequencer mix; call fdsp_bank_process_events and fdsp_sum_voices");
        }
        launch_mix_tree(b->mix_part, d_mix, R, (b->V + 63) / 64, s);
    } else if (sustained)
        b->ops->render(b->slots, b->stride, b->V, d_in, d_out, frames, 0, FDSP_LAYOUT_VOICE_MINOR, mode, b->aux, b->ring,
                       b->ring_cap, s);
    else
        b->ops->render_events(b->slots, b->stride, b->V, d_in, d_out, frames, b->ev, b->ev_fade, b->seq_time, b->sr, mode,
                              b->aux, b->ring, b->ring_cap, s);
    b->last_kernel = fd::tl_opts.last_kernel;
    HIPCHK(hipGetLastError());
    if (!capturing) {
        if (timing || s != b->stream) HIPCHK(hipEventRecord(b->e1, s));
        b->timed = timing;
        b->ext_pending = s != b->stream;
    }
    b->seq_time = t_end;
    return FDSP_OK;
}
}  // namespace

int fdsp_bank_process_events(fdsp_bank* b, size_t frames, const float* d_in, float* d_out, int mode, void* stream) {
    if (b && frames > 0 && !d_out) return fail(FDSP_EINVAL, "d_out is NULL");
    return events_render(b, frames, d_in, d_out, nullptr, mode, stream);
}

int fdsp_bank_process_events_mix(fdsp_bank* b, size_t frames, const float* d_in, float* d_mix, int mode, void* stream) {
    if (b && frames > 0 && !d_mix) return fail(FDSP_EINVAL, "d_mix is NULL");
    return events_render(b, frames, d_in, nullptr, d_mix, mode, stream);
}


namespace {

// grow-only staging buffer; `pinned` picks hipHostMalloc over hipMalloc
hipError_t stage_reserve(float** p, size_t* have, size_t want, bool pinned) {
    if (want <= *have) return hipSuccess;
    if (*p) {
        hipError_t e = pinned ? hipHostFree(*p) : hipFree(*p);
        *p = nullptr;
        *have = 0;
        if (e != hipSuccess) return e;
    }
    size_t n = want + want / 2;
    hipError_t e = pinned ? hipHostMalloc((void**)p, n * sizeof(float), hipHostMallocDefault)
                          : hipMalloc((void**)p, n * sizeof(float));
    if (e != hipSuccess) {
        n = want;
        e = pinned ? hipHostMalloc((void**)p, n * sizeof(float), hipHostMallocDefault)
                   : hipMalloc((void**)p, n * sizeof(float));
    }
    if (e == hipSuccess) *have = n;
    return e;
}

}  // namespace

int fdsp_bank_process_host(fdsp_bank* b, size_t frames, const float* h_in, float* h_out, int layout,
                           size_t frame_stride, int mode) {
    if (!b) return fail(FDSP_EINVAL, "bank is NULL");
    DeviceGuard guard(b->device);
    if (frames == 0) return FDSP_OK;
    if (!h_out) return fail(FDSP_EINVAL, "h_out is NULL");
    const bool planar = layout == FDSP_LAYOUT_PLANAR;
    const size_t row = planar ? frame_stride : frames;
    if (planar && frame_stride < frames) return fail(FDSP_EINVAL, "frame_stride < frames");
    const size_t ni = (size_t)fdsp_bank_inputs(b), no = (size_t)fdsp_bank_outputs(b);
    const size_t n_in = ni * row * b->V, n_out = no * row * b->V;
    if (ni > 0 && !h_in) return fail(FDSP_EINVAL, "h_in is NULL but the graph has inputs");
    hipError_t e = hipSuccess;
    // planar rows longer than `frames`: only the first `frames` samples of each row cross the bus, and the caller's
    // padding is left untouched (AudioNode::process does not write past `size`, audionode.rs:85)
    const bool strided = planar && frame_stride > frames;
    // small transfers (a real-time host's 64-frame blocks): the kernel reads and writes pinned host memory over the
    // bus itself — two copy-engine round trips less than staging through HBM (tools/host_sweep.py: 26-45 us against
    // 44-54 us per call up to 1024 voices; above ~1 MB the pageable copy path wins)
    const bool zc = !strided && n_in <= (size_t)fd::g_zero_copy_max.load() && n_out <= (size_t)fd::g_zero_copy_max.load() &&
                    stage_reserve(&b->pin_in, &b->pin_in_n, n_in, true) == hipSuccess &&
                    stage_reserve(&b->pin_out, &b->pin_out_n, n_out, true) == hipSuccess;
    int rc = FDSP_OK;
    if (zc) {
        if (n_in) memcpy(b->pin_in, h_in, n_in * sizeof(float));
        rc = fdsp_bank_process(b, frames, n_in ? b->pin_in : nullptr, b->pin_out, layout, frame_stride, mode, nullptr);
        if (rc != FDSP_OK) return rc;
        e = sync_bank_stream(b);
        if (e != hipSuccess) return fail(FDSP_EDEVICE, hipGetErrorString(e));
        memcpy(h_out, b->pin_out, n_out * sizeof(float));
        return FDSP_OK;
    }
    e = stage_reserve(&b->st_in, &b->st_in_n, n_in, false);
    if (e == hipSuccess) e = stage_reserve(&b->st_out, &b->st_out_n, n_out, false);
    if (e != hipSuccess) return fail(FDSP_ENOMEM, hipGetErrorString(e));
    if (n_in) {
        if (strided)
            e = hipMemcpy2DAsync(b->st_in, row * sizeof(float), h_in, row * sizeof(float), frames * sizeof(float),
                                 ni * b->V, hipMemcpyHostToDevice, b->stream);
        else
            e = hipMemcpyAsync(b->st_in, h_in, n_in * sizeof(float), hipMemcpyHostToDevice, b->stream);
    }
    if (e == hipSuccess)
        rc = fdsp_bank_process(b, frames, n_in ? b->st_in : nullptr, b->st_out, layout, frame_stride, mode, nullptr);
    if (e == hipSuccess && rc == FDSP_OK) {
        if (strided)
            e = hipMemcpy2DAsync(h_out, row * sizeof(float), b->st_out, row * sizeof(float), frames * sizeof(float),
                                 no * b->V, hipMemcpyDeviceToHost, b->stream);
        else
            e = hipMemcpyAsync(h_out, b->st_out, n_out * sizeof(float), hipMemcpyDeviceToHost, b->stream);
    }
    if (e == hipSuccess) e = sync_bank_stream(b);
    if (e != hipSuccess) return fail(FDSP_EDEVICE, hipGetErrorString(e));
    return rc;
}

int fdsp_bank_synchronize(fdsp_bank* b) {
    if (!b) return fail(FDSP_EINVAL, "bank is NULL");
    DeviceGuard guard(b->device);
    HIPCHK(sync_bank_stream(b));
    b->async_param_pending = false;
    return FDSP_OK;
}

int fdsp_bank_last_kernel_ms(fdsp_bank* b, float* ms) {
    if (!b || !ms) return fail(FDSP_EINVAL, "bank or ms NULL");
    DeviceGuard guard(b->device);
    if (!b->timed) return fail(FDSP_EINVAL, "no process call recorded yet");
    HIPCHK(hipEventSynchronize(b->e1));
    HIPCHK(hipEventElapsedTime(ms, b->e0, b->e1));
    return FDSP_OK;
}

int fdsp_mix_stereo(const float* d_voices, const float* d_pan, float* d_mix, size_t frames, size_t voices,
                    void* stream) {
    if (!d_voices || !d_mix) return fail(FDSP_EINVAL, "NULL buffer");
    if (frames == 0 || voices == 0) return FDSP_OK;
    if (frames > 65535u * 256u) return fail(FDSP_EINVAL, "fdsp_mix_stereo: at most 16 776 960 frames per call");
    DeviceGuard guard(device_of(d_voices));  // the kernels run where the voices live, whatever device is current
    hipStream_t s = (hipStream_t)stream;
    const size_t G = (voices + 63) / 64;
    float* w = nullptr;  // [2][voices] weights, then [G][2][frames] partial mixes
    HIPCHK(hipMallocAsync((void**)&w, (2 * voices + G * 2 * frames) * sizeof(float), s));
    hipLaunchKernelGGL(k_pan_weights, dim3((unsigned)((voices + 255) / 256)), dim3(256), 0, s, d_pan, w, w + voices,
                       voices, ~(size_t)0);
    hipError_t e = launch_mix_rows<true>(d_voices, w, w + voices, w + 2 * voices, d_mix, frames, voices, s);
    hipFreeAsync(w, s);
    HIPCHK(e);
    return FDSP_OK;
}

int fdsp_wavetable_upload(int set, int n_tables, const float* h_pitches, const int* h_lengths, const float* h_data) {
    if (!h_pitches || !h_lengths || !h_data) return fail(FDSP_EINVAL, "NULL table data");
    return upload_table_set(set, n_tables, h_pitches, h_lengths, h_data);
}

int fdsp_wavetable_build(int set) { return build_default_table_set(set); }
int fdsp_wavetable_compute(int set, int* n_tables, float* h_pitches, int* h_lengths, float* h_data, size_t capacity) {
    if (!n_tables) return fail(FDSP_EINVAL, "NULL n_tables");
    std::vector<float> pitches, data;
    std::vector<int> lengths;
    const int rc = compute_default_table_set(set, pitches, lengths, data);
    if (rc != FDSP_OK) return rc;
    *n_tables = (int)pitches.size();
    if (h_pitches) std::memcpy(h_pitches, pitches.data(), sizeof(float) * pitches.size());
    if (h_lengths) std::memcpy(h_lengths, lengths.data(), sizeof(int) * lengths.size());
    if (h_data) {
        if (capacity < data.size()) return fail(FDSP_EINVAL, "capacity too small");
        std::memcpy(h_data, data.data(), sizeof(float) * data.size());
    }
    return FDSP_OK;
}
int fdsp_wave_upload(int slot, int channels, size_t length, const float* h_data) { return upload_wave(slot, channels, length, h_data); }

int fdsp_wavetable_get(int set, int* n_tables, float* h_pitches, int* h_lengths, float* h_data, size_t capacity) {
    if (set < 0 || set >= fd::WT_SETS || !n_tables) return fail(FDSP_EINVAL, "bad set");
    std::lock_guard<std::mutex> lock(g_aux_mutex);
    const HostTableSet& t = g_tables[set];
    *n_tables = t.n;
    if (t.n == 0) return FDSP_OK;
    if (h_pitches) std::memcpy(h_pitches, t.pitch.data(), sizeof(float) * (size_t)t.n);
    if (h_lengths) std::memcpy(h_lengths, t.len.data(), sizeof(int) * (size_t)t.n);
    if (h_data) {
        if (capacity < t.data.size()) return fail(FDSP_EINVAL, "capacity too small");
        std::memcpy(h_data, t.data.data(), t.data.size() * sizeof(float));
    }
    return FDSP_OK;
}

int fdsp_sum_voices(const float* d_in, float* d_out, size_t channels, size_t frames, size_t voices, void* stream) {
    if (!d_in || !d_out) return fail(FDSP_EINVAL, "NULL buffer");
    if (channels == 0 || frames == 0 || voices == 0) return FDSP_OK;
    const size_t rows = channels * frames, G = (voices + 63) / 64;
    if (rows > 65535u * 256u) return fail(FDSP_EINVAL, "fdsp_sum_voices: at most 16 776 960 rows (channels x frames) per call");
    DeviceGuard guard(device_of(d_in));
    hipStream_t s = (hipStream_t)stream;
    float* part = nullptr;  // [G][rows] partial sums
    HIPCHK(hipMallocAsync((void**)&part, G * rows * sizeof(float), s));
    hipError_t e = launch_mix_rows<false>(d_in, nullptr, nullptr, part, d_out, rows, voices, s);
    hipFreeAsync(part, s);
    HIPCHK(e);
    return FDSP_OK;
}

int fdsp_sum_instances(const float* d_in, float* d_out, size_t rows, size_t instances, void* stream) {
    if (!d_in || !d_out) return fail(FDSP_EINVAL, "NULL buffer");
    if (rows == 0 || instances == 0) return FDSP_OK;
    DeviceGuard guard(device_of(d_in));
    // planar output [instance][rows] IS the tree kernel's [groups][rows] shape: the aligned binary tree over the instances
    launch_mix_tree(d_in, d_out, rows, instances, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return FDSP_OK;
}

int fdsp_svf_coefs(int mode, float sr, float cutoff, float q, float gain, float* out6) {
    if (!out6 || mode < 0 || mode > FDSP_SVF_HIGHSHELF) return fail(FDSP_EINVAL, "bad svf mode or NULL out");
    fd::SvfCoefs c = fd::svf_coefs(mode, sr, cutoff, q, gain);
    out6[0] = c.a1; out6[1] = c.a2; out6[2] = c.a3; out6[3] = c.m0; out6[4] = c.m1; out6[5] = c.m2;
    return FDSP_OK;
}

int fdsp_biquad_coefs(int kind, float sr, float f, float q, float gain, float* out5) {
    if (!out5 || kind < 0 || kind > FDSP_BQ_BELL) return fail(FDSP_EINVAL, "bad biquad kind or NULL out");
    fd::BiquadCoefs c = fd::biquad_coefs(kind, sr, f, q, gain);
    out5[0] = c.a1; out5[1] = c.a2; out5[2] = c.b0; out5[3] = c.b1; out5[4] = c.b2;
    return FDSP_OK;
}

double fdsp_rnd1(uint64_t x) { return fd::rnd1(x); }
float fdsp_libm_sinf(float x) { return fd::sinf_musl(x); }
float fdsp_libm_cosf(float x) { return fd::cosf_musl(x); }
uint64_t fdsp_hash1(uint64_t x) { return fd::hash1(x); }

}  // extern "C"

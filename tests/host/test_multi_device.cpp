// One process, several host threads, several GPUs through the C ABI (include/fundsp_hip.h: fdsp_bank_create_on,
// fdsp_comm_create_local, fdsp_mix_allreduce).  Plain C++ against the C header -- what a Rust host would do via FFI.
//
//   test_multi_device --host   no device needed: the multi-device entry points exist, reject bad arguments, and report
//                              FDSP_EDEVICE (never a CPU fallback) from two concurrent threads
//   test_multi_device --gpu    on a box with N >= 1 GPUs: max(2, N) threads, thread t drives a bank of config-3 FM
//                              voices on device t % N (its own voice shard), renders it, mixes it down on its device and
//                              joins an RCCL all-reduce of the [2][frames] partial mixes; checks:
//                                * every shard equals the same voices rendered single-threaded on device 0, bit for bit
//                                * the all-reduced mix equals the sum of the partial mixes (exactly for the 1- and
//                                  2-rank cases, within a few ulps beyond: RCCL's summation order)
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "fundsp_hip.h"

static std::atomic<int> failures{0};
#define EXPECT(cond)                                                    \
    do {                                                                \
        if (!(cond)) {                                                  \
            std::printf("FAIL %s:%d: %s (%s)\n", __FILE__, __LINE__, #cond, fdsp_last_error()); \
            failures++;                                                 \
        }                                                               \
    } while (0)

static const double SR = 48000.0;

// per-voice parameters of BASELINE config 3 (fundsp_amd/workloads.py::fm_svf_params) from the voice index
static void params(uint64_t v, float* f, float* m, float* fc, float* q) {
    double u[4];
    for (int k = 0; k < 4; k++) u[k] = fdsp_rnd1(4 * v + (uint64_t)k);
    const double ff = 55.0 * std::exp2(5.0 * u[0]);
    *f = (float)ff;
    *m = (float)(0.5 + 7.5 * u[1]);
    *fc = (float)std::fmin(ff * std::exp2(4.0 * u[2]), 0.45 * SR);
    *q = (float)(0.5 + 3.5 * u[3]);
}

static fdsp_bank* make_bank(int device, size_t first, size_t count) {
    fdsp_bank* b = nullptr;
    if (fdsp_bank_create_on(device, "fm_svf", count, 0, &b) != FDSP_OK) return nullptr;
    std::vector<float> f(count), m(count), fc(count), q(count);
    std::vector<uint64_t> seed(count);
    for (size_t i = 0; i < count; i++) {
        params(first + i, &f[i], &m[i], &fc[i], &q[i]);
        seed[i] = first + i;
    }
    int rc = fdsp_bank_set_param(b, "0.0.0.0.0.0:value[0]", f.data(), 0, count);
    rc |= fdsp_bank_set_param(b, "0.0.0.0:scalar", f.data(), 0, count);
    rc |= fdsp_bank_set_param(b, "0.0.0:scalar", m.data(), 0, count);
    rc |= fdsp_bank_set_param(b, "0.0:scalar", f.data(), 0, count);
    rc |= fdsp_bank_set_param(b, "1:cutoff", fc.data(), 0, count);
    rc |= fdsp_bank_set_param(b, "1:q", q.data(), 0, count);
    rc |= fdsp_bank_set_sample_rate(b, SR);
    rc |= fdsp_bank_set_seed(b, seed.data(), 0, count);
    if (rc != FDSP_OK) {
        fdsp_bank_destroy(b);
        return nullptr;
    }
    return b;
}

static int host_mode() {
    EXPECT(fdsp_device_count() >= 0);
    fdsp_bank* b = nullptr;
    EXPECT(fdsp_bank_create_on(0, "no_such_kind", 8, 0, &b) == FDSP_EINVAL && b == nullptr);
    EXPECT(fdsp_bank_create_on(0, "fm_svf", 0, 0, &b) == FDSP_EINVAL);
    EXPECT(fdsp_bank_device(nullptr) == FDSP_EINVAL);
    EXPECT(fdsp_comm_ranks(nullptr) == FDSP_EINVAL);
    fdsp_comm* c = nullptr;
    EXPECT(fdsp_comm_create_local(0, nullptr, &c) != FDSP_OK && c == nullptr);
    if (fdsp_device_count() == 0) {  // the product has no CPU fallback: both threads must see FDSP_EDEVICE
        std::vector<std::thread> th;
        for (int t = 0; t < 2; t++)
            th.emplace_back([t] {
                fdsp_bank* bb = nullptr;
                EXPECT(fdsp_bank_create_on(t, "fm_svf", 64, 0, &bb) == FDSP_EDEVICE && bb == nullptr);
                fdsp_comm* cc = nullptr;
                EXPECT(fdsp_comm_create_local(1, nullptr, &cc) == FDSP_EDEVICE && cc == nullptr);
            });
        for (auto& x : th) x.join();
        std::printf("no HIP device: multi-device entry points report FDSP_EDEVICE\n");
    }
    return failures;
}

static int gpu_mode() {
    const int ndev = fdsp_device_count();
    EXPECT(ndev >= 1);
    if (ndev < 1) return failures;
    const int nthreads = ndev > 2 ? ndev : 2;
    const size_t per = 640, T = 64 * 9 + 24;  // voices per shard, frames (a multiple of 8)
    // reference: all voices on device 0 from this thread
    std::vector<float> want(nthreads * per * T);
    {
        fdsp_bank* b = make_bank(0, 0, nthreads * per);
        EXPECT(b != nullptr);
        if (!b) return failures;
        EXPECT(fdsp_bank_device(b) == 0);
        EXPECT(fdsp_bank_process_host(b, T, nullptr, want.data(), FDSP_LAYOUT_PLANAR, T, FDSP_MODE_PROCESS) == FDSP_OK);
        fdsp_bank_destroy(b);
    }
    // a local communicator over the devices the threads use (ranks = distinct devices; with one GPU that is one rank and
    // only thread 0 joins the collective)
    const int nranks = ndev < nthreads ? ndev : nthreads;
    fdsp_comm* comm = nullptr;
    EXPECT(fdsp_comm_create_local(nranks, nullptr, &comm) == FDSP_OK);
    if (!comm) return failures;
    EXPECT(fdsp_comm_ranks(comm) == nranks && fdsp_comm_local_slots(comm) == nranks);
    std::vector<std::vector<float>> got(nthreads, std::vector<float>(per * T));
    std::vector<std::vector<float>> partial(nthreads, std::vector<float>(2 * T)), reduced(nthreads, std::vector<float>(2 * T));
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++)
        th.emplace_back([&, t] {
            const int dev = t % ndev;
            fdsp_bank* b = make_bank(dev, (size_t)t * per, per);
            EXPECT(b != nullptr);
            if (!b) return;
            EXPECT(fdsp_bank_device(b) == dev);
            // device-resident render in the voice-minor layout + mix-down + (one thread per rank) the all-reduce
            EXPECT(hipSetDevice(dev) == hipSuccess);
            float *d_out = nullptr, *d_mix = nullptr;
            hipStream_t s = nullptr;
            EXPECT(hipMalloc((void**)&d_out, per * T * sizeof(float)) == hipSuccess);
            EXPECT(hipMalloc((void**)&d_mix, 2 * T * sizeof(float)) == hipSuccess);
            EXPECT(hipStreamCreate(&s) == hipSuccess);
            // (a clone renders the same shard with the mix-down FUSED into the launch: fdsp_bank_process_mix, no voice-out buffer)
            fdsp_bank* bf = nullptr;
            EXPECT(fdsp_bank_clone(b, &bf) == FDSP_OK);
            float* d_fused = nullptr;
            EXPECT(hipMalloc((void**)&d_fused, 2 * T * sizeof(float)) == hipSuccess);
            std::vector<float> fused(2 * T);
            if (bf) {
                EXPECT(fdsp_bank_process_mix(bf, T, nullptr, d_fused, FDSP_MIX_PAN, FDSP_MODE_PROCESS, s) == FDSP_OK);
                EXPECT(hipMemcpyAsync(fused.data(), d_fused, 2 * T * sizeof(float), hipMemcpyDeviceToHost, s) == hipSuccess);
            }
            EXPECT(fdsp_bank_process(b, T, nullptr, d_out, FDSP_LAYOUT_VOICE_MINOR, 0, FDSP_MODE_PROCESS, s) == FDSP_OK);
            EXPECT(fdsp_mix_stereo(d_out, nullptr, d_mix, T, per, s) == FDSP_OK);
            EXPECT(hipMemcpyAsync(partial[t].data(), d_mix, 2 * T * sizeof(float), hipMemcpyDeviceToHost, s) == hipSuccess);
            EXPECT(hipStreamSynchronize(s) == hipSuccess);
            EXPECT(std::memcmp(fused.data(), partial[t].data(), 2 * T * sizeof(float)) == 0);  // one summation order: bit for bit
            if (bf) fdsp_bank_destroy(bf);
            hipFree(d_fused);
            const bool joins = t < nranks;
            if (joins) {
                EXPECT(fdsp_mix_allreduce(comm, t, d_mix, 2 * T, s) == FDSP_OK);
                EXPECT(fdsp_comm_wait(comm, t, s) == FDSP_OK);   // the copy below is ordered behind the collective
                EXPECT(hipMemcpyAsync(reduced[t].data(), d_mix, 2 * T * sizeof(float), hipMemcpyDeviceToHost, s) == hipSuccess);
            }
            std::vector<float> vm(per * T);
            EXPECT(hipMemcpyAsync(vm.data(), d_out, per * T * sizeof(float), hipMemcpyDeviceToHost, s) == hipSuccess);
            EXPECT(hipStreamSynchronize(s) == hipSuccess);
            for (size_t v = 0; v < per; v++)
                for (size_t i = 0; i < T; i++) got[t][v * T + i] = vm[i * per + v];  // [frame][voice] -> [voice][frame]
            hipFree(d_out);
            hipFree(d_mix);
            hipStreamDestroy(s);
            fdsp_bank_destroy(b);
        });
    for (auto& x : th) x.join();
    for (int t = 0; t < nthreads; t++)
        EXPECT(std::memcmp(got[t].data(), want.data() + (size_t)t * per * T, per * T * sizeof(float)) == 0);
    // the reduced mix of every joining rank = sum over the joining ranks' partial mixes
    for (int t = 0; t < nranks; t++)
        for (size_t i = 0; i < 2 * T; i++) {
            double sum = 0.0, mag = 0.0;
            for (int r = 0; r < nranks; r++) {
                sum += partial[r][i];
                mag += std::fabs(partial[r][i]);
            }
            const double tol = nranks <= 2 ? 0.0 : 4.0 * 1.2e-7 * mag;
            if (std::fabs((double)reduced[t][i] - (nranks <= 2 ? (double)(float)sum : sum)) > tol) {
                EXPECT(!"all-reduced mix differs from the sum of the partial mixes");
                break;
            }
        }
    fdsp_comm_destroy(comm);
    std::printf("%d thread(s) on %d device(s), %d rank(s): shards bit-exact, fused mix-down == mix of the voice-out render, all-reduce checked\n", nthreads, ndev, nranks);
    return failures;
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "--host";
    const int f = mode == "--gpu" ? gpu_mode() : host_mode();
    std::printf("%d failure(s)\n", f);
    return f != 0;
}

// tools/ubench_launch.hip -- what a DEPENDENT chain of small kernels costs on this box, whatever the kernels compute: the floor under
// "one 64-frame block per launch" (BASELINE config 2).  Design tool, not part of the product.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_launch.hip -o tools/ubench_launch && tools/ubench_launch
// Rows: kernel = 16 workgroups x 64 threads that spin for `spin` ticks of the 100 MHz wall clock (0 = returns at once);
//   stream   : N launches back to back on one stream, us per launch (host enqueue rate or device dispatch, whichever is slower)
//   graph    : the same N launches captured into ONE HIP graph, replayed; us per kernel node
//   events   : one launch bracketed by an event pair, hipEventElapsedTime (what fdsp_bank_last_kernel_ms reports)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ void k_spin(unsigned long long cycles, float* sink) {
    const unsigned long long t0 = wall_clock64();   // 100 MHz constant clock
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(1);
    if (cycles == 0xffffffffffffffffull) sink[threadIdx.x] = 1.0f;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    float* sink;
    CHK(hipMalloc((void**)&sink, 4096));
    hipStream_t s;
    CHK(hipStreamCreate(&s));
    const int N = 64, REP = 50;
    for (unsigned long long spin : {0ull, 100ull, 200ull, 400ull}) {   // wall_clock64 counts at 100 MHz: 100 = 1 us
        for (int warm = 0; warm < 200; warm++) hipLaunchKernelGGL(k_spin, dim3(16), dim3(64), 0, s, spin, sink);
        CHK(hipStreamSynchronize(s));
        double t0 = now_us();
        for (int r = 0; r < REP; r++)
            for (int k = 0; k < N; k++) hipLaunchKernelGGL(k_spin, dim3(16), dim3(64), 0, s, spin, sink);
        CHK(hipStreamSynchronize(s));
        const double us_stream = (now_us() - t0) / (REP * N);
        hipGraph_t g;
        hipGraphExec_t ge;
        CHK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int k = 0; k < N; k++) hipLaunchKernelGGL(k_spin, dim3(16), dim3(64), 0, s, spin, sink);
        CHK(hipStreamEndCapture(s, &g));
        CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 5; w++) CHK(hipGraphLaunch(ge, s));
        CHK(hipStreamSynchronize(s));
        t0 = now_us();
        for (int r = 0; r < REP; r++) CHK(hipGraphLaunch(ge, s));
        CHK(hipStreamSynchronize(s));
        const double us_graph = (now_us() - t0) / (REP * N);
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0));
        CHK(hipEventCreate(&e1));
        std::vector<float> ms;
        for (int r = 0; r < 200; r++) {
            CHK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(k_spin, dim3(16), dim3(64), 0, s, spin, sink);
            CHK(hipEventRecord(e1, s));
            CHK(hipEventSynchronize(e1));
            float m;
            CHK(hipEventElapsedTime(&m, e0, e1));
            ms.push_back(m);
        }
        std::sort(ms.begin(), ms.end());
        printf("spin %4llu ticks (%.1f us of work): stream %.2f us/launch | graph replay %.2f us/node | event pair median %.2f us\n", spin, spin / 100.0, us_stream, us_graph,
               ms[ms.size() / 2] * 1e3);
        CHK(hipGraphExecDestroy(ge));
        CHK(hipGraphDestroy(g));
    }
    return 0;
}

// tools/ubench_power.hip -- what does a VALU instruction COST IN ENERGY on an MI355X?  (design input, not product)
//
// Why: the headline kernel runs the socket at 1340-1350 W of its 1400 W cap and the power manager answers every gain in VALU
// utilisation with a lower shader clock (profiles/r03_power_probe.txt: 2.03 GHz for the kernel, 2.20 GHz for a variant that issues
// 7 % fewer instructions per cycle, the same watts).  Under a power cap the bound is joules per sample, not cycles per sample.
// This tool runs one instruction STREAM on every SIMD of the chip for a few seconds and reads the socket's energy accumulator
// (rsmi_dev_energy_count_get) around it: watts, sustained clock (cycles from s_memtime / wall time), wave-instructions per
// second, and from the difference to the scalar-only stream the energy per wave64 instruction.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench_power tools/ubench_power.hip -lrocm_smi64
// usage: ubench_power [seconds per stream, default 2.5]
#include <hip/hip_runtime.h>
#include <rocm_smi/rocm_smi.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define R2(x) x x
#define R4(x) R2(R2(x))
#define R8(x) R4(R2(x))
#define R16(x) R4(R4(x))
#define R32(x) R8(R4(x))
// one "round" = 16 instructions over the 16 data registers (8 pairs for the packed ones); a block = 16 rounds = 256 instructions
#define RND(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#define RNDP(M) M(0) M(2) M(4) M(6) M(8) M(10) M(12) M(14) M(0) M(2) M(4) M(6) M(8) M(10) M(12) M(14)
// data registers v0..v15, constants: v16 = a (-1.0), v17 = b (lane-varying), v[18:19] = {a, a}, v[20:21] = {b, b'}; s[20:21] = {a, a}, s[22:23] = {b, b}
#define FMA(i) "v_fma_f32 v" #i ", v" #i ", v16, v17\n"
#define MUL(i) "v_mul_f32_e32 v" #i ", v16, v" #i "\n"
#define ADD(i) "v_add_f32_e32 v" #i ", v17, v" #i "\n"
#define MAX3(i) "v_max3_f32 v" #i ", v" #i ", v16, v17\n"
#define BFI(i) "v_bfi_b32 v" #i ", v16, v" #i ", v17\n"
#define XOR(i) "v_xor_b32_e32 v" #i ", v17, v" #i "\n"
#define PKFMA(i) "v_pk_fma_f32 v[" #i ":" #i "+1], v[" #i ":" #i "+1], v[18:19], v[20:21]\n"
#define PKMUL(i) "v_pk_mul_f32 v[" #i ":" #i "+1], v[" #i ":" #i "+1], v[18:19]\n"
#define PKADD(i) "v_pk_add_f32 v[" #i ":" #i "+1], v[" #i ":" #i "+1], v[20:21]\n"
#define PKFMA_S(i) "v_pk_fma_f32 v[" #i ":" #i "+1], v[" #i ":" #i "+1], s[20:21], v[20:21]\n"
#define PKMUL_S(i) "v_pk_mul_f32 v[" #i ":" #i "+1], v[" #i ":" #i "+1], s[20:21]\n"
#define PKADD_S(i) "v_pk_add_f32 v[" #i ":" #i "+1], v[" #i ":" #i "+1], s[22:23]\n"
#define FMA_S(i) "v_fma_f32 v" #i ", v" #i ", s20, v17\n"
#define SNOP(i) "s_nop 0\n"

enum Stream { S_SNOP, S_FMA, S_MULADD, S_MAX3, S_BFI, S_XOR, S_PKFMA, S_PKMULADD, S_PKFMA_S, S_PKMULADD_S, S_FMA_S, S_N };
static const char* stream_name[S_N] = {"s_nop only (waves resident, no VALU)", "v_fma_f32 (VGPR operands)", "v_mul_f32 + v_add_f32 (e32, alternating)",
                                       "v_max3_f32", "v_bfi_b32", "v_xor_b32_e32", "v_pk_fma_f32 (VGPR pairs)", "v_pk_mul_f32 + v_pk_add_f32 (VGPR pairs)",
                                       "v_pk_fma_f32 (constants in SGPR pairs)", "v_pk_mul_f32 + v_pk_add_f32 (constants in SGPR pairs)", "v_fma_f32 (one SGPR operand)"};
static const double stream_flops[S_N] = {0, 2, 1, 0, 0, 0, 4, 2, 4, 2, 2};  // per lane per instruction

struct Rec { uint64_t cycles, rt; };

template <int K>
__global__ __launch_bounds__(1024) void k_stream(Rec* rec, int reps, float a, float b, int waves) {
    extern __shared__ float lds[];
    const int w = threadIdx.x >> 6;
    float bl = b * (1.0f + 0.013f * (float)(threadIdx.x & 63)) + 0.001f * (float)blockIdx.x;  // lane-varying addend: every lane toggles differently
    asm volatile(
        "v_mov_b32 v16, %0\n v_mov_b32 v17, %1\n v_mov_b32 v18, %0\n v_mov_b32 v19, %0\n v_mov_b32 v20, %1\n v_mul_f32 v21, 0.75, %1\n"
        "v_mul_f32 v0, 0.11, %1\n v_mul_f32 v1, 0.23, %1\n v_mul_f32 v2, 0.31, %1\n v_mul_f32 v3, 0.43, %1\n v_mul_f32 v4, 0.57, %1\n v_mul_f32 v5, 0.61, %1\n"
        "v_mul_f32 v6, 0.73, %1\n v_mul_f32 v7, 0.87, %1\n v_mul_f32 v8, 0.93, %1\n v_mul_f32 v9, 1.07, %1\n v_mul_f32 v10, 1.13, %1\n v_mul_f32 v11, 1.29, %1\n"
        "v_mul_f32 v12, 1.37, %1\n v_mul_f32 v13, 1.41, %1\n v_mul_f32 v14, 1.53, %1\n v_mul_f32 v15, 1.67, %1\n"
        "v_readfirstlane_b32 s20, v16\n s_mov_b32 s21, s20\n v_readfirstlane_b32 s22, v17\n s_mov_b32 s23, s22\n"
        :: "v"(a), "v"(bl)
        : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "s20", "s21", "s22", "s23");
    __syncthreads();
    // lane masks: 1 = the low 32 lanes execute the stream, 2 = the low 16 (what does an instruction cost when half / three quarters of its lanes are off?)
    if (waves == 1) asm volatile("s_mov_b64 exec, 0xffffffff");
    if (waves == 2) asm volatile("s_mov_b64 exec, 0xffff");
    uint64_t t0, t1, r0, r1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
#define BODY(X) asm volatile(".p2align 3\n" X ::: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15")
        if constexpr (K == S_SNOP) BODY(R16(RND(SNOP)));
        if constexpr (K == S_FMA) BODY(R16(RND(FMA)));
        if constexpr (K == S_MULADD) BODY(R8(RND(MUL) RND(ADD)));
        if constexpr (K == S_MAX3) BODY(R16(RND(MAX3)));
        if constexpr (K == S_BFI) BODY(R16(RND(BFI)));
        if constexpr (K == S_XOR) BODY(R16(RND(XOR)));
        if constexpr (K == S_PKFMA) BODY(R16(RNDP(PKFMA)));
        if constexpr (K == S_PKMULADD) BODY(R8(RNDP(PKMUL) RNDP(PKADD)));
        if constexpr (K == S_PKFMA_S) BODY(R16(RNDP(PKFMA_S)));
        if constexpr (K == S_PKMULADD_S) BODY(R8(RNDP(PKMUL_S) RNDP(PKADD_S)));
        if constexpr (K == S_FMA_S) BODY(R16(RND(FMA_S)));
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
    asm volatile("s_mov_b64 exec, -1");
    float acc;
    asm volatile("v_add_f32 %0, v0, v1\n v_add_f32 %0, %0, v2\n v_add_f32 %0, %0, v3\n v_add_f32 %0, %0, v15" : "=v"(acc));
    if (acc == 123.456f) lds[threadIdx.x] = acc;  // keep the registers alive
    if ((threadIdx.x & 63) == 0) { rec[blockIdx.x * 16 + w].cycles = t1 - t0; rec[blockIdx.x * 16 + w].rt = r1 - r0; }
}

// HBM stream: plain 16-byte stores over a large buffer (what the renderer's output costs)
__global__ __launch_bounds__(256) void k_fill(float4* p, size_t n, float x) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = float4{x, x + 1.0f, x + 2.0f, x + (float)i};
}

static uint64_t energy_uj(double* res_out = nullptr) {
    uint64_t c = 0, ts = 0;
    float res = 0;
    if (rsmi_dev_energy_count_get(0, &c, &res, &ts) != RSMI_STATUS_SUCCESS) return 0;
    if (res_out) *res_out = res;
    return (uint64_t)((double)c * res);
}
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double socket_w() { uint64_t p = 0; rsmi_dev_current_socket_power_get(0, &p); return (double)p * 1e-6; }
static double sclk_mhz() { rsmi_frequencies_t f{}; rsmi_dev_gpu_clk_freq_get(0, RSMI_CLK_TYPE_SYS, &f); return f.current < 32 ? (double)f.frequency[f.current] * 1e-6 : 0; }

struct Sampler {  // socket power / sclk at ~10 Hz from a side thread (the launch loop never waits for rocm_smi)
    std::thread th; volatile bool stop = false; double wsum = 0, fsum = 0, fmin = 1e9; int n = 0;
    void start() { stop = false; wsum = fsum = 0; fmin = 1e9; n = 0; th = std::thread([this] { while (!stop) { double w = socket_w(), f = sclk_mhz(); wsum += w; fsum += f; if (f < fmin) fmin = f; n++; std::this_thread::sleep_for(std::chrono::milliseconds(100)); } }); }
    void end() { stop = true; th.join(); }
};

template <int K>
double run_stream(Rec* d_rec, int waves_per_simd, double seconds, double base_w, int lane_mask = 0) {
    const int threads = 256 * waves_per_simd, grid = 256, reps = 4000;  // 256 instructions x 4000 = 1.02 M instructions per wave per launch
    const size_t lds = 100 * 1024;                                      // one workgroup per CU
    hipFuncSetAttribute((const void*)k_stream<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1, ev[4];
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto& e : ev) hipEventCreate(&e);
    auto launch = [&]() { k_stream<K><<<grid, threads, lds>>>(d_rec, reps, -1.0f, 0.7321f, lane_mask); };
    long launches = 0;
    auto pump = [&](double secs) {  // keep four launches queued at all times
        const double T = now_s();
        int slot = 0;
        while (now_s() - T < secs) { hipEventSynchronize(ev[slot]); launch(); launches++; hipEventRecord(ev[slot]); slot = (slot + 1) & 3; }
    };
    for (auto& e : ev) hipEventRecord(e);
    pump(0.7);  // warm-up: clock and power settle
    Sampler sm; sm.start();
    const uint64_t E0 = energy_uj();
    const double T0 = now_s();
    hipEventRecord(e0);
    launches = 0;
    pump(seconds);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    const double T1 = now_s();
    const uint64_t E1 = energy_uj();
    sm.end();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<Rec> h(grid * 16);
    hipMemcpy(h.data(), d_rec, sizeof(Rec) * grid * 16, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0, rtmax = 0; int nw = 0;
    for (int b = 0; b < grid; b++) for (int w = 0; w < 4 * waves_per_simd; w++) { const Rec& r = h[b * 16 + w]; cyc += (double)r.cycles; rt += (double)r.rt; if ((double)r.rt > rtmax) rtmax = (double)r.rt; nw++; }
    const double insts_per_wave = 256.0 * reps;
    const double waves = (double)grid * 4 * waves_per_simd;
    const double rate = insts_per_wave * waves * launches / (ms * 1e-3);  // wave-instructions per second, whole chip, launch gaps included
    const double watts = (double)(E1 - E0) * 1e-6 / (T1 - T0);
    const double ghz = cyc / (rt * 10.0);                                 // shader cycles per 10 ns tick of s_memrealtime, inside the loop
    const double busy = rtmax * 10e-9 * launches / (ms * 1e-3);           // fraction of the wall time a kernel was running
    if (lane_mask) printf("[%d of 64 lanes] ", lane_mask == 1 ? 32 : 16);
    printf("%-56s %d/SIMD | %7.1f W | clock %5.3f GHz (smi avg %4.0f min %4.0f) | %5.2f cyc/inst per wave, %5.2f per SIMD | %5.2f T wave-inst/s | busy %4.2f", stream_name[K],
           waves_per_simd, watts, ghz, sm.fsum / sm.n, sm.fmin, cyc / nw / insts_per_wave, cyc / nw / insts_per_wave / waves_per_simd, rate * 1e-12, busy);
    if (base_w > 0 && K != S_SNOP) printf(" | %5.2f nJ/wave-inst | %5.1f TF", (watts - base_w) / rate * 1e9, rate * 64 * stream_flops[K] * 1e-12);
    printf("\n");
    fflush(stdout);
    return watts;
}

template <int NT>
__global__ __launch_bounds__(256) void k_fill2(float* p, size_t n, float x) {  // 4-byte stores, one 256-byte run per wave instruction (the renderer's store shape)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(x + (float)i, &p[i]); else p[i] = x + (float)i;
    }
}
template <class F>
void run_fill(const char* name, F launch, size_t bytes, double seconds, double idle_w) {
    double t = now_s();
    while (now_s() - t < 0.7) { launch(0.0f); hipDeviceSynchronize(); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    Sampler sm; sm.start();
    const uint64_t E0 = energy_uj(); const double T0 = now_s();
    hipEventRecord(e0);
    long launches = 0;
    while (now_s() - T0 < seconds) { for (int k = 0; k < 4; k++) launch((float)launches); launches += 4; hipDeviceSynchronize(); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    const double T1 = now_s(); const uint64_t E1 = energy_uj();
    sm.end();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double watts = (double)(E1 - E0) * 1e-6 / (T1 - T0), gbs = (double)bytes * launches / (ms * 1e-3) * 1e-9;
    printf("%-56s        | %7.1f W | smi clock avg %4.0f min %4.0f | %7.1f GB/s written | %5.1f pJ per byte over the idle socket\n", name, watts, sm.fsum / sm.n, sm.fmin, gbs, (watts - idle_w) / (gbs * 1e9) * 1e12);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 2.5;
    if (rsmi_init(0) != RSMI_STATUS_SUCCESS) { printf("rsmi_init failed\n"); return 2; }
    double idle_w = 0;
    uint64_t cap = 0;
    rsmi_dev_power_cap_get(0, 0, &cap);
    double res = 0;
    energy_uj(&res);
    printf("# power cap %.0f W; energy counter resolution %.3f uJ; %g s per stream\n", (double)cap * 1e-6, res, seconds);
    Rec* d_rec;
    hipMalloc((void**)&d_rec, sizeof(Rec) * 256 * 16);
    {   // idle socket
        std::this_thread::sleep_for(std::chrono::milliseconds(500));
        const uint64_t E0 = energy_uj(); const double T0 = now_s();
        std::this_thread::sleep_for(std::chrono::milliseconds(1500));
        const uint64_t E1 = energy_uj(); const double T1 = now_s();
        idle_w = (double)(E1 - E0) * 1e-6 / (T1 - T0);
        printf("%-58s         | %7.1f W | sclk smi %6.0f MHz\n", "idle (no kernel)", idle_w, sclk_mhz());
    }
    if (argc > 2 && atoi(argv[2]) == 1) {  // lane-mask mode: the same streams with 64 / 32 / 16 active lanes, two waves per SIMD
        const double b2 = run_stream<S_SNOP>(d_rec, 2, seconds, 0);
        for (int m : {0, 1, 2}) {
            run_stream<S_FMA>(d_rec, 2, seconds, b2, m);
            run_stream<S_MULADD>(d_rec, 2, seconds, b2, m);
            run_stream<S_PKFMA_S>(d_rec, 2, seconds, b2, m);
            run_stream<S_PKMULADD>(d_rec, 2, seconds, b2, m);
            run_stream<S_PKMULADD_S>(d_rec, 2, seconds, b2, m);
        }
        return 0;
    }
    double base[5] = {0, 0, 0, 0, 0};
    for (int n : {1, 2, 4}) {
        base[n] = run_stream<S_SNOP>(d_rec, n, seconds, 0);  // scalar-only stream: its watts are the base of the per-instruction energies at this occupancy
        run_stream<S_FMA>(d_rec, n, seconds, base[n]);
        run_stream<S_MULADD>(d_rec, n, seconds, base[n]);
        run_stream<S_FMA_S>(d_rec, n, seconds, base[n]);
        run_stream<S_MAX3>(d_rec, n, seconds, base[n]);
        run_stream<S_BFI>(d_rec, n, seconds, base[n]);
        run_stream<S_XOR>(d_rec, n, seconds, base[n]);
        run_stream<S_PKFMA>(d_rec, n, seconds, base[n]);
        run_stream<S_PKMULADD>(d_rec, n, seconds, base[n]);
        run_stream<S_PKFMA_S>(d_rec, n, seconds, base[n]);
        run_stream<S_PKMULADD_S>(d_rec, n, seconds, base[n]);
    }
    {   // HBM write streams
        const size_t bytes = (size_t)12 << 30;
        float4* buf;
        if (hipMalloc((void**)&buf, bytes) == hipSuccess) {
            run_fill("HBM fill, 16-byte stores", [&](float x) { k_fill<<<256 * 16, 256>>>(buf, bytes / 16, x); }, bytes, seconds, idle_w);
            run_fill("HBM fill, 4-byte stores (256-byte run per wave)", [&](float x) { k_fill2<0><<<256 * 16, 256>>>((float*)buf, bytes / 4, x); }, bytes, seconds, idle_w);
            run_fill("HBM fill, 4-byte nontemporal stores", [&](float x) { k_fill2<1><<<256 * 16, 256>>>((float*)buf, bytes / 4, x); }, bytes, seconds, idle_w);
        }
    }
    return 0;
}

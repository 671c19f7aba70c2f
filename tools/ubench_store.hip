// tools/ubench_store.hip -- what a sample store costs the SIMD it is issued on, gfx950 (design input, not product).
//
// profiles/r03_ubench_issue_v3.txt has ONE store line: "15 fma + 1 buffer_store_dword", two waves per SIMD: 7.96 cycles per
// instruction per wave against 4.08 for pure fma pairs -- i.e. every dword store of 64 lanes takes ~31 cycles of the SIMD away from
// BOTH waves (47 with the whole chip storing).  The voice kernels store one dword per voice-frame and channel; this tool asks
// what the alternatives cost, same harness as ubench_issue (per-wave s_memtime around straight-line blocks, waves w and w + 4 of
// a workgroup share a SIMD, role A = waves with bit 2 clear, role B = the others, placement verified from HW_ID):
//   * the same 1 KiB per wave as 4 x buffer_store_dword / 2 x dwordx2 / 1 x dwordx4 (rows of 256 B / 512 B / 1 KiB per wave);
//   * the store stream next to a PURE FMA wave: what the partner loses per store;
//   * LDS traffic of a store-through-LDS scheme: ds_write_b32 per frame, ds_read_b128 per four frames.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench_store tools/ubench_store.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

#define R2(x) x x
#define R4(x) R2(R2(x))
#define R8(x) R4(R2(x))
#define R16(x) R4(R4(x))
#define R32(x) R8(R4(x))
#define R64(x) R8(R8(x))
#define FMA(i) "v_fma_f32 %" #i ", %" #i ", %[ka], %[kb]\n"
#define F8 FMA(0) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7)
#define F15 F8 FMA(0) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6)
#define F31 F8 F8 F8 FMA(0) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6)
#define F63 F8 F8 F8 F8 F8 F8 F8 FMA(0) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6)

enum Kind { FMA_ONLY, ST1, ST2, ST4, ST1_GLOBAL, ST4_GLOBAL, DSW1, DSR4, DSW1_DSR4, IDLE, NKINDS };
static const char* kind_name[NKINDS] = {"64 fma", "4 x (15 fma + buffer_store_dword)", "2 x (31 fma + buffer_store_dwordx2)", "63 fma + buffer_store_dwordx4",
                                        "4 x (15 fma + global_store_dword)", "63 fma + global_store_dwordx4", "4 x (15 fma + ds_write_b32)",
                                        "63 fma + ds_read_b128", "4 x (15 fma + ds_write_b32), then 63 fma + ds_read_b128", "idle"};
static int kind_insts(int k) { return k == IDLE ? 0 : k == DSW1_DSR4 ? 128 * 8 : 64 * 8; }

typedef int v4i_ __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f_ __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ uint64_t memtime() { return __builtin_readcyclecounter(); }
static __device__ __forceinline__ uint64_t memrealtime() {
    uint64_t t;
    asm volatile("s_memrealtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}

template <int K>
__device__ __forceinline__ void block(float* x, float a, float b, int vo1, int vo2, int vo4, int ldsw, int ldsr, v4i_ rsrc, float* gp1, v4f* gp4, v4f& q) {
#define XOPS "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
    if constexpr (K == FMA_ONLY) asm volatile(R8(F8 F8 F8 F8 F8 F8 F8 F8) : XOPS : [ka] "v"(a), [kb] "v"(b));
    if constexpr (K == ST1)   // per 64 instructions: 4 rows of 256 B (lane * 4), scalar offsets 0 / 256 / 512 / 768 in the instruction
        asm volatile(R8(F15 "buffer_store_dword %0, %[vo], %[rs], 0 offen\n" F15 "buffer_store_dword %1, %[vo], %[rs], 0 offen offset:256\n"
                        F15 "buffer_store_dword %2, %[vo], %[rs], 0 offen offset:512\n" F15 "buffer_store_dword %3, %[vo], %[rs], 0 offen offset:768\n")
                     : XOPS : [ka] "v"(a), [kb] "v"(b), [vo] "v"(vo1), [rs] "s"(rsrc) : "memory");
    if constexpr (K == ST2) {
        const v2f_ q2 = {q.x, q.y};
        asm volatile(R8(F31 "buffer_store_dwordx2 %[d], %[vo], %[rs], 0 offen\n" F31 "buffer_store_dwordx2 %[d], %[vo], %[rs], 0 offen offset:512\n")
                     : XOPS : [ka] "v"(a), [kb] "v"(b), [vo] "v"(vo2), [rs] "s"(rsrc), [d] "v"(q2) : "memory");
    }
    if constexpr (K == ST4)
        asm volatile(R8(F63 "buffer_store_dwordx4 %[d], %[vo], %[rs], 0 offen\n") : XOPS : [ka] "v"(a), [kb] "v"(b), [vo] "v"(vo4), [rs] "s"(rsrc), [d] "v"(q) : "memory");
    if constexpr (K == ST1_GLOBAL)
        asm volatile(R8(F15 "global_store_dword %[p], %0, off\n" F15 "global_store_dword %[p], %1, off offset:256\n" F15 "global_store_dword %[p], %2, off offset:512\n"
                        F15 "global_store_dword %[p], %3, off offset:768\n")
                     : XOPS : [ka] "v"(a), [kb] "v"(b), [p] "v"(gp1) : "memory");
    if constexpr (K == ST4_GLOBAL) asm volatile(R8(F63 "global_store_dwordx4 %[p], %[d], off\n") : XOPS : [ka] "v"(a), [kb] "v"(b), [p] "v"(gp4), [d] "v"(q) : "memory");
    if constexpr (K == DSW1)
        asm volatile(R8(F15 "ds_write_b32 %[ad], %0\n" F15 "ds_write_b32 %[ad], %1 offset:272\n" F15 "ds_write_b32 %[ad], %2 offset:544\n" F15 "ds_write_b32 %[ad], %3 offset:816\n")
                     : XOPS : [ka] "v"(a), [kb] "v"(b), [ad] "v"(ldsw) : "memory");
    if constexpr (K == DSR4) asm volatile(R8(F63 "ds_read_b128 %[d], %[ad]\n") "s_waitcnt lgkmcnt(0)\n" : XOPS, [d] "=&v"(q) : [ka] "v"(a), [kb] "v"(b), [ad] "v"(ldsr) : "memory");
    if constexpr (K == DSW1_DSR4)
        asm volatile(R8(F15 "ds_write_b32 %[aw], %0\n" F15 "ds_write_b32 %[aw], %1 offset:272\n" F15 "ds_write_b32 %[aw], %2 offset:544\n" F15 "ds_write_b32 %[aw], %3 offset:816\n"
                        F63 "ds_read_b128 %[d], %[ar]\n")
                     "s_waitcnt lgkmcnt(0)\n"
                     : XOPS, [d] "=&v"(q) : [ka] "v"(a), [kb] "v"(b), [aw] "v"(ldsw), [ar] "v"(ldsr) : "memory");
    if constexpr (K == IDLE) __builtin_amdgcn_s_sleep(127);
}

struct Rec {
    uint64_t cycles, real;
    uint32_t hw, xcc;
    float sink;
    uint32_t pad;
};

template <int KA, int KB>
__global__ __launch_bounds__(1024) void k(Rec* rec, float* rows, int reps, float a, float b) {
    extern __shared__ float lds[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = 0.25f + 0.001f * (float)lane + 0.01f * i;
    v4f q = {x[0], x[1], x[2], x[3]};
    // a scratch region of 4 KiB per wave, rewritten over and over: one dword row = 256 B, a dwordx4 row = 1 KiB
    float* wrow = rows + ((size_t)blockIdx.x * 16 + w) * 1024;
    const uint64_t rowp = reinterpret_cast<uint64_t>(wrow);
    const v4i_ rsrc = {__builtin_amdgcn_readfirstlane((int)(uint32_t)rowp), __builtin_amdgcn_readfirstlane((int)(uint32_t)(rowp >> 32) & 0xffff), 4096, 0x00020000};
    // LDS: a tile of [16 frames][68 floats] per wave (the mix tile's shape): lane writes its column, reads 16 B runs of a row
    const int ldsw = (w * 16 * 68 + lane) * 4, ldsr = (w * 16 * 68 + (lane >> 2) * 68 + (lane & 3) * 16) * 4;
    lds[(w * 16 * 68 + lane) % (15 * 1024 / 4)] = x[0];
    __syncthreads();
    const uint64_t t0 = memtime(), r0 = memrealtime();
    if (((w >> 2) & 1) == 0) {
#pragma unroll 1
        for (int it = 0; it < reps; it++) block<KA>(x, a, b, lane * 4, lane * 8, lane * 16, ldsw, ldsr, rsrc, wrow + lane, reinterpret_cast<v4f*>(wrow) + lane, q);
    } else {
#pragma unroll 1
        for (int it = 0; it < reps; it++) block<KB>(x, a, b, lane * 4, lane * 8, lane * 16, ldsw, ldsr, rsrc, wrow + lane, reinterpret_cast<v4f*>(wrow) + lane, q);
    }
    const uint64_t t1 = memtime(), r1 = memrealtime();
    float s = q.x + q.y;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i];
    if (lane == 0) {
        Rec& o = rec[blockIdx.x * (blockDim.x >> 6) + w];
        o.cycles = t1 - t0;
        o.real = r1 - r0;
        o.hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
        o.xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) & 0xf;
        o.sink = s;
    }
}

static Rec* d_rec;
static float* d_rows;

template <int KA, int KB>
void run(int nwaves_per_simd, int grid, const char* note = "") {
    const int wpb = 4 * nwaves_per_simd, reps = 64;
    const size_t lds = grid > 1 ? 100 * 1024 : 64 * 1024;
    hipFuncSetAttribute((const void*)k<KA, KB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((k<KA, KB>), dim3(grid), dim3(64 * wpb), lds, 0, d_rec, d_rows, reps, 0.5f, 0.001953125f);
    hipDeviceSynchronize();
    std::vector<Rec> h((size_t)grid * wpb);
    hipMemcpy(h.data(), d_rec, h.size() * sizeof(Rec), hipMemcpyDeviceToHost);
    std::map<uint64_t, int> per_simd;
    double cyc[2] = {0, 0}, mhz = 0;
    int n[2] = {0, 0};
    for (int bI = 0; bI < grid; bI++)
        for (int w = 0; w < wpb; w++) {
            const Rec& q = h[(size_t)bI * wpb + w];
            const uint64_t simd = (q.hw >> 4) & 3, cu = (q.hw >> 8) & 0xf, sh = (q.hw >> 12) & 1, se = (q.hw >> 13) & 7;
            per_simd[((uint64_t)q.xcc << 20) | (se << 12) | (sh << 8) | (cu << 4) | simd]++;
            const int role = (w >> 2) & 1;
            cyc[role] += (double)q.cycles;
            n[role]++;
            mhz += (double)q.cycles / ((double)q.real / 100.0);
        }
    mhz /= (double)(grid * wpb);
    bool placed = (int)per_simd.size() == grid * 4;
    for (auto& kv : per_simd) placed = placed && kv.second == nwaves_per_simd;
    const int kinds[2] = {KA, KB};
    printf("%-4s n/SIMD=%d grid=%-3d clock %4.0f MHz placement %s |", note, nwaves_per_simd, grid, mhz, placed ? "ok " : "BAD");
    for (int role = 0; role < 2; role++) {
        if (!n[role]) continue;
        const double per_trip = cyc[role] / n[role] / reps;
        const int insts = kind_insts(kinds[role]);
        printf(" [%s: %.0f cyc/trip", kind_name[kinds[role]], per_trip);
        if (insts > 0) printf(" = %.2f cyc/inst, %.1f cyc per 64-instruction group (1 KiB of samples per wave)", per_trip / insts, per_trip / insts * 64);
        printf("]");
    }
    printf("\n");
    fflush(stdout);
}

template <int K>
void sweep() {
    run<K, K>(1, 1);
    run<K, K>(2, 1);
    run<K, FMA_ONLY>(2, 1);
    run<K, K>(2, 256, "chip");
    run<K, FMA_ONLY>(2, 256, "chip");
}

int main() {
    hipMalloc((void**)&d_rec, 1 << 20);
    hipMalloc((void**)&d_rows, (size_t)256 * 16 * 4096);
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL((k<FMA_ONLY, FMA_ONLY>), dim3(256), dim3(512), 64 * 1024, 0, d_rec, d_rows, 256, 0.5f, 0.001f);
    hipDeviceSynchronize();
    printf("# cycles = shader clock; a 64-instruction group carries 1 KiB of samples per wave in every store kind; role A | role B share each SIMD\n");
    sweep<FMA_ONLY>();
    sweep<ST1>();
    sweep<ST2>();
    sweep<ST4>();
    sweep<ST1_GLOBAL>();
    sweep<ST4_GLOBAL>();
    sweep<DSW1>();
    sweep<DSR4>();
    sweep<DSW1_DSR4>();
    return 0;
}

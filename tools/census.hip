// tools/census.hip -- where do the waves of a launch land? (design input for the bank launch geometry)
// Each wave records XCC id + HW_ID (SE / CU / SIMD) and then spins long enough that the whole grid is co-resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <cstdint>

__global__ void k(uint32_t* rec, int spin) {
    uint32_t hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
    uint32_t xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)); // HW_REG_XCC_ID
    float x = threadIdx.x;
    for (int i = 0; i < spin; i++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(1.0001f));
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
    if ((threadIdx.x & 63) == 0) { rec[2 * wave] = hw; rec[2 * wave + 1] = xcc; }
    if (x == 12345.f) rec[0] = 0;
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 1 << 20);
    for (int cfg = 0; cfg < 4; cfg++) {
        int block = cfg == 0 ? 64 : cfg == 1 ? 256 : cfg == 2 ? 64 : 128;
        int waves = cfg == 2 ? 2048 : 1024;
        int grid = waves * 64 / block;
        hipMemset(d, 0, 1 << 20);
        hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, d, 200000);
        hipDeviceSynchronize();
        std::vector<uint32_t> h(2 * waves);
        hipMemcpy(h.data(), d, 2 * waves * 4, hipMemcpyDeviceToHost);
        std::map<uint32_t, int> per_simd, per_cu;
        for (int w = 0; w < waves; w++) {
            uint32_t hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
            uint32_t simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            uint32_t cukey = (xcc << 16) | (se << 8) | (sh << 4) | cu;
            per_cu[cukey]++;
            per_simd[(cukey << 2) | simd]++;
        }
        std::map<int, int> hist_simd, hist_cu;
        for (auto& kv : per_simd) hist_simd[kv.second]++;
        for (auto& kv : per_cu) hist_cu[kv.second]++;
        printf("block=%d grid=%d waves=%d: distinct CUs=%zu distinct SIMDs=%zu | waves-per-SIMD histogram:", block, grid, waves,
               per_cu.size(), per_simd.size());
        for (auto& kv : hist_simd) printf(" %dx%d", kv.first, kv.second);
        printf(" | waves-per-CU histogram:");
        for (auto& kv : hist_cu) printf(" %dx%d", kv.first, kv.second);
        printf("\n");
        if (cfg == 0) { printf("sample hw_id words:"); for (int w = 0; w < 8; w++) printf(" %08x/%x", h[2*w], h[2*w+1]); printf("\n"); }
    }
    return 0;
}

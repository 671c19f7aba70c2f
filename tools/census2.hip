// tools/census2.hip -- which SIMD does wave w of a workgroup land on when two workgroups of NW waves share a CU?
// (design input for the role order of the time-split kernel).  Each wave records HW_ID and spins until the grid is resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <cstdint>

template <int LDS_KB>
__global__ void k(uint32_t* rec, int spin) {
    __shared__ float pad[LDS_KB * 256];
    uint32_t hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
    uint32_t xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    float x = threadIdx.x;
    pad[threadIdx.x] = x;
    for (int i = 0; i < spin; i++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(1.0001f));
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
    if ((threadIdx.x & 63) == 0) { rec[2 * wave] = hw; rec[2 * wave + 1] = xcc; }
    if (x == 12345.f) rec[0] = (uint32_t)pad[5];
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 1 << 22);
    for (int nw : {4, 5, 6, 7, 8}) {
        const int grid = 512, waves = grid * nw;
        hipMemset(d, 0, 1 << 22);
        hipLaunchKernelGGL(k<64>, dim3(grid), dim3(64 * nw), 0, 0, d, 300000);
        hipDeviceSynchronize();
        std::vector<uint32_t> h(2 * waves);
        hipMemcpy(h.data(), d, 2 * waves * 4, hipMemcpyDeviceToHost);
        std::map<uint32_t, std::vector<std::pair<int, int>>> cu_waves;  // cu -> (global wave, simd)
        std::map<int, int> hist;                                         // waves per SIMD
        std::map<uint32_t, int> per_simd;
        std::map<std::vector<int>, int> patterns;                        // simd sequence of one workgroup -> count
        for (int b = 0; b < grid; b++) {
            std::vector<int> pat;
            for (int w = 0; w < nw; w++) {
                uint32_t hw = h[2 * (b * nw + w)], xcc = h[2 * (b * nw + w) + 1] & 0xf;
                uint32_t simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
                uint32_t cukey = (xcc << 16) | (se << 8) | (sh << 4) | cu;
                per_simd[(cukey << 2) | simd]++;
                cu_waves[cukey].push_back({b * nw + w, (int)simd});
                pat.push_back((int)simd);
            }
            patterns[pat]++;
        }
        for (auto& kv : per_simd) hist[kv.second]++;
        printf("NW=%d: %zu CUs; waves-per-SIMD histogram:", nw, cu_waves.size());
        for (auto& kv : hist) printf(" %dx%d", kv.first, kv.second);
        printf(" | workgroup SIMD patterns:");
        for (auto& kv : patterns) { printf(" ["); for (int s : kv.first) printf("%d", s); printf("]x%d", kv.second); }
        printf("\n  first CU:");
        for (auto& p : cu_waves.begin()->second) printf(" b%dw%d->S%d", p.first / nw, p.first % nw, p.second);
        printf("\n");
    }
    return 0;
}

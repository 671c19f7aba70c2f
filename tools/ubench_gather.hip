// tools/ubench_gather.hip -- what the wavetable gathers of config 4 cost a CU by where the tables live, gfx950 (design input, not
// product).  Every lane is a voice of the config-4 population (f log-uniform in [55, 1760) Hz, or [160, 1760) for the "short
// tables" rows): it walks its table pair of the saw set's geometry (40 tables, 8192 ... 32 floats + 3 of padding, 160 KiB) with
// its own phase increment and reads the 16 bytes of four taps from each of the two tables per step -- WaveSynth's access pattern
// without its arithmetic.  Rows:
//   global      global_load_dwordx4 (address space 1), the product's form
//   flat        the same addresses through generic pointers (flat_load_dwordx4)
//   flat+lds    generic pointers, every table of <= 512 floats (4 150 floats, 16.6 KB) copied to LDS: lanes of one instruction go
//               to LDS or to memory by their address
//   ds          short-table voices only, explicit LDS reads
// and the checksum of everything read, which must not depend on the row (4-byte aligned 16-byte reads through every path).
// W waves per workgroup gather (2 = config 4's two oscillator waves per CU), one workgroup per CU on the whole chip.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench_gather tools/ubench_gather.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
constexpr int NT = 40, PAD = 3, LDS_FLOATS = 4608;

struct Set {
    int off[NT], len[NT];
    float pitch[NT];
    int lds_from;  // first float of the tables that have an LDS copy
    int total;
};

enum Mode { GLOBAL, FLAT, FLAT_LDS, DS, NMODES };
static const char* mode_name[NMODES] = {"global", "flat", "flat+lds", "ds"};

template <int MODE>
__global__ __launch_bounds__(512) void k(const float* __restrict__ data, Set s, float fmin, float octaves, int steps, uint64_t* cycles, double* sums) {
    __shared__ float lt[LDS_FLOATS];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < s.total - s.lds_from && i < LDS_FLOATS; i += blockDim.x) lt[i] = data[s.lds_from + i];
    __syncthreads();
    const uint32_t v = (blockIdx.x * (blockDim.x >> 6) + w) * 64 + lane;
    uint32_t h = v * 2654435761u + 12345u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
    const float f = fmin * exp2f(octaves * u);
    int t = 0;
    while (t + 1 < NT - 2 && s.pitch[t + 1] <= f) t++;
    const int o1 = s.off[t + 1], o2 = s.off[t + 2];
    const uint32_t m1 = (uint32_t)s.len[t + 1] - 1u, m2 = (uint32_t)s.len[t + 2] - 1u;
    const float* g1 = data + o1;
    const float* g2 = data + o2;
    const float *p1 = g1, *p2 = g2;                       // generic
    if (MODE == FLAT_LDS || MODE == DS) {
        if (o1 >= s.lds_from) p1 = lt + (o1 - s.lds_from);
        if (o2 >= s.lds_from) p2 = lt + (o2 - s.lds_from);
    }
    const float d = f * (1.0f / 48000.0f);
    float ph = (float)((h >> 3) & 1023) * (1.0f / 1024.0f);
    float acc = 0.0f;
    const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 4
    for (int it = 0; it < steps; it++) {
        ph += d;
        ph -= floorf(ph);
        const uint32_t i1 = (uint32_t)((float)(m1 + 1u) * ph) & m1, i2 = (uint32_t)((float)(m2 + 1u) * ph) & m2;
        f4u a, b;
        if (MODE == GLOBAL) {
            a = *(const __attribute__((address_space(1))) f4u*)(g1 + i1);
            b = *(const __attribute__((address_space(1))) f4u*)(g2 + i2);
        } else if (MODE == DS) {
            a = *(const __attribute__((address_space(3))) f4u*)(p1 + i1);
            b = *(const __attribute__((address_space(3))) f4u*)(p2 + i2);
        } else {
            a = *(const f4u*)(p1 + i1);
            b = *(const f4u*)(p2 + i2);
        }
        acc += (a.x - a.w) + (a.y - a.z) + (b.x - b.w) + (b.y - b.z);
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    double tot = (double)acc;
    for (int o = 32; o; o >>= 1) tot += __shfl_xor(tot, o);
    if (lane == 0) {
        cycles[blockIdx.x * (blockDim.x >> 6) + w] = t1 - t0;
        sums[blockIdx.x * (blockDim.x >> 6) + w] = tot;
    }
}

static Set make_set(std::vector<float>& data) {
    Set s{};
    int off = 0;
    for (int i = 0; i < NT; i++) {  // the saw set's lengths: 8192, 4 x 4096, 4 x 2048, ... 4 x 128, 3 x 64, 12 x 32
        const float p = 20.0f * std::pow(2.0f, i / 4.0f);
        int len = i == 0 ? 8192 : 4096 >> ((i - 1) / 4);
        if (len < 32) len = 32;
        if (i >= 25 && i <= 27) len = 64;
        if (i >= 28) len = 32;
        s.pitch[i] = p; s.len[i] = len; s.off[i] = off;
        off += len + PAD;
    }
    s.total = off;
    s.lds_from = s.total;
    for (int i = NT - 1; i >= 0 && s.len[i] <= 512; i--) s.lds_from = s.off[i];
    data.resize(off);
    for (int i = 0; i < off; i++) data[i] = (float)((i * 2654435761u >> 20) & 1023) - 511.0f;
    return s;
}

int main() {
    std::vector<float> h;
    Set s = make_set(h);
    printf("# set: %d floats, LDS copy from float %d (%d floats = %.1f KB)\n", s.total, s.lds_from, s.total - s.lds_from, (s.total - s.lds_from) * 4 / 1024.0);
    if (s.total - s.lds_from > LDS_FLOATS) { printf("LDS copy does not fit\n"); return 1; }
    float* d;
    uint64_t* dc;
    double* ds;
    hipMalloc((void**)&d, h.size() * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc((void**)&dc, 256 * 8 * 8);
    hipMalloc((void**)&ds, 256 * 8 * 8);
    const int steps = 20000;
    auto run = [&](auto kern, int mode, int waves, float fmin, float oct) {
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), 0, 0, d, s, fmin, oct, steps, dc, ds);
        hipDeviceSynchronize();
        std::vector<uint64_t> c(256 * waves);
        std::vector<double> q(256 * waves);
        hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(q.data(), ds, q.size() * 8, hipMemcpyDeviceToHost);
        double cyc = 0, sum = 0;
        for (size_t i = 0; i < c.size(); i++) { cyc += (double)c[i]; sum += q[i]; }
        cyc /= (double)c.size() * steps;
        printf("%-9s f in [%4.0f, 1760) %d waves per CU: %7.1f cycles per step (two 16-byte gathers of 64 lanes) per wave = %6.1f per CU and gather; checksum %.0f\n",
               mode_name[mode], fmin, waves, cyc, cyc / (2.0 * waves), sum);
        fflush(stdout);
    };
    for (int waves : {2, 4, 8}) {
        run(k<GLOBAL>, GLOBAL, waves, 55.0f, 5.0f);
        run(k<FLAT>, FLAT, waves, 55.0f, 5.0f);
        run(k<FLAT_LDS>, FLAT_LDS, waves, 55.0f, 5.0f);
        run(k<GLOBAL>, GLOBAL, waves, 160.0f, 3.459f);
        run(k<FLAT_LDS>, FLAT_LDS, waves, 160.0f, 3.459f);
        run(k<DS>, DS, waves, 160.0f, 3.459f);
    }
    return 0;
}

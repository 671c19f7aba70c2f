// tools/proto_wt_interp_cross_frame.hip -- design prototype (round 5): the shipping formulation (optimal4x44_2 over the two frames of a pair) in isolation: 76 VALU per frame pair, 17 of them v_mov.
#include <hip/hip_runtime.h>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ v2f optimal4x44_2(v2f a0, v2f a1, v2f a2, v2f a3, v2f x) {
    v2f z = x - (float)0.5;
    v2f even1 = a2 + a1, odd1 = a2 - a1, even2 = a3 + a0, odd2 = a3 - a0;
    v2f c0 = even1 * (float)0.4656725512077848 + even2 * (float)0.03432729708429672;
    v2f c1 = odd1 * (float)0.5374383075356016 + odd2 * (float)0.1542946255730746;
    v2f c2 = even1 * (float)-0.25194210134021744 + even2 * (float)0.2519474493593906;
    v2f c3 = odd1 * (float)-0.46896069955075126 + odd2 * (float)0.15578800670302476;
    v2f c4 = even1 * (float)0.00986988334359864 + even2 * (float)-0.00989340017126506;
    return (((c4 * z + c3) * z + c2) * z + c1) * z + c0;
}
__global__ void k(const float* __restrict__ tab, const unsigned* __restrict__ idx, const float* __restrict__ frac, float* __restrict__ out, float w) {
    const int i = threadIdx.x + blockIdx.x * blockDim.x;
    const unsigned ia0 = idx[i * 4 + 0], ia1 = idx[i * 4 + 1], ib0 = idx[i * 4 + 2], ib1 = idx[i * 4 + 3];
    const f4u qa0 = *(const __attribute__((address_space(1))) f4u*)(tab + ia0);
    const f4u qa1 = *(const __attribute__((address_space(1))) f4u*)(tab + ia1);
    const f4u qb0 = *(const __attribute__((address_space(1))) f4u*)(tab + ib0);
    const f4u qb1 = *(const __attribute__((address_space(1))) f4u*)(tab + ib1);
    const v2f ea = optimal4x44_2(v2f{qa0.x, qa1.x}, v2f{qa0.y, qa1.y}, v2f{qa0.z, qa1.z}, v2f{qa0.w, qa1.w}, v2f{frac[i * 4 + 0], frac[i * 4 + 1]});
    const v2f eb = optimal4x44_2(v2f{qb0.x, qb1.x}, v2f{qb0.y, qb1.y}, v2f{qb0.z, qb1.z}, v2f{qb0.w, qb1.w}, v2f{frac[i * 4 + 2], frac[i * 4 + 3]});
    const v2f o = ea * (1.0f - w) + eb * w;
    reinterpret_cast<v2f*>(out)[i] = o;
}

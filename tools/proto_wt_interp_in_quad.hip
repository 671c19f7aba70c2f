// tools/proto_wt_interp_in_quad.hip -- design prototype (round 5): the 4-point wavetable interpolation of a frame pair over two tables with the packing done INSIDE each tap quad
// ((a2 + a1, a3 + a0) and (a2 - a1, a3 - a0) are v_pk_add_f32 with a swapped op_sel; an asm barrier keeps the coefficient sums scalar).  hipcc -S: 30 v_pk_mul + 19 v_pk_add + 20 v_add + 5 v_mov
// = 74 VALU per frame pair -- against 30 + 29 + 17 v_mov = 76 for the shipping cross-frame packing (proto_wt_interp_cross_frame.hip): no gain, the packing is not where the moves are.
#include <hip/hip_runtime.h>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
// one (table, frame) quad -> the five polynomial coefficients, packed INSIDE the quad
struct C5 { float c0, c1, c2, c3, c4; };
__device__ __forceinline__ C5 coefs(f4u q) {
    const v2f lo = v2f{q.x, q.y}, hi = v2f{q.z, q.w};
    const v2f sw = v2f{lo.y, lo.x};
    const v2f E = hi + sw;   // (a2 + a1, a3 + a0) = (even1, even2)
    const v2f O = hi - sw;   // (a2 - a1, a3 - a0) = (odd1, odd2)
    const v2f m0 = E * v2f{(float)0.4656725512077848, (float)0.03432729708429672};
    const v2f m1 = O * v2f{(float)0.5374383075356016, (float)0.1542946255730746};
    const v2f m2 = E * v2f{(float)-0.25194210134021744, (float)0.2519474493593906};
    const v2f m3 = O * v2f{(float)-0.46896069955075126, (float)0.15578800670302476};
    const v2f m4 = E * v2f{(float)0.00986988334359864, (float)-0.00989340017126506};
    C5 r{m0.x + m0.y, m1.x + m1.y, m2.x + m2.y, m3.x + m3.y, m4.x + m4.y};
    asm volatile("" : "+v"(r.c0), "+v"(r.c1), "+v"(r.c2), "+v"(r.c3), "+v"(r.c4));   // keep the five sums scalar: no re-packing across quads
    return r;
}
__global__ void k(const float* __restrict__ tab, const unsigned* __restrict__ idx, const float* __restrict__ frac, float* __restrict__ out, float w) {
    const int i = threadIdx.x + blockIdx.x * blockDim.x;
    // two frames x two tables
    const unsigned ia0 = idx[i * 4 + 0], ia1 = idx[i * 4 + 1], ib0 = idx[i * 4 + 2], ib1 = idx[i * 4 + 3];
    const f4u qa0 = *(const __attribute__((address_space(1))) f4u*)(tab + ia0);
    const f4u qa1 = *(const __attribute__((address_space(1))) f4u*)(tab + ia1);
    const f4u qb0 = *(const __attribute__((address_space(1))) f4u*)(tab + ib0);
    const f4u qb1 = *(const __attribute__((address_space(1))) f4u*)(tab + ib1);
    const C5 a0 = coefs(qa0), a1 = coefs(qa1), b0 = coefs(qb0), b1 = coefs(qb1);
    const v2f za = v2f{frac[i * 4 + 0], frac[i * 4 + 1]} - 0.5f, zb = v2f{frac[i * 4 + 2], frac[i * 4 + 3]} - 0.5f;
    const v2f ea = (((v2f{a0.c4, a1.c4} * za + v2f{a0.c3, a1.c3}) * za + v2f{a0.c2, a1.c2}) * za + v2f{a0.c1, a1.c1}) * za + v2f{a0.c0, a1.c0};
    const v2f eb = (((v2f{b0.c4, b1.c4} * zb + v2f{b0.c3, b1.c3}) * zb + v2f{b0.c2, b1.c2}) * zb + v2f{b0.c1, b1.c1}) * zb + v2f{b0.c0, b1.c0};
    const v2f o = ea * (1.0f - w) + eb * w;
    reinterpret_cast<v2f*>(out)[i] = o;
}

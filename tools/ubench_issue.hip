// tools/ubench_issue.hip -- VALU issue micro-benchmark for gfx950, third take (design input, not product).
//
// What was wrong with ubench_valu / ubench_valu2 (Weak 3): they timed whole grids with hipEvents and
// ASSUMED where the waves landed.  Here
//   * every wave reads s_memtime (shader cycles) around a STRAIGHT-LINE block of 512 instructions (one asm statement,
//     repeated REPS times by a 3-instruction scalar loop) and reports cycles per instruction itself;
//   * every wave records HW_ID: the host checks that each (CU, SIMD) holds exactly the intended number of waves;
//   * one workgroup of 256 x n threads puts n waves on each SIMD of ONE CU (waves w and w + 4 share a SIMD,
//     profiles/r02_census_wave_placement.txt); a grid of 256 such workgroups with 100 KiB of LDS each gives the same
//     placement on every CU, i.e. the clock the whole chip sustains under that load (s_memrealtime, 100 MHz);
//   * two ROLES per launch: waves with bit 2 of their index clear run stream KA, the others KB -- the two waves that
//     share a SIMD in the engine's pipeline kernel run different streams, and so can this.
// The disassembly is checked on the build host (tools/ubench_issue_check.sh): no s_nop / s_waitcnt inside a block.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I fundsp_amd/csrc -o tools/ubench_issue tools/ubench_issue.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "fd_math.hpp"

using fd::v2f;

#define R2(x) x x
#define R4(x) R2(R2(x))
#define R8(x) R4(R2(x))
#define R16(x) R4(R4(x))
#define R32(x) R8(R4(x))
#define R64(x) R8(R8(x))
#define R128(x) R16(R8(x))
#define R256(x) R16(R16(x))
#define R512(x) R64(R8(x))

// plain ops on %0..%15 (floats), constants %16 %17; packed ops on %0..%7 (register pairs), constants %8 %9
#define FMA(i) "v_fma_f32 %" #i ", %" #i ", %16, %17\n"
#define MUL(i) "v_mul_f32_e32 %" #i ", %16, %" #i "\n"
#define ADD(i) "v_add_f32_e32 %" #i ", %17, %" #i "\n"
#define MAX3(i) "v_max3_f32 %" #i ", %" #i ", %16, %17\n"
#define BFI(i) "v_bfi_b32 %" #i ", %16, %" #i ", %17\n"
#define CND(i) "v_cndmask_b32_e32 %" #i ", %16, %" #i ", vcc\n"
#define PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %9\n"
// operand sources other than VGPRs: SGPR (%18 = an SGPR holding a), SGPR pair (%10), literal, inline constant, lane masks
#define FMA_S(i) "v_fma_f32 %" #i ", %" #i ", %18, %17\n"
#define MUL_S(i) "v_mul_f32_e32 %" #i ", %18, %" #i "\n"
#define MUL_LIT(i) "v_mul_f32_e32 %" #i ", 0x3f000001, %" #i "\n"
#define ADD_INL(i) "v_add_f32_e32 %" #i ", 0.5, %" #i "\n"
#define ADD_E64(i) "v_add_f32_e64 %" #i ", %17, %" #i "\n"
#define CND_E64(i) "v_cndmask_b32_e64 %" #i ", %16, %" #i ", %19\n"
#define CMPCND(i) "v_cmp_lt_f32_e32 vcc, %16, %" #i "\nv_cndmask_b32_e32 %" #i ", %16, %" #i ", vcc\n"
#define CMP(i) "v_cmp_lt_f32_e32 vcc, %16, %" #i "\n"
#define CMP_E64(i) "v_cmp_lt_f32_e64 %19, %16, %" #i "\n"
#define AND(i) "v_and_b32_e32 %" #i ", %16, %" #i "\n"
#define LSHL(i) "v_lshlrev_b32_e32 %" #i ", 1, %" #i "\n"
#define BFE(i) "v_bfe_i32 %" #i ", %" #i ", 0, 1\n"
#define BITOP3_S(i) "v_bitop3_b32 %" #i ", %" #i ", %16, %18 bitop3:0x78\n"
#define MAXABS(i) "v_max_f32_e64 %" #i ", |%" #i "|, |%16|\n"
#define PKMUL_S(i) "v_pk_mul_f32 %" #i ", %" #i ", %10 op_sel_hi:[1,0]\n"
#define PKFMA_S(i) "v_pk_fma_f32 %" #i ", %" #i ", %10, %9 op_sel_hi:[1,0,1]\n"
#define DSW(i) "ds_write_b64 %20, %" #i "\n"
#define FMAN(i) "v_fma_f32 %" #i ", %" #i ", %[ka], %[kb]\n"
// mixed block: packed on pairs %0..%3, plain on %4..%11, constants %12 %13 (pairs) and %14 %15
#define MPK(i) "v_pk_fma_f32 %" #i ", %" #i ", %12, %13\n"
#define MPL(i) "v_fma_f32 %" #i ", %" #i ", %14, %15\n"

enum Kind {
    FMA_D1, FMA_K2, FMA_K4, FMA_K8, FMA_K16, MUL_K8, ADD_D1, ADD_K8, MAX3_K8, BFI_K8, CND_K8,
    PKFMA_D1, PKFMA_K2, PKFMA_K4, PKFMA_K8, PKMUL_K8, PKADD_K8,
    MIX_1PK_1PL,   // pk, plain, pk, plain ... all independent (8 chains each way)
    MIX_1PK_2PL,   // pk, plain, plain ...
    FMA_S_K8, MUL_S_K8, MUL_LIT_K8, ADD_INL_K8, ADD_E64_K8, CND_E64_K8, CMPCND_K8, CMP_K8, CMP_E64_K8, AND_K8, LSHL_K8, BFE_K8,
    BITOP3_S_K8, MAXABS_K8, PKMUL_S_K8, PKFMA_S_K8,
    PK_DSW,        // 15 pk_fma + 1 ds_write_b64 (a hand-over store per frame pair)
    PKFMA_AL0, PKFMA_AL4,   // the pk_fma block starting at an address = 0 / 4 (mod 8): do 8-byte instructions that straddle cost more?
    FMA_AL0, FMA_AL4, MIX48_AL0, MIX48_AL4,  // same for v_fma_f32, and for a (4-byte, 8-byte) alternating stream
    FMA_LO32, FMA_LO16, PKFMA_LO32,  // EXEC = the low 32 / 16 lanes only: does a half-empty wave64 skip its second pass?
    DSR_WAIT,      // ds_read2st64_b64 + s_waitcnt lgkmcnt(0) + 62 dependent-free VALU: the exposed hand-over read of a stage
    BUFST16,       // 15 v_fma + 1 buffer_store_dword
    IDLE,          // the wave does nothing (s_sleep) -- the partner runs alone on the SIMD
    SINE4,         // C++: four independent wide_sin2 evaluations per trip (the oscillator stage's feed-forward work)
    SVF8,          // C++: eight frames of the lowpass SVF recurrence per trip (the filter's serial work)
    SINE4_SVF8,    // C++: both in one wave, as the engine's stage 1 has them
    NKINDS
};
static const char* kind_name[NKINDS] = {
    "fma dep-1", "fma 2 chains", "fma 4 chains", "fma 8 chains", "fma 16 chains", "mul_e32 8 chains", "add_e32 dep-1",
    "add_e32 8 chains", "max3 8 chains", "bfi 8 chains", "cndmask 8 chains", "pk_fma dep-1", "pk_fma 2 chains",
    "pk_fma 4 chains", "pk_fma 8 chains", "pk_mul 8 chains", "pk_add 8 chains", "pk,plain alternating", "pk,plain,plain",
    "fma, SGPR operand", "mul_e32, SGPR operand", "mul_e32, literal", "add_e32, inline const", "add_e64 (VOP3, 2 src)", "cndmask_e64 (SGPR-pair mask)",
    "cmp+cndmask (vcc)", "cmp_e32 -> vcc", "cmp_e64 -> SGPR pair", "and_e32", "lshlrev_e32", "bfe_i32", "bitop3, SGPR operand", "max_f32_e64 |a|,|b|",
    "pk_mul, SGPR-pair operand", "pk_fma, SGPR-pair operand", "15 pk_fma + 1 ds_write_b64",
    "pk_fma, block at 0 mod 8", "pk_fma, block at 4 mod 8", "fma, block at 0 mod 8", "fma, block at 4 mod 8", "mul_e32,fma alternating at 0 mod 8", "mul_e32,fma alternating at 4 mod 8",
    "fma, EXEC = low 32 lanes", "fma, EXEC = low 16 lanes", "pk_fma, EXEC = low 32 lanes", "ds_read2st64_b64 + wait + 62 fma", "15 fma + 1 buffer_store_dword",
    "idle", "C++ 4 x wide_sin2", "C++ 8 x lowpass SVF frame", "C++ 4 x wide_sin2 + 8 x SVF"};
// instructions per block of the asm kinds; the C++ kinds are counted from the disassembly (tools/ubench_issue_check.sh
// prints the loop's size) and given here as VALU instructions per trip
static int kind_insts(int k) { return k == IDLE ? 0 : k >= SINE4 ? -1 : 512; }
#define SOPS , "s"(a), "s"(mask64), "v"(ldsaddr)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i_ __attribute__((ext_vector_type(4)));
struct Regs {
    float x[16];
    v2f y[8];
    v4f q;
};

template <int K>
__device__ __forceinline__ void block(Regs& r, float a, float b, v2f a2, v2f b2, float& guard, uint64_t mask64, uint64_t a2s, int ldsaddr, v4i_ rsrc) {
    float* x = r.x;
    v2f* y = r.y;
#define XOPS "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15])
#define YOPS "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7])
    if constexpr (K == FMA_D1) asm volatile(R512(FMA(0)) : XOPS : "v"(a), "v"(b));
    if constexpr (K == FMA_K2) asm volatile(R256(FMA(0) FMA(1)) : XOPS : "v"(a), "v"(b));
    if constexpr (K == FMA_K4) asm volatile(R128(FMA(0) FMA(1) FMA(2) FMA(3)) : XOPS : "v"(a), "v"(b));
    if constexpr (K == FMA_K8) asm volatile(R64(FMA(0) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7)) : XOPS : "v"(a), "v"(b));
    if constexpr (K == FMA_K16)
        asm volatile(R32(FMA(0) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7) FMA(8) FMA(9) FMA(10) FMA(11) FMA(12) FMA(13) FMA(14) FMA(15)) : XOPS : "v"(a), "v"(b));
    if constexpr (K == MUL_K8) asm volatile(R64(MUL(0) MUL(1) MUL(2) MUL(3) MUL(4) MUL(5) MUL(6) MUL(7)) : XOPS : "v"(a), "v"(b));
    if constexpr (K == ADD_D1) asm volatile(R512(ADD(0)) : XOPS : "v"(a), "v"(b));
    if constexpr (K == ADD_K8) asm volatile(R64(ADD(0) ADD(1) ADD(2) ADD(3) ADD(4) ADD(5) ADD(6) ADD(7)) : XOPS : "v"(a), "v"(b));
    if constexpr (K == MAX3_K8) asm volatile(R64(MAX3(0) MAX3(1) MAX3(2) MAX3(3) MAX3(4) MAX3(5) MAX3(6) MAX3(7)) : XOPS : "v"(a), "v"(b));
    if constexpr (K == BFI_K8) asm volatile(R64(BFI(0) BFI(1) BFI(2) BFI(3) BFI(4) BFI(5) BFI(6) BFI(7)) : XOPS : "v"(a), "v"(b));
    if constexpr (K == CND_K8) asm volatile(R64(CND(0) CND(1) CND(2) CND(3) CND(4) CND(5) CND(6) CND(7)) : XOPS : "v"(a), "v"(b) : "vcc");
    if constexpr (K == PKFMA_D1) asm volatile(R512(PKFMA(0)) : YOPS : "v"(a2), "v"(b2));
    if constexpr (K == PKFMA_K2) asm volatile(R256(PKFMA(0) PKFMA(1)) : YOPS : "v"(a2), "v"(b2));
    if constexpr (K == PKFMA_K4) asm volatile(R128(PKFMA(0) PKFMA(1) PKFMA(2) PKFMA(3)) : YOPS : "v"(a2), "v"(b2));
    if constexpr (K == PKFMA_K8) asm volatile(R64(PKFMA(0) PKFMA(1) PKFMA(2) PKFMA(3) PKFMA(4) PKFMA(5) PKFMA(6) PKFMA(7)) : YOPS : "v"(a2), "v"(b2));
    if constexpr (K == PKMUL_K8) asm volatile(R64(PKMUL(0) PKMUL(1) PKMUL(2) PKMUL(3) PKMUL(4) PKMUL(5) PKMUL(6) PKMUL(7)) : YOPS : "v"(a2), "v"(b2));
    if constexpr (K == PKADD_K8) asm volatile(R64(PKADD(0) PKADD(1) PKADD(2) PKADD(3) PKADD(4) PKADD(5) PKADD(6) PKADD(7)) : YOPS : "v"(a2), "v"(b2));
    if constexpr (K == MIX_1PK_1PL)
        asm volatile(R64(MPK(0) MPL(4) MPK(1) MPL(5) MPK(2) MPL(6) MPK(3) MPL(7))
                     : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
                     : "v"(a2), "v"(b2), "v"(a), "v"(b));
    if constexpr (K == MIX_1PK_2PL)
        asm volatile(R32(MPK(0) MPL(4) MPL(5) MPK(1) MPL(6) MPL(7) MPK(2) MPL(8) MPL(9) MPK(3) MPL(10) MPL(11) MPK(0) MPL(4) MPL(5) MPK(1))
                     : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
                     : "v"(a2), "v"(b2), "v"(a), "v"(b));
    // operands 16 a, 17 b (VGPR), 18 a (SGPR), 19 an SGPR pair (lane mask), 20 LDS address
#define X8(M) R64(M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7))
    if constexpr (K == FMA_S_K8) asm volatile(X8(FMA_S) : XOPS : "v"(a), "v"(b) SOPS);
    if constexpr (K == MUL_S_K8) asm volatile(X8(MUL_S) : XOPS : "v"(a), "v"(b) SOPS);
    if constexpr (K == MUL_LIT_K8) asm volatile(X8(MUL_LIT) : XOPS : "v"(a), "v"(b) SOPS);
    if constexpr (K == ADD_INL_K8) asm volatile(X8(ADD_INL) : XOPS : "v"(a), "v"(b) SOPS);
    if constexpr (K == ADD_E64_K8) asm volatile(X8(ADD_E64) : XOPS : "v"(a), "v"(b) SOPS);
    if constexpr (K == CND_E64_K8) asm volatile(X8(CND_E64) : XOPS : "v"(a), "v"(b) SOPS);
    if constexpr (K == CMPCND_K8) asm volatile(R32(CMPCND(0) CMPCND(1) CMPCND(2) CMPCND(3) CMPCND(4) CMPCND(5) CMPCND(6) CMPCND(7)) : XOPS : "v"(a), "v"(b) SOPS : "vcc");
    if constexpr (K == CMP_K8) asm volatile(X8(CMP) : XOPS : "v"(a), "v"(b) SOPS : "vcc");
    if constexpr (K == CMP_E64_K8) { uint64_t m2 = mask64; asm volatile(X8(CMP_E64) : XOPS : "v"(a), "v"(b), "s"(a), "s"(m2), "v"(ldsaddr)); }
    if constexpr (K == AND_K8) asm volatile(X8(AND) : XOPS : "v"(a), "v"(b) SOPS);
    if constexpr (K == LSHL_K8) asm volatile(X8(LSHL) : XOPS : "v"(a), "v"(b) SOPS);
    if constexpr (K == BFE_K8) asm volatile(X8(BFE) : XOPS : "v"(a), "v"(b) SOPS);
    if constexpr (K == BITOP3_S_K8) asm volatile(X8(BITOP3_S) : XOPS : "v"(a), "v"(b) SOPS);
    if constexpr (K == MAXABS_K8) asm volatile(X8(MAXABS) : XOPS : "v"(a), "v"(b) SOPS);
    // packed with an SGPR pair: operands 8 a2, 9 b2 (VGPR pairs), 10 a2 in an SGPR pair
    if constexpr (K == PKMUL_S_K8) asm volatile(R64(PKMUL_S(0) PKMUL_S(1) PKMUL_S(2) PKMUL_S(3) PKMUL_S(4) PKMUL_S(5) PKMUL_S(6) PKMUL_S(7)) : YOPS : "v"(a2), "v"(b2), "s"(a2s));
    if constexpr (K == PKFMA_S_K8) asm volatile(R64(PKFMA_S(0) PKFMA_S(1) PKFMA_S(2) PKFMA_S(3) PKFMA_S(4) PKFMA_S(5) PKFMA_S(6) PKFMA_S(7)) : YOPS : "v"(a2), "v"(b2), "s"(a2s));
    if constexpr (K == PK_DSW)
        asm volatile(R32(PKFMA(0) PKFMA(1) PKFMA(2) PKFMA(3) PKFMA(4) PKFMA(5) PKFMA(6) PKFMA(7) PKFMA(0) PKFMA(1) PKFMA(2) PKFMA(3) PKFMA(4) PKFMA(5) PKFMA(6) "ds_write_b64 %10, %7\n")
                     : YOPS : "v"(a2), "v"(b2), "v"(ldsaddr) : "memory");
    if constexpr (K == PKFMA_AL0) asm volatile(".p2align 3\n" R64(PKFMA(0) PKFMA(1) PKFMA(2) PKFMA(3) PKFMA(4) PKFMA(5) PKFMA(6) PKFMA(7)) : YOPS : "v"(a2), "v"(b2));
    if constexpr (K == PKFMA_AL4) asm volatile(".p2align 3\ns_nop 0\n" R64(PKFMA(0) PKFMA(1) PKFMA(2) PKFMA(3) PKFMA(4) PKFMA(5) PKFMA(6) PKFMA(7)) : YOPS : "v"(a2), "v"(b2));
    if constexpr (K == FMA_AL0) asm volatile(".p2align 3\n" R64(FMA(0) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7)) : XOPS : "v"(a), "v"(b));
    if constexpr (K == FMA_AL4) asm volatile(".p2align 3\ns_nop 0\n" R64(FMA(0) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7)) : XOPS : "v"(a), "v"(b));
    if constexpr (K == MIX48_AL0) asm volatile(".p2align 3\n" R64(MUL(0) FMA(1) MUL(2) FMA(3) MUL(4) FMA(5) MUL(6) FMA(7)) : XOPS : "v"(a), "v"(b));
    if constexpr (K == MIX48_AL4) asm volatile(".p2align 3\ns_nop 0\n" R64(MUL(0) FMA(1) MUL(2) FMA(3) MUL(4) FMA(5) MUL(6) FMA(7)) : XOPS : "v"(a), "v"(b));
    if constexpr (K == FMA_LO32) asm volatile("s_mov_b64 exec, 0xffffffff\n" R64(FMA(0) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7)) "s_mov_b64 exec, -1\n" : XOPS : "v"(a), "v"(b));
    if constexpr (K == FMA_LO16) asm volatile("s_mov_b64 exec, 0xffff\n" R64(FMA(0) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7)) "s_mov_b64 exec, -1\n" : XOPS : "v"(a), "v"(b));
    if constexpr (K == PKFMA_LO32) asm volatile("s_mov_b64 exec, 0xffffffff\n" R64(PKFMA(0) PKFMA(1) PKFMA(2) PKFMA(3) PKFMA(4) PKFMA(5) PKFMA(6) PKFMA(7)) "s_mov_b64 exec, -1\n" : YOPS : "v"(a2), "v"(b2));
    if constexpr (K == DSR_WAIT)  // 8 x (one read + wait + 62 fma) = 512 instructions per block
        asm volatile(R8("ds_read2st64_b64 %[d], %[ad] offset1:1\ns_waitcnt lgkmcnt(0)\n" R2(R2(R2(FMAN(0) FMAN(1) FMAN(2) FMAN(3) FMAN(4) FMAN(5) FMAN(6)))) FMAN(7) FMAN(0) FMAN(1) FMAN(2) FMAN(3) FMAN(4))
                     : XOPS, [d] "=&v"(r.q) : [ka] "v"(a), [kb] "v"(b), [ad] "v"(ldsaddr) : "memory");
    if constexpr (K == BUFST16)
        asm volatile(R32(FMA(0) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7) FMA(0) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) "buffer_store_dword %0, %[vo], %[rs], 0 offen\n")
                     : XOPS : "v"(a), "v"(b), [vo] "v"(ldsaddr), [rs] "s"(rsrc) : "memory");
    if constexpr (K == IDLE) __builtin_amdgcn_s_sleep(127);
    if constexpr (K == SINE4 || K == SINE4_SVF8) {
        // phases advance like the oscillator's (a few thousandths of a turn per frame), argument = phase * TAU
#pragma unroll
        for (int i = 0; i < 4; i++) {
            y[i] = y[i] + b2;
            y[4 + i] = fd::wide_sin2(y[i] * 6.2831855f, guard);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) asm volatile("" : "+v"(y[4 + i]));  // the samples are "stored": evaluated every trip
    }
    if constexpr (K == SVF8 || K == SINE4_SVF8) {
        // FixedSvfLp packed path (fd_nodes.hpp SvfCore, lowpass shortcut): x[0] ic1eq, x[1] ic2eq, x[2..4] a1 a2 a3
        float ic1 = x[0], ic2 = x[1];
        const float a1 = x[2], a2c = x[3], a3 = x[4];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float v0 = K == SINE4_SVF8 ? (i & 1 ? y[4 + i / 2].y : y[4 + i / 2].x) : x[8 + i];
            const float v3 = v0 - ic2;
            const float v1 = a1 * ic1 + a2c * v3;
            const float v2 = ic2 + a2c * ic1 + a3 * v3;
            guard = __builtin_fmaxf(guard, __builtin_fmaxf(__builtin_fabsf(v1), __builtin_fabsf(v2)));
            ic1 = __builtin_fmaf(2.0f, v1, -ic1);
            ic2 = __builtin_fmaf(2.0f, v2, -ic2);
            x[8 + i] = v2;  // "output"
            if (K == SVF8) asm volatile("" : "+v"(x[8 + i]));
        }
        x[0] = ic1;
        x[1] = ic2;
    }
}

__device__ __forceinline__ uint64_t memtime() {
    uint64_t t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ uint64_t memrealtime() {
    uint64_t t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

struct Rec {
    uint64_t cycles, real;
    uint32_t hw, xcc;
    float sink;
    uint32_t pad;
};

template <int KA, int KB>
__global__ __launch_bounds__(1024) void k(Rec* rec, int reps, float a, float b, int lds_floats, int prio_b) {
    extern __shared__ float lds[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Regs r;
#pragma unroll
    for (int i = 0; i < 16; i++) r.x[i] = 0.25f + 0.001f * (float)(threadIdx.x & 63) + 0.01f * i;
#pragma unroll
    for (int i = 0; i < 8; i++) r.y[i] = v2f{r.x[i], r.x[i + 8]};
    r.x[2] = 0.02f; r.x[3] = 0.1f; r.x[4] = 0.01f;  // SVF coefficients of a lowpass well inside its stable range
    if (lds_floats > 0) lds[threadIdx.x % lds_floats] = r.x[0];
    const v2f a2 = {a, a}, b2 = {b, b * 0.5f};
    float guard = 0.0f;
    const uint64_t mask64 = __builtin_amdgcn_ballot_w64(r.x[0] < r.x[5]);   // a lane mask in an SGPR pair
    const uint32_t abits = (uint32_t)__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a));
    const uint64_t a2s = ((uint64_t)abits << 32) | abits;  // {a, a} in an SGPR pair
    const int ldsaddr = (threadIdx.x & 1023) * 8;
    // a scratch row per wave for the store kind: 16 KiB behind the records, rewritten over and over
    const uint64_t rowp = reinterpret_cast<uint64_t>(rec) + (1 << 20) + (uint64_t)blockIdx.x * 16384;
    const v4i_ rsrc = {__builtin_amdgcn_readfirstlane((int)(uint32_t)rowp), __builtin_amdgcn_readfirstlane((int)(uint32_t)(rowp >> 32) & 0xffff), 16384, 0x00020000};
    if (prio_b && ((w >> 2) & 1)) __builtin_amdgcn_s_setprio(1);
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(r.x[0]), "v"(r.x[1]) : "vcc");
    __syncthreads();
    const uint64_t t0 = memtime(), r0 = memrealtime();
    if (((w >> 2) & 1) == 0) {
#pragma unroll 1
        for (int it = 0; it < reps; it++) block<KA>(r, a, b, a2, b2, guard, mask64, a2s, ldsaddr, rsrc);
    } else {
#pragma unroll 1
        for (int it = 0; it < reps; it++) block<KB>(r, a, b, a2, b2, guard, mask64, a2s, ldsaddr, rsrc);
    }
    const uint64_t t1 = memtime(), r1 = memrealtime();
    float s = guard;
#pragma unroll
    for (int i = 0; i < 16; i++) s += r.x[i];
#pragma unroll
    for (int i = 0; i < 8; i++) s += r.y[i].x + r.y[i].y;
    s += r.q.x;
    if ((threadIdx.x & 63) == 0) {
        Rec& o = rec[blockIdx.x * (blockDim.x >> 6) + w];
        o.cycles = t1 - t0;
        o.real = r1 - r0;
        o.hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
        o.xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) & 0xf;
        o.sink = s;
    }
}

static Rec* d_rec;

// insts[role]: VALU instructions per trip of the role's stream (asm kinds: 512; C++ kinds: from the disassembly)
template <int KA, int KB>
void run(int nwaves_per_simd, int grid, int instsA, int instsB, const char* note = "", int prio_b = 0) {
    const int wpb = 4 * nwaves_per_simd, reps = 64;
    const size_t lds = grid > 1 ? 100 * 1024 : 16 * 1024;  // one workgroup per CU when the whole chip is loaded
    hipFuncSetAttribute((const void*)k<KA, KB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; rep++)
        hipLaunchKernelGGL((k<KA, KB>), dim3(grid), dim3(64 * wpb), lds, 0, d_rec, reps, 0.5f, 0.001953125f, (int)(lds / 4), prio_b);
    hipDeviceSynchronize();
    std::vector<Rec> h((size_t)grid * wpb);
    hipMemcpy(h.data(), d_rec, h.size() * sizeof(Rec), hipMemcpyDeviceToHost);
    std::map<uint64_t, int> per_simd;
    double cyc[2] = {0, 0}, cmin[2] = {1e30, 1e30}, cmax[2] = {0, 0}, mhz = 0;
    int n[2] = {0, 0};
    for (int bI = 0; bI < grid; bI++)
        for (int w = 0; w < wpb; w++) {
            const Rec& q = h[(size_t)bI * wpb + w];
            const uint64_t simd = (q.hw >> 4) & 3, cu = (q.hw >> 8) & 0xf, sh = (q.hw >> 12) & 1, se = (q.hw >> 13) & 7;
            per_simd[((uint64_t)q.xcc << 20) | (se << 12) | (sh << 8) | (cu << 4) | simd]++;
            const int role = (w >> 2) & 1;
            cyc[role] += (double)q.cycles;
            cmin[role] = std::min(cmin[role], (double)q.cycles);
            cmax[role] = std::max(cmax[role], (double)q.cycles);
            n[role]++;
            mhz += (double)q.cycles / ((double)q.real / 100.0);
        }
    mhz /= (double)(grid * wpb);
    bool placed = true;
    for (auto& kv : per_simd) placed = placed && kv.second == nwaves_per_simd;
    placed = placed && (int)per_simd.size() == grid * 4;
    const int insts[2] = {instsA, instsB};
    const int kinds[2] = {KA, KB};
    printf("%-4s n/SIMD=%d grid=%-3d clock %4.0f MHz placement %s%s |", note, nwaves_per_simd, grid, mhz, placed ? "ok " : "BAD", prio_b ? " role B at s_setprio 1" : "");
    double simd_rate = 0;  // instructions per cycle per SIMD, all roles
    for (int role = 0; role < 2; role++) {
        if (!n[role]) continue;
        const double per_trip = cyc[role] / n[role] / reps;
        printf(" [%s: %.0f cyc/trip (min %.0f max %.0f)", kind_name[kinds[role]], per_trip, cmin[role] / reps, cmax[role] / reps);
        if (insts[role] > 0) {
            printf(" = %.2f cyc/inst", per_trip / insts[role]);
            simd_rate += (double)insts[role] / per_trip * (n[role] / (double)(grid * 4));
        }
        printf("]");
    }
    if (simd_rate > 0) printf(" SIMD: %.2f cyc per instruction issued", 1.0 / simd_rate);
    printf("\n");
    fflush(stdout);
}

template <int K>
void sweep(int insts = 512) {
    run<K, K>(1, 1, insts, insts);
    run<K, K>(2, 1, insts, insts);
    run<K, K>(4, 1, insts, insts);
    run<K, K>(2, 256, insts, insts, "chip");
    run<K, K>(4, 256, insts, insts, "chip");
}

int main(int argc, char** argv) {
    // VALU instructions per trip of the C++ streams (tools/ubench_issue_check.sh counts them in the disassembly)
    int n_sine4 = argc > 1 ? atoi(argv[1]) : 0, n_svf8 = argc > 2 ? atoi(argv[2]) : 0, n_both = argc > 3 ? atoi(argv[3]) : 0;
    hipMalloc((void**)&d_rec, (1 << 20) + 256 * 16384);   // records + the store kind's scratch rows
    {  // warm the clocks up
        for (int i = 0; i < 200; i++) hipLaunchKernelGGL((k<FMA_K8, FMA_K8>), dim3(256), dim3(512), 0, 0, d_rec, 256, 0.5f, 0.001f, 0, 0);
        hipDeviceSynchronize();
    }
    printf("# cycles = s_memtime ticks (shader clock); clock = s_memtime / s_memrealtime(100 MHz); grid=1: one CU, the rest of the chip idle;\n"
           "# 'chip': 256 workgroups, one per CU (100 KiB LDS each).  512-instruction straight-line blocks x 64 trips.\n");
    printf("## plain VALU\n");
    sweep<FMA_D1>(); sweep<FMA_K2>(); sweep<FMA_K4>(); sweep<FMA_K8>(); sweep<FMA_K16>(); sweep<MUL_K8>(); sweep<ADD_D1>(); sweep<ADD_K8>();
    sweep<MAX3_K8>(); sweep<BFI_K8>(); sweep<CND_K8>();
    printf("## operand sources and lane masks (8 independent chains each)\n");
    sweep<FMA_S_K8>(); sweep<MUL_S_K8>(); sweep<MUL_LIT_K8>(); sweep<ADD_INL_K8>(); sweep<ADD_E64_K8>(); sweep<AND_K8>(); sweep<LSHL_K8>(); sweep<BFE_K8>();
    sweep<BITOP3_S_K8>(); sweep<MAXABS_K8>(); sweep<CND_E64_K8>(); sweep<CMP_K8>(); sweep<CMP_E64_K8>(); sweep<CMPCND_K8>();
    printf("## code alignment, partial EXEC, exposed LDS reads, stores\n");
    sweep<PKFMA_AL0>(); sweep<PKFMA_AL4>(); sweep<FMA_AL0>(); sweep<FMA_AL4>(); sweep<MIX48_AL0>(); sweep<MIX48_AL4>();
    sweep<FMA_LO32>(); sweep<FMA_LO16>(); sweep<PKFMA_LO32>(); sweep<DSR_WAIT>(); sweep<BUFST16>();
    printf("## packed f32\n");
    sweep<PKMUL_S_K8>(); sweep<PKFMA_S_K8>(); sweep<PK_DSW>();
    sweep<PKFMA_D1>(); sweep<PKFMA_K2>(); sweep<PKFMA_K4>(); sweep<PKFMA_K8>(); sweep<PKMUL_K8>(); sweep<PKADD_K8>();
    printf("## mixes inside one wave\n");
    sweep<MIX_1PK_1PL>(); sweep<MIX_1PK_2PL>();
    printf("## two different streams on one SIMD (waves w, w+4): role A | role B\n");
    run<PKFMA_K8, IDLE>(2, 1, 512, 0);
    run<FMA_K8, IDLE>(2, 1, 512, 0);
    run<FMA_D1, IDLE>(2, 1, 512, 0);
    run<PKFMA_K8, FMA_D1>(2, 1, 512, 512);
    run<PKFMA_K8, FMA_K2>(2, 1, 512, 512);
    run<PKFMA_K8, FMA_K8>(2, 1, 512, 512);
    run<PKFMA_K4, FMA_K4>(2, 1, 512, 512);
    run<PKFMA_D1, FMA_D1>(2, 1, 512, 512);
    run<PKFMA_K8, FMA_D1>(4, 1, 512, 512);
    run<PKFMA_K8, FMA_K8>(4, 1, 512, 512);
    run<PKFMA_K8, FMA_D1>(2, 256, 512, 512, "chip");
    run<PKFMA_K8, FMA_D1>(4, 256, 512, 512, "chip");
    printf("## ... with the second (younger) role at s_setprio 1\n");
    run<PKFMA_K8, FMA_D1>(2, 1, 512, 512, "", 1);
    run<PKFMA_K8, FMA_K8>(2, 1, 512, 512, "", 1);
    run<PKFMA_K8, MUL_K8>(2, 1, 512, 512, "", 1);
    run<PKFMA_K8, MUL_K8>(2, 1, 512, 512, "", 0);
    run<FMA_K8, MUL_K8>(2, 1, 512, 512, "", 1);
    run<PKFMA_K8, PKFMA_K8>(2, 1, 512, 512, "", 1);
    run<PKFMA_K8, FMA_D1>(4, 1, 512, 512, "", 1);
    run<PKFMA_K8, FMA_D1>(2, 256, 512, 512, "chip", 1);
    printf("## the engine's own arithmetic (C++, compiled like the engine): VALU instructions per trip %d / %d / %d\n", n_sine4, n_svf8, n_both);
    run<SINE4, IDLE>(2, 1, n_sine4, 0);
    run<SVF8, IDLE>(2, 1, n_svf8, 0);
    run<SINE4_SVF8, IDLE>(2, 1, n_both, 0);
    run<SINE4, SINE4>(2, 1, n_sine4, n_sine4);
    run<SINE4, SVF8>(2, 1, n_sine4, n_svf8);
    run<SINE4, SINE4_SVF8>(2, 1, n_sine4, n_both);
    run<SINE4, SINE4_SVF8>(2, 256, n_sine4, n_both, "chip");
    run<SINE4, SINE4_SVF8>(4, 1, n_sine4, n_both);
    run<SINE4, SINE4_SVF8>(4, 256, n_sine4, n_both, "chip");
    run<SINE4, SINE4>(4, 256, n_sine4, n_sine4, "chip");
    run<SVF8, SVF8>(4, 256, n_svf8, n_svf8, "chip");
    return 0;
}

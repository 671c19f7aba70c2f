// tools/looplab/lab_template.hip -- carrier of the loop lab (design tool): two roles per SIMD (waves w, w + 4), each running a
// loop BODY that tools/looplab/make_lab.py splices into the compiled assembly in place of the marker comments below -- the
// engine's own hot loops, instruction for instruction, so that single changes to them (alignment padding, no stores, no
// SALU, operand kinds ...) can be timed in isolation: per-wave s_memtime cycles per trip, role A alone / role B alone / both.
#include <hip/hip_runtime.h>
#include <cstdint>
struct Rec { uint64_t cycles; uint32_t hw, role; };
#define CLOB_V "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119"
#define CLOB_S "s0","s1","s2","s3","s4","s5","s6","s7","s8","s9","s10","s11","s12","s13","s14","s15","s16","s17","s18","s19","s40","s41","s42","s43","s44","s45","s46","s47","s60","s61","s62","s63","s64","s65","s66","s67","s68","s69","s70","s71","s72","s73","s74","s75","s76","s77","s78","s79","s80","s81","s82","s83","s84","s85","s86","s87","s88","s89","s90","s91","s92","s93","s94","s95","s96","s97","s98","s99","vcc","scc","memory"
#ifndef LAB_WAVES
#define LAB_WAVES 8  // waves per workgroup: 8 = two per SIMD (the engine's kernel), 16 = four per SIMD
#endif
#ifndef LAB_TILE
#define LAB_TILE 8  // trips between barriers in TILE mode (8 = 64-frame tiles)
#endif
#ifndef LAB_A_SPLIT
#define LAB_A_SPLIT 1  // 2: role A is time-split over two waves per SIMD (12-wave workgroups: A, B, A), each doing half of the trips
#endif
#ifndef LAB_MERGED
#define LAB_MERGED 0  // 1: every wave runs body A then body B each trip (the unsplit voice), no roles
#endif
__global__ __launch_bounds__(LAB_WAVES * 64) void lab(Rec* rec, int reps, int mode, int prio_b) {
    extern __shared__ float lds[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = LAB_MERGED ? 1 : (w >> 2) & 1;
    lds[threadIdx.x] = 0.0f;
    // benign values in every register the bodies read (finite floats; LDS addresses inside the allocation; a null buffer
    // resource: the stores of a body are issued and dropped)
    asm volatile("; LAB_INIT" ::: CLOB_V, CLOB_S);
    if ((prio_b == 1 && role == 1) || (prio_b == 2 && role == 0)) __builtin_amdgcn_s_setprio(1);
    __syncthreads();
    uint64_t t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (LAB_MERGED) {
#pragma unroll 1
        for (int it = 0; it < reps; it++) {
            asm volatile("; LAB_BODY_A" ::: CLOB_V, CLOB_S);
            asm volatile("; LAB_BODY_B" ::: CLOB_V, CLOB_S);
        }
    } else if (mode & 4) {
        // TILE mode: the pipeline kernel's coupling -- every role does 8 trips (one 64-frame tile), then the workgroup's barrier
#pragma unroll 1
        for (int tile = 0; tile < reps / LAB_TILE; tile++) {
            if (role == 0) {
                if (mode & 1) {
#pragma unroll 1
                    for (int it = 0; it < LAB_TILE / LAB_A_SPLIT; it++) asm volatile("; LAB_BODY_A" ::: CLOB_V, CLOB_S);
                }
            } else if (mode & 2) {
#pragma unroll 1
                for (int it = 0; it < LAB_TILE; it++) asm volatile("; LAB_BODY_B" ::: CLOB_V, CLOB_S);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    } else if (role == 0) {
        if (mode & 1) {
#pragma unroll 1
            for (int it = 0; it < reps; it++) asm volatile("; LAB_BODY_A" ::: CLOB_V, CLOB_S);
        }
    } else {
        if (mode & 2) {
#pragma unroll 1
            for (int it = 0; it < reps; it++) asm volatile("; LAB_BODY_B" ::: CLOB_V, CLOB_S);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0) {
        Rec& o = rec[blockIdx.x * LAB_WAVES + w];
        o.cycles = t1 - t0;
        o.hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
        o.role = (uint32_t)role;
    }
}

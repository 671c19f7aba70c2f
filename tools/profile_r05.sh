#!/bin/bash
# tools/profile_r05.sh -- round 5, on the GPU box: rocprofv3 kernel stats of the bench's workloads (configs 3, 2, 4 in both gate shapes, 5),
# the two HBM counter passes (profiles/pmc_latest.json), and the STALL ATTRIBUTION of the headline kernel: three SQ counter passes
# (8 counters each, never combined with a trace) over `bench.py --steps 2 --warmup 1`, repeated on the stage knock-out builds when
# they are present (variants/libfundsp_hip_c3_k1.so = only stage 1 runs, _k2 = only stage 0; tools/variants/README.md).
OUT=$PWD/gpurun_out/prof_r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
kt() {  # name, command...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_$name -o kt -- "$@" > $OUT/kt_$name.out 2> $OUT/kt_$name.log
  local DB=$(find $OUT/kt_$name -name "*.db" | head -1)
  python tools/rocprof_summary.py $DB "r05: rocprofv3 --kernel-trace --stats -- $*" > $OUT/kernel_stats_$name.txt
}
kt c3 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-secondary
kt c2 python bench.py --config 2 --steps 20 --warmup 5 --cpu-seconds 0 --no-secondary
kt c5 python bench.py --config 5 --steps 10 --warmup 3 --cpu-seconds 0 --no-secondary
kt c4 python bench.py --config 4 --steps 10 --warmup 3 --cpu-seconds 0 --no-secondary
ONLY=4v kt c4v python tools/probe_c4var.py
bash tools/pmc_hbm_pass.sh r05 > $OUT/pmc_hbm.log 2>&1
PASS_A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
PASS_B="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
PASS_C="SQ_LDS_IDX_ACTIVE SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_BRANCH SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
sq() {  # tag, lib ("" = the product library)
  local tag=$1 lib=$2
  for p in A B C; do
    local ctrs; eval ctrs=\$PASS_$p
    FUNDSP_HIP_LIB=$lib timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/sq_${tag}_$p -o pmc -- python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-secondary > $OUT/sq_${tag}_$p.log 2>&1
    local CSV=$(find $OUT/sq_${tag}_$p -name "*counter_collection.csv" | head -1)
    python tools/pmc_summary.py $CSV "r05: SQ pass $p ($ctrs), python bench.py --steps 2 --warmup 1, library: ${lib:-product}" > $OUT/sq_${tag}_$p.txt
  done
}
sq full ""
[ -f variants/libfundsp_hip_c3_k1.so ] && sq stage1_alone variants/libfundsp_hip_c3_k1.so
[ -f variants/libfundsp_hip_c3_k2.so ] && sq stage0_alone variants/libfundsp_hip_c3_k2.so
rm -rf $OUT/kt_*/ $OUT/sq_*/ $OUT/pmc_*/ 2>/dev/null
ls $OUT

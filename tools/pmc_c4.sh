#!/bin/bash
# tools/pmc_c4.sh -- PMC passes over the config-4 kernel (32768 voices x 48000 frames): the workload's voices, every voice >= 400 Hz (short
# tables), every voice identical (all lanes gather from the same lines).  Separate passes per counter group (no trace domains with --pmc).
cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/pmc_c4
mkdir -p $OUT
export TMPDIR=/tmp
declare -A G
G[sq]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
G[vmem]="SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL"
G[tcp]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
G[ic]="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
for grp in sq vmem tcp ic; do
  for cond in normal fmin400 uniform; do
    case $cond in normal) A="";; fmin400) A="--fmin 400";; uniform) A="--uniform";; esac
    timeout 180 rocprofv3 --pmc ${G[$grp]} --output-format csv -d $OUT/${grp}_$cond -o pmc -- python tools/c4_ab.py --splits 0 $A > $OUT/${grp}_$cond.log 2>&1
    CSV=$(find $OUT/${grp}_$cond -name "*counter_collection.csv" | head -1)
    python tools/pmc_summary.py $CSV "r04 config 4 ($cond voices): $grp counters" | grep -A12 "k_render_pipe<" > $OUT/${grp}_$cond.txt
  done
done
cat $OUT/*.txt > gpurun_out/r04_pmc_c4.txt

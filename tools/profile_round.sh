#!/bin/bash
# tools/profile_round.sh TAG -- the rocprofv3 passes behind profiles/<TAG>_*: kernel-trace stats of the three bench
# configs, then separate PMC passes (HBM FETCH/WRITE with calibration kernels, SQ counters).  Run on the GPU box.
# CFGS="4 4fast" limits the kernel-trace passes, PMC=0 skips the counter passes.
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for cfg in ${CFGS:-3 3fast 2 4 4fast 5}; do
  ARGS="--config ${cfg%fast} --steps 10 --warmup 3 --cpu-seconds 0 --no-secondary"
  if [ "$cfg" != "${cfg%fast}" ]; then ARGS="$ARGS --math fast"; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt$cfg -o kt -- python bench.py $ARGS > $OUT/bench_c$cfg.json 2> $OUT/kt$cfg.log
  DB=$(find $OUT/kt$cfg -name "*.db" | head -1)
  python tools/rocprof_summary.py $DB "$TAG: rocprofv3 --kernel-trace --stats -- python bench.py $ARGS" > $OUT/kernel_stats_c$cfg.txt
done
if [ "${PMC:-1}" = "0" ]; then ls $OUT; exit 0; fi
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$ctr -o pmc -- python tools/traffic_workload.py > $OUT/pmc_$ctr.log 2>&1
  CSV=$(find $OUT/pmc_$ctr -name "*counter_collection.csv" | head -1)
  python tools/pmc_summary.py $CSV "$TAG: $ctr pass (KB), tools/traffic_workload.py: 3 x config-3 render (65536 x 48000) + fill + copy calibration" > $OUT/pmc_$ctr.txt
done
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -o pmc -- python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-secondary > $OUT/pmc_sq.log 2>&1
CSV=$(find $OUT/pmc_sq -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py $CSV "$TAG: SQ counters, python bench.py --steps 2 --warmup 1 (config 3)" > $OUT/pmc_sq.txt
ls $OUT; tail -3 $OUT/pmc_sq.log

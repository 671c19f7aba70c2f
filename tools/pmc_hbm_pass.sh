#!/bin/bash
# tools/pmc_hbm_pass.sh TAG -- only the two HBM counter passes of profile_round.sh (FETCH_SIZE, WRITE_SIZE; separate rocprofv3 --pmc runs
# of tools/traffic_workload.py with its copy / fill calibration kernels), each under a short timeout.  Run on the GPU box.
TAG=${1:-r04b}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 100 rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$ctr -o pmc -- python tools/traffic_workload.py > $OUT/pmc_$ctr.log 2>&1
  CSV=$(find $OUT/pmc_$ctr -name "*counter_collection.csv" | head -1)
  python tools/pmc_summary.py $CSV "$TAG: $ctr pass (KB), tools/traffic_workload.py: 3 x config-3 render (65536 x 48000) + fill + copy calibration" > $OUT/pmc_$ctr.txt
done
ls $OUT

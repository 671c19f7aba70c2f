#!/bin/bash
# tools/pmc_hbm_others.sh TAG -- HBM counter passes (FETCH_SIZE, WRITE_SIZE; separate rocprofv3 --pmc runs, never with a trace) over the OTHER
# kernels' workloads of tools/traffic_workload.py (TRAFFIC_WORKLOAD = c2 | c4v | c5 | c5r4), each with the same fill / copy calibration.
TAG=${1:-r05f}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for wl in ${WORKLOADS:-c5 c5r4 c2 c4v}; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    TRAFFIC_WORKLOAD=$wl timeout 120 rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_${wl}_$ctr -o pmc -- python tools/traffic_workload.py > $OUT/pmc_${wl}_$ctr.log 2>&1
    CSV=$(find $OUT/pmc_${wl}_$ctr -name "*counter_collection.csv" | head -1)
    python tools/pmc_summary.py $CSV "$TAG: $ctr pass (KB), TRAFFIC_WORKLOAD=$wl tools/traffic_workload.py: 3 x render + fill + copy calibration" > $OUT/pmc_${wl}_$ctr.txt
  done
done
rm -rf $OUT/pmc_*/
ls $OUT

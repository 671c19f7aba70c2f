#!/bin/bash
# tools/profile_r04.sh -- round 4: rocprofv3 kernel stats of config 4 with the fused mix-down (bench.py --config 4 --mix) and of the default
# bench, then separate FETCH_SIZE / WRITE_SIZE passes over tools/traffic_mix.py (voice-out vs fused mix, with fill + copy calibration).
OUT=$PWD/gpurun_out/prof_r04
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--config 4 --mix --steps 10 --warmup 3 --cpu-seconds 0 --no-secondary"
rocprofv3 --kernel-trace --stats -d $OUT/kt4mix -o kt -- python bench.py $ARGS > $OUT/bench_c4mix.json 2> $OUT/kt4mix.log
DB=$(find $OUT/kt4mix -name "*.db" | head -1)
python tools/rocprof_summary.py $DB "r04: rocprofv3 --kernel-trace --stats -- python bench.py $ARGS" > $OUT/kernel_stats_c4mix.txt
ARGS="--steps 10 --warmup 3 --cpu-seconds 0 --no-secondary"
rocprofv3 --kernel-trace --stats -d $OUT/kt3 -o kt -- python bench.py $ARGS > $OUT/bench_c3.json 2> $OUT/kt3.log
DB=$(find $OUT/kt3 -name "*.db" | head -1)
python tools/rocprof_summary.py $DB "r04: rocprofv3 --kernel-trace --stats -- python bench.py $ARGS" > $OUT/kernel_stats_c3.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_mix_$ctr -o pmc -- python tools/traffic_mix.py > $OUT/pmc_mix_$ctr.log 2>&1
  CSV=$(find $OUT/pmc_mix_$ctr -name "*counter_collection.csv" | head -1)
  python tools/pmc_summary.py $CSV "r04: $ctr pass (KB), tools/traffic_mix.py: config 4 shard (32768 x 48000): 2 x voice-out render, 2 x fused mix-down (k_render_pipe_mix + k_mix_tree), fill + copy calibration on the 12.58 GB voice-out buffer" > $OUT/pmc_mix_$ctr.txt
done
ls $OUT

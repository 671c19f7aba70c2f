#!/bin/bash
# tools/ab_r03.sh -- round-3 same-box A/B of the variant libraries (tools/build_variants.sh) on the headline workload; two
# interleaved passes so that run-to-run noise is visible next to the differences.  Usage: ab_r03.sh name[:split] ...
run() {  # name split
  if [ "$1" != "default" ]; then export FUNDSP_HIP_LIB=$PWD/variants/libfundsp_hip_$1.so; else unset FUNDSP_HIP_LIB; fi
  echo -n "$1 split=$2: "
  python bench.py --steps 8 --warmup 2 --cpu-seconds 0 --no-secondary --pipe-split $2 2>/dev/null | tail -1 | \
    python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'])"
}
for pass in 1 2; do
  for spec in "$@"; do
    name=${spec%%:*}; split=1; if [ "$spec" != "$name" ]; then split=${spec##*:}; fi
    run $name $split
  done
done

#!/bin/bash
# tools/ab_r02.sh -- same-box A/B of the variant libraries (tools/build_variants.sh) on the headline workload; two
# interleaved passes so that box-to-box and run-to-run noise is visible next to the differences.
run() {  # name lib math split
  if [ -n "$2" ]; then export FUNDSP_HIP_LIB=$PWD/variants/libfundsp_hip_$2.so; else unset FUNDSP_HIP_LIB; fi
  echo -n "$1 math=$3 split=$4: "
  python bench.py --steps 12 --warmup 3 --cpu-seconds 0 --no-secondary --math $3 --pipe-split $4 2>/dev/null | tail -1 | \
    python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'])"
}
for pass in 1 2; do
  run default "" exact 1
  run nolp nolp exact 1
  run noguard noguard exact 1
  run plain plain exact 1
  run default "" exact 3
  run default "" fast 1
  run fastplain fastplain fast 1
  run default "" fast 3
done

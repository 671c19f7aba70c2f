# tools/ab_variants.sh -- A/B the variant libraries built by tools/build_variants.sh on the headline workload
for v in "" plain unified svfpk; do
  for ps in 0 2 3; do
    if [ -n "$v" ]; then export FUNDSP_HIP_LIB=$PWD/variants/libfundsp_hip_$v.so; else unset FUNDSP_HIP_LIB; fi
    echo -n "variant=${v:-default} split=$ps: "; python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --pipe-split $ps 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'])"
  done
done

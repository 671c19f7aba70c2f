cd $GRAFT_REPO_ROOT
for pass in 1 2; do
for v in default alignmix; do
  if [ "$v" != "default" ]; then export FUNDSP_HIP_LIB=$PWD/variants/libfundsp_hip_$v.so; else unset FUNDSP_HIP_LIB; fi
  echo "== $v"; timeout 200 python tools/mix_bench.py 2>/dev/null | grep "fused\|config" | tr -d '\n'; echo
done
done

# same box: non-temporal sample stores (the default since round 4) against plain stores (variants/st0 = -DFD_PIPE_STORE_AUX=0 in the fm units)
cd $GRAFT_REPO_ROOT
head() { if [ "$1" != "default" ]; then export FUNDSP_HIP_LIB=$PWD/variants/libfundsp_hip_$1.so; else unset FUNDSP_HIP_LIB; fi
  echo -n "headline, $2: "
  timeout 200 python bench.py --steps 8 --warmup 2 --cpu-seconds 0 --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], 'ms per step, kernel', r['roofline']['kernel_ms_avg'])"; }
for pass in 1 2; do
head default "nt stores (default)"
head st0 "plain stores"
done
unset FUNDSP_HIP_LIB; echo "shards, nt stores (default):"; timeout 200 python tools/shards_bench.py 1 2>&1 | grep voices
export FUNDSP_HIP_LIB=$PWD/variants/libfundsp_hip_st0.so; echo "shards, plain stores:"; timeout 200 python tools/shards_bench.py 1 2>&1 | grep voices
unset FUNDSP_HIP_LIB; echo "shards, nt stores (default), again:"; timeout 200 python tools/shards_bench.py 1 2>&1 | grep voices

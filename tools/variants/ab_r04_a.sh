cd $GRAFT_REPO_ROOT
run() { if [ "$1" != "default" ]; then export FUNDSP_HIP_LIB=$PWD/variants/libfundsp_hip_$1.so; else unset FUNDSP_HIP_LIB; fi; shift; python tools/c4_ab.py "$@" 2>&1 | grep -v amdgpu.ids; }
for pass in 1 2; do
run default --splits 0,1 --label default
run splitprio --splits 1
run splitpair --splits 1
run splitk6 --splits 0,1 --label knock6_stage0_alone
run wrap --splits 0 --label store_wrap
done
c3() { if [ "$1" != "default" ]; then export FUNDSP_HIP_LIB=$PWD/variants/libfundsp_hip_$1.so; else unset FUNDSP_HIP_LIB; fi; echo -n "config3 $1: "; python bench.py --steps 8 --warmup 2 --cpu-seconds 0 --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['roofline'].get('power'))"; }
for pass in 1 2; do c3 default; c3 wrapfm; done

cd $GRAFT_REPO_ROOT
run() { if [ "$1" != "default" ]; then export FUNDSP_HIP_LIB=$PWD/variants/libfundsp_hip_$1.so; else unset FUNDSP_HIP_LIB; fi; shift; timeout 120 python tools/c4_ab.py "$@" 2>&1 | grep -v amdgpu.ids; }
for pass in 1 2; do
run default --splits 0 --mix --label "default"
run feednt --splits 0 --mix --label "FD_FEED_NT=1 (non-temporal gate feed)"
run stnt --splits 0 --mix --check --label "FD_PIPE_STORE_AUX=2 (non-temporal sample stores)"
run stntfeed --splits 0 --mix --check --label "FD_PIPE_STORE_AUX=2 + FD_FEED_NT=1"
done

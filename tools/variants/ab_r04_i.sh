cd $GRAFT_REPO_ROOT
run() { if [ "$1" != "default" ]; then export FUNDSP_HIP_LIB=$PWD/variants/libfundsp_hip_$1.so; else unset FUNDSP_HIP_LIB; fi; shift; timeout 120 python tools/c4_ab.py "$@" 2>&1 | grep -v amdgpu.ids; }
run default --splits 0 --mix --check --label "default (one frame pair gathered ahead)"
run pf2 --splits 0 --mix --check --label "FD_WT_PREFETCH=2 (two pairs ahead)"
run pf4 --splits 0 --mix --check --label "FD_WT_PREFETCH=4 (four pairs ahead)"
run feednt --splits 0 --mix --check --label "FD_FEED_NT=1 (non-temporal gate feed)"
run pf2nt --splits 0 --mix --check --label "FD_WT_PREFETCH=2 + FD_FEED_NT=1"
run default --splits 0 --mix --label "default again"

cd $GRAFT_REPO_ROOT
run() { if [ "$1" != "default" ]; then export FUNDSP_HIP_LIB=$PWD/variants/libfundsp_hip_$1.so; else unset FUNDSP_HIP_LIB; fi; shift; python tools/c4_ab.py "$@" 2>&1 | grep -v amdgpu.ids; }
for pass in 1 2; do
run k6 --splits 0 --label "stage0_alone"
run k6 --splits 0 --uniform --label "stage0_alone_uniform_voices(all_lanes_same_lines)"
run k6 --splits 0 --fmin 400 --label "stage0_alone_fmin400(short_tables)"
run default --splits 0 --label "whole_kernel"
run default --splits 0 --uniform --label "whole_kernel_uniform_voices"
run default --splits 0 --fmin 400 --label "whole_kernel_fmin400"
done

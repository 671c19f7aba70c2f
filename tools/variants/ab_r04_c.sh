cd $GRAFT_REPO_ROOT
run() { if [ "$1" != "default" ]; then export FUNDSP_HIP_LIB=$PWD/variants/libfundsp_hip_$1.so; else unset FUNDSP_HIP_LIB; fi; shift; timeout 120 python tools/c4_ab.py "$@" 2>&1 | grep -v amdgpu.ids; }
for pass in 1 2; do
run default --splits 0 --label "whole_kernel"
run wtnt --splits 0 --check --label "whole_kernel_nt_gathers"
run k6 --splits 0 --label "stage0_alone"
run wtntk6 --splits 0 --label "stage0_alone_nt_gathers"
run wtnt --splits 0 --fmin 400 --label "whole_kernel_nt_gathers_fmin400"
done

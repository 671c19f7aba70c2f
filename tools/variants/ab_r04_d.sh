cd $GRAFT_REPO_ROOT
run() { if [ "$1" != "default" ]; then export FUNDSP_HIP_LIB=$PWD/variants/libfundsp_hip_$1.so; else unset FUNDSP_HIP_LIB; fi; shift; timeout 120 python tools/c4_ab.py "$@" 2>&1 | grep -v amdgpu.ids; }
for pass in 1 2; do
run default --splits 0 --mix --label "whole_kernel"
run wtsame --splits 0 --mix --label "whole_kernel_both_taps_one_table(measurement_only)"
done

cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_streams.py tests/test_gpu_jit.py -x -q 2>&1 | tail -3
bash tools/pmc_hbm_pass.sh r05c > gpurun_out/pmc_r05c.log 2>&1
rm -rf gpurun_out/prof_r05c/pmc_FETCH_SIZE gpurun_out/prof_r05c/pmc_WRITE_SIZE

#!/bin/bash
# tools/profile_r06.sh -- round 6, on the GPU box: rocprofv3 kernel stats of the bench's workloads (config 3 = the default command, 2, 4 in
# the Var-gate shape, 5, the generic FDN probe and the reverb3_stereo probe), then the HBM counter passes: the headline's (profiles/pmc_latest.json) and the other
# kernels' including the new generic FDN kernel.  Counter passes are never combined with a trace.
OUT=$PWD/gpurun_out/prof_r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
kt() {  # name, command...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_$name -o kt -- "$@" > $OUT/kt_$name.out 2> $OUT/kt_$name.log
  local DB=$(find $OUT/kt_$name -name "*.db" | head -1)
  python tools/rocprof_summary.py $DB "r06: rocprofv3 --kernel-trace --stats -- $*" > $OUT/kernel_stats_$name.txt
}
kt c3 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-secondary
kt c2 python bench.py --config 2 --steps 20 --warmup 5 --cpu-seconds 0 --no-secondary
kt c5 python bench.py --config 5 --steps 10 --warmup 3 --cpu-seconds 0 --no-secondary
kt c4 python bench.py --config 4 --steps 10 --warmup 3 --cpu-seconds 0 --no-secondary
kt fdn python tools/probe_fdn_generic.py
kt rv3 python tools/probe_reverb3.py
kt criterion python bench.py --criterion --cpu-seconds 4
bash tools/pmc_hbm_pass.sh r06 > $OUT/pmc_hbm.log 2>&1
WORKLOADS="rv3 fdn16 c5 c4v" bash tools/pmc_hbm_others.sh r06 > $OUT/pmc_others.log 2>&1
rm -rf $OUT/kt_*/ $OUT/pmc_*/ 2>/dev/null
ls $OUT

#!/bin/bash
# tools/power_probe.sh [variant ...] -- is the headline kernel clock- or power-limited?  Runs the headline bench for ~15 s per
# library and samples rocm-smi (socket power, sclk, temperature) once a second while it runs.  Design tool.
for name in "$@"; do
  if [ "$name" != "default" ]; then export FUNDSP_HIP_LIB=$PWD/variants/libfundsp_hip_$name.so; else unset FUNDSP_HIP_LIB; fi
  echo "== $name"
  python bench.py --steps 3000 --warmup 2 --cpu-seconds 0 --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('ms_per_step', r['ms_per_step'], 'kernel', r['roofline']['kernel_ms_avg'])" &
  BP=$!
  sleep 6   # import + spin-up
  for k in 1 2 3 4 5 6 7 8; do
    /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|junction|Temperature \(Sensor edge|hotspot" | tr '\n' ' '; echo
    sleep 1
  done
  wait $BP
done
echo "== caps"; /opt/rocm/bin/rocm-smi --showmaxpower 2>/dev/null | grep -i power

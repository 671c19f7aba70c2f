for rep in 1 2 3; do
for lib in "" variants/libfundsp_hip_fdn_unaligned.so; do
for r in stereo 4; do
FUNDSP_HIP_LIB=$lib python bench.py --config 5 --reverb $r --steps 20 --warmup 5 --cpu-seconds 0 --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep lib=${lib:-product} reverb=$r', d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline'].get('power',{}).get('socket_watts'), d['roofline'].get('power',{}).get('sclk_mhz'))"
done; done; done

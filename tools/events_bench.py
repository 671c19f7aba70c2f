"""Voice scheduler (fdsp_bank_process_events) against plain rendering of the same bank: 65 536 FM voices x 12 000 frames,
every voice inside its event for the whole launch, a quarter of them fading.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fundsp_amd as F
from fundsp_amd import workloads as W
V, T, SR = 65536, 12000, 48000.0
b = W.make_fm_svf_bank(V, SR)
out = torch.empty((1, T, V), dtype=torch.float32, device="cuda")
def best(fn, n=5):
    ms = []
    for _ in range(n):
        fn(); torch.cuda.synchronize(); ms.append(b.last_kernel_ms())
    return min(ms)
t_plain = best(lambda: b.process(T, out=out))
start = np.zeros(V); end = np.full(V, 10.0)
fade_in = np.where(np.arange(V) % 4 == 0, 0.1, 0.0)
b.set_events(start, end, fade_in=fade_in, fade_out=0.0)
def ev():
    b.events_rewind(0.0)
    b.process_events(T, out=out)
t_ev = best(ev)
def ev_sustained():          # a launch after every fade has ended: every voice sustains it -> dispatched to the render kernels
    b.events_rewind(0.2)
    b.process_events(T, out=out)
t_sus = best(ev_sustained)
print(f"events plain render {t_plain:.3f} ms ({V*T/t_plain/1e3:.0f} Msamples/s)  scheduler with fades {t_ev:.3f} ms "
      f"({V*T/t_ev/1e3:.0f} Msamples/s)  fully sustained launch {t_sus:.3f} ms ({V*T/t_sus/1e3:.0f} Msamples/s)")

"""Real-time style rendering (one 64-frame block per launch) captured into a HIP graph: N block launches of
fdsp_bank_process recorded once on a capturing stream, then replayed.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fundsp_amd as F
from fundsp_amd import workloads as W
V, SR, NB = 65536, 48000.0, 16
p = W.fm_svf_params(V, SR)
ref = W.make_fm_svf_bank(V, SR, params=p)
want = ref.process(64 * NB * 3)                       # three replays' worth in one launch
b = W.make_fm_svf_bank(V, SR, params=p)
outs = [torch.empty((1, 64, V), dtype=torch.float32, device="cuda") for _ in range(NB)]
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    b.process(64, out=outs[0]); b.reset(); b.set_seed(p["seed"])     # warm the kernels outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for k in range(NB):
            b.process(64, out=outs[k])
    got = []
    for r in range(3):
        g.replay()
        torch.cuda.synchronize()
        got.append(torch.cat(outs, dim=1).clone())
    torch.cuda.synchronize()
got = torch.cat(got, dim=1)
b2 = W.make_fm_svf_bank(V, SR, params=p)
loop_out = torch.cat([b2.process(64).clone() for _ in range(NB * 3)], dim=1)
torch.cuda.synchronize()
print("plain 64-frame launches equal one long launch:", bool(torch.equal(loop_out.view(torch.int32), want.view(torch.int32))),
      " graph equals plain launches:", bool(torch.equal(loop_out.view(torch.int32), got.view(torch.int32))))
same = bool(torch.equal(got.view(torch.int32), want.view(torch.int32)))
print("graph replay equals one long launch:", same)
if not same:
    d = (got.view(torch.int32) != want.view(torch.int32))[0]
    print("  first differing frame:", int(d.any(dim=1).nonzero()[0]), " voices differing there:", int(d[int(d.any(dim=1).nonzero()[0])].sum()))
def timed(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
with torch.cuda.stream(s):
    t_graph = timed(lambda: g.replay())
    def loop():
        for k in range(NB): b.process(64, out=outs[k])
    t_loop = timed(loop)
print(f"graphcap {NB} blocks of 64 frames x {V} voices: launch loop {t_loop/NB*1e6:.1f} us/block, graph replay {t_graph/NB*1e6:.1f} us/block")

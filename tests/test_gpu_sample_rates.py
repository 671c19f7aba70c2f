"""AudioNode::set_sample_rate (audionode.rs:68) over a bank's life: construction at one rate, a render, a change of rate in
mid-stream (filters recompute their coefficients and keep their state, delay lines are re-sized and cleared
(delay.rs:105-113), oscillators keep their phase), another render -- against the oracle walked through the same calls."""
import numpy as np
import pytest

import oracle as O
from fundsp_amd import LAYOUT_VOICE_MINOR, MODE_PROCESS, MODE_TICK
from fundsp_amd import graph as GR
from test_gpu_parity import assert_bit_equal, oracle_render, run_bank

pytestmark = pytest.mark.gpu

GRAPHS = {
    "filters": lambda m: m.lowpass_hz(1200.0, 1.5) >> m.bell_hz(900.0, 1.2, 2.0) >> m.moog_hz(2000.0, 0.3) >> m.dcblock_hz(10.0),
    "oscillators": lambda m: m.sine_hz(441.0) + m.saw_hz(110.0) * 0.3 + m.pass_() * 0.0,
    "poly_dsf": lambda m: (m.pass_() * 100.0 + 300.0) >> ((m.dsf_saw_r(0.6) * 0.5) & (m.poly_saw() * 0.25)) >> m.lowpole_hz(3000.0),
    "delays": lambda m: m.allnest_c(0.5, m.delay(0.002)) >> m.feedback(m.delay(0.001) * 0.7) >> m.butterpass_hz(4000.0),
    "dynamics": lambda m: m.pass_() * 3.0 >> m.limiter(0.002, 0.02) >> m.follow(0.01) >> m.declick(),
    "adsr_gate": lambda m: m.adsr_live(0.003, 0.004, 0.5, 0.005) * 0.5,
}
RATES = [(44100.0, 96000.0), (96000.0, 22050.0), (11025.5, 48000.0), (192000.0, 44100.0)]
RING = 1024   # >= 0.002 s at 192 kHz, a power of two


@pytest.mark.parametrize("name", list(GRAPHS))
def test_sample_rate_changes_in_mid_stream(gpu, name):
    V, T1, T2 = 5, 64 * 2 + 5, 64 * 3 + 9
    rng = np.random.default_rng(31)
    x = (rng.standard_normal((V, 1, T1 + T2)) * 0.5).astype(np.float32)
    if "saw" in name or name == "oscillators":
        t = O.Wavetable.get("saw")
        offs = np.concatenate([[0], np.cumsum(t.lengths)])
        gpu.wavetable_upload("saw", t.pitches, [t.data[offs[i]:offs[i + 1]] for i in range(len(t.lengths))])
    for sr1, sr2 in RATES:
        for mode in (MODE_PROCESS, MODE_TICK):
            b = gpu.Bank.from_graph(GRAPHS[name](GR), V, ring_frames=RING, sample_rate=sr1)
            b.set_seed(np.arange(V, dtype=np.uint64))
            xa, xb = np.ascontiguousarray(x[:, :, :T1]), np.ascontiguousarray(x[:, :, T1:])
            got1 = run_bank(b, xa, T1, LAYOUT_VOICE_MINOR, mode)
            b.set_sample_rate(sr2)
            got2 = run_bank(b, xb, T2, LAYOUT_VOICE_MINOR, mode)
            for v in range(V):
                n = GRAPHS[name](O)
                n.set_sample_rate(sr1)
                n.set_seed(v)
                assert_bit_equal(got1[v], oracle_render(n, xa[v], T1, mode), f"{name} {sr1} voice {v} mode {mode}")
                n.set_sample_rate(sr2)
                assert_bit_equal(got2[v], oracle_render(n, xb[v], T2, mode), f"{name} {sr1}->{sr2} voice {v} mode {mode}")

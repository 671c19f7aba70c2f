"""BASELINE config 4 in the reference's own gate shape: the envelope's gate is a shared variable, not an audio stream --
`var(&control) >> adsr_live(..)` (examples/live_adsr.rs:72; SURVEY.md 8(d) writes `dc(gate) >> adsr_live`).  Var::process reads the
variable ONCE per 64-sample block and splats it (src/shared.rs:122-125), so the host flips it between launches
(fdsp_bank_set_param_all on the Var's slot = Shared::set_value, src/shared.rs:98-101).  The kind `saw_moog_var_adsr_pan` has no
input: no gate stream in HBM, no loader wave, no feed ring.

Bar: bit-exact against the oracle's graph of the same shape, rendered launch by launch with the variable set in between; the fused
mix-down (mode B) equals the mix of the voice-out render bit for bit."""
import numpy as np
import pytest

import oracle as O
from mix_order import mix_order_reference
from fundsp_amd import LAYOUT_PLANAR, LAYOUT_VOICE_MINOR, MIX_SUM, MODE_PROCESS, MODE_TICK
from fundsp_amd import workloads as W
from test_gpu_parity import assert_bit_equal

pytestmark = pytest.mark.gpu
SR = 48000.0
ADSR = (0.005, 0.01, 0.6, 0.01)


@pytest.fixture(scope="module")
def tables(gpu):
    gpu.wavetable_build("saw")
    return True


def oracle_voice(p, v, adsr):
    """-> (graph, its var node)"""
    gate = O.var(0.0)
    g = (((O.dc(float(p["f"][v])) >> O.saw()) | O.dc(float(p["fc"][v])) | O.dc(float(p["q"][v]))) >> O.moog()) \
        * (gate >> O.adsr_live(*adsr)) >> O.pan(float(p["pan"][v]))
    g.set_sample_rate(SR)
    g.set_seed(int(p["seed"][v]))
    return g, gate


def oracle_plan(p, v, adsr, plan, mode):
    g, gate = oracle_voice(p, v, adsr)
    parts = []
    for value, n in plan:
        gate.set_value(value)
        parts.append(g.render_blocks(length=n) if mode == MODE_PROCESS else g.render_ticks(length=n))
    return np.concatenate(parts, axis=1)


def device_plan(bank, plan, layout=LAYOUT_VOICE_MINOR, mode=MODE_PROCESS, per_voice_gate=False):
    """-> [V][2][frames]: one launch per plan entry, the Var slot set before each"""
    import torch

    parts = []
    for value, n in plan:
        if per_voice_gate:   # the per-voice form of the setter (a host array)
            bank.set_param(W.C4V_SLOTS["gate"], np.full(bank.voices, value, dtype=np.float32))
        else:
            bank.set_param(W.C4V_SLOTS["gate"], float(value))
        if layout == LAYOUT_VOICE_MINOR:
            parts.append(bank.process(n, layout=layout, mode=mode).permute(2, 0, 1))
        else:
            parts.append(bank.process(n, layout=layout, frame_stride=n + 64 - n % 64, mode=mode)[:, :, :n])
    torch.cuda.synchronize()
    return torch.cat(parts, dim=2).cpu().numpy()


# low block first (adsr_live attacks on a low -> high change only), a ragged launch in the middle (a new block starts with the next
# launch, as it would with separate process() calls), a re-trigger during the release
PLAN = [(0.0, 64), (1.0, 64 * 9), (0.0, 64 * 3 + 7), (0.5, 64 * 5), (0.0, 64 * 12 + 3)]


@pytest.mark.parametrize("layout", [LAYOUT_VOICE_MINOR, LAYOUT_PLANAR])
@pytest.mark.parametrize("mode", [MODE_PROCESS, MODE_TICK])
def test_config4_var_gate_voice(gpu, tables, layout, mode):
    V = 130
    p = W.saw_moog_params(V, SR)
    b = W.make_saw_moog_var_bank(V, SR, params=p, adsr=ADSR, prime=False)
    assert b.inputs() == 0 and b.outputs() == 2
    got = device_plan(b, PLAN, layout, mode, per_voice_gate=(layout == LAYOUT_PLANAR))
    for v in range(0, V, 9):
        assert_bit_equal(got[v], oracle_plan(p, v, ADSR, PLAN, mode), f"config 4 (var gate) voice {v}")
    assert np.abs(got).max() > 0.05


def test_var_gate_kind_takes_the_pipeline_kernel_and_has_no_graph_input(gpu, tables):
    """What this checks is the launch shape only (the samples are checked against the oracle's `var(g) >> adsr_live` graph above and below):
    the Var kind has no graph input -- so no loader wave and no feed ring -- and a sustained launch takes the stage pipeline.  (The two
    config-4 kinds cannot be compared sample by sample: `var >> adsr_live` hashes differently from `pass >> adsr_live`, the envelope's
    jitter stream is another one.)"""
    V, T = 64 * 5, 64 * 8
    p = W.saw_moog_params(V, SR)
    b = W.make_saw_moog_var_bank(V, SR, params=p, adsr=ADSR)
    assert b.inputs() == 0 and b.outputs() == 2
    b.set_param(W.C4V_SLOTS["gate"], 1.0)
    out = b.process(T)
    assert b.get_option("last_kernel") == 2, "the launch must take the pipeline kernel"
    assert float(out.abs().max()) > 0.05
    s = W.make_saw_moog_bank(V, SR, params=p, adsr=ADSR)
    assert s.inputs() == 1 and s.outputs() == 2, "the stream-gate kind keeps its audio-rate gate input"


@pytest.mark.parametrize("V", [130, 64 * 300 + 5, 32768])
def test_config4_var_gate_banks(gpu, tables, V):
    """Pipeline kernel geometries: one voice group per workgroup (130), two (19 205: ragged, odd group count), the per-GPU shard of the
    config (32 768).  Spot voices vs the oracle; mode B == sum_voices(voice-out) == the order's statement, state included."""
    import torch

    plan = [(1.0, 64 * 6), (0.0, 64 * 5)]
    p = W.saw_moog_params(V, SR)
    b = W.make_saw_moog_var_bank(V, SR, params=p, adsr=ADSR)      # primed: one low block
    ref = b.clone()
    got = device_plan(b, plan)
    rng = np.random.default_rng(V)
    full = [(0.0, 64)] + plan
    for v in np.concatenate([[0, 63, 64, V - 1], rng.integers(0, V, 4)]):
        want = oracle_plan(p, int(v), ADSR, full, MODE_PROCESS)[:, 64:]
        assert_bit_equal(got[int(v)], want, f"config 4 (var gate) V={V} voice {v}")
    mixes = []
    for value, n in plan:
        ref.set_param(W.C4V_SLOTS["gate"], float(value))
        mixes.append(ref.process_mix(n, mix=MIX_SUM))
    mix = torch.cat(mixes, dim=1).cpu().numpy()
    vo = np.ascontiguousarray(got.transpose(1, 2, 0))                          # [2][T][V]
    assert_bit_equal(mix, gpu.sum_voices(torch.from_numpy(vo).cuda()).cpu().numpy(), f"fused vs sum_voices(voice-out), V={V}")
    assert_bit_equal(mix, mix_order_reference(vo), "vs the order's statement")
    assert_bit_equal(b.get_state(), ref.get_state(), "state after the launches")
    assert np.abs(mix).max() > 0.05


def test_set_param_all_is_stream_ordered(gpu, tables):
    """fdsp_bank_set_param_all fills the slot on the device behind the last render: back-to-back set / render pairs without a host
    wait in between give what the same pairs with waits give."""
    import torch

    V = 64 * 40
    p = W.saw_moog_params(V, SR)
    a = W.make_saw_moog_var_bank(V, SR, params=p, adsr=ADSR)
    b = a.clone()
    plan = [(1.0, 64 * 4), (0.0, 64 * 4), (1.0, 64 * 4)]
    outs = []
    for value, n in plan:                                       # no synchronisation between the calls
        a.set_param(W.C4V_SLOTS["gate"], value)
        outs.append(a.process(n))
    torch.cuda.synchronize()
    for (value, n), o in zip(plan, outs):
        b.set_param(W.C4V_SLOTS["gate"], value)
        torch.cuda.synchronize()
        want = b.process(n)
        torch.cuda.synchronize()
        assert torch.equal(o, want)
    assert np.array_equal(a.get_slot(W.C4V_SLOTS["gate"]), np.ones(V, dtype=np.float32))

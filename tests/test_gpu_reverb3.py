"""GPU parity of reverb3_stereo(time, diffusion, lowpole_hz(cutoff)) (prelude.rs:1858-1871; the allpass-loop reverb Reverb<F> of
reverb.rs:152-279) through its lane-per-frame kernel (fdsp_reverb3_stereo_create, fd_reverb3.hip): bit-exact against the oracle's per-sample
restatement of Reverb::tick -- both layouts, ragged launches, denormals kept, reset() and set_sample_rate() with the reference's quirks (the
input diffusers survive both; a change of rate keeps every allpass's pending sample, the feedback sample and the filter values) -- and
against the run-time compiled lane-per-voice rendering of the same node (fundsp_amd.graph.reverb3_stereo through fd_jit.hip)."""
import numpy as np
import pytest

import oracle as O
from fundsp_amd import LAYOUT_PLANAR, LAYOUT_VOICE_MINOR, MODE_PROCESS, MODE_TICK
from fundsp_amd import graph as GR
from test_gpu_fdn import run
from test_gpu_parity import assert_bit_equal

pytestmark = pytest.mark.gpu
SR = 48000.0


def oracle_rv3(time, diffusion, cutoff, sr=SR):
    n = O.reverb3_stereo(time, diffusion, lambda: O.lowpole_hz(float(cutoff)))
    n.set_sample_rate(sr)
    return n


def signal(V, T, seed):
    rng = np.random.default_rng(seed)
    x = (rng.random((V, 2, T), dtype=np.float32) * 2 - 1).astype(np.float32)
    x[:, :, T // 2:] = 0.0
    if V > 1:
        x[1] *= 1e-36                    # an instance whose tail lives among the denormals (kept: no Feedback node, no flush)
    if V > 2:
        x[2] = 0.0
        x[2, 0, 0] = 1.0                 # an impulse on the left channel
    return x


@pytest.mark.parametrize("mode", [MODE_PROCESS, MODE_TICK])
@pytest.mark.parametrize("layout", [LAYOUT_PLANAR, LAYOUT_VOICE_MINOR])
def test_reverb3_stereo_matches_oracle(gpu, layout, mode):
    V, T = 5, 64 * 320 + 11             # > one trip through all eight blocks (9 500 frames) and back
    x = signal(V, T, 31)
    b = gpu.Bank.reverb3_stereo(V, 2.0, 0.6, 1800.0)
    b.set_sample_rate(SR)
    assert b.inputs() == 2 and b.outputs() == 2
    cuts = [0, 64 * 7, 64 * 7 + 13, 64 * 100 + 13, T]       # a ragged launch in the middle: the next one starts a new block
    got = run(b, x, layout, mode, cuts)
    assert b.get_option("last_kernel") == 6
    for v in range(V):
        n = oracle_rv3(2.0, 0.6, 1800.0)
        want = np.concatenate([n.render_blocks(x[v][:, a:e]) if mode == MODE_PROCESS else n.render_ticks(x[v][:, a:e]) for a, e in zip(cuts[:-1], cuts[1:])], axis=1)
        assert_bit_equal(got[v], want, f"reverb3_stereo instance {v}")
    assert np.abs(got[0][:, 15000:]).max() > 1e-4           # the loop recirculates
    assert 0.0 < np.abs(got[1]).max() < 1e-30               # the quiet instance stays where it is: nothing was flushed
    # reset() leaves the input diffusers' lines alone (reverb.rs:211-224): the second render differs from the first and equals the oracle's
    n = oracle_rv3(2.0, 0.6, 1800.0)
    n.render_blocks(x[0])
    n.reset()
    b.reset()
    again = run(b, x[:, :, :64 * 30], layout, mode, [0, 64 * 30])
    assert_bit_equal(again[0], n.render_blocks(x[0][:, :64 * 30]), "after reset(): the diffusers still hold the end of the first render")


def test_reverb3_sample_rate_change_keeps_what_the_reference_keeps(gpu):
    """Reverb::set_sample_rate (reverb.rs:226-238) resizes and empties the loop's lines (Delay::set_sample_rate) but touches neither `pre` nor any
    allpass's z, nor the feedback sample, nor the filters' values: a bank moved from 48 kHz to 44.1 kHz in mid-tail continues exactly like the
    oracle node moved the same way -- and unlike a fresh one."""
    V, T1, T2 = 3, 64 * 200 + 5, 64 * 260
    x = signal(V, T1 + T2, 41)
    x[:, :, T1:] = 0.0                                      # the second half is the tail alone
    x[:, :, : T1] = (np.random.default_rng(5).random((V, 2, T1), dtype=np.float32) * 2 - 1).astype(np.float32)
    b = gpu.Bank.reverb3_stereo(V, 3.0, 0.3, 6000.0)
    b.set_sample_rate(SR)
    run(b, x[:, :, :T1], LAYOUT_PLANAR, MODE_PROCESS, [0, T1])
    b.set_sample_rate(44100.0)
    got = run(b, x[:, :, T1:], LAYOUT_PLANAR, MODE_PROCESS, [0, 64 * 3 + 1, T2])
    for v in range(V):
        n = oracle_rv3(3.0, 0.3, 6000.0)
        n.render_blocks(x[v][:, :T1])
        n.set_sample_rate(44100.0)
        assert_bit_equal(got[v], n.render_blocks(x[v][:, T1:]), f"instance {v} after the move to 44.1 kHz")
    assert np.abs(got[0][:, :64]).max() > 1e-3              # what survived the move sounds at once (a fresh node would be silent)
    with pytest.raises(gpu.FdspError, match="128 samples"):
        b.set_sample_rate(8000.0)
    assert run(b, x[:, :, :64], LAYOUT_PLANAR, MODE_PROCESS, [0, 64]).shape == (V, 2, 64)    # the bank keeps working at the rate it had


def test_reverb3_clone_and_staged_voice_minor(gpu):
    V, T = 70, 64 * 40 + 3
    x = signal(V, T, 51)
    a = gpu.Bank.reverb3_stereo(V, 1.5, 1.0, 9000.0)
    a.set_sample_rate(SR)
    b = a.clone()
    vm = run(a, x, LAYOUT_VOICE_MINOR, MODE_PROCESS, [0, 64 * 9 + 3, T])     # >= 64 instances: through the planar staging copy
    pl = run(b, x, LAYOUT_PLANAR, MODE_PROCESS, [0, 64 * 9 + 3, T])
    assert_bit_equal(vm, pl, "voice-minor (staged) == planar, on a clone")
    for v in (0, 63, 64, V - 1):
        assert_bit_equal(vm[v], oracle_rv3(1.5, 1.0, 9000.0).render_blocks(x[v]), f"instance {v}")
    c = a.clone()                                             # a clone in mid-tail continues like the original
    z = np.zeros((V, 2, 64 * 6), dtype=np.float32)
    assert_bit_equal(run(c, z, LAYOUT_PLANAR, MODE_PROCESS, [0, 64 * 6]), run(a, z, LAYOUT_PLANAR, MODE_PROCESS, [0, 64 * 6]), "clone in mid-tail")


def test_from_graph_takes_the_reverb3_kernel_and_equals_the_run_time_compiled_node(gpu):
    """graph.reverb3_stereo(time, diffusion, lowpole_hz(cutoff)) with scalar arguments IS the stock node: Bank.from_graph builds the lane-per-frame bank;
    fdn_kernel=False compiles the Reverb3<OnePole> node at run time and renders it one lane per voice.  Identical samples."""
    V, T = 4, 64 * 170 + 5
    mk = lambda: GR.reverb3_stereo(2.5, 0.5, lambda: GR.lowpole_hz(3000.0))
    fast = gpu.Bank.from_graph(mk(), V, sample_rate=SR)
    slow = gpu.Bank.from_graph(mk(), V, ring_frames=2048, sample_rate=SR, fdn_kernel=False)
    assert fast.kind == "reverb3_stereo" and slow.kind.startswith("jit_")
    x = signal(V, T, 61)
    a = run(fast, x, LAYOUT_VOICE_MINOR, MODE_PROCESS, [0, 64 * 50, T])
    b = run(slow, x, LAYOUT_VOICE_MINOR, MODE_PROCESS, [0, 64 * 50, T])
    assert fast.get_option("last_kernel") == 6 and slow.get_option("last_kernel") != 6
    assert_bit_equal(a, b, "lane-per-frame kernel vs run-time compiled lane-per-voice node")
    # per-voice cutoffs, or another loop filter, stay with the run-time compiler
    g = GR.reverb3_stereo(2.5, 0.5, lambda: GR.lowpole_hz(np.linspace(900.0, 4000.0, V).astype(np.float32)))
    assert getattr(g, "reverb3_plan", None) is None
    g = GR.reverb3_stereo(2.5, 0.5, lambda: GR.dcblock_hz(30.0))
    assert getattr(g, "reverb3_plan", None) is None
    g = GR.reverb3_stereo(2.5, 0.5, lambda: GR.highshelf_hz(5000.0, 1.0, 0.9))          # a FixedSvf with scalar parameters: the kernel's other filter
    assert g.reverb3_plan["svf"] == 8


def test_reverb3_bad_arguments(gpu):
    for args in ((0.0, 0.5, 1000.0), (2.0, 1.5, 1000.0), (2.0, 0.5, 0.0)):
        with pytest.raises(gpu.FdspError, match="reverb3"):
            gpu.Bank.reverb3_stereo(2, *args)


SVF_FILTERS = [  # loop filters of the FixedSvf family: (graph / oracle constructor name, args)
    ("highshelf_hz", (5000.0, 1.0, 0.8912509)),     # examples/keys.rs:134: highshelf_hz(5000.0, 1.0, db_amp(-1.0))
    ("lowpass_hz", (3000.0, 0.7)),
    ("bell_hz", (1200.0, 2.0, 0.7)),
    ("notch_hz", (2000.0, 1.5)),
]


@pytest.mark.parametrize("name,args", SVF_FILTERS)
def test_reverb3_with_a_fixed_svf_as_the_loop_filter(gpu, name, args):
    """reverb3_stereo(time, diffusion, <FixedSvf>) -- the highshelf_hz of the reference's examples and three other modes -- through the same kernel
    (fdsp_reverb3_stereo_svf_create: the sixteen SVF recurrences on the eight serial lanes): built with Bank.from_graph, bit-exact against the oracle
    over ragged launches, a reset, and a change of rate in mid-tail (the coefficients follow the rate, the filters' states stay)."""
    V, T = 4, 64 * 250 + 9
    x = signal(V, T, 71)
    mk_o = lambda: O.reverb3_stereo(1.8, 0.7, lambda: getattr(O, name)(*args))
    b = gpu.Bank.from_graph(GR.reverb3_stereo(1.8, 0.7, lambda: getattr(GR, name)(*args)), V, sample_rate=SR)
    assert b.kind == "reverb3_stereo"
    cuts = [0, 64 * 5 + 3, 64 * 120 + 3, T]
    got = run(b, x, LAYOUT_PLANAR, MODE_PROCESS, cuts)
    assert b.get_option("last_kernel") == 6
    for v in range(V):
        n = mk_o()
        n.set_sample_rate(SR)
        want = np.concatenate([n.render_blocks(x[v][:, a:e]) for a, e in zip(cuts[:-1], cuts[1:])], axis=1)
        assert_bit_equal(got[v], want, f"reverb3_stereo with {name}{args}, instance {v}")
    assert np.abs(got[0][:, 12000:]).max() > 1e-4
    # on with the tail at another rate, then a reset
    b.set_sample_rate(44100.0)
    z = np.zeros((V, 2, 64 * 40), dtype=np.float32)
    tail = run(b, z, LAYOUT_VOICE_MINOR, MODE_TICK, [0, 64 * 40])
    n = mk_o()
    n.set_sample_rate(SR)
    n.render_blocks(x[0])
    n.set_sample_rate(44100.0)
    assert_bit_equal(tail[0], n.render_ticks(z[0]), "the tail after the move to 44.1 kHz")
    b.reset()
    n.reset()
    assert_bit_equal(run(b, x[:, :, :640], LAYOUT_PLANAR, MODE_PROCESS, [0, 640])[0], n.render_blocks(x[0][:, :640]), "after reset()")


def test_reverb3_svf_bad_arguments(gpu):
    with pytest.raises(gpu.FdspError, match="svf_mode"):
        gpu.Bank.reverb3_stereo(2, 2.0, 0.5, 1000.0, svf=11)
    with pytest.raises(gpu.FdspError, match="svf_mode"):
        gpu.Bank.reverb3_stereo(2, 2.0, 0.5, 1000.0, svf="lowpass", q=0.0)


def test_reverb3_at_bench_size(gpu):
    """The bench line's shape (bench.py "rv3": 2 048 instances x 48 000 frames, planar, built with Bank.from_graph): spot instances against the oracle over
    the whole second; the same second in two ragged launches equals the one launch."""
    import torch

    V, T = 2048, 48000
    b = gpu.Bank.from_graph(GR.reverb3_stereo(2.0, 0.5, lambda: GR.lowpole_hz(8000.0)), V, sample_rate=SR)
    c = b.clone()
    assert b.kind == "reverb3_stereo"
    g = torch.Generator(device="cuda").manual_seed(123)
    x = torch.rand((V, 2, T), dtype=torch.float32, device="cuda", generator=g) * 2 - 1
    one = b.process(T, x, layout=LAYOUT_PLANAR, frame_stride=T)
    cut = 64 * 411 + 29
    a1 = c.process(cut, x[:, :, :cut].contiguous(), layout=LAYOUT_PLANAR, frame_stride=cut)
    a2 = c.process(T - cut, x[:, :, cut:].contiguous(), layout=LAYOUT_PLANAR, frame_stride=T - cut)
    assert torch.equal(torch.cat([a1, a2], dim=2).view(torch.int32), one.view(torch.int32)), "two launches == one"
    for v in (0, 3, 1023, 1024, V - 1):
        assert_bit_equal(one[v].cpu().numpy(), oracle_rv3(2.0, 0.5, 8000.0).render_blocks(x[v].cpu().numpy()), f"instance {v} of {V}, {T} frames")

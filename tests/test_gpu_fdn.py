"""GPU parity of the GENERIC Hadamard feedback delay network (fdsp_fdn_create; prelude.rs:1323-1345 and its "Mono Reverb" example :1334):

    split::<N>() | multisplit::<U2, N/2>()  >>  fdn::<N, _>(stacki(|i| delay(t_i) >> fir(w)))  >>  join::<N>() | multijoin::<U2, N/2>()

through the lane-per-frame FDN kernel, bit-exact against the oracle's generic Feedback graph in both executors (Join::process scales every
term, ::tick divides the sum), both layouts, ragged launches, state carried across launches, clone, reset, another sample rate -- and against
the run-time compiled lane-per-voice rendering of the same graph (two independent device formulations)."""
import numpy as np
import pytest

import oracle as O
from fundsp_amd import LAYOUT_PLANAR, LAYOUT_VOICE_MINOR, MODE_PROCESS, MODE_TICK
from fundsp_amd import graph as GR
from test_gpu_parity import assert_bit_equal

pytestmark = pytest.mark.gpu
SR = 48000.0


def delays_of(n, lo=0.01, hi=0.03):
    """delay(lerp(0.01, 0.03, rnd1(i))) of the doc example (prelude.rs:1334; rnd1: math.rs:569-576), as f32 seconds"""
    r = np.array([O.lib().o_math_rnd1(i) for i in range(n)], dtype=np.float64)
    return [float(np.float32(np.float32(lo) * (np.float32(1.0) - np.float32(x)) + np.float32(hi) * np.float32(x))) for x in r]


def oracle_net(n, delays, w, nin, nout):
    line = O.stacki(n, lambda i: O.delay(delays[i]) >> O.fir(*w))
    head = O.split(n) if nin == 1 else O.multisplit(2, n // 2)
    tail = O.join(n) if nout == 1 else O.multijoin(2, n // 2)
    net = head >> O.fdn(line) >> tail
    net.set_sample_rate(SR)
    return net


def device_graph(n, delays, w, nin, nout):
    line = GR.stacki(n, lambda i: GR.delay(delays[i]) >> GR.fir(*w))
    head = GR.split(n) if nin == 1 else GR.multisplit(2, n // 2)
    tail = GR.join(n) if nout == 1 else GR.multijoin(2, n // 2)
    return head >> GR.fdn(line) >> tail


def run(bank, x, layout, mode, cuts):
    """x: [V][nin][T] -> [V][nout][T], rendered launch by launch"""
    import torch

    parts = []
    for a, e in zip(cuts[:-1], cuts[1:]):
        n = e - a
        if layout == LAYOUT_PLANAR:
            xi = torch.from_numpy(np.ascontiguousarray(x[:, :, a:e])).cuda()
            parts.append(bank.process(n, xi, layout=layout, frame_stride=n, mode=mode).cpu().numpy())
        else:
            xi = torch.from_numpy(np.ascontiguousarray(x[:, :, a:e].transpose(1, 2, 0))).cuda()
            parts.append(bank.process(n, xi, layout=layout, mode=mode).cpu().numpy().transpose(2, 0, 1))
    return np.concatenate(parts, axis=2)


CASES = [  # lines, FIR weights, inputs, outputs
    (16, (0.2, 0.4, 0.2), 1, 1),        # the doc example
    (32, (-0.21, -0.45, -0.2), 2, 2),
    (8, (0.55, 0.4), 2, 1),
    (4, (0.93,), 1, 2),
    (2, (0.45, 0.45), 1, 1),
]


@pytest.mark.parametrize("mode", [MODE_PROCESS, MODE_TICK])
@pytest.mark.parametrize("layout", [LAYOUT_PLANAR, LAYOUT_VOICE_MINOR])
@pytest.mark.parametrize("n,w,nin,nout", CASES)
def test_generic_fdn_matches_oracle(gpu, n, w, nin, nout, layout, mode):
    V, T = 6, 64 * 90 + 13
    rng = np.random.default_rng(100 + n)
    x = (rng.random((V, nin, T), dtype=np.float32) * 2 - 1).astype(np.float32)
    x[:, :, 2 * T // 3:] = 0.0
    x[1] *= 1e-30                                     # an instance that lives in the denormal range (Feedback graphs flush)
    x[2] = 0.0
    x[2, 0, 0] = 1.0                                  # an impulse
    delays = delays_of(n)
    b = gpu.Bank.fdn(V, n, delays, len(w), w, nin, nout)
    b.set_sample_rate(SR)
    assert b.inputs() == nin and b.outputs() == nout
    cuts = [0, 64 * 9, 64 * 9 + 7, 64 * 40 + 7, T]   # a ragged launch in the middle: the next one starts a new block
    got = run(b, x, layout, mode, cuts)
    assert b.get_option("last_kernel") == 6
    for v in range(V):
        net = oracle_net(n, delays, w, nin, nout)
        want = [net.render_blocks(x[v][:, a:e]) if mode == MODE_PROCESS else net.render_ticks(x[v][:, a:e]) for a, e in zip(cuts[:-1], cuts[1:])]
        assert_bit_equal(got[v], np.concatenate(want, axis=1), f"fdn<{n}> fir{len(w)} {nin}->{nout} instance {v}")
    assert np.abs(got[2, :, 2000:]).max() > 1e-6     # the impulse actually recirculated
    # reset: the same render again
    b.reset()
    assert_bit_equal(run(b, x[:, :, :700], layout, mode, [0, 700]), got[:, :, :700], "after reset")


def test_generic_fdn_clone_and_other_sample_rate(gpu):
    import torch

    n, w, V, T = 16, (0.2, 0.4, 0.2), 5, 64 * 30
    delays = delays_of(n)
    rng = np.random.default_rng(7)
    x = (rng.random((V, 1, T), dtype=np.float32) * 2 - 1).astype(np.float32)
    b = gpu.Bank.fdn(V, n, delays, 3, w)
    b.set_sample_rate(SR)
    a1 = run(b, x[:, :, :1000], LAYOUT_PLANAR, MODE_PROCESS, [0, 1000])
    c = b.clone()
    a2 = run(b, x[:, :, 1000:], LAYOUT_PLANAR, MODE_PROCESS, [0, T - 1000])
    c2 = run(c, x[:, :, 1000:], LAYOUT_PLANAR, MODE_PROCESS, [0, T - 1000])
    assert_bit_equal(c2, a2, "the clone continues like the original")
    net = oracle_net(n, delays, w, 1, 1)
    assert_bit_equal(np.concatenate([a1, a2], axis=2)[3], net.render_blocks(x[3]), "chunked == the oracle's one pass")
    # another rate: Delay::set_sample_rate resizes and empties the lines (delay.rs:105-113); the FIR history and the feedback value stay
    # (the oracle graph goes the same way: one second at 48 kHz, then the move)
    b.set_sample_rate(96000.0)
    net = oracle_net(n, delays, w, 1, 1)
    net.render_blocks(x[1])
    net.set_sample_rate(96000.0)
    got = run(b, x, LAYOUT_VOICE_MINOR, MODE_PROCESS, [0, T])
    assert_bit_equal(got[1], net.render_blocks(x[1]), "96 kHz")
    del torch


@pytest.mark.parametrize("n,w,nin,nout", CASES[:3])
def test_from_graph_takes_the_fdn_kernel_and_equals_the_run_time_compiled_graph(gpu, n, w, nin, nout):
    """Bank.from_graph recognises the shape (graph.fdn_plan) and builds the lane-per-frame bank; fdn_kernel=False compiles the same graph
    at run time and renders it one lane per voice.  Two device formulations that share no kernel code: identical samples."""
    import torch

    V, T = 4, 64 * 60 + 5
    delays = delays_of(n)
    g = device_graph(n, delays, w, nin, nout)
    assert GR.fdn_plan(g) is not None
    fast = gpu.Bank.from_graph(g, V, sample_rate=SR)
    slow = gpu.Bank.from_graph(device_graph(n, delays, w, nin, nout), V, sample_rate=SR, fdn_kernel=False)
    assert fast.kind == "fdn" and slow.kind.startswith("jit_")
    rng = np.random.default_rng(n)
    x = (rng.random((V, nin, T), dtype=np.float32) * 2 - 1).astype(np.float32)
    a = run(fast, x, LAYOUT_VOICE_MINOR, MODE_PROCESS, [0, 64 * 20, T])
    b = run(slow, x, LAYOUT_VOICE_MINOR, MODE_PROCESS, [0, 64 * 20, T])
    assert fast.get_option("last_kernel") == 6 and slow.get_option("last_kernel") != 6
    assert_bit_equal(a, b, "lane-per-frame kernel vs run-time compiled lane-per-voice graph")
    del torch


def test_short_delays_and_bad_arguments(gpu):
    with pytest.raises(gpu.FdspError, match="128 samples"):
        gpu.Bank.fdn(2, 4, [0.001] * 4, 1, [0.5])                       # 44 samples at the construction rate
    with pytest.raises(gpu.FdspError, match="lines"):
        gpu.Bank.fdn(2, 12, [0.01] * 12, 1, [0.5])
    with pytest.raises(gpu.FdspError, match="taps"):
        gpu.Bank.fdn(2, 4, [0.01] * 4, 4, [0.5] * 4)
    # from_graph falls back to the run-time compiled graph when a delay is shorter than two blocks
    g = device_graph(4, [0.001, 0.01, 0.012, 0.013], (0.5,), 1, 1)
    b = gpu.Bank.from_graph(g, 3, sample_rate=SR)
    assert b.kind.startswith("jit_")
    # a bank that fits at 48 kHz refuses a rate at which a delay drops under two blocks, and stays as it was
    f = gpu.Bank.fdn(2, 4, [0.004] * 4, 1, [0.5])
    f.set_sample_rate(SR)
    with pytest.raises(gpu.FdspError, match="128 samples"):
        f.set_sample_rate(16000.0)
    assert f.inputs() == 1


@pytest.mark.parametrize("kind", ["fdn8", "reverb_stereo", "reverb4_stereo"])
def test_voice_minor_launches_of_larger_banks_take_the_staging_copy(gpu, kind):
    """Banks of 64 instances or more take voice-minor buffers through a planar staging copy (fd_fdn.hip "voice-minor I/O"): the same samples
    as the planar render of a twin bank and as the oracle, over ragged launches (the staging buffer grows with the longest launch)."""
    V, T = 70, 64 * 12 + 9
    rng = np.random.default_rng(5)
    if kind == "fdn8":
        delays, w = delays_of(8), (0.55, 0.4)
        mk = lambda: gpu.Bank.fdn(V, 8, delays, 2, w, 2, 1)
        net = lambda: oracle_net(8, delays, w, 2, 1)
    elif kind == "reverb_stereo":
        mk = lambda: gpu.Bank.reverb_stereo(V, 10.0, 2.0, 0.5)
        net = lambda: O.reverb_stereo(10.0, 2.0, 0.5)
    else:
        mk = lambda: gpu.Bank.reverb4_stereo(V, 20.0, 2.0)
        net = lambda: O.reverb4_stereo(20.0, 2.0)
    a, b = mk(), mk()
    a.set_sample_rate(SR)
    b.set_sample_rate(SR)
    x = (rng.random((V, a.inputs(), T), dtype=np.float32) * 2 - 1).astype(np.float32)
    cuts = [0, 64 * 2 + 5, 64 * 9 + 5, T]
    vm = run(a, x, LAYOUT_VOICE_MINOR, MODE_PROCESS, cuts)
    pl = run(b, x, LAYOUT_PLANAR, MODE_PROCESS, cuts)
    assert_bit_equal(vm, pl, f"{kind}: voice-minor (staged) == planar")
    for v in (0, 63, 64, V - 1):
        n = net()
        n.set_sample_rate(SR)
        want = np.concatenate([n.render_blocks(x[v][:, s:e]) for s, e in zip(cuts[:-1], cuts[1:])], axis=1)
        assert_bit_equal(vm[v], want, f"{kind} instance {v}")


def test_generic_fdn_at_bench_size(gpu):
    """The bench line's shape (bench.py "fdn16": 4 096 instances of the prelude's 16-line example x 48 000 frames, planar, built with
    Bank.from_graph): spot instances against the oracle over the whole second, and the same second in two launches equals the one launch."""
    import torch

    V, T, n, w = 4096, 48000, 16, (0.2, 0.4, 0.2)
    delays = delays_of(n)
    b = gpu.Bank.from_graph(device_graph(n, delays, w, 1, 1), V, sample_rate=SR)
    assert b.kind == "fdn"
    g = torch.Generator(device="cuda").manual_seed(99)
    x = torch.rand((V, 1, T), dtype=torch.float32, device="cuda", generator=g) * 2 - 1
    one = b.process(T, x, layout=LAYOUT_PLANAR, frame_stride=T)
    b.reset()
    cut = 64 * 301 + 17
    a1 = b.process(cut, x[:, :, :cut].contiguous(), layout=LAYOUT_PLANAR, frame_stride=cut)
    a2 = b.process(T - cut, x[:, :, cut:].contiguous(), layout=LAYOUT_PLANAR, frame_stride=T - cut)
    assert torch.equal(torch.cat([a1, a2], dim=2).view(torch.int32), one.view(torch.int32)), "two launches == one"
    for v in (0, 1, 2047, 2048, V - 1):
        net = oracle_net(n, delays, w, 1, 1)
        assert_bit_equal(one[v].cpu().numpy(), net.render_blocks(x[v].cpu().numpy()), f"instance {v} of {V}, {T} frames")


@pytest.mark.parametrize("kind", ["fdn16", "reverb_stereo", "reverb4_stereo"])
def test_a_sample_rate_change_in_mid_tail_keeps_what_the_reference_keeps(gpu, kind):
    """A change of rate empties the delay lines (Delay::set_sample_rate, delay.rs:105-113) and nothing else: the FIRs keep their history
    (fir.rs:52-54), Feedback keeps its value (feedback.rs:125-127).  A bank moved from 48 kHz to 44.1 kHz while its tail sounds continues
    like the oracle graph moved the same way (round 6; before, the move cleared those too -- right only for a silent bank)."""
    V, T1, T2 = 3, 64 * 120 + 7, 64 * 130
    rng = np.random.default_rng(77)
    if kind == "fdn16":
        delays, w = delays_of(16), (0.2, 0.4, 0.2)
        b = gpu.Bank.fdn(V, 16, delays, 3, w)
        net = lambda: oracle_net(16, delays, w, 1, 1)
    elif kind == "reverb_stereo":
        b = gpu.Bank.reverb_stereo(V, 10.0, 2.0, 0.5)
        net = lambda: O.reverb_stereo(10.0, 2.0, 0.5)
    else:
        b = gpu.Bank.reverb4_stereo(V, 20.0, 2.0)
        net = lambda: O.reverb4_stereo(20.0, 2.0)
    b.set_sample_rate(SR)
    x = (rng.random((V, b.inputs(), T1 + T2), dtype=np.float32) * 2 - 1).astype(np.float32)
    x[:, :, T1:] = 0.0
    run(b, x[:, :, :T1], LAYOUT_PLANAR, MODE_PROCESS, [0, T1])
    b.set_sample_rate(44100.0)
    got = run(b, x[:, :, T1:], LAYOUT_PLANAR, MODE_PROCESS, [0, 64 * 2 + 9, T2])
    for v in range(V):
        n = net()
        n.set_sample_rate(SR)
        n.render_blocks(x[v][:, :T1])
        n.set_sample_rate(44100.0)
        assert_bit_equal(got[v], n.render_blocks(x[v][:, T1:]), f"{kind} instance {v} after the move")
    assert np.abs(got[:, :, :4]).max() > 1e-4       # the FIR history and the feedback value sound at once; the lines themselves start empty

"""Stream semantics of the C ABI: fdsp_bank_process on a caller's (non-blocking) stream is ordered against the bank's own
lifecycle / parameter work in both directions, and block-by-block rendering can be captured into a HIP graph and replayed
(the real-time pattern: one AudioNode::process block per launch)."""
import numpy as np
import pytest

from fundsp_amd import workloads as W

pytestmark = pytest.mark.gpu
SR = 48000.0


def ints(t):
    import torch

    return t.view(torch.int32)


def test_reset_and_params_do_not_overtake_a_render_on_a_caller_stream(gpu):
    import torch

    V = 16384
    p = W.fm_svf_params(V, SR)
    want = W.make_fm_svf_bank(V, SR, params=p).process(512)
    b = W.make_fm_svf_bank(V, SR, params=p)
    s = torch.cuda.Stream()                       # non-blocking: no implicit ordering with the bank's stream
    with torch.cuda.stream(s):
        for _ in range(20):
            b.process(4096)                       # still running when reset() is issued on the bank's stream
            b.reset()
            b.set_seed(p["seed"])
            got = b.process(512)
            assert torch.equal(ints(got), ints(want))
        # a state snapshot taken right after a render on the side stream sees that render's final state
        b.process(2048)
        snap = b.get_state()
        nxt = b.process(256).clone()
        b.set_state(snap)
        assert torch.equal(ints(b.process(256)), ints(nxt))
    torch.cuda.synchronize()


def test_block_launches_captured_into_a_hip_graph(gpu):
    import torch

    V, NB = 4096, 8
    p = W.fm_svf_params(V, SR)
    want = W.make_fm_svf_bank(V, SR, params=p).process(64 * NB * 3)
    b = W.make_fm_svf_bank(V, SR, params=p)
    outs = [torch.empty((1, 64, V), dtype=torch.float32, device="cuda") for _ in range(NB)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        b.process(64, out=outs[0])                # load the kernels outside the capture
        b.reset()
        b.set_seed(p["seed"])
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for k in range(NB):
                b.process(64, out=outs[k])        # recorded, not run: no host synchronisation inside the capture
        chunks = []
        for _ in range(3):                        # each replay continues from the state the previous one left
            g.replay()
            chunks.append(torch.cat(outs, dim=1).clone())
        torch.cuda.synchronize()
    got = torch.cat(chunks, dim=1)
    assert torch.equal(ints(got), ints(want))


def test_destroying_a_bank_waits_for_its_render_on_a_caller_stream(gpu):
    """fdsp_bank_destroy right after an asynchronous render on a caller's stream: the output is complete and correct
    (the teardown waits for the render's completion event before freeing the slots the kernel is reading)."""
    import torch

    V = 16384
    p = W.fm_svf_params(V, SR)
    want = W.make_fm_svf_bank(V, SR, params=p).process(4096)
    s = torch.cuda.Stream()
    for _ in range(5):
        b = W.make_fm_svf_bank(V, SR, params=p)
        out = torch.empty((1, 4096, V), dtype=torch.float32, device="cuda")
        with torch.cuda.stream(s):
            b.process(4096, out=out)              # asynchronous on s
        b.close()                                 # fdsp_bank_destroy while the kernel may still be running
        junk = [W.make_fm_svf_bank(V, SR, params=p) for _ in range(2)]   # re-use the freed slot memory at once
        s.synchronize()
        assert torch.equal(ints(out), ints(want))
        del junk


def test_device_side_fill_and_stream_capture(gpu):
    """fdsp_bank_set_param_all fills on the device and does not wait; a capture of renders on a caller's stream cannot wait either, so it is
    refused while such a fill is still queued (the captured graph could otherwise replay before the fill lands) and accepted after
    fdsp_bank_synchronize -- and then replays the variable's value of capture time block after block."""
    import torch

    gpu.wavetable_build("saw")
    V, NB = 64 * 20, 4
    p = W.saw_moog_params(V, SR)
    adsr = (0.005, 0.01, 0.6, 0.01)
    ref = W.make_saw_moog_var_bank(V, SR, params=p, adsr=adsr)
    ref.set_param(W.C4V_SLOTS["gate"], 1.0)
    want = ref.process(64 * NB * 2)
    b = W.make_saw_moog_var_bank(V, SR, params=p, adsr=adsr)
    outs = [torch.empty((2, 64, V), dtype=torch.float32, device="cuda") for _ in range(NB)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        c = b.clone()
        c.process(64, out=outs[0])                # load the kernels outside the capture (on a clone: b's state stays)
        torch.cuda.synchronize()
        b.set_param(W.C4V_SLOTS["gate"], 1.0)    # queued on the bank's stream, not waited for
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            with pytest.raises(gpu.FdspError, match="fdsp_bank_synchronize"):   # the refusal names its remedy
                b.process(64, out=outs[0])
        del g
        b.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for k in range(NB):
                b.process(64, out=outs[k])
        chunks = []
        for _ in range(2):
            g.replay()
            chunks.append(torch.cat(outs, dim=1).clone())
        torch.cuda.synchronize()
    assert torch.equal(ints(torch.cat(chunks, dim=1)), ints(want))


def test_run_time_compiled_small_bank_captures_its_first_launch(gpu):
    """A run-time compiled generator chain on a small bank takes the time-split kernels, which live in the kind's second module: that module is
    built when the BANK is created (KindOps::prepare_render), so even the bank's very first launches can be recorded into a HIP graph."""
    import torch
    from fundsp_amd import graph as GR

    V, NB = 64 * 6, 4
    p = W.fm_svf_params(V, SR)
    g = GR.sine_hz(p["f"]) * p["f"] * p["m"] + p["f"] >> GR.sine() >> GR.lowpass_hz(p["fc"], p["q"])
    want = W.make_fm_svf_bank(V, SR, params=p).process(64 * NB)
    b = gpu.Bank.from_graph(g, V, sample_rate=SR)
    b.set_seed(p["seed"])
    outs = [torch.empty((1, 64, V), dtype=torch.float32, device="cuda") for _ in range(NB)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for k in range(NB):
                b.process(64, out=outs[k])        # the first launches this bank ever sees
        assert b.get_option("last_kernel") == 4
        gr.replay()
        torch.cuda.synchronize()
    assert torch.equal(ints(torch.cat(outs, dim=1)), ints(want))


@pytest.mark.parametrize("how", ["switched", "default"])
def test_run_time_compiled_fast_small_bank_captures_its_first_launch(gpu, how):
    """The same for the tolerance mode: FastOf<G> is a module of its own (and its time-split kernels another), built when a FAST bank of the kind
    is created (process-wide default) or when fdsp_bank_set_option switches a bank to FAST -- not inside the bank's first render, which
    a host may be capturing.  A graph nobody has compiled before, so that a lazy compile could not hide in a cache."""
    import torch
    import fundsp_amd as F
    from fundsp_amd import graph as GR

    V, NB = 64 * 7, 3
    p = W.fm_svf_params(V, SR)
    mk = lambda: GR.sine_hz(p["f"]) * p["f"] * (p["m"] + (0.25 if how == "switched" else 0.5)) + p["f"] >> GR.sine() >> GR.lowpass_hz(p["fc"], p["q"])
    if how == "default":
        assert gpu.lib().fdsp_set_option(b"math", F.MATH_FAST) == 0
    try:
        b = gpu.Bank.from_graph(mk(), V, sample_rate=SR)
        r = gpu.Bank.from_graph(mk(), V, sample_rate=SR)
    finally:
        gpu.lib().fdsp_set_option(b"math", F.MATH_EXACT)
    if how == "switched":
        b.set_option("math", F.MATH_FAST)
        r.set_option("math", F.MATH_FAST)
    assert b.get_option("math") == F.MATH_FAST and b.get_option("math_has_fast_variant") == 1
    b.set_seed(p["seed"])
    r.set_seed(p["seed"])
    outs = [torch.empty((1, 64, V), dtype=torch.float32, device="cuda") for _ in range(NB)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for k in range(NB):
                b.process(64, out=outs[k])        # the first launches this bank ever sees
        assert b.get_option("last_kernel") == 4
        gr.replay()
        torch.cuda.synchronize()
    want = r.process(64 * NB)
    assert torch.equal(ints(torch.cat(outs, dim=1)), ints(want))
    exact = gpu.Bank.from_graph(mk(), V, sample_rate=SR)
    exact.set_seed(p["seed"])
    assert not torch.equal(ints(exact.process(64 * NB)), ints(want)), "the FAST bank must not have rendered exactly"


@pytest.mark.parametrize("layout_name", ["voice_minor_staged", "planar"])
def test_fdn_bank_block_launches_captured_into_a_hip_graph(gpu, layout_name):
    """The lane-per-frame FDN banks (here the generic network, fdsp_fdn_create) under stream capture: block launches recorded once and
    replayed equal one uncaptured render -- in the planar layout, and with voice-minor buffers through the bank's planar staging copy
    (64 instances or more; sized by a launch before the capture, like the partial-mix buffer)."""
    import torch
    import fundsp_amd as F

    V, NB = 80, 5
    delays = [0.004 + 0.0007 * i for i in range(8)]
    mk = lambda: gpu.Bank.fdn(V, 8, delays, 3, [0.2, 0.45, 0.2], 2, 2)
    planar = layout_name == "planar"
    b, r = mk(), mk()
    b.set_sample_rate(SR)
    r.set_sample_rate(SR)
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.rand((V, 2, 64 * NB * 2) if planar else (2, 64 * NB * 2, V), device="cuda", generator=g) * 2 - 1
    kw = dict(layout=F.LAYOUT_PLANAR, frame_stride=64) if planar else dict(layout=F.LAYOUT_VOICE_MINOR)
    want = r.process(64 * NB * 2, x, **(dict(layout=F.LAYOUT_PLANAR, frame_stride=64 * NB * 2) if planar else kw))
    ins = [torch.empty((V, 2, 64) if planar else (2, 64, V), device="cuda") for _ in range(NB)]
    outs = [torch.empty_like(t) for t in ins]
    s = torch.cuda.Stream()
    chunks = []
    with torch.cuda.stream(s):
        ins[0].zero_()
        b.process(64, ins[0], out=outs[0], **kw)        # loads the kernels and sizes the staging buffer before the capture ...
        b.reset()                                        # ... and leaves no trace: silence through zeroed rings, then reset
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for k in range(NB):
                b.process(64, ins[k], out=outs[k], **kw)
        for rep in range(2):
            for k in range(NB):
                sl = slice((rep * NB + k) * 64, (rep * NB + k + 1) * 64)
                ins[k].copy_(x[:, :, sl] if planar else x[:, sl, :])
            gr.replay()
            chunks.append(torch.cat(outs, dim=2 if planar else 1).clone())
        torch.cuda.synchronize()
    got = torch.cat(chunks, dim=2 if planar else 1)
    assert torch.equal(ints(got), ints(want))

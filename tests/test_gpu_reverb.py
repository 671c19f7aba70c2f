"""GPU parity for BASELINE config 5: reverb_stereo (32-line FDN, prelude.rs:1732-1762) -- the lane-per-frame kernel
(default) and the lane-per-delay-line kernel vs the oracle's per-sample restatement.  Bit-exact, including the ordered
32-term pan sum and the Hadamard stages."""
import numpy as np
import pytest

import oracle as O
from fundsp_amd import LAYOUT_PLANAR, LAYOUT_VOICE_MINOR, MODE_PROCESS, MODE_TICK
from test_gpu_parity import assert_bit_equal

pytestmark = pytest.mark.gpu
SR = 48000.0


def render(bank, x, layout):
    """x: [V][2][T] -> [V][2][T]"""
    import torch

    V, _, T = x.shape
    if layout == LAYOUT_PLANAR:
        out = bank.process(T, torch.from_numpy(x).cuda(), layout=LAYOUT_PLANAR, frame_stride=T)
        torch.cuda.synchronize()
        return out.cpu().numpy()
    inp = torch.from_numpy(np.ascontiguousarray(x.transpose(1, 2, 0))).cuda()
    out = bank.process(T, inp, layout=LAYOUT_VOICE_MINOR)
    torch.cuda.synchronize()
    return out.cpu().numpy().transpose(2, 0, 1)


@pytest.fixture(params=[0, 1], ids=["lane_per_frame", "lane_per_line"])
def fdn_kernel(gpu, request):
    assert gpu.lib().fdsp_set_option(b"fdn_kernel", request.param) == 0
    yield request.param
    gpu.lib().fdsp_set_option(b"fdn_kernel", 0)


@pytest.mark.parametrize("layout", [LAYOUT_PLANAR, LAYOUT_VOICE_MINOR])
def test_reverb_stereo_matches_oracle(gpu, layout, fdn_kernel):
    V, T = 5, 64 * 150 + 17  # > 2 trips around the longest line (3980 samples); odd V: half-empty last wave
    rng = np.random.default_rng(12)
    x = (rng.random((V, 2, T), dtype=np.float32) * 2 - 1).astype(np.float32)
    x[1, :, 200:] = 0.0      # an impulse-like burst that decays
    x[2] = 0.0
    x[2, 0, 0] = 1.0         # pure impulse on the left channel
    b = gpu.Bank.reverb_stereo(V, 10.0, 2.0, 0.5)
    b.set_sample_rate(SR)
    assert b.inputs() == 2 and b.outputs() == 2
    got = render(b, x, layout)
    for v in range(V):
        n = O.reverb_stereo(10.0, 2.0, 0.5)
        n.set_sample_rate(SR)
        assert_bit_equal(got[v], n.render_blocks(x[v]), f"reverb instance {v}")
    assert np.abs(got[2, :, 4000:]).max() > 1e-4  # the impulse actually recirculated


def test_reverb_chunked_calls_reset_and_other_rooms(gpu, fdn_kernel):
    V, T = 3, 64 * 70
    rng = np.random.default_rng(13)
    x = (rng.random((V, 2, T), dtype=np.float32) * 2 - 1).astype(np.float32)
    b = gpu.Bank.reverb_stereo(V, 20.0, 5.0, 0.2)
    b.set_sample_rate(SR)
    one = render(b, x, LAYOUT_PLANAR)
    b.reset()
    a1 = render(b, np.ascontiguousarray(x[:, :, :1000]), LAYOUT_PLANAR)      # ragged chunk: 1000 = 15*64 + 40
    a2 = render(b, np.ascontiguousarray(x[:, :, 1000:]), LAYOUT_PLANAR)
    assert_bit_equal(np.concatenate([a1, a2], axis=2), one, "chunked == whole")
    n = O.reverb_stereo(20.0, 5.0, 0.2)
    n.set_sample_rate(SR)
    assert_bit_equal(one[1], n.render_blocks(x[1]), "room 20 m")


def test_reverb_rejects_tiny_delays(gpu):
    with pytest.raises(gpu.FdspError):
        gpu.Bank.reverb_stereo(1, 0.1, 2.0, 0.5)  # 10 cm room: delays shorter than a 64-sample block


@pytest.mark.parametrize("mode", [MODE_PROCESS, MODE_TICK])
@pytest.mark.parametrize("layout", [LAYOUT_PLANAR, LAYOUT_VOICE_MINOR])
def test_reverb4_stereo_bank_matches_oracle(gpu, layout, mode):
    """reverb4_stereo(room_size, time) (prelude.rs:1873-1941) -- two 16-line Hadamard FDNs in series with a MultiJoin / MultiSplit between
    them -- through the lane-per-frame FDN kernel (fdsp_reverb4_stereo_create): bit-exact against the oracle's generic Feedback graph in
    both executors (MultiJoin::process scales every term, ::tick divides the sum: audionode.rs:697-720), ragged launch lengths, a silent
    tail (the feedback decays into the flush-to-zero range), state carried across launches, clone."""
    import torch

    V, T = 6, 64 * 200 + 13                                       # (the first sound leaves the second network after >= 94 ms = 4 500 frames)
    rng = np.random.default_rng(17)
    x = (rng.random((V, 2, T), dtype=np.float32) * 2 - 1).astype(np.float32)
    x[:, :, 2 * T // 3:] = 0.0
    x[1] *= 1e-30                                                   # an instance that lives in the denormal range from the start
    b = gpu.Bank.reverb4_stereo(V, 20.0, 2.0)
    b.set_sample_rate(SR)
    assert b.inputs() == 2 and b.outputs() == 2
    cuts = [0, 64 * 20, 64 * 20 + 7, 64 * 150 + 7, T]              # a ragged launch in the middle: the next one starts a new block
    parts = []
    for a, e in zip(cuts[:-1], cuts[1:]):
        n = e - a
        if layout == LAYOUT_PLANAR:
            xi = torch.from_numpy(np.ascontiguousarray(x[:, :, a:e])).cuda()
            parts.append(b.process(n, xi, layout=layout, frame_stride=n, mode=mode).cpu().numpy())
        else:
            xi = torch.from_numpy(np.ascontiguousarray(x[:, :, a:e].transpose(1, 2, 0))).cuda()
            parts.append(b.process(n, xi, layout=layout, mode=mode).cpu().numpy().transpose(2, 0, 1))
        if a == 0:
            c = b.clone()
    got = np.concatenate(parts, axis=2)
    assert b.get_option("last_kernel") == 6
    for v in range(V):
        n = O.reverb4_stereo(20.0, 2.0)
        n.set_sample_rate(SR)
        want = []
        for a, e in zip(cuts[:-1], cuts[1:]):
            want.append(n.render_blocks(x[v][:, a:e]) if mode == MODE_PROCESS else n.render_ticks(x[v][:, a:e]))
        assert_bit_equal(got[v], np.concatenate(want, axis=1), f"reverb4_stereo instance {v}")
    assert np.abs(got[0]).max() > 0.01 and np.abs(got[:, :, :4000]).max() == 0.0
    # the clone taken after the first launch continues exactly like the original
    a, e = cuts[1], cuts[2]
    xi = torch.from_numpy(np.ascontiguousarray(x[:, :, a:e])).cuda()
    cc = c.process(e - a, xi, layout=LAYOUT_PLANAR, frame_stride=e - a, mode=mode).cpu().numpy()
    assert_bit_equal(cc, got[:, :, a:e], "clone")


def test_reverb4_dedicated_kernel_equals_the_run_time_compiled_graph(gpu):
    """Two independent device formulations of reverb4_stereo(20, 2): the lane-per-frame FDN kernel (fdsp_reverb4_stereo_create: one wave per
    instance, the 32 lines in registers) and the generic run-time compiled graph (Feedback nodes, lane per voice, rings in HBM) -- 300 instances,
    12 800 frames, bit for bit, no oracle involved."""
    import torch
    from fundsp_amd import graph as GR

    V, T = 300, 64 * 200
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.rand((V, 2, T), dtype=torch.float32, device="cuda", generator=g) * 2 - 1
    a = gpu.Bank.reverb4_stereo(V, 20.0, 2.0)
    a.set_sample_rate(SR)
    ya = a.process(T, x, layout=LAYOUT_PLANAR, frame_stride=T)
    b = gpu.Bank.from_graph(GR.reverb4_stereo(20.0, 2.0), V, sample_rate=SR, fdn_kernel=False)
    assert b.kind.startswith("jit_") and gpu.Bank.from_graph(GR.reverb4_stereo(20.0, 2.0), 2, sample_rate=SR).kind == "reverb4_stereo"
    yb = b.process(T, x, layout=LAYOUT_PLANAR, frame_stride=T)
    torch.cuda.synchronize()
    assert float(ya.abs().max()) > 0.05
    assert torch.equal(ya.view(torch.int32), yb.view(torch.int32))


def test_reverb4_stereo_sample_rate_change_and_reset(gpu):
    """Delay::set_sample_rate resizes and resets the lines when the rate changes (delay.rs:105-113): a reverb4_stereo bank moved from 48 kHz to
    44.1 kHz renders what a fresh oracle graph at 44.1 kHz renders; reset() after some audio gives the same again; a too small room * rate is refused."""
    import torch

    V, T = 3, 64 * 150 + 5
    rng = np.random.default_rng(23)
    x = (rng.random((V, 2, T), dtype=np.float32) * 2 - 1).astype(np.float32)
    b = gpu.Bank.reverb4_stereo(V, 25.0, 1.5)
    b.set_sample_rate(SR)
    b.process(64 * 10, torch.from_numpy(np.ascontiguousarray(x[:, :, :640])).cuda(), layout=LAYOUT_PLANAR, frame_stride=640)
    b.set_sample_rate(44100.0)
    got = b.process(T, torch.from_numpy(x).cuda(), layout=LAYOUT_PLANAR, frame_stride=T).cpu().numpy()
    b.reset()
    again = b.process(T, torch.from_numpy(x).cuda(), layout=LAYOUT_PLANAR, frame_stride=T).cpu().numpy()
    for v in range(V):
        n = O.reverb4_stereo(25.0, 1.5)
        n.set_sample_rate(44100.0)
        assert_bit_equal(got[v], n.render_blocks(x[v]), f"reverb4_stereo at 44.1 kHz, instance {v}")
    assert_bit_equal(again, got, "after reset()")
    assert np.abs(got).max() > 0.01
    tiny = gpu.Bank.reverb4_stereo(2, 15.0, 2.0)
    with pytest.raises(gpu.FdspError):
        tiny.set_sample_rate(2000.0)          # the shortest line would be 95 samples: not longer than two blocks
    tiny.set_sample_rate(SR)                  # the bank keeps working at a rate that fits


def test_from_graph_takes_the_dedicated_kernels_for_the_stock_reverbs(gpu):
    """graph.reverb_stereo(..) / graph.reverb4_stereo(..) ARE the stock reverbs: Bank.from_graph builds their lane-per-frame banks (as it does for the
    generic fdn network and reverb3_stereo); fdn_kernel=False compiles the same graph at run time.  reverb_stereo(10, 2, 0.5) -- BASELINE
    config 5 spelled as a graph -- both ways, bit for bit; inside a larger graph the marker is gone and the run-time compiler renders it."""
    import torch
    from fundsp_amd import graph as GR

    V, T = 5, 64 * 140 + 3
    fast = gpu.Bank.from_graph(GR.reverb_stereo(10.0, 2.0, 0.5), V, sample_rate=SR)
    slow = gpu.Bank.from_graph(GR.reverb_stereo(10.0, 2.0, 0.5), V, ring_frames=4096, sample_rate=SR, fdn_kernel=False)
    assert fast.kind == "reverb_stereo" and slow.kind.startswith("jit_")
    rng = np.random.default_rng(3)
    x = (rng.random((V, 2, T), dtype=np.float32) * 2 - 1).astype(np.float32)
    x[:, :, T // 2:] = 0.0
    a, b = render(fast, x, LAYOUT_VOICE_MINOR), render(slow, x, LAYOUT_VOICE_MINOR)
    assert fast.get_option("last_kernel") == 6 and slow.get_option("last_kernel") != 6
    assert_bit_equal(a, b, "reverb_stereo: dedicated kernel vs run-time compiled graph")
    n = O.reverb_stereo(10.0, 2.0, 0.5)
    n.set_sample_rate(SR)
    assert_bit_equal(a[4], n.render_blocks(x[4]), "... and the oracle")
    wrapped = (GR.pass_() | GR.pass_()) >> GR.reverb_stereo(10.0, 2.0, 0.5)
    assert getattr(wrapped, "stock_reverb", None) is None
    del torch

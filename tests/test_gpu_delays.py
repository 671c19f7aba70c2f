"""GPU parity for the delay-line row: Delay, Tap (cubic), TapLinear, AllNest -- rings in HBM laid out
[ring][position][voice].  Bit-exact vs the oracle, including the reference's exact identity tick>>tick>>tick == delay(3)."""
import numpy as np
import pytest

import oracle as O
from fundsp_amd import LAYOUT_PLANAR, LAYOUT_VOICE_MINOR, MODE_PROCESS, MODE_TICK
from test_gpu_parity import assert_bit_equal, noise_input, oracle_render, run_bank

pytestmark = pytest.mark.gpu
SR = 48000.0


@pytest.mark.parametrize("layout", [LAYOUT_VOICE_MINOR, LAYOUT_PLANAR])
def test_delay(gpu, layout):
    V, T = 70, 700
    times = np.concatenate([[0.0, 1.0 / SR, 3.0 / SR], np.linspace(0.0002, 0.01, V - 3)]).astype(np.float32)
    b = gpu.Bank("delay", V, ring_frames=512)
    b.set_param(":time", times)
    b.set_sample_rate(SR)
    x = noise_input(V, 1, T, seed=51)
    got = run_bank(b, x, T, layout, MODE_PROCESS)
    for v in range(V):
        n = O.delay(float(times[v]))
        n.set_sample_rate(SR)
        assert_bit_equal(got[v], n.render_blocks(x[v]), f"delay voice {v}")
    # exact structural identity of the reference (tests/test_basic.rs:520-529): delay(3 samples) == three ticks
    assert_bit_equal(got[2, 0, 3:], x[2, 0, :-3], "delay(3/sr) shifts by exactly 3")
    assert_bit_equal(got[0], x[0], "delay(0) is a pass-through (delay.rs:116-124 with a length-1 ring)")
    # reset clears the line
    b.reset()
    got2 = run_bank(b, x, T, layout, MODE_PROCESS)
    assert_bit_equal(got2, got, "reset determinism")


@pytest.mark.parametrize("linear", [False, True])
@pytest.mark.parametrize("mode", [MODE_PROCESS, MODE_TICK])
def test_taps(gpu, linear, mode):
    V, T = 64, 64 * 6 + 5
    rng = np.random.default_rng(52)
    max_d = 0.01
    b = gpu.Bank("tap_linear" if linear else "tap", V, ring_frames=1024)
    b.set_param(":min_delay", 0.0 if linear else 0.0005)
    b.set_param(":max_delay", max_d)
    b.set_sample_rate(SR)
    x = noise_input(V, 2, T, seed=53)
    x[:, 1, :] = (0.0002 + 0.012 * rng.random((V, 1))).astype(np.float32)          # constant per voice (some beyond max)
    x[1, 1, :] = 0.004 + 0.003 * np.sin(np.arange(T) / 9.0)                          # modulated delay (chorus-like)
    x[2, 1, :] = 0.0                                                                  # below min: clamped
    got = run_bank(b, x, T, LAYOUT_VOICE_MINOR, mode)
    for v in range(0, V, 3):
        n = O.tap_linear(0.0, max_d) if linear else O.tap(0.0005, max_d)
        n.set_sample_rate(SR)
        assert_bit_equal(got[v], oracle_render(n, x[v], T, mode), f"tap linear={linear} voice {v}")


@pytest.mark.parametrize("inner", ["delay", "tick", "pass"])
def test_allnest(gpu, inner):
    V, T = 64, 900
    b = gpu.Bank(f"allnest_{inner}", V, ring_frames=256) if inner == "delay" else gpu.Bank(f"allnest_{inner}", V)
    eta = np.linspace(-0.8, 0.8, V).astype(np.float32)
    b.set_param(":coefficient", eta)
    if inner == "delay":
        b.set_param("0:time", 0.003)
    b.set_sample_rate(SR)
    x = noise_input(V, 1, T, seed=54)
    got = run_bank(b, x, T, LAYOUT_PLANAR, MODE_PROCESS)
    for v in range(0, V, 7):
        child = O.delay(0.003) if inner == "delay" else O.tick() if inner == "tick" else O.pass_()
        n = O.allnest_c(float(eta[v]), child)
        n.set_sample_rate(SR)
        assert_bit_equal(got[v], n.render_blocks(x[v]), f"allnest {inner} voice {v}")
    # allpass property of the nested allpass (tests/test_flow.rs:251-283), on the GPU output itself
    imp = np.zeros((V, 1, 4096), dtype=np.float32)
    imp[:, 0, 0] = 1.0
    b.reset()
    h = run_bank(b, imp, 4096, LAYOUT_PLANAR, MODE_PROCESS)
    if inner != "delay":
        mag = np.abs(np.fft.rfft(h[5, 0].astype(np.float64)))
        assert np.all(np.abs(mag - 1.0) < 1e-4)


def test_ring_kind_needs_capacity(gpu):
    with pytest.raises(gpu.FdspError):
        gpu.Bank("delay", 8)


def test_ring_capacity_too_small_is_an_error_not_a_shorter_delay(gpu):
    """The reference resizes a Delay / Tap / Limiter buffer when time * sample_rate grows (delay.rs:105-113); a bank's
    rings are fixed at creation, so asking for more than `ring_frames` must FAIL the next render loudly -- and stop
    failing once the parameters fit again."""
    V, T = 64, 64
    b = gpu.Bank("delay", V, ring_frames=256)
    b.set_param(":time", 0.004)          # 192 + 1 positions at 48 kHz: fits
    b.set_sample_rate(SR)
    x = noise_input(V, 1, T, seed=5)
    run_bank(b, x, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    b.set_sample_rate(96000.0)           # 384 + 1 positions: does not fit in 256
    with pytest.raises(gpu.FdspError, match="ring capacity too small.*385"):
        run_bank(b, x, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    with pytest.raises(gpu.FdspError, match="ring capacity too small"):      # stays an error until fixed
        run_bank(b, x, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    b.set_sample_rate(SR)
    run_bank(b, x, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    one = np.full(V, 0.004, dtype=np.float32)
    one[17] = 0.02                       # a single voice asks for 961 positions
    b.set_param(":time", one)
    with pytest.raises(gpu.FdspError, match="961"):
        run_bank(b, x, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    # corrected by PARTIAL-range updates only: the complaint must not survive the fix ...
    b.set_param(":time", np.full(1, 0.004, dtype=np.float32), first=17)
    run_bank(b, x, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    # ... and a complaint of voices OUTSIDE the last updated range must not be lost
    b.set_param(":time", np.full(1, 0.02, dtype=np.float32), first=40)
    b.set_param(":time", np.full(2, 0.003, dtype=np.float32), first=3)
    with pytest.raises(gpu.FdspError, match="961"):
        run_bank(b, x, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)


def test_from_graph_sizes_the_rings_itself(gpu):
    """Bank.from_graph without ring_frames: Graph.ring_frames derives the capacity from the delay / tap / limiter
    parameters at the construction rate and the target rate; the render equals the oracle."""
    from fundsp_amd import graph as GR

    build = lambda m: m.delay(0.0105) >> m.allnest_c(0.4, m.delay(0.003)) >> (m.pass_() * 2.0 >> m.limiter(0.004, 0.03))
    g = build(GR)
    assert g.ring_frames(48000.0) == max(505, 256 + 192 + 0)   # delay 0.0105 s -> 504 + 1; limiter 0.004 s -> 256 + 192
    assert g.ring_frames(96000.0) == 1009
    V, T = 66, 64 * 4 + 7
    b = gpu.Bank.from_graph(g, V, sample_rate=SR)
    x = noise_input(V, 1, T, seed=8)
    got = run_bank(b, x, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    n = build(O)
    n.set_sample_rate(SR)
    assert_bit_equal(got[3], oracle_render(n, x[3], T, MODE_PROCESS), "auto-sized rings")

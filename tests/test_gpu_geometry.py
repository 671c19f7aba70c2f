"""Launch geometry must not show in the samples.  The pipeline kernel picks the voice groups per workgroup (1 / 2 / 4), the
hand-over tile length that goes with it, and -- for two groups -- an order of roles over the SIMDs from the size of
the bank; small banks of heavy graphs also change the tile plan.  Every ahead-of-time graph kind is rendered in banks
that land in each regime, with per-voice seeds and inputs, and the first 192 voices must be bit-identical to a 192-voice
bank of the same voices (whose kernels the oracle parity tests cover)."""
import numpy as np
import pytest

from fundsp_amd import LAYOUT_PLANAR, LAYOUT_VOICE_MINOR, MODE_PROCESS, MODE_TICK
from test_gpu_parity import assert_bit_equal, noise_input

pytestmark = pytest.mark.gpu
SR = 48000.0
GRAPH_KINDS = ["sine_hz", "sine_hz_lowpass_hz", "noise_biquad", "fm_svf", "saw_moog_adsr_pan", "oversample_fm",
               "oversample_shape", "resample_fm", "svf_shape", "svf_shape_svf"]
SMALL = 192


def render(gpu, kind, V, T, x, layout, mode):
    import torch

    b = gpu.Bank(kind, V)
    b.set_sample_rate(SR)
    b.set_seed(np.arange(V, dtype=np.uint64) * 3 + 11)
    inp = None
    if b.inputs():
        xs = x[:V]
        inp = torch.from_numpy(np.ascontiguousarray(xs.transpose(1, 2, 0)) if layout == LAYOUT_VOICE_MINOR else xs).cuda()
    out = b.process(T, inp, layout=layout, mode=mode, **({"frame_stride": T} if layout == LAYOUT_PLANAR else {}))
    torch.cuda.synchronize()
    o = out[:, :, :SMALL].permute(2, 0, 1) if layout == LAYOUT_VOICE_MINOR else out[:SMALL]
    return o.contiguous().cpu().numpy()


@pytest.mark.parametrize("kind", GRAPH_KINDS)
def test_samples_do_not_depend_on_bank_size(gpu, kind):
    if kind == "saw_moog_adsr_pan":
        gpu.wavetable_build("saw")
    T = 64 * 6                                           # planar pipeline wants 16-byte rows: a multiple of 4
    ni = gpu.Bank(kind, 1).inputs()
    sizes = (SMALL, 10048, 20032, 40000, 70016)          # <= CUs, <= 2 CUs, <= 4 CUs groups; beyond 4 CUs groups
    x = noise_input(max(sizes), max(ni, 1), T, seed=3) if ni else None
    if x is not None and kind == "saw_moog_adsr_pan":    # a gate: low, high, low
        x[:, 0, :] = 0.0
        x[:, 0, 3:250] = 1.0
    for layout, mode in ((LAYOUT_VOICE_MINOR, MODE_PROCESS), (LAYOUT_PLANAR, MODE_PROCESS), (LAYOUT_VOICE_MINOR, MODE_TICK)):
        want = render(gpu, kind, SMALL, T, x, layout, mode)
        for V in sizes[1:]:
            with np.errstate(all="ignore"):
                assert_bit_equal(render(gpu, kind, V, T, x, layout, mode), want, f"{kind} V={V} layout={layout} mode={mode}")


@pytest.mark.parametrize("kind", GRAPH_KINDS + ["saw_moog_var_adsr_pan", "sine", "fixed_svf", "biquad", "moog_hz", "noise"])
def test_every_dispatcher_route_of_an_ahead_of_time_kind_gives_the_same_samples(gpu, kind):
    """The routes a launch can take -- "pipe_split" 0 (single-wave kernels), 1 (the default choice), 2 (pipelines forced) x voice-minor, planar rows
    with 16-byte runs, planar rows of an odd tight length (the single-wave planar kernel whatever the option says) -- for the ahead-of-time kinds:
    all bit-identical per executor.  (The run-time compiled kinds have the same check in tests/test_gpu_graph_fuzz.py, where a tight planar row
    first exposed a miscompiled kernel variant.)"""
    import torch

    if kind not in gpu.kinds():
        pytest.skip(f"no ahead-of-time kind {kind}")
    if "saw_moog" in kind:
        gpu.wavetable_build("saw")
    V, T = 130, 64 * 4 + 13
    ni = gpu.Bank(kind, 1).inputs()
    x = noise_input(V, max(ni, 1), T, seed=7) if ni else None
    if x is not None and kind == "saw_moog_adsr_pan":
        x[:, 0, :] = 0.0
        x[:, 0, 3:200] = 1.0
    for mode in (MODE_PROCESS, MODE_TICK):
        ref = None
        for split in (1, 0, 2):
            for layout in ("voice-minor", "planar", "planar, tight rows"):
                b = gpu.Bank(kind, V)
                b.set_sample_rate(SR)
                b.set_option("pipe_split", split)
                b.set_seed(np.arange(V, dtype=np.uint64) * 3 + 11)
                if layout == "voice-minor":
                    xi = None if x is None else torch.from_numpy(np.ascontiguousarray(x.transpose(1, 2, 0))).cuda()
                    got = b.process(T, xi, mode=mode).cpu().numpy().transpose(2, 0, 1)
                else:
                    fs = T if layout.endswith("tight rows") else (T + 63) // 64 * 64
                    xi = None
                    if x is not None:
                        buf = np.zeros((V, ni, fs), dtype=np.float32)
                        buf[:, :, :T] = x
                        xi = torch.from_numpy(buf).cuda()
                    got = b.process(T, xi, layout=LAYOUT_PLANAR, frame_stride=fs, mode=mode).cpu().numpy()[:, :, :T]
                if ref is None:
                    ref = got
                else:
                    with np.errstate(all="ignore"):
                        assert_bit_equal(got, ref, f"{kind} mode {mode} pipe_split {split} {layout} != pipe_split 1 voice-minor")

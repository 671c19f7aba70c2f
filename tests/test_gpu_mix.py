"""The fused mix-down (fdsp_bank_process_mix, SURVEY.md 8(d) "mode B" / 8(e) "per-GPU on-device tree-sum"): the render kernels
reduce over the voices themselves -- the per-voice output never exists in HBM.  Reference shape: `voice >> pan(p)` per voice
(src/pan.rs:50-76) and the sum over voices (src/audionode.rs:2406-2462).

Bar: the fused mix equals fdsp_sum_voices / fdsp_mix_stereo of the VOICE-OUT render BIT FOR BIT (same fixed order), equals the
order's numpy statement applied to the oracle-checked voice-out samples, and is within sqrt(V) * 6e-8 * max|x| (+ 2 ulp of the mix) of
a serial f64 mix of the oracle's voices (the tolerance SURVEY 8(e) states for a re-ordered f32 sum)."""
import numpy as np
import pytest

import oracle as O
from mix_order import mix_order_reference
from fundsp_amd import LAYOUT_VOICE_MINOR, MIX_PAN, MIX_SUM, MODE_PROCESS, MODE_TICK
from fundsp_amd import workloads as W
from test_gpu_parity import assert_bit_equal

pytestmark = pytest.mark.gpu
SR = 48000.0


@pytest.fixture(scope="module")
def tables(gpu):
    gpu.wavetable_build("saw")
    return True


def serial_tolerance(voices_out, mix):
    """a re-ordered f32 sum of V terms against the serial mix: sqrt(V) * 6e-8 * max|x| -- plus two ulp of the largest
    mix value: the f32 RESULT cannot sit closer than half an ulp to the f64 mix, whatever the order (the first GPU run measured
    2.79e-6 against sqrt(200) * 6e-8 * 3.23 = 2.74e-6 on sums of magnitude 12, whose ulp is 9.5e-7)."""
    return np.sqrt(voices_out.shape[-1]) * 6e-8 * np.abs(voices_out).max() + 2.0 ** -22 * np.abs(mix).max()


def fm_bank(gpu, V, voice0=0):
    p = W.fm_svf_params(V, SR, voice0)
    b = W.make_fm_svf_bank(V, SR, params=p)
    pan = (-1.0 + 2.0 * W.rnd1(np.arange(voice0, voice0 + V, dtype=np.uint64) + np.uint64(777))).astype(np.float32)
    b.set_pan(pan)
    return b, p, pan


@pytest.mark.parametrize("V,T", [(200, 64 * 5 + 13), (64 * 7, 64 * 4), (8192, 64 * 6), (32768, 64 * 4 + 9), (64 * 7, 64), (64 * 9 + 5, 128), (64 * 300, 192), (65536, 64)])
@pytest.mark.parametrize("mode", [MODE_PROCESS, MODE_TICK])
def test_fm_pan_mix_equals_mix_of_voice_out(gpu, V, T, mode):
    """Config-3 voices, MIX_PAN: fused == fdsp_mix_stereo(voice-out) == the order's numpy statement, bit for bit; a ragged last
    voice group (200), the time-split kernels (8 192: one group per CU, T % 64 == 0; 32 768 with a ragged T: the pipeline)."""
    import torch

    if mode == MODE_TICK and V > 8192:
        pytest.skip("tick mode: the small cases cover it")
    b, p, pan = fm_bank(gpu, V)
    ref = b.clone()
    mix = b.process_mix(T, mix=MIX_PAN, mode=mode).cpu().numpy()
    out = ref.process(T, layout=LAYOUT_VOICE_MINOR, mode=mode)            # [1][T][V]
    unfused = gpu.mix_stereo(out[0], torch.from_numpy(pan).cuda()).cpu().numpy()
    assert_bit_equal(mix, unfused, f"fused vs mix_stereo(voice-out), V={V}")
    x = out[0].cpu().numpy()
    ang = (np.clip(pan, -1, 1).astype(np.float32) + np.float32(1)) * (np.float32(np.pi) * np.float32(0.25))
    wl = np.array([O.lib().o_math_cosf(float(a)) for a in ang], dtype=np.float32)
    wr = np.array([O.lib().o_math_sinf(float(a)) for a in ang], dtype=np.float32)
    assert_bit_equal(mix[0], mix_order_reference(x * wl[None, :]), "left vs the order's statement")
    assert_bit_equal(mix[1], mix_order_reference(x * wr[None, :]), "right vs the order's statement")
    # the voices' state after a fused launch is the state after a voice-out launch
    assert_bit_equal(b.get_state(), ref.get_state(), "state after the launch")
    assert np.abs(mix).max() > 0.1


@pytest.mark.parametrize("V,T", [(1, 1), (5, 3), (63, 8), (64, 65), (65, 129)])
def test_tiny_banks_and_launches(gpu, V, T):
    """One voice, one frame; fewer voices than a quarter; launches shorter than a chunk and one frame past a block: fused == unfused,
    both mix kinds, both modes."""
    import torch

    for mode in (MODE_PROCESS, MODE_TICK):
        b, p, pan = fm_bank(gpu, V)
        ref = b.clone()
        ref2 = b.clone()
        out = ref.process(T, layout=LAYOUT_VOICE_MINOR, mode=mode)
        assert_bit_equal(b.process_mix(T, mix=MIX_PAN, mode=mode).cpu().numpy(), gpu.mix_stereo(out[0], torch.from_numpy(pan).cuda()).cpu().numpy(), f"PAN V={V} T={T}")
        assert_bit_equal(ref2.process_mix(T, mix=MIX_SUM, mode=mode).cpu().numpy(), gpu.sum_voices(out).cpu().numpy(), f"SUM V={V} T={T}")


def test_fm_mix_against_the_oracles_serial_mix(gpu):
    """Voices from the CPU oracle, panned and added one after the other in f64: the fused f32 mix is within the stated bound."""
    V, T = 200, 64 * 3 + 5
    b, p, pan = fm_bank(gpu, V)
    mix = b.process_mix(T, mix=MIX_PAN).cpu().numpy()
    want, _ = O.bank_render(3, [p["f"], p["m"], p["fc"], p["q"]], p["seed"], T, SR, True, 1, 4)   # [T][V]
    ang = (np.clip(pan, -1, 1).astype(np.float32) + np.float32(1)) * (np.float32(np.pi) * np.float32(0.25))
    wl = np.array([O.lib().o_math_cosf(float(a)) for a in ang], dtype=np.float32)
    wr = np.array([O.lib().o_math_sinf(float(a)) for a in ang], dtype=np.float32)
    serial = np.stack([(want * wl[None, :]).astype(np.float64).sum(axis=1), (want * wr[None, :]).astype(np.float64).sum(axis=1)])
    assert np.abs(mix - serial).max() <= serial_tolerance(want, serial)
    # ... and exactly the order's statement applied to the oracle's own samples (the voices are bit-exact)
    assert_bit_equal(mix[0], mix_order_reference(want * wl[None, :]), "left vs oracle voices in the stated order")


def test_pan_mix_is_the_oracles_panner_per_voice(gpu):
    """FDSP_MIX_PAN against the reference's own shape: every voice as `voice >> pan(p)` through the ORACLE's Panner node (pan.rs:50-76),
    the stereo voices added in the mix-down's order -- bit for bit, pan positions beyond +-1 included (clamp11)."""
    V, T = 70, 64 * 2 + 11
    p = W.fm_svf_params(V, SR)
    b = W.make_fm_svf_bank(V, SR, params=p)
    pan = np.linspace(-1.2, 1.2, V).astype(np.float32)
    b.set_pan(pan)
    mix = b.process_mix(T, mix=MIX_PAN).cpu().numpy()
    want, _ = O.bank_render(3, [p["f"], p["m"], p["fc"], p["q"]], p["seed"], T, SR, True, 0, 4)      # the oracle's voices, [V][T]
    voices = [O.pan(float(pan[v])).render_blocks(want[v][None, :]) for v in range(V)]                # each through the oracle's Panner: [2][T]
    stereo = np.stack(voices, axis=-1)                                  # [2][T][V]
    assert_bit_equal(mix, mix_order_reference(stereo), "fused PAN mix vs the oracle's `voice >> pan(p)` voices in the stated order")


def test_fm_sum_mix_mono(gpu):
    """MIX_SUM of a mono graph: [1][T] = fdsp_sum_voices of the voice-out render (the Sequencer's mix of its events)."""
    V, T = 64 * 9 + 11, 64 * 5
    b, p, _ = fm_bank(gpu, V)
    ref = b.clone()
    mix = b.process_mix(T, mix=MIX_SUM)
    out = ref.process(T)
    assert mix.shape == (1, T)
    assert_bit_equal(mix.cpu().numpy(), gpu.sum_voices(out).cpu().numpy(), "sum mix vs sum_voices(voice-out)")


@pytest.mark.parametrize("V,T", [(130, 64 * 30 + 7), (64 * 300 + 5, 64 * 6), (32768, 64 * 5)])
def test_config4_mix_equals_sum_of_voice_out(gpu, tables, V, T):
    """BASELINE config 4 (`... >> pan(p)` per voice, "RCCL stereo mix-down"): MIX_SUM of the two output channels.  One group per
    workgroup (130 voices), two (19 205: ragged, odd group count), and the per-GPU shard of the config (32 768: two per workgroup)."""
    import torch

    adsr = (0.005, 0.01, 0.6, 0.01)
    p = W.saw_moog_params(V, SR)
    b = W.make_saw_moog_bank(V, SR, params=p, adsr=adsr)
    ref = b.clone()
    gate = torch.from_numpy(np.broadcast_to(W.gate_signal(T, SR, on_frame=1, off_seconds=1200 / SR)[None, :, None], (1, T, V)).copy()).cuda()
    mix = b.process_mix(T, gate, mix=MIX_SUM)
    out = ref.process(T, gate)                                             # [2][T][V]
    assert mix.shape == (2, T)
    assert_bit_equal(mix.cpu().numpy(), gpu.sum_voices(out).cpu().numpy(), f"fused vs sum_voices(voice-out), V={V}")
    assert_bit_equal(mix.cpu().numpy(), mix_order_reference(out.cpu().numpy()), "vs the order's statement")
    assert_bit_equal(b.get_state(), ref.get_state(), "state after the launch")
    assert np.abs(mix.cpu().numpy()).max() > 0.05
    with pytest.raises(gpu.FdspError):
        b.process_mix(T, gate, mix=MIX_PAN)                                # a stereo graph is not panned again


def test_config4_mix_small_against_oracle_voices(gpu, tables):
    from test_gpu_config4 import config4_oracle_voice
    from test_gpu_parity import oracle_render

    V, T = 70, 64 * 20 + 3
    adsr = (0.005, 0.01, 0.6, 0.01)
    p = W.saw_moog_params(V, SR)
    b = W.make_saw_moog_bank(V, SR, params=p, adsr=adsr)
    import torch
    g1 = W.gate_signal(T, SR, on_frame=1, off_seconds=900 / SR)
    gate = torch.from_numpy(np.broadcast_to(g1[None, :, None], (1, T, V)).copy()).cuda()
    mix = b.process_mix(T, gate, mix=MIX_SUM).cpu().numpy()
    voices = np.stack([oracle_render(config4_oracle_voice(p, v, adsr), g1[None, :], T, MODE_PROCESS) for v in range(V)], axis=-1)  # [2][T][V]
    assert_bit_equal(mix, mix_order_reference(voices), "fused mix vs the oracle's voices in the stated order")
    assert np.abs(mix - voices.astype(np.float64).sum(axis=-1)).max() <= serial_tolerance(voices, mix)


def test_chunked_mix_equals_whole_and_reserve(gpu):
    """A launch of 64 * 6 frames == six launches of 64 (T < 256 goes through the pipeline kernel in a mix launch as well);
    fdsp_bank_mix_reserve sizes the partial buffer ahead; kinds without the fused kernels say so."""
    V = 64 * 3 + 7
    b, p, _ = fm_bank(gpu, V)
    c = b.clone()
    c.mix_reserve(64)
    whole = b.process_mix(64 * 6, mix=MIX_PAN).cpu().numpy()
    parts = [c.process_mix(64, mix=MIX_PAN).cpu().numpy() for _ in range(6)]
    assert_bit_equal(whole, np.concatenate(parts, axis=1), "chunked == whole")
    lone = gpu.Bank("noise", 64)
    with pytest.raises(gpu.FdspError) as e:
        lone.process_mix(64, mix=MIX_SUM)
    assert e.value.code == -4 and "fdsp_sum_voices" in str(e.value)


def test_noise_biquad_sum_mix(gpu):
    V, T = 1024, 64 * 8 + 3
    b = W.make_noise_biquad_bank(V, SR)
    ref = b.clone()
    mix = b.process_mix(T, mix=MIX_SUM).cpu().numpy()
    out = ref.process(T)
    assert_bit_equal(mix, gpu.sum_voices(out).cpu().numpy(), "config-2 voices: fused sum vs sum_voices(voice-out)")


def test_shards_add_up_to_the_whole_bank_exactly_at_aligned_splits(gpu):
    """contiguous voice shards, one all-reduce of the partial mixes.  The groups' partials are added in an ALIGNED
    binary tree, so a bank split at a power-of-two group boundary reproduces the whole bank's mix with ONE addition per sample:
    mix(whole) == mix(first half) + mix(second half) bit for bit (what a 2-GPU all-reduce computes); an unaligned split agrees
    within the re-ordering tolerance."""
    V, T = 64 * 16, 64 * 5
    whole, p, pan = fm_bank(gpu, V)
    want = whole.process_mix(T, mix=MIX_PAN).cpu().numpy()
    halves = []
    for first in (0, V // 2):
        b, _, _ = fm_bank(gpu, V // 2, voice0=first)
        halves.append(b.process_mix(T, mix=MIX_PAN).cpu().numpy())
    assert_bit_equal(want, halves[0] + halves[1], "whole == half + half at an aligned split")
    parts = []
    for first, n in ((0, 64 * 5 + 3), (64 * 5 + 3, V - (64 * 5 + 3))):
        b, _, _ = fm_bank(gpu, n, voice0=first)
        parts.append(b.process_mix(T, mix=MIX_PAN).cpu().numpy())
    assert np.abs(want - (parts[0] + parts[1])).max() <= np.sqrt(V) * 6e-8 * 4.0 + 2.0 ** -22 * np.abs(want).max()


def test_run_time_compiled_graph_mixes(gpu):
    """fdsp_bank_process_mix on a run-time compiled graph (the Rust front door's path): the mix kernels are compiled on first use."""
    import torch
    from fundsp_amd import graph as G

    V, T = 64 * 5 + 9, 64 * 6 + 5
    f = (110.0 * 2.0 ** (3.0 * W.rnd1(np.arange(V, dtype=np.uint64)))).astype(np.float32)
    g = (G.dc(f) >> G.sine()) * 0.5 >> G.lowpass_hz(1200.0, 0.8)
    b = gpu.Bank.from_graph(g, V, sample_rate=SR)
    b.set_seed(np.arange(V, dtype=np.uint64) + 11)
    assert b.get_option("has_fused_mix") == 1
    ref = b.clone()
    pan = np.linspace(-1, 1, V).astype(np.float32)
    b.set_pan(pan)
    mix = b.process_mix(T, mix=MIX_PAN).cpu().numpy()
    out = ref.process(T)
    assert_bit_equal(mix, gpu.mix_stereo(out[0], torch.from_numpy(pan).cuda()).cpu().numpy(), "run-time compiled graph: fused vs mix_stereo(voice-out)")
    summed = b.clone()
    s1 = summed.process_mix(T, mix=MIX_SUM).cpu().numpy()
    o2 = b.process(T)
    assert_bit_equal(s1, gpu.sum_voices(o2).cpu().numpy(), "the next block, MIX_SUM")


def test_run_time_compiled_wide_graph_has_no_fused_mix_and_says_so(gpu):
    """A run-time compiled graph of four outputs: a mix tile of eight frames does not fit beside the pipeline's tiles at four voice groups
    per workgroup (MixGeom), so the kind has no fused mix-down -- "has_fused_mix" says 0 and fdsp_bank_process_mix answers FDSP_ENOTSUP
    at once (it used to spend a hiprtc compile on a module that could not build).  Three outputs fit."""
    from fundsp_amd import graph as G

    V, T = 64 * 3, 64 * 2
    f = (110.0 * 2.0 ** (3.0 * W.rnd1(np.arange(V, dtype=np.uint64)))).astype(np.float32)
    osc = lambda k: (G.dc(f * (k + 1)) >> G.sine()) * 0.5 >> G.lowpass_hz(1200.0, 0.8)   # noqa: E731
    wide = gpu.Bank.from_graph(osc(0) | osc(1) | osc(2) | osc(3), V, sample_rate=SR)
    assert wide.outputs() == 4 and wide.get_option("has_fused_mix") == 0
    with pytest.raises(gpu.FdspError):
        wide.process_mix(T, mix=MIX_SUM)
    three = gpu.Bank.from_graph(osc(0) | osc(1) | osc(2), V, sample_rate=SR)
    three.set_seed(np.arange(V, dtype=np.uint64))
    if three.get_option("has_fused_mix") == 1:      # (a graph without a pipeline plan has none either)
        three.mix_reserve(T)                          # compiles and loads the mix kernels ahead of the first launch
        ref = three.clone()
        mix = three.process_mix(T, mix=MIX_SUM).cpu().numpy()
        assert_bit_equal(mix, gpu.sum_voices(ref.process(T)).cpu().numpy(), "three-output run-time compiled graph: fused vs sum_voices")


def test_run_time_compiled_filter_chain_mixes_with_lds_weights(gpu):
    """A graph with an audio input and two compute stages -- 12 waves per workgroup of four voice groups: the pan weights of the
    fused mix-down live in LDS and the flush is a loop there (render_pipe_body TIGHT); a moog in the chain makes it `heavy`, so the
    voice-out render of the small bank takes narrower workgroups than the mix launch: the partial mixes do not depend on that."""
    import torch
    from fundsp_amd import graph as G

    V, T = 64 * 6 + 17, 64 * 7 + 3
    g = G.lowpass_hz(900.0, 0.7) >> G.moog_hz(1500.0, 0.4) >> G.highpass_hz(120.0, 0.7)
    b = gpu.Bank.from_graph(g, V, sample_rate=SR)
    rng = np.random.default_rng(21)
    x = torch.from_numpy((rng.random((1, T, V), dtype=np.float32) - 0.5).astype(np.float32)).cuda()
    ref = b.clone()
    pan = (rng.random(V, dtype=np.float32) * 2 - 1).astype(np.float32)
    b.set_pan(pan)
    mix = b.process_mix(T, x, mix=MIX_PAN).cpu().numpy()
    out = ref.process(T, x)
    assert_bit_equal(mix, gpu.mix_stereo(out[0], torch.from_numpy(pan).cuda()).cpu().numpy(), "filter chain with an input: fused PAN vs mix_stereo(voice-out)")
    s1 = ref.clone().process_mix(T, x, mix=MIX_SUM).cpu().numpy()
    assert_bit_equal(s1, gpu.sum_voices(ref.process(T, x)).cpu().numpy(), "the next block, MIX_SUM")


def test_mix_on_a_caller_stream_and_in_a_hip_graph(gpu):
    """Stream rules of fdsp_bank_process: a caller's stream, and a stream capture after fdsp_bank_mix_reserve / set_pan (no allocation
    inside the capture); an unreserved bank refuses to mix during a capture."""
    import torch

    V, T = 64 * 4, 64
    b, p, pan = fm_bank(gpu, V)
    ref = b.clone()
    want = np.concatenate([ref.process_mix(T, mix=MIX_PAN).cpu().numpy() for _ in range(3)], axis=1)
    b.mix_reserve(T)
    outs = [torch.empty((2, T), dtype=torch.float32, device="cuda") for _ in range(3)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for k in range(3):
                b.process_mix(T, mix=MIX_PAN, out=outs[k])
        g.replay()
    torch.cuda.synchronize()
    assert_bit_equal(np.concatenate([o.cpu().numpy() for o in outs], axis=1), want, "three captured mix launches == three plain ones")
    fresh, _, _ = fm_bank(gpu, V)
    with torch.cuda.stream(s):
        g2 = torch.cuda.CUDAGraph()
        with pytest.raises(gpu.FdspError):
            with torch.cuda.graph(g2, stream=s):
                fresh.process_mix(T, mix=MIX_SUM, out=outs[0])


def test_sum_instances_planar(gpu):
    import torch

    n, ch, fs = 37, 2, 96
    rng = np.random.default_rng(9)
    x = (rng.random((n, ch, fs), dtype=np.float32) - 0.5).astype(np.float32)
    got = gpu.sum_instances(torch.from_numpy(x).cuda()).cpu().numpy()
    level = [x[i] for i in range(n)]
    while len(level) > 1:
        nxt = [level[i] + level[i + 1] for i in range(0, len(level) - 1, 2)]
        if len(level) & 1:
            nxt.append(level[-1])
        level = nxt
    assert_bit_equal(got, level[0], "sum over instances: aligned binary tree")


INVENTORY_MIX = ["noise_moog", "stack_binop_sub", "comb_allpass_chain", "saw_filter_env", "chorus_tap", "pulse_resonator",
                 "bus_branch_thru", "busf_resonators", "svf_q_forms", "pulse_wave", "multitap_allnest_panner", "limiter_stereo"]


@pytest.mark.parametrize("name", INVENTORY_MIX)
def test_inventory_graphs_mix_or_refuse(gpu, name):
    """A sweep of the run-time compiled inventory (tests/test_gpu_jit.py GRAPHS: rings, wavetables, inputs, two outputs, stages of
    every weight) through fdsp_bank_process_mix: a graph either mixes bit-equal to sum_voices / mix_stereo of its voice-out render
    -- padded lanes of the last voice group included, V is not a multiple of 64 -- or refuses with FDSP_ENOTSUP and leaves the
    bank where it was (the voice-out render that follows equals the one of an untouched clone)."""
    import torch
    from fundsp_amd import FdspError
    from fundsp_amd._lib import ENOTSUP
    from fundsp_amd import graph as G
    from test_gpu_jit import GRAPHS, noise_input

    build, ni, ring = GRAPHS[name]
    g = build(G)
    V, T = 64 * 3 + 21, 64 * 5 + 9
    for kind in G.uses_wavetables(g):
        t = O.Wavetable.get(kind)
        offs = np.concatenate([[0], np.cumsum(t.lengths)])
        gpu.wavetable_upload(kind, t.pitches, [t.data[offs[i]:offs[i + 1]] for i in range(len(t.lengths))])
    b = gpu.Bank.from_graph(g, V, ring_frames=ring, sample_rate=SR)
    b.set_seed(np.arange(V, dtype=np.uint64) * 977 + 3)
    x = None
    if ni:
        xs = noise_input(V, ni, T, seed=5)                       # [voice][channel][frame]
        x = torch.from_numpy(np.ascontiguousarray(xs.transpose(1, 2, 0))).cuda()   # voice-minor
    ref = b.clone()
    modes = [MIX_SUM] + ([MIX_PAN] if b.outputs() == 1 else [])
    pan = np.cos(np.arange(V, dtype=np.float32)).astype(np.float32)
    for mix in modes:
        fused, plain = ref.clone(), ref.clone()
        if mix == MIX_PAN:
            fused.set_pan(pan)
        try:
            got = fused.process_mix(T, x, mix=mix).cpu().numpy()
        except FdspError as e:
            assert e.code == ENOTSUP, e
            with pytest.raises(FdspError):   # no launch happened: there is no event pair to read
                fused.last_kernel_ms()
            assert_bit_equal(fused.process(T, x).cpu().numpy(), plain.process(T, x).cpu().numpy(), f"{name}: a refused mix leaves the bank alone")
            continue
        out = plain.process(T, x)
        want = gpu.sum_voices(out) if mix == MIX_SUM else gpu.mix_stereo(out[0], torch.from_numpy(pan).cuda())
        assert_bit_equal(got, want.cpu().numpy(), f"{name}: fused mix {mix} vs the mix of the voice-out render")
        assert_bit_equal(fused.process(T, x).cpu().numpy(), plain.process(T, x).cpu().numpy(), f"{name}: state after the fused mix")

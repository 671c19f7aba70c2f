"""BASELINE.json's full sizes on the device, checked through size-independent properties: spot bit-parity of randomly
chosen voices against the oracle, chunked == whole (state carry), voice independence (a voice's samples do not depend
on which bank or launch geometry it sits in), finiteness.  Outputs stay in HBM; only the spot voices cross PCIe."""
import numpy as np
import pytest

import oracle as O
from fundsp_amd import LAYOUT_PLANAR, LAYOUT_VOICE_MINOR, MIX_PAN, MIX_SUM, MODE_PROCESS
from fundsp_amd import workloads as W
from test_gpu_config4 import config4_oracle_voice, tables  # noqa: F401
from test_gpu_parity import assert_bit_equal

pytestmark = pytest.mark.gpu
SR = 48000.0


def test_config3_full_size(gpu):
    import torch

    V, T = 65536, 48000
    p = W.fm_svf_params(V, SR)
    bank = W.make_fm_svf_bank(V, SR, params=p)
    out = bank.process(T)                                     # [1][T][V], 12.6 GB
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all())
    assert float(out.abs().max()) < 16.0                      # resonant lowpass of a unit sine: bounded by its peak gain
    rng = np.random.default_rng(2026)
    spots = np.concatenate([[0, 63, 64, V - 1], rng.integers(0, V, 12)])
    got = out[0][:, torch.from_numpy(spots).cuda()].t().contiguous().cpu().numpy()
    want, _ = O.bank_render(3, [p["f"][spots], p["m"][spots], p["fc"][spots], p["q"][spots]], p["seed"][spots], T, SR,
                            process_mode=True, out_layout=0, threads=8)
    assert_bit_equal(got, want, "config 3 full size, spot voices")
    # chunked == whole: two launches of 375 blocks each
    b2 = W.make_fm_svf_bank(V, SR, params=p)
    a = b2.process(T // 2)
    b = b2.process(T // 2)
    assert torch.equal(out[:, :T // 2], a) and torch.equal(out[:, T // 2:], b)
    del a, b, b2
    # voice independence: the same voices in a small bank (different grid, partially filled waves)
    small = W.make_fm_svf_bank(200, SR, voice0=30000)
    s = small.process(T)
    assert torch.equal(out[0][:, 30000:30200], s[0])
    del s, small
    # mode B at full size (the bench's config3_mix_pan_fused launch shape: 750 blocks, 1 024 voice groups): every voice panned and summed
    # inside the render launch == fdsp_mix_stereo of the voice-out render above, bit for bit
    pan = (-1.0 + 2.0 * W.rnd1(np.arange(V, dtype=np.uint64) + np.uint64(777))).astype(np.float32)
    bm = W.make_fm_svf_bank(V, SR, params=p)
    bm.set_pan(pan)
    fused = bm.process_mix(T, mix=MIX_PAN)
    unfused = gpu.mix_stereo(out[0], torch.from_numpy(pan).cuda())
    assert fused.shape == (2, T) and torch.equal(fused, unfused), "config 3 full size: fused mix-down vs mix_stereo(voice-out)"
    del bm, fused, unfused
    # the reference-native planar layout at full size (planar pipeline kernel): the same samples, transposed
    b3 = W.make_fm_svf_bank(V, SR, params=p)
    planar = b3.process(T, layout=LAYOUT_PLANAR, frame_stride=T)   # [V][1][T]
    torch.cuda.synchronize()
    for v0 in range(0, V, 8192):                                  # compare in slabs: no 12.6 GB transpose temporary
        assert torch.equal(planar[v0:v0 + 8192, 0, :], out[0][:, v0:v0 + 8192].t())


def test_config4_full_size(gpu, tables):
    import torch

    V, T = 32768, 48000
    adsr = (0.01, 0.1, 0.6, 0.2)
    p = W.saw_moog_params(V, SR)
    bank = W.make_saw_moog_bank(V, SR, params=p, adsr=adsr)
    gate_np = W.gate_signal(T, SR)
    gate = torch.from_numpy(gate_np).cuda()[None, :, None].expand(1, T, V).contiguous()
    out = bank.process(T, gate)                               # [2][T][V]
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all())
    rng = np.random.default_rng(2027)
    spots = np.concatenate([[0, V - 1], rng.integers(0, V, 4)])
    got = out[:, :, torch.from_numpy(spots).cuda()].permute(2, 0, 1).contiguous().cpu().numpy()
    for k, v in enumerate(spots):
        want = config4_oracle_voice(p, int(v), adsr).render_blocks(gate_np[None, :])
        assert_bit_equal(got[k], want, f"config 4 full size, voice {v}")
    mix = gpu.sum_voices(out)                                 # the per-GPU partial of the stereo mix-down
    assert mix.shape == (2, T) and bool(torch.isfinite(mix).all())
    # mode B at full size (the bench's config4_mix_single_rank launch shape): the fused mix-down == sum_voices of the voice-out render
    bm = W.make_saw_moog_bank(V, SR, params=p, adsr=adsr)
    fused = bm.process_mix(T, gate, mix=MIX_SUM)
    assert torch.equal(fused, mix), "config 4 full size: fused mix-down vs sum_voices(voice-out)"
    assert float(mix.abs().max()) > 1.0


def test_config4_var_gate_full_size(gpu, tables):
    """Config 4 in the reference's gate shape (`var(gate) >> adsr_live`, tests/test_gpu_config4_var.py) at the bench's sizes: one note per
    second = two launches of 24 000 frames with the Var slot set in between; spot voices vs the oracle, chunked == whole, mode B == the
    mix of the voice-out render."""
    import torch
    from test_gpu_config4_var import device_plan, oracle_plan  # noqa: F401

    V, T = 32768, 48000
    adsr = (0.01, 0.1, 0.6, 0.2)
    p = W.saw_moog_params(V, SR)
    plan = W.gate_plan(T, SR)
    assert plan == [(1.0, 24000), (0.0, 24000)]
    bank = W.make_saw_moog_var_bank(V, SR, params=p, adsr=adsr)      # primed with one low block
    bm = bank.clone()
    bc = bank.clone()
    outs = []
    for value, n in plan:
        bank.set_param(W.C4V_SLOTS["gate"], value)
        outs.append(bank.process(n))                                 # [2][n][V]
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(o).all()) for o in outs)
    rng = np.random.default_rng(2028)
    spots = np.concatenate([[0, V - 1], rng.integers(0, V, 4)])
    idx = torch.from_numpy(spots).cuda()
    got = torch.cat([o[:, :, idx] for o in outs], dim=1).permute(2, 0, 1).contiguous().cpu().numpy()
    for k, v in enumerate(spots):
        want = oracle_plan(p, int(v), adsr, [(0.0, 64)] + plan, MODE_PROCESS)[:, 64:]
        assert_bit_equal(got[k], want, f"config 4 (var gate) full size, voice {v}")
    mixes = []
    for value, n in plan:
        bm.set_param(W.C4V_SLOTS["gate"], value)
        mixes.append(bm.process_mix(n, mix=MIX_SUM))
    for o, m in zip(outs, mixes):
        assert torch.equal(m, gpu.sum_voices(o)), "config 4 (var gate) full size: fused mix-down vs sum_voices(voice-out)"
    assert float(mixes[0].abs().max()) > 1.0
    # chunked == whole: the first half in two launches of 12 000 frames (187.5 blocks: a ragged launch starts a new block, so compare the
    # block-aligned form 11 968 + 12 032)
    bc.set_param(W.C4V_SLOTS["gate"], 1.0)
    a = bc.process(11968)
    b = bc.process(12032)
    assert torch.equal(outs[0][:, :11968], a) and torch.equal(outs[0][:, 11968:], b)


def test_config5_full_size(gpu):
    import torch

    V, T = 2048, 48000
    bank = gpu.Bank.reverb_stereo(V, 10.0, 2.0, 0.5)
    bank.set_sample_rate(SR)
    g = torch.Generator(device="cuda").manual_seed(99)
    x = torch.rand((V, 2, T), dtype=torch.float32, device="cuda", generator=g) * 2 - 1
    out = bank.process(T, x, layout=LAYOUT_PLANAR, frame_stride=T)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all())
    for v in (0, 1, 1027, V - 1):
        n = O.reverb_stereo(10.0, 2.0, 0.5)
        n.set_sample_rate(SR)
        assert_bit_equal(out[v].cpu().numpy(), n.render_blocks(x[v].cpu().numpy()), f"config 5 full size, instance {v}")
    b2 = gpu.Bank.reverb_stereo(V, 10.0, 2.0, 0.5)
    b2.set_sample_rate(SR)
    a = b2.process(1000, x[:, :, :1000].contiguous(), layout=LAYOUT_PLANAR, frame_stride=1000)   # ragged chunk
    b = b2.process(T - 1000, x[:, :, 1000:].contiguous(), layout=LAYOUT_PLANAR, frame_stride=T - 1000)
    assert torch.equal(out[:, :, :1000], a) and torch.equal(out[:, :, 1000:], b)


def test_reverb4_stereo_full_size(gpu):
    """reverb4_stereo(20, 2) at config 5's sizes (2048 instances x 48 000 frames) through the lane-per-frame FDN kernel (two 16-line
    networks in series): spot instances vs the oracle's generic Feedback graph, chunked == whole."""
    import torch

    V, T = 2048, 48000
    bank = gpu.Bank.reverb4_stereo(V, 20.0, 2.0)
    bank.set_sample_rate(SR)
    g = torch.Generator(device="cuda").manual_seed(199)
    x = torch.rand((V, 2, T), dtype=torch.float32, device="cuda", generator=g) * 2 - 1
    out = bank.process(T, x, layout=LAYOUT_PLANAR, frame_stride=T)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all()) and float(out.abs().max()) > 0.1
    for v in (0, 1, 1027, V - 1):
        n = O.reverb4_stereo(20.0, 2.0)
        n.set_sample_rate(SR)
        assert_bit_equal(out[v].cpu().numpy(), n.render_blocks(x[v].cpu().numpy()), f"reverb4_stereo full size, instance {v}")
    b2 = gpu.Bank.reverb4_stereo(V, 20.0, 2.0)
    b2.set_sample_rate(SR)
    a = b2.process(64 * 100, x[:, :, :6400].contiguous(), layout=LAYOUT_PLANAR, frame_stride=6400)
    b = b2.process(T - 6400, x[:, :, 6400:].contiguous(), layout=LAYOUT_PLANAR, frame_stride=T - 6400)
    assert torch.equal(out[:, :, :6400], a) and torch.equal(out[:, :, 6400:], b)


def test_config5_with_the_documented_bus_full_size(gpu):
    """README.md:436's `multipass() & 0.2 * reverb_stereo(20.0, 2.0, 1.0)` at config 5's sizes (2048 instances x 48 000 frames), built from the graph
    (graph.bus_plan -> fdsp_bank_set_bus): spot instances against the oracle's WHOLE graph (Bus, MultiPass, Unop and the reverb), and -- the
    size-independent property -- every sample of every instance equal to `in + 0.2 * (the bare reverb's output)`, one rounding per operation,
    evaluated by separate elementwise kernels over the whole buffers."""
    import torch

    from fundsp_amd import BUS_DRY_WET
    from fundsp_amd import graph as GR

    V, T = 2048, 48000
    bank = gpu.Bank.from_graph(GR.multipass(2) & 0.2 * GR.reverb_stereo(20.0, 2.0, 1.0), V, sample_rate=SR)
    assert bank.kind == "reverb_stereo" and bank.get_bus()[0] == BUS_DRY_WET
    g = torch.Generator(device="cuda").manual_seed(299)
    x = torch.rand((V, 2, T), dtype=torch.float32, device="cuda", generator=g) * 2 - 1
    out = bank.process(T, x, layout=LAYOUT_PLANAR, frame_stride=T)
    torch.cuda.synchronize()
    for v in (0, 1027, V - 1):
        n = O.multipass(2) & 0.2 * O.reverb_stereo(20.0, 2.0, 1.0)
        n.set_sample_rate(SR)
        assert_bit_equal(out[v].cpu().numpy(), n.render_blocks(x[v].cpu().numpy()), f"multipass() & 0.2 * reverb_stereo full size, instance {v}")
    bare = gpu.Bank.reverb_stereo(V, 20.0, 2.0, 1.0)
    bare.set_sample_rate(SR)
    y = bare.process(T, x, layout=LAYOUT_PLANAR, frame_stride=T)
    torch.cuda.synchronize()
    wet = y * torch.tensor(0.2, dtype=torch.float32, device="cuda")     # (one kernel per operation: no contraction)
    assert torch.equal(out, x + wet)

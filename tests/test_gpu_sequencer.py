"""On-device voice scheduler vs the oracle's Sequencer restatement (sequencer.rs): every voice's faded contribution must
match its event's contribution bit for bit, in process and tick semantics; the mix is their sum."""
import os

import numpy as np
import pytest

import oracle as O
from fundsp_amd import FADE_POWER, FADE_SMOOTH, MODE_PROCESS, MODE_TICK
from fundsp_amd import workloads as W
from test_gpu_config4 import tables  # noqa: F401  (fixture: the oracle's wavetables installed on the device)
from test_gpu_parity import assert_bit_equal

pytestmark = pytest.mark.gpu
SR = 48000.0


def random_events(V, T, rng):
    start = rng.integers(0, T // 2, V).astype(np.float64)
    dur = rng.integers(40, T // 2, V).astype(np.float64)
    start[0], dur[0] = 0.0, float(T + 100)                 # plays through the whole launch
    start[1], dur[1] = 37.0, 3.0                           # shorter than one SIMD item
    start[2], dur[2] = 64.0, 64.0                          # exactly one sequencer block
    start[3], dur[3] = float(T + 10), 50.0                 # never starts inside the launch
    start += rng.random(V) * 0.4 - 0.2                     # off-grid times: rounding to the sample grid
    fin = np.minimum(dur - 1, rng.integers(0, 300, V)) * (rng.random(V) < 0.8)    # push(): fades <= duration
    fout = np.minimum(dur - 1, rng.integers(0, 300, V)) * (rng.random(V) < 0.8)
    fade = rng.integers(0, 2, V).astype(np.int32)
    return start / SR, (start + dur) / SR, fin / SR, fout / SR, fade


@pytest.mark.parametrize("seed", [91 + k for k in range(int(os.environ.get("FUNDSP_FUZZ_EVENTS", "1")))])   # more for a hunt
@pytest.mark.parametrize("mode", [MODE_PROCESS, MODE_TICK])
def test_fm_voices_with_events(gpu, mode, seed):
    import torch

    V, T = 70, 64 * 9 + 21
    rng = np.random.default_rng(seed)
    p = W.fm_svf_params(V, SR)
    start, end, fin, fout, fade = random_events(V, T, rng)
    b = W.make_fm_svf_bank(V, SR, params=p)
    b.set_events(start, end, fin, fout, fade)
    got = b.process_events(T, mode=mode)
    torch.cuda.synchronize()
    got = got.cpu().numpy().transpose(2, 0, 1)             # [V][1][T]
    seq = O.Sequencer(0, 1, SR)
    for v in range(V):
        f, m = float(p["f"][v]), float(p["m"][v])
        n = O.sine_hz(f) * f * m + f >> O.sine() >> O.lowpass_hz(float(p["fc"][v]), float(p["q"][v]))
        n.set_seed(int(p["seed"][v]))
        seq.push(start[v], end[v], int(fade[v]), fin[v], fout[v], n)
    mix, per = seq.render(T, process=(mode == MODE_PROCESS))
    for v in range(V):
        assert_bit_equal(got[v], per[v], f"event voice {v}")
    assert np.any(got[0] != 0) and not np.any(got[3] != 0)
    assert abs(b.events_time() - seq.time()) == 0.0
    summed = gpu.sum_voices(torch.from_numpy(np.ascontiguousarray(got.transpose(1, 2, 0))).cuda()).cpu().numpy()
    assert np.allclose(summed, mix, atol=1e-4)             # tree order vs the sequencer's serial order
    # the Sequencer's OUTPUT in one launch (fdsp_bank_process_events_mix): bit-equal to the sum of the per-event render above, the same
    # clock afterwards, and the same voice state
    b2 = W.make_fm_svf_bank(V, SR, params=p)
    b2.set_events(start, end, fin, fout, fade)
    fused = b2.process_events_mix(T, mode=mode).cpu().numpy()
    assert_bit_equal(fused, summed, "Sequencer output fused vs sum_voices(process_events)")
    assert b2.events_time() == b.events_time()
    assert_bit_equal(b2.get_state(), b.get_state(), "voice state after the fused launch")


def test_sustained_voices_take_the_packed_path(gpu):
    """Blocks in which every voice of a wave is inside its event and no fade is running render through the packed
    two-frame path (a plain process(size) of the unit); waves enter and leave that state at different times here --
    fades at the start, staggered ends, a ragged last block -- and every sample must still match the Sequencer oracle."""
    import torch

    V, T = 64 * 2 + 20, 64 * 12 + 29
    rng = np.random.default_rng(93)
    p = W.fm_svf_params(V, SR)
    start = np.zeros(V)
    start[64:128] = 100.3                                   # second wave starts later, off the block grid
    dur = np.full(V, float(T + 50))
    dur[128:] = rng.integers(300, 700, V - 128)             # third (partial) wave: staggered ends
    fin = np.where(np.arange(V) % 3 == 0, 150.0, 0.0)       # some voices fade in over 2.3 blocks
    fout = np.zeros(V)
    fout[128:] = 90.0
    fade = (np.arange(V) % 2).astype(np.int32)
    start, end, fin, fout = start / SR, (start + dur) / SR, fin / SR, fout / SR
    b = W.make_fm_svf_bank(V, SR, params=p)
    b.set_events(start, end, fin, fout, fade)
    got = b.process_events(T, mode=MODE_PROCESS)
    torch.cuda.synchronize()
    got = got.cpu().numpy().transpose(2, 0, 1)
    seq = O.Sequencer(0, 1, SR)
    for v in range(V):
        f, m = float(p["f"][v]), float(p["m"][v])
        n = O.sine_hz(f) * f * m + f >> O.sine() >> O.lowpass_hz(float(p["fc"][v]), float(p["q"][v]))
        n.set_seed(int(p["seed"][v]))
        seq.push(start[v], end[v], int(fade[v]), fin[v], fout[v], n)
    _, per = seq.render(T, process=True)
    for v in range(V):
        assert_bit_equal(got[v], per[v], f"sustained voice {v}")


@pytest.mark.parametrize("mode", [MODE_PROCESS, MODE_TICK])
def test_fully_sustained_launch_is_a_plain_render(gpu, mode):
    """A launch that every voice sustains (inside its event, no fade running) is dispatched to the render kernels instead
    of the scheduler kernel; three launches -- fades, sustained, ends -- must still add up to the Sequencer oracle's
    samples, and the clock must advance identically."""
    import torch

    V = 64 * 3 + 5
    T1, T2, T3 = 64 * 5, 64 * 6, 64 * 4 + 11          # whole sequencer blocks for the first two launches
    T = T1 + T2 + T3
    rng = np.random.default_rng(94)
    p = W.fm_svf_params(V, SR)
    start = rng.integers(0, 40, V).astype(np.float64) + rng.random(V) * 0.3
    fin = rng.integers(0, 200, V).astype(np.float64)                     # all fade-ins over before T1 = 320
    end = (T1 + T2) + rng.integers(60, T3, V).astype(np.float64)         # all ends inside the third launch
    fout = rng.integers(0, 50, V).astype(np.float64)                     # fade-outs start after T1 + T2
    fade = rng.integers(0, 2, V).astype(np.int32)
    start, end, fin, fout = start / SR, end / SR, fin / SR, fout / SR
    b = W.make_fm_svf_bank(V, SR, params=p)
    b.set_events(start, end, fin, fout, fade)
    outs = [b.process_events(n, mode=mode) for n in (T1, T2, T3)]
    torch.cuda.synchronize()
    got = torch.cat(outs, dim=1).cpu().numpy().transpose(2, 0, 1)
    seq = O.Sequencer(0, 1, SR)
    for v in range(V):
        f, m = float(p["f"][v]), float(p["m"][v])
        n = O.sine_hz(f) * f * m + f >> O.sine() >> O.lowpass_hz(float(p["fc"][v]), float(p["q"][v]))
        n.set_seed(int(p["seed"][v]))
        seq.push(start[v], end[v], int(fade[v]), fin[v], fout[v], n)
    _, per = seq.render(T, process=(mode == MODE_PROCESS))
    for v in range(V):
        assert_bit_equal(got[v], per[v], f"voice {v}")
    assert abs(b.events_time() - seq.time()) == 0.0
    # the same three launches mixed in the launch (the sustained one takes the pipeline kernel with the fused mix-down): the Sequencer's
    # output, bit-equal to the sum of the per-event samples
    b2 = W.make_fm_svf_bank(V, SR, params=p)
    b2.set_events(start, end, fin, fout, fade)
    mixes, kernels = [], []
    for n in (T1, T2, T3):
        mixes.append(b2.process_events_mix(n, mode=mode))
        kernels.append(b2.get_option("last_kernel"))
    fused = torch.cat(mixes, dim=1).cpu().numpy()
    assert kernels[0] == 5 and kernels[2] == 5 and kernels[1] in (1, 2, 4), kernels   # scheduler kernel | render kernel | scheduler kernel
    assert_bit_equal(fused, gpu.sum_voices(torch.cat(outs, dim=1)).cpu().numpy(), "three mixed launches vs sum_voices of the per-event render")
    assert b2.events_time() == b.events_time()


def test_gated_voices_with_inputs_and_two_launches(gpu, tables):
    """A kind with an input (saw >> moog * adsr >> pan, gate in) scheduled per voice, rendered in two launches of whole
    sequencer blocks: the clock and every voice's state carry over."""
    import torch

    V, T1, T2 = 64, 64 * 4, 64 * 3 + 17
    T = T1 + T2
    rng = np.random.default_rng(92)
    p = W.saw_moog_params(V, SR)
    adsr = (0.005, 0.01, 0.6, 0.01)
    b = W.make_saw_moog_bank(V, SR, params=p, adsr=adsr)
    start, end, fin, fout, fade = random_events(V, T, rng)
    b.set_events(start, end, fin, fout, fade)
    gate = np.zeros((V, 1, T), dtype=np.float32)
    gate[:, 0, 5:T - 120] = 1.0
    g = torch.from_numpy(np.ascontiguousarray(gate.transpose(1, 2, 0))).cuda()
    o1 = b.process_events(T1, g[:, :T1].contiguous())
    o2 = b.process_events(T2, g[:, T1:].contiguous())
    torch.cuda.synchronize()
    got = torch.cat([o1, o2], dim=1).cpu().numpy().transpose(2, 0, 1)
    from test_gpu_config4 import config4_oracle_voice

    seq = O.Sequencer(1, 2, SR)
    for v in range(V):
        seq.push(start[v], end[v], int(fade[v]), fin[v], fout[v], config4_oracle_voice(p, v, adsr))
    _, per = seq.render(T, True, inputs=gate)
    for v in range(0, V, 5):
        assert_bit_equal(got[v], per[v], f"gated event voice {v}")


def test_event_validation(gpu):
    b = gpu.Bank("sine_hz", 8)
    with pytest.raises(gpu.FdspError):
        b.set_events(0.0, 0.01, fade_in=0.02)              # fade longer than the event (Sequencer::push asserts)
    with pytest.raises(gpu.FdspError):
        b.process_events(64)                               # no events set
    b.set_events(np.zeros(8), np.full(8, 0.01), fade=FADE_POWER)
    b.process_events(64)
    b.events_rewind(0.0)
    assert b.events_time() == 0.0

"""Invariants of the oracle's Sequencer restatement (oracle/o_sequencer.c <- src/sequencer.rs): event placement on the
sample grid, fade curves, equal-amplitude / equal-power crossfades, tick vs process, units starting at their own start."""
import numpy as np

import oracle as O

SR = 48000.0


def test_event_placement_and_process_tick_agreement():
    for start_s, end_s in ((0, 300), (37, 37 + 200), (64, 191), (100, 101), (5, 64 * 3 + 9)):
        outs = []
        for process in (True, False):
            s = O.Sequencer(0, 1, SR)
            s.push(start_s / SR, end_s / SR, O.FADE_SMOOTH, 0.0, 0.0, O.constant(1.0))
            mix, per = s.render(64 * 4 + 20, process)
            want = np.zeros(64 * 4 + 20, dtype=np.float32)
            want[start_s:end_s] = 1.0
            assert np.array_equal(mix[0], want), (start_s, end_s, process)
            assert np.array_equal(per[0, 0], want)
            outs.append(mix)
        assert np.array_equal(outs[0], outs[1])
        assert abs(s.time() - (64 * 4 + 20) / SR) < 1e-12


def test_fades_are_monotone_and_tick_close_to_process():
    T, a, b = 64 * 20, 100, 1100
    for ease in (O.FADE_SMOOTH, O.FADE_POWER):
        res = []
        for process in (True, False):
            s = O.Sequencer(0, 1, SR)
            s.push(a / SR, b / SR, ease, 300 / SR, 200 / SR, O.constant(1.0))
            res.append(s.render(T, process)[0][0])
        y = res[0]
        assert np.all(y[:a] == 0) and np.all(y[b:] == 0)
        assert np.all(np.diff(y[a:a + 300]) >= -1e-6) and np.all(np.diff(y[b - 200:b]) <= 1e-6)
        assert abs(y[a + 150] - (0.5 if ease == O.FADE_SMOOTH else np.sin(np.pi / 4))) < 2e-2
        assert np.allclose(y[a + 300:b - 200], 1.0, atol=1e-6)
        assert np.max(np.abs(res[0] - res[1])) < 2e-4      # block-accumulated vs per-sample fade phase


def test_crossfades():
    T, x0, n = 64 * 12, 200, 256
    for ease, combine in ((O.FADE_SMOOTH, lambda u, v: u + v), (O.FADE_POWER, lambda u, v: u * u + v * v)):
        s = O.Sequencer(0, 1, SR)
        s.push(0.0, (x0 + n) / SR, ease, 0.0, n / SR, O.constant(1.0))
        s.push(x0 / SR, T / SR, ease, n / SR, 0.0, O.constant(1.0))
        _, per = s.render(T, True)
        c = combine(per[0, 0, x0:x0 + n].astype(np.float64), per[1, 0, x0:x0 + n].astype(np.float64))
        assert np.max(np.abs(c - 1.0)) < (1e-5 if ease == O.FADE_SMOOTH else 4e-3)   # Bhaskara sine: ~0.2 % off


def test_units_start_at_their_own_start_and_mix_is_the_sum():
    T = 64 * 6 + 11
    starts = [0, 13, 64, 150]
    s = O.Sequencer(0, 1, SR)
    refs = []
    for i, st in enumerate(starts):
        n = O.sine_hz(440.0 * (i + 1))
        n.set_seed(i + 1)
        s.push(st / SR, (T - 5) / SR, O.FADE_SMOOTH, 0.0, 0.0, n)
        r = O.sine_hz(440.0 * (i + 1))
        r.set_sample_rate(SR)
        r.set_seed(i + 1)
        refs.append(r)
    mix, per = s.render(T, False)                       # tick path: each unit ticks from its start sample
    for i, st in enumerate(starts):
        want = refs[i].render_ticks(length=T - 5 - st)[0]
        assert np.array_equal(per[i, 0, st:T - 5], want)
    assert np.allclose(mix[0], per[:, 0].sum(axis=0), atol=1e-6)

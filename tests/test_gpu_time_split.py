"""The time-split kernel (fd_device.hpp k_render_ts): banks small enough to leave most SIMDs idle -- the 2-, 4-, 8-GPU
shards of the 65 536-voice metric -- render a three-stage chain with its oscillator stages split over TIME as well (two
waves per stage, each advancing the phase through the whole block but evaluating the sine for its own half).  Same
arithmetic per frame as every other kernel: bit-exact against the oracle, against the pipeline kernel, across launches,
in tolerance mode, and through the rollback path (a phase of -0.0)."""
import numpy as np
import pytest

import oracle as O
from fundsp_amd import LAYOUT_VOICE_MINOR, MATH_FAST, MODE_PROCESS
from fundsp_amd import workloads as W
from test_gpu_parity import assert_bit_equal, run_bank

pytestmark = pytest.mark.gpu
SR = 48000.0


@pytest.fixture
def time_split(gpu):
    def set_(v):
        assert gpu.lib().fdsp_set_option(b"time_split", v) == 0
    yield set_
    set_(1)


def test_time_split_equals_oracle_and_pipeline_kernel(gpu, time_split):
    V, T = 64 * 10 + 37, 64 * 24          # ragged last voice group; three launches of 8 blocks each
    p = W.fm_svf_params(V, SR)
    want, _ = O.bank_render(3, [p["f"], p["m"], p["fc"], p["q"]], p["seed"], T, SR, True, 0, 8)   # [voice][frame]
    for split in (1, 0):
        time_split(split)
        b = W.make_fm_svf_bank(V, SR, params=p)
        got = np.concatenate([run_bank(b, None, T // 3, LAYOUT_VOICE_MINOR, MODE_PROCESS)[:, 0, :] for _ in range(3)], axis=1)
        assert_bit_equal(got, want, f"time_split={split}, three consecutive launches")
        # every kernel renders the same samples, so ask which one ran: 4 = time-split, 2 = pipeline
        assert b.get_option("last_kernel") == (4 if split else 2)
    # the option per BANK (process-wide value untouched): two banks side by side, one on each kernel
    time_split(1)
    b1, b0 = W.make_fm_svf_bank(V, SR, params=p), W.make_fm_svf_bank(V, SR, params=p)
    b0.set_option("time_split", 0)
    g1 = run_bank(b1, None, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)[:, 0, :]
    g0 = run_bank(b0, None, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)[:, 0, :]
    assert (b1.get_option("last_kernel"), b0.get_option("last_kernel")) == (4, 2)
    assert_bit_equal(g1, want, "per-bank time_split=default")
    assert_bit_equal(g0, want, "per-bank time_split=0")
    b0.set_option("time_split", -1)          # back to the process-wide default
    b0.reset(); b0.set_seed(p["seed"])
    run_bank(b0, None, 64 * 4, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    assert b0.get_option("last_kernel") == 4
    b0.set_option("pipe_split", 0)
    run_bank(b0, None, 64 * 4, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    assert b0.get_option("last_kernel") == 1  # single-wave kernel
    # a launch that is not a multiple of 64 frames falls back to the pipeline kernel and continues the same state
    time_split(1)
    b = W.make_fm_svf_bank(V, SR, params=p)
    a = run_bank(b, None, 64 * 8, LAYOUT_VOICE_MINOR, MODE_PROCESS)[:, 0, :]
    c = run_bank(b, None, 64 * 4 + 13, LAYOUT_VOICE_MINOR, MODE_PROCESS)[:, 0, :]
    want2, _ = O.bank_render(3, [p["f"], p["m"], p["fc"], p["q"]], p["seed"], 64 * 12 + 13, SR, True, 0, 8)
    assert_bit_equal(np.concatenate([a, c], axis=1), want2, "time-split launch followed by a ragged launch")


def test_time_split_two_groups_per_cu_layout(gpu, time_split):
    """Banks of more than one voice group per CU (the 2-GPU shard of the headline) take the 14-wave workgroup of the
    three-way time split (two groups per workgroup; ragged last workgroup: an odd number of groups plus a ragged group).
    Against the pipeline kernel over the whole bank and against the oracle on voices of the first, a middle and the last
    workgroup; round 2's 2 + 1 + 1 layout (time_split = 2) must agree as well."""
    V, T = 64 * 301 + 5, 64 * 6
    p = W.fm_svf_params(V, SR)
    outs = {}
    for split in (1, 2, 0):
        time_split(split)
        b = W.make_fm_svf_bank(V, SR, params=p)
        outs[split] = run_bank(b, None, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)[:, 0, :]
        assert b.get_option("last_kernel") == (4 if split else 2)
    assert_bit_equal(outs[1], outs[0], "3 + 3 + 1 time split == pipeline kernel, 302 voice groups")
    assert_bit_equal(outs[2], outs[0], "2 + 1 + 1 time split == pipeline kernel")
    pick = np.array([0, 63, 64, 127, 64 * 150 + 17, 64 * 300, V - 6, V - 1])
    want, _ = O.bank_render(3, [p["f"][pick], p["m"][pick], p["fc"][pick], p["q"][pick]], p["seed"][pick], T, SR, True, 0, 4)
    assert_bit_equal(outs[1][pick], want, "time split vs oracle")


def test_time_split_odd_group_count_leaves_a_dead_second_group(gpu, time_split):
    """64 * 300 + 5 voices = 301 voice groups (> one per CU): the last 14-wave workgroup holds ONE live group; the waves of its second
    group (v0 == stride: `live` false in render_ts3_body<G, 2>) load and store nothing and only take part in the barriers
    (the 302-group case above never reached that branch)."""
    V, T = 64 * 300 + 5, 64 * 5
    p = W.fm_svf_params(V, SR)
    outs = {}
    for split in (1, 0):
        time_split(split)
        b = W.make_fm_svf_bank(V, SR, params=p)
        outs[split] = run_bank(b, None, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)[:, 0, :]
        assert b.get_option("last_kernel") == (4 if split else 2)
    assert_bit_equal(outs[1], outs[0], "3 + 3 + 1 time split == pipeline kernel, 301 voice groups")
    pick = np.array([0, 64 * 299 + 63, 64 * 300, V - 1])
    want, _ = O.bank_render(3, [p["f"][pick], p["m"][pick], p["fc"][pick], p["q"][pick]], p["seed"][pick], T, SR, True, 0, 4)
    assert_bit_equal(outs[1][pick], want, "time split vs oracle, last workgroup")


def test_time_split_rollback_path_and_negative_frequencies(gpu, time_split):
    """A modulator / carrier phase of exactly -0.0 at a block start sends that block down the packed path's rollback
    (Sine::begin_block), in the split waves too; huge modulation indices trip the |quadrant| < 8192 guard mid-block."""
    V, T = 64 * 3, 64 * 6
    p = W.fm_svf_params(V, SR)
    p["m"][5] = np.float32(2.0e6)         # carrier frequency input ~ 1e9 Hz: phases run past the guard inside a block
    p["f"][9] = np.float32(-440.0)        # negative frequencies: phases decrease
    outs = []
    for split in (1, 0):
        time_split(split)
        b = W.make_fm_svf_bank(V, SR, params=p)
        for slot in (W.FM_SLOTS["mod_phase"], W.FM_SLOTS["car_phase"]):
            ph = b.get_slot(slot)
            ph[7] = np.float32(-0.0)
            b.set_param(slot, ph)
        with np.errstate(all="ignore"):
            outs.append(run_bank(b, None, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)[:, 0, :])
    assert_bit_equal(outs[0], outs[1], "time-split == pipeline kernel on rollback blocks")
    assert np.isfinite(outs[0][9]).all()


def test_time_split_in_tolerance_mode(gpu, time_split):
    V, T = 64 * 4, 64 * 16
    p = W.fm_svf_params(V, SR)
    outs = []
    for split in (1, 0):
        time_split(split)
        b = W.make_fm_svf_bank(V, SR, params=p)
        b.set_option("math", MATH_FAST)
        outs.append(run_bank(b, None, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)[:, 0, :])
    assert_bit_equal(outs[0], outs[1], "FDSP_MATH_FAST: time-split == pipeline kernel")
    want, _ = O.bank_render(3, [p["f"], p["m"], p["fc"], p["q"]], p["seed"], T, SR, True, 0, 4)
    assert np.max(np.abs(outs[0] - want)) < 1e-4 and not np.array_equal(outs[0], want)


@pytest.mark.parametrize("V", [64 * 10 + 37, 64 * 300 + 5])
def test_real_time_blocks_take_the_time_split_and_pipeline_kernels(gpu, time_split, V):
    """Launches of ONE to three 64-frame blocks (a real-time host's AudioNode::process calls): small banks of oscillator chains take the
    time-split kernels from one block on, every other launch of a chain worth cutting the stage pipeline (fd_engine.hpp FD_TS_MIN_T,
    fd_device.hpp PipeMinT), shorter launches the single-wave kernel -- one state, one stream of samples whichever kernel ran: a run of
    mixed launch lengths equals the oracle's continuous render bit for bit (one group per CU and the 14-wave two-group layout)."""
    time_split(1)
    p = W.fm_svf_params(V, SR)
    lengths = [64, 64, 128, 192, 64, 256, 128, 16, 37]   # (process semantics: a launch is cut into blocks of 64 from ITS first frame, so only
    aligned = sum(lengths[:-1])                            #  launches that start on a multiple of 64 continue the oracle's one long render)
    want, _ = O.bank_render(3, [p["f"], p["m"], p["fc"], p["q"]], p["seed"], aligned, SR, True, 0, 8)
    sel = [0, 1, 63, 64, V // 2, V - 38, V - 1]
    b = W.make_fm_svf_bank(V, SR, params=p)
    ref = W.make_fm_svf_bank(V, SR, params=p)
    ref.set_option("pipe_split", 0)  # the single-wave kernel at every length
    got, families = [], []
    for T in lengths:
        got.append(run_bank(b, None, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)[:, 0, :])
        families.append(b.get_option("last_kernel"))
        assert_bit_equal(got[-1], run_bank(ref, None, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)[:, 0, :], f"T = {T}: family {families[-1]} vs the single-wave kernel")
    assert families == [4, 4, 4, 4, 4, 4, 4, 1, 1], families
    assert_bit_equal(np.concatenate(got[:-1], axis=1)[sel], want[sel], "mixed launch lengths vs the oracle's continuous render")
    # a bank too large for the time split: the stage pipeline from one block on, the single wave below
    time_split(0)
    b = W.make_fm_svf_bank(V, SR, params=p)
    got, families = [], []
    for T in lengths:
        got.append(run_bank(b, None, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)[:, 0, :])
        families.append(b.get_option("last_kernel"))
    assert families == [2, 2, 2, 2, 2, 2, 2, 1, 1], families
    assert_bit_equal(np.concatenate(got[:-1], axis=1)[sel], want[sel], "pipeline kernel from one block on vs the oracle")


def test_planar_real_time_blocks_take_the_planar_pipeline(gpu):
    """The reference's own buffer shape ([voice][channel][frames], planar) at real-time launch lengths: from 16 frames on a launch takes
    the planar pipeline (loader / stages / storer waves, family 3), below that the single-wave kernel -- same samples either way, with and
    without a graph input (config 3's voices, config 4's gated voices), process and tick semantics."""
    from fundsp_amd import LAYOUT_PLANAR, MODE_TICK
    import fundsp_amd as F

    F.wavetable_build("saw")
    V = 64 * 5 + 9
    lengths = [64, 16, 128, 8, 192, 40, 64]
    for make, ni in ((W.make_fm_svf_bank, 0), (W.make_saw_moog_bank, 1)):
        for mode in (MODE_PROCESS, MODE_TICK):
            b, ref = make(V, SR), make(V, SR)
            ref.set_option("pipe_split", 0)  # the single-wave kernel at every length
            families = []
            for k, T in enumerate(lengths):
                x = None
                if ni:
                    x = np.zeros((V, ni, T), dtype=np.float32)
                    x[:, 0, :] = 1.0 if k % 3 else 0.0   # gate edges between the launches
                    x[:, 0, T // 2:] = 1.0
                got = run_bank(b, x, T, LAYOUT_PLANAR, mode)
                families.append(b.get_option("last_kernel"))
                assert_bit_equal(got, run_bank(ref, x, T, LAYOUT_PLANAR, mode), f"{make.__name__} mode {mode} T = {T}: family {families[-1]} vs the single-wave kernel")
                assert ref.get_option("last_kernel") == 1
            assert families == [3, 3, 3, 1, 3, 3, 3], families


@pytest.mark.parametrize("name", ["modulated_svf", "saw_filter_env"])
def test_run_time_compiled_graphs_at_real_time_launch_lengths(gpu, name):
    """Run-time compiled graphs carry their own launch-length threshold (the meta block's PipeMinT): short launches in both layouts equal
    the single-wave kernel's samples bit for bit, and the families are the ahead-of-time kinds' (2 / 3 = voice-minor / planar pipeline)."""
    from fundsp_amd import LAYOUT_PLANAR
    from fundsp_amd import graph as G
    from test_gpu_jit import GRAPHS
    from test_gpu_parity import noise_input

    build, ni, ring = GRAPHS[name]
    g = build(G)
    V = 64 * 4 + 11
    for kind in G.uses_wavetables(g):
        t = O.Wavetable.get(kind)
        offs = np.concatenate([[0], np.cumsum(t.lengths)])
        gpu.wavetable_upload(kind, t.pitches, [t.data[offs[i]:offs[i + 1]] for i in range(len(t.lengths))])
    lengths = [64, 16, 128, 8, 200]
    for layout, fam in ((LAYOUT_VOICE_MINOR, 2), (LAYOUT_PLANAR, 3)):
        b = gpu.Bank.from_graph(g, V, ring_frames=ring, sample_rate=SR)
        b.set_seed(np.arange(V, dtype=np.uint64) * 31 + 7)
        ref = b.clone()
        ref.set_option("pipe_split", 0)
        families = []
        for k, T in enumerate(lengths):
            x = noise_input(V, ni, T, seed=100 + k) if ni else None
            got = run_bank(b, x, T, layout, MODE_PROCESS)
            families.append(b.get_option("last_kernel"))
            assert_bit_equal(got, run_bank(ref, x, T, layout, MODE_PROCESS), f"{name} layout {layout} T = {T}: family {families[-1]} vs the single-wave kernel")
        want = [fam, fam if layout == LAYOUT_PLANAR else 1, fam, 1, fam]
        assert families == want or all(f == 1 for f in families), (families, want)   # (a graph without a stage plan stays on the single wave)


@pytest.mark.parametrize("V,T", [(1024, 64 * 12), (64 * 5 + 9, 64), (64 * 301 + 5, 64 * 3)])
def test_config2_biquad_split_at_its_seam(gpu, time_split, V, T):
    """BASELINE config 2 (`noise() >> biquad`, 1024 voices): the DF1 biquad is a chain of two stages cut at the seam of its own
    expression (biquad.rs:186-188: `(b0*x0 + b1*x1) + b2*x2` | `(p - a1*y1) - a2*y2`), so behind the counter-based Noise the chain has
    three stages and a small bank takes the three-way time-split kernel: noise and the feed-forward half in three waves each, the
    serial wave carries the recurrence alone.  Every kernel family renders the oracle's samples: time split (4), stage pipeline (2:
    noise | whole biquad with the packed feed-forward half), single wave (1); one group per CU, a ragged lone block, two per workgroup."""
    p = W.noise_biquad_params(V, SR)
    pick = np.unique(np.concatenate([[0, 63, 64 % V, V - 1], np.random.default_rng(V).integers(0, V, 12)]))
    want, _ = O.bank_render(2, [p["fc"][pick], p["q"][pick]], p["seed"][pick], 2 * T + 13, SR, True, 0, 4)      # [voice][frame]
    outs = {}
    for name, opts, family in (("time split", {}, 4), ("pipeline", {"time_split": 0, "pipe_split": 2}, 2), ("single wave", {"pipe_split": 0}, 1)):
        b = W.make_noise_biquad_bank(V, SR, params=p)
        for k, v in opts.items():
            b.set_option(k, v)
        a = run_bank(b, None, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)[:, 0, :]
        assert b.get_option("last_kernel") == family, name
        c = run_bank(b, None, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)[:, 0, :]      # the state carried into a second launch
        d = run_bank(b, None, 13, LAYOUT_VOICE_MINOR, MODE_PROCESS)[:, 0, :]     # ... and into a ragged one (pipeline / single wave)
        outs[name] = np.concatenate([a, c, d], axis=1)
        assert_bit_equal(outs[name][pick], want, f"config 2, {name} vs oracle")
    assert_bit_equal(outs["time split"], outs["single wave"], "time split == single wave, every voice")
    assert_bit_equal(outs["pipeline"], outs["single wave"], "pipeline == single wave, every voice")

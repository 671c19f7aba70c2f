"""Synthetic workloads of BASELINE.json's configs: per-voice parameters derived from the
voice index with the reference's own indexed RNG `rnd1` (src/math.rs:569-576), so that any two implementations
(the HIP engine, the CPU oracle, a future Rust harness) build identical banks from the voice index alone.
Pure numpy: no device, no oracle.
"""
import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def rnd1(x):
    """rnd1 (math.rs:569-576), vectorised over uint64 -> float64 in [0, 1)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x ^ np.uint64(0x5555555555555555)
        x = x * np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return (x >> np.uint64(11)).astype(np.float64) * (1.0 / float(1 << 53))


def hash1(x):
    """hash1 (math.rs:592-599), vectorised over uint64."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x ^ np.uint64(0x5555555555555555)
        x = x * np.uint64(0x517CC1B727220A95)
        x = (x ^ (x >> np.uint64(32))) * np.uint64(0xD6E8FEB86659FD93)
        x = (x ^ (x >> np.uint64(32))) * np.uint64(0xD6E8FEB86659FD93)
        x = x ^ (x >> np.uint64(32))
    return x


def fm_svf_params(voices, sample_rate=48000.0, voice0=0):
    """Config 3: sine_hz(f) * f * m + f >> sine() >> lowpass_hz(fc, q), one parameter set per voice.

    f in [55, 1760) Hz log-uniform, m in [0.5, 8), fc = f * 2^[0,4) clamped below 0.45*sr, q in [0.5, 4);
    u_k = rnd1(4 v + k); oscillator phases from set_seed(v).
    """
    v = np.arange(voice0, voice0 + voices, dtype=np.uint64)
    u = [rnd1(np.uint64(4) * v + np.uint64(k)) for k in range(4)]
    f = 55.0 * np.exp2(5.0 * u[0])
    m = 0.5 + 7.5 * u[1]
    fc = np.minimum(f * np.exp2(4.0 * u[2]), 0.45 * sample_rate)
    q = 0.5 + 3.5 * u[3]
    return dict(f=f.astype(np.float32), m=m.astype(np.float32), fc=fc.astype(np.float32), q=q.astype(np.float32),
                seed=v.copy())


def noise_biquad_params(voices, sample_rate=48000.0, voice0=0):
    """Config 2: white noise through a per-voice lowpass biquad (one BiquadBank<f32x8> lane per voice).

    fc = 20 * 1000^u Hz (20 Hz - 20 kHz log-uniform), q = 0.5 * 20^u' (0.5 - 10), u = rnd1(2v), u' = rnd1(2v+1);
    noise seed = hash1(v) (Noise::reset: state = lo32(h ^ h >> 32), noise.rs:192-195).
    """
    v = np.arange(voice0, voice0 + voices, dtype=np.uint64)
    fc = 20.0 * np.power(1000.0, rnd1(np.uint64(2) * v))
    q = 0.5 * np.power(20.0, rnd1(np.uint64(2) * v + np.uint64(1)))
    fc = np.minimum(fc, 0.49 * sample_rate)
    return dict(fc=fc.astype(np.float32), q=q.astype(np.float32), seed=hash1(v))


# slot names of the fused graphs (see fdsp_kind_slot_name / include/fundsp_hip.h)
FM_SLOTS = dict(
    f_const="0.0.0.0.0.0:value[0]",  # constant(f) feeding the modulator
    f_mul="0.0.0.0:scalar",          # * f
    m_mul="0.0.0:scalar",            # * m
    f_add="0.0:scalar",              # + f
    cutoff="1:cutoff", q="1:q", gain="1:gain", mode="1:mode",
    mod_phase="0.0.0.0.0.1:phase", car_phase="0.1:phase",
)


def make_fm_svf_bank(voices, sample_rate=48000.0, voice0=0, params=None):
    """Build the config-3 bank on the current device, exactly like constructing each voice's graph in Rust,
    calling set_sample_rate(sr) and set_seed(v)."""
    from .bank import Bank

    p = params or fm_svf_params(voices, sample_rate, voice0)
    b = Bank("fm_svf", voices)
    b.set_param(FM_SLOTS["f_const"], p["f"])
    b.set_param(FM_SLOTS["f_mul"], p["f"])
    b.set_param(FM_SLOTS["m_mul"], p["m"])
    b.set_param(FM_SLOTS["f_add"], p["f"])
    b.set_param(FM_SLOTS["cutoff"], p["fc"])
    b.set_param(FM_SLOTS["q"], p["q"])
    b.set_sample_rate(sample_rate)
    b.set_seed(p["seed"])
    return b


def make_noise_biquad_bank(voices, sample_rate=48000.0, voice0=0, params=None, kind="noise_biquad"):
    from .bank import Bank, biquad_coefs

    p = params or noise_biquad_params(voices, sample_rate, voice0)
    coefs = np.stack([biquad_coefs("lowpass", sample_rate, float(fc), float(q)) for fc, q in zip(p["fc"], p["q"])])
    b = Bank(kind, voices)
    for i, n in enumerate(("a1", "a2", "b0", "b1", "b2")):
        b.set_param(f"1:{n}", coefs[:, i])
    b.set_sample_rate(sample_rate)
    b.set_param("0:has_seed", 1.0)
    b.set_param_u64("0:seed", p["seed"])
    b.reset()  # Noise::reset picks the seed up (noise.rs:192-195); filter state is already zero
    return b


def saw_moog_params(voices, sample_rate=48000.0, voice0=0):
    """Config 4 voice: ((dc(f) >> saw() | dc(fc) | dc(q)) >> moog()) * adsr_live(0.01, 0.1, 0.6, 0.2) >> pan(p).

    f in [55, 1760) Hz log-uniform, fc = f * 2^[1,5) clamped below 0.4*sr, q in [0.1, 0.7), pan in [-1, 1);
    u_k = rnd1(4 v + k); oscillator phase from set_seed(v).
    """
    v = np.arange(voice0, voice0 + voices, dtype=np.uint64)
    u = [rnd1(np.uint64(4) * v + np.uint64(k)) for k in range(4)]
    f = 55.0 * np.exp2(5.0 * u[0])
    fc = np.minimum(f * np.exp2(1.0 + 4.0 * u[1]), 0.4 * sample_rate)
    q = 0.1 + 0.6 * u[2]
    pan = -1.0 + 2.0 * u[3]
    return dict(f=f.astype(np.float32), fc=fc.astype(np.float32), q=q.astype(np.float32), pan=pan.astype(np.float32),
                seed=v.copy())


def gate_signal(frames, sample_rate=48000.0, on_frame=1, off_seconds=0.5):
    """adsr_live only attacks on a low->high gate transition (adsr.rs:37-43): low at frame 0, high until
    `off_seconds`, then low (release)."""
    g = np.zeros(frames, dtype=np.float32)
    g[on_frame:min(frames, int(off_seconds * sample_rate))] = 1.0
    return g


C4_SLOTS = dict(f="0.0.0.0.0.0:value[0]", fc="0.0.0.0.1:value[0]", q="0.0.0.1:value[0]", pan="1:pan",
                attack="0.1:attack", decay="0.1:decay", sustain="0.1:sustain", release="0.1:release")


def make_saw_moog_bank(voices, sample_rate=48000.0, voice0=0, params=None, adsr=(0.01, 0.1, 0.6, 0.2)):
    from .bank import Bank

    p = params or saw_moog_params(voices, sample_rate, voice0)
    b = Bank("saw_moog_adsr_pan", voices)
    for k in ("f", "fc", "q", "pan"):
        b.set_param(C4_SLOTS[k], p[k])
    for k, val in zip(("attack", "decay", "sustain", "release"), adsr):
        b.set_param(C4_SLOTS[k], float(val))
    b.set_sample_rate(sample_rate)
    b.set_seed(p["seed"])
    return b


# ---- config 4 in the reference's own gate shape: var(gate) >> adsr_live (examples/live_adsr.rs:72) ----------------------------------
C4V_SLOTS = dict(f="0.0.0.0.0.0:value[0]", fc="0.0.0.0.1:value[0]", q="0.0.0.1:value[0]", pan="1:pan", gate="0.1.0:value",
                 attack="0.1.1:attack", decay="0.1.1:decay", sustain="0.1.1:sustain", release="0.1.1:release")


def gate_plan(frames, sample_rate=48000.0, off_seconds=0.5, block=64):
    """One note per `frames`: [(gate, frames), ...] -- the gate high for `off_seconds` (rounded down to whole blocks), then low.
    The value of a `var(..)` is read once per 64-sample block (Var::process, shared.rs:122-125), so a host that changes it does so
    BETWEEN blocks: a plan is a list of launches with the variable set before each."""
    on = min(frames, int(off_seconds * sample_rate) // block * block)
    return [(g, n) for g, n in ((1.0, on), (0.0, frames - on)) if n > 0]


def make_saw_moog_var_bank(voices, sample_rate=48000.0, voice0=0, params=None, adsr=(0.01, 0.1, 0.6, 0.2), prime=True):
    """Config 4 voice with the gate as a shared variable, the shape of the reference's live_adsr example:
    ((dc(f) >> saw() | dc(fc) | dc(q)) >> moog()) * (var(gate) >> adsr_live(a, d, s, r)) >> pan(p); same parameters as
    saw_moog_params.  The graph has NO input: the gate lives in the per-voice slot C4V_SLOTS["gate"] (fdsp_bank_set_param* plays
    Shared::set_value).  `prime`: render one block with the gate low, as in the reference example where the control starts at 0.0 while
    audio already runs -- adsr_live attacks on a low -> high change only (adsr.rs:37-43)."""
    from .bank import Bank

    p = params or saw_moog_params(voices, sample_rate, voice0)
    b = Bank("saw_moog_var_adsr_pan", voices)
    for k in ("f", "fc", "q", "pan"):
        b.set_param(C4V_SLOTS[k], p[k])
    for k, val in zip(("attack", "decay", "sustain", "release"), adsr):
        b.set_param(C4V_SLOTS[k], float(val))
    b.set_param(C4V_SLOTS["gate"], 0.0)
    b.set_sample_rate(sample_rate)
    b.set_seed(p["seed"])
    if prime:
        b.process(64)
    return b

"""The reference's OWN benchmark graphs (benches/benchmark.rs, a criterion harness: each bench renders 1 s of ONE graph at 44.1 kHz
with Wave::render on one thread) in both notations -- the engine's (fundsp_amd.graph) and the oracle's (tests/oracle.py).

Nine of the thirteen benches are graphs of nodes on the path and are here; the other four are not:
`resynth` (FFT node), `chorus` (closures over the `funutd` crate's hashes, crate source absent), `wrap` / `netpass` (the `Net`
executor: the control plane, not the arithmetic -- `pass` is the same graph statically typed).

Used by tests/test_gpu_criterion.py (bit parity of a bank of instances against the oracle, per instance) and by
`bench.py --criterion` (the bank's rate next to the oracle's rate for ONE instance on one host thread = what criterion times).
Test / bench infrastructure like tests/oracle.py: the product does not import it."""
import numpy as np

SAMPLE_RATE = 44100.0      # benchmark.rs: Wave::render(44100.0, 1.0, ..)
SECONDS = 1.0
FRAMES = 44100             # Wave::render: (duration * sample_rate).round() frames in blocks of 64 + a remainder of 4 (wave.rs:441-470)

# the two closures as device functors (contract: Envelope<FN>, fd_nodes.hpp); their host twins are the oracle's o_envfn_criterion_* (fundsp_oracle.c)
ENVELOPE_SRC = ("struct EnvCriterionEnvelope { static constexpr int OUT = 1; template <class V> FD_HD void visit(V&) {} FD_HD void init() {} "
                "FD_HD void eval(float t, float* out) const { out[0] = expf_musl(-t) * sinf_musl(t * 1.0f * F32_TAU); } };")     # benchmark.rs:57
PHASER_SRC = ("struct EnvCriterionPhase { static constexpr int OUT = 1; template <class V> FD_HD void visit(V&) {} FD_HD void init() {} "
              "FD_HD void eval(float t, float* out) const { out[0] = sinf_musl(t * 0.1f * F32_TAU) * 0.5f + 0.5f; } };")           # benchmark.rs:94


def _is_engine(m):
    return m.__name__.endswith("graph")


def _db_amp_f32(O, db):
    """db_amp::<f32>(db) = exp10(db / 20) = (x * LN_10 as f32).exp()   (math.rs:76-78, 294-296), with the oracle's libm expf"""
    f = np.float32
    return float(O.m_expf(f(db) / f(20.0) * f(2.302585092994046)))


def table(m, O):
    """name -> (graph in notation `m`, ring_frames the engine's bank needs, benchmark.rs line).  `O` = the oracle module (its libm gives the
    f32 constants both notations must agree on)."""
    eng = _is_engine(m)
    gain = _db_amp_f32(O, 3.0)
    if eng:
        env = m.envelope("EnvCriterionEnvelope", ENVELOPE_SRC)
        phaser = m.phaser(0.5, "EnvCriterionPhase", PHASER_SRC)
    else:
        env = m.envelope_c("o_envfn_criterion_envelope")
        phaser = m.phaser_lfo(0.5, m.envelope_c("o_envfn_criterion_phaser"))
    return {
        "sine": (m.sumi(100, lambda i: m.sine_hz(float(np.float32(100.0) * np.float32(i + 1)))), 0, 4),                    # :4-10
        "pass": (m.dc(1.0, 2.0) * 2.0 >> m.pass_() + m.pass_() >> m.pass_(), 0, 25),                                       # :25-31
        "wavetable": (m.saw_hz(110.0), 0, 50),                                                                              # :50-52
        "envelope": (m.noise() * env, 0, 54),                                                                               # :54-60
        "oversample": (m.noise() >> m.oversample(m.pass_()), 0, 62),                                                        # :62-64
        "equalizer": (m.noise() >> m.pipei(10, lambda i: m.bell_hz(float(np.float32(1000.0) + np.float32(1000.0) * np.float32(i)), 1.0, gain)), 0, 70),  # :70-77
        "reverb": ((m.noise() | m.noise()) >> m.reverb_stereo(10.0, 1.0, 0.5), 4096, 79),                                   # :79-85
        "limiter": (m.noise() >> m.limiter(0.1, 1.0), 16384, 87),                                                            # :87-89
        "phaser": (m.noise() >> phaser, 0, 91),                                                                             # :91-97
    }


NOT_ON_THE_PATH = {
    "resynth": "FFT node (resynth.rs), out of scope (DESIGN 8)",
    "chorus": "closures over the funutd crate's hashes (prelude.rs:2669); crate source absent from the reference tree",
    "wrap": "Net::wrap of the `pass` graph: the dynamic executor, not the arithmetic",
    "netpass": "Net front-end of the `pass` graph: the dynamic executor, not the arithmetic",
}

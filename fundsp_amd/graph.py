"""FunDSP graph notation on the host, lowered to a run-time compiled voice-bank kernel.

This mirrors the part of the reference's `prelude32` + `combinator.rs` surface that describes a *voice*: the same opcode
names (`sine_hz`, `lowpass_hz`, `moog`, `saw`, `adsr_live`, `pan`, ...) and the same operators (`>>` Pipe, `|` Stack,
`*`/`+`/`-` between nodes = Binop, with numbers = Unop; Python's precedence of these operators matches Rust's), so a
graph reads like its Rust original:

    g = sine_hz(f) * f * m + f >> sine() >> lowpass_hz(fc, q)          # README.md:98-103
    bank = Bank.from_graph(g, voices=65536)

Every numeric argument may be a scalar (all voices) or a length-`voices` array (one value per voice).  A graph object is
only a description: `type` is the combinator type FunDSP would build, spelled with the device templates of
fundsp_amd/csrc/fd_nodes.hpp, and `params` the per-node parameter values addressed the way the C ABI names slots
("<path>:<field>").  The arithmetic runs on the GPU through the C ABI (fdsp_graph_compile + fdsp_bank_*).
"""
import hashlib

import numpy as np

from .bank import BQ_KINDS, SVF_MODES

SHAPES = dict(clip=0, clip_to=1, tanh=2, atan=3, softsign=4, crush=5, soft_crush=6, adaptive_tanh=7)
SHAPES.update({f"adaptive_{k}": 8 + v for k, v in list(SHAPES.items())[:7] if k != "tanh"})  # Adaptive<S>, shape.rs:162-201


class Graph:
    __array_ufunc__ = None  # let `ndarray * Graph` reach Graph.__rmul__

    def __init__(self, type_str, nin, nout, params=(), rings=0, source=""):
        self.type, self.nin, self.nout, self.rings = type_str, nin, nout, rings
        self.params = list(params)  # (path tuple, field, value, is_u64)
        self.source = source        # C++ definitions (closure functors) the type refers to

    # --- combinators (combinator.rs:289-488)
    def _pair(self, other, tmpl, nin, nout):
        ps = [((0,) + p, f, v, u) for p, f, v, u in self.params] + [((1,) + p, f, v, u) for p, f, v, u in other.params]
        return Graph(f"{tmpl}<{self.type},{other.type}>", nin, nout, ps, self.rings + other.rings, _merge(self.source, other.source))

    def __rshift__(self, other):
        if self.nout != other.nin:
            raise TypeError(f"Pipe arity mismatch: {self.nout} outputs >> {other.nin} inputs")
        g = self._pair(other, "Pipe", self.nin, other.nout)
        g.pipe_parts = (self, other)   # Bank.from_graph renders `generator >> stock reverb / network` as a chain of two banks (bank.Chain)
        return g

    def __or__(self, other):
        return self._pair(other, "Stack", self.nin + other.nin, self.nout + other.nout)

    def __and__(self, other):  # Bus, combinator.rs `&`
        if self.nin != other.nin or self.nout != other.nout:
            raise TypeError("Bus arity mismatch")
        g = self._pair(other, "Bus", self.nin, self.nout)
        g.bus_parts = (self, other)    # bus_plan: `multipass() & wet * reverb` around a lane-per-frame bank (fdsp_bank_set_bus)
        return g

    def __xor__(self, other):  # Branch, combinator.rs `^`
        if self.nin != other.nin:
            raise TypeError("Branch arity mismatch")
        return self._pair(other, "Branch", self.nin, self.nout + other.nout)

    def __invert__(self):      # Thru, combinator.rs `!`
        return Graph(f"Thru<{self.type}>", self.nin, self.nin, [((0,) + p, f, v, u) for p, f, v, u in self.params],
                     self.rings, self.source)

    def _binop(self, other, op):
        if self.nout != other.nout:
            raise TypeError("Binop arity mismatch")
        ps = [((0,) + p, f, v, u) for p, f, v, u in self.params] + [((1,) + p, f, v, u) for p, f, v, u in other.params]
        return Graph(f"Binop<{op},{self.type},{other.type}>", self.nin + other.nin, self.nout, ps, self.rings + other.rings,
                     _merge(self.source, other.source))

    def _unop(self, u, scalar=None):
        ps = [((0,) + p, f, v, uu) for p, f, v, uu in self.params]
        if scalar is not None:
            ps.append(((), "scalar", scalar, False))
        g = Graph(f"Unop<{self.type},{u}>", self.nin, self.nout, ps, self.rings, self.source)
        g.unop_parts = (self, u, scalar)
        return g

    def __mul__(self, o): return self._binop(o, "OpMul") if isinstance(o, Graph) else self._unop("UMulScalar", o)
    def __rmul__(self, o): return self._unop("UMulScalar", o)
    def __add__(self, o): return self._binop(o, "OpAdd") if isinstance(o, Graph) else self._unop("UAddScalar", o)
    def __radd__(self, o): return self._unop("UAddScalar", o)
    def __sub__(self, o): return self._binop(o, "OpSub") if isinstance(o, Graph) else self._unop("UAddScalar", -np.asarray(o, dtype=np.float32))
    def __rsub__(self, o): return self._unop("UNegAddScalar", o)
    def __neg__(self): return self._unop("UNeg")

    # --- builders (combinator.rs:263-267)
    def _set(self, field, value, u64=False):
        self.params.append(((), field, value, u64))
        return self

    def phase(self, p):
        self._set("has_initial_phase", 1.0)
        return self._set("initial_phase", p)

    def seed(self, s):
        self._set("has_seed", 1.0)
        return self._set("seed", s, u64=True)

    def ring_frames(self, sample_rate):
        """Ring capacity (positions per ring) this graph needs at `sample_rate`, from its delay / tap / limiter parameters
        (delay.rs:108-110, 204-206, 440-442; dynamics.rs:159-171), or None when a node sizes its ring from data the graph
        does not hold (Pluck, Hold, Reverb3: pass ring_frames yourself)."""
        if self.rings == 0:
            return 0
        if any(k in self.type for k in ("Pluck", "Hold", "Reverb3")):
            return None
        sr = float(sample_rate)

        def pow2(n):
            return 1 << max(0, int(n) - 1).bit_length()

        need = 1
        cubic = "TapT<false" in self.type
        for _p, field, value, _u in self.params:
            v = float(np.max(np.asarray(value, dtype=np.float32)))
            if field == "time" and "Delay" in self.type:
                need = max(need, int(round(v * sr)) + 1)
            elif field == "max_delay":
                blen = np.ceil(np.float32(v) * np.float32(sr)) + (11 if cubic else 2)
                need = max(need, pow2(int(blen)))
            elif field == "attack_time" and "Limiter" in self.type:
                ln = max(1, int(round(sr * v)))
                need = max(need, pow2(ln) + ln + (ln & 1))
        return need

    def kind_name(self):
        return "jit_" + hashlib.sha1((self.type + "\0" + self.source).encode()).hexdigest()[:16]

    def slot_values(self):
        """[(slot name, value, is_u64)] in assignment order."""
        return [(".".join(str(i) for i in p) + ":" + f, v, u) for p, f, v, u in self.params]


def _merge(a, b):
    """Union of two source preludes (a definition must not appear twice in the compiled translation unit)."""
    a, b = a.strip(), b.strip()
    if not b or b in a:
        return a
    if not a or a in b:
        return b
    return a + "\n" + b


def _leaf(t, nin, nout, rings=0, **fields):
    return Graph(t, nin, nout, [((), k, v, False) for k, v in fields.items()], rings)


# --- prelude32 opcodes -------------------------------------------------------------------------------------------
def constant(*v): return Graph(f"Constant<{len(v)}>", 0, len(v), [((), f"value[{i}]", x, False) for i, x in enumerate(v)])
dc = constant
def pass_(): return _leaf("Pass", 1, 1)
def tick(n=1): return _leaf(f"Tick<{n}>", n, n)
def sine(): return _leaf("Sine", 1, 1)
def sine_hz(f): return constant(f) >> sine()
def noise(): return _leaf("Noise", 0, 1)
white = noise
def _fsvf(mode, f, q, gain=1.0): return _leaf("FixedSvf", 1, 1, mode=float(SVF_MODES[mode]), cutoff=f, q=q, gain=gain)
def lowpass_hz(f, q): return _fsvf("lowpass", f, q)
def highpass_hz(f, q): return _fsvf("highpass", f, q)
def bandpass_hz(f, q): return _fsvf("bandpass", f, q)
def notch_hz(f, q): return _fsvf("notch", f, q)
def peak_hz(f, q): return _fsvf("peak", f, q)
def allpass_hz(f, q): return _fsvf("allpass", f, q)
def bell_hz(f, q, gain): return _fsvf("bell", f, q, gain)
def lowshelf_hz(f, q, gain): return _fsvf("lowshelf", f, q, gain)
def highshelf_hz(f, q, gain): return _fsvf("highshelf", f, q, gain)
def _svf(mode): return _leaf("Svf<4>" if SVF_MODES[mode] >= 6 else "Svf<3>", 4 if SVF_MODES[mode] >= 6 else 3, 1, mode=float(SVF_MODES[mode]))
def lowpass(): return _svf("lowpass")
def highpass(): return _svf("highpass")
def bandpass(): return _svf("bandpass")
def notch(): return _svf("notch")
def peak(): return _svf("peak")
def allpass(): return _svf("allpass")
def bell(): return _svf("bell")
def lowshelf(): return _svf("lowshelf")
def highshelf(): return _svf("highshelf")
def morph(): return _leaf("Morph", 4, 1)
def biquad(a1, a2, b0, b1, b2): return _leaf("Biquad", 1, 1, a1=a1, a2=a2, b0=b0, b1=b1, b2=b2)
def butterpass_hz(f): return _leaf("ButterLowpass<1>", 1, 1, cutoff=f)
def butterpass(): return _leaf("ButterLowpass<2>", 2, 1)
def resonator_hz(center, bandwidth): return _leaf("Resonator<1>", 1, 1, center=center, q=bandwidth)
def resonator(): return _leaf("Resonator<3>", 3, 1, center=440.0, q=110.0)   # prelude32.rs:521
def moog_hz(f, q): return _leaf("Moog<1>", 1, 1, cutoff=f, q=q)
def moog(): return _leaf("Moog<3>", 3, 1)
def fir(*w): return Graph(f"Fir<{len(w)}>", 1, 1, [((), f"w[{i}]", x, False) for i, x in enumerate(w)])
def fir3(gain):  # prelude.rs:863-867 (f32 arithmetic)
    g = np.asarray(gain, dtype=np.float32)
    alpha = (g + np.float32(1.0)) / np.float32(2.0)
    beta = (np.float32(1.0) - alpha) / np.float32(2.0)
    return fir(beta, alpha, beta)
def lowpole_hz(f): return _leaf("OnePole<OP_LOWPOLE,1>", 1, 1, cutoff=f)
def lowpole(): return _leaf("OnePole<OP_LOWPOLE,2>", 2, 1)
def highpole_hz(f): return _leaf("OnePole<OP_HIGHPOLE,1>", 1, 1, cutoff=f)
def highpole(): return _leaf("OnePole<OP_HIGHPOLE,2>", 2, 1)
def dcblock_hz(f): return _leaf("OnePole<OP_DCBLOCK,1>", 1, 1, cutoff=f)
def allpole_delay(d): return _leaf("OnePole<OP_ALLPOLE,1>", 1, 1, delay=d)
def allpole(): return _leaf("OnePole<OP_ALLPOLE,2>", 2, 1)
def pinkpass(): return _leaf("Pinkpass", 1, 1)
def lowrez_hz(cutoff, q): return _leaf("Rez<1>", 1, 1, bandpass=0.0, cutoff=cutoff, q=q)
def lowrez(): return _leaf("Rez<3>", 3, 1, bandpass=0.0)
def bandrez_hz(center, q): return _leaf("Rez<1>", 1, 1, bandpass=1.0, cutoff=center, q=q)
def bandrez(): return _leaf("Rez<3>", 3, 1, bandpass=1.0)
def follow(t): return _leaf("Follow", 1, 1, response_time=t)
def afollow(a, r): return _leaf("AFollow", 1, 1, attack_time=a, release_time=r)
def mls_bits(n): return _leaf("Mls", 0, 1, bits=float(n))
def mls(): return mls_bits(29)
def lfo_exp(a=1.0, k=1.0):        # lfo(|t| a * exp(-t * k))   prelude32.rs:602
    return Graph("Envelope<EnvExp>", 0, 1, [((0,), "a", a, False), ((0,), "k", k, False)])
def lfo_sine_hz(hz, lo=-1.0, hi=1.0):  # lfo(|t| lerp11(lo, hi, sin_hz(hz, t)))
    return Graph("Envelope<EnvSineHz>", 0, 1, [((0,), "hz", hz, False), ((0,), "lo", lo, False), ((0,), "hi", hi, False)])
def envelope(functor, source, outputs=1, **params):
    """envelope(|t| ...) / lfo(|t| ...) (prelude32.rs:581-611) with the closure given as a C++ functor: `functor` is
    its type name, `source` its definition (contract: Envelope<FN> in fd_nodes.hpp), `params` its per-voice fields."""
    return Graph(f"Envelope<{functor}>", 0, outputs, [((0,), k, v, False) for k, v in params.items()], 0, source)
lfo = envelope
def lfo2_exp():  # lfo2(|t, speed| exp(-t * speed))   prelude32.rs:623
    return Graph("EnvelopeIn<EnvInExp>", 1, 1)
def envelope_in(functor, source, inputs, outputs=1, **params):
    """envelope2 / lfo2 / envelope_in / lfo_in (prelude32.rs:625-745): closure(t, inputs) as a C++ functor
    (contract: EnvelopeIn<FN> in fd_nodes.hpp)."""
    return Graph(f"EnvelopeIn<{functor}>", inputs, outputs, [((0,), k, v, False) for k, v in params.items()], 0, source)
lfo_in = envelope_in
def pluck(frequency, gain_per_second, damping):  # excitation: Bank.set_ring(0, rnd_stream)
    return _leaf("Pluck", 1, 1, rings=2, frequency=frequency, gain_per_second=gain_per_second, high_frequency_damping=damping)
def dsf_saw(): return _leaf("Dsf<2>", 2, 1, harmonic_spacing=1.0, roughness=0.5)
def dsf_saw_r(r): return _leaf("Dsf<1>", 1, 1, harmonic_spacing=1.0, roughness=r)
def dsf_square(): return _leaf("Dsf<2>", 2, 1, harmonic_spacing=2.0, roughness=0.5)
def dsf_square_r(r): return _leaf("Dsf<1>", 1, 1, harmonic_spacing=2.0, roughness=r)
def delay(t): return _leaf("Delay", 1, 1, rings=1, time=t)
def tap(min_delay, max_delay): return _leaf("TapT<false>", 2, 1, rings=1, min_delay=min_delay, max_delay=max_delay)
def tap_linear(min_delay, max_delay): return _leaf("TapT<true>", 2, 1, rings=1, min_delay=min_delay, max_delay=max_delay)
def multitap(n, min_delay, max_delay): return _leaf(f"TapT<false,{n}>", 1 + n, 1, rings=1, min_delay=min_delay, max_delay=max_delay)
def multitap_linear(n, min_delay, max_delay): return _leaf(f"TapT<true,{n}>", 1 + n, 1, rings=1, min_delay=min_delay, max_delay=max_delay)
def multitick(n): return _leaf(f"Tick<{n}>", n, n)
def panner(): return _leaf("PannerT<2>", 2, 2)
def allnest(x):  # prelude32.rs:1112: coefficient on input 1
    return Graph(f"AllNest<{x.type},2>", 2, 1, [((0,) + p, f, v, u) for p, f, v, u in x.params], x.rings, x.source)
def allnest_c(coefficient, x):
    return Graph(f"AllNest<{x.type}>", 1, 1, [((0,) + p, f, v, u) for p, f, v, u in x.params] + [((), "coefficient", coefficient, False)], x.rings, x.source)
def resample(x):  # prelude32.rs:1021: x is a generator, input 0 = speed
    assert x.nin == 0
    return Graph(f"Resample<{x.type}>", 1, x.nout, [((0,) + p, f, v, u) for p, f, v, u in x.params], x.rings, x.source)
def oversample(x):  # prelude32.rs:983
    return Graph(f"Oversampler<{x.type}>", x.nin, x.nout, [((0,) + p, f, v, u) for p, f, v, u in x.params], x.rings, x.source)
def saw(): return _leaf("WaveSynth<0>", 1, 1)
def square(): return _leaf("WaveSynth<1>", 1, 1)
def triangle(): return _leaf("WaveSynth<2>", 1, 1)
def saw_hz(f): return constant(f) >> saw()
def square_hz(f): return constant(f) >> square()
def triangle_hz(f): return constant(f) >> triangle()
def organ(): return _leaf("WaveSynth<4>", 1, 1)                    # prelude32.rs organ/soft_saw/hammond: WaveSynth over
def soft_saw(): return _leaf("WaveSynth<5>", 1, 1)                 # organ_table / soft_saw_table / hammond_table
def hammond(): return _leaf("WaveSynth<6>", 1, 1)
def organ_hz(f): return constant(f) >> organ()
def soft_saw_hz(f): return constant(f) >> soft_saw()
def hammond_hz(f): return constant(f) >> hammond()
def playwave_at(slot, channel, start_point, end_point, loop_point=None):   # prelude32.rs:2234; the Wave: wave_upload(slot, ..)
    # the fourth field 2 marks u32 slots: Bank.from_graph uploads their words bit for bit
    return Graph(f"WavePlayer<{slot}>", 0, 1, [((), "channel", channel, 2), ((), "start_point", start_point, 2),
                                               ((), "end_point", end_point, 2),
                                               ((), "loop_point", 0xFFFFFFFF if loop_point is None else loop_point, 2)])
def playwave(slot, channel, length, loop_point=None): return playwave_at(slot, channel, 0, length, loop_point)   # prelude32.rs:2225
def pulse(): return _leaf("PulseWave", 2, 1)                       # input 0 frequency, input 1 pulse width (wavetable.rs:437)
def ramp(): return _leaf("PhaseOsc<OSC_RAMP>", 1, 1)
def poly_saw(): return _leaf("PhaseOsc<OSC_POLYSAW>", 1, 1)
def poly_square(): return _leaf("PhaseOsc<OSC_POLYSQUARE>", 1, 1)
def poly_pulse(): return _leaf("PhaseOsc<OSC_POLYPULSE>", 2, 1)
def rossler(): return _leaf("Chaos<false>", 1, 1)
def lorenz(): return _leaf("Chaos<true>", 1, 1)
def adsr_live(a, d, s, r): return _leaf("AdsrLive", 1, 1, attack=a, decay=d, sustain=s, release=r)
def pan(p): return _leaf("Panner", 1, 2, pan=p)
def shape(kind, p0=1.0, p1=0.0, smoothing=0.0):
    return _leaf("Shaper", 1, 1, shape=float(SHAPES[kind]), shape_p0=p0, shape_p1=p1, shape_smoothing=smoothing)



# --- routing (prelude32.rs:129-222, 1103-1156) and the function forms of the operators (:1330-1470)
def zero(): return constant(0.0)
def multizero(n): return constant(*([0.0] * n))
def multipass(n): return _leaf(f"MultiPass<{n}>", n, n)
def sink(): return _leaf("Sink<1>", 1, 0)
def multisink(n): return _leaf(f"Sink<{n}>", n, 0)
def split(n): return _leaf(f"Split<{n}>", 1, n)
def multisplit(m, n): return _leaf(f"MultiSplit<{m},{n}>", m, m * n)
def join(n): return _leaf(f"Join<{n}>", n, 1)
def multijoin(m, n): return _leaf(f"MultiJoin<{m},{n}>", m * n, m)
def reverse(n): return _leaf(f"Reverse<{n}>", n, n)
def impulse(n=1): return _leaf(f"Impulse<{n}>", 0, n)
def _feedback(x, y, op):
    if x.nin != x.nout or (y is not None and (y.nin != x.nout or y.nout != x.nout)):
        raise TypeError("feedback: the enclosed nodes need as many outputs as inputs")
    ps = [((0,) + p, f, v, u) for p, f, v, u in x.params]
    if y is None:
        return Graph(f"Feedback<{x.type},{op}>", x.nin, x.nout, ps, x.rings, x.source)
    ps += [((1,) + p, f, v, u) for p, f, v, u in y.params]
    return Graph(f"Feedback2<{x.type},{y.type},{op}>", x.nin, x.nout, ps, x.rings + y.rings, _merge(x.source, y.source))
def feedback(x): return _feedback(x, None, "FbId")                 # prelude32.rs:1040
def feedback2(x, y): return _feedback(x, y, "FbId")                # prelude32.rs:1061
def fdn(x): return _feedback(x, None, "FbHadamard")                # prelude32.rs:1323
def fdn2(x, y): return _feedback(x, y, "FbHadamard")               # prelude32.rs:1340
METER_MODES = dict(sample=0, peak=1, rms=2)
def _meter(mode, timescale, monitor):
    ts = np.asarray(timescale, dtype=np.float64).view(np.uint64)   # the f64 timescale travels as its bit pattern
    return Graph(f"MeterT<{METER_MODES[mode]},{'true' if monitor else 'false'}>", 1, 1, [((), "timescale", ts, True)])
def meter(mode, timescale=0.1): return _meter(mode, timescale, False)     # prelude32.rs:300: meter(Meter::Peak(t)) etc.
def monitor(mode, timescale=0.1): return _meter(mode, timescale, True)    # level readable from the ":state" slot
def var(value): return _leaf("Var", 0, 1, value=value)                    # a Shared value = a per-voice parameter
def hold(variability):  # prelude32.rs:830; the Rnd draws: Bank.set_ring(<this node's ring>, hold_stream(draws))
    return _leaf("Hold", 2, 1, rings=1, variability=variability)
def hold_hz(f, variability): return (pass_() | dc(f)) >> hold(variability)                    # prelude32.rs:843
def hold_stream(draws):
    """[voices][n] f64 draws of `Rnd::from_u64(hash).f64()` -> the [voices][2n] f32 words Hold's ring takes."""
    d = np.ascontiguousarray(draws, dtype=np.float64)
    return d.view(np.float32).reshape(d.shape[0], -1)
def mixer(matrix):                                                        # Mixer::new pan.rs:108, matrix[out][in]
    m = np.asarray(matrix, dtype=object)
    n_out, n_in = len(matrix), len(matrix[0])
    return Graph(f"Mixer<{n_in},{n_out}>", n_in, n_out,
                 [((), f"matrix[{i * n_in + j}]", matrix[i][j], False) for i in range(n_out) for j in range(n_in)])
def rotate(angle, gain):                                                  # prelude32.rs:2432; libm cos / sin on the host
    from ._lib import lib
    c, s_ = np.float32(lib().fdsp_libm_cosf(float(angle))), np.float32(lib().fdsp_libm_sinf(float(angle)))
    g = np.float32(gain)
    return mixer([[c * g, -s_ * g], [s_ * g, c * g]])
def var_fn(value, functor, source, outputs=1):
    """var_fn(&shared, |x| ..): functor with `static FD_HD void f(float value, float* out)`; the value is the ":value" slot."""
    return Graph(f"VarFn<{functor},{outputs}>", 0, outputs, [((), "value", value, False)], 0, source)
def envelope2(functor, source, outputs=1, **params): return envelope_in(functor, source, 1, outputs, **params)   # prelude32.rs:625
def envelope3(functor, source, outputs=1, **params): return envelope_in(functor, source, 2, outputs, **params)   # prelude32.rs:669
lfo2, lfo3 = envelope2, envelope3
def biquad_bank(coefs):
    """biquad_bank() (prelude32.rs:2711) with Setting::biquad(a1, a2, b0, b1, b2).index(i) applied: `coefs` = 8 rows of
    (a1, a2, b0, b1, b2), each entry a scalar or a per-voice array."""
    ps = [((i,), n, coefs[i][k], False) for i in range(8) for k, n in enumerate(("a1", "a2", "b0", "b1", "b2"))]
    return Graph("BiquadBank", 8, 8, ps)
def limiter(attack, release):                                             # prelude32.rs:1275; needs ring_frames
    return _leaf("Limiter<1>", 1, 1, rings=2, attack_time=attack, release_time=release)
def limiter_stereo(attack, release):                                      # prelude32.rs:1286
    return _leaf("Limiter<2>", 2, 2, rings=3, attack_time=attack, release_time=release)
def thru(x): return ~x
def bus(x, y): return x & y
def branch(x, y): return x ^ y
def stack(x, y): return x | y
def pipe(x, y): return x >> y
def sum_(x, y): return x + y
def product(x, y): return x * y
def add(*x): return multipass(len(x)) + dc(*x)                    # prelude32.rs:391: MultiPass + dc(x)
def sub(*x): return multipass(len(x)) - dc(*x)                    # prelude32.rs:409
def mul(*x): return multipass(len(x)) * dc(*x)                    # prelude32.rs:427
def declick(): return _leaf("Declick", 1, 1, duration=0.010)
def declick_s(t): return _leaf("Declick", 1, 1, duration=t)


def map_(functor, source, inputs, outputs=1):
    """map(|i: &Frame<f32, I>| ..) (prelude32.rs:332) with the closure as a C++ functor type `functor` whose definition
    `source` provides `static __device__ void f(const float* in, float* out)` (contract: Map<FN,NI,NO> in fd_nodes.hpp)."""
    return Graph(f"Map<{functor},{inputs},{outputs}>", inputs, outputs, [], 0, source)


def shape_fn(functor, source):
    """shape_fn(|x| ..) (prelude32.rs:1181): functor with `static __device__ float f(float x)` (ShaperFn<FN>)."""
    return Graph(f"ShaperFn<{functor}>", 1, 1, [], 0, source)


def _multi(tmpl, nodes, nin_mul, nout_mul, extra=""):
    nodes = list(nodes)
    t = nodes[0].type
    if any(n.type != t for n in nodes):
        raise TypeError("the N nodes of busi/stacki/branchi/sumi/pipei must have one type (as in Rust)")
    ps, src = [], ""
    for i, n in enumerate(nodes):
        ps += [((i,) + p, f, v, u) for p, f, v, u in n.params]
        src = _merge(src, n.source)
    k = len(nodes)
    return Graph(f"{tmpl}<{k},{t}{extra}>", nodes[0].nin * (k if nin_mul else 1), nodes[0].nout * (k if nout_mul else 1),
                 ps, sum(n.rings for n in nodes), src)


def busi(n, f): return _multi("MultiBus", [f(i) for i in range(n)], False, False)        # prelude.rs:1385
def stacki(n, f): return _multi("MultiStack", [f(i) for i in range(n)], True, True)      # prelude.rs:1452
def branchi(n, f): return _multi("MultiBranch", [f(i) for i in range(n)], False, True)   # prelude.rs:1342
def sumi(n, f): return _multi("Reduce", [f(i) for i in range(n)], True, False, ",OpAdd")  # prelude.rs:1565
def pipei(n, f):                                                                          # prelude.rs:1620
    g = _multi("PipeN", [f(i) for i in range(n)], False, False)
    if g.nin != g.nout:
        raise TypeError("pipei needs as many outputs as inputs")
    return g
def _frac(n, i): return np.float32(i / (n - 1)) if n > 1 else np.float32(0.5)
def busf(n, f): return busi(n, lambda i: f(_frac(n, i)))
def stackf(n, f): return stacki(n, lambda i: f(_frac(n, i)))
def branchf(n, f): return branchi(n, lambda i: f(_frac(n, i)))
def sumf(n, f): return sumi(n, lambda i: f(_frac(n, i)))
def pipef(n, f): return pipei(n, lambda i: f(_frac(n, i)))


# --- opcodes the prelude composes from the ones above
def _svf_q(mode, q):                                                                  # prelude.rs:2127-2140 etc.
    f = _svf(mode)
    f.params.append(((), "q", q, False))
    return (multipass(2) | dc(q)) >> f
def _svf_qg(mode, q, gain):                                                           # prelude.rs:2425-2446 etc.
    f = _svf(mode)
    f.params += [((), "q", q, False), ((), "gain", gain, False)]
    return (multipass(2) | dc(q, gain)) >> f
def lowpass_q(q): return _svf_q("lowpass", q)
def highpass_q(q): return _svf_q("highpass", q)
def bandpass_q(q): return _svf_q("bandpass", q)
def notch_q(q): return _svf_q("notch", q)
def peak_q(q): return _svf_q("peak", q)
def allpass_q(q): return _svf_q("allpass", q)
def bell_q(q, gain): return _svf_qg("bell", q, gain)
def lowshelf_q(q, gain): return _svf_qg("lowshelf", q, gain)
def highshelf_q(q, gain): return _svf_qg("highshelf", q, gain)
def lowrez_q(q): return (multipass(2) | dc(q)) >> lowrez()
def bandrez_q(q): return (multipass(2) | dc(q)) >> bandrez()
def moog_q(q): return (multipass(2) | dc(q)) >> _leaf("Moog<3>", 3, 1, cutoff=1000.0, q=q)   # prelude32.rs:560
def morph_hz(f, q, m): return (pass_() | dc(f, q, m)) >> morph()
def pink(): return white() >> pinkpass()                                              # prelude32.rs:1299
def brown(): return white() >> lowpole_hz(10.0) * dc(13.7)                            # prelude32.rs:1305
def dcblock(): return dcblock_hz(10.0)
def clip(): return shape("clip", 1.0)
def clip_to(lo, hi): return shape("clip_to", lo, hi)
def ramp_hz(f): return dc(f) >> ramp()
def poly_saw_hz(f): return dc(f) >> poly_saw()
def poly_square_hz(f): return dc(f) >> poly_square()
def poly_pulse_hz(f, width): return dc(f, width) >> poly_pulse()


# waveshapes for the nonlinear biquads: (kind, p0, p1) as in shape()
def Clip(h=1.0): return ("clip", h, 0.0)
def ClipTo(lo, hi): return ("clip_to", lo, hi)
def Tanh(h=1.0): return ("tanh", h, 0.0)
def Atan(h=1.0): return ("atan", h, 0.0)
def Softsign(h=1.0): return ("softsign", h, 0.0)
def Crush(levels): return ("crush", levels, 0.0)
def SoftCrush(levels): return ("soft_crush", levels, 0.0)


def _nlbiquad(dirty, nin, mode, shp, **fields):
    kind, p0, p1 = shp
    ps = [((), "mode", float(BQ_KINDS[mode]), False)] + [((), k, v, False) for k, v in fields.items()]
    for path in ((0,), (1,)) if dirty else ((0,),):
        ps += [(path, "shape", float(SHAPES[kind]), False), (path, "shape_p0", p0, False), (path, "shape_p1", p1, False)]
    return Graph(f"NlBiquad<{'true' if dirty else 'false'},{nin}>", nin, 1, ps)


# biquad with nonlinear feedback (f..) / nonlinear state shaping (d..)  prelude32.rs:2440-2660
def fresonator_hz(shp, center, q): return _nlbiquad(False, 1, "resonator", shp, center=center, q=q)
def flowpass_hz(shp, cutoff, q): return _nlbiquad(False, 1, "lowpass", shp, center=cutoff, q=q)
def fhighpass_hz(shp, cutoff, q): return _nlbiquad(False, 1, "highpass", shp, center=cutoff, q=q)
def fbell_hz(shp, center, q, gain): return _nlbiquad(False, 1, "bell", shp, center=center, q=q, gain=gain)
def dresonator_hz(shp, center, q): return _nlbiquad(True, 1, "resonator", shp, center=center, q=q)
def dlowpass_hz(shp, cutoff, q): return _nlbiquad(True, 1, "lowpass", shp, center=cutoff, q=q)
def dhighpass_hz(shp, cutoff, q): return _nlbiquad(True, 1, "highpass", shp, center=cutoff, q=q)
def dbell_hz(shp, center, q, gain): return _nlbiquad(True, 1, "bell", shp, center=center, q=q, gain=gain)
def fresonator(shp): return _nlbiquad(False, 3, "resonator", shp)
def flowpass(shp): return _nlbiquad(False, 3, "lowpass", shp)
def fhighpass(shp): return _nlbiquad(False, 3, "highpass", shp)
def fbell(shp): return _nlbiquad(False, 4, "bell", shp)
def dresonator(shp): return _nlbiquad(True, 3, "resonator", shp)
def dlowpass(shp): return _nlbiquad(True, 3, "lowpass", shp)
def dhighpass(shp): return _nlbiquad(True, 3, "highpass", shp)
def dbell(shp): return _nlbiquad(True, 4, "bell", shp)



# --- effects the prelude composes (prelude.rs:2719-2753).  The Rust closure delay_f / phase_f arrives as an Envelope
# functor (see envelope()): `functor` type name, `source` its definition ("" for the built-in EnvSineHz / EnvExp).
def flanger(feedback_amount, minimum_delay, maximum_delay, functor, source="", **params):
    return pass_() & feedback2((pass_() | lfo(functor, source, **params)) >> tap(minimum_delay, maximum_delay),
                               shape("tanh", feedback_amount))


def phaser(feedback_amount, functor, source="", **params):
    # lfo(move |t| lerp(2.0, 20.0, clamp01(phase_f(t)))): the wrapper is a functor around the caller's
    wrap = f"PhaserLfo_{functor}"
    wsrc = (f"struct {wrap} {{ static constexpr int OUT = 1; {functor} f; "
            "template <class V> FD_HD void visit(V& v) { f.visit(v); } FD_HD void init() { f.init(); } "
            "FD_HD void eval(float t, float* out) const { float p; f.eval(t, &p); out[0] = lerpf(2.0f, 20.0f, clamp01f(p)); } };")
    inner = ((pass_() | lfo(wrap, _merge(source, wsrc), **params))
             >> pipei(10, lambda _i: add(0.0, 0.1) >> ~allpole())
             >> (mul(feedback_amount) | sink()))
    return pass_() & feedback(inner)


REVERB4_DELAYS = [0.059326634, 0.04778291, 0.06995449, 0.0393001, 0.041604012, 0.06215825, 0.052269846, 0.043227978,
                  0.06966107, 0.031615064, 0.068442, 0.037332155, 0.032944717, 0.034493037, 0.06787566, 0.038824916,
                  0.068260126, 0.068044715, 0.0688076, 0.066724524, 0.051293883, 0.06023173, 0.040897705, 0.031507637,
                  0.060309593, 0.049584292, 0.04532072, 0.056379095, 0.035180368, 0.041291796, 0.046129026, 0.05504605]


def _db_amp(db): return float(np.exp(np.float64(db) / 20.0 * np.float64(2.302585092994046)))  # exp10: math.rs:76-78, 294-296


def _smooth9(x):  # math.rs:431-437 in f32
    f = np.float32
    x = f(x)
    x2 = x * x
    return ((((f(70) * x - f(315)) * x + f(540)) * x - f(420)) * x + f(126)) * x2 * x2 * x


def reverb3_stereo(time, diffusion, make_filter):
    """reverb3_stereo(time, diffusion, filter) (prelude.rs:1858, reverb.rs:152-279): allpass loop reverb; make_filter()
    builds the 1-in 1-out loop filter (cloned 16 times by the reference).  Needs 76 delay rings of >= 1200 frames."""
    f = np.float32
    fs = [make_filter() for _ in range(16)]
    if any(x.type != fs[0].type or x.nin != 1 or x.nout != 1 for x in fs):
        raise TypeError("reverb3_stereo: the loop filter is one 1-in 1-out node type")
    coeff = f(0.5 * (1.0 - diffusion) + 0.9 * diffusion)
    a = f(_db_amp(-60.0) ** (0.035 / time))
    ps = [((), "a", a, False), ((), "coefficient", coeff, False)]
    for b in range(8):                      # visit order of Reverb3: pre 0..3, then per block ap0 x4, ap1 x4, f0, f1, delay
        for k, flt in ((4 + b * 11 + 8, fs[2 * b]), (4 + b * 11 + 9, fs[2 * b + 1])):
            ps += [((k,) + p, fld, v, u) for p, fld, v, u in flt.params]
    g = Graph(f"Reverb3<{fs[0].type}>", 2, 2, ps, 76 + 16 * fs[0].rings, fs[0].source)
    # The stock shapes -- reverb3_stereo(time, diffusion, lowpole_hz(cutoff)) or (time, diffusion, <a FixedSvf: lowpass_hz .. highshelf_hz>) with the same scalar
    # parameters in all sixteen filters -- have a dedicated lane-per-frame kernel (fdsp_reverb3_stereo[_svf]_create); Bank.from_graph takes it when the graph IS this node
    def uniform(field):   # the one scalar value all sixteen filters hold in `field`, else None
        vals = {float(np.float32(np.asarray(v))) for x in fs for _p, fld, v, _u in x.params if fld == field and np.asarray(v).ndim == 0}
        n = sum(1 for x in fs for _p, fld, _v, _u in x.params if fld == field)
        return vals.pop() if len(vals) == 1 and n == 16 else None

    if np.asarray(time).ndim == 0 and np.asarray(diffusion).ndim == 0:
        if fs[0].type == "OnePole<OP_LOWPOLE,1>" and all(len(x.params) == 1 for x in fs) and uniform("cutoff") is not None:
            g.reverb3_plan = dict(time=float(time), diffusion=float(diffusion), cutoff=uniform("cutoff"))
        elif fs[0].type == "FixedSvf" and all(len(x.params) == 4 for x in fs) and None not in [uniform(k) for k in ("mode", "cutoff", "q", "gain")]:
            g.reverb3_plan = dict(time=float(time), diffusion=float(diffusion), cutoff=uniform("cutoff"), svf=int(uniform("mode")), q=uniform("q"), gain=uniform("gain"))
    return g


def reverb4_stereo_delays(delays, time):  # prelude.rs:1917-1941: two 16-line Hadamard FDNs in series
    f = np.float32
    a = f(_db_amp(-60.0) ** (0.03 * 10.0 / 10.0 / time))
    w = (-a / f(4.0), -a / f(2.0), -a / f(4.0))
    line1 = stacki(16, lambda i: delay(float(f(delays[i]))) >> fir(*w))
    line2 = stacki(16, lambda i: delay(float(f(delays[16 + i]))) >> fir(*w))
    pans = sumf(16, lambda x: pan(f(-1.0) * (f(1.0) - _smooth9(x)) + f(1.0) * _smooth9(x)))
    return (multisplit(2, 8) >> fdn(line1) >> multijoin(2, 8) >> multisplit(2, 8) >> fdn(line2)
            >> pans * dc(1.0 / 4.0, 1.0 / 4.0))


def reverb4_stereo(room_size, time):  # prelude.rs:1873-1914
    f = np.float32
    scale = max(f(room_size), f(15.0)) / f(10.0)
    g = reverb4_stereo_delays([f(d) * scale for d in REVERB4_DELAYS], time)
    if np.asarray(room_size).ndim == 0 and np.asarray(time).ndim == 0:   # the stock reverb itself: Bank.from_graph takes its dedicated kernel
        g.stock_reverb = ("reverb4_stereo", (float(room_size), float(time)))
    return g


REVERB_DELAYS = [0.073904, 0.052918, 0.066238, 0.066387, 0.037783, 0.080073, 0.050961, 0.075900, 0.043646, 0.072095, 0.056194, 0.045961,
                 0.058934, 0.068016, 0.047529, 0.058156, 0.072972, 0.036084, 0.062715, 0.076377, 0.044339, 0.076725, 0.077884, 0.046126,
                 0.067741, 0.049800, 0.051709, 0.082923, 0.070121, 0.079315, 0.055039, 0.081859]   # prelude.rs:1739-1744


def reverb_stereo(room_size, time, damping):
    """reverb_stereo(room_size, time, damping) (prelude.rs:1732-1762): multisplit::<U2, U16>() >> fdn::<U32>(stacki(|i| delay(DELAYS[i] * room / 10)
    >> fir3(1 - damping) * a)) >> sumf::<U32>(pan) * dc((1/16, 1/16)) -- BASELINE config 5 as a graph (scalar arguments)."""
    f = np.float32
    a = f(_db_amp(-60.0) ** (0.03 * room_size / 10.0 / time))                       # :1746
    gain = f(1.0) - f(damping)
    alpha = (gain + f(1.0)) / f(2.0)
    beta = (f(1.0) - alpha) / f(2.0)
    w = (beta * a, alpha * a, beta * a)                                              # fir3(gain).weights() * a  :1747, :863-867
    line = stacki(32, lambda i: delay(float(f(REVERB_DELAYS[i] * room_size / 10.0))) >> fir(*w))   # delay(t as f32) :1749
    pans = sumf(32, lambda x: pan(f(-1.0) * (f(1.0) - _smooth9(x)) + f(1.0) * _smooth9(x)))
    g = multisplit(2, 16) >> fdn(line) >> pans * dc(1.0 / 16.0, 1.0 / 16.0)
    g.stock_reverb = ("reverb_stereo", (float(room_size), float(time), float(damping)))
    return g


def _parse_type(t):
    """`Name<arg, ..>` -> (name, [children]) with every argument parsed the same way (numbers and names become leaves)."""
    pos = 0

    def node():
        nonlocal pos
        a = pos
        while pos < len(t) and t[pos] not in "<>,":
            pos += 1
        name, kids = t[a:pos].strip(), []
        if pos < len(t) and t[pos] == "<":
            pos += 1
            while True:
                kids.append(node())
                if t[pos] == ",":
                    pos += 1
                    continue
                pos += 1  # '>'
                break
        return name, kids

    return node()


def fdn_plan(g):
    """The arguments of fdsp_fdn_create when `g` is the Hadamard feedback delay network the prelude documents (prelude.rs:1323-1345,
    the "Mono Reverb" example :1334) --

        split(N) | multisplit(2, N/2)  >>  fdn(stacki(N, lambda i: delay(t_i) >> fir(w..)))  >>  join(N) | multijoin(2, N/2)

    with N in 2, 4, 8, 16, 32, one to three FIR weights shared by all lines, and every parameter a scalar (the same network in every
    instance) -- else None.  Such a graph renders through the lane-per-frame FDN kernel (one wave per instance, the lines in registers,
    coalesced ring rows) instead of lane-per-voice; Bank.from_graph takes that route when every delay exceeds two blocks."""
    name, kids = _parse_type(g.type)
    chain = []

    def flat(n, path):  # the Pipe chain, left to right, with the parameter path of every element
        if n[0] == "Pipe" and len(n[1]) == 2:
            flat(n[1][0], path + (0,))
            flat(n[1][1], path + (1,))
        else:
            chain.append((n, path))

    flat((name, kids), ())
    if len(chain) != 3:
        return None
    (sp, _), (fb, fpath), (jn, _) = chain
    try:
        if sp[0] == "Split":
            nin, n_in = 1, int(sp[1][0][0])
        elif sp[0] == "MultiSplit" and int(sp[1][0][0]) == 2:
            nin, n_in = 2, 2 * int(sp[1][1][0])
        else:
            return None
        if jn[0] == "Join":
            nout, n_out = 1, int(jn[1][0][0])
        elif jn[0] == "MultiJoin" and int(jn[1][0][0]) == 2:
            nout, n_out = 2, 2 * int(jn[1][1][0])
        else:
            return None
        if fb[0] != "Feedback" or fb[1][1][0] != "FbHadamard":
            return None
        st = fb[1][0]
        if st[0] != "MultiStack":
            return None
        n, line = int(st[1][0][0]), st[1][1]
        if line[0] != "Pipe" or line[1][0][0] != "Delay" or line[1][1][0] != "Fir":
            return None
        taps = int(line[1][1][1][0][0])
    except (IndexError, ValueError):
        return None
    if n not in (2, 4, 8, 16, 32) or n_in != n or n_out != n or not 1 <= taps <= 3:
        return None
    vals = {}
    for path, field, value, _u in g.params:
        v = np.asarray(value)
        if v.ndim != 0:
            return None       # per-voice parameters: the generic kernels
        vals[(tuple(path), field)] = v
    try:
        delays = [float(np.float32(vals[(fpath + (0, i, 0), "time")])) for i in range(n)]   # delay(t: f32) -> Delay::new(t as f64)
        ws = [[np.float32(vals[(fpath + (0, i, 1), f"w[{j}]")]) for j in range(taps)] for i in range(n)]
    except KeyError:
        return None
    if any(w != ws[0] for w in ws[1:]):
        return None           # the kernel keeps one set of weights for all lines
    if len(vals) != n * (1 + taps):
        return None           # something else carries parameters (builders on the nodes): not this shape
    return dict(lines=n, delays=delays, taps=taps, weights=[float(x) for x in ws[0]], inputs=nin, outputs=nout)


def lane_per_frame_shape(g):
    """whether `g` is one of the nodes / networks with a lane-per-frame kernel (what Bank.from_graph builds as an FDN / reverb bank)"""
    return getattr(g, "stock_reverb", None) is not None or getattr(g, "reverb3_plan", None) is not None or fdn_plan(g) is not None


def bus_plan(g):
    """(inner graph, mode, wet, dry) -- the arguments of fdsp_bank_set_bus -- when `g` is a gain and / or a dry bus around `inner`, the shapes the
    reference's documentation gives its reverbs (README.md:436, wave.rs:514, CHANGES.md:203, net.rs:681):

        wet * inner                        (FDSP_BUS_WET)        Unop<X, FrameMulScalar>, combinator.rs:477-488
        multipass() & wet * inner          (FDSP_BUS_DRY_WET)    Bus, audionode.rs:1842-1877; either order of the `&`,
        dry * multipass() & wet * inner                          with or without the factors (`pass()` for a mono network)

    with scalar factors (the same bus in every instance); else None.  Bank.from_graph folds such a bus into the inner bank's render kernel when
    `inner` has a lane-per-frame kernel (lane_per_frame_shape)."""
    def scaled(x):
        up = getattr(x, "unop_parts", None)
        if up is not None and up[1] == "UMulScalar" and np.asarray(up[2]).ndim == 0:
            return up[0], float(np.float32(up[2]))
        return x, None

    bp = getattr(g, "bus_parts", None)
    if bp is None:
        inner, wet = scaled(g)
        return None if wet is None else (inner, 1, wet, 1.0)
    for dry_side, wet_side in (bp, bp[::-1]):
        d, dry = scaled(dry_side)
        if not (d.type == "Pass" or d.type.startswith("MultiPass<")) or d.params:
            continue
        inner, wet = scaled(wet_side)
        return inner, 2, 1.0 if wet is None else wet, 1.0 if dry is None else dry
    return None


def uses_wavetables(g):
    sets = (("saw", "WaveSynth<0>"), ("square", "WaveSynth<1>"), ("triangle", "WaveSynth<2>"), ("organ", "WaveSynth<4>"),
            ("soft_saw", "WaveSynth<5>"), ("hammond", "WaveSynth<6>"), ("saw", "PulseWave"))
    return sorted({k for k, t in sets if t in g.type})

"""ctypes front-end for the CPU oracle (oracle/libfundsp_oracle.so) in FunDSP graph notation.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The opcode names and operator overloads follow the reference's prelude32 (src/prelude32.rs) and
combinator.rs (`>>` Pipe, `|` Stack, `*`/`+`/`-` with nodes = Binop, with floats = Unop) so that tests
read like the reference's own tests.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(_ORACLE_DIR, "libfundsp_oracle.so")


def build(force=False):
    srcs = [os.path.join(_ORACLE_DIR, f) for f in ("fundsp_oracle.c", "o_bank.c", "fundsp_oracle.h", "o_math.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-s"])
    return _SO


_lib = None

SVF_MODES = dict(lowpass=0, highpass=1, bandpass=2, notch=3, peak=4, allpass=5, bell=6, lowshelf=7, highshelf=8)
BQ_KINDS = dict(butter=0, resonator=1, lowpass=2, highpass=3, bell=4)
ADD, SUB, MUL = 0, 1, 2
NEG, ID, ADD_SCALAR, NEG_ADD_SCALAR, MUL_SCALAR = range(5)
DEFAULT_SR = 44100.0


class BankJob(C.Structure):
    _fields_ = [
        ("config", C.c_int), ("process_mode", C.c_int), ("out_layout", C.c_int), ("threads", C.c_int),
        ("sample_rate", C.c_double), ("voices", C.c_size_t), ("frames", C.c_size_t),
        ("p0", C.POINTER(C.c_float)), ("p1", C.POINTER(C.c_float)), ("p2", C.POINTER(C.c_float)),
        ("p3", C.POINTER(C.c_float)), ("seed", C.POINTER(C.c_uint64)),
    ]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    P, f, i, d, u64 = C.c_void_p, C.c_float, C.c_int, C.c_double, C.c_uint64
    fp = C.POINTER(C.c_float)
    sig = {
        "o_constant": (P, [i, fp]), "o_pass": (P, []), "o_sine": (P, []), "o_noise": (P, []),
        "o_fixed_svf": (P, [i, f, f, f]), "o_svf": (P, [i, f, f, f]), "o_biquad": (P, [f, f, f, f, f]),
        "o_butter_lowpass": (P, [i, f]), "o_resonator": (P, [i, f, f]), "o_biquad_bank": (P, []),
        "o_biquad_bank_set": (None, [P, i, f, f, f, f, f]), "o_moog": (P, [i, f, f]), "o_fir": (P, [i, fp]),
        "o_tick_node": (P, [i]), "o_delay": (P, [d]),
        "o_pipe": (P, [P, P]), "o_stack": (P, [P, P]), "o_binop": (P, [i, P, P]), "o_unop": (P, [i, P, f]),
        "o_free": (None, [P]), "o_inputs": (i, [P]), "o_outputs": (i, [P]), "o_reset": (None, [P]),
        "o_set_sample_rate": (None, [P, d]), "o_set_seed": (None, [P, u64]),
        "o_sine_set_phase": (None, [P, f]), "o_noise_set_seed": (None, [P, u64]),
        "o_sine_hash": (u64, [P]), "o_sine_phase": (f, [P]), "o_noise_state": (C.c_uint32, [P]),
        "o_tick": (None, [P, fp, fp]), "o_process": (None, [P, i, fp, fp]),
        "o_wave_render": (C.c_size_t, [P, d, d, fp, C.c_size_t]),
        "o_render_blocks": (None, [P, C.c_size_t, i, fp, fp]), "o_render_ticks": (None, [P, C.c_size_t, fp, fp]),
        "o_svf_coefs": (None, [i, f, f, f, f, fp]), "o_biquad_coefs": (None, [i, f, f, f, f, fp]),
        "o_moog_coefs": (None, [f, f, f, fp]),
        "o_math_sinf": (f, [f]), "o_math_cosf": (f, [f]), "o_math_tanf": (f, [f]), "o_math_tanhf": (f, [f]),
        "o_math_expf": (f, [f]), "o_math_expm1f": (f, [f]), "o_math_wide_sinf": (f, [f]),
        "o_math_rnd1": (d, [u64]), "o_math_hash1": (u64, [u64]), "o_math_atto": (u64, [u64, u64]),
        "o_math_hash32x": (C.c_uint32, [C.c_uint32]),
        "o_bank_render": (d, [C.POINTER(BankJob), fp]),
        "o_bank_render_fast": (d, [C.POINTER(BankJob), fp]), "o_fast_simd_flavour": (C.c_char_p, []),
        "o_make_wavetable": (i, [i, i, fp, C.POINTER(C.c_int), C.c_size_t, fp]),
        "o_wavetable_create": (P, [i, fp, C.POINTER(C.c_int), fp]), "o_wavetable_free": (None, [P]),
        "o_wavesynth": (P, [P, i]), "o_phasesynth": (P, [P]), "o_waveplayer": (P, [fp, i, C.c_size_t, i, C.c_size_t, C.c_size_t, C.c_long]), "o_wrap": (P, [P, u64]), "o_wavesynth_set_phase": (None, [P, f]),
        "o_adsr_live": (P, [f, f, f, f]), "o_panner": (P, [i, f]),
        "o_onepole": (P, [i, i, f]), "o_pinkpass": (P, []), "o_morph": (P, [f, f, f]),
        "o_rez": (P, [i, f, f, f]), "o_follow": (P, [f]), "o_afollow": (P, [f, f]), "o_mls": (P, [C.c_uint]),
        "o_mls_set_seed": (None, [P, C.c_uint64]), "o_oversample": (P, [P]), "o_resample": (P, [P]), "o_dsf": (P, [i, f, f]), "o_envelope": (P, [f, i, P, P]), "o_envelope_in": (P, [f, i, i, P, P]), "o_pluck": (P, [f, f, f, fp, C.c_size_t]),
        "o_math_powf": (f, [f, f]),
        "o_seq_new": (P, [i, i, d]), "o_seq_free": (None, [P]), "o_seq_push": (i, [P, d, d, i, d, d, P]),
        "o_seq_render": (None, [P, C.c_size_t, i, fp, fp, fp]), "o_seq_time": (d, [P]), "o_mls_period": (C.c_uint64, [C.c_uint]),
        "o_tap": (P, [i, f, f]), "o_allnest": (P, [f, P]), "o_multitap": (P, [i, i, f, f]), "o_allnest2": (P, [P]),
        "o_shaper": (P, [i, f, f]), "o_shaper_adaptive": (P, [i, f, f, f]), "o_phase_osc": (P, [i]), "o_osc_set_phase": (None, [P, f]), "o_chaos": (P, [i]),
        "o_nlbiquad": (P, [i, i, i, i, f, f, f, f, f]), "o_math_atanf": (f, [f]), "o_math_wide_atanf": (f, [f]),
        "o_adaptive_smoothing": (d, [f, d]),
        "o_multipass": (P, [i]), "o_sink": (P, [i]), "o_split": (P, [i, i]), "o_join": (P, [i, i]),
        "o_reverse": (P, [i]), "o_impulse": (P, [i]), "o_map": (P, [i, i, P, P]),
        "o_shape_fn": (P, [P, P]), "o_declick": (P, [f]),
        "o_feedback": (P, [P, P, i]), "o_meter": (P, [i, d, i]), "o_meter_level": (f, [P]), "o_var": (P, [f]),
        "o_var_set": (None, [P, f]), "o_mixer": (P, [i, i, fp]), "o_hold": (P, [f, C.POINTER(C.c_double), C.c_size_t]), "o_var_fn": (P, [f, i, P, P]), "o_limiter": (P, [i, f, f]),
        "o_branch": (P, [P, P]), "o_bus": (P, [P, P]), "o_thru": (P, [P]), "o_multi": (P, [i, i, C.POINTER(P), i]),
        "o_reverb_stereo": (P, [d, d, d]), "o_reverb3": (P, [d, d, C.POINTER(P)]),
        "o_reverb_stereo_params": (None, [d, d, d, d, fp, C.POINTER(C.c_int), fp, fp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


class Node:
    """Owning handle of an oracle graph. Combining nodes transfers ownership to the new parent."""

    def __init__(self, ptr, children=()):
        if not ptr:
            raise ValueError("oracle: arity mismatch while combining nodes")
        self.ptr = ptr
        self._owned = True
        self.children = list(children)
        for c in children:
            c._owned = False

    def __del__(self):
        if getattr(self, "_owned", False) and self.ptr and _lib is not None:
            _lib.o_free(self.ptr)
            self.ptr = None

    # --- AudioNode / AudioUnit surface (audionode.rs:29-369, audiounit.rs:21-371)
    def inputs(self): return lib().o_inputs(self.ptr)
    def outputs(self): return lib().o_outputs(self.ptr)
    def reset(self): lib().o_reset(self.ptr)
    def set_sample_rate(self, sr): lib().o_set_sample_rate(self.ptr, float(sr))
    def set_seed(self, seed): lib().o_set_seed(self.ptr, int(seed) & (2**64 - 1))

    def tick(self, frame=()):
        fin = np.asarray(frame, dtype=np.float32).reshape(-1)
        assert fin.size == self.inputs()
        out = np.zeros(max(self.outputs(), 1), dtype=np.float32)
        lib().o_tick(self.ptr, _fptr(fin) if fin.size else None, _fptr(out))
        return out[: self.outputs()]

    def process(self, size, inp=None):
        """inp: [inputs][64] planar block. Returns [outputs][64] (values past `size` undefined)."""
        out = np.zeros((max(self.outputs(), 1), 64), dtype=np.float32)
        if self.inputs():
            inp = np.ascontiguousarray(inp, dtype=np.float32)
            assert inp.shape == (self.inputs(), 64)
        lib().o_process(self.ptr, int(size), _fptr(inp) if self.inputs() else None, _fptr(out))
        return out[: self.outputs()]

    def render_blocks(self, x=None, length=None, block=64):
        """Feed [inputs][length] through process() in `block`-sized chunks -> [outputs][length]."""
        if self.inputs():
            x = np.ascontiguousarray(np.atleast_2d(np.asarray(x, dtype=np.float32)))
            length = x.shape[1]
        out = np.zeros((self.outputs(), length), dtype=np.float32)
        lib().o_render_blocks(self.ptr, length, block, _fptr(x) if self.inputs() else None, _fptr(out))
        return out

    def render_ticks(self, x=None, length=None):
        if self.inputs():
            x = np.ascontiguousarray(np.atleast_2d(np.asarray(x, dtype=np.float32)))
            length = x.shape[1]
        out = np.zeros((self.outputs(), length), dtype=np.float32)
        lib().o_render_ticks(self.ptr, length, _fptr(x) if self.inputs() else None, _fptr(out))
        return out

    # --- graph notation (combinator.rs:289-488)
    def __rshift__(self, other): return Node(lib().o_pipe(self.ptr, other.ptr), (self, other))
    def __or__(self, other): return Node(lib().o_stack(self.ptr, other.ptr), (self, other))
    def __and__(self, other): return Node(lib().o_bus(self.ptr, other.ptr), (self, other))      # `&` combinator.rs:447
    def __xor__(self, other): return Node(lib().o_branch(self.ptr, other.ptr), (self, other))   # `^` combinator.rs:426
    def __invert__(self): return Node(lib().o_thru(self.ptr), (self,))                          # `!` combinator.rs:404

    def _bin(self, other, op, uop, scalar=None):
        if isinstance(other, Node):
            return Node(lib().o_binop(op, self.ptr, other.ptr), (self, other))
        return Node(lib().o_unop(uop, self.ptr, float(other if scalar is None else scalar)), (self,))

    def __mul__(self, o): return self._bin(o, MUL, MUL_SCALAR)
    def __rmul__(self, o): return self._bin(o, MUL, MUL_SCALAR)
    def __add__(self, o): return self._bin(o, ADD, ADD_SCALAR)
    def __radd__(self, o): return self._bin(o, ADD, ADD_SCALAR)
    def __sub__(self, o):
        if isinstance(o, Node):
            return self._bin(o, SUB, None)
        return self._bin(o, None, ADD_SCALAR, scalar=-float(o))  # combinator.rs: `x - f32` = AddScalar(-y)
    def __rsub__(self, o): return self._bin(o, None, NEG_ADD_SCALAR)
    def __neg__(self): return Node(lib().o_unop(NEG, self.ptr, 0.0), (self,))

    # builders (combinator.rs:263-267 `.phase()`, `.seed()`)
    def phase(self, p):
        lib().o_sine_set_phase(self.ptr, float(p))
        return self

    def seed(self, s):
        lib().o_noise_set_seed(self.ptr, int(s) & (2**64 - 1))
        return self

    def set_value(self, value):
        """Shared::set_value (shared.rs:98-101) of the variable a `var(..)` node reads."""
        lib().o_var_set(self.ptr, float(value))
        return self

    def wave_phase(self, p):
        lib().o_wavesynth_set_phase(self.ptr, float(p))
        return self


# --- prelude32 opcodes -------------------------------------------------------------------------------------
def constant(*v):
    a = np.asarray(v, dtype=np.float32).reshape(-1)
    return Node(lib().o_constant(a.size, _fptr(a)))


dc = constant
def zero(): return constant(0.0)                                   # prelude32.rs:129
def pass_(): return Node(lib().o_pass())
def multipass(n): return Node(lib().o_multipass(n))               # prelude32.rs:147
def sink(): return Node(lib().o_sink(1))                           # prelude32.rs:170
def multisink(n): return Node(lib().o_sink(n))
def split(n): return Node(lib().o_split(1, n))                     # prelude32.rs:1103
def multisplit(m, n): return Node(lib().o_split(m, n))
def join(n): return Node(lib().o_join(1, n))                       # prelude32.rs:1130
def multijoin(m, n): return Node(lib().o_join(m, n))
def reverse(n): return Node(lib().o_reverse(n))                    # prelude32.rs:190
def impulse(n=1): return Node(lib().o_impulse(n))                  # prelude32.rs:2480
def feedback(x): return Node(lib().o_feedback(x.ptr, None, 0), (x,))            # prelude32.rs:1040
def feedback2(x, y): return Node(lib().o_feedback(x.ptr, y.ptr, 0), (x, y))     # prelude32.rs:1061
def fdn(x): return Node(lib().o_feedback(x.ptr, None, 1), (x,))                 # prelude32.rs:1323
def fdn2(x, y): return Node(lib().o_feedback(x.ptr, y.ptr, 1), (x, y))          # prelude32.rs:1340
METER_MODES = dict(sample=0, peak=1, rms=2)
def meter(mode, timescale=0.1): return Node(lib().o_meter(METER_MODES[mode], timescale, 0))    # prelude32.rs:300 meter(Meter::Peak(t))
def monitor(mode, timescale=0.1): return Node(lib().o_meter(METER_MODES[mode], timescale, 1))  # prelude32.rs monitor(&shared, meter)
def meter_level(n): return np.float32(lib().o_meter_level(n.ptr))
def var(value): return Node(lib().o_var(value))                                               # prelude32.rs var(&shared)
def hold(variability, draws):                                                                 # prelude32.rs:830 (+ the Rnd stream)
    d = np.ascontiguousarray(draws, dtype=np.float64)
    return Node(lib().o_hold(variability, d.ctypes.data_as(C.POINTER(C.c_double)), d.size))
def hold_hz(f, variability, draws): return (pass_() | dc(f)) >> hold(variability, draws)     # prelude32.rs:843
def mixer(matrix):                                                                            # Mixer::new pan.rs:108
    m = np.ascontiguousarray(matrix, dtype=np.float32)
    return Node(lib().o_mixer(m.shape[1], m.shape[0], _fptr(m)))
def rotate(angle, gain):                                                                      # prelude32.rs:2432
    c, s_ = m_cosf(angle), m_sinf(angle)
    g = np.float32(gain)
    return mixer([[c * g, -s_ * g], [s_ * g, c * g]])
def var_fn(value, fn, outputs=1):                                                             # prelude32.rs var_fn(&shared, f)
    def cb(inp, out, _ctx):
        r = fn(np.float32(inp[0]))
        r = [r] if np.isscalar(r) else list(r)
        for k in range(outputs):
            out[k] = float(np.float32(r[k]))
    c = MAP_FN(cb)
    n = Node(lib().o_var_fn(value, outputs, C.cast(c, C.c_void_p), None))
    n._callback = c
    return n
def biquad_bank_coefs(coefs):                                                                 # prelude32.rs:2711 + Setting::biquad
    n = biquad_bank()
    for i in range(8):
        set_biquad_bank(n, i, coefs[i])
    return n
def envelope2(fn): return envelope_in(lambda t, i: fn(t, i[0]), 1)                           # prelude32.rs:625
def envelope3(fn): return envelope_in(lambda t, i: fn(t, i[0], i[1]), 2)                     # prelude32.rs:669
lfo3 = envelope3
def limiter(attack, release): return Node(lib().o_limiter(1, attack, release))                # prelude32.rs:1275
def limiter_stereo(attack, release): return Node(lib().o_limiter(2, attack, release))         # prelude32.rs:1286
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


MAP_FN = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p)


def map_(fn, inputs, outputs=1):                                   # prelude32.rs:332 map / shape_fn :1322
    """fn(frame: np.float32[inputs]) -> float or sequence of `outputs` floats plays the Rust closure (use np.float32
    arithmetic and the oracle's own libm inside it)."""
    def cb(inp, out, _ctx):
        r = fn(np.array([inp[k] for k in range(inputs)], dtype=np.float32))
        r = [r] if np.isscalar(r) else list(r)
        for k in range(outputs):
            out[k] = float(np.float32(r[k]))
    c = MAP_FN(cb)
    n = Node(lib().o_map(inputs, outputs, C.cast(c, C.c_void_p), None))
    n._callback = c
    return n


def shape_fn(fn):                                                 # prelude32.rs:1181: Shaper<ShapeFn<S>>
    def cb(inp, out, _ctx):
        out[0] = float(np.float32(fn(np.float32(inp[0]))))
    c = MAP_FN(cb)
    n = Node(lib().o_shape_fn(C.cast(c, C.c_void_p), None))
    n._callback = c
    return n


def declick(): return Node(lib().o_declick(0.010))                # prelude32.rs:1167
def declick_s(t): return Node(lib().o_declick(t))                 # prelude32.rs:1174


def _multi(kind, nodes, op=0):
    nodes = list(nodes)
    arr = (C.c_void_p * len(nodes))(*[n.ptr for n in nodes])
    return Node(lib().o_multi(kind, len(nodes), arr, op), nodes)


def busi(n, f): return _multi(0, [f(i) for i in range(n)])        # prelude32.rs busi/busf: MultiBus
def stacki(n, f): return _multi(1, [f(i) for i in range(n)])      # MultiStack
def branchi(n, f): return _multi(2, [f(i) for i in range(n)])     # MultiBranch
def sumi(n, f): return _multi(3, [f(i) for i in range(n)], ADD)   # Reduce<.., FrameAdd>
def pipei(n, f): return _multi(4, [f(i) for i in range(n)])       # Chain
def _frac(n, i): return np.float32(i / (n - 1)) if n > 1 else np.float32(0.5)              # prelude.rs busf etc.: closure of i / (n - 1)
def busf(n, f): return busi(n, lambda i: f(_frac(n, i)))
def stackf(n, f): return stacki(n, lambda i: f(_frac(n, i)))
def branchf(n, f): return branchi(n, lambda i: f(_frac(n, i)))
def sumf(n, f): return sumi(n, lambda i: f(_frac(n, i)))
def pipef(n, f): return pipei(n, lambda i: f(_frac(n, i)))
def sine(): return Node(lib().o_sine())
def sine_hz(f): return constant(f) >> sine()                       # prelude.rs:349
def noise(): return Node(lib().o_noise())
white = noise
def _fsvf(mode, f, q, gain=1.0): return Node(lib().o_fixed_svf(SVF_MODES[mode], f, q, gain))
def lowpass_hz(f, q): return _fsvf("lowpass", f, q)                # prelude.rs:2111
def highpass_hz(f, q): return _fsvf("highpass", f, q)
def bandpass_hz(f, q): return _fsvf("bandpass", f, q)
def notch_hz(f, q): return _fsvf("notch", f, q)
def peak_hz(f, q): return _fsvf("peak", f, q)
def allpass_hz(f, q): return _fsvf("allpass", f, q)
def bell_hz(f, q, gain): return _fsvf("bell", f, q, gain)
def lowshelf_hz(f, q, gain): return _fsvf("lowshelf", f, q, gain)
def highshelf_hz(f, q, gain): return _fsvf("highshelf", f, q, gain)
def svf(mode, f=440.0, q=1.0, gain=1.0): return Node(lib().o_svf(SVF_MODES[mode], f, q, gain))
def lowpass(): return svf("lowpass")                               # prelude.rs: Svf with (audio, cutoff, q) inputs
def biquad(a1, a2, b0, b1, b2): return Node(lib().o_biquad(a1, a2, b0, b1, b2))
def butterpass_hz(f): return Node(lib().o_butter_lowpass(1, f))
def butterpass(): return Node(lib().o_butter_lowpass(2, 440.0))
def resonator_hz(center, bandwidth): return Node(lib().o_resonator(1, center, bandwidth))  # prelude32.rs:534: passed straight to Resonator::new
def resonator(): return Node(lib().o_resonator(3, 440.0, 110.0))   # prelude32.rs:521: Resonator::new(440, 110), 3 inputs
def biquad_bank(): return Node(lib().o_biquad_bank())
def moog_hz(f, q): return Node(lib().o_moog(1, f, q))
def moog(): return Node(lib().o_moog(3, 1000.0, 0.1))             # prelude.rs:551-553
def fir(*w):
    a = np.asarray(w, dtype=np.float32).reshape(-1)
    return Node(lib().o_fir(a.size, _fptr(a)))
def tick(channels=1): return Node(lib().o_tick_node(channels))
def delay(t): return Node(lib().o_delay(float(np.float32(t))))   # prelude32.rs:893 delay(t: f32) -> Delay::new(t as f64)


# --- wavetables (wavetable.rs:44-123, 493-623).  Tables are DATA shared by the oracle and the engine in parity
# tests; this numpy builder follows make_wave / Wavetable::new (f64 spectrum set-up, f32 polar partials, inverse FFT,
# global peak normalisation).  The reference's microfft f32 butterflies are not restated: table bits are unpinned.
def _smooth5(x): return ((x * 6 - 15) * x + 10) * x * x * x


WT_KINDS = dict(saw=0, square=1, triangle=2, organ=4, soft_saw=5, hammond=6)


def make_wavetable_arrays(kind="saw"):
    """Wavetable::new(20, 20000, 4, ..) for a built-in table: the oracle's C restatement (oracle/o_wavetable.c: make_wave
    with the f32 radix-2 inverse FFT restated from microfft, global normalisation in f32)."""
    pitches = np.zeros(64, dtype=np.float32)
    lengths = np.zeros(64, dtype=np.int32)
    data = np.zeros(64 * 8192, dtype=np.float32)
    n = lib().o_make_wavetable(WT_KINDS[kind], 64, _fptr(pitches), lengths.ctypes.data_as(C.POINTER(C.c_int)), data.size, _fptr(data))
    assert n > 0
    offs = np.concatenate([[0], np.cumsum(lengths[:n])])
    return pitches[:n].copy(), [data[offs[k]:offs[k + 1]].copy() for k in range(n)]


def make_wavetable_arrays_f64(kind="saw", min_pitch=20.0, max_pitch=20000.0, tables_per_octave=4.0):
    """Independent double-precision construction of the same tables (numpy complex128 inverse FFT): the yardstick that
    bounds the f32 FFT restatement's error in tests/test_wavetable_build.py -- not used by any parity test."""
    def phase(i):
        if kind == "saw": return 0.0 if (i & 1) == 1 else 0.5
        if kind in ("square", "hammond"): return 0.0
        if kind == "triangle": return 0.5 if (i & 3) == 3 else 0.0
        if kind in ("organ", "soft_saw"): return 0.5 if (i & 3) == 3 else (0.0 if (i & 1) == 1 else 0.5)  # wavetable.rs:555-563
        raise KeyError(kind)

    def amplitude(i):
        if kind == "saw": return 1.0 / i
        if kind == "square": return 1.0 / i if (i & 1) == 1 else 0.0
        if kind == "triangle": return 1.0 / (i * i) if (i & 1) == 1 else 0.0
        if kind == "soft_saw": return 1.0 / (i * i)                                                       # :590
        z = (i & -i).bit_length() - 1
        j = i >> z
        if kind == "organ": return 1.0 / (i + j * j * j)                                                  # :564-568
        if kind == "hammond":                                                                             # :602-619
            f = 1.0 / ((z + 1) * (z + 1))
            if i <= 3: return 1.0
            return f if j in (1, 3) else (0.2 * f if j == 9 else 0.0)
        raise KeyError(kind)

    pitches, waves = [], []
    pitch, p_factor = float(min_pitch), 2.0 ** (1.0 / tables_per_octave)
    while pitch <= max_pitch:
        harmonics = int(np.floor(22000.0 / pitch))
        target = 4 * harmonics
        length = int(min(8192, max(32, 1 << max(0, (target - 1).bit_length()))))
        a = np.zeros(length, dtype=np.complex128)
        for i in range(1, harmonics + 1):
            f = pitch * i
            w = amplitude(i) * _smooth5(min(1.0, max(0.0, (f - 22000.0) / (20000.0 - 22000.0))))
            if w > 0.0:
                ang = np.float32(2 * np.pi * phase(i))
                a[i] = np.float32(w) * np.complex64(np.cos(ang) + 1j * np.sin(ang))
        wave = (np.fft.ifft(a) * length).imag.astype(np.float32)
        pitches.append(np.float32(pitch))
        waves.append(wave)
        pitch *= p_factor
    peak = max(float(np.max(np.abs(w))) for w in waves)
    z = np.float32(1.0 / peak)
    waves = [(w * z).astype(np.float32) for w in waves]
    return np.array(pitches, dtype=np.float32), waves


class Wavetable:
    _cache = {}

    def __init__(self, pitches, waves):
        self.pitches = np.ascontiguousarray(pitches, dtype=np.float32)
        self.lengths = np.array([len(w) for w in waves], dtype=np.int32)
        self.data = np.ascontiguousarray(np.concatenate(waves), dtype=np.float32)
        self.ptr = lib().o_wavetable_create(len(waves), _fptr(self.pitches),
                                            self.lengths.ctypes.data_as(C.POINTER(C.c_int)), _fptr(self.data))

    @classmethod
    def get(cls, kind):
        if kind not in cls._cache:
            cls._cache[kind] = cls(*make_wavetable_arrays(kind))
        return cls._cache[kind]


def wavesynth(kind="saw", outputs=1):
    t = Wavetable.get(kind)
    n = Node(lib().o_wavesynth(t.ptr, outputs))
    n._table = t  # keep the shared table alive
    return n


def saw(): return wavesynth("saw")                                    # prelude.rs:2003
def square(): return wavesynth("square")
def triangle(): return wavesynth("triangle")
def saw_hz(f): return constant(f) >> saw()
def organ(): return wavesynth("organ")                                # prelude32.rs organ / soft_saw / hammond
def soft_saw(): return wavesynth("soft_saw")
def hammond(): return wavesynth("hammond")
def organ_hz(f): return constant(f) >> organ()
def soft_saw_hz(f): return constant(f) >> soft_saw()
def hammond_hz(f): return constant(f) >> hammond()


def playwave_at(wave, channel, start_point, end_point, loop_point=None):       # prelude32.rs:2234
    w = np.ascontiguousarray(np.atleast_2d(wave), dtype=np.float32)
    n = Node(lib().o_waveplayer(_fptr(w), w.shape[0], w.shape[1], channel, start_point, end_point,
                                -1 if loop_point is None else loop_point))
    n._wave = w   # the node borrows the samples
    return n


def playwave(wave, channel, loop_point=None):                                    # prelude32.rs:2225
    return playwave_at(wave, channel, 0, np.atleast_2d(wave).shape[1], loop_point)


def phasesynth(kind="saw"):
    t = Wavetable.get(kind)
    n = Node(lib().o_phasesynth(t.ptr))
    n._table = t
    return n


def pulse():
    """PulseWave::new (wavetable.rs:452-459): built from the oracle's own combinators, wrapped under ID 44."""
    inner = (wavesynth("saw", 2) | pass_()) >> (pass_() | (pass_() + pass_()) >> phasesynth("saw")) >> pass_() - pass_()
    n = Node(lib().o_wrap(inner.ptr, 44), (inner,))
    return n
def adsr_live(a, d, s, r): return Node(lib().o_adsr_live(a, d, s, r))  # adsr.rs:21
def pan(p): return Node(lib().o_panner(1, p))                         # prelude.rs:1250


SHAPES = dict(clip=0, clip_to=1, tanh=2, atan=3, softsign=4, crush=5, soft_crush=6, adaptive_tanh=7)
SHAPES.update({f"adaptive_{k}": 8 + v for k, v in list(SHAPES.items())[:7] if k != "tanh"})
OSCS = dict(ramp=0, poly_saw=1, poly_square=2, poly_pulse=3)


def lowpole_hz(f): return Node(lib().o_onepole(0, 1, f))
def lowpole(): return Node(lib().o_onepole(0, 2, 440.0))
def highpole_hz(f): return Node(lib().o_onepole(1, 1, f))
def highpole(): return Node(lib().o_onepole(1, 2, 440.0))
def dcblock_hz(f): return Node(lib().o_onepole(2, 1, f))
def allpole_delay(d): return Node(lib().o_onepole(3, 1, d))
def allpole(): return Node(lib().o_onepole(3, 2, 1.0))
def pinkpass(): return Node(lib().o_pinkpass())
def lowrez_hz(cutoff, q): return Node(lib().o_rez(1, 0.0, cutoff, q))        # prelude.rs:2566
def lowrez(): return Node(lib().o_rez(3, 0.0, 440.0, 1.0))                    # prelude.rs:2559 Rez::new(0, 440, 1)
def bandrez_hz(center, q): return Node(lib().o_rez(1, 1.0, center, q))       # prelude.rs:2590
def bandrez(): return Node(lib().o_rez(3, 1.0, 440.0, 1.0))                   # prelude.rs:2583
def follow(t): return Node(lib().o_follow(t))                                # prelude32.rs:1251
def afollow(a, r): return Node(lib().o_afollow(a, r))                        # prelude32.rs:1266
def mls_bits(n): return Node(lib().o_mls(n))                                 # prelude32.rs:772
def mls(): return mls_bits(29)
ENV_FN = C.CFUNCTYPE(None, C.c_float, C.POINTER(C.c_float), C.c_void_p)


def envelope(fn, outputs=1, interval=0.002):                                  # prelude32.rs:581 envelope / :604 lfo
    """fn(t: np.float32) -> float or sequence of `outputs` floats plays the Rust closure.  Use np.float32 arithmetic and
    the oracle's own libm (m_expf, m_sinf, ...) inside it so that it computes what the f32 closure computes."""
    def cb(t, out, _ctx):
        r = fn(np.float32(t))
        r = [r] if np.isscalar(r) else list(r)
        for k in range(outputs):
            out[k] = float(np.float32(r[k]))
    c = ENV_FN(cb)
    n = Node(lib().o_envelope(np.float32(interval), outputs, C.cast(c, C.c_void_p), None))
    n._callback = c  # keep the trampoline alive as long as the node
    return n


lfo = envelope


def envelope_c(symbol, outputs=1, interval=0.002):
    """envelope(..) whose closure is one of the oracle's own C functions (`symbol`, signature o_env_fn): no Python callback per envelope
    sample -- for CPU timings of graphs with closures (tests/criterion_graphs.py)."""
    L = lib()
    return Node(L.o_envelope(np.float32(interval), outputs, C.cast(getattr(L, symbol), C.c_void_p), None))


ENVIN_FN = C.CFUNCTYPE(None, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p)


def envelope_in(fn, inputs, outputs=1, interval=0.002):                       # prelude32.rs:716 envelope_in / :732 lfo_in
    """fn(t, [inputs...]) -> float or sequence; envelope2 / lfo2 are the 1-input case (prelude32.rs:625-661)."""
    def cb(t, inp, out, _ctx):
        r = fn(np.float32(t), [np.float32(inp[k]) for k in range(inputs)])
        r = [r] if np.isscalar(r) else list(r)
        for k in range(outputs):
            out[k] = float(np.float32(r[k]))
    c = ENVIN_FN(cb)
    n = Node(lib().o_envelope_in(np.float32(interval), inputs, outputs, C.cast(c, C.c_void_p), None))
    n._callback = c
    return n


lfo_in = envelope_in
def lfo2(fn): return envelope_in(lambda t, i: fn(t, i[0]), 1)
def m_expf(x): return np.float32(lib().o_math_expf(float(x)))
def m_sinf(x): return np.float32(lib().o_math_sinf(float(x)))
def m_cosf(x): return np.float32(lib().o_math_cosf(float(x)))


def pluck(frequency, gain_per_second, damping, excitation):                    # prelude32.rs:1812 (+ the Rnd stream)
    e = np.ascontiguousarray(excitation, dtype=np.float32)
    return Node(lib().o_pluck(frequency, gain_per_second, damping, _fptr(e), e.size))
def dsf_saw(): return Node(lib().o_dsf(2, 1.0, 0.5))                         # prelude32.rs:1773
def dsf_saw_r(r): return Node(lib().o_dsf(1, 1.0, r))                        # prelude32.rs:1781
def dsf_square(): return Node(lib().o_dsf(2, 2.0, 0.5))                      # prelude32.rs:1789
def dsf_square_r(r): return Node(lib().o_dsf(1, 2.0, r))                     # prelude32.rs:1797
def resample(x): return Node(lib().o_resample(x.ptr), [x])                    # prelude32.rs:1021
def oversample(x): return Node(lib().o_oversample(x.ptr), [x])                # prelude32.rs:983                                               # prelude32.rs:784
def morph(): return Node(lib().o_morph(440.0, 1.0, 0.0))


def tap(min_delay, max_delay): return Node(lib().o_tap(0, min_delay, max_delay))                  # prelude.rs:910
def tap_linear(min_delay, max_delay): return Node(lib().o_tap(1, min_delay, max_delay))           # prelude.rs:948
def allnest_c(coefficient, x): return Node(lib().o_allnest(coefficient, x.ptr), (x,))             # prelude.rs allnest_c
def allnest(x): return Node(lib().o_allnest2(x.ptr), (x,))                                        # prelude32.rs:1112
def multitap(n, min_delay, max_delay): return Node(lib().o_multitap(0, n, min_delay, max_delay))  # prelude32.rs:928
def multitap_linear(n, min_delay, max_delay): return Node(lib().o_multitap(1, n, min_delay, max_delay))  # :965
def multitick(n): return tick(n)                                                                  # prelude32.rs:878
def panner(): return Node(lib().o_panner(2, 0.0))                                                 # prelude32.rs:1223


def shape(kind, p0=1.0, p1=0.0): return Node(lib().o_shaper(SHAPES[kind], p0, p1))          # prelude.rs:1194
def shape_adaptive(inner, p0, p1, timescale):  # shape(Adaptive::new(timescale, S)) shape.rs:173-183
    return Node(lib().o_shaper_adaptive(SHAPES[inner], p0, p1, timescale))
def ramp(): return Node(lib().o_phase_osc(0))
def poly_saw(): return Node(lib().o_phase_osc(1))
def poly_square(): return Node(lib().o_phase_osc(2))
def poly_pulse(): return Node(lib().o_phase_osc(3))
def rossler(): return Node(lib().o_chaos(0))
def lorenz(): return Node(lib().o_chaos(1))


def nlbiquad(dirty, inputs, mode, shape_kind, p0=1.0, p1=0.0, center=440.0, q=1.0, gain=1.0):
    """dbell_hz(Tanh(1.0), 1000, 10, 2) == nlbiquad(True, 1, "bell", "tanh", 1.0, 0, 1000, 10, 2) etc. (prelude.rs:2912-3100)"""
    return Node(lib().o_nlbiquad(int(dirty), inputs, BQ_KINDS[mode], SHAPES[shape_kind], p0, p1, center, q, gain))


# waveshapes, and the opcodes the prelude composes from others (same spelling as fundsp_amd.graph)
def Clip(h=1.0): return ("clip", h, 0.0)
def ClipTo(lo, hi): return ("clip_to", lo, hi)
def Tanh(h=1.0): return ("tanh", h, 0.0)
def Atan(h=1.0): return ("atan", h, 0.0)
def Softsign(h=1.0): return ("softsign", h, 0.0)
def Crush(levels): return ("crush", levels, 0.0)
def SoftCrush(levels): return ("soft_crush", levels, 0.0)
def fresonator_hz(s, center, q): return nlbiquad(False, 1, "resonator", s[0], s[1], s[2], center, q)     # prelude32.rs:2653
def flowpass_hz(s, cutoff, q): return nlbiquad(False, 1, "lowpass", s[0], s[1], s[2], cutoff, q)         # :2593
def fhighpass_hz(s, cutoff, q): return nlbiquad(False, 1, "highpass", s[0], s[1], s[2], cutoff, q)       # :2545
def fbell_hz(s, center, q, gain): return nlbiquad(False, 1, "bell", s[0], s[1], s[2], center, q, gain)   # :2496
def dresonator_hz(s, center, q): return nlbiquad(True, 1, "resonator", s[0], s[1], s[2], center, q)      # :2623
def dlowpass_hz(s, cutoff, q): return nlbiquad(True, 1, "lowpass", s[0], s[1], s[2], cutoff, q)          # :2569
def dhighpass_hz(s, cutoff, q): return nlbiquad(True, 1, "highpass", s[0], s[1], s[2], cutoff, q)        # :2521
def dbell_hz(s, center, q, gain): return nlbiquad(True, 1, "bell", s[0], s[1], s[2], center, q, gain)    # :2469
def _svf_q(mode, q): return (multipass(2) | dc(q)) >> svf(mode, 440.0, q)                                # prelude.rs:2127-2140
def lowpass_q(q): return _svf_q("lowpass", q)
def highpass_q(q): return _svf_q("highpass", q)
def bandpass_q(q): return _svf_q("bandpass", q)
def notch_q(q): return _svf_q("notch", q)
def peak_q(q): return _svf_q("peak", q)
def allpass_q(q): return _svf_q("allpass", q)
def bell_q(q, gain): return (multipass(2) | dc(q, gain)) >> svf("bell", 440.0, q, gain)                  # prelude.rs:2425-2446
def lowshelf_q(q, gain): return (multipass(2) | dc(q, gain)) >> svf("lowshelf", 440.0, q, gain)
def highshelf_q(q, gain): return (multipass(2) | dc(q, gain)) >> svf("highshelf", 440.0, q, gain)
def lowrez_q(q): return (multipass(2) | dc(q)) >> lowrez()                                               # prelude32.rs:2168
def bandrez_q(q): return (multipass(2) | dc(q)) >> bandrez()
def moog_q(q): return (multipass(2) | dc(q)) >> Node(lib().o_moog(3, 1000.0, q))                         # prelude32.rs:560
def morph_hz(f, q, m): return (pass_() | dc(f, q, m)) >> morph()                                         # prelude32.rs:2218
def pink(): return white() >> pinkpass()                                                                 # prelude32.rs:1299
def brown(): return white() >> lowpole_hz(10.0) * dc(13.7)                                               # prelude32.rs:1305
def dcblock(): return dcblock_hz(10.0)                                                                   # prelude32.rs:1160
def clip(): return shape("clip", 1.0)                                                                    # prelude32.rs:1201
def clip_to(lo, hi): return shape("clip_to", lo, hi)
def ramp_hz(f): return dc(f) >> ramp()
def poly_saw_hz(f): return dc(f) >> poly_saw()
def poly_square_hz(f): return dc(f) >> poly_square()
def poly_pulse_hz(f, width): return dc(f, width) >> poly_pulse()


def flanger(feedback_amount, minimum_delay, maximum_delay, delay_f):                                    # prelude.rs:2719-2730
    return pass_() & feedback2((pass_() | lfo(delay_f)) >> tap(minimum_delay, maximum_delay), shape("tanh", feedback_amount))
def phaser(feedback_amount, phase_f):                                                                   # prelude.rs:2743-2753
    f32 = np.float32
    def wrapped(t):
        p = f32(phase_f(t))
        p = f32(min(max(p, f32(0.0)), f32(1.0)))
        return f32(2.0) * (f32(1.0) - p) + f32(20.0) * p
    return phaser_lfo(feedback_amount, lfo(wrapped))
def phaser_lfo(feedback_amount, wrapped_lfo):
    """phaser(..) around an lfo node that already computes lerp(2.0, 20.0, clamp01(phase_f(t))) (e.g. envelope_c of a C closure)"""
    return pass_() & feedback((pass_() | wrapped_lfo) >> pipei(10, lambda _i: add(0.0, 0.1) >> ~allpole())
                              >> (mul(feedback_amount) | sink()))


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
    return reverb4_stereo_delays([f(d) * scale for d in REVERB4_DELAYS], time)


def reverb3_stereo(time, diffusion, make_filter):                                                      # prelude.rs:1858
    """make_filter() builds one fresh 1-in 1-out loop filter (the reference clones `filter` 16 times)."""
    fs = [make_filter() for _ in range(16)]
    arr = (C.c_void_p * 16)(*[f.ptr for f in fs])
    return Node(lib().o_reverb3(time, diffusion, arr), fs)


def reverb_stereo(room_size, time, damping): return Node(lib().o_reverb_stereo(room_size, time, damping))  # prelude.rs:1732


def reverb_stereo_params(room_size, time, damping, sample_rate):
    w = np.zeros(3, np.float32); d = np.zeros(32, np.int32); wl = np.zeros(32, np.float32); wr = np.zeros(32, np.float32)
    lib().o_reverb_stereo_params(room_size, time, damping, sample_rate, _fptr(w), d.ctypes.data_as(C.POINTER(C.c_int)),
                                 _fptr(wl), _fptr(wr))
    return w, d, wl, wr


def set_biquad_bank(node, index, coefs):
    lib().o_biquad_bank_set(node.ptr, index, *[float(c) for c in coefs])


def wave_render(sample_rate, duration, node):
    """Wave::render (wave.rs:441-466) -> [channels][length] float32."""
    length = int(round(duration * sample_rate))
    out = np.zeros((node.outputs(), length), dtype=np.float32)
    n = lib().o_wave_render(node.ptr, float(sample_rate), float(duration), _fptr(out), length)
    assert n == length
    return out


def svf_coefs(mode, sr, cutoff, q, gain=1.0):
    out = np.zeros(6, dtype=np.float32)
    lib().o_svf_coefs(SVF_MODES[mode], sr, cutoff, q, gain, _fptr(out))
    return out


def biquad_coefs(kind, sr, f, q=1.0, gain=1.0):
    out = np.zeros(5, dtype=np.float32)
    lib().o_biquad_coefs(BQ_KINDS[kind], sr, f, q, gain, _fptr(out))
    return out


def moog_coefs(sr, cutoff, q):
    out = np.zeros(3, dtype=np.float32)
    lib().o_moog_coefs(sr, cutoff, q, _fptr(out))
    return out


def bank_render(config, params, seeds, frames, sample_rate=48000.0, process_mode=True, out_layout=1, threads=1,
                store=True, lib=None, fast=False):
    """Render V voices of BASELINE config 2 or 3. params: list of float32 [V] arrays. Returns (out, seconds).
    `lib`: another build of the same oracle (bench.py's -march=native flavour) whose o_bank_render to call.
    `fast`: the monomorphised SIMD form of the config-3 process() path (oracle/o_fast.c), bit-identical to the tree walk."""
    V = len(seeds)
    ps = [np.ascontiguousarray(p, dtype=np.float32) for p in params] + [None] * (4 - len(params))
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    job = BankJob(config, int(process_mode), out_layout if store else 2, threads, sample_rate, V, frames,
                  *[_fptr(p) for p in ps], seeds.ctypes.data_as(C.POINTER(C.c_uint64)))
    out = None
    if store:
        out = np.zeros((frames, V) if out_layout == 1 else (V, frames), dtype=np.float32)
    L = lib or globals()["lib"]()
    if fast and config == 2:     # BiquadBank<f32x8>: eight voices per SIMD instruction (oracle/o_fast.c)
        L.o_biquad_bank8_render.restype = C.c_double
        L.o_biquad_bank8_render.argtypes = [C.POINTER(BankJob), C.POINTER(C.c_float)]
        secs = L.o_biquad_bank8_render(C.byref(job), _fptr(out))
        assert secs >= 0.0, "o_biquad_bank8_render takes config 2 in process mode only"
        return out, secs
    if fast:
        L.o_bank_render_fast.restype = C.c_double
        L.o_bank_render_fast.argtypes = [C.POINTER(BankJob), C.POINTER(C.c_float)]
        secs = L.o_bank_render_fast(C.byref(job), _fptr(out))
        assert secs >= 0.0, "o_bank_render_fast takes config 3 in process mode only"
    else:
        secs = L.o_bank_render(C.byref(job), _fptr(out))
    return out, secs


class C4Job(C.Structure):
    _fields_ = [
        ("threads", C.c_int), ("fast", C.c_int), ("gate_var", C.c_int), ("n_plan", C.c_int),
        ("sample_rate", C.c_double), ("voices", C.c_size_t), ("frames", C.c_size_t),
        ("p0", C.POINTER(C.c_float)), ("p1", C.POINTER(C.c_float)), ("p2", C.POINTER(C.c_float)), ("p3", C.POINTER(C.c_float)),
        ("seed", C.POINTER(C.c_uint64)), ("adsr", C.c_float * 4), ("gate", C.POINTER(C.c_float)), ("plan", C.POINTER(C.c_float)),
    ]


def c4_bank_render(p, adsr, frames, sample_rate=48000.0, gate=None, plan=None, threads=1, fast=True, store=True, lib=None):
    """V voices of BASELINE config 4 through the C bank driver (oracle/o_fast.c): `gate` [frames] = the stream-gate shape
    (`... * adsr_live(..)`, the gate is the graph's input), `plan` [(value, frames), ..] = the reference's shape `var(g) >> adsr_live(..)`
    with the variable set before every entry.  fast: the monomorphised process() (fundsp_oracle.c o_c4_block), else the tree walk.
    -> (out [V][2][frames] or None, seconds)"""
    V = len(p["seed"])
    arrs = [np.ascontiguousarray(p[k], dtype=np.float32) for k in ("f", "fc", "q", "pan")]
    seeds = np.ascontiguousarray(p["seed"], dtype=np.uint64)
    g = None if gate is None else np.ascontiguousarray(gate, dtype=np.float32).reshape(-1)
    pl = None if plan is None else np.ascontiguousarray([[v, n] for v, n in plan], dtype=np.float32).reshape(-1)
    assert (g is None) != (pl is None) and (g is None or g.size >= frames)
    job = C4Job(threads, int(fast), int(pl is not None), 0 if pl is None else len(plan), sample_rate, V, frames,
                *[_fptr(a) for a in arrs], seeds.ctypes.data_as(C.POINTER(C.c_uint64)), (C.c_float * 4)(*[float(a) for a in adsr]), _fptr(g), _fptr(pl))
    out = np.zeros((V, 2, frames), dtype=np.float32) if store else None
    L = lib or globals()["lib"]()
    L.o_c4_bank_render.restype = C.c_double
    L.o_c4_bank_render.argtypes = [C.POINTER(C4Job), C.POINTER(C.c_float)]
    secs = L.o_c4_bank_render(C.byref(job), _fptr(out))
    assert secs >= 0.0
    return out, secs


def reverb_bank_render(instances, x, sample_rate=48000.0, room=10.0, time=2.0, damping=0.5, threads=1, fast=True, store=True, lib=None):
    """`instances` x reverb_stereo(room, time, damping) on the stereo input x [2][frames] -> (out [instances][2][frames] or None, seconds)"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    assert x.ndim == 2 and x.shape[0] == 2
    frames = x.shape[1]
    out = np.zeros((instances, 2, frames), dtype=np.float32) if store else None
    L = lib or globals()["lib"]()
    L.o_reverb_bank_render.restype = C.c_double
    L.o_reverb_bank_render.argtypes = [C.c_int, C.c_int, C.c_double, C.c_size_t, C.c_size_t, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    secs = L.o_reverb_bank_render(threads, int(fast), sample_rate, instances, frames, room, time, damping, _fptr(x), _fptr(out))
    return out, secs


def graph_bank_render(which, params, instances, x, sample_rate=48000.0, threads=1, store=True, lib=None, fast=False):
    """CPU leg (fast: the monomorphised block form, else the tree walk) of round 6's lane-per-frame kinds (o_fast.c o_graph_bank_render): which = "reverb3" (params = time, diffusion, lowpole cutoff;
    x [2][frames]) or "fdn16" (params = 16 delays + 3 FIR weights; x [1][frames]) -> (out [instances][outputs][frames] or None, seconds)"""
    k = {"reverb3": 0, "fdn16": 1}[which]
    nch = 2 if k == 0 else 1
    x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float32)
    assert x.shape[0] == nch
    frames = x.shape[1]
    p = (C.c_double * len(params))(*[float(v) for v in params])
    out = np.zeros((instances, nch, frames), dtype=np.float32) if store else None
    L = lib or globals()["lib"]()
    L.o_graph_bank_render.restype = C.c_double
    L.o_graph_bank_render.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_double, C.c_size_t, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    secs = L.o_graph_bank_render(threads, k, int(fast), p, sample_rate, instances, frames, _fptr(x), _fptr(out))
    return out, secs


FADE_POWER, FADE_SMOOTH = 0, 1


class Sequencer:
    """sequencer.rs Sequencer (ReplayMode::None): push events, render.  Every event may have its own input stream."""

    def __init__(self, inputs, outputs, sample_rate):
        self.ptr = lib().o_seq_new(inputs, outputs, float(sample_rate))
        self.inputs, self.outputs, self.n = inputs, outputs, 0

    def __del__(self):
        if self.ptr and _lib is not None:
            _lib.o_seq_free(self.ptr)
            self.ptr = None

    def push(self, start, end, fade, fade_in, fade_out, node):
        idx = lib().o_seq_push(self.ptr, float(start), float(end), int(fade), float(fade_in), float(fade_out), node.ptr)
        if idx < 0:
            raise ValueError("sequencer: arity mismatch or fade longer than the event")
        node._owned = False
        self._keep = getattr(self, "_keep", []) + [node]
        self.n += 1
        return idx

    def render(self, length, process=True, inputs=None):
        """-> (mix [outputs][length], per_event [events][outputs][length]); inputs [events][inputs][length] or None"""
        x = np.zeros((max(self.n, 1), max(self.inputs, 1), length), dtype=np.float32)
        if inputs is not None and self.inputs:
            x[:self.n, :self.inputs] = np.asarray(inputs, dtype=np.float32).reshape(self.n, self.inputs, length)
        mix = np.zeros((self.outputs, length), dtype=np.float32)
        per = np.zeros((max(self.n, 1), self.outputs, length), dtype=np.float32)
        lib().o_seq_render(self.ptr, length, 1 if process else 0, _fptr(x), _fptr(mix), _fptr(per))
        return mix, per[:self.n]

    def time(self): return lib().o_seq_time(self.ptr)

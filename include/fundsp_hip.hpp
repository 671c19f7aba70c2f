// fundsp_hip.hpp -- header-only C++17 host side above the C ABI (fundsp_hip.h).
//
// FunDSP is compiled code (Rust); this image has no Rust toolchain, so the host-side mirror of the reference's operator
// interface for the voice path is C++ (INTEGRATION.md shows the Rust binding a maintainer would add).  The names and
// argument meanings follow the reference:
//   * `An` + the graph operators of src/combinator.rs:289-488   ( >>  |  &  ^  !  *  +  - , unary - )
//   * the prelude32 opcodes of src/prelude32.rs                    ( sine_hz, lowpass_hz, moog, saw, adsr_live, pan, ... )
//   * the AudioNode surface of src/audionode.rs:29-369 on `Bank`   ( inputs, outputs, reset, set_sample_rate, tick,
//     process, set_hash -> set_seed, Clone -> clone ), and Wave::render (src/wave.rs:441-466) as `render`.
// A graph is the combinator TYPE FunDSP would build, spelled with the device templates of fundsp_amd/csrc/fd_nodes.hpp,
// plus its per-node parameters; Bank::from_graph compiles it into one fused kernel (fdsp_graph_compile_src).
//
// Error behaviour: arity mismatches are compile-time errors in Rust; here they throw fundsp_hip::Error when the graph
// is put together (host) or compiled (device).  Errors of the C ABI become fundsp_hip::Error with the code and message.
// AudioNode::process itself is infallible in the reference; Bank::process throws only on misuse / device failure.
//
// Every numeric opcode argument is a `P`: one value for all voices, or one value per voice.
#ifndef FUNDSP_HIP_HPP
#define FUNDSP_HIP_HPP

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "fundsp_hip.h"

namespace fundsp_hip {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error("fundsp_hip error " + std::to_string(c) + ": " + m), code(c) {}
};
inline void check(int rc) {
    if (rc < 0) throw Error(rc, fdsp_last_error());
}

constexpr size_t MAX_BUFFER_SIZE = FDSP_MAX_BUFFER_SIZE;  // src/lib.rs:48
constexpr double DEFAULT_SR = FDSP_DEFAULT_SR;            // src/lib.rs:42

// one value for every voice, or one per voice
struct P {
    std::vector<float> v;
    P(float x) : v{x} {}
    P(double x) : v{(float)x} {}
    P(int x) : v{(float)x} {}
    P(std::vector<float> per_voice) : v(std::move(per_voice)) {}
};

struct Param {
    std::vector<int> path;  // child indices from the root (0 = left / inner, 1 = right), as the C ABI names slots
    std::string field;
    std::vector<float> values;
    bool is_u64 = false;
    std::vector<uint64_t> u64s;
    std::string slot() const {
        std::string s;
        for (size_t i = 0; i < path.size(); i++) s += (i ? "." : "") + std::to_string(path[i]);
        return s + ":" + field;
    }
};

// An<X> of the reference: a node expression.  `type` is X.
class An {
 public:
    std::string type, source;
    int inputs = 0, outputs = 0, rings = 0;
    std::vector<Param> params;

    An() = default;
    An(std::string t, int nin, int nout, int nrings = 0) : type(std::move(t)), inputs(nin), outputs(nout), rings(nrings) {}
    An& with(const std::string& field, const P& p) {
        params.push_back(Param{{}, field, p.v, false, {}});
        return *this;
    }
    // a parameter of an enclosed object, e.g. the fields of an Envelope's functor live under child 0: "0:<field>"
    An& with_child(int child, const std::string& field, const P& p) {
        params.push_back(Param{{child}, field, p.v, false, {}});
        return *this;
    }
    // combinator.rs:263-267 `.phase(x)` / `.seed(x)`
    An phase(P p) const {
        An a = *this;
        a.with("has_initial_phase", 1.0f).with("initial_phase", p);
        return a;
    }
    An seed(uint64_t s) const {
        An a = *this;
        a.with("has_seed", 1.0f);
        Param q{{}, "seed", {}, true, {s}};
        a.params.push_back(q);
        return a;
    }
};

namespace detail {
inline std::string merge(const std::string& a, const std::string& b) {
    if (b.empty() || a.find(b) != std::string::npos) return a;
    if (a.empty() || b.find(a) != std::string::npos) return b;
    return a + "\n" + b;
}
inline void adopt(An& dst, const An& child, int index) {
    for (Param p : child.params) {
        p.path.insert(p.path.begin(), index);
        dst.params.push_back(std::move(p));
    }
    dst.rings += child.rings;
    dst.source = merge(dst.source, child.source);
}
inline An pair(const char* tmpl, const An& x, const An& y, int nin, int nout, const std::string& extra = "") {
    An a(std::string(tmpl) + "<" + extra + x.type + "," + y.type + ">", nin, nout);
    adopt(a, x, 0);
    adopt(a, y, 1);
    return a;
}
inline An unop(const An& x, const char* u, const P* scalar) {
    An a("Unop<" + x.type + "," + u + ">", x.inputs, x.outputs);
    adopt(a, x, 0);
    if (scalar) a.with("scalar", *scalar);
    return a;
}
inline An leaf(const std::string& t, int nin, int nout, int rings = 0) { return An(t, nin, nout, rings); }
[[noreturn]] inline void arity(const std::string& what) { throw Error(FDSP_EINVAL, what); }
}  // namespace detail

// ---- graph operators (combinator.rs:289-488) ---------------------------------------------------------------------
inline An operator>>(const An& x, const An& y) {  // Pipe
    if (x.outputs != y.inputs)
        detail::arity("Pipe arity mismatch: " + std::to_string(x.outputs) + " outputs >> " + std::to_string(y.inputs) + " inputs");
    return detail::pair("Pipe", x, y, x.inputs, y.outputs);
}
inline An operator|(const An& x, const An& y) { return detail::pair("Stack", x, y, x.inputs + y.inputs, x.outputs + y.outputs); }
inline An operator&(const An& x, const An& y) {  // Bus
    if (x.inputs != y.inputs || x.outputs != y.outputs) detail::arity("Bus arity mismatch");
    return detail::pair("Bus", x, y, x.inputs, x.outputs);
}
inline An operator^(const An& x, const An& y) {  // Branch
    if (x.inputs != y.inputs) detail::arity("Branch arity mismatch");
    return detail::pair("Branch", x, y, x.inputs, x.outputs + y.outputs);
}
inline An operator!(const An& x) {  // Thru
    An a("Thru<" + x.type + ">", x.inputs, x.inputs);
    detail::adopt(a, x, 0);
    return a;
}
inline An binop(const char* op, const An& x, const An& y) {
    if (x.outputs != y.outputs) detail::arity("Binop arity mismatch");
    return detail::pair("Binop", x, y, x.inputs + y.inputs, x.outputs, std::string(op) + ",");
}
inline An operator*(const An& x, const An& y) { return binop("OpMul", x, y); }
inline An operator+(const An& x, const An& y) { return binop("OpAdd", x, y); }
inline An operator-(const An& x, const An& y) { return binop("OpSub", x, y); }
inline An operator*(const An& x, const P& s) { return detail::unop(x, "UMulScalar", &s); }
inline An operator*(const P& s, const An& x) { return detail::unop(x, "UMulScalar", &s); }
inline An operator+(const An& x, const P& s) { return detail::unop(x, "UAddScalar", &s); }
inline An operator+(const P& s, const An& x) { return detail::unop(x, "UAddScalar", &s); }
inline An operator-(const An& x, const P& s) {  // `x - f32` = FrameAddScalar(-y)
    P neg = s;
    for (float& f : neg.v) f = -f;
    return detail::unop(x, "UAddScalar", &neg);
}
inline An operator-(const P& s, const An& x) { return detail::unop(x, "UNegAddScalar", &s); }
inline An operator-(const An& x) { return detail::unop(x, "UNeg", nullptr); }

// ---- prelude32 opcodes -------------------------------------------------------------------------------------------
inline An constant(std::initializer_list<P> v) {
    An a("Constant<" + std::to_string(v.size()) + ">", 0, (int)v.size());
    int i = 0;
    for (const P& p : v) a.with("value[" + std::to_string(i++) + "]", p);
    return a;
}
inline An constant(P v) { return constant({std::move(v)}); }
inline An dc(P v) { return constant({std::move(v)}); }
inline An dc(std::initializer_list<P> v) { return constant(v); }
inline An zero() { return constant(0.0f); }
inline An pass() { return detail::leaf("Pass", 1, 1); }
inline An multipass(int n) { return detail::leaf("MultiPass<" + std::to_string(n) + ">", n, n); }
inline An sink() { return detail::leaf("Sink<1>", 1, 0); }
inline An multisink(int n) { return detail::leaf("Sink<" + std::to_string(n) + ">", n, 0); }
inline An split(int n) { return detail::leaf("Split<" + std::to_string(n) + ">", 1, n); }
inline An join(int n) { return detail::leaf("Join<" + std::to_string(n) + ">", n, 1); }
inline An multisplit(int m, int n) { return detail::leaf("MultiSplit<" + std::to_string(m) + "," + std::to_string(n) + ">", m, m * n); }
inline An multijoin(int m, int n) { return detail::leaf("MultiJoin<" + std::to_string(m) + "," + std::to_string(n) + ">", m * n, m); }
inline An reverse(int n) { return detail::leaf("Reverse<" + std::to_string(n) + ">", n, n); }
inline An impulse(int n = 1) { return detail::leaf("Impulse<" + std::to_string(n) + ">", 0, n); }
inline An tick() { return detail::leaf("Tick<1>", 1, 1); }
inline An sine() { return detail::leaf("Sine", 1, 1); }
inline An sine_hz(P f) { return constant(std::move(f)) >> sine(); }  // prelude.rs:349
inline An noise() { return detail::leaf("Noise", 0, 1); }
inline An white() { return noise(); }
inline An mls_bits(int n) { return detail::leaf("Mls", 0, 1).with("bits", (float)n); }
inline An mls() { return mls_bits(29); }

inline An fixed_svf(int mode, P cutoff, P q, P gain = 1.0f) {
    return detail::leaf("FixedSvf", 1, 1).with("mode", (float)mode).with("cutoff", cutoff).with("q", q).with("gain", gain);
}
inline An lowpass_hz(P f, P q) { return fixed_svf(FDSP_SVF_LOWPASS, f, q); }  // prelude.rs:2111
inline An highpass_hz(P f, P q) { return fixed_svf(FDSP_SVF_HIGHPASS, f, q); }
inline An bandpass_hz(P f, P q) { return fixed_svf(FDSP_SVF_BANDPASS, f, q); }
inline An notch_hz(P f, P q) { return fixed_svf(FDSP_SVF_NOTCH, f, q); }
inline An peak_hz(P f, P q) { return fixed_svf(FDSP_SVF_PEAK, f, q); }
inline An allpass_hz(P f, P q) { return fixed_svf(FDSP_SVF_ALLPASS, f, q); }
inline An bell_hz(P f, P q, P gain) { return fixed_svf(FDSP_SVF_BELL, f, q, gain); }
inline An lowshelf_hz(P f, P q, P gain) { return fixed_svf(FDSP_SVF_LOWSHELF, f, q, gain); }
inline An highshelf_hz(P f, P q, P gain) { return fixed_svf(FDSP_SVF_HIGHSHELF, f, q, gain); }
inline An svf(int mode) {  // Svf with (audio, cutoff, q[, gain]) inputs
    const int nin = mode >= FDSP_SVF_BELL ? 4 : 3;
    return detail::leaf("Svf<" + std::to_string(nin) + ">", nin, 1).with("mode", (float)mode);
}
inline An lowpass() { return svf(FDSP_SVF_LOWPASS); }
inline An highpass() { return svf(FDSP_SVF_HIGHPASS); }
inline An bandpass() { return svf(FDSP_SVF_BANDPASS); }
inline An notch() { return svf(FDSP_SVF_NOTCH); }
inline An peak() { return svf(FDSP_SVF_PEAK); }
inline An allpass() { return svf(FDSP_SVF_ALLPASS); }
inline An bell() { return svf(FDSP_SVF_BELL); }
inline An lowshelf() { return svf(FDSP_SVF_LOWSHELF); }
inline An highshelf() { return svf(FDSP_SVF_HIGHSHELF); }
inline An lowpass_q(P q) { return (multipass(2) | dc(q)) >> svf(FDSP_SVF_LOWPASS).with("q", q); }  // prelude.rs:2127
inline An highpass_q(P q) { return (multipass(2) | dc(q)) >> svf(FDSP_SVF_HIGHPASS).with("q", q); }
inline An bandpass_q(P q) { return (multipass(2) | dc(q)) >> svf(FDSP_SVF_BANDPASS).with("q", q); }
inline An morph() { return detail::leaf("Morph", 4, 1); }
inline An biquad(P a1, P a2, P b0, P b1, P b2) {
    return detail::leaf("Biquad", 1, 1).with("a1", a1).with("a2", a2).with("b0", b0).with("b1", b1).with("b2", b2);
}
inline An butterpass_hz(P f) { return detail::leaf("ButterLowpass<1>", 1, 1).with("cutoff", f); }
inline An butterpass() { return detail::leaf("ButterLowpass<2>", 2, 1); }
inline An resonator_hz(P center, P bandwidth) { return detail::leaf("Resonator<1>", 1, 1).with("center", center).with("q", bandwidth); }
inline An resonator() { return detail::leaf("Resonator<3>", 3, 1).with("center", 440.0f).with("q", 110.0f); }
inline An moog_hz(P f, P q) { return detail::leaf("Moog<1>", 1, 1).with("cutoff", f).with("q", q); }
inline An moog() { return detail::leaf("Moog<3>", 3, 1); }  // prelude.rs:551-553
inline An moog_q(P q) { return (multipass(2) | dc(q)) >> detail::leaf("Moog<3>", 3, 1).with("cutoff", 1000.0f).with("q", q); }
inline An fir(std::initializer_list<P> w) {
    An a("Fir<" + std::to_string(w.size()) + ">", 1, 1);
    int i = 0;
    for (const P& p : w) a.with("w[" + std::to_string(i++) + "]", p);
    return a;
}
inline An lowpole_hz(P f) { return detail::leaf("OnePole<OP_LOWPOLE,1>", 1, 1).with("cutoff", f); }
inline An lowpole() { return detail::leaf("OnePole<OP_LOWPOLE,2>", 2, 1); }
inline An highpole_hz(P f) { return detail::leaf("OnePole<OP_HIGHPOLE,1>", 1, 1).with("cutoff", f); }
inline An highpole() { return detail::leaf("OnePole<OP_HIGHPOLE,2>", 2, 1); }
inline An dcblock_hz(P f) { return detail::leaf("OnePole<OP_DCBLOCK,1>", 1, 1).with("cutoff", f); }
inline An dcblock() { return dcblock_hz(10.0f); }
inline An allpole_delay(P d) { return detail::leaf("OnePole<OP_ALLPOLE,1>", 1, 1).with("delay", d); }
inline An allpole() { return detail::leaf("OnePole<OP_ALLPOLE,2>", 2, 1); }
inline An pinkpass() { return detail::leaf("Pinkpass", 1, 1); }
inline An pink() { return white() >> pinkpass(); }                       // prelude32.rs:1299
inline An brown() { return white() >> lowpole_hz(10.0f) * dc(13.7f); }    // prelude32.rs:1305
inline An lowrez_hz(P c, P q) { return detail::leaf("Rez<1>", 1, 1).with("bandpass", 0.0f).with("cutoff", c).with("q", q); }
inline An bandrez_hz(P c, P q) { return detail::leaf("Rez<1>", 1, 1).with("bandpass", 1.0f).with("cutoff", c).with("q", q); }
inline An lowrez() { return detail::leaf("Rez<3>", 3, 1).with("bandpass", 0.0f); }
inline An bandrez() { return detail::leaf("Rez<3>", 3, 1).with("bandpass", 1.0f); }
inline An follow(P t) { return detail::leaf("Follow", 1, 1).with("response_time", t); }
inline An afollow(P a, P r) { return detail::leaf("AFollow", 1, 1).with("attack_time", a).with("release_time", r); }
inline An declick() { return detail::leaf("Declick", 1, 1).with("duration", 0.010f); }
inline An declick_s(P t) { return detail::leaf("Declick", 1, 1).with("duration", t); }
inline An limiter(P attack, P release) { return detail::leaf("Limiter<1>", 1, 1, 2).with("attack_time", attack).with("release_time", release); }
inline An limiter_stereo(P attack, P release) { return detail::leaf("Limiter<2>", 2, 2, 3).with("attack_time", attack).with("release_time", release); }
inline An var(P value) { return detail::leaf("Var", 0, 1).with("value", value); }

enum Shape { CLIP = 0, CLIP_TO, TANH, ATAN, SOFTSIGN, CRUSH, SOFT_CRUSH, ADAPTIVE_TANH,  // shape.rs:35-201
             ADAPTIVE /* + inner shape: Adaptive<S>, e.g. ADAPTIVE + ATAN */ };
inline An shape(Shape kind, P p0 = 1.0f, P p1 = 0.0f) {                                     // prelude.rs:1194
    return detail::leaf("Shaper", 1, 1).with("shape", (float)kind).with("shape_p0", p0).with("shape_p1", p1);
}
inline An clip() { return shape(CLIP, 1.0f); }
inline An clip_to(P lo, P hi) { return shape(CLIP_TO, lo, hi); }

inline An delay(P t) { return detail::leaf("Delay", 1, 1, 1).with("time", t); }
inline An tap(P lo, P hi) { return detail::leaf("TapT<false>", 2, 1, 1).with("min_delay", lo).with("max_delay", hi); }
inline An tap_linear(P lo, P hi) { return detail::leaf("TapT<true>", 2, 1, 1).with("min_delay", lo).with("max_delay", hi); }
inline An multitap(int n, P lo, P hi) { return detail::leaf("TapT<false," + std::to_string(n) + ">", 1 + n, 1, 1).with("min_delay", lo).with("max_delay", hi); }
inline An allnest_c(P coefficient, const An& x) {
    An a("AllNest<" + x.type + ">", 1, 1);
    detail::adopt(a, x, 0);
    a.with("coefficient", coefficient);
    return a;
}
inline An allnest(const An& x) {
    An a("AllNest<" + x.type + ",2>", 2, 1);
    detail::adopt(a, x, 0);
    return a;
}

inline An wavesynth(int set) { return detail::leaf("WaveSynth<" + std::to_string(set) + ">", 1, 1); }
inline An saw() { return wavesynth(0); }
inline An square() { return wavesynth(1); }
inline An triangle() { return wavesynth(2); }
inline An organ() { return wavesynth(4); }
inline An soft_saw() { return wavesynth(5); }
inline An hammond() { return wavesynth(6); }
inline An saw_hz(P f) { return constant(std::move(f)) >> saw(); }
inline An square_hz(P f) { return constant(std::move(f)) >> square(); }
inline An triangle_hz(P f) { return constant(std::move(f)) >> triangle(); }
inline An pulse() { return detail::leaf("PulseWave", 2, 1); }
inline An ramp() { return detail::leaf("PhaseOsc<OSC_RAMP>", 1, 1); }
inline An poly_saw() { return detail::leaf("PhaseOsc<OSC_POLYSAW>", 1, 1); }
inline An poly_square() { return detail::leaf("PhaseOsc<OSC_POLYSQUARE>", 1, 1); }
inline An poly_pulse() { return detail::leaf("PhaseOsc<OSC_POLYPULSE>", 2, 1); }
inline An rossler() { return detail::leaf("Chaos<false>", 1, 1); }
inline An lorenz() { return detail::leaf("Chaos<true>", 1, 1); }
inline An dsf_saw_r(P r) { return detail::leaf("Dsf<1>", 1, 1).with("harmonic_spacing", 1.0f).with("roughness", r); }
inline An dsf_square_r(P r) { return detail::leaf("Dsf<1>", 1, 1).with("harmonic_spacing", 2.0f).with("roughness", r); }
inline An adsr_live(P a, P d, P s, P r) {  // adsr.rs:21
    return detail::leaf("AdsrLive", 1, 1).with("attack", a).with("decay", d).with("sustain", s).with("release", r);
}
inline An pan(P p) { return detail::leaf("Panner", 1, 2).with("pan", p); }  // prelude.rs:1250
inline An panner() { return detail::leaf("PannerT<2>", 2, 2); }
inline An oversample(const An& x) {
    An a("Oversampler<" + x.type + ">", x.inputs, x.outputs);
    detail::adopt(a, x, 0);
    return a;
}
inline An resample(const An& x) {
    if (x.inputs != 0) detail::arity("resample: the enclosed node is a generator");
    An a("Resample<" + x.type + ">", 1, x.outputs);
    detail::adopt(a, x, 0);
    return a;
}

// closures: the Rust closure is a C++ functor type whose definition travels as `source` (contracts in fd_nodes.hpp)
// (functor parameters: .with_child(0, "<field>", value))
inline An envelope(const std::string& functor, const std::string& source, int outputs = 1) {  // envelope / lfo, prelude32.rs:581-611
    An a("Envelope<" + functor + ">", 0, outputs);
    a.source = source;
    return a;
}
inline An lfo(const std::string& functor, const std::string& source, int outputs = 1) { return envelope(functor, source, outputs); }
inline An map(const std::string& functor, const std::string& source, int inputs, int outputs) {  // prelude32.rs:332
    An a("Map<" + functor + "," + std::to_string(inputs) + "," + std::to_string(outputs) + ">", inputs, outputs);
    a.source = source;
    return a;
}
inline An shape_fn(const std::string& functor, const std::string& source) {  // prelude32.rs:1181
    An a("ShaperFn<" + functor + ">", 1, 1);
    a.source = source;
    return a;
}

// feedback.rs: feedback / feedback2 (FrameId), fdn / fdn2 (FrameHadamard)
inline An feedback_with(const char* op, const An& x, const An* y) {
    if (x.inputs != x.outputs || (y && (y->inputs != x.outputs || y->outputs != x.outputs)))
        detail::arity("feedback: the enclosed nodes need as many outputs as inputs");
    An a;
    if (y) {
        a = An("Feedback2<" + x.type + "," + y->type + "," + op + ">", x.inputs, x.outputs);
        detail::adopt(a, x, 0);
        detail::adopt(a, *y, 1);
    } else {
        a = An("Feedback<" + x.type + "," + op + ">", x.inputs, x.outputs);
        detail::adopt(a, x, 0);
    }
    return a;
}
inline An feedback(const An& x) { return feedback_with("FbId", x, nullptr); }
inline An feedback2(const An& x, const An& y) { return feedback_with("FbId", x, &y); }
inline An fdn(const An& x) { return feedback_with("FbHadamard", x, nullptr); }
inline An fdn2(const An& x, const An& y) { return feedback_with("FbHadamard", x, &y); }

// N-fold closure forms (busi / stacki / branchi / sumi / pipei and the ..f variants, prelude.rs:1342-1640)
inline An multi(const char* tmpl, int n, const std::function<An(int)>& f, bool nin_mul, bool nout_mul, const char* extra = "") {
    if (n < 1) detail::arity("N-fold combinator needs N > 0");
    std::vector<An> nodes;
    for (int i = 0; i < n; i++) nodes.push_back(f(i));
    for (const An& x : nodes)
        if (x.type != nodes[0].type) detail::arity("the N nodes of an N-fold combinator have one type");
    An a(std::string(tmpl) + "<" + std::to_string(n) + "," + nodes[0].type + extra + ">", nodes[0].inputs * (nin_mul ? n : 1),
         nodes[0].outputs * (nout_mul ? n : 1));
    for (int i = 0; i < n; i++) detail::adopt(a, nodes[i], i);
    return a;
}
inline An busi(int n, const std::function<An(int)>& f) { return multi("MultiBus", n, f, false, false); }
inline An stacki(int n, const std::function<An(int)>& f) { return multi("MultiStack", n, f, true, true); }
inline An branchi(int n, const std::function<An(int)>& f) { return multi("MultiBranch", n, f, false, true); }
inline An sumi(int n, const std::function<An(int)>& f) { return multi("Reduce", n, f, true, false, ",OpAdd"); }
inline An pipei(int n, const std::function<An(int)>& f) { return multi("PipeN", n, f, false, false); }
inline float frac(int n, int i) { return n > 1 ? (float)((double)i / (double)(n - 1)) : 0.5f; }
inline An busf(int n, const std::function<An(float)>& f) { return busi(n, [&](int i) { return f(frac(n, i)); }); }
inline An stackf(int n, const std::function<An(float)>& f) { return stacki(n, [&](int i) { return f(frac(n, i)); }); }
inline An branchf(int n, const std::function<An(float)>& f) { return branchi(n, [&](int i) { return f(frac(n, i)); }); }
inline An sumf(int n, const std::function<An(float)>& f) { return sumi(n, [&](int i) { return f(frac(n, i)); }); }
inline An pipef(int n, const std::function<An(float)>& f) { return pipei(n, [&](int i) { return f(frac(n, i)); }); }

// ---- more of the prelude (same constructions as fundsp_amd/graph.py) ------------------------------------------------
inline An add(std::initializer_list<P> v) { return multipass((int)v.size()) + constant(v); }   // prelude32.rs:391
inline An sub(std::initializer_list<P> v) { return multipass((int)v.size()) - constant(v); }   // prelude32.rs:409
inline An mul(std::initializer_list<P> v) { return multipass((int)v.size()) * constant(v); }   // prelude32.rs:427
inline An add(P v) { return add({std::move(v)}); }
inline An sub(P v) { return sub({std::move(v)}); }
inline An mul(P v) { return mul({std::move(v)}); }
inline An mixer(const std::vector<std::vector<P>>& matrix) {  // Mixer::new pan.rs:108, matrix[out][in]
    const int n_out = (int)matrix.size(), n_in = (int)matrix[0].size();
    An a("Mixer<" + std::to_string(n_in) + "," + std::to_string(n_out) + ">", n_in, n_out);
    for (int i = 0; i < n_out; i++)
        for (int j = 0; j < n_in; j++) a.with("matrix[" + std::to_string(i * n_in + j) + "]", matrix[i][j]);
    return a;
}
inline An rotate(float angle, float gain) {  // prelude32.rs:2432, libm cos / sin as the engine restates them
    const float c = fdsp_libm_cosf(angle), s = fdsp_libm_sinf(angle);
    return mixer({{c * gain, -s * gain}, {s * gain, c * gain}});
}
enum MeterMode { METER_SAMPLE = 0, METER_PEAK = 1, METER_RMS = 2 };  // dynamics.rs:316-320
inline An meter_node(MeterMode mode, double timescale, bool monitor) {
    An a("MeterT<" + std::to_string((int)mode) + "," + (monitor ? "true" : "false") + ">", 1, 1);
    uint64_t bits;
    std::memcpy(&bits, &timescale, 8);
    Param q{{}, "timescale", {}, true, {bits}};  // the f64 timescale travels as its bit pattern
    a.params.push_back(q);
    return a;
}
inline An meter(MeterMode mode, double timescale = 0.1) { return meter_node(mode, timescale, false); }    // prelude32.rs:300
inline An monitor(MeterMode mode, double timescale = 0.1) { return meter_node(mode, timescale, true); }   // level: ":state" slot
inline An hold(P variability) { return detail::leaf("Hold", 2, 1, 1).with("variability", variability); }  // draws: Bank::set_ring
inline An hold_hz(P f, P variability) { return (pass() | dc(f)) >> hold(variability); }                   // prelude32.rs:831
inline An envelope_in(const std::string& functor, const std::string& source, int inputs, int outputs = 1) {  // prelude32.rs:625-745
    An a("EnvelopeIn<" + functor + ">", inputs, outputs);
    a.source = source;
    return a;
}
inline An flanger(P feedback_amount, P minimum_delay, P maximum_delay, const An& delay_lfo) {  // prelude.rs:2719-2730
    return pass() & feedback2((pass() | delay_lfo) >> tap(minimum_delay, maximum_delay), shape(TANH, feedback_amount));
}
inline An reverb4_stereo_delays(const std::vector<float>& delays, double time) {  // prelude.rs:1917-1941
    if (delays.size() != 32) detail::arity("reverb4_stereo_delays takes 32 delay times");
    const float a = (float)std::pow(std::exp(-60.0 / 20.0 * 2.302585092994046), 0.03 * 10.0 / 10.0 / time);
    auto line = [&](int first) {
        return stacki(16, [&](int i) { return delay(delays[(size_t)(first + i)]) >> fir({-a / 4.0f, -a / 2.0f, -a / 4.0f}); });
    };
    auto smooth9 = [](float x) {
        const float x2 = x * x;
        return ((((70.0f * x - 315.0f) * x + 540.0f) * x - 420.0f) * x + 126.0f) * x2 * x2 * x;
    };
    An pans = sumf(16, [&](float x) { return pan(-1.0f * (1.0f - smooth9(x)) + 1.0f * smooth9(x)); });
    return multisplit(2, 8) >> fdn(line(0)) >> multijoin(2, 8) >> multisplit(2, 8) >> fdn(line(16)) >> pans * dc({0.25f, 0.25f});
}
inline An reverb4_stereo(double room_size, double time) {  // prelude.rs:1873-1914
    static const float d[32] = {0.059326634f, 0.04778291f, 0.06995449f, 0.0393001f, 0.041604012f, 0.06215825f, 0.052269846f,
                                0.043227978f, 0.06966107f, 0.031615064f, 0.068442f, 0.037332155f, 0.032944717f, 0.034493037f,
                                0.06787566f, 0.038824916f, 0.068260126f, 0.068044715f, 0.0688076f, 0.066724524f, 0.051293883f,
                                0.06023173f, 0.040897705f, 0.031507637f, 0.060309593f, 0.049584292f, 0.04532072f, 0.056379095f,
                                0.035180368f, 0.041291796f, 0.046129026f, 0.05504605f};
    const float scale = std::max((float)room_size, 15.0f) / 10.0f;
    std::vector<float> delays(32);
    for (int i = 0; i < 32; i++) delays[(size_t)i] = d[i] * scale;
    return reverb4_stereo_delays(delays, time);
}
inline An reverb_stereo(double room_size, double time, double damping) {  // prelude.rs:1732-1763 (the graph; Bank::reverb_stereo is its dedicated kernel)
    static const double d[32] = {0.073904, 0.052918, 0.066238, 0.066387, 0.037783, 0.080073, 0.050961, 0.075900, 0.043646, 0.072095, 0.056194,
                                 0.045961, 0.058934, 0.068016, 0.047529, 0.058156, 0.072972, 0.036084, 0.062715, 0.076377, 0.044339, 0.076725,
                                 0.077884, 0.046126, 0.067741, 0.049800, 0.051709, 0.082923, 0.070121, 0.079315, 0.055039, 0.081859};
    const float a = (float)std::pow(std::exp(-60.0 / 20.0 * 2.302585092994046), 0.03 * room_size / 10.0 / time);  // pow(db_amp(-60.0), ..) as f32 :1746
    const float gain = 1.0f - (float)damping, alpha = (gain + 1.0f) / 2.0f, beta = (1.0f - alpha) / 2.0f;          // fir3(gain).weights() :863-867
    An line = stacki(32, [&](int i) { return delay((float)(d[i] * room_size / 10.0)) >> fir({beta * a, alpha * a, beta * a}); });
    auto smooth9 = [](float x) {
        const float x2 = x * x;
        return ((((70.0f * x - 315.0f) * x + 540.0f) * x - 420.0f) * x + 126.0f) * x2 * x2 * x;
    };
    An pans = sumf(32, [&](float x) { return pan(-1.0f * (1.0f - smooth9(x)) + 1.0f * smooth9(x)); });
    return multisplit(2, 16) >> fdn(line) >> pans * dc({1.0f / 16.0f, 1.0f / 16.0f});
}
// playwave_at(wave, channel, start, end, loop) (prelude32.rs:2234) over sample slot `slot` (fdsp_wave_upload); the
// u32 parameters travel as raw words
inline An playwave_at(int slot, uint32_t channel, uint32_t start_point, uint32_t end_point, int64_t loop_point = -1) {
    An a("WavePlayer<" + std::to_string(slot) + ">", 0, 1);
    auto word = [](uint32_t u) {
        float f;
        std::memcpy(&f, &u, 4);
        return std::vector<float>{f};
    };
    a.params.push_back(Param{{}, "channel", word(channel), false, {}});
    a.params.push_back(Param{{}, "start_point", word(start_point), false, {}});
    a.params.push_back(Param{{}, "end_point", word(end_point), false, {}});
    a.params.push_back(Param{{}, "loop_point", word(loop_point < 0 ? 0xFFFFFFFFu : (uint32_t)loop_point), false, {}});
    return a;
}
inline An playwave(int slot, uint32_t channel, uint32_t length, int64_t loop_point = -1) { return playwave_at(slot, channel, 0, length, loop_point); }

// ---- Bank: V voices of one graph behind the AudioNode surface ---------------------------------------------------
class Bank {
 public:
    Bank() = default;
    // a kind compiled into the library ("fm_svf", "fixed_svf", ...) or registered by fdsp_graph_compile
    Bank(const std::string& kind, size_t voices, size_t ring_frames = 0) : kind_(kind), ring_frames_(ring_frames) {
        check(ring_frames ? fdsp_bank_create_ring(kind.c_str(), voices, ring_frames, &h_) : fdsp_bank_create(kind.c_str(), voices, &h_));
    }
    // compile the graph (one fused kernel set, cached by type + source), create the bank, apply the graph's parameters
    // `flush_denormals`: for the FRONT half of a Chain whose other half has a Feedback node (reverb_stereo, reverb4_stereo, an fdn network) --
    // Feedback::new switches the constructing thread to FTZ + DAZ (feedback.rs:96, denormal.rs:18), so the reference renders the whole graph
    // flushed; a front compiled on its own has no Feedback in its type, and is handed to the run-time compiler (which flushes kinds whose type
    // expression names one) under an alias that does.  Same type, same kernels, same slot names.
    static Bank from_graph(const An& g, size_t voices, size_t ring_frames = 0, double sample_rate = DEFAULT_SR, bool flush_denormals = false) {
        if (g.rings > 0 && ring_frames == 0) throw Error(FDSP_EINVAL, "this graph has delay lines: pass ring_frames");
        std::string ctype = g.type, csrc = g.source;
        if (flush_denormals && ctype.find("Feedback") == std::string::npos) {
            const std::string alias = "FeedbackThreadFlushed_" + std::to_string(std::hash<std::string>{}(ctype + '\0' + csrc));
            csrc += (csrc.empty() ? "" : "\n") + ("using " + alias + " = " + ctype + ";");
            ctype = alias;
        }
        const std::string name = "cpp_" + std::to_string(std::hash<std::string>{}(ctype + '\0' + csrc));
        if (fdsp_kind_by_name(name.c_str()) < 0) check(fdsp_graph_compile_src(name.c_str(), ctype.c_str(), csrc.c_str()));
        // the shared wavetables of the reference (saw_table() etc., wavetable.rs:493-623) are built on first use
        static const std::pair<const char*, int> tables[] = {{"WaveSynth<0", 0}, {"PulseWave", 0}, {"WaveSynth<1", 1}, {"WaveSynth<2", 2},
                                                             {"WaveSynth<4", 4}, {"WaveSynth<5", 5}, {"WaveSynth<6", 6}};
        for (const auto& t : tables)
            if (g.type.find(t.first) != std::string::npos) {
                int n = 0;
                check(fdsp_wavetable_get(t.second, &n, nullptr, nullptr, nullptr, 0));
                if (n == 0) check(fdsp_wavetable_build(t.second));
            }
        Bank b(name, voices, ring_frames);
        for (const Param& p : g.params) {
            const std::string slot = p.slot();
            if (p.is_u64) {
                std::vector<uint64_t> u(voices, p.u64s.size() == 1 ? p.u64s[0] : 0);
                if (p.u64s.size() == voices) u = p.u64s;
                check(fdsp_bank_set_param_u64(b.h_, slot.c_str(), u.data(), 0, voices));
            } else if (p.values.size() == 1) {
                std::vector<float> all(voices, p.values[0]);  // copied as words: raw u32 parameters keep every bit
                check(fdsp_bank_set_param(b.h_, slot.c_str(), all.data(), 0, voices));
            } else if (p.values.size() == voices) {
                check(fdsp_bank_set_param(b.h_, slot.c_str(), p.values.data(), 0, voices));
            } else {
                throw Error(FDSP_EINVAL, slot + ": a per-voice parameter needs one value per voice");
            }
        }
        b.set_sample_rate(sample_rate);
        b.reset();  // `.phase()` / `.seed()` store only, the constructor's reset applies them (combinator.rs:263-267)
        return b;
    }
    // reverb_stereo(room_size, time, damping) (prelude.rs:1732-1762): `instances` 32-line FDN reverbs, stereo in / out,
    // rendered by the dedicated lane-per-frame kernel
    static Bank reverb_stereo(size_t instances, double room_size, double time, double damping) {
        Bank b;
        check(fdsp_reverb_stereo_create(instances, room_size, time, damping, &b.h_));
        b.kind_ = "reverb_stereo";
        return b;
    }
    // reverb4_stereo(room_size, time) (prelude.rs:1873-1941): two 16-line FDNs in series, the same kernel family
    static Bank reverb4_stereo(size_t instances, double room_size, double time) {
        Bank b;
        check(fdsp_reverb4_stereo_create(instances, room_size, time, &b.h_));
        b.kind_ = "reverb4_stereo";
        return b;
    }
    // reverb3_stereo(time, diffusion, lowpole_hz(cutoff)) (prelude.rs:1858-1871, reverb.rs:152-279): the allpass-loop reverb through its lane-per-frame kernel
    static Bank reverb3_stereo(size_t instances, double time, double diffusion, float lowpole_cutoff_hz) {
        Bank b;
        check(fdsp_reverb3_stereo_create(instances, time, diffusion, lowpole_cutoff_hz, &b.h_));
        b.kind_ = "reverb3_stereo";
        return b;
    }
    // ... with a FixedSvf as the loop filter, e.g. svf_mode = FDSP_SVF_HIGHSHELF for highshelf_hz(5000.0, 1.0, db_amp(-1.0)) (examples/keys.rs:134)
    static Bank reverb3_stereo_svf(size_t instances, double time, double diffusion, int svf_mode, float cutoff_hz, float q, float gain = 1.0f) {
        Bank b;
        check(fdsp_reverb3_stereo_svf_create(instances, time, diffusion, svf_mode, cutoff_hz, q, gain, &b.h_));
        b.kind_ = "reverb3_stereo";
        return b;
    }
    // split / multisplit >> fdn::<N, _>(stacki(|i| delay(delays[i]) >> fir(weights))) >> join / multijoin (prelude.rs:1323-1345, the
    // documented "Mono Reverb" :1334 with inputs = outputs = 1): the generic Hadamard network through the same lane-per-frame kernel family;
    // delays.size() = N in 2, 4, 8, 16, 32, one to three FIR weights, every delay longer than 128 samples at the bank's sample rate
    static Bank fdn(size_t instances, const std::vector<double>& delays, const std::vector<float>& weights, int inputs = 1, int outputs = 1) {
        Bank b;
        check(fdsp_fdn_create(instances, (int)delays.size(), delays.data(), (int)weights.size(), weights.data(), inputs, outputs, &b.h_));
        b.kind_ = "fdn";
        return b;
    }
    Bank(Bank&& o) noexcept { *this = std::move(o); }
    Bank& operator=(Bank&& o) noexcept {
        if (this != &o) {
            close();
            h_ = o.h_;
            kind_ = std::move(o.kind_);
            ring_frames_ = o.ring_frames_;
            o.h_ = nullptr;
        }
        return *this;
    }
    Bank(const Bank&) = delete;
    Bank& operator=(const Bank&) = delete;
    // `Clone` (audionode.rs:35): a new bank that continues exactly where this one stands (fdsp_bank_clone: slots, delay
    // rings, sample rate, arithmetic mode, launch options, scheduler events, reverb line state)
    Bank clone() const {
        Bank b;
        check(fdsp_bank_clone(h_, &b.h_));
        b.kind_ = kind_;
        b.ring_frames_ = ring_frames_;
        return b;
    }
    // A gain and a dry bus around a reverb / network bank, folded into its render kernel (fdsp_bank_set_bus): FDSP_BUS_WET = `wet * node`,
    // FDSP_BUS_DRY_WET = `dry * multipass() & wet * node` -- README.md:436 `multipass() & 0.2 * reverb_stereo(20.0, 2.0, 1.0)` is
    // set_bus(FDSP_BUS_DRY_WET, 0.2f) --, FDSP_BUS_NONE = the node alone.  Unop<X, FrameMulScalar> + Bus + MultiPass: audionode.rs:1190-1228, 1842-1877, 373-403
    void set_bus(int mode, float wet = 1.0f, float dry = 1.0f) { check(fdsp_bank_set_bus(h_, mode, wet, dry)); }
    // launch options of this bank ("pipe_split", "time_split", "fdn_kernel", "timing", "math"; -1 = process-wide default)
    void set_option(const std::string& name, int value) { check(fdsp_bank_set_option(h_, name.c_str(), value)); }
    int get_option(const std::string& name) const {
        const int v = fdsp_bank_get_option(h_, name.c_str());
        if (v < 0) check(v);
        return v;
    }
    ~Bank() { close(); }
    void close() {
        if (h_) fdsp_bank_destroy(h_);
        h_ = nullptr;
    }
    fdsp_bank* handle() const { return h_; }

    // AudioNode surface (audionode.rs:29-369); channels are per voice
    int inputs() const { return fdsp_bank_inputs(h_); }
    int outputs() const { return fdsp_bank_outputs(h_); }
    size_t voices() const { return fdsp_bank_voices(h_); }
    void reset() { check(fdsp_bank_reset(h_)); }
    void set_sample_rate(double sr) { check(fdsp_bank_set_sample_rate(h_, sr)); }
    void set_seed(uint64_t seed) {  // AudioNode::set_seed :366-368, the same seed for every voice
        std::vector<uint64_t> s(voices(), seed);
        check(fdsp_bank_set_seed(h_, s.data(), 0, s.size()));
    }
    void set_seed(const std::vector<uint64_t>& per_voice) { check(fdsp_bank_set_seed(h_, per_voice.data(), 0, per_voice.size())); }
    // AudioNode::set(Setting): slots are addressed "<path>:<field>"
    void set(const std::string& slot, float value) { check(fdsp_bank_set_param_all(h_, slot.c_str(), value)); }
    void set(const std::string& slot, const std::vector<float>& per_voice, size_t first = 0) {
        check(fdsp_bank_set_param(h_, slot.c_str(), per_voice.data(), first, per_voice.size()));
    }
    // state a Rust caller supplies because it comes from a crate outside the reference tree (Pluck's excitation, Hold's
    // Rnd draws): data[count][frames] into ring node `ring_index` for voices first .. first + count
    void set_ring(int ring_index, const float* data, size_t frames, size_t first, size_t count) {
        check(fdsp_bank_set_ring(h_, ring_index, data, frames, first, count));
    }
    // AudioNode::tick for every voice: input [V][inputs], output [V][outputs]
    void tick(const float* input, float* output) {
        check(fdsp_bank_process_host(h_, 1, input, output, FDSP_LAYOUT_PLANAR, 1, FDSP_MODE_TICK));
    }
    // AudioNode::process(size, &BufferRef, &mut BufferMut): planar blocks [V * channels][64] f32, size <= 64
    void process(size_t size, const float* input, float* output) {
        if (size > MAX_BUFFER_SIZE) throw Error(FDSP_EINVAL, "process: size exceeds MAX_BUFFER_SIZE");
        check(fdsp_bank_process_host(h_, size, input, output, FDSP_LAYOUT_PLANAR, MAX_BUFFER_SIZE, FDSP_MODE_PROCESS));
    }
    // device-resident rendering of any length (the engine applies Wave::render's 64-sample blocking itself)
    void process_device(size_t frames, const float* d_in, float* d_out, int layout = FDSP_LAYOUT_VOICE_MINOR, size_t frame_stride = 0,
                        int mode = FDSP_MODE_PROCESS, void* stream = nullptr) {
        check(fdsp_bank_process(h_, frames, d_in, d_out, layout, frame_stride, mode, stream));
    }
    // render + mix-down in one launch (fdsp_bank_process_mix): `voice >> pan(p)` per voice and the sum over the voices
    // (pan.rs:50-76, Reduce audionode.rs:2406-2462) without the per-voice output ever existing in HBM.  d_mix: [2][frames] for
    // FDSP_MIX_PAN (mono graphs, positions from set_pan, default centre), [outputs][frames] for FDSP_MIX_SUM.  d_in voice-minor.
    void process_mix(size_t frames, const float* d_in, float* d_mix, int mix = FDSP_MIX_SUM, int mode = FDSP_MODE_PROCESS, void* stream = nullptr) {
        check(fdsp_bank_process_mix(h_, frames, d_in, d_mix, mix, mode, stream));
    }
    void set_pan(const std::vector<float>& per_voice, size_t first = 0) { check(fdsp_bank_set_pan(h_, per_voice.data(), first, per_voice.size())); }
    void mix_reserve(size_t frames) { check(fdsp_bank_mix_reserve(h_, frames)); }  // AudioNode::allocate for the mix path
    void synchronize() { check(fdsp_bank_synchronize(h_)); }
    // Sequencer with one event per voice (sequencer.rs:355-398, 838-951): push = set_events (rows of start, end, fade_in,
    // fade_out in seconds; `fade` FDSP_FADE_POWER / FDSP_FADE_SMOOTH per voice or nullptr), process = process_events
    void set_events(const std::vector<double>& events_x4, const int* fade = nullptr, size_t first = 0) {
        check(fdsp_bank_set_events(h_, events_x4.data(), fade, first, events_x4.size() / 4));
    }
    void process_events(size_t frames, const float* d_in, float* d_out, int mode = FDSP_MODE_PROCESS, void* stream = nullptr) {
        check(fdsp_bank_process_events(h_, frames, d_in, d_out, mode, stream));
    }
    // ... the Sequencer's mixed output in the same launch: d_mix [outputs][frames]
    void process_events_mix(size_t frames, const float* d_in, float* d_mix, int mode = FDSP_MODE_PROCESS, void* stream = nullptr) {
        check(fdsp_bank_process_events_mix(h_, frames, d_in, d_mix, mode, stream));
    }
    void events_rewind(double time) { check(fdsp_bank_events_rewind(h_, time)); }
    double events_time() const { return fdsp_bank_events_time(h_); }

 private:
    fdsp_bank* h_ = nullptr;
    std::string kind_;
    size_t ring_frames_ = 0;
};

// ---- Chain: `source >> effect` as two banks piped by the host -----------------------------------------------------
// AttoHash::hash (math.rs:649-658)
inline uint64_t atto(uint64_t state, uint64_t data) { return (((state << 5) | (state >> 59)) ^ data) * 0x517cc1b727220a95ULL; }

// `G::ping(true, AttoHash::new(G::ID))` of a graph's TYPE -- the hash a combinator's constructor hands down to its nodes (audionode.rs:871-876,
// 1389-1394) -- evaluated on the device by a one-slot probe kind whose constructor stores it (compiled once per type).
inline uint64_t probe_hash(const An& g) {
    static const char* probe =
        "template <class G> struct HashProbe { static constexpr int IN = 0, OUT = 1, RINGS = 0; static constexpr uint64_t ID = 0; uint64_t h;\n"
        "  template <class V> FD_HD void visit(V& v) { v.u64(h, STATE, \"probe\"); } FD_HD void bind(Ctx&) {}\n"
        "  FD_HD void init() { G g; h = g.ping(true, G::ID); } FD_HD void update(double) {} FD_HD void reset() {}\n"
        "  FD_HD uint64_t ping(bool, uint64_t x) { return x; } FD_HD void begin_block(int) {} FD_HD bool tripped() const { return false; }\n"
        "  FD_HD void end_simd() {} template <int PH> FD_HD void step(const float*, float* out) { out[0] = 0.0f; } FD_STEP2_VIA_STEP };\n";
    const std::string name = "cpp_probe_" + std::to_string(std::hash<std::string>{}(g.type + '\0' + g.source));
    if (fdsp_kind_by_name(name.c_str()) < 0) {
        const std::string src = g.source + "\n" + probe, type = "HashProbe<" + g.type + ">";
        check(fdsp_graph_compile_src(name.c_str(), type.c_str(), src.c_str()));
    }
    Bank b(name, 1);
    float words[2] = {0.0f, 0.0f};
    check(fdsp_bank_get_state(b.handle(), words));
    uint32_t lo, hi;
    std::memcpy(&lo, &words[0], 4);
    std::memcpy(&hi, &words[1], 4);
    return ((uint64_t)hi << 32) | lo;
}

// Two banks in series: the first bank's output blocks are the second's input blocks.  What it is for: a graph whose halves want different kernel
// families -- `(noise() | noise()) >> reverb_stereo(10.0, 1.0, 0.5)`, the reference's own `reverb` bench (benches/benchmark.rs:79-85), compiled as ONE
// lane-per-voice graph reads its 32 delay lines one lane per instance; as Chain(Bank::from_graph(noise() | noise(), V), Bank::reverb_stereo(V, ..)) the
// network runs in its lane-per-frame kernel (two to three orders of magnitude faster).  With `whole` = the graph the chain stands for, the source starts
// from the hash the Pipe's constructor would hand it (Pipe::ping, audionode.rs:1459: its left side sees hash.hash(Pipe::ID) of what the probe ping of
// the whole type returned) and the chain renders what the one graph renders; without, both halves keep the construction hash of a stand-alone node --
// two AudioNodes piped by hand.  set_seed = AudioNode::set_seed of the Pipe; the effect's own ping is skipped (exact for the stock reverbs and
// networks: none of their nodes keeps hashed state).  (Python: fundsp_amd.Chain / Bank.from_graph; Rust: INTEGRATION.md section 9.)
class Chain {
 public:
    static constexpr uint64_t PIPE_ID = 6;  // audionode.rs:1426
    Chain(Bank&& source, Bank&& effect, const An* whole = nullptr) : source_(std::move(source)), effect_(std::move(effect)) {
        if (source_.voices() != effect_.voices() || source_.outputs() != effect_.inputs())
            throw Error(FDSP_EINVAL, "Chain: the source's outputs are not the effect's inputs");
        if (whole) {
            has_ctor_ = true;
            ctor_ = probe_hash(*whole);
            source_.set_seed(atto(ctor_, PIPE_ID));
            source_.reset();
        }
    }
    Bank& source() { return source_; }
    Bank& effect() { return effect_; }
    int inputs() const { return source_.inputs(); }
    int outputs() const { return effect_.outputs(); }
    size_t voices() const { return source_.voices(); }
    void reset() { source_.reset(); effect_.reset(); }
    void set_sample_rate(double sr) { source_.set_sample_rate(sr); effect_.set_sample_rate(sr); }
    void set_seed() {  // re-apply the construction hash (a chain built with `whole`; otherwise the stand-alone source's own)
        if (has_ctor_) source_.set_seed(atto(ctor_, PIPE_ID));
    }
    void set_seed(uint64_t seed) { source_.set_seed(atto(seed, PIPE_ID)); }
    void set_seed(std::vector<uint64_t> per_voice) {
        for (uint64_t& s : per_voice) s = atto(s, PIPE_ID);
        source_.set_seed(per_voice);
    }
    // AudioNode::process(size, input, output): planar blocks [V * channels][64] f32 through both halves
    void process(size_t size, const float* input, float* output) {
        mid_.resize(voices() * (size_t)source_.outputs() * MAX_BUFFER_SIZE);
        source_.process(size, input, mid_.data());
        effect_.process(size, mid_.data(), output);
    }
    // device-resident rendering: d_mid = a device buffer of the source's output shape in the launch's layout; both launches go to `stream`, or --
    // stream == nullptr means "the bank's own stream" and two banks' own streams do not order each other -- the host waits for the source first
    void process_device(size_t frames, const float* d_in, float* d_mid, float* d_out, int layout = FDSP_LAYOUT_VOICE_MINOR, size_t frame_stride = 0,
                        int mode = FDSP_MODE_PROCESS, void* stream = nullptr) {
        source_.process_device(frames, d_in, d_mid, layout, frame_stride, mode, stream);
        if (!stream) source_.synchronize();
        effect_.process_device(frames, d_mid, d_out, layout, frame_stride, mode, stream);
    }

 private:
    Bank source_, effect_;
    std::vector<float> mid_;
    bool has_ctor_ = false;
    uint64_t ctor_ = 0;
};

// the Arc<Wave> of playwave(): [channels][length] f32 into sample slot 0..7
inline void wave_upload(int slot, int channels, size_t length, const float* data) { check(fdsp_wave_upload(slot, channels, length, data)); }

// Wave::render (wave.rs:441-466): set the sample rate, chop `duration` into <= 64-sample blocks, call process.
// Returns [V * outputs][length] planar.
inline std::vector<float> render(double sample_rate, double duration, Bank& bank) {
    if (bank.inputs() != 0) throw Error(FDSP_EINVAL, "render: the node must be a generator");
    bank.set_sample_rate(sample_rate);
    const size_t length = (size_t)std::llround(duration * sample_rate);
    const size_t rows = bank.voices() * (size_t)bank.outputs();
    std::vector<float> wave(rows * length), block(rows * MAX_BUFFER_SIZE);
    for (size_t i = 0; i < length;) {
        const size_t n = std::min(length - i, MAX_BUFFER_SIZE);
        bank.process(n, nullptr, block.data());
        for (size_t r = 0; r < rows; r++) std::memcpy(&wave[r * length + i], &block[r * MAX_BUFFER_SIZE], n * sizeof(float));
        i += n;
    }
    return wave;
}

}  // namespace fundsp_hip

#endif  // FUNDSP_HIP_HPP

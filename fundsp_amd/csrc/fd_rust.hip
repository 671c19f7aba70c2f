// fd_rust.hip -- the Rust-side front door of the graph -> kernel compiler.
//
// A FunDSP graph is a statically typed combinator tree; what a Rust host can hand over without any extra crate is the
// string `core::any::type_name::<X>()` of its graph `An<X>`, e.g. for `sine_hz(440.0) >> lowpass_hz(1000.0, 1.0)`:
//
//   fundsp::combinator::An<fundsp::audionode::Pipe<fundsp::audionode::Pipe<fundsp::audionode::Constant<typenum::uint::
//   UInt<typenum::uint::UTerm, typenum::bit::B1>>, fundsp::oscillator::Sine<f32>>, fundsp::svf::FixedSvf<f32,
//   fundsp::svf::LowpassMode<f32>>>>
//
// This file parses that spelling (paths, generic arguments, typenum's binary UInt<.., Bn> integers, `{{closure}}`
// segments) and rewrites it with the engine's device templates (fd_nodes.hpp):  Pipe<Pipe<Constant<1>,Sine>,FixedSvf>
// plus the PARAMETERS the Rust type itself carries -- filter modes (LowpassMode ..), shape kinds (Tanh ..), biquad
// modes -- as slot presets ("1:mode=0").  Values that live in Rust fields (frequencies, Q ..) are not in the type; the host
// sets them through fdsp_bank_set_param exactly as for any other kind.  What neither the type nor a field carries in a
// form the engine can read -- which shared wavetable a WaveSynth holds, the Meter mode, closures -- comes in `hints`:
//   "wavesynth=saw,square;meter=peak,rms;envelope=EnvExp;envelope_in=MyFn;map=MidSide;shape_fn=SoftFold"
// (lists are consumed in the order the nodes appear in the type, left to right).
// Reference: combinator.rs:178-488 (An, operators), audionode.rs:850 (Binop), :1232 (Unop), :1375 (Pipe), :1496 (Stack).
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/fundsp_hip.h"

namespace fd {
int api_fail(int code, const std::string& msg);
}

namespace {

struct TypeNode {
    std::string full, last;  // whole path, last path segment
    std::vector<TypeNode> args;
};

struct Parser {
    const std::string& s;
    size_t i = 0;
    std::string err;
    explicit Parser(const std::string& str) : s(str) {}
    void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t')) i++; }
    bool parse(TypeNode* out) {
        ws();
        if (i < s.size() && s[i] == '&') {  // &T / &mut T: transparent
            i++;
            ws();
            if (s.compare(i, 4, "mut ") == 0) i += 4;
            return parse(out);
        }
        if (i < s.size() && s[i] == '(') {  // tuple
            i++;
            out->full = out->last = "()";
            ws();
            while (i < s.size() && s[i] != ')') {
                TypeNode a;
                if (!parse(&a)) return false;
                out->args.push_back(a);
                ws();
                if (i < s.size() && s[i] == ',') i++;
                ws();
            }
            if (i >= s.size()) return fail("unterminated tuple");
            i++;
            return true;
        }
        // path: segments of [A-Za-z0-9_] or {{closure}} / {closure#0}, separated by ::
        size_t start = i;
        size_t seg = i;
        for (;;) {
            if (i < s.size() && s[i] == '{') {
                int depth = 0;
                while (i < s.size()) {
                    if (s[i] == '{') depth++;
                    if (s[i] == '}') depth--;
                    i++;
                    if (depth == 0) break;
                }
            } else {
                while (i < s.size() && (isalnum((unsigned char)s[i]) || s[i] == '_')) i++;
            }
            if (i + 1 < s.size() && s[i] == ':' && s[i + 1] == ':') {
                i += 2;
                seg = i;
                continue;
            }
            break;
        }
        if (i == start) return fail("expected a type name");
        out->full = s.substr(start, i - start);
        out->last = s.substr(seg, i - seg);
        ws();
        if (i < s.size() && s[i] == '<') {
            i++;
            for (;;) {
                TypeNode a;
                if (!parse(&a)) return false;
                out->args.push_back(a);
                ws();
                if (i < s.size() && s[i] == ',') {
                    i++;
                    continue;
                }
                if (i < s.size() && s[i] == '>') {
                    i++;
                    break;
                }
                return fail("expected , or > in generic arguments");
            }
        }
        return true;
    }
    bool fail(const std::string& m) {
        err = m + " at offset " + std::to_string(i);
        return false;
    }
};

struct Hints {
    std::map<std::string, std::vector<std::string>> lists;
    std::map<std::string, size_t> used;
    explicit Hints(const char* h) {
        if (!h) return;
        std::string s(h), key, cur;
        std::vector<std::string> vals;
        auto flush_val = [&] {
            if (!cur.empty()) vals.push_back(cur);
            cur.clear();
        };
        auto flush_key = [&] {
            flush_val();
            if (!key.empty()) lists[key] = vals;
            key.clear();
            vals.clear();
        };
        bool in_key = true;
        for (char c : s) {
            if (c == ' ') continue;
            if (in_key) {
                if (c == '=') in_key = false;
                else key += c;
            } else if (c == ',') flush_val();
            else if (c == ';') {
                flush_key();
                in_key = true;
            } else cur += c;
        }
        flush_key();
    }
    bool next(const std::string& key, std::string* out) {
        auto it = lists.find(key);
        size_t& k = used[key];
        if (it == lists.end() || k >= it->second.size()) return false;
        *out = it->second[k++];
        return true;
    }
};

struct Translator {
    Hints hints;
    std::vector<std::pair<std::string, float>> presets;  // slot name -> value
    std::string err;
    explicit Translator(const char* h) : hints(h) {}

    static std::string path_str(const std::vector<int>& p) {
        std::string s;
        for (size_t i = 0; i < p.size(); i++) s += (i ? "." : "") + std::to_string(p[i]);
        return s;
    }
    void preset(const std::vector<int>& p, const std::string& field, float v) { presets.push_back({path_str(p) + ":" + field, v}); }
    bool fail(const std::string& m) {
        if (err.empty()) err = m;
        return false;
    }
    // typenum unsigned: UTerm = 0, UInt<U, Bn> = 2 * U + n; the prelude's aliases U0 .. U128 are accepted too
    bool unum(const TypeNode& n, int* out) {
        if (n.last == "UTerm") {
            *out = 0;
            return true;
        }
        if (n.last == "UInt" && n.args.size() == 2) {
            int hi = 0;
            if (!unum(n.args[0], &hi)) return false;
            if (n.args[1].last != "B0" && n.args[1].last != "B1") return fail("bad typenum bit " + n.args[1].full);
            *out = 2 * hi + (n.args[1].last == "B1");
            return true;
        }
        if (n.last.size() >= 2 && n.last[0] == 'U' && isdigit((unsigned char)n.last[1])) {
            *out = atoi(n.last.c_str() + 1);
            return true;
        }
        return fail("expected a typenum integer, got " + n.full);
    }
    static std::vector<int> sub(const std::vector<int>& p, int i) {
        std::vector<int> q = p;
        q.push_back(i);
        return q;
    }
    static int index_of(const char* const* names, int n, const std::string& s) {
        for (int i = 0; i < n; i++)
            if (s == names[i]) return i;
        return -1;
    }
    bool shape_kind(const TypeNode& n, int* kind) {
        static const char* const names[] = {"Clip", "ClipTo", "Tanh", "Atan", "Softsign", "Crush", "SoftCrush", "Adaptive"};
        *kind = index_of(names, 8, n.last);
        if (*kind < 0) return fail("unknown Shape type " + n.full);
        if (*kind == 7) {  // Adaptive<S>: 7 for S = Tanh (the reference's own spelling), 8 + S otherwise (fd_nodes.hpp SH_ADAPTIVE)
            if (n.args.size() != 1) return fail(n.full + ": expected Adaptive<S>");
            const int inner = index_of(names, 7, n.args[0].last);
            if (inner < 0) return fail("Adaptive<S>: S must be one of the seven plain shapes, got " + n.args[0].full);
            if (inner != 2) *kind = 8 + inner;
        }
        return true;
    }

    bool tr(const TypeNode& n, const std::vector<int>& p, std::string* o) {
        const std::string& t = n.last;
        const auto& a = n.args;
        auto need = [&](size_t k) { return a.size() == k ? true : fail(n.full + ": expected " + std::to_string(k) + " generic arguments"); };
        int k = 0, m = 0;
        std::string x, y;
        if (t == "An") return need(1) && tr(a[0], p, o);
        if (t == "Pipe" || t == "Stack" || t == "Branch" || t == "Bus") {
            if (!need(2) || !tr(a[0], sub(p, 0), &x) || !tr(a[1], sub(p, 1), &y)) return false;
            *o = t + "<" + x + "," + y + ">";
            return true;
        }
        if (t == "Thru") {
            if (!need(1) || !tr(a[0], sub(p, 0), &x)) return false;
            *o = "Thru<" + x + ">";
            return true;
        }
        if (t == "Binop") {  // Binop<B, X, Y>  audionode.rs:850
            if (!need(3)) return false;
            static const char* const rs[] = {"FrameAdd", "FrameSub", "FrameMul"};
            static const char* const en[] = {"OpAdd", "OpSub", "OpMul"};
            const int b = index_of(rs, 3, a[0].last);
            if (b < 0) return fail("unsupported FrameBinop " + a[0].full);
            if (!tr(a[1], sub(p, 0), &x) || !tr(a[2], sub(p, 1), &y)) return false;
            *o = std::string("Binop<") + en[b] + "," + x + "," + y + ">";
            return true;
        }
        if (t == "Unop") {  // Unop<X, U>  audionode.rs:1232
            if (!need(2)) return false;
            static const char* const rs[] = {"FrameNeg", "FrameAddScalar", "FrameNegAddScalar", "FrameMulScalar"};
            static const char* const en[] = {"UNeg", "UAddScalar", "UNegAddScalar", "UMulScalar"};
            const int u = index_of(rs, 4, a[1].last);
            if (u < 0) return fail("unsupported FrameUnop " + a[1].full);
            if (!tr(a[0], sub(p, 0), &x)) return false;
            *o = "Unop<" + x + "," + en[u] + ">";
            return true;
        }
        // ---- leaves with one typenum arity
        struct Arity1 { const char* rust; const char* engine; size_t nargs; size_t which; };
        static const Arity1 arity1[] = {
            {"Constant", "Constant", 1, 0}, {"MultiPass", "MultiPass", 1, 0}, {"Sink", "Sink", 1, 0}, {"Split", "Split", 1, 0},
            {"Join", "Join", 1, 0}, {"Reverse", "Reverse", 1, 0}, {"Impulse", "Impulse", 1, 0}, {"Tick", "Tick", 1, 0},
            {"Fir", "Fir", 1, 0}, {"Dsf", "Dsf", 1, 0}, {"Limiter", "Limiter", 1, 0},
            {"ButterLowpass", "ButterLowpass", 2, 1}, {"Resonator", "Resonator", 2, 1}, {"Moog", "Moog", 2, 1}, {"Rez", "Rez", 2, 1},
        };
        for (const auto& e : arity1)
            if (t == e.rust) {
                if (!need(e.nargs) || !unum(a[e.which], &k)) return false;
                *o = std::string(e.engine) + "<" + std::to_string(k) + ">";
                return true;
            }
        if (t == "MultiSplit" || t == "MultiJoin") {
            if (!need(2) || !unum(a[0], &m) || !unum(a[1], &k)) return false;
            *o = t + "<" + std::to_string(m) + "," + std::to_string(k) + ">";
            return true;
        }
        if (t == "Mixer") {  // Mixer<M, N>: M inputs, N outputs
            if (!need(2) || !unum(a[0], &m) || !unum(a[1], &k)) return false;
            *o = "Mixer<" + std::to_string(m) + "," + std::to_string(k) + ">";
            return true;
        }
        // ---- plain leaves
        struct Plain { const char* rust; const char* engine; };
        static const Plain plain[] = {
            {"Pass", "Pass"}, {"Sine", "Sine"}, {"Noise", "Noise"}, {"Mls", "Mls"}, {"Morph", "Morph"}, {"Biquad", "Biquad"},
            {"BiquadBank", "BiquadBank"}, {"Pinkpass", "Pinkpass"}, {"Follow", "Follow"}, {"AFollow", "AFollow"}, {"Delay", "Delay"},
            {"PulseWave", "PulseWave"}, {"Ramp", "PhaseOsc<OSC_RAMP>"}, {"PolySaw", "PhaseOsc<OSC_POLYSAW>"},
            {"PolySquare", "PhaseOsc<OSC_POLYSQUARE>"}, {"PolyPulse", "PhaseOsc<OSC_POLYPULSE>"}, {"Rossler", "Chaos<false>"},
            {"Lorenz", "Chaos<true>"}, {"Pluck", "Pluck"}, {"Hold", "Hold"}, {"Declick", "Declick"}, {"Var", "Var"},
        };
        for (const auto& e : plain)
            if (t == e.rust) {
                *o = e.engine;
                return true;
            }
        if (t == "DCBlock") {
            *o = "OnePole<OP_DCBLOCK,1>";
            return true;
        }
        if (t == "Lowpole" || t == "Highpole" || t == "Allpole") {
            if (!need(2) || !unum(a[1], &k)) return false;
            *o = std::string("OnePole<") + (t == "Lowpole" ? "OP_LOWPOLE" : t == "Highpole" ? "OP_HIGHPOLE" : "OP_ALLPOLE") + "," + std::to_string(k) + ">";
            return true;
        }
        if (t == "FixedSvf" || t == "Svf") {  // the MODE is a type parameter in Rust, a per-voice parameter here
            if (!need(2)) return false;
            static const char* const modes[] = {"LowpassMode", "HighpassMode", "BandpassMode", "NotchMode", "PeakMode", "AllpassMode",
                                                "BellMode", "LowshelfMode", "HighshelfMode"};
            const int mode = index_of(modes, 9, a[1].last);
            if (mode < 0) return fail("unknown SvfMode " + a[1].full);
            preset(p, "mode", (float)mode);
            *o = t == "FixedSvf" ? "FixedSvf" : (mode >= 6 ? "Svf<4>" : "Svf<3>");
            return true;
        }
        if (t == "Tap" || t == "TapLinear") {
            if (!need(1) || !unum(a[0], &k)) return false;
            *o = std::string("TapT<") + (t == "Tap" ? "false" : "true") + (k == 1 ? "" : "," + std::to_string(k)) + ">";
            return true;
        }
        if (t == "AllNest") {  // AllNest<N, X>: N = 1 fixed coefficient, N = 2 coefficient input
            if (!need(2) || !unum(a[0], &k) || !tr(a[1], sub(p, 0), &x)) return false;
            *o = "AllNest<" + x + (k == 2 ? ",2" : "") + ">";
            return true;
        }
        if (t == "Oversampler" || t == "Resample") {
            if (!need(1) || !tr(a[0], sub(p, 0), &x)) return false;
            *o = t + "<" + x + ">";
            return true;
        }
        if (t == "Panner") {
            if (!need(1) || !unum(a[0], &k)) return false;
            *o = k == 1 ? "Panner" : "PannerT<2>";
            return true;
        }
        if (t == "WaveSynth") {  // which Arc<Wavetable> it holds is a field: hint, default saw
            static const char* const sets[] = {"saw", "square", "triangle", "user3", "organ", "soft_saw", "hammond", "user7"};
            std::string h = "saw";
            hints.next("wavesynth", &h);
            const int set = index_of(sets, 8, h);
            if (set < 0) return fail("unknown wavetable set in hints: " + h);
            if (!need(1) || !unum(a[0], &k)) return false;
            if (k != 1) return fail("WaveSynth<U2> only occurs inside PulseWave");
            *o = "WaveSynth<" + std::to_string(set) + ">";
            return true;
        }
        if (t == "MeterNode" || t == "Monitor") {  // Meter::{Sample, Peak(t), Rms(t)} is a field: hint, default peak
            std::string h = "peak";
            hints.next("meter", &h);
            const int mode = h == "sample" ? 0 : h == "peak" ? 1 : h == "rms" ? 2 : -1;
            if (mode < 0) return fail("unknown meter mode in hints: " + h);
            *o = "MeterT<" + std::to_string(mode) + "," + (t == "Monitor" ? "true" : "false") + ">";
            return true;
        }
        if (t == "Shaper") {
            if (!need(1) || !shape_kind(a[0], &k)) return false;
            preset(p, "shape", (float)k);
            *o = "Shaper";
            return true;
        }
        if (t == "ShaperFn" || t == "Map" || t == "Envelope" || t == "EnvelopeIn") {
            // closures: adsr_live's is recognised by its path; the others name the functor that stands in for them
            if (t == "EnvelopeIn" && n.full.find("adsr") == std::string::npos) {
                for (const auto& arg : a)
                    if (arg.full.find("adsr_live") != std::string::npos || arg.full.find("adsr::") != std::string::npos) {
                        *o = "AdsrLive";
                        return true;
                    }
            }
            const char* key = t == "ShaperFn" ? "shape_fn" : t == "Map" ? "map" : t == "Envelope" ? "envelope" : "envelope_in";
            std::string fn;
            if (!hints.next(key, &fn)) return fail(n.last + " holds a Rust closure: name the functor that stands in for it in hints (" + key + "=..)");
            if (t == "Map") {  // Map<M, I, O>
                if (!need(3) || !unum(a[1], &m) || !unum(a[2], &k)) return false;
                *o = "Map<" + fn + "," + std::to_string(m) + "," + std::to_string(k) + ">";
            } else {
                *o = t + "<" + fn + ">";
            }
            return true;
        }
        if (t == "Feedback" || t == "Feedback2") {  // Feedback<N, X, U> / Feedback2<N, X, Y, U>
            const bool two = t == "Feedback2";
            if (!need(two ? 4 : 3)) return false;
            const std::string& u = a[two ? 3 : 2].last;
            const char* fb = u == "FrameId" ? "FbId" : u == "FrameHadamard" ? "FbHadamard" : nullptr;
            if (!fb) return fail("unsupported feedback FrameUnop " + a[two ? 3 : 2].full);
            if (!tr(a[1], sub(p, 0), &x)) return false;
            if (two && !tr(a[2], sub(p, 1), &y)) return false;
            *o = t + "<" + x + (two ? "," + y : "") + "," + fb + ">";
            return true;
        }
        if (t == "MultiBus" || t == "MultiStack" || t == "MultiBranch" || t == "Chain" || t == "Reduce") {
            // N copies of one node type: presets of the element type apply to every copy (paths i.*)
            if (!need(t == "Reduce" ? 3 : 2) || !unum(a[0], &k)) return false;
            std::string op;
            if (t == "Reduce") {
                const std::string& b = a[2].last;
                op = b == "FrameAdd" ? ",OpAdd" : b == "FrameSub" ? ",OpSub" : b == "FrameMul" ? ",OpMul" : "";
                if (op.empty()) return fail("unsupported FrameBinop " + a[2].full);
            }
            for (int i = 0; i < k; i++) {
                Hints saved = hints;  // every copy consumes the same hints
                if (!tr(a[1], sub(p, i), &x)) return false;
                if (i + 1 < k) hints = saved;
            }
            *o = (t == "Chain" ? std::string("PipeN") : t) + "<" + std::to_string(k) + "," + x + op + ">";
            return true;
        }
        // nonlinear biquads: {Fixed}FbBiquad<f32, M, S> / {Fixed}DirtyBiquad<f32, M, S>  biquad.rs:494-920
        if (t == "FbBiquad" || t == "FixedFbBiquad" || t == "DirtyBiquad" || t == "FixedDirtyBiquad") {
            if (!need(3)) return false;
            static const char* const bm[] = {"ButterBiquad", "ResonatorBiquad", "LowpassBiquad", "HighpassBiquad", "BellBiquad"};
            const int mode = index_of(bm, 5, a[1].last);
            if (mode < 1) return fail("unsupported BiquadMode " + a[1].full);
            if (!shape_kind(a[2], &k)) return false;
            const bool dirty = t.find("Dirty") != std::string::npos, fixed = t.find("Fixed") == 0;
            preset(p, "mode", (float)mode);
            preset(sub(p, 0), "shape", (float)k);
            if (dirty) preset(sub(p, 1), "shape", (float)k);
            *o = std::string("NlBiquad<") + (dirty ? "true" : "false") + "," + (fixed ? "1" : (mode == 4 ? "4" : "3")) + ">";
            return true;
        }
        return fail("no device template for the Rust type " + n.full);
    }
};

// prelude64 graphs (`F = f64`: Sine<f64>, FixedSvf<f64, LowpassMode<f64>>, Moog<f64, U1>, Envelope<f64, ..>,
// BiquadBank<wide::f64x4_::f64x4> ...; reference src/prelude64.rs:338, :1924, :2711, src/biquad_bank.rs:14-24,
// src/lib.rs:521) keep their recurrences and phases in f64 -- and BiquadBank<f64x4> has FOUR lanes, not eight.  The
// engine's nodes are the prelude32 ones (f32 state), so such a graph must be refused, not rendered
// with f32 state: no prelude32 type carries an f64 parameter anywhere, so any f64 / f64xN argument means prelude64.
const TypeNode* find_f64(const TypeNode& n) {
    if (n.last == "f64" || n.last.compare(0, 4, "f64x") == 0) return &n;
    for (const auto& a : n.args)
        if (const TypeNode* hit = find_f64(a)) return hit;
    return nullptr;
}
const TypeNode* find_f64_owner(const TypeNode& n) {  // the innermost node that has the f64 as a direct argument
    for (const auto& a : n.args) {
        if (a.last == "f64" || a.last.compare(0, 4, "f64x") == 0) return &n;
        if (const TypeNode* hit = find_f64_owner(a)) return hit;
    }
    return nullptr;
}

int translate(const char* rust_type_name, const char* hints, std::string* expr, std::string* presets) {
    if (!rust_type_name || !*rust_type_name) return fd::api_fail(FDSP_EINVAL, "type name missing");
    const std::string src(rust_type_name);
    Parser ps(src);
    TypeNode root;
    if (!ps.parse(&root)) return fd::api_fail(FDSP_EINVAL, "cannot parse the Rust type name: " + ps.err);
    ps.ws();
    if (ps.i != src.size()) return fd::api_fail(FDSP_EINVAL, "trailing characters after the Rust type name at offset " + std::to_string(ps.i));
    if (const TypeNode* f = find_f64(root)) {
        const TypeNode* owner = find_f64_owner(root);
        return fd::api_fail(FDSP_EINVAL, "F = f64: the engine renders prelude32 (F = f32) graphs only; `" + (owner ? owner->last : root.last) +
                                             "<.." + f->last + "..>` is a prelude64 node (f64 state" +
                                             (f->last != "f64" ? ", " + f->last.substr(4) + " lanes" : "") + ") and would not match the reference if rendered with f32 state");
    }
    Translator tr(hints);
    if (!tr.tr(root, {}, expr)) return fd::api_fail(FDSP_EINVAL, tr.err);
    presets->clear();
    for (const auto& kv : tr.presets) {
        char buf[64];
        snprintf(buf, sizeof buf, "%.9g", kv.second);
        *presets += kv.first + "=" + buf + "\n";
    }
    return FDSP_OK;
}

}  // namespace

namespace fd {
int rust_translate(const char* rust_type_name, const char* hints, std::string* expr, std::string* presets) {
    return translate(rust_type_name, hints, expr, presets);
}
}  // namespace fd

extern "C" int fdsp_rust_type_to_expr(const char* rust_type_name, const char* hints, char* out_expr, size_t expr_cap,
                                      char* out_presets, size_t presets_cap) {
    std::string expr, presets;
    if (int rc = translate(rust_type_name, hints, &expr, &presets)) return rc;
    if (out_expr) {
        if (expr.size() + 1 > expr_cap) return fd::api_fail(FDSP_EINVAL, "out_expr too small: need " + std::to_string(expr.size() + 1));
        std::memcpy(out_expr, expr.c_str(), expr.size() + 1);
    }
    if (out_presets) {
        if (presets.size() + 1 > presets_cap) return fd::api_fail(FDSP_EINVAL, "out_presets too small: need " + std::to_string(presets.size() + 1));
        std::memcpy(out_presets, presets.c_str(), presets.size() + 1);
    }
    return FDSP_OK;
}

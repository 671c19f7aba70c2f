// fd_kinds_graph.hip -- fused voice graphs of the BASELINE configs.  The types spell out exactly what the
// reference's operator overloads build (combinator.rs:289-488; Rust precedence `*` > `+` > `>>`).
#include "fd_engine.hpp"

namespace fd {
// sine_hz(f) = constant(f) >> sine()                       prelude.rs:349
using SineHz = Pipe<Constant<1>, Sine>;
// config 1: sine_hz(440) >> lowpass_hz(1000, 1)
using SineHzLowpass = Pipe<SineHz, FixedSvf>;
// config 2 voice: noise() >> biquad(..) -- arithmetic of one BiquadBank<f32x8> lane fed by white noise
using NoiseBiquad = Pipe<Noise, Biquad>;
// config 3: sine_hz(f) * f * m + f >> sine() >> lowpass_hz(fc, q)     (README.md:98-103)
using FmMod = Unop<Unop<Unop<SineHz, UMulScalar>, UMulScalar>, UAddScalar>;
using FmSvf = Pipe<Pipe<FmMod, Sine>, FixedSvf>;

void register_graph_kinds(std::vector<KindOps>& out) {
    out.push_back(make_kind<SineHz>("sine_hz"));
    out.push_back(make_kind<SineHzLowpass>("sine_hz_lowpass_hz"));
    out.push_back(make_kind<NoiseBiquad>("noise_biquad"));
    out.push_back(make_kind<FmSvf>("fm_svf"));
}
}  // namespace fd

// fd_kinds_graph.hip -- fused voice graphs of the BASELINE configs.  The types spell out exactly what the
// reference's operator overloads build (combinator.rs:289-488; Rust precedence `*` > `+` > `>>`).
// (sine_hz, sine_hz_lowpass_hz and fm_svf -- configs 1 and 3 -- are in fd_kinds_fm.hip.)
#include "fd_kinds_graph.hpp"

namespace fd {
// filter chains with an audio input (exercise the loader wave together with 2 / 3 compute stages):
//   lowpass_hz(fc, q) >> shape(..)   and   lowpass_hz(fc, q) >> shape(..) >> highpass_hz(fc2, q2)
using SvfShape = Pipe<FixedSvf, Shaper>;
using SvfShapeSvf = Pipe<SvfShape, FixedSvf>;

// oversample(..) (prelude32.rs:983): the README's canonical FM patch  oversample(sine_hz(f) * f * m + f >> sine())
// (README.md:1631), and an oversampled waveshaper  oversample(shape(..))  as the 1-in 1-out case
using OversampleFm = Oversampler<Pipe<FmMod, Sine>>;
using OversampleShape = Oversampler<Shaper>;
// resample(sine_hz(f) * f * m + f >> sine()) (prelude32.rs:1021): the FM pair played back at a per-sample speed
using ResampleFm = Resample<Pipe<FmMod, Sine>>;

void register_fm_kinds(std::vector<KindOps>& out);  // fd_kinds_fm.hip
void attach_fm_mix(std::vector<KindOps>& out);      // fd_kinds_fm_mix.hip
void attach_graph_mix(std::vector<KindOps>& out);   // fd_kinds_graph_mix.hip

void register_graph_kinds(std::vector<KindOps>& out) {
    register_fm_kinds(out);
    out.push_back(make_kind<NoiseBiquad>("noise_biquad"));
    out.push_back(make_kind<SawMoogAdsrPan>("saw_moog_adsr_pan"));
    out.push_back(make_kind<SawMoogVarAdsrPan>("saw_moog_var_adsr_pan"));
    out.push_back(make_kind<OversampleFm>("oversample_fm"));
    out.push_back(make_kind<OversampleShape>("oversample_shape"));
    out.push_back(make_kind<ResampleFm>("resample_fm"));
    out.push_back(make_kind<SvfShape>("svf_shape"));
    out.push_back(make_kind<SvfShapeSvf>("svf_shape_svf"));
    // render + mix-down in one launch for the kinds of the BASELINE configs
    attach_fm_mix(out);
    attach_graph_mix(out);
}
}  // namespace fd

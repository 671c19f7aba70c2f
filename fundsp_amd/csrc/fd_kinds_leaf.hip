// fd_kinds_leaf.hip -- bank kernels for single leaf nodes (each replaces that node's AudioNode::process).
#include "fd_engine.hpp"

namespace fd {
void register_leaf_kinds(std::vector<KindOps>& out) {
    out.push_back(make_kind<Pass>("pass"));
    out.push_back(make_kind<Sine>("sine"));
    out.push_back(make_kind<Noise>("noise"));
    out.push_back(make_kind<FixedSvf>("fixed_svf"));
    out.push_back(make_kind<Svf<3>>("svf3"));
    out.push_back(make_kind<Svf<4>>("svf4"));
    out.push_back(make_kind<Biquad>("biquad"));
    out.push_back(make_kind<BiquadT<98>>("biquad_bank"));
    out.push_back(make_kind<ButterLowpass<1>>("butterpass_hz"));
    out.push_back(make_kind<ButterLowpass<2>>("butterpass"));
    out.push_back(make_kind<Resonator<1>>("resonator_hz"));
    out.push_back(make_kind<Resonator<3>>("resonator"));
    out.push_back(make_kind<Moog<1>>("moog_hz"));
    out.push_back(make_kind<Moog<3>>("moog"));
    out.push_back(make_kind<Fir<2>>("fir2"));
    out.push_back(make_kind<Fir<3>>("fir3"));
    out.push_back(make_kind<Tick<1>>("tick"));
    out.push_back(make_kind<WaveSynth<0>>("saw"));
    out.push_back(make_kind<WaveSynth<1>>("square"));
    out.push_back(make_kind<WaveSynth<2>>("triangle"));
    out.push_back(make_kind<AdsrLive>("adsr_live"));
    out.push_back(make_kind<Panner>("pan"));
    out.push_back(make_kind<OnePole<OP_LOWPOLE, 1>>("lowpole_hz"));
    out.push_back(make_kind<OnePole<OP_LOWPOLE, 2>>("lowpole"));
    out.push_back(make_kind<OnePole<OP_HIGHPOLE, 1>>("highpole_hz"));
    out.push_back(make_kind<OnePole<OP_HIGHPOLE, 2>>("highpole"));
    out.push_back(make_kind<OnePole<OP_DCBLOCK, 1>>("dcblock_hz"));
    out.push_back(make_kind<OnePole<OP_ALLPOLE, 1>>("allpole_delay"));
    out.push_back(make_kind<OnePole<OP_ALLPOLE, 2>>("allpole"));
    out.push_back(make_kind<Pinkpass>("pinkpass"));
    out.push_back(make_kind<Rez<1>>("rez_hz"));
    out.push_back(make_kind<Rez<3>>("rez"));
    out.push_back(make_kind<Follow>("follow"));
    out.push_back(make_kind<AFollow>("afollow"));
    out.push_back(make_kind<Mls>("mls"));
    out.push_back(make_kind<Pluck>("pluck"));
    out.push_back(make_kind<Envelope<EnvExp>>("lfo_exp"));
    out.push_back(make_kind<Envelope<EnvSineHz>>("lfo_sine_hz"));
    out.push_back(make_kind<EnvelopeIn<EnvInExp>>("lfo2_exp"));
    out.push_back(make_kind<Dsf<1>>("dsf_r"));
    out.push_back(make_kind<Dsf<2>>("dsf"));
    out.push_back(make_kind<Morph>("morph"));
    out.push_back(make_kind<Delay>("delay"));
    out.push_back(make_kind<TapT<false>>("tap"));
    out.push_back(make_kind<TapT<true>>("tap_linear"));
    out.push_back(make_kind<AllNest<Delay>>("allnest_delay"));
    out.push_back(make_kind<AllNest<Tick<1>>>("allnest_tick"));
    out.push_back(make_kind<AllNest<Pass>>("allnest_pass"));
    out.push_back(make_kind<Shaper>("shape"));
    out.push_back(make_kind<PhaseOsc<OSC_RAMP>>("ramp"));
    out.push_back(make_kind<PhaseOsc<OSC_POLYSAW>>("poly_saw"));
    out.push_back(make_kind<PhaseOsc<OSC_POLYSQUARE>>("poly_square"));
    out.push_back(make_kind<PhaseOsc<OSC_POLYPULSE>>("poly_pulse"));
    out.push_back(make_kind<Chaos<false>>("rossler"));
    out.push_back(make_kind<Chaos<true>>("lorenz"));
    out.push_back(make_kind<NlBiquad<false, 1>>("fbiquad_hz"));
    out.push_back(make_kind<NlBiquad<true, 1>>("dbiquad_hz"));
    out.push_back(make_kind<NlBiquad<false, 3>>("fbiquad3"));
    out.push_back(make_kind<NlBiquad<false, 4>>("fbiquad4"));
    out.push_back(make_kind<NlBiquad<true, 3>>("dbiquad3"));
    out.push_back(make_kind<NlBiquad<true, 4>>("dbiquad4"));
}
}  // namespace fd

// fd_kinds_graph_mix.hip -- the pipeline kernels with the fused mix-down (fd_device.hpp k_render_pipe_mix) of the config-2 and
// config-4 voices: BASELINE config 4 is "256k-voice subtractive synth ... RCCL stereo mix-down" -- its voices end in a Panner, so
// the per-GPU partial of the mix is the plain sum of the two output channels over the voices (MIX_SUM).  A translation unit of its
// own: the instantiations build in parallel with the voice-out kernels.
#include "fd_kinds_graph.hpp"

namespace fd {
void attach_graph_mix(std::vector<KindOps>& out) {
    attach_mix<NoiseBiquad>(out, "noise_biquad");
    attach_mix<SawMoogAdsrPan>(out, "saw_moog_adsr_pan");
    attach_mix<SawMoogVarAdsrPan>(out, "saw_moog_var_adsr_pan");
}
}  // namespace fd

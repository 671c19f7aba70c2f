"""fundsp_amd -- MI355X (gfx950) voice-bank renderer behind FunDSP's AudioNode::process boundary.

The compute path is hand-written HIP (fundsp_amd/csrc) behind the C ABI in include/fundsp_hip.h; this package is
the thin host-side mirror of the reference's AudioNode surface used by tests and bench.py.
"""
from ._lib import (BUS_DRY_WET, BUS_NONE, BUS_WET, DEFAULT_SR, FADE_POWER, FADE_SMOOTH, LAYOUT_PLANAR, LAYOUT_VOICE_MINOR, MATH_EXACT, MATH_FAST, MAX_BUFFER_SIZE, MIX_PAN, MIX_SUM, MODE_PROCESS, MODE_TICK,  # noqa: F401
                   FdspError, lib)
from .bank import (Bank, Chain, Comm, biquad_coefs, kind_slots, kinds, mix_stereo, sum_instances, sum_voices, svf_coefs, wavetable_build,  # noqa: F401
                   wave_upload, wavetable_compute, wavetable_get, wavetable_upload)

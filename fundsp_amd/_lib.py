"""ctypes binding of the C ABI declared in include/fundsp_hip.h (libfundsp_hip.so, built in-tree for gfx950).

There is no CPU fallback: if the HIP library is missing the import of anything that needs it raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FUNDSP_HIP_LIB overrides the in-tree library (A/B builds of the same C ABI, e.g. tools/build_variants.sh)
SO_PATH = os.environ.get("FUNDSP_HIP_LIB") or os.path.join(_HERE, "libfundsp_hip.so")

OK, EINVAL, ENOMEM, EDEVICE, ENOTSUP = 0, -1, -2, -3, -4
LAYOUT_VOICE_MINOR, LAYOUT_PLANAR = 0, 1
MODE_PROCESS, MODE_TICK = 0, 1
MATH_EXACT, MATH_FAST = 0, 1
BUS_NONE, BUS_WET, BUS_DRY_WET = 0, 1, 2  # fdsp_bank_set_bus: the node alone | wet * node | dry * multipass() & wet * node
MIX_SUM, MIX_PAN = 1, 2  # fdsp_bank_process_mix: sum the output channels over the voices | pan a mono graph per voice, then sum
FADE_POWER, FADE_SMOOTH = 0, 1  # sequencer.rs Fade::Power / Fade::Smooth
MAX_BUFFER_SIZE = 64
DEFAULT_SR = 44100.0

# every symbol include/fundsp_hip.h declares: name -> (restype, argtypes)
_P, _i, _f, _d, _sz, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t, C.c_uint64
_fp, _u64p, _cs = C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.c_char_p
SYMBOLS = {
    "fdsp_last_error": (_cs, []),
    "fdsp_kind_count": (_i, []),
    "fdsp_kind_name": (_cs, [_i]),
    "fdsp_kind_by_name": (_i, [_cs]),
    "fdsp_set_option": (_i, [_cs, _i]),
    "fdsp_bank_set_option": (_i, [_P, _cs, _i]),
    "fdsp_bank_get_option": (_i, [_P, _cs]),
    "fdsp_graph_compile": (_i, [_cs, _cs]),
    "fdsp_graph_compile_src": (_i, [_cs, _cs, _cs]),
    "fdsp_graph_check": (_i, [_cs]),
    "fdsp_rust_type_to_expr": (_i, [_cs, _cs, C.c_char_p, _sz, C.c_char_p, _sz]),
    "fdsp_graph_compile_rust": (_i, [_cs, _cs, _cs, _cs]),
    "fdsp_kind_inputs": (_i, [_i]),
    "fdsp_kind_outputs": (_i, [_i]),
    "fdsp_kind_slot_count": (_i, [_i]),
    "fdsp_kind_slot_name": (_cs, [_i, _i]),
    "fdsp_kind_slot_kind": (_i, [_i, _i]),
    "fdsp_bank_create": (_i, [_cs, _sz, C.POINTER(_P)]),
    "fdsp_bank_create_ring": (_i, [_cs, _sz, _sz, C.POINTER(_P)]),
    "fdsp_reverb_stereo_create": (_i, [_sz, _d, _d, _d, C.POINTER(_P)]),
    "fdsp_reverb4_stereo_create": (_i, [_sz, _d, _d, C.POINTER(_P)]),
    "fdsp_reverb4_stereo_create_on": (_i, [_i, _sz, _d, _d, C.POINTER(_P)]),
    "fdsp_reverb3_stereo_create": (_i, [_sz, _d, _d, _f, C.POINTER(_P)]),
    "fdsp_reverb3_stereo_create_on": (_i, [_i, _sz, _d, _d, _f, C.POINTER(_P)]),
    "fdsp_reverb3_stereo_svf_create": (_i, [_sz, _d, _d, _i, _f, _f, _f, C.POINTER(_P)]),
    "fdsp_reverb3_stereo_svf_create_on": (_i, [_i, _sz, _d, _d, _i, _f, _f, _f, C.POINTER(_P)]),
    "fdsp_fdn_create": (_i, [_sz, _i, C.POINTER(_d), _i, C.POINTER(C.c_float), _i, _i, C.POINTER(_P)]),
    "fdsp_fdn_create_on": (_i, [_i, _sz, _i, C.POINTER(_d), _i, C.POINTER(C.c_float), _i, _i, C.POINTER(_P)]),
    "fdsp_bank_set_bus": (_i, [_P, _i, _f, _f]),
    "fdsp_bank_get_bus": (_i, [_P, C.POINTER(_i), C.POINTER(_f), C.POINTER(_f)]),
    "fdsp_jit_compiler": (C.c_char_p, []),
    "fdsp_device_count": (_i, []),
    "fdsp_bank_create_on": (_i, [_i, _cs, _sz, _sz, C.POINTER(_P)]),
    "fdsp_reverb_stereo_create_on": (_i, [_i, _sz, _d, _d, _d, C.POINTER(_P)]),
    "fdsp_bank_device": (_i, [_P]),
    "fdsp_bank_destroy": (None, [_P]),
    "fdsp_bank_clone": (_i, [_P, C.POINTER(_P)]),
    "fdsp_bank_inputs": (_i, [_P]),
    "fdsp_bank_outputs": (_i, [_P]),
    "fdsp_bank_voices": (_sz, [_P]),
    "fdsp_bank_set_sample_rate": (_i, [_P, _d]),
    "fdsp_bank_reset": (_i, [_P]),
    "fdsp_bank_set_seed": (_i, [_P, _u64p, _sz, _sz]),
    "fdsp_bank_slot_count": (_i, [_P]),
    "fdsp_bank_slot_name": (_cs, [_P, _i]),
    "fdsp_bank_slot_kind": (_i, [_P, _i]),
    "fdsp_bank_set_param": (_i, [_P, _cs, _fp, _sz, _sz]),
    "fdsp_bank_set_param_all": (_i, [_P, _cs, _f]),
    "fdsp_bank_set_param_u64": (_i, [_P, _cs, _u64p, _sz, _sz]),
    "fdsp_bank_get_slot": (_i, [_P, _cs, _fp, _sz, _sz]),
    "fdsp_bank_get_state": (_i, [_P, _fp]),
    "fdsp_bank_set_state": (_i, [_P, _fp]),
    "fdsp_bank_process": (_i, [_P, _sz, _P, _P, _i, _sz, _i, _P]),
    "fdsp_bank_process_host": (_i, [_P, _sz, _fp, _fp, _i, _sz, _i]),
    "fdsp_bank_set_ring": (_i, [_P, _i, _fp, _sz, _sz, _sz]),
    "fdsp_bank_set_events": (_i, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int), _sz, _sz]),
    "fdsp_bank_process_events": (_i, [_P, _sz, _P, _P, _i, _P]),
    "fdsp_bank_process_events_mix": (_i, [_P, _sz, _P, _P, _i, _P]),
    "fdsp_bank_events_rewind": (_i, [_P, _d]),
    "fdsp_bank_events_time": (_d, [_P]),
    "fdsp_bank_synchronize": (_i, [_P]),
    "fdsp_bank_last_kernel_ms": (_i, [_P, C.POINTER(C.c_float)]),
    "fdsp_bank_process_mix": (_i, [_P, _sz, _P, _P, _i, _i, _P]),
    "fdsp_bank_set_pan": (_i, [_P, _fp, _sz, _sz]),
    "fdsp_bank_mix_reserve": (_i, [_P, _sz]),
    "fdsp_mix_stereo": (_i, [_P, _P, _P, _sz, _sz, _P]),
    "fdsp_sum_voices": (_i, [_P, _P, _sz, _sz, _sz, _P]),
    "fdsp_sum_instances": (_i, [_P, _P, _sz, _sz, _P]),
    "fdsp_comm_create_local": (_i, [_i, C.POINTER(C.c_int), C.POINTER(_P)]),
    "fdsp_comm_unique_id": (_i, [_P]),
    "fdsp_comm_create_rank": (_i, [_P, _i, _i, _i, C.POINTER(_P)]),
    "fdsp_comm_destroy": (None, [_P]),
    "fdsp_comm_ranks": (_i, [_P]),
    "fdsp_comm_local_slots": (_i, [_P]),
    "fdsp_comm_device": (_i, [_P, _i]),
    "fdsp_mix_allreduce": (_i, [_P, _i, _P, _sz, _P]),
    "fdsp_mix_allreduce_all": (_i, [_P, C.POINTER(_P), _sz, C.POINTER(_P)]),
    "fdsp_comm_wait": (_i, [_P, _i, _P]),
    "fdsp_wavetable_build": (_i, [_i]),
    "fdsp_wave_upload": (_i, [_i, _i, _sz, _fp]),
    "fdsp_wavetable_upload": (_i, [_i, _i, _fp, C.POINTER(C.c_int), _fp]),
    "fdsp_wavetable_get": (_i, [_i, C.POINTER(C.c_int), _fp, C.POINTER(C.c_int), _fp, _sz]),
    "fdsp_wavetable_compute": (_i, [_i, C.POINTER(C.c_int), _fp, C.POINTER(C.c_int), _fp, _sz]),
    "fdsp_svf_coefs": (_i, [_i, _f, _f, _f, _f, _fp]),
    "fdsp_biquad_coefs": (_i, [_i, _f, _f, _f, _f, _fp]),
    "fdsp_rnd1": (_d, [_u64]),
    "fdsp_libm_sinf": (_f, [_f]),
    "fdsp_libm_cosf": (_f, [_f]),
    "fdsp_hash1": (_u64, [_u64]),
}

_lib = None


class FdspError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"fundsp_hip error {code}: {msg}")
        self.code = code


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                f"{SO_PATH} not found: build the HIP engine first (python -c 'import __graft_entry__ as g; g.build()' "
                "or make -C fundsp_amd/csrc). There is no CPU fallback.")
        # PyTorch-ROCm ships its own HIP runtime; if ours initialises first in the process, torch later reports
        # "No HIP GPUs are available".  Device memory handed to the C ABI comes from torch, so let torch go first.
        try:
            import torch

            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(L, name)  # AttributeError here = header/library mismatch
            except AttributeError:
                # the in-tree library must export everything; an A/B build selected through FUNDSP_HIP_LIB (tools/variants: an older
                # source tree with retired switches) may predate an entry point -- the calls it lacks then fail where they are made
                if not os.environ.get("FUNDSP_HIP_LIB"):
                    raise
                continue
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != OK:
        raise FdspError(rc, lib().fdsp_last_error().decode())
    return rc

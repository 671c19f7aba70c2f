"""Host-side mirror of the reference's AudioNode surface for a voice bank (V lock-step instances of one graph).

Method names and argument meaning follow `AudioNode` / `AudioUnit` (reference src/audionode.rs:29-369,
src/audiounit.rs:21-371): inputs/outputs, reset, set_sample_rate, set_seed, process, tick, and `set(...)` via
named per-voice parameters.  Everything calls the C ABI (include/fundsp_hip.h); torch is used only to own
device memory and to pick the current HIP stream.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FdspError  # noqa: E402
from ._lib import FADE_POWER, FADE_SMOOTH, LAYOUT_PLANAR, LAYOUT_VOICE_MINOR, MIX_PAN, MIX_SUM, MODE_PROCESS, MODE_TICK, check, lib

SVF_MODES = dict(lowpass=0, highpass=1, bandpass=2, notch=3, peak=4, allpass=5, bell=6, lowshelf=7, highshelf=8)
BQ_KINDS = dict(butter=0, resonator=1, lowpass=2, highpass=3, bell=4)


def kinds():
    L = lib()
    return [L.fdsp_kind_name(k).decode() for k in range(L.fdsp_kind_count())]


def kind_slots(kind):
    L = lib()
    k = L.fdsp_kind_by_name(kind.encode())
    if k < 0:
        raise KeyError(kind)
    return [(L.fdsp_kind_slot_name(k, i).decode(), L.fdsp_kind_slot_kind(k, i)) for i in range(L.fdsp_kind_slot_count(k))]


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Bank:
    """V voices of one compiled voice graph on the current HIP device."""

    def __init__(self, kind, voices, _handle=None, ring_frames=0, device=-1):
        """`device`: HIP device index the bank lives on (-1 = the calling thread's current device)."""
        self._h = C.c_void_p()
        self.kind = kind
        if _handle is None:
            check(lib().fdsp_bank_create_on(int(device), kind.encode(), int(voices), int(ring_frames), C.byref(self._h)))
        else:
            self._h = _handle
        self.voices = int(voices)
        self.sample_rate = _lib.DEFAULT_SR

    @classmethod
    def from_graph(cls, graph, voices, ring_frames=0, sample_rate=None, fdn_kernel=True, flush_denormals=False):
        """Compile `graph` (fundsp_amd.graph notation) into a fused kernel set at run time and build a bank of `voices`
        instances of it with the graph's parameters applied (scalars to every voice, arrays per voice).
        A graph that IS a Hadamard feedback delay network (graph.fdn_plan: split >> fdn(stacki(delay >> fir)) >> join, uniform parameters,
        every delay longer than two blocks) becomes a bank of the lane-per-frame FDN kernel instead (fdsp_fdn_create) -- the same samples,
        two orders of magnitude faster than one lane per voice; `fdn_kernel=False` keeps the run-time compiled form.
        `flush_denormals`: compile the graph with f32 denormals flushed although it has no Feedback node of its own -- the front half of a chain
        whose other half has one (Feedback::new's prevent_denormals() sets FTZ + DAZ for the constructing thread, feedback.rs:96, denormal.rs:18: the
        reference renders the WHOLE graph flushed, and so does the oracle)."""
        from . import graph as G

        stock = getattr(graph, "stock_reverb", None) if fdn_kernel else None
        if stock is not None and graph.type.startswith("Pipe<") and "Feedback<" in graph.type:
            # reverb_stereo(..) / reverb4_stereo(..) themselves (graph.py marks the object its constructor returns): their lane-per-frame kernels
            try:
                b = cls.reverb_stereo(voices, *stock[1]) if stock[0] == "reverb_stereo" else cls.reverb4_stereo(voices, *stock[1])
                if sample_rate is not None:
                    b.set_sample_rate(sample_rate)
                b.reset()
                return b
            except FdspError:
                pass   # (a room too small for the kernel's two-block rule at this rate: the run-time compiled graph renders it)
        rv3 = getattr(graph, "reverb3_plan", None) if fdn_kernel else None
        if rv3 is not None and graph.type.startswith("Reverb3<"):   # reverb3_stereo(time, diffusion, lowpole_hz(cutoff)) itself: its lane-per-frame kernel
            if min(_lib.DEFAULT_SR, float(sample_rate or _lib.DEFAULT_SR)) >= 14200.0:
                b = cls.reverb3_stereo(voices, **rv3)
                if sample_rate is not None:
                    b.set_sample_rate(sample_rate)
                b.reset()
                return b
        plan = G.fdn_plan(graph) if fdn_kernel else None
        if plan is not None:
            rates = {_lib.DEFAULT_SR, float(sample_rate or _lib.DEFAULT_SR)}   # the bank is constructed at DEFAULT_SR and then moved
            if all(int(np.floor(t * r + 0.5)) >= 128 for t in plan["delays"] for r in rates):
                b = cls.fdn(voices, **plan)
                if sample_rate is not None:
                    b.set_sample_rate(sample_rate)
                b.reset()
                return b
        bus = G.bus_plan(graph) if fdn_kernel else None
        if bus is not None and G.lane_per_frame_shape(bus[0]):
            # `multipass() & 0.2 * reverb_stereo(..)` (README.md:436) and its relatives: the reverb's own bank with the gain and the dry bus folded into
            # its kernel's epilogue (fdsp_bank_set_bus) -- the Unop, MultiPass and Bus nodes cost two multiplications and an addition per output sample
            eff = cls.from_graph(bus[0], voices, sample_rate=sample_rate)
            if isinstance(eff, Bank) and eff.kind in LANE_PER_FRAME_KINDS:
                eff.set_bus(*bus[1:])
                return eff
            eff.close()   # (a room too small for the kernel's two-block rule: the whole graph renders lane-per-voice)
        parts = getattr(graph, "pipe_parts", None) if fdn_kernel else None
        pbus = G.bus_plan(parts[1]) if parts is not None else None
        if parts is not None and (G.lane_per_frame_shape(parts[1]) or (pbus is not None and G.lane_per_frame_shape(pbus[0]))):
            # `front >> stock reverb / network [with its bus]` -- the reference's own `reverb` bench, (noise() | noise()) >> reverb_stereo(..); an
            # instrument or an effect chain in front of `multipass() & 0.2 * reverb_stereo(..)` (examples/beep.rs:105) --: compiled as ONE
            # lane-per-voice graph the delay lines are read one lane per instance; as a chain the front keeps its fused kernel (inputs, delay
            # rings and all) and the network its lane-per-frame kernel -- the same samples (the front is seeded as the Pipe would seed it, and
            # flushes denormals when the network has a Feedback node, as the one graph would), 200-700 x faster
            eff = cls.from_graph(parts[1], voices, sample_rate=sample_rate)
            if isinstance(eff, Bank) and eff.kind in LANE_PER_FRAME_KINDS:
                src = cls.from_graph(parts[0], voices, ring_frames=ring_frames, sample_rate=sample_rate, flush_denormals=flush_denormals or "Feedback<" in parts[1].type)
                return Chain(src, eff, construction_hash=probe_hash(graph))
            eff.close()
        name, ctype, csrc = graph.kind_name(), graph.type, graph.source
        if flush_denormals and "Feedback" not in ctype:
            # the run-time compiler flushes a kind whose type expression names a Feedback node (fd_jit.hip): hand it the same type under an alias
            # that does -- the type itself, every kernel specialisation keyed on it and the slot names are unchanged
            import hashlib

            alias = "FeedbackThreadFlushed_" + hashlib.sha1((ctype + "\0" + csrc).encode()).hexdigest()[:16]
            csrc = (csrc + "\n" if csrc else "") + f"using {alias} = {ctype};"
            ctype = alias
            name = "jit_" + hashlib.sha1((ctype + "\0" + csrc).encode()).hexdigest()[:16]
        rc = lib().fdsp_graph_compile_src(name.encode(), ctype.encode(), csrc.encode() if csrc else None)
        if rc < 0:
            check(rc)
        for kind in G.uses_wavetables(graph):
            if wavetable_get(kind)[0] is None:
                wavetable_build(kind)
        if graph.rings and not ring_frames:
            # the bank is constructed at DEFAULT_SR and then moved to sample_rate: the rings must hold both
            need = [graph.ring_frames(r) for r in {_lib.DEFAULT_SR, float(sample_rate or _lib.DEFAULT_SR)}]
            if None in need:
                raise ValueError("this graph has a node whose ring size the graph cannot derive (Pluck / Hold / Reverb3): pass ring_frames")
            ring_frames = max(need)
        b = cls(name, voices, ring_frames=ring_frames)
        for slot, value, is_u64 in graph.slot_values():
            if is_u64 == 2:   # a u32 slot: one word per voice, uploaded as raw bits
                v = np.asarray(value, dtype=np.uint32)
                b.set_param(slot, (np.full(voices, v, dtype=np.uint32) if v.ndim == 0 else v).view(np.float32))
            elif is_u64:
                v = np.asarray(value, dtype=np.uint64)
                b.set_param_u64(slot, np.full(voices, v, dtype=np.uint64) if v.ndim == 0 else v)
            else:
                v = np.asarray(value, dtype=np.float32)
                b.set_param(slot, float(v) if v.ndim == 0 else v)
        if sample_rate is not None:
            b.set_sample_rate(sample_rate)
        b.reset()  # builders such as .phase() / .seed() take effect on reset (combinator.rs:263-267)
        return b

    @classmethod
    def reverb_stereo(cls, instances, room_size, time, damping):
        """Bank of `instances` x reverb_stereo(room_size, time, damping) (32-line FDN, prelude.rs:1732)."""
        h = C.c_void_p()
        check(lib().fdsp_reverb_stereo_create(int(instances), float(room_size), float(time), float(damping), C.byref(h)))
        return cls("reverb_stereo", instances, _handle=h)

    @classmethod
    def reverb4_stereo(cls, instances, room_size, time):
        """Bank of `instances` x reverb4_stereo(room_size, time) (two 16-line FDNs in series, prelude.rs:1873-1941)."""
        h = C.c_void_p()
        check(lib().fdsp_reverb4_stereo_create(int(instances), float(room_size), float(time), C.byref(h)))
        return cls("reverb4_stereo", instances, _handle=h)

    @classmethod
    def reverb3_stereo(cls, instances, time, diffusion, cutoff, device=-1, svf=None, q=1.0, gain=1.0):
        """Bank of `instances` x reverb3_stereo(time, diffusion, filter) (prelude.rs:1858-1871, reverb.rs:152-279: the allpass-loop reverb) through its
        lane-per-frame kernel.  The loop filter: lowpole_hz(cutoff) (the documented one; fdsp_reverb3_stereo_create), or with `svf` = "lowpass" ..
        "highshelf" the FixedSvf of that mode at (cutoff, q, gain) -- e.g. svf="highshelf" for the highshelf_hz(5000, 1, db_amp(-1)) of the
        reference's examples (fdsp_reverb3_stereo_svf_create)."""
        h = C.c_void_p()
        if svf is None:
            check(lib().fdsp_reverb3_stereo_create_on(int(device), int(instances), float(time), float(diffusion), float(cutoff), C.byref(h)))
        else:
            check(lib().fdsp_reverb3_stereo_svf_create_on(int(device), int(instances), float(time), float(diffusion), int(SVF_MODES[svf] if isinstance(svf, str) else svf),
                                                          float(cutoff), float(q), float(gain), C.byref(h)))
        return cls("reverb3_stereo", instances, _handle=h)

    @classmethod
    def fdn(cls, instances, lines, delays, taps, weights, inputs=1, outputs=1, device=-1):
        """Bank of `instances` x `split / multisplit >> fdn(stacki(lines, |i| delay(delays[i]) >> fir(weights))) >> join / multijoin`
        (prelude.rs:1323-1345) through the lane-per-frame FDN kernel (fdsp_fdn_create)."""
        h = C.c_void_p()
        d = (C.c_double * int(lines))(*[float(x) for x in delays])
        w = (C.c_float * int(taps))(*[float(x) for x in weights])
        check(lib().fdsp_fdn_create_on(int(device), int(instances), int(lines), d, int(taps), w, int(inputs), int(outputs), C.byref(h)))
        return cls("fdn", instances, _handle=h)

    def clone(self):
        """AudioNode: Clone -- a new bank that continues exactly where this one stands (fdsp_bank_clone: slots, rings, sample
        rate, arithmetic mode, launch options, scheduler events, reverb state)."""
        h = C.c_void_p()
        check(lib().fdsp_bank_clone(self._h, C.byref(h)))
        b = Bank(self.kind, self.voices, _handle=h)
        b.sample_rate = self.sample_rate
        return b

    def device(self):
        return lib().fdsp_bank_device(self._h)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().fdsp_bank_destroy(self._h)
            self._h = None

    def __del__(self):
        # at interpreter shutdown the module globals may already be gone; the process is about to release the device anyway
        if _lib is None or getattr(_lib, "_lib", None) is None:
            return
        self.close()

    # --- AudioNode surface
    def inputs(self):
        return lib().fdsp_bank_inputs(self._h)

    def outputs(self):
        return lib().fdsp_bank_outputs(self._h)

    def set_bus(self, mode, wet=1.0, dry=1.0):
        """A gain and a dry bus around a reverb / network bank, folded into its render kernel (fdsp_bank_set_bus): BUS_WET = `wet * node`,
        BUS_DRY_WET = `dry * multipass() & wet * node` (README.md:436: `multipass() & 0.2 * reverb_stereo(20.0, 2.0, 1.0)`), BUS_NONE = the node alone."""
        check(lib().fdsp_bank_set_bus(self._h, int(mode), float(wet), float(dry)))

    def get_bus(self):
        m, w, d = C.c_int(), C.c_float(), C.c_float()
        check(lib().fdsp_bank_get_bus(self._h, C.byref(m), C.byref(w), C.byref(d)))
        return m.value, w.value, d.value

    def set_sample_rate(self, sample_rate):
        check(lib().fdsp_bank_set_sample_rate(self._h, float(sample_rate)))
        self.sample_rate = float(sample_rate)

    def reset(self):
        check(lib().fdsp_bank_reset(self._h))

    def set_seed(self, seeds=None, first=0):
        """AudioNode::set_seed per voice. seeds=None re-applies the construction-time hash."""
        if seeds is None:
            check(lib().fdsp_bank_set_seed(self._h, None, first, self.voices - first))
            return
        s = np.ascontiguousarray(seeds, dtype=np.uint64)
        check(lib().fdsp_bank_set_seed(self._h, s.ctypes.data_as(C.POINTER(C.c_uint64)), first, s.size))

    # --- parameters (Setting)
    def slots(self):
        L = lib()
        return [(L.fdsp_bank_slot_name(self._h, i).decode(), L.fdsp_bank_slot_kind(self._h, i))
                for i in range(L.fdsp_bank_slot_count(self._h))]

    def set_param(self, name, values, first=0):
        if np.isscalar(values):
            check(lib().fdsp_bank_set_param_all(self._h, name.encode(), float(values)))
            return
        v = np.ascontiguousarray(values, dtype=np.float32)
        check(lib().fdsp_bank_set_param(self._h, name.encode(), _fptr(v), first, v.size))

    def set_param_u64(self, name, values, first=0):
        v = np.ascontiguousarray(values, dtype=np.uint64)
        check(lib().fdsp_bank_set_param_u64(self._h, name.encode(), v.ctypes.data_as(C.POINTER(C.c_uint64)), first, v.size))

    def get_slot(self, name, first=0, count=None):
        count = self.voices - first if count is None else count
        out = np.zeros(count, dtype=np.float32)
        check(lib().fdsp_bank_get_slot(self._h, name.encode(), _fptr(out), first, count))
        return out

    def get_state(self):
        out = np.zeros((len(self.slots()), self.voices), dtype=np.float32)
        check(lib().fdsp_bank_get_state(self._h, _fptr(out)))
        return out

    def set_state(self, state):
        s = np.ascontiguousarray(state, dtype=np.float32)
        assert s.shape == (len(self.slots()), self.voices)
        check(lib().fdsp_bank_set_state(self._h, _fptr(s)))

    # --- hot path
    def process(self, frames, inp=None, out=None, layout=LAYOUT_VOICE_MINOR, frame_stride=None, mode=MODE_PROCESS,
                stream=None):
        """Render `frames` samples per voice into a device tensor (torch, HBM-resident I/O).

        voice-minor: inp [inputs, frames, V], out [outputs, frames, V]
        planar:      inp [V, inputs, frame_stride], out [V, outputs, frame_stride]
        """
        import torch

        frames = int(frames)
        ni, no = self.inputs(), self.outputs()
        if layout == LAYOUT_PLANAR and frame_stride is None:
            frame_stride = max(frames, 1)
        fs = int(frame_stride or 0)
        if out is None:
            shape = (no, frames, self.voices) if layout == LAYOUT_VOICE_MINOR else (self.voices, no, fs)
            out = torch.empty(shape, dtype=torch.float32, device="cuda")
        assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous()
        if layout == LAYOUT_PLANAR:
            assert fs >= frames, f"frame_stride {fs} < frames {frames}"
        # the kernel trusts the pointers: a buffer smaller than the layout implies would be read / written out of bounds
        need_out = no * frames * self.voices if layout == LAYOUT_VOICE_MINOR else self.voices * no * fs
        assert out.numel() >= need_out, f"out has {out.numel()} floats, the {'voice-minor' if layout == LAYOUT_VOICE_MINOR else 'planar'} layout needs {need_out}"
        d_in = None
        if ni:
            assert inp is not None and inp.is_cuda and inp.dtype == torch.float32 and inp.is_contiguous()
            need_in = ni * frames * self.voices if layout == LAYOUT_VOICE_MINOR else self.voices * ni * fs
            assert inp.numel() >= need_in, f"inp has {inp.numel()} floats, the layout needs {need_in}"
            d_in = C.c_void_p(inp.data_ptr())
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        check(lib().fdsp_bank_process(self._h, frames, d_in, C.c_void_p(out.data_ptr()), layout, fs, mode,
                                      C.c_void_p(stream) if stream else None))
        return out

    def process_mix(self, frames, inp=None, mix=MIX_SUM, out=None, mode=MODE_PROCESS, stream=None):
        """Render `frames` samples per voice and reduce them over the voices in the same launch (fdsp_bank_process_mix): the
        per-voice output never exists in HBM.  mix = MIX_SUM -> [outputs, frames]; MIX_PAN (mono graphs, positions from
        set_pan) -> [2, frames].  inp: voice-minor [inputs, frames, V].  Same summation order as sum_voices / mix_stereo."""
        import torch

        frames = int(frames)
        ni, nm = self.inputs(), (2 if mix == MIX_PAN else self.outputs())
        if out is None:
            out = torch.empty((nm, frames), dtype=torch.float32, device="cuda")
        assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.numel() >= nm * frames
        d_in = None
        if ni:
            assert inp is not None and inp.is_cuda and inp.dtype == torch.float32 and inp.is_contiguous()
            assert inp.numel() >= ni * frames * self.voices, f"inp has {inp.numel()} floats, needs {ni * frames * self.voices}"
            d_in = C.c_void_p(inp.data_ptr())
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        check(lib().fdsp_bank_process_mix(self._h, frames, d_in, C.c_void_p(out.data_ptr()), int(mix), mode,
                                          C.c_void_p(stream) if stream else None))
        return out

    def set_pan(self, pan, first=0):
        """Pan position (-1 .. 1) per voice for MIX_PAN (Panner, pan.rs:13-17); every voice starts in the centre."""
        p = np.ascontiguousarray(pan, dtype=np.float32).reshape(-1)
        check(lib().fdsp_bank_set_pan(self._h, _fptr(p), first, p.size))

    def mix_reserve(self, frames):
        """Size the bank's partial-mix buffer for launches of up to `frames` frames (AudioNode::allocate semantics)."""
        check(lib().fdsp_bank_mix_reserve(self._h, int(frames)))

    def set_ring(self, ring_index, data, first=0):
        """Upload ring contents [voices][frames] (e.g. Pluck's excitation stream into ring 0)."""
        d = np.ascontiguousarray(data, dtype=np.float32)
        assert d.ndim == 2
        check(lib().fdsp_bank_set_ring(self._h, int(ring_index), _fptr(d), d.shape[1], first, d.shape[0]))

    # --- voice scheduler: Sequencer::push / process with one event per voice (sequencer.rs:355-398, 838-951)
    def set_events(self, start, end, fade_in=0.0, fade_out=0.0, fade=FADE_SMOOTH, first=0):
        """Per-voice events in seconds on the sequencer clock; scalars broadcast.  Resets nothing else."""
        start = np.atleast_1d(np.asarray(start, dtype=np.float64))
        n = start.size
        ev = np.empty((n, 4), dtype=np.float64)
        ev[:, 0] = start
        ev[:, 1] = np.broadcast_to(np.asarray(end, dtype=np.float64), (n,))
        ev[:, 2] = np.broadcast_to(np.asarray(fade_in, dtype=np.float64), (n,))
        ev[:, 3] = np.broadcast_to(np.asarray(fade_out, dtype=np.float64), (n,))
        fd = np.ascontiguousarray(np.broadcast_to(np.asarray(fade, dtype=np.int32), (n,)))
        check(lib().fdsp_bank_set_events(self._h, ev.ctypes.data_as(C.POINTER(C.c_double)),
                                         fd.ctypes.data_as(C.POINTER(C.c_int)), first, n))

    def events_rewind(self, time=0.0):
        check(lib().fdsp_bank_events_rewind(self._h, float(time)))

    def events_time(self):
        return lib().fdsp_bank_events_time(self._h)

    def process_events(self, frames, inp=None, out=None, mode=MODE_PROCESS, stream=None):
        """Sequencer rendering: out [outputs, frames, V] = every voice's faded contribution (0 outside its event)."""
        import torch

        frames = int(frames)
        ni, no = self.inputs(), self.outputs()
        if out is None:
            out = torch.empty((no, frames, self.voices), dtype=torch.float32, device="cuda")
        assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous()
        assert out.numel() >= no * frames * self.voices, f"out has {out.numel()} floats, needs {no * frames * self.voices}"
        d_in = None
        if ni:
            assert inp is not None and inp.is_cuda and inp.dtype == torch.float32 and inp.is_contiguous()
            assert inp.numel() >= ni * frames * self.voices, f"inp has {inp.numel()} floats, needs {ni * frames * self.voices}"
            d_in = C.c_void_p(inp.data_ptr())
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        check(lib().fdsp_bank_process_events(self._h, frames, d_in, C.c_void_p(out.data_ptr()), mode,
                                             C.c_void_p(stream) if stream else None))
        return out

    def process_events_mix(self, frames, inp=None, out=None, mode=MODE_PROCESS, stream=None):
        """Sequencer rendering, mixed: out [outputs, frames] = the sum of the events' faded contributions (fdsp_bank_process_events_mix:
        one launch, the mix-down's fixed order; equals sum_voices(process_events(..)) bit for bit)."""
        import torch

        frames = int(frames)
        ni, no = self.inputs(), self.outputs()
        if out is None:
            out = torch.empty((no, frames), dtype=torch.float32, device="cuda")
        assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.numel() >= no * frames
        d_in = None
        if ni:
            assert inp is not None and inp.is_cuda and inp.dtype == torch.float32 and inp.is_contiguous()
            assert inp.numel() >= ni * frames * self.voices, f"inp has {inp.numel()} floats, needs {ni * frames * self.voices}"
            d_in = C.c_void_p(inp.data_ptr())
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        check(lib().fdsp_bank_process_events_mix(self._h, frames, d_in, C.c_void_p(out.data_ptr()), mode,
                                                 C.c_void_p(stream) if stream else None))
        return out

    def process_host(self, frames, inp=None, layout=LAYOUT_PLANAR, frame_stride=None, mode=MODE_PROCESS, out=None):
        """Same with numpy buffers (staged, synchronous). Planar default: [V][channels][frame_stride].
        `out` (optional, right shape, C-contiguous f32) is written in place; planar padding past `frames` is left alone."""
        frames = int(frames)
        ni, no = self.inputs(), self.outputs()
        if layout == LAYOUT_PLANAR and frame_stride is None:
            frame_stride = max(frames, 1)
        fs = int(frame_stride or 0)
        shape = (no, frames, self.voices) if layout == LAYOUT_VOICE_MINOR else (self.voices, no, fs)
        if out is None:
            out = np.zeros(shape, dtype=np.float32)
        assert out.shape == shape and out.dtype == np.float32 and out.flags.c_contiguous, (out.shape, shape)
        h_in = None
        if ni:
            inp = np.ascontiguousarray(inp, dtype=np.float32)
            exp = (ni, frames, self.voices) if layout == LAYOUT_VOICE_MINOR else (self.voices, ni, fs)
            assert inp.shape == exp, (inp.shape, exp)
            h_in = _fptr(inp)
        check(lib().fdsp_bank_process_host(self._h, frames, h_in, _fptr(out), layout, fs, mode))
        return out

    def tick(self, frame=None):
        """AudioNode::tick for every voice: frame [V, inputs] -> [V, outputs]."""
        ni = self.inputs()
        inp = None
        if ni:
            inp = np.ascontiguousarray(frame, dtype=np.float32).reshape(self.voices, ni, 1)
        out = self.process_host(1, inp, layout=LAYOUT_PLANAR, frame_stride=1, mode=MODE_TICK)
        return out[:, :, 0]

    def synchronize(self):
        check(lib().fdsp_bank_synchronize(self._h))

    def set_option(self, name, value):
        """Per-bank engine option: "math" = MATH_EXACT (0, default: bit-identical to the reference arithmetic) or
        MATH_FAST (1, tolerance mode, include/fundsp_hip.h)."""
        check(lib().fdsp_bank_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name):
        rc = lib().fdsp_bank_get_option(self._h, name.encode())
        if rc < 0:
            check(rc)
        return rc

    def last_kernel_ms(self):
        ms = C.c_float()
        check(lib().fdsp_bank_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value


PIPE_ID = 6   # Pipe::ID (audionode.rs:1375-1492; fd_nodes.hpp)
LANE_PER_FRAME_KINDS = ("reverb_stereo", "reverb4_stereo", "reverb3_stereo", "fdn")

# A one-slot node whose constructor stores what `G::ping(true, AttoHash::new(G::ID))` returns -- the hash a combinator's constructor hands
# down to its nodes (audionode.rs:871-876, 1389-1394) -- so that the host can read it: the probe walks the TYPE only, no node state.
_PROBE_SRC = """
template <class G> struct HashProbe {
    static constexpr int IN = 0, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 0;
    uint64_t h;
    template <class V> FD_HD void visit(V& v) { v.u64(h, STATE, "probe"); }
    FD_HD void bind(Ctx&) {}
    FD_HD void init() { G g; h = g.ping(true, G::ID); }
    FD_HD void update(double) {}
    FD_HD void reset() {}
    FD_HD uint64_t ping(bool, uint64_t x) { return x; }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    template <int PH> FD_HD void step(const float*, float* out) { out[0] = 0.0f; }
    FD_STEP2_VIA_STEP
};
"""
_probe_cache = {}


def probe_hash(graph):
    """`G::ping(true, AttoHash::new(G::ID))` of the graph's type, evaluated on the device by a one-slot probe kind (compiled once per type)."""
    import hashlib

    key = graph.type + "\0" + graph.source
    if key not in _probe_cache:
        name = "jit_probe_" + hashlib.sha1(key.encode()).hexdigest()[:16]
        src = (graph.source + "\n" if graph.source else "") + _PROBE_SRC
        rc = lib().fdsp_graph_compile_src(name.encode(), f"HashProbe<{graph.type}>".encode(), src.encode())
        if rc < 0:
            check(rc)
        b = Bank(name, 1)
        w = b.get_state()[:2, 0].view(np.uint32)
        b.close()
        _probe_cache[key] = (int(w[1]) << 32) | int(w[0])
    return _probe_cache[key]



def atto(state, data):
    """AttoHash::hash (math.rs:649-658) on numpy u64 (scalars or arrays)."""
    s = np.asarray(state, dtype=np.uint64)
    with np.errstate(over="ignore"):
        r = (s << np.uint64(5)) | (s >> np.uint64(59))
        return (r ^ np.uint64(data)) * np.uint64(0x517CC1B727220A95)


class Chain:
    """Two banks in series, composed on the host: `source >> effect` where each half has the kernel family that suits it -- e.g. the
    reference's own `reverb` bench, `(noise() | noise()) >> reverb_stereo(10, 1, 0.5)` (benches/benchmark.rs:79-85): compiled as ONE
    lane-per-voice graph the 32 delay lines are read one lane per instance (uncoalesced rings); as a chain the generator runs in its fused
    kernel and the network in its lane-per-frame kernel (fdsp_reverb_stereo_create), two launches on one stream with a buffer in HBM
    between them, two to three orders of magnitude faster.  `Bank.from_graph` builds it by itself for `generator >> stock reverb / network`.

    Hashes.  `construction_hash` = what Pipe::new's probe ping returns for the WHOLE graph (`probe_hash`): the source then starts from the
    hash the Pipe's constructor would hand it -- Pipe::ping (audionode.rs:1459) gives its left side atto(hash, Pipe::ID) -- and the chain
    renders what the one graph renders.  Without it (a chain the host puts together from two banks) every half keeps the construction hash
    of a stand-alone node, as two AudioNodes piped by hand would.  `set_seed(seeds)` = AudioNode::set_seed of the Pipe.  The effect's own
    ping is skipped: exact for the stock reverbs and networks (no node of theirs keeps hashed state)."""

    def __init__(self, source, effect, construction_hash=None):
        if source.voices != effect.voices or source.outputs() != effect.inputs():
            raise ValueError(f"chain mismatch: {source.voices} x {source.outputs()} outputs into {effect.voices} x {effect.inputs()} inputs")
        self.source, self.effect, self.voices = source, effect, source.voices
        self.kind = f"chain({source.kind} >> {effect.kind})"
        self.sample_rate = effect.sample_rate
        self._mid = self._stream = None
        self._ctor = construction_hash
        if construction_hash is not None:
            self.set_seed(None)
            self.source.reset()

    def inputs(self):
        return self.source.inputs()

    def outputs(self):
        return self.effect.outputs()

    def set_sample_rate(self, sample_rate):
        self.source.set_sample_rate(sample_rate)
        self.effect.set_sample_rate(sample_rate)
        self.sample_rate = float(sample_rate)

    def reset(self):
        self.source.reset()
        self.effect.reset()

    def set_seed(self, seeds=None):
        """AudioNode::set_seed per instance; None re-applies the construction-time hash (as Bank.set_seed)."""
        if seeds is None:
            if self._ctor is None:
                self.source.set_seed(None)
            else:
                self.source.set_seed(np.full(self.voices, atto(np.uint64(self._ctor), PIPE_ID), dtype=np.uint64))
            return
        self.source.set_seed(atto(np.ascontiguousarray(seeds, dtype=np.uint64), PIPE_ID))

    def get_option(self, name):
        return self.effect.get_option(name)

    def set_option(self, name, value):
        self.source.set_option(name, value)
        self.effect.set_option(name, value)

    def clone(self):
        """AudioNode: Clone -- both halves continue exactly where they stand"""
        c = Chain(self.source.clone(), self.effect.clone())
        c._ctor, c.sample_rate = self._ctor, self.sample_rate
        return c

    def last_kernel_ms(self):
        return self.source.last_kernel_ms() + self.effect.last_kernel_ms()

    def synchronize(self):
        self.source.synchronize()
        self.effect.synchronize()

    def close(self):
        self.source.close()
        self.effect.close()

    @staticmethod
    def frame_stride(frames):
        """default row stride of a planar chain launch: `frames` rounded up to whole blocks"""
        return (int(frames) + 63) // 64 * 64

    def process(self, frames, inp=None, out=None, layout=LAYOUT_VOICE_MINOR, frame_stride=None, mode=MODE_PROCESS, stream=None):
        """As Bank.process: voice-minor out [outputs, frames, V] / planar out [V, outputs, frame_stride] (planar default stride: frames
        rounded up to whole blocks).  A launch has ONE layout and stride for its input and output, and the buffer between the halves is the
        source's output and the effect's input, so every buffer of the call has them.  Both launches go to ONE stream, nothing waits in
        between on the host: `stream` if given, else a stream of the chain's own that is ordered behind torch's current stream before the
        launches and in front of it after them (the C ABI reads a NULL stream -- torch's default stream -- as "the bank's own stream", and
        two banks' own streams do not order each other)."""
        import torch

        frames = int(frames)
        V = self.voices
        if layout == LAYOUT_PLANAR:
            fs = int(frame_stride) if frame_stride else self.frame_stride(frames)
            mshape, oshape = (V, self.source.outputs(), fs), (V, self.effect.outputs(), fs)
        else:
            fs = 0
            mshape, oshape = (self.source.outputs(), frames, V), (self.effect.outputs(), frames, V)
        if self._mid is None or tuple(self._mid.shape) != mshape:
            self._mid = torch.empty(mshape, dtype=torch.float32, device="cuda")
        if out is None:
            out = torch.empty(oshape, dtype=torch.float32, device="cuda")
        own = None
        if stream is None:
            if self._stream is None:
                self._stream = torch.cuda.Stream()
            own, cur = self._stream, torch.cuda.current_stream()
            own.wait_stream(cur)
            stream = own.cuda_stream
        self.source.process(frames, inp, self._mid, layout=layout, frame_stride=fs or None, mode=mode, stream=stream)
        self.effect.process(frames, self._mid, out, layout=layout, frame_stride=fs or None, mode=mode, stream=stream)
        if own is not None:
            cur.wait_stream(own)
        return out


def svf_coefs(mode, sample_rate, cutoff, q, gain=1.0):
    out = np.zeros(6, dtype=np.float32)
    check(lib().fdsp_svf_coefs(SVF_MODES[mode], sample_rate, cutoff, q, gain, _fptr(out)))
    return out


def biquad_coefs(kind, sample_rate, f, q=1.0, gain=1.0):
    """BiquadCoefs::{butter_lowpass,resonator,lowpass,highpass,bell} (biquad.rs:27-116) -> (a1,a2,b0,b1,b2)."""
    out = np.zeros(5, dtype=np.float32)
    check(lib().fdsp_biquad_coefs(BQ_KINDS[kind], sample_rate, f, q, gain, _fptr(out)))
    return out


WT_SETS = dict(saw=0, square=1, triangle=2, user=3, organ=4, soft_saw=5, hammond=6, user2=7)


def wave_upload(slot, data):
    """Install a shared sample buffer [channels][length] f32 (the Arc<Wave> of playwave(), wave.rs:739) in sample slot 0..7."""
    d = np.ascontiguousarray(np.atleast_2d(data), dtype=np.float32)
    check(lib().fdsp_wave_upload(int(slot), d.shape[0], d.shape[1], _fptr(d)))


def wavetable_build(kind):
    """Generate and install a built-in shared wavetable set (saw_table / square_table / triangle_table / organ_table /
    soft_saw_table / hammond_table, wavetable.rs:493-623)."""
    check(lib().fdsp_wavetable_build(WT_SETS[kind]))


def wavetable_upload(kind, pitches, waves):
    """Install caller-provided tables: pitches [n] ascending, waves = list of power-of-two-length float32 arrays."""
    p = np.ascontiguousarray(pitches, dtype=np.float32)
    lengths = np.array([len(w) for w in waves], dtype=np.int32)
    data = np.ascontiguousarray(np.concatenate(waves), dtype=np.float32)
    check(lib().fdsp_wavetable_upload(WT_SETS[kind], len(waves), _fptr(p), lengths.ctypes.data_as(C.POINTER(C.c_int)),
                                      _fptr(data)))


def wavetable_get(kind):
    n = C.c_int()
    check(lib().fdsp_wavetable_get(WT_SETS[kind], C.byref(n), None, None, None, 0))
    if n.value == 0:
        return None, None
    p = np.zeros(n.value, dtype=np.float32)
    lengths = np.zeros(n.value, dtype=np.int32)
    check(lib().fdsp_wavetable_get(WT_SETS[kind], C.byref(n), _fptr(p), lengths.ctypes.data_as(C.POINTER(C.c_int)), None, 0))
    data = np.zeros(int(lengths.sum()), dtype=np.float32)
    check(lib().fdsp_wavetable_get(WT_SETS[kind], C.byref(n), None, None, _fptr(data), data.size))
    offs = np.concatenate([[0], np.cumsum(lengths)])
    return p, [data[offs[i]:offs[i + 1]] for i in range(n.value)]


def wavetable_compute(kind):
    """The built-in table set as fdsp_wavetable_build would install it, computed on the host (no device needed)."""
    n = C.c_int()
    check(lib().fdsp_wavetable_compute(WT_SETS[kind], C.byref(n), None, None, None, 0))
    p = np.zeros(n.value, dtype=np.float32)
    lengths = np.zeros(n.value, dtype=np.int32)
    check(lib().fdsp_wavetable_compute(WT_SETS[kind], C.byref(n), _fptr(p), lengths.ctypes.data_as(C.POINTER(C.c_int)), None, 0))
    data = np.zeros(int(lengths.sum()), dtype=np.float32)
    check(lib().fdsp_wavetable_compute(WT_SETS[kind], C.byref(n), None, None, _fptr(data), data.size))
    offs = np.concatenate([[0], np.cumsum(lengths)])
    return p, [data[offs[i]:offs[i + 1]] for i in range(n.value)]


class Comm:
    """RCCL communicator for the stereo mix-down across GPUs (include/fundsp_hip.h, fdsp_comm_*).

    Comm.local(devices)  one process, several GPUs;  Comm.rank(id, nranks, rank, device)  one process per GPU, with the
    128-byte id from Comm.unique_id() on rank 0 shipped out of band (torch.distributed, MPI, a file)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def local(cls, devices):
        d = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        check(lib().fdsp_comm_create_local(len(devices), d, C.byref(h)))
        return cls(h)

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        check(lib().fdsp_comm_unique_id(buf))
        return bytes(buf.raw)

    @classmethod
    def rank(cls, unique_id, nranks, rank, device=-1):
        h = C.c_void_p()
        check(lib().fdsp_comm_create_rank(C.create_string_buffer(unique_id, 128), nranks, rank, device, C.byref(h)))
        return cls(h)

    def ranks(self):
        return lib().fdsp_comm_ranks(self._h)

    def allreduce(self, mix, slot=0, after_stream=None):
        """Sum the [2, frames] device tensor `mix` in place across the ranks, on the communicator's side stream, ordered
        behind `after_stream` (default: torch's current stream).  Returns at once; wait() orders a consumer behind it."""
        import torch

        assert mix.is_cuda and mix.dtype == torch.float32 and mix.is_contiguous()
        if after_stream is None:
            after_stream = torch.cuda.current_stream(mix.device).cuda_stream
        check(lib().fdsp_mix_allreduce(self._h, slot, C.c_void_p(mix.data_ptr()), mix.numel(),
                                       C.c_void_p(after_stream) if after_stream else None))
        return mix

    def allreduce_all(self, mixes, after_streams=None):
        """ONE host thread driving a Comm.local([...]) communicator of several GPUs: sum `mixes[k]` (the [2, frames] tensor on
        slot k's device) across all slots in one grouped call (fdsp_mix_allreduce_all: ncclGroupStart / End).  Per-slot
        allreduce() calls issued one after the other from a single thread would wait for each other."""
        import torch

        n = lib().fdsp_comm_local_slots(self._h)
        assert len(mixes) == n, f"{n} local slots, {len(mixes)} tensors"
        count = mixes[0].numel()
        for m in mixes:
            assert m.is_cuda and m.dtype == torch.float32 and m.is_contiguous() and m.numel() == count
        if after_streams is None:
            after_streams = [torch.cuda.current_stream(m.device).cuda_stream for m in mixes]
        ptrs = (C.c_void_p * n)(*[C.c_void_p(m.data_ptr()) for m in mixes])
        strs = (C.c_void_p * n)(*[C.c_void_p(s) if s else C.c_void_p(None) for s in after_streams])
        check(lib().fdsp_mix_allreduce_all(self._h, ptrs, count, strs))
        return mixes

    def wait(self, slot=0, stream=None):
        """stream=None blocks the HOST until the slot's last all-reduce is complete; stream="current" makes torch's current
        stream wait for it; any other value is a raw hipStream_t handle that waits (the NULL stream cannot be named this
        way: a handle of 0 means "block the host", as in the C ABI)."""
        if stream == "current":
            import torch

            stream = torch.cuda.current_stream().cuda_stream
            if not stream:       # torch's current stream IS the NULL stream: order it by blocking the host
                stream = None
        check(lib().fdsp_comm_wait(self._h, slot, C.c_void_p(stream) if stream else None))

    def close(self):
        if self._h:
            lib().fdsp_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        if _lib is None or getattr(_lib, "_lib", None) is None:
            return
        self.close()


def sum_voices(x, stream=None):
    """[channels, frames, V] voice-minor device tensor -> [channels, frames] (deterministic order)."""
    import torch

    assert x.is_cuda and x.dim() == 3 and x.is_contiguous()
    ch, frames, V = x.shape
    out = torch.empty((ch, frames), dtype=torch.float32, device=x.device)
    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    check(lib().fdsp_sum_voices(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), ch, frames, V,
                                C.c_void_p(stream) if stream else None))
    return out


def sum_instances(x, stream=None):
    """Sum over the instances of a planar render [instances, channels, frame_stride] -> [channels, frame_stride]
    (fdsp_sum_instances: the aligned binary tree of the mix-down's order, over the instances)."""
    import torch

    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 3
    n, ch, fs = x.shape
    out = torch.empty((ch, fs), dtype=torch.float32, device=x.device)
    if stream is None:
        stream = torch.cuda.current_stream(x.device).cuda_stream
    check(lib().fdsp_sum_instances(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), ch * fs, n, C.c_void_p(stream) if stream else None))
    return out


def mix_stereo(voices_out, pan=None, stream=None):
    """On-device stereo mix-down of a voice-minor mono render [frames, V] -> [2, frames]."""
    import torch

    assert voices_out.is_cuda and voices_out.dim() == 2 and voices_out.is_contiguous()
    frames, V = voices_out.shape
    mix = torch.empty((2, frames), dtype=torch.float32, device=voices_out.device)
    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    check(lib().fdsp_mix_stereo(C.c_void_p(voices_out.data_ptr()), C.c_void_p(pan.data_ptr()) if pan is not None else None,
                                C.c_void_p(mix.data_ptr()), frames, V, C.c_void_p(stream) if stream else None))
    return mix

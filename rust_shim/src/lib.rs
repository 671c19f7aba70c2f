//! `HipBank`: V voices of one FunDSP graph rendered by the MI355X engine, behind the reference's own `AudioNode` trait.
//!
//! The trait surface, the combinators, `Net` and `Wave::render` stay in Rust (BASELINE.json north_star); what moves is
//! `AudioNode::process` of the voices.  A bank of `V` voices with `I` inputs and `O` outputs per voice is an `AudioNode`
//! with `V * I` inputs and `V * O` outputs: `BufferRef` / `BufferMut` are planar `[channel][64] f32`
//! (src/buffer.rs:8-12), which is exactly `FDSP_LAYOUT_PLANAR` with `frame_stride = 64` when the channels are ordered
//! voice-major -- the slices are handed over without repacking.
//!
//! ```ignore
//! use fundsp::prelude32::*;
//! use fundsp_hip::HipBank;
//! // 512 voices of the README's FM patch; the graph TYPE is compiled for the device from its own type name
//! let voice = sine_hz(110.0) * 110.0 * 2.0 + 110.0 >> sine() >> lowpass_hz(900.0, 1.0);
//! let mut bank: HipBank<U0, U512> = HipBank::from_graph(&voice, 512).unwrap();
//! bank.set_param("0.0.0.0.0.0:value[0]", &freqs).unwrap();          // per-voice field values, by slot
//! let wave = Wave::render(48000.0, 1.0, &mut An(bank));              // unchanged executor: 64-sample process() blocks
//! ```

use core::ffi::{c_char, c_int, c_void};
use core::marker::PhantomData;
use fundsp::prelude32::*;
use std::ffi::{CStr, CString};

pub mod ffi {
    //! `extern "C"` mirror of include/fundsp_hip.h (the subset the shim uses; same names, same argument order).
    use super::*;

    #[repr(C)]
    pub struct FdspBank {
        _private: [u8; 0],
    }
    #[repr(C)]
    pub struct FdspComm {
        _private: [u8; 0],
    }
    pub const FDSP_OK: c_int = 0;
    pub const FDSP_LAYOUT_VOICE_MINOR: c_int = 0;
    pub const FDSP_LAYOUT_PLANAR: c_int = 1;
    pub const FDSP_MODE_PROCESS: c_int = 0;
    pub const FDSP_MODE_TICK: c_int = 1;
    pub const FDSP_MATH_EXACT: c_int = 0;
    pub const FDSP_MATH_FAST: c_int = 1;
    pub const FDSP_MIX_SUM: c_int = 1;
    pub const FDSP_MIX_PAN: c_int = 2;

    #[link(name = "fundsp_hip")]
    unsafe extern "C" {
        pub fn fdsp_last_error() -> *const c_char;
        pub fn fdsp_device_count() -> c_int;
        pub fn fdsp_kind_by_name(name: *const c_char) -> c_int;
        pub fn fdsp_graph_compile_rust(name: *const c_char, rust_type_name: *const c_char, hints: *const c_char, source: *const c_char) -> c_int;
        pub fn fdsp_bank_create_on(device: c_int, kind: *const c_char, voices: usize, ring_frames: usize, out: *mut *mut FdspBank) -> c_int;
        pub fn fdsp_reverb_stereo_create_on(device: c_int, instances: usize, room_size: f64, time: f64, damping: f64, out: *mut *mut FdspBank) -> c_int;
        pub fn fdsp_reverb4_stereo_create_on(device: c_int, instances: usize, room_size: f64, time: f64, out: *mut *mut FdspBank) -> c_int;
        pub fn fdsp_reverb3_stereo_create_on(device: c_int, instances: usize, time: f64, diffusion: f64, lowpole_cutoff_hz: f32, out: *mut *mut FdspBank) -> c_int;
        pub fn fdsp_reverb3_stereo_svf_create_on(device: c_int, instances: usize, time: f64, diffusion: f64, svf_mode: c_int, cutoff_hz: f32, q: f32, gain: f32, out: *mut *mut FdspBank) -> c_int;
        pub fn fdsp_fdn_create_on(device: c_int, instances: usize, lines: c_int, delays: *const f64, taps: c_int, weights: *const f32, inputs: c_int, outputs: c_int, out: *mut *mut FdspBank) -> c_int;
        pub fn fdsp_bank_destroy(bank: *mut FdspBank);
        pub fn fdsp_bank_clone(bank: *const FdspBank, out: *mut *mut FdspBank) -> c_int; // Clone: slots, rings, sample rate, options, events
        pub fn fdsp_bank_inputs(bank: *const FdspBank) -> c_int;
        pub fn fdsp_bank_outputs(bank: *const FdspBank) -> c_int;
        pub fn fdsp_bank_voices(bank: *const FdspBank) -> usize;
        pub fn fdsp_bank_device(bank: *const FdspBank) -> c_int;
        pub fn fdsp_bank_set_sample_rate(bank: *mut FdspBank, sample_rate: f64) -> c_int; // AudioNode::set_sample_rate
        pub fn fdsp_bank_reset(bank: *mut FdspBank) -> c_int; // AudioNode::reset
        pub fn fdsp_bank_set_seed(bank: *mut FdspBank, seeds: *const u64, first: usize, count: usize) -> c_int; // set_seed / ping
        pub fn fdsp_bank_set_param(bank: *mut FdspBank, name: *const c_char, values: *const f32, first: usize, count: usize) -> c_int;
        pub fn fdsp_bank_set_param_all(bank: *mut FdspBank, name: *const c_char, value: f32) -> c_int;
        pub fn fdsp_bank_set_option(bank: *mut FdspBank, name: *const c_char, value: c_int) -> c_int;
        pub fn fdsp_bank_set_bus(bank: *mut FdspBank, mode: c_int, wet: f32, dry: f32) -> c_int;
        pub fn fdsp_jit_compiler() -> *const c_char;
        pub fn fdsp_bank_slot_count(bank: *const FdspBank) -> c_int;
        pub fn fdsp_bank_get_state(bank: *mut FdspBank, slots: *mut f32) -> c_int; // Clone
        pub fn fdsp_bank_set_state(bank: *mut FdspBank, slots: *const f32) -> c_int;
        pub fn fdsp_bank_process_host(bank: *mut FdspBank, frames: usize, input: *const f32, output: *mut f32, layout: c_int,
                                      frame_stride: usize, mode: c_int) -> c_int; // AudioNode::process on host buffers
        pub fn fdsp_bank_process(bank: *mut FdspBank, frames: usize, d_in: *const f32, d_out: *mut f32, layout: c_int,
                                 frame_stride: usize, mode: c_int, stream: *mut c_void) -> c_int; // device-resident I/O
        pub fn fdsp_bank_process_mix(bank: *mut FdspBank, frames: usize, d_in: *const f32, d_mix: *mut f32, mix: c_int, mode: c_int,
                                     stream: *mut c_void) -> c_int; // render + reduce over the voices in one launch
        pub fn fdsp_bank_process_events_mix(bank: *mut FdspBank, frames: usize, d_in: *const f32, d_mix: *mut f32, mode: c_int, stream: *mut c_void) -> c_int; // Sequencer output
        pub fn fdsp_bank_set_pan(bank: *mut FdspBank, pan: *const f32, first: usize, count: usize) -> c_int;
        pub fn fdsp_bank_mix_reserve(bank: *mut FdspBank, frames: usize) -> c_int; // AudioNode::allocate for the mix path
        pub fn fdsp_bank_synchronize(bank: *mut FdspBank) -> c_int;
        pub fn fdsp_sum_voices(d_in: *const f32, d_out: *mut f32, channels: usize, frames: usize, voices: usize, stream: *mut c_void) -> c_int;
        pub fn fdsp_sum_instances(d_in: *const f32, d_out: *mut f32, rows: usize, instances: usize, stream: *mut c_void) -> c_int;
        pub fn fdsp_mix_stereo(d_voices: *const f32, d_pan: *const f32, d_mix: *mut f32, frames: usize, voices: usize, stream: *mut c_void) -> c_int;
        pub fn fdsp_comm_create_local(n: c_int, devices: *const c_int, out: *mut *mut FdspComm) -> c_int;
        pub fn fdsp_comm_unique_id(id128: *mut c_void) -> c_int;
        pub fn fdsp_comm_create_rank(id128: *const c_void, nranks: c_int, rank: c_int, device: c_int, out: *mut *mut FdspComm) -> c_int;
        pub fn fdsp_comm_destroy(comm: *mut FdspComm);
        pub fn fdsp_mix_allreduce(comm: *mut FdspComm, slot: c_int, d_mix: *mut f32, count: usize, after_stream: *mut c_void) -> c_int;
        pub fn fdsp_comm_wait(comm: *mut FdspComm, slot: c_int, stream: *mut c_void) -> c_int;
    }
}
use ffi::*;

/// The engine's last error message for this thread.
pub fn last_error() -> String {
    unsafe { CStr::from_ptr(fdsp_last_error()).to_string_lossy().into_owned() }
}

fn check(rc: c_int) -> Result<(), String> {
    if rc == FDSP_OK { Ok(()) } else { Err(last_error()) }
}

/// A bank of voices on one GPU.  `NI` = voices x inputs per voice, `NO` = voices x outputs per voice (typenum sizes, as
/// every `AudioNode` declares them).
///
/// The `AudioNode` face has a practical ceiling: `tick()` builds `Frame<f32, NO>` on the stack (a `GenericArray`: 4 B per
/// channel -- 256 KB at `U65536`) and every combinator above the bank instantiates typenum arithmetic over `NI` / `NO`.  The
/// reference uses `Size` at channel counts of tens (`U32` in its FDN reverbs, src/prelude.rs:1732-1762); use the trait face at
/// bank sizes of that order (a few hundred voices at most) and [`HipBank::render_device`] / [`HipBank::render_mix`] for the
/// BASELINE-sized banks, where `NI` / `NO` are only the arity check of `new` (pass `U0` and skip it with `new_unchecked_arity`).
pub struct HipBank<NI: Size<f32>, NO: Size<f32>> {
    bank: *mut FdspBank,
    kind: CString,
    voices: usize,
    ring_frames: usize,
    device: c_int,
    /// `tick` / `process` are infallible in FunDSP.  A failed launch (lost device, wrong arity) zeroes the output and is
    /// recorded here -- in release builds too -- for the host to poll with `take_error()`.
    last_error: Option<String>,
    _marker: PhantomData<(NI, NO)>,
}

// The handle may move between threads and is used by one thread at a time (`process(&mut self)`); `&self` methods do
// not touch the device.
unsafe impl<NI: Size<f32>, NO: Size<f32>> Send for HipBank<NI, NO> {}
unsafe impl<NI: Size<f32>, NO: Size<f32>> Sync for HipBank<NI, NO> {}

impl<NI: Size<f32>, NO: Size<f32>> HipBank<NI, NO> {
    /// `voices` instances of an ahead-of-time kind ("fm_svf", "fixed_svf", "biquad_bank" ...) on `device` (-1 = current).
    pub fn new(kind: &str, voices: usize, ring_frames: usize, device: i32) -> Result<Self, String> {
        let kind = CString::new(kind).map_err(|e| e.to_string())?;
        let mut bank: *mut FdspBank = core::ptr::null_mut();
        check(unsafe { fdsp_bank_create_on(device as c_int, kind.as_ptr(), voices, ring_frames, &mut bank) })?;
        let (i, o) = unsafe { (fdsp_bank_inputs(bank) as usize, fdsp_bank_outputs(bank) as usize) };
        if i * voices != NI::USIZE || o * voices != NO::USIZE {
            unsafe { fdsp_bank_destroy(bank) };
            return Err(format!("arity mismatch: the bank has {}x{} inputs and {}x{} outputs", voices, i, voices, o));
        }
        let device = unsafe { fdsp_bank_device(bank) };
        Ok(Self { bank, kind, voices, ring_frames, device, last_error: None, _marker: PhantomData })
    }

    /// The same without the typenum arity check: for banks far beyond the channel counts `Size` is meant for (65 536 voices),
    /// which are driven through `render_device` / `render_mix` only.  `tick` / `process` of such a bank report an arity error.
    pub fn new_unchecked_arity(kind: &str, voices: usize, ring_frames: usize, device: i32) -> Result<Self, String> {
        let kind = CString::new(kind).map_err(|e| e.to_string())?;
        let mut bank: *mut FdspBank = core::ptr::null_mut();
        check(unsafe { fdsp_bank_create_on(device as c_int, kind.as_ptr(), voices, ring_frames, &mut bank) })?;
        let device = unsafe { fdsp_bank_device(bank) };
        Ok(Self { bank, kind, voices, ring_frames, device, last_error: None, _marker: PhantomData })
    }

    /// `instances` x `reverb_stereo(room_size, time, damping)` (src/prelude.rs:1732-1762) through the dedicated lane-per-frame FDN kernel
    /// (one wave per instance, the 32 delay lines in registers) instead of the generic lane-per-voice `Feedback` kernel `from_graph` builds
    /// for the same type: bit-identical output, an order of magnitude faster.  Each instance has two inputs and two outputs.
    pub fn reverb_stereo(instances: usize, room_size: f64, time: f64, damping: f64, device: i32) -> Result<Self, String> {
        let mut bank: *mut FdspBank = core::ptr::null_mut();
        check(unsafe { fdsp_reverb_stereo_create_on(device as c_int, instances, room_size, time, damping, &mut bank) })?;
        Self::adopt(bank, "reverb_stereo", instances)
    }

    /// `instances` x `reverb4_stereo(room_size, time)` (src/prelude.rs:1873-1941: two 16-line Hadamard networks in series), same kernel family.
    pub fn reverb4_stereo(instances: usize, room_size: f64, time: f64, device: i32) -> Result<Self, String> {
        let mut bank: *mut FdspBank = core::ptr::null_mut();
        check(unsafe { fdsp_reverb4_stereo_create_on(device as c_int, instances, room_size, time, &mut bank) })?;
        Self::adopt(bank, "reverb4_stereo", instances)
    }

    /// `instances` x `reverb3_stereo(time, diffusion, lowpole_hz(cutoff))` (src/prelude.rs:1858-1871; `Reverb<F>`, src/reverb.rs:152-279) through its
    /// lane-per-frame kernel: bit-identical to the `Reverb3` node `from_graph` builds for the type, two orders of magnitude faster.
    pub fn reverb3_stereo(instances: usize, time: f64, diffusion: f64, lowpole_cutoff_hz: f32, device: i32) -> Result<Self, String> {
        let mut bank: *mut FdspBank = core::ptr::null_mut();
        check(unsafe { fdsp_reverb3_stereo_create_on(device as c_int, instances, time, diffusion, lowpole_cutoff_hz, &mut bank) })?;
        Self::adopt(bank, "reverb3_stereo", instances)
    }

    /// ... with a `FixedSvf` as the loop filter: `svf_mode` 0..8 = lowpass, highpass, bandpass, notch, peak, allpass, bell, lowshelf, highshelf
    /// (`reverb3_stereo(time, diffusion, highshelf_hz(5000.0, 1.0, db_amp(-1.0)))` of examples/keys.rs:134 is mode 8).
    pub fn reverb3_stereo_svf(instances: usize, time: f64, diffusion: f64, svf_mode: i32, cutoff_hz: f32, q: f32, gain: f32, device: i32) -> Result<Self, String> {
        let mut bank: *mut FdspBank = core::ptr::null_mut();
        check(unsafe { fdsp_reverb3_stereo_svf_create_on(device as c_int, instances, time, diffusion, svf_mode as c_int, cutoff_hz, q, gain, &mut bank) })?;
        Self::adopt(bank, "reverb3_stereo", instances)
    }

    /// `instances` x the Hadamard feedback delay network of the prelude's own example (src/prelude.rs:1323-1345):
    /// `split::<N>() >> fdn::<N, _>(stacki::<N, _, _>(|i| delay(delays[i]) >> fir(weights))) >> join::<N>()` with `inputs = outputs = 1`,
    /// `multisplit::<U2, _>` / `multijoin::<U2, _>` around it with 2.  `delays.len()` = N in 2, 4, 8, 16, 32; one to three FIR weights; every
    /// delay longer than 128 samples at the bank's sample rate.  Same kernel family as the reverbs: bit-identical to the `Feedback` graph
    /// `from_graph` would build for the type, two orders of magnitude faster.
    pub fn fdn(instances: usize, delays: &[f64], weights: &[f32], inputs: usize, outputs: usize, device: i32) -> Result<Self, String> {
        let mut bank: *mut FdspBank = core::ptr::null_mut();
        check(unsafe {
            fdsp_fdn_create_on(device as c_int, instances, delays.len() as c_int, delays.as_ptr(), weights.len() as c_int, weights.as_ptr(), inputs as c_int, outputs as c_int, &mut bank)
        })?;
        Self::adopt(bank, "fdn", instances)
    }

    fn adopt(bank: *mut FdspBank, kind: &str, voices: usize) -> Result<Self, String> {
        let device = unsafe { fdsp_bank_device(bank) };
        Ok(Self { bank, kind: CString::new(kind).unwrap(), voices, ring_frames: 0, device, last_error: None, _marker: PhantomData })
    }

    fn arity_ok(&self) -> bool {
        let (i, o) = unsafe { (fdsp_bank_inputs(self.bank) as usize, fdsp_bank_outputs(self.bank) as usize) };
        i * self.voices == NI::USIZE && o * self.voices == NO::USIZE
    }

    /// `voices` instances of the graph TYPE `X`: `core::any::type_name::<X>()` goes to the engine's front door
    /// (fdsp_graph_compile_rust), which builds the fused device kernel and remembers the parameters the type carries
    /// (filter modes, shape kinds).  Field values of `_voice` are NOT read: set them per voice with [`Self::set_param`].
    pub fn from_graph<X: AudioNode>(_voice: &An<X>, voices: usize) -> Result<Self, String> {
        Self::from_graph_with::<X>(voices, 0, -1, None, None)
    }

    pub fn from_graph_with<X: AudioNode>(voices: usize, ring_frames: usize, device: i32, hints: Option<&str>,
                                         functor_source: Option<&str>) -> Result<Self, String> {
        let type_name = core::any::type_name::<An<X>>();
        // a stable kind name per graph type
        let mut h = AttoHash::new(0x4849_5042);
        for b in type_name.bytes() {
            h = h.hash(b as u64);
        }
        let name = format!("rust_{:016x}", h.state());
        let cname = CString::new(name.clone()).unwrap();
        let ctype = CString::new(type_name).map_err(|e| e.to_string())?;
        let chints = hints.map(|s| CString::new(s).unwrap());
        let csrc = functor_source.map(|s| CString::new(s).unwrap());
        let rc = unsafe {
            fdsp_graph_compile_rust(cname.as_ptr(), ctype.as_ptr(), chints.as_ref().map_or(core::ptr::null(), |c| c.as_ptr()),
                                    csrc.as_ref().map_or(core::ptr::null(), |c| c.as_ptr()))
        };
        if rc < 0 {
            return Err(last_error());
        }
        Self::new(&name, voices, ring_frames, device)
    }

    pub fn voices(&self) -> usize { self.voices }
    pub fn device(&self) -> i32 { self.device }

    /// The error of a failed `tick` / `process` since the last call (those two cannot return one): `None` = all launches
    /// succeeded.  A failed launch has already zeroed its output block.
    pub fn take_error(&mut self) -> Option<String> { self.last_error.take() }

    /// One value per voice for the named slot (`"<path>:<field>"`, listed by `fdsp_kind_slot_name`): what
    /// `Setting::center(..)`, `.q(..)`, `Constant` values etc. are on a single node.
    pub fn set_param(&mut self, slot: &str, values: &[f32]) -> Result<(), String> {
        let c = CString::new(slot).map_err(|e| e.to_string())?;
        check(unsafe { fdsp_bank_set_param(self.bank, c.as_ptr(), values.as_ptr(), 0, values.len()) })
    }

    pub fn set_param_all(&mut self, slot: &str, value: f32) -> Result<(), String> {
        let c = CString::new(slot).map_err(|e| e.to_string())?;
        check(unsafe { fdsp_bank_set_param_all(self.bank, c.as_ptr(), value) })
    }

    /// `AudioNode::set_seed` per voice (`ping(false, AttoHash::new(seed))`).
    pub fn set_seeds(&mut self, seeds: &[u64]) -> Result<(), String> {
        check(unsafe { fdsp_bank_set_seed(self.bank, seeds.as_ptr(), 0, seeds.len()) })
    }

    /// `dry * multipass() & wet * node` around a reverb / network bank (src/audionode.rs:1842-1877 `Bus`, :1190-1228 `FrameMulScalar`), folded
    /// into the bank's render kernel: README.md:436's `multipass() & 0.2 * reverb_stereo(20.0, 2.0, 1.0)` is `set_bus(Some(1.0), 0.2)`;
    /// `None` for the dry side leaves `wet * node` alone.  Factors of 1.0 are the nodes a graph leaves out (x * 1.0 == x).
    pub fn set_bus(&mut self, dry: Option<f32>, wet: f32) -> Result<(), String> {
        check(unsafe { fdsp_bank_set_bus(self.bank, if dry.is_some() { 2 } else { 1 }, wet, dry.unwrap_or(1.0)) })
    }

    /// Which hiprtc compiles run-time compiled graphs in this process (`"linked: <path>"` for a Rust host; see include/fundsp_hip.h).
    pub fn jit_compiler() -> String {
        unsafe { std::ffi::CStr::from_ptr(fdsp_jit_compiler()) }.to_string_lossy().into_owned()
    }

    /// Tolerance mode (FDSP_MATH_FAST): FMA polynomials for feed-forward transcendentals, recurrences exact.
    pub fn set_fast_math(&mut self, on: bool) -> Result<(), String> {
        let c = CString::new("math").unwrap();
        check(unsafe { fdsp_bank_set_option(self.bank, c.as_ptr(), if on { FDSP_MATH_FAST } else { FDSP_MATH_EXACT }) })
    }

    /// Raw handle for device-resident rendering (`fdsp_bank_process`, `fdsp_mix_stereo`, `fdsp_mix_allreduce`).
    pub fn raw(&mut self) -> *mut FdspBank { self.bank }

    /// The big-block entry (the reference's `BigBlockAdapter::process_big`, src/audiounit.rs:510-528, without the 64-sample
    /// chopping: the kernel blocks internally exactly like `Wave::render`).  Device-resident I/O, voice-minor:
    /// `input` = `[inputs][frames][voices]`, `output` = `[outputs][frames][voices]` f32 in HBM; `stream` = a `hipStream_t` or null
    /// (the bank's own stream).  Asynchronous: [`Self::synchronize`] or the caller's stream order the read-back.
    /// This -- not `process()` -- is the throughput path: `process()` follows `AudioNode`'s contract on HOST buffers and moves
    /// `voices x (inputs + outputs) x 256 B` over PCIe per 64-frame block (16.7 MB per block at 65 536 mono voices).
    pub fn render_device(&mut self, frames: usize, input: DevicePtr<'_>, output: DevicePtrMut<'_>, stream: *mut c_void) -> Result<(), String> {
        let (i, o) = unsafe { (fdsp_bank_inputs(self.bank) as usize, fdsp_bank_outputs(self.bank) as usize) };
        if input.len < i * frames * self.voices || output.len < o * frames * self.voices {
            return Err(format!("render_device: buffers of {} / {} floats, the launch needs {} / {}", input.len, output.len,
                               i * frames * self.voices, o * frames * self.voices));
        }
        check(unsafe { fdsp_bank_process(self.bank, frames, input.ptr, output.ptr, FDSP_LAYOUT_VOICE_MINOR, 0, FDSP_MODE_PROCESS, stream) })
    }

    /// Render AND mix down in one launch (`fdsp_bank_process_mix`): what `voice >> pan(p)` per voice followed by the sum over
    /// the voices computes (src/pan.rs:50-76, src/audionode.rs:2406-2462), without the per-voice output ever existing in HBM.
    /// `pan = true`: mono graphs, every voice panned with its own position ([`Self::set_pan`]); `mix` = `[2][frames]`.
    /// `pan = false`: the sum of every output channel over the voices; `mix` = `[outputs][frames]`.
    /// Fixed summation order (include/fundsp_hip.h): bit-identical from launch to launch and across launch geometries.
    pub fn render_mix(&mut self, frames: usize, input: DevicePtr<'_>, mix: DevicePtrMut<'_>, pan: bool, stream: *mut c_void) -> Result<(), String> {
        let (i, o) = unsafe { (fdsp_bank_inputs(self.bank) as usize, fdsp_bank_outputs(self.bank) as usize) };
        let nm = if pan { 2 } else { o };
        if input.len < i * frames * self.voices || mix.len < nm * frames {
            return Err(format!("render_mix: buffers of {} / {} floats, the launch needs {} / {}", input.len, mix.len, i * frames * self.voices, nm * frames));
        }
        check(unsafe { fdsp_bank_process_mix(self.bank, frames, input.ptr, mix.ptr, if pan { FDSP_MIX_PAN } else { FDSP_MIX_SUM }, FDSP_MODE_PROCESS, stream) })
    }

    /// Pan position (-1 .. 1) per voice for `render_mix(.., pan = true, ..)`; every voice starts in the centre (`pan(0.0)`).
    pub fn set_pan(&mut self, pan: &[f32]) -> Result<(), String> {
        check(unsafe { fdsp_bank_set_pan(self.bank, pan.as_ptr(), 0, pan.len()) })
    }

    /// `AudioNode::allocate` for the mix path: size the bank's partial-mix buffer for launches of up to `frames` frames, so that
    /// `render_mix` never allocates (real-time loops, HIP graph capture).
    pub fn allocate_mix(&mut self, frames: usize) -> Result<(), String> { check(unsafe { fdsp_bank_mix_reserve(self.bank, frames) }) }

    /// Wait for the bank's own stream (renders launched with a null `stream`).
    pub fn synchronize(&mut self) -> Result<(), String> { check(unsafe { fdsp_bank_synchronize(self.bank) }) }
}

/// A borrowed run of f32 in DEVICE memory (hipMalloc'd by the host application): pointer + length in floats.  The shim never
/// dereferences it on the host; the length is what the launch-size check needs.  A generator's input is `DevicePtr::null()`.
#[derive(Clone, Copy)]
pub struct DevicePtr<'a> { ptr: *const f32, len: usize, _life: PhantomData<&'a f32> }
pub struct DevicePtrMut<'a> { ptr: *mut f32, len: usize, _life: PhantomData<&'a mut f32> }
impl<'a> DevicePtr<'a> {
    /// # Safety
    /// `ptr` must be device memory of at least `len` floats on the bank's device, alive for `'a` and not written while a launch reads it.
    pub unsafe fn new(ptr: *const f32, len: usize) -> Self { Self { ptr, len, _life: PhantomData } }
    pub fn null() -> Self { Self { ptr: core::ptr::null(), len: 0, _life: PhantomData } }
}
impl<'a> DevicePtrMut<'a> {
    /// # Safety
    /// `ptr` must be device memory of at least `len` floats on the bank's device, alive for `'a`, with no other reader or writer while a launch runs.
    pub unsafe fn new(ptr: *mut f32, len: usize) -> Self { Self { ptr, len, _life: PhantomData } }
}

impl<NI: Size<f32>, NO: Size<f32>> AudioNode for HipBank<NI, NO> {
    const ID: u64 = 0x4849_5042; // "HIPB"
    type Inputs = NI;
    type Outputs = NO;

    fn reset(&mut self) {
        // infallible in FunDSP (src/audionode.rs:52); a failed device call is kept for take_error()
        if unsafe { fdsp_bank_reset(self.bank) } != FDSP_OK {
            self.last_error = Some(last_error());
        }
    }

    fn set_sample_rate(&mut self, sample_rate: f64) {
        let rc = unsafe { fdsp_bank_set_sample_rate(self.bank, sample_rate) };
        debug_assert!(rc == FDSP_OK, "{}", last_error());
    }

    #[inline]
    fn tick(&mut self, input: &Frame<f32, Self::Inputs>) -> Frame<f32, Self::Outputs> {
        // one sample of every voice: planar rows of length 1 are [voice][channel] = the Frame's own order
        let mut out: Frame<f32, Self::Outputs> = Frame::default();
        if !self.arity_ok() {
            self.last_error = Some("tick: the bank was created with new_unchecked_arity; drive it through render_device / render_mix".into());
            return out;
        }
        let inp = if NI::USIZE > 0 { input.as_slice().as_ptr() } else { core::ptr::null() };
        let rc = unsafe { fdsp_bank_process_host(self.bank, 1, inp, out.as_mut_slice().as_mut_ptr(), FDSP_LAYOUT_PLANAR, 1, FDSP_MODE_TICK) };
        if rc != FDSP_OK {
            self.last_error = Some(last_error());
            out = Frame::default(); // silence, not stale samples
            debug_assert!(false, "{}", last_error());
        }
        out
    }

    fn process(&mut self, size: usize, input: &BufferRef, output: &mut BufferMut) {
        // BufferRef(&[F32x]) / BufferMut(&mut [F32x]): contiguous [channel][64] f32, 32-byte aligned, channel = voice-major
        if !self.arity_ok() {
            self.last_error = Some("process: the bank was created with new_unchecked_arity; drive it through render_device / render_mix".into());
            for ch in 0..NO::USIZE {
                output.channel_f32_mut(ch)[..size].fill(0.0);
            }
            return;
        }
        let inp = if NI::USIZE > 0 { input.channel_f32(0).as_ptr() } else { core::ptr::null() };
        let out = output.channel_f32_mut(0).as_mut_ptr();
        let rc = unsafe { fdsp_bank_process_host(self.bank, size, inp, out, FDSP_LAYOUT_PLANAR, MAX_BUFFER_SIZE, FDSP_MODE_PROCESS) };
        // process() is infallible in FunDSP: an error here is a programming error (wrong arity, lost device).  The block is
        // silenced (never left holding stale samples) and the error kept for take_error(), in release builds too.
        if rc != FDSP_OK {
            self.last_error = Some(last_error());
            for ch in 0..NO::USIZE {
                output.channel_f32_mut(ch)[..size].fill(0.0);
            }
            debug_assert!(false, "{}", last_error());
        }
    }

    fn set_hash(&mut self, hash: u64) {
        // ping() reaches a leaf with ONE hash; a bank is V leaves: every voice gets it (a bank that should de-correlate
        // its voices calls set_seeds with one seed per voice instead, as the parity tests do)
        let seeds = vec![hash; self.voices];
        if unsafe { fdsp_bank_set_seed(self.bank, seeds.as_ptr(), 0, self.voices) } != FDSP_OK {
            self.last_error = Some(last_error()); // set_hash is infallible (src/audionode.rs:136): kept for take_error()
        }
    }

    fn route(&mut self, input: &SignalFrame, _frequency: f64) -> SignalFrame {
        Routing::Arbitrary(0.0).route(input, self.outputs())
    }
}

impl<NI: Size<f32>, NO: Size<f32>> Clone for HipBank<NI, NO> {
    /// FunDSP nodes are `Clone` (Net and Sequencer clone their units): a second bank that continues exactly where this one
    /// stands.  `fdsp_bank_clone` copies everything a later `set_param` / `reset` / `process` depends on -- the slots, the
    /// delay rings, the SAMPLE RATE and the arithmetic mode (a clone rebuilt from the slot words alone would re-derive its
    /// coefficients at the 44.1 kHz default on the next parameter change), launch options, scheduler events and clock.
    /// `Clone::clone` cannot fail, so a failed copy (device lost, out of memory) panics with the engine's message.
    fn clone(&self) -> Self {
        let mut bank: *mut FdspBank = core::ptr::null_mut();
        let rc = unsafe { fdsp_bank_clone(self.bank, &mut bank) };
        assert!(rc == FDSP_OK && !bank.is_null(), "fdsp_bank_clone: {}", last_error());
        Self { bank, kind: self.kind.clone(), voices: self.voices, ring_frames: self.ring_frames, device: self.device,
               last_error: None, _marker: PhantomData }
    }
}

impl<NI: Size<f32>, NO: Size<f32>> Drop for HipBank<NI, NO> {
    fn drop(&mut self) {
        unsafe { fdsp_bank_destroy(self.bank) }
    }
}

/*
 * fundsp_hip.h -- C ABI of the MI355X (gfx950) voice-bank engine for FunDSP leaf DSP nodes.
 *
 * Drop-in boundary.  FunDSP has no FFI of its own; the operator boundary this library replaces is the Rust
 * trait method
 *     fn process(&mut self, size: usize, input: &BufferRef, output: &mut BufferMut)   src/audionode.rs:85
 * (dyn twin AudioUnit::process, src/audiounit.rs:45) together with the lifecycle methods a node honours:
 *     reset :52, set_sample_rate :68, tick :79, set(Setting) :130, set_hash :136, ping :156, set_seed :366.
 * A *bank* is V independent voice instances of one compiled voice graph evaluated in lock-step, one wavefront
 * lane per voice.  INTEGRATION.md shows the `extern "C"` block + `impl AudioNode for HipBank` a FunDSP
 * maintainer would add on the Rust side.
 *
 * Conventions
 *  - All functions return 0 on success or a negative FDSP_E* code; fdsp_last_error() gives the message.
 *    Nothing unwinds across the ABI (AudioNode::process itself is infallible, audionode.rs:85).
 *  - Plain pointers and sizes only.  `d_*` pointers are device (HBM) pointers, `h_*` are host pointers.
 *  - A bank handle is not thread-safe but is thread-movable (AudioNode: Send, `process(&mut self)`).
 *  - Sample data is IEEE f32; internal state is f32 (prelude32, F = f32).
 *
 * Buffer layouts (`layout` argument)
 *  - FDSP_LAYOUT_VOICE_MINOR: element (channel c, frame t, voice v) at  (c*frames + t)*voices + v.
 *      Device-native: lane-consecutive addresses, no staging.
 *  - FDSP_LAYOUT_PLANAR:      element (voice v, channel c, frame t) at  (v*channels + c)*frame_stride + t.
 *      With frame_stride = 64 this is exactly an array of the reference's BufferArray/BufferRef blocks
 *      ([channel][8 x f32x8], src/buffer.rs:8-12, 356-365), one per voice; the engine transposes 64x64 tiles
 *      through LDS so HBM access stays coalesced.
 *
 * Render modes (`mode` argument)
 *  - FDSP_MODE_PROCESS: AudioNode::process semantics.  `frames` is chopped into <=64-sample blocks exactly
 *      like Wave::render (src/wave.rs:452-464); nodes that override `process` with different arithmetic
 *      (Sine: unwrapped in-block phase + f32x8 sin, src/oscillator.rs:74-86) use it for the full 8-sample
 *      SIMD items of each block and `tick` for the remainder.
 *  - FDSP_MODE_TICK: every sample through AudioNode::tick (src/audionode.rs:79).
 */
#ifndef FUNDSP_HIP_H
#define FUNDSP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDSP_OK 0
#define FDSP_EINVAL (-1)   /* bad argument (unknown kind / parameter name / range / layout) */
#define FDSP_ENOMEM (-2)   /* device allocation failed */
#define FDSP_EDEVICE (-3)  /* HIP runtime error or no gfx950 device */
#define FDSP_ENOTSUP (-4)  /* the request is valid but this kind / bank has no kernel for it (the message says what to call instead) */

#define FDSP_LAYOUT_VOICE_MINOR 0
#define FDSP_LAYOUT_PLANAR 1

#define FDSP_MODE_PROCESS 0
#define FDSP_MODE_TICK 1

#define FDSP_MAX_BUFFER_SIZE 64 /* MAX_BUFFER_SIZE, src/lib.rs:48 */
#define FDSP_DEFAULT_SR 44100.0 /* DEFAULT_SR, src/lib.rs:42 */

/* SvfMode values for the "mode" parameter of fixed_svf / svf3 / svf4 (src/svf.rs:281-742) */
enum { FDSP_SVF_LOWPASS = 0, FDSP_SVF_HIGHPASS, FDSP_SVF_BANDPASS, FDSP_SVF_NOTCH, FDSP_SVF_PEAK,
       FDSP_SVF_ALLPASS, FDSP_SVF_BELL, FDSP_SVF_LOWSHELF, FDSP_SVF_HIGHSHELF };
/* BiquadCoefs constructors (src/biquad.rs:27-116) */
enum { FDSP_BQ_BUTTER_LOWPASS = 0, FDSP_BQ_RESONATOR, FDSP_BQ_LOWPASS, FDSP_BQ_HIGHPASS, FDSP_BQ_BELL };

typedef struct fdsp_bank fdsp_bank;

const char* fdsp_last_error(void);

/* ---- voice-graph kinds compiled into the library -------------------------------------------------------
 * Leaves (each replaces that node's AudioNode::process):
 *   "sine" oscillator.rs:21  "noise" noise.rs:173  "fixed_svf" svf.rs:861  "svf3"/"svf4" svf.rs:748
 *   "biquad" biquad.rs:136  "biquad_bank" biquad_bank.rs:14 (voice = instance*8 + lane)
 *   "butterpass_hz"/"butterpass" biquad.rs:227  "resonator_hz"/"resonator" biquad.rs:310
 *   "moog_hz"/"moog" moog.rs:17  "fir2"/"fir3" fir.rs:14  "tick" delay.rs:19  "pass" audionode.rs:408
 * Fused BASELINE graphs (combinator semantics of audionode.rs Pipe/Unop/Constant are fused in registers):
 *   "sine_hz"            constant(f) >> sine()                                         prelude.rs:349
 *   "sine_hz_lowpass_hz" sine_hz(f) >> lowpass_hz(fc, q)                    (BASELINE config 1)
 *   "noise_biquad"       noise() >> biquad(..)        one BiquadBank lane  (BASELINE config 2)
 *   "fm_svf"             sine_hz(f) * f * m + f >> sine() >> lowpass_hz(fc, q) (BASELINE config 3)
 *   "saw_moog_adsr_pan"  ((dc(f) >> saw() | dc(fc) | dc(q)) >> moog()) * adsr_live(a,d,s,r) >> pan(p)
 *                        1 input (gate), 2 outputs                            (BASELINE config 4 voice)
 * Delay lines (need fdsp_bank_create_ring): "delay" delay.rs:72, "tap" :148, "tap_linear" :386,
 *   "allnest_delay"/"allnest_tick"/"allnest_pass" :294 (AllNest around Delay / Tick / Pass)
 * Shapers and oscillators: "shape" shape.rs:205 (per-voice shape kind), "ramp"/"poly_saw"/"poly_square"/"poly_pulse"
 *   oscillator.rs:441-760, "rossler"/"lorenz" :323-435; nonlinear biquads "fbiquad_hz"/"dbiquad_hz"/"fbiquad3"/
 *   "fbiquad4"/"dbiquad3"/"dbiquad4" biquad.rs:494-920
 * More leaves: "saw"/"square"/"triangle" wavetable.rs:249 (need fdsp_wavetable_build/upload first),
 *   "adsr_live" adsr.rs:21 + envelope.rs:185, "pan" pan.rs:26
 */
int fdsp_kind_count(void);
const char* fdsp_kind_name(int kind);
int fdsp_kind_by_name(const char* name); /* -1 if unknown */
/* Engine options.  "pipe_split" (default 1): render Pipe-chain graphs of the ahead-of-time kinds in the voice-minor
 * layout with the multi-wave pipeline split (the chain stages of each 64-voice group cut into 2 or 3 consecutive
 * segments that run in 2 or 3 waves sharing a SIMD; identical samples, better issue-slot utilisation at one
 * voice-wave per SIMD).  1 = best plan, 2 / 3 = exactly that many stages (if the graph allows), 0 = single-wave kernel.
 * With 1 the launch length decides as well: a chain worth cutting takes the pipeline from ONE 64-frame block on (a light graph such
 * as noise >> biquad from four blocks on, planar launches from 16 frames on), small banks of oscillator chains the time-split kernels
 * from one whole block on, anything shorter the single-wave kernel; 2 / 3 force the pipeline at any length.  Same samples always. */
int fdsp_set_option(const char* name, int value);
/* "math" (default FDSP_MATH_EXACT): the arithmetic of banks created afterwards.
 *   FDSP_MATH_EXACT  every node evaluates the reference's own algorithm operation for operation (no contraction, the
 *                    restated libm / wide functions): bit-identical to the CPU oracle.  The headline mode.
 *   FDSP_MATH_FAST   tolerance mode: state recurrences (phases, filter states, envelopes) stay operation for operation,
 *                    but feed-forward transcendental evaluations may use the engine's own forms -- today the f32x8
 *                    sine of Sine::process (FMA polynomial, 13 instead of 28 operations, within 1.2e-7 of it) and the
 *                    saturating tanh of Moog::process (hardware exp2 / reciprocal, odd polynomial below |x| = 0.25;
 *                    within 2.3e-7 absolute / 6.3e-7 relative over all f32; the BASELINE config-4 voice within 1e-4 of the exact mode,
 *                    measured 4e-7).  Stated tolerance of the sine,
 *                    measured in tests/test_gpu_math_fast.py on the BASELINE config-3 FM voices against the exact mode:
 *                    <= 1e-4 absolute over the reference's own check window (441 samples, tests/test_basic.rs:21-47);
 *                    <= 1e-3 max, <= 5e-5 rms over a full second (measured 2.6e-4 / 1.2e-5) -- an FM patch integrates
 *                    every last-bit difference of the modulator into the carrier phase: the reference's own tick and
 *                    process paths drift ~0.2 apart over the same second.  Kinds without such a node render exactly.
 * Per bank: fdsp_bank_set_option(bank, "math", v) / fdsp_bank_get_option(bank, "math");
 * fdsp_bank_get_option(bank, "math_has_fast_variant") tells whether FAST changes anything for the bank's kind. */
#define FDSP_MATH_EXACT 0
#define FDSP_MATH_FAST 1
int fdsp_bank_set_option(fdsp_bank* bank, const char* name, int value);
int fdsp_bank_get_option(const fdsp_bank* bank, const char* name);
/* The launch options -- "pipe_split", "time_split" (default 1: banks of <= 2 voice groups per CU of eligible graphs take
 * the time-split kernel), "fdn_kernel", "timing" -- exist per bank as well: fdsp_bank_set_option(bank, name, v) overrides
 * the process-wide value of fdsp_set_option for that bank (-1 = follow it again).  They are resolved per launch on the
 * calling thread, so hosts driving different banks from different threads do not see each other's choices.
 * "timing" (default 1): every render records a HIP event pair around the kernel (fdsp_bank_last_kernel_ms); 0 drops
 * the pair -- a real-time host rendering one 64-frame block per call saves two event records per launch.
 * fdsp_bank_get_option(bank, "last_kernel") (read-only): the kernel family the most recent render launch took --
 * 1 single-wave, 2 pipeline, 3 planar pipeline, 4 time-split, 5 voice scheduler, 6 / 7 reverb lane-per-frame / -line,
 * 8 the chain of waves that renders a wide sum of generators (sumi / busi of >= 8 oscillators) on a small bank. */
/* fdsp_bank_get_option(bank, "has_fused_mix") (read-only): 1 if fdsp_bank_process_mix has kernels for the bank's kind. */
/* "host_zero_copy_max" (default 262144): fdsp_bank_process_host calls moving at most this many floats per direction
 * let the kernel read/write pinned host memory directly instead of staging through HBM (lower per-block latency). */
/* "fdn_kernel" (default 0): reverb banks render with one lane per FRAME (0) or one lane per DELAY LINE (1); identical
 * samples, the former is the faster formulation. */

/* ---- run-time compiled voice graphs (graph -> kernel compiler) -------------------------------------------
 * `type_expr` is the graph's combinator TYPE, exactly what FunDSP's operators build (src/combinator.rs:289-488),
 * spelled with the engine's node templates (fundsp_amd/csrc/fd_nodes.hpp), e.g.
 *     sine_hz(f) >> lowpass_hz(fc, q)   ==  "Pipe<Pipe<Constant<1>,Sine>,FixedSvf>"
 *     (noise() | dc(fc) | dc(q)) >> moog()  ==  "Pipe<Stack<Stack<Noise,Constant<1>>,Constant<1>>,Moog<3>>"
 * The graph is compiled with hiprtc from the library's own headers (same flags as the ahead-of-time kinds) into one
 * fused kernel set and registered under `name`; fdsp_bank_create(name, ..) then works as for built-in kinds.
 * Returns the kind index (>= 0) or a negative error (fdsp_last_error() holds the compiler log).
 * fdsp_graph_check() only compiles (no device needed) -- a syntax / arity check of a type expression. */
int fdsp_graph_compile(const char* name, const char* type_expr);
/* Same, with C++ source compiled in front of the graph type (inside namespace fd): definitions the type expression
 * refers to -- typically the functor that stands in for the Rust closure of envelope(|t| ...) / lfo(|t| ...), see
 * Envelope<FN> in fundsp_amd/csrc/fd_nodes.hpp for the functor contract (OUT, visit, init, eval). */
int fdsp_graph_compile_src(const char* name, const char* type_expr, const char* source);
/* Which compiler compiles these graphs.  The promise of a run-time compiled kind -- the same samples as the same graph compiled ahead of time -- needs
 * the headers, the flags AND the compiler of the library's own build.  A host process may already hold another ROCm's libhiprtc / libamd_comgr under
 * the same sonames (a PyTorch wheel bundles ROCm 7.0's next to this image's 7.2 and loads them first); the dynamic loader then hands this library THAT
 * copy.  The library detects it and loads the libhiprtc of the ROCm it was built against into a link-map namespace of its own (dlmopen), where it finds
 * its own comgr.  Returns a description: "linked: <path>" (the process's hiprtc is the right one) or "isolated: <path> (the process's own is <path>)".
 * Environment: FDSP_HIPRTC=linked keeps the process's copy whatever it is, FDSP_HIPRTC=<path of a libhiprtc.so> names another one;
 * FDSP_JIT_DUMP=<directory> writes the generated source and code object of every compiled module there.  Why it matters: the bundled 7.0 compiler
 * miscompiles one kernel variant of `(mls() ^ impulse()) + c` (DESIGN.md section 0, round 6 "third part, 4"; tests/test_gpu_jit_compiler.py). */
const char* fdsp_jit_compiler(void);
int fdsp_graph_check(const char* type_expr);
/* The Rust side of the compiler: hand over `core::any::type_name::<X>()` of the graph `An<X>` as it is, e.g.
 *   fundsp::combinator::An<fundsp::audionode::Pipe<fundsp::audionode::Pipe<fundsp::audionode::Constant<typenum::uint::
 *   UInt<typenum::uint::UTerm, typenum::bit::B1>>, fundsp::oscillator::Sine<f32>>, fundsp::svf::FixedSvf<f32,
 *   fundsp::svf::LowpassMode<f32>>>>
 * fdsp_rust_type_to_expr rewrites it with the engine's templates ("Pipe<Pipe<Constant<1>,Sine>,FixedSvf>") and lists
 * the parameters the Rust TYPE carries as "slot=value" lines ("1:mode=0": filter modes, shape kinds, biquad modes;
 * combinator.rs:178-488, audionode.rs:850,1232,1375,1496).  fdsp_graph_compile_rust compiles the result under `name`
 * (like fdsp_graph_compile_src; `source` = functor definitions for closures, or NULL) and remembers those presets:
 * every bank created from the kind gets them applied.  Field VALUES (frequencies, Q ..) are not in a type: set them with
 * fdsp_bank_set_param as for any kind.  `hints` (or NULL) supplies what neither the type nor a slot carries, consumed
 * in the order the nodes appear in the type:
 *   "wavesynth=saw,square;meter=peak,rms;envelope=EnvExp;envelope_in=MyFn;map=MidSide;shape_fn=SoftFold" */
int fdsp_rust_type_to_expr(const char* rust_type_name, const char* hints, char* out_expr, size_t expr_cap,
                           char* out_presets, size_t presets_cap);
int fdsp_graph_compile_rust(const char* name, const char* rust_type_name, const char* hints, const char* source);
/* Host-only introspection of a kind (no device needed): arity and the named per-voice slots. */
int fdsp_kind_inputs(int kind);
int fdsp_kind_outputs(int kind);
int fdsp_kind_slot_count(int kind);
const char* fdsp_kind_slot_name(int kind, int slot);
int fdsp_kind_slot_kind(int kind, int slot);

/* ---- lifecycle (mirrors constructors / AudioNode::{set_sample_rate,reset,set_seed}) --------------------- */
/* Creates a bank of `voices` instances on the current HIP device.  State after creation equals the
 * reference constructor: DEFAULT_SR, default parameters, combinator construction-time ping (audionode.rs:871-876). */
int fdsp_bank_create(const char* kind, size_t voices, fdsp_bank** out);
/* Kinds that contain delay lines ("delay", "tap", "tap_linear", "allnest_delay": src/delay.rs) keep their rings in HBM,
 * laid out [ring node][position][voice].  `ring_frames` is the capacity (positions) of every ring of the graph and must
 * cover the longest delay at the highest sample rate that will be set: Delay needs round(time*sr)+1, Tap
 * next_pow2(ceil(max_delay*sr)+11), TapLinear next_pow2(ceil(max_delay*sr)+2) (delay.rs:108-110,204-206,440-442). */
int fdsp_bank_create_ring(const char* kind, size_t voices, size_t ring_frames, fdsp_bank** out);
/* reverb_stereo(room_size, time, damping) (src/prelude.rs:1732-1762): bank of `instances` independent 32-line FDN
 * reverbs, 2 inputs / 2 outputs each, all with the same parameters.  Default mapping: one wave per instance, one lane
 * per FRAME of a 64-sample block (the 32 lines in 32 registers, the Hadamard as register butterflies; option
 * "fdn_kernel" = 1 selects the lane-per-delay-line formulation, identical samples); delay rings in HBM; flushes f32
 * denormals like the reference does after Feedback::new (src/feedback.rs:96, src/denormal.rs:18).  The handle works
 * with set_sample_rate / reset / process / clone / destroy.  Layouts: FDSP_LAYOUT_PLANAR ([instance][channel][frame_stride]) is what a
 * lane-per-frame kernel reads and writes in 256-byte runs; FDSP_LAYOUT_VOICE_MINOR buffers of banks with 64 instances or more go
 * through a planar staging copy owned by the bank (one tiled transpose in, one out: + 8 bytes per channel and instance-frame; it grows
 * with the longest launch, outside stream captures only -- a captured launch that finds it too small gathers directly), smaller banks
 * gather them directly. */
int fdsp_reverb_stereo_create(size_t instances, double room_size, double time, double damping, fdsp_bank** out);
/* reverb4_stereo(room_size, time) (src/prelude.rs:1873-1941): TWO 16-line Hadamard networks in series --
 * multisplit::<U2,U8>() >> fdn(16 x delay >> fir3) >> multijoin::<U2,U8>() >> multisplit::<U2,U8>() >> fdn(16 x delay >> fir3)
 * >> sumf::<U16>(pan) * dc((1/4, 1/4)) -- through the same lane-per-frame kernel (both networks' ring reads, FIR outputs and feedback
 * are known at the head of a block; 272 B per instance-frame as well).  FDSP_MODE_PROCESS / FDSP_MODE_TICK differ in MultiJoin's
 * arithmetic exactly like the reference (src/audionode.rs:697-720).  Same handle semantics as reverb_stereo banks; "fdn_kernel" is ignored. */
int fdsp_reverb4_stereo_create(size_t instances, double room_size, double time, fdsp_bank** out);
/* The generic Hadamard feedback delay network as the prelude documents it (src/prelude.rs:1323-1345, "Mono Reverb" :1334):
 *     split::<N>() >> fdn::<N, _>(stacki::<N, _, _>(|i| delay(delays[i]) >> fir(weights))) >> join::<N>()
 * `instances` independent networks of `lines` = N delay lines (2, 4, 8, 16 or 32), Delay::new(delays[i]) seconds each
 * (src/delay.rs:82-113), every line followed by the same Fir of `taps` = 1..3 weights (src/fir.rs:14-70), Feedback with FrameHadamard
 * around them (src/feedback.rs:35-57,108-146).  `inputs` = 1 puts split::<N>() in front (src/audionode.rs:527-568), 2
 * multisplit::<U2, N/2>() (:571-613: line k takes channel k % 2); `outputs` = 1 puts join::<N>() behind (:617-660), 2
 * multijoin::<U2, N/2>() (:668-730; FDSP_MODE_PROCESS scales every term by 1/n and adds, FDSP_MODE_TICK adds and divides, like the
 * reference's two executors).  Rendered by the lane-per-frame kernel of the reverbs (one wave per instance, the lines in registers, ring
 * rows as 256-byte runs: 8 * lines + 4 * (inputs + outputs) bytes per instance-frame), which needs every delay to exceed two blocks
 * (128 samples) at the bank's sample rate -- FDSP_EINVAL otherwise; such a graph still renders lane-per-voice through
 * fdsp_graph_compile.  Flushes f32 denormals like every graph with a Feedback node.  Handle semantics of the reverb banks. */
/* reverb3_stereo(time, diffusion, lowpole_hz(cutoff)) (src/prelude.rs:1858-1871): the allpass-loop reverb Reverb<F> of src/reverb.rs:152-279
 * with the documented loop filter, a one-pole lowpass (src/filter.rs:19-66).  `instances` independent reverbs, 2 inputs / 2 outputs each, all
 * with the same parameters.  One wave per instance, one lane per FRAME of a 64-sample block: all 76 delay lines of the structure (4 input
 * diffusers, 8 x (4 + 4) Schroeder allpasses, 8 block delays) are longer than two blocks, so their reads are known at the head of a block and
 * every allpass is feed-forward inside it; the sixteen loop filters -- the only recurrences in time -- run on eight lanes between the two
 * allpass layers.  624 B per instance-frame (76 ring reads + 76 ring writes + 2 in + 2 out).  The graph the run-time compiler builds for the
 * same node renders the same samples one lane per voice, two to three orders of magnitude slower.  Like the reference: no process
 * override (FDSP_MODE_PROCESS == FDSP_MODE_TICK), IEEE denormals kept (no Feedback node), reset() and set_sample_rate() leave the input
 * diffusers alone (src/reverb.rs:211-238), a change of rate empties the lines but keeps every allpass's pending sample, the feedback sample
 * and the filters' values.  Needs >= 14.2 kHz (every delay longer than 128 samples).  Handle semantics and layouts of the reverb banks. */
int fdsp_reverb3_stereo_create(size_t instances, double time, double diffusion, float lowpole_cutoff_hz, fdsp_bank** out);
/* The same reverb with a FixedSvf as the loop filter -- reverb3_stereo(time, diffusion, highshelf_hz(5000.0, 1.0, db_amp(-1.0))) is what the
 * reference's examples put there (examples/keys.rs:134): svf_mode = FDSP_SVF_LOWPASS .. FDSP_SVF_HIGHSHELF (the modes of fdsp_svf_coefs),
 * `gain` an amplitude (bell / shelves).  Same kernel; the sixteen filters' recurrences (src/svf.rs:995-1006) run on the eight serial lanes. */
int fdsp_reverb3_stereo_svf_create(size_t instances, double time, double diffusion, int svf_mode, float cutoff_hz, float q, float gain, fdsp_bank** out);
int fdsp_fdn_create(size_t instances, int lines, const double* delays, int taps, const float* weights, int inputs, int outputs, fdsp_bank** out);
/* A gain and a dry bus around a reverb / network bank -- the shape the reference's documentation gives its reverbs:
 *     multipass() & 0.2 * reverb_stereo(20.0, 2.0, 1.0)                       README.md:436 ("to add 20% reverb to a stereo signal")
 *     0.2 * reverb_stereo(10.0, 1.0, 0.5) & multipass()                       src/wave.rs:514
 *     wet * reverb_stereo(10.0, time) & (1.0 - wet) * multipass()             CHANGES.md:203
 * `wet * node` is Unop<X, FrameMulScalar> (src/combinator.rs:477-488, src/audionode.rs:1190-1228: every output sample times the scalar),
 * `x & y` is Bus<X, Y> (src/audionode.rs:1842-1877: both sides take the node's input and the outputs are added, one addition per sample in
 * tick :1862-1866 and in process :1868-1877), multipass() hands its input on (:373-403).  The render kernels fold these nodes into their
 * epilogue (the block's input frames are still in registers there; no second pass over the output, no buffer for the wet signal):
 *     FDSP_BUS_NONE     out = node(in)                          (the bank as created)
 *     FDSP_BUS_WET      out = wet * node(in)                    `wet * node`; `dry` ignored
 *     FDSP_BUS_DRY_WET  out = dry * in + wet * node(in)         `dry * multipass() & wet * node` in either order of the `&` (f32 addition commutes);
 *                                                               needs inputs == outputs (Bus: both sides have the same arities)
 * one f32 rounding per multiplication and per addition, as the reference's three nodes; pass 1.0 for a factor the graph does not have (x * 1.0 is x
 * bit for bit, so `multipass() & 0.2 * node` is (FDSP_BUS_DRY_WET, 0.2, 1.0) and `multipass() & node` (src/net.rs:681) is (.., 1.0, 1.0)).  A host-side
 * setting: takes effect at the next launch, survives reset / set_sample_rate, travels with fdsp_bank_clone.  Banks of the reverb / network
 * constructors above only (FDSP_ENOTSUP otherwise: a compiled graph carries its bus in its type).  The nodes' pings are no concern of the bank: none
 * of these nodes keeps hashed state (a generator in FRONT of the bus is seeded by the host from the whole graph's construction hash). */
#define FDSP_BUS_NONE 0
#define FDSP_BUS_WET 1
#define FDSP_BUS_DRY_WET 2
int fdsp_bank_set_bus(fdsp_bank* bank, int mode, float wet, float dry);
int fdsp_bank_get_bus(const fdsp_bank* bank, int* mode, float* wet, float* dry);
/* Several GPUs from one process.  A bank lives on ONE device, fixed at creation: the `_on` constructors take the HIP
 * device index (-1 = the calling thread's current device, which is what the constructors above use).  Every entry point
 * that takes a bank makes the bank's device current for its own duration and restores the caller's, so a host thread
 * may interleave banks of different GPUs freely; a bank is still single-threaded (AudioNode::process(&mut self)),
 * different banks may be driven from different threads.  Shared wavetables / waves are kept per device and installed
 * on every device that has banks.  `ring_frames` = 0 for kinds without delay lines. */
int fdsp_device_count(void);
int fdsp_bank_create_on(int device, const char* kind, size_t voices, size_t ring_frames, fdsp_bank** out);
int fdsp_reverb_stereo_create_on(int device, size_t instances, double room_size, double time, double damping, fdsp_bank** out);
int fdsp_reverb4_stereo_create_on(int device, size_t instances, double room_size, double time, fdsp_bank** out);
int fdsp_reverb3_stereo_create_on(int device, size_t instances, double time, double diffusion, float lowpole_cutoff_hz, fdsp_bank** out);
int fdsp_reverb3_stereo_svf_create_on(int device, size_t instances, double time, double diffusion, int svf_mode, float cutoff_hz, float q, float gain, fdsp_bank** out);
int fdsp_fdn_create_on(int device, size_t instances, int lines, const double* delays, int taps, const float* weights, int inputs, int outputs, fdsp_bank** out);
int fdsp_bank_device(const fdsp_bank* bank);
void fdsp_bank_destroy(fdsp_bank* bank);
/* `Clone` (every AudioNode is Clone, src/audionode.rs:35; Net and Sequencer clone their units): a new bank of the same
 * kind on the same device that continues exactly where `bank` stands -- all slots (parameters, coefficients, state), the
 * delay rings, the sample rate, the arithmetic mode and launch options, the scheduler's events and clock, a reverb's
 * line state.  Rendering the clone and the original with the same input gives the same samples. */
int fdsp_bank_clone(const fdsp_bank* bank, fdsp_bank** out);
int fdsp_bank_inputs(const fdsp_bank* bank);   /* AudioNode::Inputs  */
int fdsp_bank_outputs(const fdsp_bank* bank);  /* AudioNode::Outputs */
size_t fdsp_bank_voices(const fdsp_bank* bank);
int fdsp_bank_set_sample_rate(fdsp_bank* bank, double sample_rate); /* audionode.rs:68 */
int fdsp_bank_reset(fdsp_bank* bank);                               /* audionode.rs:52 */
/* AudioNode::set_seed (audionode.rs:366-368) per voice: ping(false, AttoHash::new(seed[i])) for voices
 * first..first+count.  h_seeds == NULL re-applies the construction-time hash to that range. */
int fdsp_bank_set_seed(fdsp_bank* bank, const uint64_t* h_seeds, size_t first, size_t count);

/* ---- parameters (mirrors AudioNode::set(Setting), setting.rs:14-72) ------------------------------------
 * Every per-voice field of the graph is a named slot "<path>:<field>[index]": <path> = child indices from the
 * root joined by '.', e.g. in "fm_svf" the filter cutoff is "1:cutoff" and the modulator frequency constant
 * is "0.0.0.0.0.0:value".  Setting a parameter re-derives dependent coefficients immediately, like the
 * reference's setters (e.g. FixedSvf::set_cutoff_q svf.rs:944-948). */
int fdsp_bank_slot_count(const fdsp_bank* bank);
const char* fdsp_bank_slot_name(const fdsp_bank* bank, int slot);
int fdsp_bank_slot_kind(const fdsp_bank* bank, int slot); /* 0 = parameter, 1 = derived coefficient, 2 = state */
int fdsp_bank_set_param(fdsp_bank* bank, const char* name, const float* h_values, size_t first, size_t count);
/* One value for every voice (Shared::set_value of a variable all voices watch, src/shared.rs:98-101): filled ON THE DEVICE in stream order behind
 * the last render -- no host copy, no host wait.  A render on the bank's own stream simply follows it; a render on a caller's stream waits for
 * it on the host as for every setter; a caller's stream that is being CAPTURED cannot wait, so a capture started while such a fill is still
 * queued is refused (FDSP_EDEVICE): call fdsp_bank_synchronize(bank) before capturing. */
int fdsp_bank_set_param_all(fdsp_bank* bank, const char* name, float value);
int fdsp_bank_set_param_u64(fdsp_bank* bank, const char* name, const uint64_t* h_values, size_t first, size_t count);
int fdsp_bank_get_slot(fdsp_bank* bank, const char* name, float* h_values, size_t first, size_t count);
/* Full per-voice snapshot, [slot_count][voices] f32 words (FunDSP nodes are Clone: audionode.rs:29). */
int fdsp_bank_get_state(fdsp_bank* bank, float* h_slots);
int fdsp_bank_set_state(fdsp_bank* bank, const float* h_slots);

/* ---- the hot path --------------------------------------------------------------------------------------
 * Render `frames` samples of every voice.  d_in may be NULL for generators.  `frame_stride` is only used by
 * FDSP_LAYOUT_PLANAR (row length in samples, >= frames; 64 for BufferArray blocks).  `stream` is a
 * hipStream_t (NULL = the bank's own stream); the call is asynchronous with respect to the host.
 * Ordering: a render on a caller's stream starts after the bank's pending parameter / lifecycle work, and later
 * setters, reset, set_seed, get/set_state wait for that render (its completion event) -- in either direction nothing
 * overtakes.  A caller's stream that is being CAPTURED into a HIP graph is supported: the launch is recorded without
 * any host-side synchronisation or event, so a real-time host can capture its block-by-block loop once and replay it
 * (tests/test_gpu_streams.py; 65 536 voices x 64 frames: 20.8 -> 13.3 us per block).  One refusal: a capture that would start behind a
 * still-queued fdsp_bank_set_param_all (the one setter that does not wait, see its comment) returns FDSP_EDEVICE and fdsp_last_error()
 * names the remedy -- fdsp_bank_synchronize(bank) before capturing.  Nothing compiles inside a render: run-time compiled kinds build
 * what a bank needs when it is created and when fdsp_bank_set_option(bank, "math", ..) switches its arithmetic.
 * Input values: every IEEE value is accepted and treated like the reference treats it (tests/test_gpu_specials.py).
 * One input sets the COST of a sample rather than its value: the speed input of Resample<X> (resample.rs:281-303) ticks
 * the enclosed generator `speed` times per output sample, on the device as in the reference -- a speed of 1e9 is a
 * billion inner ticks in one lane, i.e. a kernel that does not return in useful time.  Bound it on the host. */
int fdsp_bank_process(fdsp_bank* bank, size_t frames, const float* d_in, float* d_out, int layout,
                      size_t frame_stride, int mode, void* stream);
/* Same with host buffers (staged through device memory; synchronous).  With voices = 1, layout PLANAR,
 * frame_stride = 64 and frames = size <= 64 this is a literal AudioNode::process call on one node. */
int fdsp_bank_process_host(fdsp_bank* bank, size_t frames, const float* h_in, float* h_out, int layout,
                           size_t frame_stride, int mode);
int fdsp_bank_synchronize(fdsp_bank* bank);
/* Device time in milliseconds of the most recent fdsp_bank_process launch (HIP events on the launch stream). */
int fdsp_bank_last_kernel_ms(fdsp_bank* bank, float* ms);

/* Upload the first `frames` positions of ring node `ring_index` (visit order) for `count` voices from `first_voice`:
 * data is [count][frames] f32.  Used for state a Rust caller must provide because it comes from a crate outside the
 * reference tree: Pluck's excitation, the stream `Rnd::from_u64(hash).f32_in(-1.0, 1.0)` of funutd that
 * Pluck::initialize_line draws (src/oscillator.rs:257-261), goes into ring 0 of the "pluck" kind; Hold's draws
 * `Rnd::from_u64(hash).f64()` (src/noise.rs:299) go into that node's ring, each f64 as two f32 words, low word first. */
int fdsp_bank_set_ring(fdsp_bank* bank, int ring_index, const float* data, size_t frames, size_t first_voice,
                       size_t count);

/* ---- on-device voice scheduler: the reference's Sequencer with one event per voice ---------------------------
 * Replaces Sequencer::push + process / tick (src/sequencer.rs:355-398, 838-951, 769-836; ReplayMode::None, no loop
 * point) for a bank whose voices are the events' units.  `events` holds 4 doubles per voice -- start_time, end_time,
 * fade_in_time, fade_out_time, seconds on the sequencer clock -- and `fade` the curve per voice (NULL = Smooth).
 * As in push(), fade times may not exceed the event's duration.  Voices without an event never play.
 * fdsp_bank_process_events renders `frames` frames from the bank's sequencer clock (0 after set_events of a fresh
 * bank; see fdsp_bank_events_rewind) into d_out [outputs][frames][voices]: each voice's own faded contribution, zeros
 * outside its event.  Their sum over voices (fdsp_sum_voices) is the Sequencer's output.  Units are processed in
 * sequencer blocks of 64 frames exactly like the reference (a unit's first and last process() block is the part of
 * the sequencer block its event overlaps), so launches other than the last should be multiples of 64 frames.
 * FDSP_MODE_TICK follows Sequencer::tick.  d_in: [inputs][frames][voices] per-voice inputs (the reference feeds
 * every event the same input; give every voice the same stream to reproduce that). */
#define FDSP_FADE_POWER 0  /* Fade::Power: equal-power crossfades (sine_ease) */
#define FDSP_FADE_SMOOTH 1 /* Fade::Smooth: equal-amplitude crossfades (smooth5), the default */
int fdsp_bank_set_events(fdsp_bank* bank, const double* events, const int* fade, size_t first_voice, size_t count);
int fdsp_bank_process_events(fdsp_bank* bank, size_t frames, const float* d_in, float* d_out, int mode, void* stream);
/* The Sequencer's OUTPUT -- the sum of its events (sequencer.rs:838-951: every active event's faded block added into the output
 * buffer) -- in one launch: d_mix [outputs][frames], the mix-down's fixed summation order (below), so it equals fdsp_sum_voices of
 * fdsp_bank_process_events' output bit for bit; the per-event samples never exist in HBM.  Graphs of at most two outputs whose kind
 * has the fused kernels (FDSP_ENOTSUP otherwise); the clock advances as in fdsp_bank_process_events. */
int fdsp_bank_process_events_mix(fdsp_bank* bank, size_t frames, const float* d_in, float* d_mix, int mode, void* stream);
int fdsp_bank_events_rewind(fdsp_bank* bank, double time); /* set the sequencer clock (Sequencer::reset -> 0.0) */
double fdsp_bank_events_time(const fdsp_bank* bank);       /* Sequencer::time() */

/* ---- the stereo mix-down, per GPU (SURVEY 8d "mode B", 8e "per-GPU on-device tree-sum") ------------------------------
 * What the reference computes for a bank of voices feeding one output: `voice >> pan(p)` per voice (Panner::process,
 * src/pan.rs:50-76, weights src/pan.rs:13-17), then the sum over the voices (Reduce / Net mixing, src/audionode.rs:2406-2462).
 * SUMMATION ORDER -- fixed, independent of the launch geometry, the same for every function of this section:
 *     partial(group of 64 consecutive voices) = (S0 + S1) + (S2 + S3),
 *         Sq = ((x[16q] + x[16q+1]) + x[16q+2]) + ... + x[16q+15]     (voices past the end of the bank count as +0.0)
 *     mix = the groups' partials added in an aligned binary tree: level by level node(2k) + node(2k+1), a node without a
 *         right sibling passes through unchanged.
 * So a fused mix equals fdsp_sum_voices / fdsp_mix_stereo of the voice-out render bit for bit, and both are within
 * sqrt(voices) * 6e-8 * max|x| of a serial mix (tests/test_gpu_mix.py).
 *
 * fdsp_bank_process_mix: render `frames` samples of every voice AND reduce them over the voices in the same launch -- the
 * per-voice output never exists in HBM (a voice group's last stage parks MC frames in LDS, transposes and writes one float per
 * channel and frame; a second, tiny launch adds the groups' partials).  d_in: voice-minor [inputs][frames][voices] or NULL.
 *   mix = FDSP_MIX_SUM: d_mix [outputs][frames] = sum over the voices of every output channel (graphs that end in a Panner:
 *         BASELINE config 4; the Sequencer's mix of its events);
 *   mix = FDSP_MIX_PAN: mono graphs; every voice is panned with its own position (fdsp_bank_set_pan, -1 .. 1, default 0 =
 *         centre) exactly like Panner::tick does (`weight * sample`), d_mix = [2][frames].
 * The bank owns the partial-mix buffer ([voice groups][channels][frames] f32, 1/64 of a voice-out render); it grows on demand,
 * fdsp_bank_mix_reserve(bank, frames) sizes it ahead of a real-time loop or a stream capture (AudioNode::allocate semantics).
 * A mix launch always takes the stage pipeline (or, for small banks of eligible graphs, the time-split kernel): "pipe_split" = 0 and the
 * launch-length thresholds of fdsp_bank_process do not apply to it, and a kind whose graph has no pipeline plan answers FDSP_ENOTSUP.
 * Stream, ordering, timing and capture rules are those of fdsp_bank_process (including the refusal of a capture behind a queued
 * fdsp_bank_set_param_all; run-time compiled graphs build their mix kernels in fdsp_bank_mix_reserve AND in fdsp_bank_set_pan, whichever
 * the host calls before its loop); a CAPTURED launch holds the partial-mix buffer of
 * capture time, so reserve for the longest launch before capturing and do not grow the reservation while such a graph is alive.  FDSP_ENOTSUP: the kind was built without the
 * fused kernels (the BASELINE kinds fm_svf, sine_hz_lowpass_hz, saw_moog_adsr_pan, noise_biquad have them) -- render
 * voice-out and call the functions below, which use the same order. */
#define FDSP_MIX_SUM 1
#define FDSP_MIX_PAN 2
int fdsp_bank_process_mix(fdsp_bank* bank, size_t frames, const float* d_in, float* d_mix, int mix, int mode, void* stream);
int fdsp_bank_set_pan(fdsp_bank* bank, const float* h_pan, size_t first, size_t count);
int fdsp_bank_mix_reserve(fdsp_bank* bank, size_t frames);

/* The same mix-down of a voice-out render that already sits in HBM.
 * d_voices: [frames][voices] mono voice outputs; d_pan: [voices] pan position in -1..1 or NULL (centre);
 * d_mix: [2][frames].  Equal-power pan weights follow Panner (src/pan.rs:13-17). */
int fdsp_mix_stereo(const float* d_voices, const float* d_pan, float* d_mix, size_t frames, size_t voices,
                    void* stream);

/* Sum over voices of a voice-minor buffer d_in [channels][frames][voices] -> d_out [channels][frames] (per-GPU partial
 * of the mix-down for graphs that already end in a Panner). */
int fdsp_sum_voices(const float* d_in, float* d_out, size_t channels, size_t frames, size_t voices, void* stream);

/* Sum over the instances of a PLANAR render d_in [instances][rows] (rows = channels x frame_stride: reverb_stereo banks, planar
 * voice banks) -> d_out [rows]: the aligned binary tree of the summation order above, taken over the instances. */
int fdsp_sum_instances(const float* d_in, float* d_out, size_t rows, size_t instances, void* stream);

/* ---- the exchange step across GPUs: all-reduce(sum) of the per-GPU partial mixes over RCCL / xGMI ----
 * The only collective of the path: `count` = 2 * frames floats per GPU, latency-bound, issued once per launch.  It runs
 * on the communicator's own side stream, ordered behind the work already queued on `after_stream` (the mix kernel), so
 * the render stream can start the next launch immediately; consumers order themselves behind it with fdsp_comm_wait.
 *   one process, n GPUs:        fdsp_comm_create_local(n, devices, &comm) (devices NULL = 0..n-1); slot k <-> devices[k];
 *                               one thread per GPU calls fdsp_mix_allreduce(comm, k, ..), or one thread calls
 *                               fdsp_mix_allreduce_all(comm, mixes, count, streams) (the calls are grouped)
 *   one process per GPU:        rank 0: fdsp_comm_unique_id(id); ship the 128 bytes to the other ranks (MPI, a file,
 *                               torch.distributed ...); every rank: fdsp_comm_create_rank(id, nranks, rank, device, &comm);
 *                               slot is always 0.
 * The sum is in place in d_mix.  Summation order across GPUs is RCCL's (ring / tree): compare mixes with a tolerance,
 * voices bit for bit (SURVEY 8e parity caveat). */
#define FDSP_COMM_ID_BYTES 128
typedef struct fdsp_comm fdsp_comm;
int fdsp_comm_create_local(int n, const int* devices, fdsp_comm** out);
int fdsp_comm_unique_id(void* id128);
int fdsp_comm_create_rank(const void* id128, int nranks, int rank, int device, fdsp_comm** out);
void fdsp_comm_destroy(fdsp_comm* comm);
int fdsp_comm_ranks(const fdsp_comm* comm);        /* ranks of the whole communicator */
int fdsp_comm_local_slots(const fdsp_comm* comm);  /* ranks owned by this process */
int fdsp_comm_device(const fdsp_comm* comm, int slot);
int fdsp_mix_allreduce(fdsp_comm* comm, int slot, float* d_mix, size_t count, void* after_stream);
int fdsp_mix_allreduce_all(fdsp_comm* comm, float* const* d_mix, size_t count, void* const* after_streams);
int fdsp_comm_wait(fdsp_comm* comm, int slot, void* stream); /* stream == NULL: block the host */

/* ---- shared wavetables (Arc<Wavetable> singletons of the reference: saw_table/square_table/triangle_table,
 * src/wavetable.rs:493-560).  `set`: 0 = saw, 1 = square, 2 = triangle, 4 = organ, 5 = soft saw,
 * 6 = hammond (organ_table/soft_saw_table/hammond_table :546-623), 3 and 7 = user.  Tables are a list of
 * (pitch, power-of-two-length wave) pairs in ascending pitch, data concatenated.  fdsp_wavetable_build() generates the
 * built-in shape with the engine's own make_wave (wavetable.rs:44-123); fdsp_wavetable_upload() installs caller data
 * (e.g. tables produced by FunDSP itself).  Must be called before rendering a kind that uses the set. */
int fdsp_wavetable_build(int set);
int fdsp_wavetable_upload(int set, int n_tables, const float* h_pitches, const int* h_lengths, const float* h_data);
/* Shared sample buffers = the reference's Arc<Wave> handed to playwave() / playwave_at() (src/wave.rs:739-797,
 * prelude32.rs:2225-2248): `slot` 0..7, data [channels][length] f32.  The WavePlayer<slot> node of a graph reads it. */
int fdsp_wave_upload(int slot, int channels, size_t length, const float* h_data);
int fdsp_wavetable_get(int set, int* n_tables, float* h_pitches, int* h_lengths, float* h_data, size_t capacity);
/* The tables fdsp_wavetable_build(set) would install, computed on the host and returned without touching a device
 * (Wavetable::new + make_wave, wavetable.rs:44-123: f64 partial weights, f32 polar insert, f32 radix-2 inverse FFT,
 * f32 peak normalisation).  Bit-identical to the oracle's restatement (tests/test_wavetable_build.py); versus a
 * double-precision FFT of the same spectrum they differ by < 1e-6.  Pass NULL data to query n_tables / lengths. */
int fdsp_wavetable_compute(int set, int* n_tables, float* h_pitches, int* h_lengths, float* h_data, size_t capacity);

/* ---- host-side helpers that restate reference coefficient constructors with the engine's own math --------- */
int fdsp_svf_coefs(int mode, float sample_rate, float cutoff, float q, float gain, float* out6);     /* svf.rs:28-221 */
int fdsp_biquad_coefs(int kind, float sample_rate, float f, float q, float gain, float* out5);       /* biquad.rs:27-116 */
double fdsp_rnd1(uint64_t x);  /* math.rs:569-576 */
/* the engine's restatement of libm::sinf / cosf (lib.rs:470-492) for host-side parameter derivation, e.g. the matrix of
 * rotate(angle, gain) (prelude32.rs:2432) or pan weights */
float fdsp_libm_sinf(float x);
float fdsp_libm_cosf(float x);
uint64_t fdsp_hash1(uint64_t x); /* math.rs:592-599 */

#ifdef __cplusplus
}
#endif
#endif

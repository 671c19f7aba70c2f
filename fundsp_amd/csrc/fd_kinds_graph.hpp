// fd_kinds_graph.hpp -- the graph types of BASELINE configs 2 and 4, shared by fd_kinds_graph.hip (their kinds) and
// fd_kinds_graph_mix.hip (their kernels with the fused mix-down).  The types spell out exactly what the reference's operator
// overloads build (combinator.rs:289-488; Rust precedence `*` > `+` > `>>`).
#pragma once

#include "fd_engine.hpp"

namespace fd {
// sine_hz(f) = constant(f) >> sine()                       prelude.rs:349
using SineHz = Pipe<Constant<1>, Sine>;
// config 2 voice: noise() >> biquad(..) -- arithmetic of one BiquadBank<f32x8> lane fed by white noise
using NoiseBiquad = Pipe<Noise, Biquad>;
// the FM pair of config 3 (README.md:98-103), here for the oversample / resample kinds below
using FmMod = Unop<Unop<Unop<SineHz, UMulScalar>, UMulScalar>, UAddScalar>;

// config 4 voice: ((dc(f) >> saw() | dc(fc) | dc(q)) >> moog()) * adsr_live(a, d, s, r) >> pan(p)
// (`|` binds loosest, `*` tightest: combinator.rs; moog() is the 3-input variant prelude.rs:551; the gate is the
// graph's one input, feeding adsr_live -- Binop inputs = X inputs (0) + Y inputs (1), audionode.rs:912)
using SawMoog = Pipe<Stack<Stack<Pipe<Constant<1>, WaveSynth<0>>, Constant<1>>, Constant<1>>, Moog<3>>;
using SawMoogAdsrPan = Pipe<Binop<OpMul, SawMoog, AdsrLive>, Panner>;
// config 4 voice in the reference's own gate shape (examples/live_adsr.rs:72 `var(&control) >> adsr_live(..)`; SURVEY 8(d) `dc(gate) >> adsr_live`):
//   ((dc(f) >> saw() | dc(fc) | dc(q)) >> moog()) * (var(gate) >> adsr_live(a, d, s, r)) >> pan(p)
// The gate is a per-voice Var slot, block-constant like Var::process (shared.rs:122-125: one atomic read per block, splat): NO graph
// input, hence no HBM gate stream, no loader wave and no feed ring in LDS -- the host flips the slot between launches
// (fdsp_bank_set_param "0.1.0:value" plays Shared::set_value).
using SawMoogVarAdsrPan = Pipe<Binop<OpMul, SawMoog, Pipe<Var, AdsrLive>>, Panner>;
// Build-time pins of the config-4 stage plan (fd_device.hpp): three compute stages cut behind the oscillator stack and behind the ladder;
// the stack's two trailing Constants stay out of the first hand-over (ConstTail), which leaves ONE channel per cut and, for two voice
// groups per workgroup, 32-frame tiles; with a gate input that makes four roles, i.e. a round loop per role.
static_assert(pipe_plan<SawMoogAdsrPan>(0).S == 3 && pipe_plan<SawMoogAdsrPan>(0).K1 == 1 && pipe_plan<SawMoogAdsrPan>(0).K2 == 2, "config 4: [saw stack] [moog] [* adsr >> pan]");
static_assert(PipeTiles<SawMoogAdsrPan, 3, 1, 2, 2>::S0::OUT == (FD_PIPE_ELIDE ? 1 : 3) && PipeTiles<SawMoogAdsrPan, 3, 1, 2, 2>::S1::IN == (FD_PIPE_ELIDE ? 1 : 3), "config 4: the first cut carries the oscillator only");
static_assert(PipeTiles<SawMoogAdsrPan, 3, 1, 2, 2>::SUB == (FD_PIPE_ELIDE ? 32 : 16) && PipeTiles<SawMoogAdsrPan, 3, 1, 2, 4>::SUB == (FD_PIPE_ELIDE ? 16 : 8), "config 4: tile lengths");
static_assert(SawMoogVarAdsrPan::IN == 0 && pipe_plan<SawMoogVarAdsrPan>(0).S == 3 && pipe_plan<SawMoogVarAdsrPan>(0).K1 == 1 && pipe_plan<SawMoogVarAdsrPan>(0).K2 == 2, "config 4, Var gate: the same three stages, no loader");
static_assert(PipeTiles<SawMoogVarAdsrPan, 3, 1, 2, 2>::SUB == (FD_PIPE_ELIDE ? 64 : 16) && PipeTiles<SawMoogVarAdsrPan, 3, 1, 2, 4>::SUB == (FD_PIPE_ELIDE ? 32 : 8), "config 4, Var gate: no feed ring, the tiles are twice as long");
static_assert(!TsPlan<SawMoogVarAdsrPan>::ok && TsPlan<NoiseBiquad>::ok, "time-split kernels: config 2's generator chain yes (a counter, a feed-forward half, the recurrence); "
                                                                        "config 4's no -- a wavetable oscillator and a ladder cannot skip frames");
// ... and of the launch lengths from which the two kinds leave the single-wave kernel (PipeMinT: measured, profiles/r04_small_t_kernels.txt)
#ifndef FD_PIPE_MIN_T
static_assert(PipeMinT<NoiseBiquad>::v == 256 && PipeMinT<SawMoogAdsrPan>::v == 64, "config 2: the pipeline from four blocks on; config 4: from one");
#endif

}  // namespace fd

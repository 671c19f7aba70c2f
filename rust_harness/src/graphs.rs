//! The inventory graphs of tests/test_gpu_jit.py::GRAPHS, spelled with the reference's own prelude32 operators.
//! Same names, same constants, same operator precedence (Python's and Rust's agree on `* + >> & ^ |`); Python's `~x`
//! is Rust's `!x` (Thru).  Closure arguments of busi/stacki/sumi/pipei are u64 indices, of busf/branchf f32 fractions
//! i / (n - 1), exactly as in prelude.rs.  Arguments the Python side computes in double before they reach an f32
//! parameter are computed here the same way (f64, then `as f32`).

use fundsp::prelude32::*;

/// Frames rendered per graph (tests/golden/make_golden.py::GRAPH_FRAMES).
pub const FRAMES: usize = 64 * 3 + 9;

/// (name, number of input channels)
pub fn names() -> Vec<(&'static str, usize)> {
    vec![
        ("noise_moog", 0),
        ("fm_pair_shaped", 0),
        ("modulated_svf", 0),
        ("stack_binop_sub", 0),
        ("comb_allpass_chain", 0),
        ("saw_filter_env", 1),
        ("chorus_tap", 1),
        ("pulse_resonator", 0),
        ("bus_branch_thru", 0),
        ("split_join", 0),
        ("busi_sines", 0),
        ("stacki_sumi", 0),
        ("branchf_filters", 1),
        ("pipei_poles", 0),
        ("busf_resonators", 0),
        ("impulse_declick", 0),
        ("svf_q_forms", 1),
        ("brown_pink", 0),
        ("nl_biquads", 1),
        ("feedback_echo", 1),
        ("fdn4", 1),
        ("fdn2_loop_filters", 1),
        ("feedback_denormal_decay", 0),
        ("pulse_wave", 0),
        ("organ_family", 0),
        ("multitap_allnest_panner", 1),
        ("multitap_linear3", 1),
        ("limiter_mono", 1),
        ("limiter_stereo", 1),
        ("meters", 1),
        ("moog_q_thru_cut", 1),
    ]
}

fn f(x: f64) -> f32 {
    x as f32
}

#[allow(clippy::precedence)]
pub fn build(name: &str) -> Box<dyn AudioUnit> {
    match name {
        "noise_moog" => Box::new(noise() >> moog_hz(1500.0, 0.4)),
        "fm_pair_shaped" => Box::new((sine_hz(110.0) + sine_hz(220.0) * 0.5) >> shape(Tanh(2.0))),
        "modulated_svf" => Box::new((noise() | sine_hz(0.7) * 800.0 + 1000.0 | dc(2.0)) >> lowpass()),
        "stack_binop_sub" => Box::new(
            (noise() | noise()) >> (lowpole_hz(500.0) | highpole_hz(2000.0)) >> (pass() - pass()),
        ),
        "comb_allpass_chain" => Box::new(
            noise() >> allnest_c(0.5, delay(0.002)) >> dcblock_hz(20.0) >> peak_hz(3000.0, 2.0) * 0.25,
        ),
        "saw_filter_env" => Box::new(
            (saw_hz(82.4) >> lowpass_hz(900.0, 3.0)) * (pass() >> adsr_live(0.002, 0.01, 0.5, 0.005)),
        ),
        "chorus_tap" => Box::new((pass() | sine_hz(1.3) * 0.001 + 0.003) >> tap(0.001, 0.005)),
        "pulse_resonator" => {
            Box::new((dc((140.0, 0.3)) >> poly_pulse()) >> resonator_hz(700.0, 40.0) >> pan(-0.4))
        }
        "bus_branch_thru" => Box::new(
            (sine_hz(440.0) & sine_hz(220.0)) >> (pass() ^ lowpole_hz(100.0)) >> (!sink() | pass()),
        ),
        "split_join" => Box::new(
            (noise() | noise())
                >> multisplit::<U2, U3>()
                >> multijoin::<U2, U3>()
                >> reverse::<U2>()
                >> join::<U2>()
                >> split::<U3>()
                >> join::<U3>(),
        ),
        "busi_sines" => Box::new(busi::<U4, _, _>(|i| sine_hz(f(100.0 * (i + 1) as f64))) * 0.25),
        "stacki_sumi" => Box::new(
            stacki::<U3, _, _>(|i| sine_hz(f(100.0 * (i + 1) as f64)))
                >> sumi::<U3, _, _>(|i| lowpole_hz(f(100.0 + i as f64))),
        ),
        "branchf_filters" => Box::new(
            branchf::<U3, _, _>(|t| lowpass_hz(f(500.0 + 1500.0 * t as f64), 1.0)) >> join::<U3>(),
        ),
        "pipei_poles" => Box::new(noise() >> pipei::<U4, _, _>(|i| lowpole_hz(f(1000.0 + 100.0 * i as f64)))),
        "busf_resonators" => Box::new(busf::<U5, _, _>(|t| {
            (noise() | dc((f(200.0 + 900.0 * t as f64), 20.0))) >> !resonator() >> resonator()
        })),
        "impulse_declick" => Box::new((impulse::<U1>() + noise()) >> declick_s(0.004)),
        "svf_q_forms" => Box::new(
            (pass() | sine_hz(2.0) * 300.0 + 1000.0) >> (lowpass_q(2.0) ^ bell_q(1.5, 2.0)) >> (pass() - pass()),
        ),
        "brown_pink" => Box::new(brown() & pink()),
        "nl_biquads" => Box::new(
            fresonator_hz(Tanh(1.0), 500.0, 2.0) >> dlowpass_hz(Softsign(0.9), 800.0, 1.0) >> clip_to(-0.5, 0.5),
        ),
        "feedback_echo" => Box::new(
            feedback::<U1, _>(delay(0.001) * 0.9)
                >> feedback2::<U1, _, _>(delay(0.0007), lowpole_hz(1500.0) * 0.8),
        ),
        "fdn4" => Box::new(
            split::<U4>()
                >> fdn::<U4, _>(stacki::<U4, _, _>(|i| {
                    delay(f(0.0005 * (i + 1) as f64)) >> fir((0.3, 0.4, 0.2))
                }))
                >> join::<U4>(),
        ),
        "fdn2_loop_filters" => Box::new(
            (pass() | noise() * 0.01)
                >> fdn2::<U2, _, _>(
                    stacki::<U2, _, _>(|i| delay(f(0.0011 * (i + 1) as f64))),
                    stacki::<U2, _, _>(|_i| lowpole_hz(3000.0) * 0.7),
                )
                >> join::<U2>(),
        ),
        "feedback_denormal_decay" => {
            Box::new(impulse::<U1>() >> feedback::<U1, _>(tick() * 0.5) >> lowpole_hz(5000.0))
        }
        "pulse_wave" => {
            Box::new((sine_hz(3.0) * 50.0 + 220.0 | sine_hz(0.7) * 0.3 + 0.5) >> pulse() * 0.2)
        }
        "organ_family" => Box::new(
            organ_hz(110.0) + soft_saw_hz(220.0) * 0.5 + (sine_hz(5.0) * 3.0 + 55.0 >> hammond()),
        ),
        "multitap_allnest_panner" => Box::new(
            (pass() | sine_hz(0.9) * 0.001 + 0.003 | sine_hz(1.7) * 0.0005 + 0.002)
                >> multitap::<U2>(0.001, 0.005)
                >> (pass() | sine_hz(3.0) * 0.6)
                >> allnest(delay(0.0013))
                >> multitick::<U1>()
                >> (pass() | sine_hz(0.5))
                >> panner(),
        ),
        "multitap_linear3" => Box::new(
            (pass() | dc((0.001, 0.0021, 0.0034))) >> multitap_linear::<U3>(0.0005, 0.004) >> join::<U1>(),
        ),
        "limiter_mono" => Box::new(pass() * 3.0 >> limiter(0.002, 0.02)),
        "limiter_stereo" => Box::new(
            (pass() * 2.0 | noise() * (sine_hz(3.0) + 1.0)) >> limiter_stereo(0.001, 0.01),
        ),
        "meters" => {
            let gain = shared(0.7);
            let level = shared(0.0);
            Box::new(
                (pass() * var(&gain))
                    >> (meter(Meter::Peak(0.01))
                        ^ meter(Meter::Rms(0.02))
                        ^ meter(Meter::Sample)
                        ^ monitor(&level, Meter::Rms(0.005))),
            )
        }
        "moog_q_thru_cut" => Box::new(
            (pass() | dc(800.0)) >> moog_q(0.5) >> clip() >> split::<U2>() >> !(sink() | sink()) >> join::<U2>(),
        ),
        _ => panic!("unknown graph {name}"),
    }
}

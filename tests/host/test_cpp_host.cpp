// C++ host side (include/fundsp_hip.hpp) against the oracle -- written the way the reference's own tests read
// (tests/test_basic.rs: check_wave :21-47, tick vs process, reset determinism; README.md:98-103 FM patch).
//
//   test_cpp_host --host   no device: graph notation, type strings, arity errors, hiprtc type check
//   test_cpp_host --gpu    on an MI355X: renders through Bank / render() and compares with the oracle bit for bit
//
// Test infrastructure: links oracle/libfundsp_oracle.so (the checker) and fundsp_amd/libfundsp_hip.so (the product).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fundsp_hip.hpp"
#include "fundsp_oracle.h"

using namespace fundsp_hip;

static int failures = 0;
#define EXPECT(cond)                                                           \
    do {                                                                       \
        if (!(cond)) {                                                         \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);        \
            failures++;                                                        \
        }                                                                      \
    } while (0)

template <class F>
static bool throws(F&& f) {
    try {
        f();
    } catch (const Error&) {
        return true;
    }
    return false;
}

static bool bit_equal(const float* a, const float* b, size_t n, const char* what) {
    for (size_t i = 0; i < n; i++)
        if (std::memcmp(&a[i], &b[i], 4) != 0) {
            std::printf("FAIL %s: sample %zu got %.9g want %.9g\n", what, i, a[i], b[i]);
            failures++;
            return false;
        }
    return true;
}

static void host_checks() {
    // a C++ host holds no other ROCm: the run-time compiler is the hiprtc the library links, the one of the ROCm it was built with
    // (a Python process answers "isolated: ..": include/fundsp_hip.h fdsp_jit_compiler, tests/test_gpu_jit_compiler.py)
    {
        const std::string who = fdsp_jit_compiler();
        EXPECT(who.rfind("linked: ", 0) == 0 && who.find("libhiprtc") != std::string::npos);
    }
    // README.md:98-103 of the reference: sine_hz(f) * f * m + f >> sine() >> lowpass_hz(fc, q)
    const float f = 110.0f, m = 2.0f;
    An fm = sine_hz(f) * f * m + f >> sine() >> lowpass_hz(1000.0f, 1.0f);
    EXPECT(fm.type == "Pipe<Pipe<Unop<Unop<Unop<Pipe<Constant<1>,Sine>,UMulScalar>,UMulScalar>,UAddScalar>,Sine>,FixedSvf>");
    EXPECT(fm.inputs == 0 && fm.outputs == 1);
    EXPECT(fdsp_graph_check(fm.type.c_str()) == 0);
    // operator table arities (tests/test_basic.rs:604-640)
    EXPECT((pass() ^ pass()).outputs == 2);
    EXPECT((sink() | zero()).inputs == 1 && (sink() | zero()).outputs == 1);
    EXPECT((!butterpass() >> lowpole()).inputs == 2);
    EXPECT((pass() & lowpole_hz(100.0f)).outputs == 1);
    EXPECT(throws([] { sine() >> (sine() | sine()); }));   // a compile-time error in Rust
    EXPECT(throws([] { pass() & sink(); }));
    EXPECT(fdsp_graph_check(fdn(stacki(3, [](int) { return pass(); })).type.c_str()) < 0);   // FrameHadamard: power of two
    // parameter addressing follows the C ABI's "<path>:<field>" slot names
    An g = sine_hz(440.0f) >> lowpass_hz(1000.0f, 1.0f);
    bool found = false;
    for (const Param& p : g.params) found = found || p.slot() == "1:cutoff";
    EXPECT(found);
    An echo = feedback(delay(0.5f) * 0.5f);   // test_basic.rs:646
    EXPECT(echo.inputs == 1 && echo.outputs == 1 && echo.rings == 1);
    EXPECT(fdsp_graph_check(echo.type.c_str()) == 0);
    EXPECT(fdsp_graph_check((busi(4, [](int i) { return sine_hz(100.0f * (i + 1)); }) >> split(2) >> join(2)).type.c_str()) == 0);
    EXPECT(throws([] { Bank b("no_such_kind", 4); }));
    // the composed opcodes build the same types as the reference's prelude would
    An rv = reverb4_stereo(20.0, 2.0);
    EXPECT(rv.inputs == 2 && rv.outputs == 2 && rv.rings == 32 && fdsp_graph_check(rv.type.c_str()) == 0);
    An fl = noise() >> flanger(0.6f, 0.002f, 0.006f, lfo("EnvSineHz", "").with_child(0, "hz", 0.7f).with_child(0, "lo", 0.002f).with_child(0, "hi", 0.006f));
    EXPECT(fl.type == "Pipe<Noise,Bus<Pass,Feedback2<Pipe<Stack<Pass,Envelope<EnvSineHz>>,TapT<false>>,Shaper,FbId>>>");
    EXPECT(fdsp_graph_check(fl.type.c_str()) == 0);
    bool lfo_slot = false;
    for (const Param& p : fl.params) lfo_slot = lfo_slot || p.slot() == "1.1.0.0.1.0:hz";
    EXPECT(lfo_slot);
    An chain = (pass() | dc(1.0f)) >> rotate(0.5f, 1.0f) >> (pass() | sink()) >> meter(METER_PEAK, 0.1) >> mul(0.5f);   // test_flow.rs:176
    EXPECT(chain.inputs == 1 && chain.outputs == 1 && fdsp_graph_check(chain.type.c_str()) == 0);
    EXPECT(playwave_at(3, 1, 10, 200, 50).type == "WavePlayer<3>");
}

static onode* oracle_fm(float f, float m, float fc, float q) {
    float c = f;
    onode* mod = o_pipe(o_constant(1, &c), o_sine());
    onode* g = o_unop(O_ADD_SCALAR, o_unop(O_MUL_SCALAR, o_unop(O_MUL_SCALAR, mod, f), m), f);
    return o_pipe(o_pipe(g, o_sine()), o_fixed_svf(O_SVF_LOWPASS, fc, q, 1.0f));
}

static void gpu_checks() {
    const double SR = 48000.0;
    // config 1: sine_hz(440) >> lowpass_hz(1000, 1), one voice, one second through Wave::render's blocking
    {
        Bank b = Bank::from_graph(sine_hz(440.0f) >> lowpass_hz(1000.0f, 1.0f), 1);
        std::vector<float> got = render(SR, 1.0, b);
        float c = 440.0f;
        onode* n = o_pipe(o_pipe(o_constant(1, &c), o_sine()), o_fixed_svf(O_SVF_LOWPASS, 1000.0f, 1.0f, 1.0f));
        std::vector<float> want(48000);
        EXPECT(o_wave_render(n, SR, 1.0, want.data(), want.size()) == 48000);
        EXPECT(got.size() == 48000);
        bit_equal(got.data(), want.data(), 48000, "config 1 render");
        // reset restores the initial state (doc-test audionode.rs:44-49)
        b.reset();
        std::vector<float> again = render(SR, 1.0, b);
        bit_equal(again.data(), got.data(), 48000, "render after reset");
        // the tick path: 100 samples, one per call, against the oracle's tick after reset
        b.reset();
        o_reset(n);
        for (int i = 0; i < 100; i++) {
            float y = 0.0f, w = 0.0f;
            b.tick(nullptr, &y);
            o_tick(n, nullptr, &w);
            if (!bit_equal(&y, &w, 1, "tick")) break;
        }
        o_free(n);
    }
    // the FM patch with per-voice parameters, 70 voices, seeds per voice; blocks of 64, 13, 0 and 64 samples
    {
        const size_t V = 70;
        std::vector<float> f(V), m(V), fc(V), q(V);
        std::vector<uint64_t> seeds(V);
        for (size_t v = 0; v < V; v++) {
            f[v] = 55.0f + 20.0f * (float)v;
            m[v] = 0.5f + 0.1f * (float)v;
            fc[v] = 300.0f + 90.0f * (float)v;
            q[v] = 0.5f + 0.04f * (float)v;
            seeds[v] = 1000 + 17 * v;
        }
        An g = sine_hz(f) * f * m + f >> sine() >> lowpass_hz(fc, q);
        Bank b = Bank::from_graph(g, V, 0, SR);
        b.set_seed(seeds);
        const size_t sizes[4] = {64, 13, 0, 64};
        std::vector<std::vector<float>> blocks;
        for (size_t s : sizes) {
            std::vector<float> out(V * 64, -7.0f);
            b.process(s, nullptr, out.data());
            blocks.push_back(out);
        }
        for (size_t v : {(size_t)0, (size_t)33, V - 1}) {
            onode* n = oracle_fm(f[v], m[v], fc[v], q[v]);
            o_set_sample_rate(n, SR);
            o_set_seed(n, seeds[v]);
            for (int k = 0; k < 4; k++) {
                float want[64] = {0};
                o_process(n, (int)sizes[k], nullptr, want);
                bit_equal(&blocks[k][v * 64], want, sizes[k], "fm process block");
                for (size_t i = sizes[k]; i < 64; i++) EXPECT(blocks[k][v * 64 + i] == -7.0f);  // samples past `size` untouched
            }
            o_free(n);
        }
        // Clone: a clone continues exactly like the original
        Bank c = b.clone();
        std::vector<float> a1(V * 64), a2(V * 64);
        b.process(64, nullptr, a1.data());
        c.process(64, nullptr, a2.data());
        bit_equal(a1.data(), a2.data(), V * 64, "clone continues identically");
    }
    // a wavetable voice through the closure-form combinators: tables are built on first use; tick == process within 1e-4
    // like check_wave (tests/test_basic.rs:21-47)
    {
        An g = busi(3, [](int i) { return saw_hz(110.0f * (float)(i + 1)) * 0.3f; }) >> (pass() ^ lowpole_hz(800.0f)) >> join(2);
        Bank b = Bank::from_graph(g, 4, 0, SR);
        std::vector<float> w = render(SR, 0.01, b);
        b.reset();
        const size_t length = w.size() / 4;
        double worst = 0.0;
        for (size_t i = 0; i < length; i++) {
            std::vector<float> y(4);
            b.tick(nullptr, y.data());
            for (size_t v = 0; v < 4; v++) worst = std::max(worst, (double)std::fabs(y[v] - w[v * length + i]));
        }
        EXPECT(worst <= 1e-4);
        float peak = 0.0f;
        for (float x : w) peak = std::max(peak, std::fabs(x));
        EXPECT(peak > 0.05f && peak < 2.0f);
    }
    // reverb_stereo(10, 2, 0.5) through its dedicated kernel: an impulse against the oracle's Feedback graph restatement
    {
        Bank b = Bank::reverb_stereo(2, 10.0, 2.0, 0.5);
        b.set_sample_rate(SR);
        onode* n = o_reverb_stereo(10.0, 2.0, 0.5);
        o_set_sample_rate(n, SR);
        EXPECT(b.inputs() == 2 && b.outputs() == 2);
        for (int blk = 0; blk < 40; blk++) {   // 2560 frames: past the shortest delay lines
            std::vector<float> x(2 * 2 * 64, 0.0f), got(2 * 2 * 64), want(2 * 64);
            if (blk == 0) x[0] = x[64] = x[128] = x[192] = 1.0f;
            b.process(64, x.data(), got.data());
            o_process(n, 64, x.data(), want.data());
            if (!bit_equal(got.data(), want.data(), 128, "reverb_stereo instance 0")) break;
            if (!bit_equal(got.data() + 128, want.data(), 128, "reverb_stereo instance 1")) break;
        }
        o_free(n);
    }
    // the prelude's own fdn example (prelude.rs:1334): split >> fdn::<U16>(stacki(delay >> fir)) >> join through Bank::fdn (fdsp_fdn_create, the
    // lane-per-frame kernel) against the oracle's generic Feedback graph, block by block
    {
        std::vector<double> delays;
        std::vector<onode*> lines;
        const float w[3] = {0.2f, 0.4f, 0.2f};
        for (int i = 0; i < 16; i++) {
            const float t = 0.01f + 0.00125f * (float)i;   // 480 .. 1380 samples at 48 kHz
            delays.push_back((double)t);
            lines.push_back(o_pipe(o_delay((double)t), o_fir(3, w)));
        }
        Bank b = Bank::fdn(2, delays, {0.2f, 0.4f, 0.2f});
        b.set_sample_rate(SR);
        onode* n = o_pipe(o_pipe(o_split(1, 16), o_feedback(o_multi(1, 16, lines.data(), 0), nullptr, 1)), o_join(1, 16));
        o_set_sample_rate(n, SR);
        EXPECT(b.inputs() == 1 && b.outputs() == 1);
        uint32_t s = 777;
        for (int blk = 0; blk < 60; blk++) {   // 3840 frames: several trips around every line
            std::vector<float> x(2 * 64, 0.0f), got(2 * 64), want(64);
            for (int i = 0; i < 64; i++) {
                s = s * 1664525u + 1013904223u;
                x[i] = x[64 + i] = blk < 30 ? (float)(s >> 8) * (1.0f / 8388608.0f) - 1.0f : 0.0f;
            }
            b.process(64, x.data(), got.data());
            o_process(n, 64, x.data(), want.data());
            if (!bit_equal(got.data(), want.data(), 64, "fdn<16> instance 0")) break;
            if (!bit_equal(got.data() + 64, want.data(), 64, "fdn<16> instance 1")) break;
        }
        o_free(n);
    }
    // reverb3_stereo(2.0, 0.5, lowpole_hz(8000)) (the reference's own example, prelude.rs:1850-1856) through Bank::reverb3_stereo against the
    // oracle's Reverb::tick restatement, block by block, a burst and its tail
    {
        Bank b = Bank::reverb3_stereo(2, 2.0, 0.5, 8000.0f);
        b.set_sample_rate(SR);
        onode* fl[16];
        for (int i = 0; i < 16; i++) fl[i] = o_onepole(0, 1, 8000.0f);
        onode* n = o_reverb3(2.0, 0.5, fl);
        o_set_sample_rate(n, SR);
        EXPECT(b.inputs() == 2 && b.outputs() == 2);
        uint32_t s = 4242;
        for (int blk = 0; blk < 200; blk++) {   // 12 800 frames: once around the loop
            std::vector<float> x(2 * 2 * 64, 0.0f), got(2 * 2 * 64), want(2 * 64);
            if (blk < 40)
                for (int i = 0; i < 128; i++) {
                    s = s * 1664525u + 1013904223u;
                    x[i] = x[128 + i] = (float)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
                }
            b.process(64, x.data(), got.data());
            o_process(n, 64, x.data(), want.data());
            if (!bit_equal(got.data(), want.data(), 128, "reverb3_stereo instance 0")) break;
            if (!bit_equal(got.data() + 128, want.data(), 128, "reverb3_stereo instance 1")) break;
        }
        o_free(n);
    }
    // the reference's own `reverb` bench, (noise() | noise()) >> reverb_stereo(10.0, 1.0, 0.5) at 44.1 kHz (benches/benchmark.rs:79-85), as a Chain of two
    // banks -- the generator's fused kernel, the network's lane-per-frame kernel -- against the oracle's ONE graph: as constructed (the hash Pipe::new's
    // probe ping hands down, read from the device by probe_hash) and after set_seed, block by block until the tail sounds
    {
        const size_t V = 2;
        const double R = 44100.0;
        const An whole = (noise() | noise()) >> reverb_stereo(10.0, 1.0, 0.5);
        Bank rev = Bank::reverb_stereo(V, 10.0, 1.0, 0.5);
        rev.set_sample_rate(R);
        Chain ch(Bank::from_graph(noise() | noise(), V, 0, R, /* flush_denormals: the one graph has a Feedback node */ true), std::move(rev), &whole);
        EXPECT(ch.inputs() == 0 && ch.outputs() == 2 && ch.voices() == V);
        onode* n = o_pipe(o_stack(o_noise(), o_noise()), o_reverb_stereo(10.0, 1.0, 0.5));
        o_set_sample_rate(n, R);
        for (int pass = 0; pass < 2; pass++) {
            if (pass == 1) {
                ch.reset();
                ch.set_seed(77);
                o_reset(n);
                o_set_seed(n, 77);
            }
            for (int blk = 0; blk < 60; blk++) {   // 3 840 frames: past the longest line's first return
                std::vector<float> got(V * 2 * 64), want(2 * 64);
                ch.process(64, nullptr, got.data());
                o_process(n, 64, nullptr, want.data());
                if (!bit_equal(got.data(), want.data(), 128, pass ? "chain after set_seed, instance 0" : "chain as constructed, instance 0")) break;
                if (!bit_equal(got.data() + 128, want.data(), 128, pass ? "chain after set_seed, instance 1" : "chain as constructed, instance 1")) break;
            }
        }
        o_free(n);
    }
    // README.md:436, "to add 20% reverb to a stereo signal": multipass() & 0.2 * reverb_stereo(20.0, 2.0, 1.0) -- the reverb's bank with the gain and the
    // dry bus folded into its kernel (Bank::set_bus) against the oracle's graph of Bus, MultiPass, Unop and the reverb, block by block
    {
        Bank b = Bank::reverb_stereo(2, 20.0, 2.0, 1.0);
        b.set_sample_rate(SR);
        b.set_bus(FDSP_BUS_DRY_WET, 0.2f);
        onode* n = o_bus(o_multipass(2), o_unop(O_MUL_SCALAR, o_reverb_stereo(20.0, 2.0, 1.0), 0.2f));
        o_set_sample_rate(n, SR);
        uint32_t s = 99;
        for (int blk = 0; blk < 120; blk++) {   // 7 680 frames: past the longest line's first return (20 m room)
            std::vector<float> x(2 * 2 * 64, 0.0f), got(2 * 2 * 64), want(2 * 64);
            if (blk < 30)
                for (int i = 0; i < 128; i++) {
                    s = s * 1664525u + 1013904223u;
                    x[i] = x[128 + i] = (float)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
                }
            b.process(64, x.data(), got.data());
            o_process(n, 64, x.data(), want.data());
            if (!bit_equal(got.data(), want.data(), 128, "multipass() & 0.2 * reverb_stereo, instance 0")) break;
            if (!bit_equal(got.data() + 128, want.data(), 128, "multipass() & 0.2 * reverb_stereo, instance 1")) break;
        }
        o_free(n);
    }
    // a filter with an input: noise through the C ABI, the same samples through the oracle
    {
        Bank b("fixed_svf", 1);
        b.set(":cutoff", 1234.0f);
        b.set(":q", 0.7f);
        b.set_sample_rate(SR);
        onode* n = o_fixed_svf(O_SVF_LOWPASS, 1234.0f, 0.7f, 1.0f);
        o_set_sample_rate(n, SR);
        float x[64], got[64], want[64];
        uint32_t s = 12345;
        for (int blk = 0; blk < 5; blk++) {
            for (float& t : x) {
                s = s * 1664525u + 1013904223u;
                t = (float)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
            }
            b.process(64, x, got);
            o_process(n, 64, x, want);
            bit_equal(got, want, 64, "fixed_svf process");
        }
        o_free(n);
    }
}

int main(int argc, char** argv) {
    const bool gpu = argc > 1 && std::strcmp(argv[1], "--gpu") == 0;
    try {
        host_checks();
        if (gpu) gpu_checks();
    } catch (const std::exception& e) {
        std::printf("FAIL: exception %s\n", e.what());
        return 2;
    }
    std::printf("%s: %d failure(s)\n", gpu ? "gpu" : "host", failures);
    return failures ? 1 : 0;
}

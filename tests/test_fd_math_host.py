"""Host-side check of the device math header: the optimistic packed sine (wide_sin2) and its scalar twin (wide_sin1)
must reproduce the general scalar restatement of wide::f32x8::sin bit for bit everywhere inside their guard domain
(|quadrant index| < 8192), including quadrant ties, tiny and negative arguments.  Built with hipcc for the host."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wide_sin2_matches_wide_sinf_on_host(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = tmp_path / "check_wide_sin2"
    cmd = [hipcc, "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-std=c++17", "-Wno-unused-result",
           "-I", os.path.join(ROOT, "fundsp_amd", "csrc"), "-o", str(exe),
           os.path.join(ROOT, "tests", "host", "check_wide_sin2.hip")]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bad 0" in r.stdout
    # tanhf_common (the ladder's packed-path saturator, guarded): every 257th bit pattern of its whole domain here; all
    # 2.18e9 of them were run once on this host (0 differences, 21 CPU-minutes) and all 2^32 on the device
    # (tests/host/check_tanh_common_device.hip, profiles/r03_tanh_common_device_exhaustive.txt)
    r = subprocess.run([str(exe), "--common", "257"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bad 0" in r.stdout


def test_device_trig_header_matches_oracle_over_the_whole_f32_range(tmp_path):
    """fd_math.hpp's sinf/cosf/tanf (host build) vs oracle/o_math.h, bit for bit, 20M arguments of every exponent incl.
    the Payne-Hanek branch, inf and NaN."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = tmp_path / "check_trig_big"
    cmd = [hipcc, "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-std=c++17", "-Wno-unused-result",
           "-I", os.path.join(ROOT, "fundsp_amd", "csrc"), "-I", os.path.join(ROOT, "oracle"), "-o", str(exe),
           os.path.join(ROOT, "tests", "host", "check_trig_big.hip")]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bad 0" in r.stdout


def test_tolerance_mode_sine_bound(tmp_path):
    """FDSP_MATH_FAST's sine (fast_sin1 / fast_sin2) vs the wide::f32x8::sin restatement and vs double sin."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = tmp_path / "check_fast_sin"
    cmd = [hipcc, "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-std=c++17", "-Wno-unused-result",
           "-I", os.path.join(ROOT, "fundsp_amd", "csrc"), "-o", str(exe),
           os.path.join(ROOT, "tests", "host", "check_fast_sin.hip")]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


def test_svf_packed_paths_match_the_oracle_on_overflow_bursts(tmp_path):
    """SvfCore::tick_fused (2*v - ic as one FMA, guarded against the |v| >= 2^127 case where the reference's unfused
    product overflows) and FixedSvfLp (lowpass output = v2), each with its per-tile rollback, vs the oracle's svf tick."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    import oracle as O
    O.lib()   # make sure oracle/libfundsp_oracle.so is built
    exe = tmp_path / "check_svf_paths"
    odir = os.path.join(ROOT, "oracle")
    cmd = [hipcc, "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-std=c++17", "-Wno-unused-result",
           "-I", os.path.join(ROOT, "fundsp_amd", "csrc"), "-I", odir, "-o", str(exe),
           os.path.join(ROOT, "tests", "host", "check_svf_paths.hip"), "-L" + odir, "-lfundsp_oracle", "-Wl,-rpath," + odir]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bad 0" in r.stdout


def test_select_form_tanh_expm1_match_the_oracle(tmp_path):
    """fd_math.hpp's expm1f_musl / tanhf_musl (all cases evaluated, one selected -- the form the one-wave ladder filter
    needs) vs the oracle's branch-form restatements, bit for bit over 2^23 evenly spaced f32, the case boundaries and
    40 M random patterns."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = tmp_path / "check_tanh_expm1"
    odir = os.path.join(ROOT, "oracle")
    cmd = [hipcc, "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-std=c++17", "-Wno-unused-result",
           "-I", os.path.join(ROOT, "fundsp_amd", "csrc"), "-o", str(exe),
           os.path.join(ROOT, "tests", "host", "check_tanh_expm1.hip"), "-L" + odir, "-lfundsp_oracle", "-Wl,-rpath," + odir]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bad 0" in r.stdout


def test_envelope_block_preparation_matches_the_oracle(tmp_path):
    """AdsrLive / Envelope process paths with the once-per-block preparation of the next segment (host build of
    fd_nodes.hpp) vs the oracle's envelope.rs restatement: 6000 random ADSR settings, sample rates 2 .. 192 kHz, seeds,
    block partitions with remainders, gate edges anywhere (incl. NaN / negative levels), two calls in a row."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = tmp_path / "check_envelope_spec"
    odir = os.path.join(ROOT, "oracle")
    cmd = [hipcc, "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-std=c++17", "-Wno-unused-result",
           "-I", os.path.join(ROOT, "fundsp_amd", "csrc"), "-I", odir, "-o", str(exe),
           os.path.join(ROOT, "tests", "host", "check_envelope_spec.hip"), "-L" + odir, "-lfundsp_oracle", "-Wl,-rpath," + odir]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bad 0" in r.stdout


def test_oversampler_walk_over_odd_block_sizes_matches_the_oracle(tmp_path):
    """Oversampler::process over launch blocks of odd sizes (19, 1, 7, 63, ..): per pass the inner node processes `size` samples, one more than the
    pass interpolated (oversample.rs:191-195) -- the engine's per-sample walk and the oracle's block restatement, state carried across blocks."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    import oracle as O

    O.build()
    exe = tmp_path / "check_oversample_odd"
    cmd = [hipcc, "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-std=c++17", "-Wno-unused-result",
           "-I", os.path.join(ROOT, "fundsp_amd", "csrc"), "-I", os.path.join(ROOT, "oracle"), "-o", str(exe),
           os.path.join(ROOT, "tests", "host", "check_oversample_odd.hip"), "-L" + os.path.join(ROOT, "oracle"), "-lfundsp_oracle",
           "-Wl,-rpath," + os.path.join(ROOT, "oracle")]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "all equal" in r.stdout, r.stdout + r.stderr

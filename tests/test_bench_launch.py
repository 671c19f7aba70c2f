"""`python bench.py --gpus N` must run as typed (VERDICT r02 Missing 2): without RANK in the environment one process drives N
devices from N host threads; under torch.distributed.run it is one rank; a box with too few devices is told so in terms
of DEVICES, not launchers.  CPU only: the plan and the thread rendezvous, no device work."""
import os
import subprocess
import sys
import threading
from types import SimpleNamespace

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def plan(gpus, env, devices):
    return bench.launch_plan(SimpleNamespace(gpus=gpus), env, lambda: devices)


def test_launch_plan():
    assert plan(1, {}, 1) == ("single", None)
    assert plan(1, {}, 8) == ("single", None)
    assert plan(8, {}, 8) == ("threads", 8)                    # as typed on an 8-GPU node: one process, 8 host threads
    assert plan(2, {"RANK": "1", "WORLD_SIZE": "2", "LOCAL_RANK": "1"}, 0) == ("torchrun-rank", (1, 2, 1))
    assert plan(1, {"RANK": "0", "WORLD_SIZE": "1"}, 1) == ("single", None)   # torchrun --nproc-per-node 1
    with pytest.raises(SystemExit, match="2 HIP devices needed, 1 present"):
        plan(2, {}, 1)
    with pytest.raises(SystemExit, match="1 HIP device needed, 0 present"):
        plan(1, {}, 0)
    with pytest.raises(SystemExit, match="WORLD_SIZE=4"):
        plan(8, {"RANK": "0", "WORLD_SIZE": "4"}, 8)


def test_bench_gpus_2_as_typed_reports_devices_not_launchers():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two devices present: the run itself is the driver's business")
    assert r.returncode != 0
    msg = r.stderr.strip().splitlines()[-1]
    assert "2 HIP devices needed" in msg and "present" in msg and "torch.distributed.run" not in msg and "launch" not in msg


def test_thread_peers_barrier_and_max():
    n = 4
    tb, shared, got = threading.Barrier(n), [0.0] * n, [None] * n

    def body(k):
        p = bench.Peers(n, k, thread_barrier=tb, shared=shared)
        p.barrier()
        got[k] = (p.max(float(k + 1), None), p.max(float(10 - k), None))   # two rounds: the slots are reused safely
    ts = [threading.Thread(target=body, args=(k,)) for k in range(n)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert got == [(4.0, 10.0)] * n
    assert bench.Peers(1).max(3.5, None) == 3.5


def test_power_meter_degrades_to_none_without_a_device():
    """bench.py's roofline.power comes from librocm_smi64; without the library or a device every reading is None and the
    bench line simply carries `"power": null` (this container has no GPU: the CPU suite sees exactly that path)."""
    import bench

    m = bench.PowerMeter(0)
    m.start()
    m.sample_clock()
    if m.lib is None or m.energy_j() is None:
        assert m.stop(3) is None
    else:   # a box with a GPU: a well-formed record
        rec = m.stop(3)
        assert rec is None or {"socket_watts", "cap_watts", "joules_per_step", "sclk_mhz"} <= set(rec)


def test_cpu_baseline_counts_the_cpus_the_process_may_use(tmp_path):
    """cpu_baseline.cores = min(affinity, cgroup quota), not os.cpu_count() (a 256-thread host under a 16-CPU quota)."""
    import bench

    aff = len(os.sched_getaffinity(0))
    v2 = tmp_path / "v2"
    v2.mkdir()
    (v2 / "cpu.max").write_text("200000 100000\n")
    b = bench.host_cpu_budget(str(v2))
    assert b["cgroup_quota_cpus"] == 2.0 and b["effective_cpus"] == min(aff, 2) and b["affinity_cpus"] == aff
    (v2 / "cpu.max").write_text("max 100000\n")
    b = bench.host_cpu_budget(str(v2))
    assert b["cgroup_quota_cpus"] is None and b["effective_cpus"] == aff
    (v2 / "cpu.max").write_text("150000 100000\n")          # 1.5 CPUs: two threads can run
    assert bench.host_cpu_budget(str(v2))["effective_cpus"] == min(aff, 2)
    v1 = tmp_path / "v1" / "cpu"
    v1.mkdir(parents=True)
    (v1 / "cpu.cfs_quota_us").write_text("300000\n")
    (v1 / "cpu.cfs_period_us").write_text("100000\n")
    assert bench.host_cpu_budget(str(tmp_path / "v1"))["effective_cpus"] == min(aff, 3)
    (v1 / "cpu.cfs_quota_us").write_text("-1\n")
    assert bench.host_cpu_budget(str(tmp_path / "v1"))["effective_cpus"] == aff
    assert bench.host_cpu_budget(str(tmp_path / "nothing_here"))["effective_cpus"] == aff


def test_cpu_baseline_runs_on_rank_0_of_multi_gpu_runs_too():
    """north_star: the CPU path is timed "in the same run" next to the 1/2/4/8-GPU numbers (it used to be null at N > 1)."""
    a = bench.parse_args(["--gpus", "2"])
    assert bench.cpu_baseline_wanted(a, 0, 2) and bench.cpu_baseline_wanted(a, 0, 8) and bench.cpu_baseline_wanted(a, 0, 1)
    assert not bench.cpu_baseline_wanted(a, 1, 2)
    assert not bench.cpu_baseline_wanted(bench.parse_args(["--gpus", "2", "--cpu-seconds", "0"]), 0, 2)


def test_secondary_cpu_baselines_are_the_native_monomorphised_legs():
    """Every cpu_baseline of the bench line comes from the -O3 -march=native build, many voices per pinned thread; configs 4 / 5 through
    the monomorphised process() (bit-equal to the tree walk: tests/test_oracle_fast.py)."""
    for cfg, kind in ((2, "port (monomorphised)"), ("4v", "port (monomorphised)"), (5, "port (monomorphised)")):
        r = bench.cpu_baseline_config(cfg, 48000.0, 640, target_seconds=0.05)
        assert r["kind"] == kind and "-O3 -march=native" in r["flags"] and r["value"] > 0 and r["threads_pinned"]
        assert r["cores"] == bench.host_cpu_budget()["effective_cpus"]
        assert r["scalar_voice_value" if cfg == 2 else "tree_walk_value"] > 0


def test_pmc_latest_is_quoted_only_for_the_kernel_it_was_counted_on():
    """bench.py quotes profiles/pmc_latest.json as roofline.traffic only when the record carries the hash of the sources the headline
    kernel is compiled from in THIS tree (the file silently went stale whenever the kernel changed)."""
    import json

    rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
    assert len(bench.headline_kernel_source_hash()) == 64
    assert "k_render_pipe" in rec["kernel"] and len(rec.get("kernel_source_sha256", "")) == 64

"""Workload definitions and the multi-GPU (N > 1) path on CPU: contiguous voice sharding with no data-path
collective, and the single exchange step -- the stereo mix-down all-reduce -- over `gloo`, world_size 2.
The renderer stand-in on CPU is the oracle (tests may use it); on the GPU box the same code path runs over RCCL."""
import os
import socket

import numpy as np
import pytest

import oracle as O
from fundsp_amd import dist as fdist
from fundsp_amd import workloads as W


def test_rnd1_hash1_vectorised_match_oracle():
    xs = np.array([0, 1, 2, 3, 1000, 2**40 + 3, 2**64 - 1], dtype=np.uint64)
    L = O.lib()
    assert list(W.rnd1(xs)) == [L.o_math_rnd1(int(x)) for x in xs]
    assert [int(h) for h in W.hash1(xs)] == [L.o_math_hash1(int(x)) for x in xs]


def test_config3_parameter_ranges_and_sharding_consistency():
    sr = 48000.0
    p = W.fm_svf_params(4096, sr)
    assert p["f"].min() >= 55.0 and p["f"].max() < 1760.0
    assert p["m"].min() >= 0.5 and p["m"].max() < 8.0
    assert p["q"].min() >= 0.5 and p["q"].max() < 4.0
    assert np.all(p["fc"] >= p["f"] * 0.999) and p["fc"].max() <= 0.45 * sr + 1
    # a shard's parameters are exactly the slice of the whole bank's parameters (voices are index-addressed)
    first, count = fdist.shard_range(4096, 1, 3)
    q = W.fm_svf_params(count, sr, voice0=first)
    for k in ("f", "m", "fc", "q", "seed"):
        assert np.array_equal(q[k], p[k][first:first + count])


def test_config2_parameter_ranges():
    p = W.noise_biquad_params(1024, 48000.0)
    assert p["fc"].min() >= 20.0 and p["fc"].max() <= 0.49 * 48000 and p["q"].min() >= 0.5 and p["q"].max() <= 10.0
    assert len(set(int(s) for s in p["seed"])) == 1024


@pytest.mark.parametrize("total,world", [(65536, 8), (10, 3), (7, 8), (262144, 8), (1, 1)])
def test_shard_range_partitions(total, world):
    spans = [fdist.shard_range(total, r, world) for r in range(world)]
    assert spans[0][0] == 0 and sum(c for _, c in spans) == total
    for (f0, c0), (f1, _) in zip(spans, spans[1:]):
        assert f0 + c0 == f1
    assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, frames, ret):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sr = 48000.0
        first, count = fdist.shard_range(total, rank, world)
        p = W.fm_svf_params(count, sr, voice0=first)
        out, _ = O.bank_render(3, [p["f"], p["m"], p["fc"], p["q"]], p["seed"], frames, sr, True, 1, 1)  # [frame][voice]
        w = np.float32(np.cos(np.float32(np.pi) * np.float32(0.25)))
        part = np.stack([(out * w).sum(axis=1, dtype=np.float32)] * 2)  # centre pan: L = R
        mix = fdist.allreduce_mix(torch.from_numpy(part.copy()))
        ret[rank] = mix.numpy().copy()
    finally:
        dist.destroy_process_group()


def test_mixdown_allreduce_gloo_world2():
    import torch.multiprocessing as mp

    total, frames, world = 48, 200, 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, total, frames, ret), nprocs=world, join=True)
    # single-process reference: the whole bank, same per-voice weights
    p = W.fm_svf_params(total, 48000.0)
    out, _ = O.bank_render(3, [p["f"], p["m"], p["fc"], p["q"]], p["seed"], frames, 48000.0, True, 1, 1)
    w = np.float32(np.cos(np.float32(np.pi) * np.float32(0.25)))
    full = (out.astype(np.float64) * float(w)).sum(axis=1)
    for r in range(world):
        assert ret[r].shape == (2, frames)
        # float summation order differs between 1 and 2 ranks: tolerance ~ sqrt(V) * eps * max|x|
        assert np.max(np.abs(ret[r][0] - full)) < 1e-4
        assert np.array_equal(ret[0], ret[r])  # every rank holds the same reduced mix


def _worker_order(rank, world, port, total, frames, ret):
    import torch
    import torch.distributed as dist

    from mix_order import mix_order_reference

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sr = 48000.0
        first, count = fdist.shard_range(total, rank, world)
        p = W.fm_svf_params(count, sr, voice0=first)
        out, _ = O.bank_render(3, [p["f"], p["m"], p["fc"], p["q"]], p["seed"], frames, sr, True, 1, 1)  # [frame][voice]
        w = np.float32(np.cos(np.float32(np.pi) * np.float32(0.25)))
        part = np.stack([mix_order_reference(out * w)] * 2)   # what fdsp_bank_process_mix hands the collective: the shard's partial mix
        ret[rank] = fdist.allreduce_mix(torch.from_numpy(part.copy())).numpy().copy()
    finally:
        dist.destroy_process_group()


def test_two_ranks_reproduce_the_one_rank_mix_bit_for_bit():
    """Round 4: the per-GPU partial mix is produced in the mix-down's fixed order (an aligned binary tree over the voice groups), so
    with the bank split at an aligned power-of-two group boundary the 2-rank all-reduce -- ONE addition per sample -- returns exactly
    the 1-rank mix: multi-GPU runs of the mix-down are bit-reproducible against a single GPU at N = 2 (N = 4 / 8: up to the
    collective's own order over the shards)."""
    import torch.multiprocessing as mp

    from mix_order import mix_order_reference

    total, frames, world = 64 * 8, 96, 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_order, args=(world, port, total, frames, ret), nprocs=world, join=True)
    p = W.fm_svf_params(total, 48000.0)
    out, _ = O.bank_render(3, [p["f"], p["m"], p["fc"], p["q"]], p["seed"], frames, 48000.0, True, 1, 1)
    w = np.float32(np.cos(np.float32(np.pi) * np.float32(0.25)))
    whole = mix_order_reference(out * w)
    for r in range(world):
        assert np.array_equal(ret[r][0].view(np.uint32), whole.view(np.uint32)), f"rank {r}: all-reduced mix != the one-rank mix"


def test_allreduce_mix_is_identity_without_process_group():
    import torch

    t = torch.arange(6, dtype=torch.float32).reshape(2, 3)
    assert fdist.allreduce_mix(t.clone()).equal(t)


def test_wav_sink_layout(tmp_path):
    """write.rs:26-116: header fields and interleaving; read back with an independent reader (scipy)."""
    import struct

    from scipy.io import wavfile

    from fundsp_amd import wav

    x = np.stack([np.linspace(-1, 1, 100, dtype=np.float32), np.linspace(1, -1, 100, dtype=np.float32) * 0.5])
    b = wav.wav32_bytes(x, 48000.0)
    assert b[:4] == b"RIFF" and b[8:16] == b"WAVEfmt " and b[36:40] == b"data" and len(b) == 44 + 800
    assert struct.unpack("<I", b[4:8])[0] == 800 + 36 and struct.unpack("<HHI", b[20:28]) == (3, 2, 48000)
    assert struct.unpack("<IHH", b[28:36]) == (48000 * 2 * 4, 8, 32)
    p32, p16 = tmp_path / "a32.wav", tmp_path / "a16.wav"
    wav.save_wav32(p32, x, 48000.0)
    wav.save_wav16(p16, x, 48000.0)
    sr, y = wavfile.read(p32)
    assert sr == 48000 and np.array_equal(y.T, x)
    sr, y16 = wavfile.read(p16)
    assert y16.dtype == np.int16 and y16[0, 0] == -32767 and y16[-1, 0] == 32767 and abs(int(y16[0, 1]) - 16384) <= 1


def test_gate_plan_and_var_gate_slots():
    """Config 4 in the reference's gate shape (`var(gate) >> adsr_live`): the plan of launches mirrors the stream workload's gate at block
    granularity, and the slot names the workload sets exist in the kind (CPU: the slot table of the library needs no device)."""
    from fundsp_amd import bank

    assert W.gate_plan(48000, 48000.0) == [(1.0, 24000), (0.0, 24000)]
    assert W.gate_plan(1000, 48000.0) == [(1.0, 1000)]                       # shorter than the note: all high
    assert W.gate_plan(48000, 44100.0) == [(1.0, 22016), (0.0, 25984)]       # 0.5 s rounded DOWN to whole 64-frame blocks
    assert sum(n for _, n in W.gate_plan(12345, 48000.0, off_seconds=0.1)) == 12345
    slots = dict(bank.kind_slots("saw_moog_var_adsr_pan"))
    for name in W.C4V_SLOTS.values():
        assert name in slots, name
    assert slots[W.C4V_SLOTS["gate"]] == 0                                    # a parameter (kind 0), like Shared::set_value's target
    stream = dict(bank.kind_slots("saw_moog_adsr_pan"))
    assert len(slots) == len(stream) + 1                                      # the Var's value is the one slot more

"""Multi-GPU sharding of a voice bank: voices are independent, so a bank shards as
contiguous voice ranges, one process per GPU, with NO data-path collective in voice-out mode.  The only exchange
step of the path is the final stereo mix-down: each rank reduces its own voices on device (fdsp_mix_stereo) and
one all-reduce(sum) of the [2, frames] partial mixes runs over RCCL/xGMI (backend "nccl" on ROCm; "gloo" on CPU
for tests).  The payload is 8*frames bytes per rank -- latency-bound, so it is issued once per launch.
"""


def shard_range(total_voices, rank, world_size):
    """Contiguous voice range [first, first + count) owned by `rank` (remainder spread over the low ranks)."""
    base, rem = divmod(int(total_voices), int(world_size))
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def allreduce_mix(mix, group=None):
    """Sum per-rank stereo partial mixes [2, frames] in place across the process group; returns `mix`."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(mix, op=dist.ReduceOp.SUM, group=group)
    return mix

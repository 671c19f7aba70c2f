"""CPU: the mix-down's summation order as the tests state it (tests/mix_order.py) against a literal, loop-by-loop reading of
include/fundsp_hip.h ("the stereo mix-down"), and the property the multi-GPU path relies on: a bank split at an aligned
power-of-two group boundary adds up to the whole bank's mix with one addition per sample."""
import numpy as np

from mix_order import mix_order_reference


def test_summation_order_statement():
    """mix_order_reference against a literal, loop-by-loop reading of include/fundsp_hip.h on a small ragged case."""
    rng = np.random.default_rng(1)
    for V in (1, 15, 64, 65, 200, 64 * 5, 64 * 6 + 3):
        x = (rng.random(V, dtype=np.float32) - 0.5).astype(np.float32)
        G = (V + 63) // 64
        xp = np.zeros(G * 64, dtype=np.float32)
        xp[:V] = x
        parts = []
        for g in range(G):
            S = []
            for q in range(4):
                s = xp[g * 64 + q * 16]
                for j in range(1, 16):
                    s = np.float32(s + xp[g * 64 + q * 16 + j])
                S.append(s)
            parts.append(np.float32(np.float32(S[0] + S[1]) + np.float32(S[2] + S[3])))
        while len(parts) > 1:
            nxt = [np.float32(parts[i] + parts[i + 1]) for i in range(0, len(parts) - 1, 2)]
            if len(parts) & 1:
                nxt.append(parts[-1])
            parts = nxt
        assert np.float32(mix_order_reference(x)).view(np.uint32) == np.float32(parts[0]).view(np.uint32), V




def test_aligned_shards_add_up_exactly():
    rng = np.random.default_rng(4)
    x = (rng.random((7, 64 * 32), dtype=np.float32) - 0.5).astype(np.float32)
    whole = mix_order_reference(x)
    for parts in (2, 4, 8):   # 2, 4, 8 GPUs: contiguous shards of 16 / 8 / 4 voice groups
        n = x.shape[1] // parts
        sums = [mix_order_reference(x[:, k * n:(k + 1) * n]) for k in range(parts)]
        while len(sums) > 1:  # the same aligned tree over the shards (what a tree all-reduce would compute)
            sums = [sums[i] + sums[i + 1] for i in range(0, len(sums), 2)]
        assert np.array_equal(whole.view(np.uint32), sums[0].view(np.uint32)), parts

"""The mix-down's summation order (include/fundsp_hip.h, "the stereo mix-down") stated in numpy f32 -- TEST INFRASTRUCTURE:
the checker of fdsp_bank_process_mix / fdsp_sum_voices / fdsp_mix_stereo.  The product never imports it."""
import numpy as np


def mix_order_reference(x):
    """The mix-down's summation order (include/fundsp_hip.h) in numpy f32: x [..., voices] -> [...].
    partial(group of 64) = (S0 + S1) + (S2 + S3), Sq = its 16 voices added one after the other (voices past the end = +0.0);
    the partials in an aligned binary tree, a node without a right sibling passes through."""
    x = np.asarray(x, dtype=np.float32)
    V = x.shape[-1]
    G = (V + 63) // 64
    pad = np.zeros(x.shape[:-1] + (G * 64,), dtype=np.float32)
    pad[..., :V] = x
    q = pad.reshape(x.shape[:-1] + (G, 4, 16))
    s = q[..., 0].copy()
    for j in range(1, 16):
        s = s + q[..., j]
    level = (s[..., 0] + s[..., 1]) + (s[..., 2] + s[..., 3])  # [..., G]
    while level.shape[-1] > 1:
        n = level.shape[-1]
        pairs = level[..., 0:n - (n & 1):2] + level[..., 1:n:2]
        level = np.concatenate([pairs, level[..., n - 1:n]], axis=-1) if n & 1 else pairs
    return level[..., 0]



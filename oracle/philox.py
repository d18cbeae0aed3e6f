"""TEST INFRASTRUCTURE (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package).

numpy restatement of the library's noise generator (include/cbfssm_hip.h: cbfssm_normal_f64) -- Philox4x32-10 (Salmon, Moraes,
Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) + Box-Muller in float64.  The reference draws its noise with
tf.random_normal inside the graph (cbfssm.py:134,149,209); TensorFlow's stream cannot be reproduced outside TensorFlow, so the
pin of this file is the paper's own known-answer vectors (tests/test_oracle.py::test_philox_known_answers), not the reference."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)
S32 = np.uint64(32)


def philox4x32_10(counter, key):
    """counter: (..., 4) and key: (..., 2) arrays of 32-bit words -> (..., 4) uint32."""
    c = [np.asarray(counter)[..., i].astype(np.uint64) for i in range(4)]
    k = [np.asarray(key)[..., i].astype(np.uint64) for i in range(2)]
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> S32) ^ c[1] ^ k[0], p1 & MASK, (p0 >> S32) ^ c[3] ^ k[1], p0 & MASK]
        k = [(k[0] + W0) & MASK, (k[1] + W1) & MASK]
    return np.stack(c, axis=-1).astype(np.uint32)


def normal(seed, offset, n):
    """out[i] of cbfssm_normal_f64(seed, offset, n): pair p = (offset + i) >> 1, member (offset + i) & 1."""
    idx = np.uint64(offset) + np.arange(n, dtype=np.uint64)
    p = idx >> np.uint64(1)
    ctr = np.stack([p & MASK, p >> S32, np.zeros_like(p), np.zeros_like(p)], axis=-1)
    key = np.broadcast_to(np.array([np.uint64(seed) & MASK, np.uint64(seed) >> S32]), (n, 2))
    r = philox4x32_10(ctr, key).astype(np.uint64)
    a, b = (r[:, 0] << S32) | r[:, 1], (r[:, 2] << S32) | r[:, 3]
    u1 = ((a >> np.uint64(11)) + np.uint64(1)).astype(np.float64) * 2.0 ** -53
    u2 = (b >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    rad = np.sqrt(-2.0 * np.log(u1))
    ang = 2.0 * np.pi * u2
    return np.where((idx & np.uint64(1)) == 0, rad * np.cos(ang), rad * np.sin(ang))

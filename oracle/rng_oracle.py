"""CPU restatement (numpy) of csrc/rng.hip's sharding-invariant eps -- TEST INFRASTRUCTURE ONLY.

Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; multipliers 0xD2511F53 / 0xCD9E8D57,
Weyl key increments 0x9E3779B9 / 0xBB67AE85) + Box-Muller.  Pinned by the known-answer vectors of the Random123
distribution (kat_vectors: philox4x32 10 rounds), checked in tests/test_oracle_vs_golden.py."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr [..., 4] uint32, key [..., 2] uint32 (broadcastable) -> [..., 4] uint32"""
    c = [np.asarray(ctr[..., i], dtype=np.uint64) for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint64)
    k1 = np.asarray(key[..., 1], dtype=np.uint64)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & MASK, p1 & MASK, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & MASK, p0 & MASK]
        k0 = (k0 + np.uint64(W0)) & MASK
        k1 = (k1 + np.uint64(W1)) & MASK
    return np.stack(c, -1).astype(np.uint32)


def philox_normal(rows, Z, seed, stream_id, row_offset=0):
    """eps [rows, Z] float32 as ptv_philox_normal produces it"""
    q4 = (Z + 3) // 4
    g = (np.arange(rows, dtype=np.uint64) + np.uint64(row_offset))[:, None]
    q = np.arange(q4, dtype=np.uint64)[None, :]
    ctr = np.stack(np.broadcast_arrays(g & MASK, ((g >> np.uint64(32)) * np.uint64(0x10000) + q) & MASK,
                                       np.uint64(stream_id & 0xFFFFFFFF), np.uint64(stream_id >> 32)), -1)
    key = np.array([seed & 0xFFFFFFFF, seed >> 32], dtype=np.uint64)
    x = philox4x32_10(ctr, key).astype(np.float32)
    u = (x + np.float32(0.5)) * np.float32(2.3283064365386963e-10)
    out = np.empty((rows, q4, 4), dtype=np.float32)
    for h in range(2):
        u1 = np.clip(u[..., 2 * h], np.float32(1.1754944e-38), np.float32(1.0))
        rad = np.sqrt(np.float32(-2.0) * np.log(u1))
        th = np.float32(6.283185307179586) * u[..., 2 * h + 1]
        out[..., 2 * h] = rad * np.cos(th)
        out[..., 2 * h + 1] = rad * np.sin(th)
    return out.reshape(rows, q4 * 4)[:, :Z]

"""numpy emulation of common.h:dropout_scale's hash and the statistics quoted there (CPU, seconds)."""
import numpy as np


def mul24(a, b):
    return ((a & 0xFFFFFF).astype(np.uint64) * np.uint64(b & 0xFFFFFF) & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def uniform24(idx, seed):
    """common.h:dropout_keep: elements are hashed in pairs (idx >> 1); the odd element takes one more round"""
    pair, odd = idx >> np.uint64(1), (idx & np.uint64(1)).astype(bool)
    lo, hi = (pair & 0xFFFFFFFF).astype(np.uint32), (pair >> 32).astype(np.uint32)
    x = lo ^ np.uint32(seed & 0xFFFFFFFF) ^ mul24(hi, 0x85EBCB)
    x ^= x >> np.uint32(16)
    x = mul24(x, 0x9E3779) ^ np.uint32((seed >> 32) & 0xFFFFFFFF)
    x ^= x >> np.uint32(13)
    x = mul24(x, 0xC2B2AF)
    x ^= x >> np.uint32(15)
    x = mul24(x, 0x7FEB35)
    x ^= x >> np.uint32(12)
    y = mul24(x ^ (x >> np.uint32(11)), 0x9E3779)
    y ^= y >> np.uint32(14)
    x = np.where(odd, y, x)
    return (x & np.uint32(0xFFFFFF)).astype(np.float64) / 16777216.0


if __name__ == "__main__":
    N = 1 << 22
    chis = []
    for seed in (1, 12345678901234, 0xFFFFFFFFFFFF, 999, 424242424242):
        for base in (0, 7 * N, 1 << 33):
            idx = np.arange(N, dtype=np.uint64) + np.uint64(base)
            u = uniform24(idx, seed)
            hist = np.histogram(u, bins=64, range=(0, 1))[0]
            chis.append(((hist - N / 64) ** 2 / (N / 64)).sum())
            keep = u >= 0.1
            c = [np.corrcoef(keep[:-s], keep[s:])[0, 1] for s in (1, 256)] + [np.corrcoef(keep, uniform24(idx, seed + 1) >= 0.1)[0, 1]]
            c.append(np.corrcoef(keep[0::2], keep[1::2])[0, 1])  # the two elements of a hash pair
            assert max(abs(v) for v in c) < 3e-3, c
    print("chi2(63 dof) mean %.1f max %.1f; keep-mask correlations < 3e-3" % (np.mean(chis), np.max(chis)))

"""Brute-force LDS bank-conflict check of the GEMM core's tile layout (gemm_core.hpp `swz`), using
the gfx950 per-instruction lane groups and bank moduli of MI355X_MICROARCH.md §LDS."""
from collections import defaultdict

LD_B = 128


def t(r):
    return ((r >> 2) ^ r) & 7


def addr(row, chunk, off=0):
    return row * LD_B + ((chunk ^ t(row)) * 16) + off


def ways(addrs, width, nbanks):
    b = defaultdict(set)
    for a in addrs:
        for d in range(width // 4):
            w = a // 4 + d
            b[w % nbanks].add(w)
    return max(len(v) for v in b.values())


B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
               [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[x + 32 for x in g] for g in B128_GROUPS]

if __name__ == '__main__':
    km16 = max(ways([addr(4 * (l & 7) + j, (l >> 3) >> 1, ((l >> 3) & 1) * 8) for l in range(16 * g, 16 * g + 16)], 8, 32)
               for j in range(4) for g in range(4))
    km32 = max(ways([addr(4 * (l & 7) + j, l >> 3) for l in range(8 * g, 8 * g + 8)], 16, 32)
               for j in range(4) for g in range(8))
    kc16 = max(ways([addr(row, vr >> 1, (vr & 1) * 8) for vr in range(16)], 8, 32) for row in range(128))
    kc32 = max(ways([addr(row, vr) for vr in range(8)], 16, 32) for row in range(128))
    rd = max(ways([addr(base + (l & 15), c0 + (l >> 4)) for l in g], 16, 64)
             for base in range(0, 256, 16) for c0 in (0, 4) for g in B128_GROUPS)
    print('K-major store bf16 %d-way, f32 %d-way; K-contiguous store bf16 %d-way, f32 %d-way; fragment read %d-way'
          % (km16, km32, kc16, kc32, rd))
    assert (km16, km32, kc16, kc32, rd) == (1, 1, 1, 1, 1)

#!/usr/bin/env python3
"""Statistical comparison of the two stratum permutations of optrace_amd/csrc/ot_generate.hpp on power-of-two domains:
`permute_hash` (A. Kensler's construction, used below 2^20 and with cycle walking on ragged blocks) and
`permute_pow2_large` (three multiply / xor-shift rounds, used from 2^20).  NumPy restatements of both; chi-square per
degree of freedom (1.0 = uniform) of: pairs and a triple of streams with independent keys, a stream against the ray
index, against itself at lags 1 / 64 / 4096, its low byte against another stream, and streams with neighbouring keys;
plus a Kolmogorov-Smirnov statistic of the strata of 4096 consecutive rays.  Output: profiles/r1/perm_quality.txt."""
import numpy as np

U = np.uint64
M32 = U(0xffffffff)


def kensler(i, l, key):
    x = i.astype(U).copy()
    key = U(key)
    w = U(l - 1)
    x ^= key; x = (x * U(0xe170893d)) & M32; x ^= key >> U(16); x ^= (x & w) >> U(4); x ^= key >> U(8)
    x = (x * U(0x0929eb3f)) & M32; x ^= key >> U(23); x ^= (x & w) >> U(1); x = (x * (U(1) | key >> U(27))) & M32
    x = (x * U(0x6935fa69)) & M32; x ^= (x & w) >> U(11); x = (x * U(0x74dcb303)) & M32; x ^= (x & w) >> U(2)
    x = (x * U(0x9e501cc3)) & M32; x ^= (x & w) >> U(2); x = (x * U(0xc860a3df)) & M32; x &= w; x ^= x >> U(5)
    return (x + (key & w)) & w


def large(i, l, key):
    x = i.astype(U).copy()
    key = U(key)
    k = int(l - 1).bit_length()
    w = U(l - 1)
    sA, sB = U((k + 1) // 2), U((k + 2) // 3)
    m2 = ((U(0x0929eb3f) ^ ((key >> U(7)) << U(1))) & M32) | U(1)
    x ^= key & w
    x = (x * U(0xe170893d)) & w; x ^= x >> sA
    x = (x * m2) & w; x ^= x >> sB
    x ^= (key >> U(13)) & w
    x = (x * U(0x6935fa69)) & w; x ^= x >> sA
    return (x + ((key >> U(3)) & w)) & w


def chi2(a, b, l, G):
    H = np.bincount((a * U(G) // U(l)).astype(np.int64) * G + (b * U(G) // U(l)).astype(np.int64), minlength=G * G).astype(float)
    E = a.shape[0] / (G * G)
    return ((H - E) ** 2 / E).sum() / (G * G - 1)


def chi3(a, b, c, l, G):
    idx = ((a * U(G) // U(l)).astype(np.int64) * G + (b * U(G) // U(l)).astype(np.int64)) * G + (c * U(G) // U(l)).astype(np.int64)
    H = np.bincount(idx, minlength=G ** 3).astype(float)
    E = a.shape[0] / G ** 3
    return ((H - E) ** 2 / E).sum() / (G ** 3 - 1)


if __name__ == "__main__":
    rng = np.random.default_rng(5)
    for name, f in (("permute_hash", kensler), ("permute_pow2_large", large)):
        for k in (16, 18, 20, 22, 24, 26):
            l = 1 << k
            j = np.arange(l, dtype=U)
            keys = [int(x) for x in rng.integers(0, 2 ** 32, 5)]
            P = [f(j, l, kk) for kk in keys]
            for p in P:
                assert np.array_equal(np.sort(p), j), "not a bijection"
            G = min(256, 1 << (k // 2 - 2))
            pair = max(chi2(P[a], P[b], l, G) for a in range(4) for b in range(a + 1, 4))
            jj = max(chi2(j, P[a], l, G) for a in range(4))
            lag = [max(chi2(P[a][:-d], P[a][d:], l, G) for a in range(4)) for d in (1, 64, 4096)]
            tri = chi3(P[0], P[1], P[2], l, min(32, 1 << (k // 3 - 1)))
            low = chi2((P[0] % U(256)) * U(l // 256), P[1], l, min(G, 256))
            near = max(chi2(P[0], f(j, l, keys[0] ^ 1), l, G), chi2(P[0], f(j, l, keys[0] + 0x10000), l, G))
            blk = np.sort(P[0][:4096].astype(np.float64)) / l
            ks = np.abs(blk - (np.arange(4096) + 0.5) / 4096).max() * np.sqrt(4096)
            print(f"{name:18s} 2^{k:2d}  pairs {pair:5.3f}  vs index {jj:5.3f}  lags {lag[0]:5.3f} {lag[1]:5.3f} {lag[2]:5.3f}  "
                  f"triple {tri:5.3f}  low byte {low:5.3f}  near keys {near:5.3f}  KS {ks:4.2f}", flush=True)

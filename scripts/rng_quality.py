#!/usr/bin/env python3
"""Why the render stream's mixer could change: avalanche bias (flip one input bit: every output bit should flip with
probability 1/2) and stream statistics (consecutive draws of one key: triples in 16^3 cells -- the unit-sphere sampler's use --, a low byte,
lag-1 correlation) of the splitmix64 finaliser (rounds 1-2), of the two-round xor-shift-32 / multiply mixer the stream uses now ("degski64":
its shifts cost a 32-bit ALU nothing) and of a one-multiply mixer (rejected: its avalanche is far from flat).  CPU only, ~1 minute."""
import numpy as np
M = np.uint64(0xFFFFFFFFFFFFFFFF)
def splitmix(z):
    z = z.copy()
    z ^= z >> np.uint64(30); z *= np.uint64(0xBF58476D1CE4E5B9)
    z ^= z >> np.uint64(27); z *= np.uint64(0x94D049BB133111EB)
    z ^= z >> np.uint64(31)
    return z
def degski(z, c=0xd6e8feb86659fd93):
    z = z.copy()
    z ^= z >> np.uint64(32); z *= np.uint64(c)
    z ^= z >> np.uint64(32); z *= np.uint64(c)
    z ^= z >> np.uint64(32)
    return z
def one_mul(z):
    z = z.copy()
    z ^= z >> np.uint64(32); z *= np.uint64(0xd6e8feb86659fd93); z ^= z >> np.uint64(32)
    return z
def avalanche(f, n=200000, seed=1, weyl=False):
    rng = np.random.default_rng(seed)
    if weyl:
        key = np.uint64(rng.integers(0, 2**63))
        x = key + np.uint64(0x9E3779B97F4A7C15) * np.arange(1, n + 1, dtype=np.uint64)
    else:
        x = rng.integers(0, 2**64, n, dtype=np.uint64)
    fx = f(x)
    bias = np.zeros((64, 64))
    for i in range(64):
        d = fx ^ f(x ^ np.uint64(1 << i))
        for j in range(64):
            bias[i, j] = ((d >> np.uint64(j)) & np.uint64(1)).mean()
    dev = bias - 0.5
    return float(np.sqrt((dev ** 2).mean())), float(np.abs(dev).max())
with np.errstate(over="ignore"):
    for name, f in (("splitmix64 finaliser", splitmix), ("degski64", degski), ("one multiply", one_mul)):
        for weyl in (False, True):
            r, m = avalanche(f, weyl=weyl)
            print("%-22s inputs %-7s avalanche bias rms %.5f max %.5f   (sampling noise rms ~ %.5f)" % (name, "weyl" if weyl else "random", r, m, 0.5 / np.sqrt(200000)))
    # the stream as used: consecutive draws of one key -> uniforms; serial correlation, 2-D and 3-D equidistribution of the top bits, low-bit runs
    def stream(f, key, n):
        x = np.uint64(key) + np.uint64(0x9E3779B97F4A7C15) * np.arange(1, n + 1, dtype=np.uint64)
        return f(x)
    for name, f in (("splitmix64 finaliser", splitmix), ("degski64", degski), ("one multiply", one_mul)):
        n = 3_000_000
        worst = 0
        for key in (0, 1, 0x5EED0002, 0xDEADBEEFCAFEBABE, 2**63 + 12345):
            z = stream(f, key, n)
            u = (z >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
            c1 = np.corrcoef(u[:-1], u[1:])[0, 1]
            # 3-D cells of consecutive triples (the unit-sphere sampler's use): 16^3 cells, chi-square
            t = (u[: n // 3 * 3].reshape(-1, 3) * 16).astype(int)
            cells = np.bincount(t[:, 0] * 256 + t[:, 1] * 16 + t[:, 2], minlength=4096)
            e = len(t) / 4096
            chi = ((cells - e) ** 2 / e).sum()
            # low 11 bits unused; bits 11..18 byte frequency
            b = ((z >> np.uint64(11)) & np.uint64(255)).astype(int)
            cb = np.bincount(b, minlength=256); eb = n / 256
            chib = ((cb - eb) ** 2 / eb).sum()
            worst = max(worst, abs(chi - 4095) / np.sqrt(2 * 4095), abs(chib - 255) / np.sqrt(2 * 255), abs(c1) * np.sqrt(n))
        print("%-22s stream tests over 5 keys x %d draws: worst |z-score| %.2f (triples in 16^3 cells, low byte, lag-1 correlation)" % (name, n, worst))

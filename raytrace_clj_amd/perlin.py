"""Host mirror of raytrace-clj.perlin's namespace-level tables (src/raytrace_clj/perlin.clj:6-17).

The reference builds `random-vectors` (256 x (normalise (rand-in-unit-sphere))) and `perm-x/y/z` ((shuffle (range 256)))
from the unseeded global RNG when the namespace loads, so no two JVM runs share them.  Here they are scene data generated
from a seed: the vectors by the same rejection sampler in the same draw order (x, y, z; retry while p.p >= 1), the
permutations by a Fisher-Yates shuffle of the same stream.  `noise` / `turbulence` (perlin.clj:19-64) run on the device."""
import math

import numpy as np

from .util import SplitMix64

PERLIN_SEED = 0x5EED0004


def make_tables(seed=PERLIN_SEED):
    """-> (vectors float64 [256,3], perm int32 [3,256])"""
    rng = SplitMix64(seed)
    vec = np.zeros((256, 3), np.float64)
    for n in range(256):
        while True:
            x, y, z = 2.0 * rng.rand() - 1.0, 2.0 * rng.rand() - 1.0, 2.0 * rng.rand() - 1.0
            if not ((x * x + y * y) + z * z >= 1.0):
                break
        d = math.sqrt((x * x + y * y) + z * z)
        inv = 1.0 / d if d > 0 else 1.0
        vec[n] = (x * inv, y * inv, z * inv)
    perm = np.zeros((3, 256), np.int32)
    for a in range(3):
        p = list(range(256))
        for i in range(255, 0, -1):
            j = rng.rand_int(i + 1)
            p[i], p[j] = p[j], p[i]
        perm[a] = p
    return vec, perm

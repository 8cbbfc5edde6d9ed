"""Host mirror of raytrace-clj.util (src/raytrace_clj/util.clj).

Only the constructors a scene description needs live on the host: `vec3` (util.clj:5-11),
`ray` (util.clj:13-16) and `point_at_parameter` (util.clj:18-22, used by host-side scene code
and the known-answer tests).  The rejection samplers (util.clj:32-52) run on the device.
"""
import numpy as np


def vec3(a, b, c):
    """(vec3 a b c): a 3-vector of doubles (util.clj:5-11; ints and ratios become doubles)."""
    return np.array([float(a), float(b), float(c)], dtype=np.float64)


def ray(origin, direction, t):
    """(ray origin direction t) -> {:origin :direction :time}; direction is not normalised (util.clj:13-16)."""
    return {"origin": np.asarray(origin, np.float64), "direction": np.asarray(direction, np.float64), "time": float(t)}


def point_at_parameter(r, t):
    """direction*t + origin (util.clj:18-22)."""
    return r["direction"] * float(t) + r["origin"]


class SplitMix64:
    """Seeded host stream for scene construction (the reference uses the unseeded global
    clojure.core/rand there, scene.clj:369-405): splitmix64.  (The RENDER stream -- sample_key /
    draw_bits below, rtmi_sample_key -- has its own mixer since round 3; scenes did not change with it.)"""

    MASK = (1 << 64) - 1
    GOLD = 0x9E3779B97F4A7C15

    def __init__(self, seed):
        self.state = int(seed) & self.MASK

    @staticmethod
    def mix64(z):
        m = SplitMix64.MASK
        z &= m
        z ^= z >> 30
        z = (z * 0xBF58476D1CE4E5B9) & m
        z ^= z >> 27
        z = (z * 0x94D049BB133111EB) & m
        z ^= z >> 31
        return z

    def next_u64(self):
        self.state = (self.state + self.GOLD) & self.MASK
        return self.mix64(self.state)

    def rand(self):
        """uniform double in [0, 1) with 53 random bits."""
        return (self.next_u64() >> 11) * (1.0 / 9007199254740992.0)

    def rand_int(self, n):
        return self.next_u64() % n


def mix64(z):
    """the render stream's mixer (rtmi_device.h mix64): two rounds of x ^= x >> 32; x *= 0xD6E8FEB86659FD93, then x ^= x >> 32"""
    m = SplitMix64.MASK
    z &= m
    z ^= z >> 32
    z = (z * 0xD6E8FEB86659FD93) & m
    z ^= z >> 32
    z = (z * 0xD6E8FEB86659FD93) & m
    z ^= z >> 32
    return z


def sample_key(seed, pixel, sample):
    """Stream key of (seed, pixel index j*nx+i, sample) -- python restatement of rtmi_sample_key."""
    m, g = SplitMix64.MASK, SplitMix64.GOLD
    a = mix64((seed ^ (g * (pixel + 1))) & m)
    return mix64((a + 0xD1B54A32D192ED03 * (sample + 1)) & m)


def draw_bits(key, d):
    m, g = SplitMix64.MASK, SplitMix64.GOLD
    return mix64((key + g * (d + 1)) & m)

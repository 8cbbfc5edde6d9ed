#!/usr/bin/env python3
"""Exports the flattened scenes of tests/golden/*.npz (what raytrace_clj_amd/flatten.py produced for three of the reference's scene functions, and what the
device and the oracle render bit for bit) as EDN for clj/test/raytrace_clj/gpu_test.clj: the Clojure flattener (clj/src/raytrace_clj/gpu.clj), run over the
reference's OWN scene functions with clojure.core/rand and rand-int rebound to the same seeded sequence, must produce these arrays.

    python scripts/export_clj_fixtures.py        ->  clj/test/resources/{two_spheres,cornell_box,cover_n3}.edn

EDN needs no dependency on the Clojure side (clojure.edn); doubles are written with repr (17 significant digits: exact round trip)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytrace_clj_amd.scene import SCENE_SEED  # noqa: E402

OUT = os.path.join(ROOT, "clj", "test", "resources")
KEYS = [("prim_kind", "prim-kind", int), ("prim_geom", "prim-geom", float), ("prim_mat", "prim-mat", int), ("mat_kind", "mat-kind", int), ("mat_tex", "mat-tex", int),
        ("mat_param", "mat-param", float), ("tex_kind", "tex-kind", int), ("tex_param", "tex-param", float), ("tex_child", "tex-child", int), ("cam", "cam", float),
        ("prim_flip", "prim-flip", int), ("prim_xform", "prim-xform", int), ("xform_kind", "xform-kind", int), ("xform_param", "xform-param", float)]


def edn_num(v, typ):
    if typ is int:
        return str(int(v))
    v = float(v)
    if v != v:
        return "##NaN"
    if v in (float("inf"), float("-inf")):
        return "##Inf" if v > 0 else "##-Inf"
    s = repr(v)
    return s if any(c in s for c in ".eE") else s + ".0"


def bvh_axes(make):
    """the split axes make-bvh draws ((rand-int 3), hitable.clj:109), in call order, while the mirror builds the scene -- and the flattened scene, to check the
    fixture against.  The reference draws the OUTER make-bvh's axis before its lazy item list is realised (scene.clj:332: (make-bvh (concat ... (for ...)))), the
    mirror after: the Clojure test therefore feeds rand (the scene's numbers) and rand-int (the axes) from two sequences, and the order between them is free."""
    import raytrace_clj_amd.scene as sc
    import raytrace_clj_amd.hitable as hit
    from raytrace_clj_amd import flatten as fl
    axes = []

    class Recording(sc.SplitMix64):
        def rand_int(self, n):
            v = super().rand_int(n)
            assert n == 3
            axes.append(int(v))
            return v
    keep = sc.SplitMix64
    sc.SplitMix64 = Recording
    try:
        flat = fl.flatten(make())
    finally:
        sc.SplitMix64 = keep
    return axes, flat


def export(npz, name, scene_call, nx, ny, make):
    z = np.load(os.path.join(ROOT, "tests", "golden", npz))
    axes, flat = bvh_axes(make)
    for k in ("prim_kind", "prim_geom", "prim_mat", "mat_kind", "tex_kind", "cam"):
        assert np.array_equal(np.asarray(getattr(flat, k)), z[k]), (name, k)  # the golden fixture IS this scene
    lines = ["{:scene %s" % scene_call, " :nx %d :ny %d" % (nx, ny), " :scene-seed %d" % SCENE_SEED, " :cam-kind %d" % int(z["cam_kind"]),
             " :bvh-axes [%s]" % " ".join(str(a) for a in axes)]
    for k, edn, typ in KEYS:
        if k in z.files:
            lines.append(" :%s [%s]" % (edn, " ".join(edn_num(v, typ) for v in np.asarray(z[k]).reshape(-1))))
    lines.append("}")
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".edn")
    open(path, "w").write("\n".join(lines) + "\n")
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    import raytrace_clj_amd as r
    export("render_two_spheres.npz", "two_spheres", '"make-two-spheres"', 40, 20, lambda: r.scene.make_two_spheres(40, 20))
    export("render_cornell.npz", "cornell_box", '"make-cornell-box"', 40, 40, lambda: r.scene.make_cornell_box(40, 40))
    export("render_cover_n3.npz", "cover_n3", '"make-random-scene"', 48, 24, lambda: r.scene.make_random_scene(48, 24, 3, False))

"""Generates the golden fixtures in this directory from the CPU oracle (oracle/rt_oracle.c) and from an
independent pure-Python restatement of the counter stream.  Run from the repo root:

    python tests/golden/make_golden.py

Fixtures are DATA (inputs + expected outputs):
  rng.json                    stream keys / raw bits / uniforms from the pure-Python restatement
  reference_kat.json          the known-answer data of the reference's own tests
                              (test/raytrace_clj/util_test.clj:44-49, hitable_test.clj:8-19,25-47,77-103,118-141)
  render_cover_n3.npz         oracle render of the cover scene n=3, 48x24x4spp (flat scene arrays included)
  render_two_spheres.npz      oracle render of make-two-spheres, 40x20x4spp
  paths_cover11_moving.npz    oracle `color` of 512 rays through the moving cover scene, with segment logs
  render_cornell.npz          oracle render (nested records) of the classic Cornell box, 40x40x8spp, with the flattened scene
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

import raytrace_clj_amd as r  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from raytrace_clj_amd.util import draw_bits, sample_key  # noqa: E402

FLAT_KEYS = ["prim_kind", "prim_geom", "prim_mat", "mat_kind", "mat_tex", "mat_param", "tex_kind", "tex_param", "tex_child", "cam"]


def flat_dict(fs):
    d = {k: np.asarray(getattr(fs, k)) for k in FLAT_KEYS}
    d["cam_kind"] = np.array(fs.cam_kind)
    return d


def main():
    orc = Oracle("f64")
    # ---- rng.json: pure-Python restatement, independent of the C code ----
    cases = []
    for seed, pix, s in [(0, 0, 0), (0x5EED0002, 0, 0), (0x5EED0002, 319999, 63), (42, 2073599, 255), ((1 << 64) - 1, (1 << 32) + 7, 4095)]:
        k = sample_key(seed, pix, s)
        bits = [draw_bits(k, d) for d in range(8)]
        cases.append({"seed": str(seed), "pixel": str(pix), "sample": str(s), "key": str(k), "bits": [str(b) for b in bits],
                      "u53": [(b >> 11) * 2.0 ** -53 for b in bits], "u24": [(b >> 40) * 2.0 ** -24 for b in bits]})
    json.dump({"cases": cases}, open(os.path.join(HERE, "rng.json"), "w"), indent=1)

    # ---- reference_kat.json: the reference's own test data ----
    grid = [[25.0 * i, 25.0 * j, 25.0 * k] for i in (-1, 0, 1) for j in (-1, 0, 1) for k in (-1, 0, 1)]
    dirs = [[5.0 * i, 5.0 * j, 5.0 * k] for i in (-1, 0, 1) for j in (-1, 0, 1) for k in (-1, 0, 1) if (i, j, k) != (0, 0, 0)]
    kat = {
        "point_at_parameter": {"origin": [1, 2, 3], "direction": [4, 5, 6], "cases": [[0, [1, 2, 3]], [1, [5, 7, 9]], [-1, [-3, -3, -3]]]},
        "sphere": {"gridpoints": grid, "directions": dirs, "radius": 1.0, "t_min": 0.0, "t_max": 3.4028234663852886e38,
                   "inward_hits": True, "outward_hits": False,
                   "grazing": [[[1, 1, 0], [-1, 0, 0]], [[1, 1, 0], [0, -1, 0]], [[1, 0, 1], [0, 0, -1]]], "inside_dir": [1, 1, 1]},
        "moving_sphere": {"t0": 0.1, "t1": 0.9, "delta": [10, 20, 30]},
        "center_at_time": {"a": [0, 0, 0], "b": [1, 2, 3], "cases": [[0, [0, 0, 0]], [1, [1, 2, 3]], [0.5, [0.5, 1.0, 1.5]]]},
        "aabb": {"vmin": [-1, -1, -1], "vmax": [1, 1, 1],
                 "hits": [[[0, 0, 0], [1, 1, 1]], [[-2, 0, 0], [1, 0, 0]], [[0, -2, 0], [0, 1, 0]], [[0, 0, -2], [0, 0, 1]],
                          [[-2, 1, 0], [1, 0, 0]], [[1, -2, 0], [0, 1, 0]], [[1, 0, -2], [0, 0, 1]]]},
        "surrounding_bbox": {"centers": [[-1, 2, -3], [1, -2, 3]], "radius": 0.1, "vmin": [-1.1, -2.1, -3.1], "vmax": [1.1, 2.1, 3.1]},
    }
    json.dump(kat, open(os.path.join(HERE, "reference_kat.json"), "w"), indent=1)

    # ---- oracle renders ----
    sc = r.scene.make_random_scene(48, 24, 3, False)
    fs = r.flatten.flatten(sc)
    lin, q, cnt = orc.render(fs, 48, 24, 4, depth=50, seed=0x5EED0002)
    np.savez_compressed(os.path.join(HERE, "render_cover_n3.npz"), nx=48, ny=24, ns=4, depth=50, seed=np.uint64(0x5EED0002),
                        linear=lin, rgb8=q, counters=cnt, **flat_dict(fs))
    sc = r.scene.make_two_spheres(40, 20)
    fs = r.flatten.flatten(sc)
    lin, q, cnt = orc.render(fs, 40, 20, 4, depth=50, seed=0x5EED0002)
    np.savez_compressed(os.path.join(HERE, "render_two_spheres.npz"), nx=40, ny=20, ns=4, depth=50, seed=np.uint64(0x5EED0002),
                        linear=lin, rgb8=q, counters=cnt, **flat_dict(fs))

    # ---- section 8(f3): classic Cornell box, rendered by the oracle evaluating the NESTED records ----
    from oracle.tree import flatten_with_tree
    fs = flatten_with_tree(r.scene.make_cornell_box(40, 40))
    lin, q, cnt = orc.render(fs, 40, 40, 8, depth=50, seed=0x5EED0002, nthreads=8)
    extra = {k: np.asarray(getattr(fs, k)) for k in ("prim_flip", "prim_xform", "xform_kind", "xform_param")}
    np.savez_compressed(os.path.join(HERE, "render_cornell.npz"), nx=40, ny=40, ns=8, depth=50, seed=np.uint64(0x5EED0002),
                        linear=lin, rgb8=q, counters=cnt, **flat_dict(fs), **extra)

    # ---- oracle paths with segment logs ----
    sc = r.scene.make_random_scene(200, 100, 11, True)
    fs = r.flatten.flatten(sc)
    rng = np.random.default_rng(11)
    n = 512
    uv = rng.random((n, 2))
    keys = np.array([sample_key(7, i, 0) for i in range(n)], np.uint64)
    cam = orc.probe_camera(fs, uv, keys)
    rays = cam[:, :7].copy()
    rgb, nseg, log, nlog = orc.probe_paths(fs, rays, keys, depth=50, ctr0=100, max_seg=8)
    np.savez_compressed(os.path.join(HERE, "paths_cover11_moving.npz"), rays=rays, keys=keys, ctr0=100, depth=50, rgb=rgb, nseg=nseg,
                        log=log, nlog=nlog, **flat_dict(fs))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()

import sys; sys.path.insert(0, '.')
import numpy as np
import raytrace_clj_amd as r
from raytrace_clj_amd import core
from oracle.oracle import Oracle
from oracle.tree import flatten_with_tree
orc = Oracle()
f = flatten_with_tree(r.scene.make_final(48, 48))
rng = np.random.default_rng(2)
n = 4096
keys = rng.integers(0, 2 ** 63, n, dtype=np.uint64)
cam = orc.probe_camera(f, rng.random((n, 2)), keys)
ctr0 = int(cam[:, 7].max())
ergb, enseg, elog, _ = orc.probe_paths(f, cam[:, :7], keys, depth=50, ctr0=ctr0, max_seg=6)
ctx = core.Context(0); ds = core.DeviceScene(f, ctx=ctx)
for accel in (1, 0):
    ctx.set_option("accel", accel)
    rgb, nseg, log, _ = ds.probe_paths(cam[:, :7], keys, depth=50, ctr0=ctr0, max_seg=6)
    bad = np.flatnonzero(~np.all(np.isclose(log, elog, rtol=1e-9, atol=1e-9), axis=(1, 2)))
    print("accel", accel, "bad paths", len(bad), "of", n)
    kinds = f.prim_kind
    first = {}
    for b in bad[:2000]:
        seg = np.flatnonzero(~np.all(np.isclose(log[b], elog[b], rtol=1e-9, atol=1e-9), axis=1))[0]
        de, dd = int(elog[b, seg, 0]), int(log[b, seg, 0])
        key = (seg, kinds[de] if elog[b, seg, 1] else -1, kinds[dd] if log[b, seg, 1] else -1, de == dd)
        first[key] = first.get(key, 0) + 1
    for k, v in sorted(first.items(), key=lambda kv: -kv[1])[:12]:
        print("  first differing seg %d: oracle kind %d, device kind %d, same prim %s : %d paths" % (k + (v,)))
    b = bad[0]; seg = np.flatnonzero(~np.all(np.isclose(log[b], elog[b], rtol=1e-9, atol=1e-9), axis=1))[0]
    print("  example path", b, "seg", seg, "\n   oracle", elog[b, seg], "\n   device", log[b, seg])
    if seg > 0: print("   prev   ", elog[b, seg - 1])

import time, sys, os
sys.path.insert(0, os.getcwd())
import raytrace_clj_amd as r
from raytrace_clj_amd import core, flatten as fl
for n in (11, 50, 150):
    sc = r.scene.make_random_scene(800, 400, n, False)
    f = fl.flatten(sc)
    ctx = core.Context(0)
    for walk in ("grid", "nogrid"):
        os.environ["RTMI_GRID"] = "0" if walk == "nogrid" else "1"
        t0 = time.perf_counter(); ds = core.DeviceScene(f, ctx=ctx); t1 = time.perf_counter()
        print("n=%d prims %d %s: scene creation %.1f ms" % (n, f.n_prims, walk, (t1 - t0) * 1e3), flush=True)
        ds.close()
    ctx.close()

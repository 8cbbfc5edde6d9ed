"""Serialises a world built from the host mirror's records (raytrace_clj_amd.hitable) into the oracle's NESTED node
table (rt_oracle.c: node_hit), i.e. the structure the reference itself evaluates: Hitlist scans, bvh-node slab tests,
FlipNormals / Translate / RotateY wrappers and Box's inner Hitlist stay nested.  TEST INFRASTRUCTURE ONLY.

attach_tree(flat, world) adds `.tree` to a FlatScene so Oracle.* evaluates the nested world; each leaf carries the
primitive index the product's flattener assigned to it, so segment logs can be compared index for index."""
import numpy as np

from raytrace_clj_amd import flatten as fl
from raytrace_clj_amd import hitable as H

(N_SPHERE, N_UVSPHERE, N_MOVING, N_RECT_XY, N_RECT_XZ, N_RECT_YZ, N_TRIANGLE, N_FLIP, N_TRANSLATE, N_ROTATE_Y, N_HITLIST,
 N_BOX, N_BVH, N_MEDIUM) = range(14)


def attach_tree(flat, world):
    leaves, keys = [], []
    fl._leaves(world, leaves, set(), keys=keys)  # world primitives only (media included, their boundaries not); keys: what every entry was recorded under
    prim_index = {k: i for i, k in enumerate(keys)}
    mat_of_prim = flat.prim_mat
    kind, a, d, prim, children = [], [], [], [], []

    def new(k, aa=(0, 0, 0), dd=(), p=-1):
        kind.append(k); a.append(list(aa)); row = np.zeros(12); row[:len(dd)] = dd; d.append(row); prim.append(p)
        return len(kind) - 1

    def leaf(k, o, dd, chain, flip):
        if state["boundary"]:
            return new(k, (0, 0, 0), dd, -1)
        i = prim_index[(id(o), chain, flip)]
        return new(k, (0, 0, int(mat_of_prim[i])), dd, i)

    state = {"boundary": False}

    def walk(o, chain, flip, boundary=False, in_list=False, slot=None):
        if boundary:
            state["boundary"] = True
            try:
                return walk(o, chain, flip)
            finally:
                state["boundary"] = False
        lid = id(o)  # (the flattener keys a medium's listing by the list OBJECT it stands in and its position there)
        if isinstance(o, (list, tuple)):
            o = H.Hitlist(list(o))
        if isinstance(o, H.Hitlist):
            n = new(N_HITLIST)
            ids = [walk(it, chain, flip, False, True, (lid, i)) for i, it in enumerate(o.items)]
            a[n] = [len(children), len(ids), 0]
            children.extend(ids)
            return n
        if isinstance(o, H.bvh_node):
            n = new(N_BVH, dd=list(o.box.vmin) + list(o.box.vmax))
            l = walk(o.left, chain, flip)
            r = l if o.right is o.left else walk(o.right, chain, flip)
            a[n] = [l, r, 0]
            return n
        if isinstance(o, H.Box):
            n = new(N_BOX, dd=list(o.p0) + list(o.p1))
            a[n] = [walk(o.sides, chain, flip, False, in_list, slot), 0, 0]
            return n
        if isinstance(o, H.FlipNormals):
            n = new(N_FLIP)
            a[n] = [walk(o.item, chain, flip ^ 1, False, in_list, slot), 0, 0]
            return n
        if isinstance(o, H.Translate):
            n = new(N_TRANSLATE, dd=list(o.offset))
            a[n] = [walk(o.item, chain + ((fl.XFORM_TRANSLATE, tuple(float(v) for v in o.offset)),), flip, False, in_list, slot), 0, 0]
            return n
        if isinstance(o, H.RotateY):
            n = new(N_ROTATE_Y, dd=[o.sin_theta, o.cos_theta])
            a[n] = [walk(o.obj, chain + ((fl.XFORM_ROTATE_Y, (float(o.sin_theta), float(o.cos_theta), 0.0)),), flip, False, in_list, slot), 0, 0]
            return n
        if isinstance(o, H.ConstantMedium):
            i = prim_index[(id(o), chain, flip) + ((slot,) if in_list else ())]  # a medium's LISTING in a Hitlist is a primitive of its own
            n = new(N_MEDIUM, dd=[o.density], p=i)
            a[n] = [walk(o.boundary, chain, flip, boundary=True), 0, int(mat_of_prim[i])]
            return n
        if isinstance(o, H.MovingSphere):
            return leaf(N_MOVING, o, list(o.center0) + [o.radius] + list(o.center1) + [o.t0, o.t1], chain, flip)
        if isinstance(o, H.UVSphere):
            return leaf(N_UVSPHERE, o, list(o.center) + [o.radius], chain, flip)
        if isinstance(o, H.Sphere):
            return leaf(N_SPHERE, o, list(o.center) + [o.radius], chain, flip)
        if isinstance(o, H.RectXY):
            return leaf(N_RECT_XY, o, [o.x0, o.y0, o.x1, o.y1, o.k], chain, flip)
        if isinstance(o, H.RectXZ):
            return leaf(N_RECT_XZ, o, [o.x0, o.z0, o.x1, o.z1, o.k], chain, flip)
        if isinstance(o, H.RectYZ):
            return leaf(N_RECT_YZ, o, [o.y0, o.z0, o.y1, o.z1, o.k], chain, flip)
        if isinstance(o, H.Triangle):
            return leaf(N_TRIANGLE, o, list(o.v0) + list(o.v1) + list(o.v2), chain, flip)
        raise TypeError(type(o).__name__)

    root = walk(world, (), 0)
    flat.tree = {"kind": np.array(kind, np.int32), "a": np.array(a, np.int32).reshape(-1, 3), "d": np.array(d, np.float64).reshape(-1, 12),
                 "prim": np.array(prim, np.int32), "children": np.array(children if children else [0], np.int32), "root": root}
    return flat


def flatten_with_tree(scene):
    """flatten({:camera :world}) + the nested world for the oracle"""
    flat = fl.flatten(scene)
    return attach_tree(flat, scene["world"])

"""The flattener: walks a {:camera :world} scene built from the reference's record types and produces
the flat arrays librtmi.so takes (layout: include/rtmi.h).

World traversal: Hitlist (hitable.clj:15-26) in item order; bvh-node (hitable.clj:97-123) left then
right, de-duplicating leaves by identity (a one-item make-bvh stores the same child twice,
hitable.clj:113-114).  The closest hit of a bvh-node tree equals the closest hit of the flat list
(both children are always visited with the un-narrowed interval, hitable.clj:100-105), so the device
scans the flat list.  Anything that is not one of the mirrored records raises UnsupportedOnGpuPath so a
caller can fall back to its own CPU path."""
import numpy as np

from . import camera as cam
from . import hitable as hit
from . import shader as shad
from . import texture as tex

PRIM_SPHERE, PRIM_UVSPHERE, PRIM_MOVING = 0, 1, 2
PRIM_RECT_XY, PRIM_RECT_XZ, PRIM_RECT_YZ, PRIM_TRIANGLE = 3, 4, 5, 6
PRIM_MEDIUM = 7        # ConstantMedium: prim_geom = density, first boundary primitive, boundary primitive count
PRIM_BOUNDARY = 16     # flag OR-ed into the kind of a primitive that only exists as (part of) a medium's boundary
XFORM_TRANSLATE, XFORM_ROTATE_Y = 0, 1
MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_DIFFUSE_LIGHT, MAT_ISOTROPIC = 0, 1, 2, 3, 4
TEX_CONSTANT, TEX_UVGRADIENT, TEX_CHECKER = 0, 1, 2
TEX_PERLIN_NOISE, TEX_PERLIN_TURB, TEX_MARBLE, TEX_FLIP_U, TEX_FLIP_V, TEX_IMAGE = 3, 4, 5, 6, 7, 8
CAM_PINHOLE, CAM_THINLENS = 0, 1
PRIM_STRIDE, TEX_STRIDE = 9, 12


class UnsupportedOnGpuPath(TypeError):
    """A record type the GPU path does not implement (rectangles, boxes, instances, media, ...)."""


class FlatScene:
    """Flat SoA description of a scene.  Attributes are numpy arrays in the layout of include/rtmi.h."""

    def __init__(self):
        self.prim_kind = np.zeros(0, np.int32)
        self.prim_geom = np.zeros((0, PRIM_STRIDE), np.float64)
        self.prim_mat = np.zeros(0, np.int32)
        self.mat_kind = np.zeros(0, np.int32)
        self.mat_tex = np.zeros(0, np.int32)
        self.mat_param = np.zeros(0, np.float64)
        self.tex_kind = np.zeros(0, np.int32)
        self.tex_param = np.zeros((0, TEX_STRIDE), np.float64)
        self.tex_child = np.zeros((0, 2), np.int32)
        self.cam_kind = CAM_PINHOLE
        self.cam = np.zeros(24, np.float64)
        # instancing (hitable.clj:375-486): per primitive, the parity of the FlipNormals wrappers around it and the chain of
        # Translate / RotateY wrappers, outermost first, as a slice of the xform table
        self.prim_flip = np.zeros(0, np.int32)
        self.prim_xform = np.zeros((0, 2), np.int32)   # first, count
        self.xform_kind = np.zeros(0, np.int32)
        self.xform_param = np.zeros((0, 3), np.float64)  # translate: offset.xyz | rotate-y: sin, cos, 0
        # procedural / image textures (texture.clj:60-138): the Perlin tables (perlin.clj:6-17, seeded) and ImageMap pixels
        self.perlin_vectors = None  # [256, 3] float64, set when a Perlin texture is present
        self.perlin_perm = None     # [3, 256] int32
        self.images = []            # [h, w, 3] uint8 arrays; TEX_IMAGE's first parameter indexes this list
        # ConstantMedium: the primitive indices of the media in the order (and multiplicity) the reference's descent calls their hit?
        self.media_calls = np.zeros(0, np.int32)
        self.media_mode = 0  # RTMI_MEDIA_DESCENT (bvh-node descent, un-narrowed t-max) | 1 = RTMI_MEDIA_HITLIST (the world is a Hitlist) | 2 = narrowed per call
        self.media_narrow_from = np.zeros(0, np.int32)  # per media call: first primitive of the Hitlist items that narrow it (= the medium's own index: none)

    n_world = None  # primitives [0, n_world) are the world; the rest are medium boundaries (None: all of them are the world)

    @property
    def n_prims(self):
        return len(self.prim_kind) if self.n_world is None else self.n_world


_LEAF_TYPES = (hit.Sphere, hit.UVSphere, hit.MovingSphere, hit.RectXY, hit.RectXZ, hit.RectYZ, hit.Triangle)


def _leaves(world, out, seen, chain=(), flip=0, in_list=False, calls=None, under_bvh=False, modes=None, ctx=None, slot=None, keys=None):
    """out gets (leaf, chain, flip): chain = the Translate/RotateY wrappers around the leaf, outermost first; keys (optional list, in step with out) the key every
    entry was recorded under.
    calls (optional list) gets (key, narrow_from) of every ConstantMedium in the order, and as often as, the reference's descent calls
    its hit?: make-bvh stores a lone item as bvh-node(L, L) (hitable.clj:113-114) and bvh-node.hit? evaluates both children
    (hitable.clj:101-102), so such a subtree is visited twice per ray -- harmless for surfaces, two random draws for a medium.  narrow_from = index (in out) of
    the first item of the Hitlist context the medium stands in: Hitlist.hit? hands it the closest hit of the items before it (hitable.clj:15-26); None = un-narrowed.
    modes (optional set) collects how the media are reached: "descent" (only bvh-nodes / wrappers above: un-narrowed t-max, hitable.clj:99-105), "hitlist" (only
    Hitlists / wrappers above) or "narrowed" (a Hitlist below bvh-nodes: round 4).
    ctx: the Hitlist context ({"lo": index of its first primitive, "dup": a shared record was skipped inside it}); slot: (id of the list, position) of the list
    item being walked -- a medium's LISTING: the same record listed twice in a list is two primitives, the same listing reached twice by the descent is one."""
    if isinstance(world, (hit.Hitlist, list, tuple)):
        items = world.items if isinstance(world, hit.Hitlist) else world
        inner = ctx if (in_list and ctx is not None) else {"lo": len(out), "dup": False, "poisoned": bool(ctx and ctx.get("poisoned"))}
        for i, it in enumerate(items):
            _leaves(it, out, seen, chain, flip, True, calls, under_bvh, modes, inner, (id(world), i), keys)
    elif isinstance(world, hit.ConstantMedium):
        if ctx is not None and ctx.get("poisoned"):
            raise UnsupportedOnGpuPath("a ConstantMedium below a bvh-node that itself sits inside a Hitlist is not supported on the GPU path (its t-max is narrowed by "
                                       "the list's earlier items but not by its bvh siblings)")
        if modes is not None:
            modes.add(("narrowed" if under_bvh else "hitlist") if in_list else "descent")
        key = (id(world), chain, flip) + ((slot,) if in_list else ())
        if key not in seen:
            if in_list and ctx["dup"]:
                raise UnsupportedOnGpuPath("a Hitlist that holds a ConstantMedium shares a record with another part of the world: its items are not contiguous in the "
                                           "flattened order, so the medium's narrowing cannot be expressed on the GPU path")
            seen.add(key)
            out.append((world, chain, flip))
            if keys is not None:
                keys.append(key)
        if calls is not None:
            calls.append((key, ctx["lo"] if in_list else None))
    elif isinstance(world, hit.bvh_node):
        below = {"lo": None, "dup": False, "poisoned": True} if in_list else None  # a bvh-node INSIDE a Hitlist: media below it are narrowed by some siblings only
        _leaves(world.left, out, seen, chain, flip, False, calls, True, modes, below, None, keys)
        _leaves(world.right, out, seen, chain, flip, False, calls, True, modes, below, None, keys)
    elif isinstance(world, hit.Box):
        _leaves(world.sides, out, seen, chain, flip, in_list, calls, under_bvh, modes, ctx, slot, keys)  # (hit? sides ...) with the caller's interval: its six rectangles splice into the enclosing list
    elif isinstance(world, hit.FlipNormals):
        _leaves(world.item, out, seen, chain, flip ^ 1, in_list, calls, under_bvh, modes, ctx, slot, keys)
    elif isinstance(world, hit.Translate):
        _leaves(world.item, out, seen, chain + ((XFORM_TRANSLATE, tuple(float(v) for v in world.offset)),), flip, in_list, calls, under_bvh, modes, ctx, slot, keys)
    elif isinstance(world, hit.RotateY):
        _leaves(world.obj, out, seen, chain + ((XFORM_ROTATE_Y, (float(world.sin_theta), float(world.cos_theta), 0.0)),), flip, in_list, calls, under_bvh, modes, ctx, slot, keys)
    elif isinstance(world, _LEAF_TYPES):
        key = (id(world), chain, flip)  # the same record under two different instances is two primitives
        if key not in seen:
            seen.add(key)
            out.append((world, chain, flip))
            if keys is not None:
                keys.append(key)
        elif in_list and ctx is not None:
            ctx["dup"] = True
    else:
        raise UnsupportedOnGpuPath("%s is not supported on the GPU path" % type(world).__name__)


class _Interner:
    def __init__(self):
        self.ids = {}
        self.rows = []

    def get(self, obj, build):
        k = id(obj)
        if k not in self.ids:
            row = build(obj)            # children first, so a checker's children have lower ids
            self.ids[k] = len(self.rows)
            self.rows.append(row)
        return self.ids[k]


def flatten(scene_or_world, camera=None, perlin_seed=None):
    """flatten({"camera": c, "world": w}) or flatten(world, camera) -> FlatScene"""
    if isinstance(scene_or_world, dict):
        world, camera = scene_or_world["world"], scene_or_world["camera"]
    else:
        world = scene_or_world
    leaves, medium_calls, modes, keys = [], [], set(), []
    _leaves(world, leaves, set(), calls=medium_calls, modes=modes, keys=keys)

    textures, materials = _Interner(), _Interner()
    uses, images = {"perlin": False}, []

    def build_tex(t):
        p = np.zeros(TEX_STRIDE)
        if isinstance(t, tex.Constant):
            p[0:3] = t.color
            return (TEX_CONSTANT, p, (-1, -1))
        if isinstance(t, tex.UVGradient):
            p[0:3], p[3:6], p[6:9], p[9:12] = t.co, t.cu, t.cv, t.cuv
            return (TEX_UVGRADIENT, p, (-1, -1))
        if isinstance(t, tex.Checkerboard):
            c0 = textures.get(t.tex0, build_tex)
            c1 = textures.get(t.tex1, build_tex)
            p[0] = t.scale
            return (TEX_CHECKER, p, (c0, c1))
        if isinstance(t, tex.PerlinNoise):
            uses["perlin"] = True
            p[0] = t.scale
            return (TEX_PERLIN_NOISE, p, (-1, -1))
        if isinstance(t, (tex.PerlinTurbulence, tex.Marble)):
            uses["perlin"] = True
            p[0], p[1] = t.scale, t.depth
            return (TEX_PERLIN_TURB if isinstance(t, tex.PerlinTurbulence) else TEX_MARBLE, p, (-1, -1))
        if isinstance(t, (tex.FlipTextureU, tex.FlipTextureV)):
            c0 = textures.get(t.tex, build_tex)
            return (TEX_FLIP_U if isinstance(t, tex.FlipTextureU) else TEX_FLIP_V, p, (c0, -1))
        if isinstance(t, tex.ImageMap):
            p[0] = len(images)
            images.append(t.image)
            return (TEX_IMAGE, p, (-1, -1))
        raise UnsupportedOnGpuPath("texture %s is not supported on the GPU path" % type(t).__name__)

    def build_mat(m):
        if isinstance(m, shad.Lambertian):
            return (MAT_LAMBERTIAN, textures.get(m.albedo, build_tex), 0.0)
        if isinstance(m, shad.Metal):
            return (MAT_METAL, textures.get(m.albedo, build_tex), m.fuzz)
        if isinstance(m, shad.Dielectric):
            return (MAT_DIELECTRIC, -1, m.ri)
        if isinstance(m, shad.DiffuseLight):
            return (MAT_DIFFUSE_LIGHT, textures.get(m.tex, build_tex), 0.0)
        if isinstance(m, shad.Isotropic):
            return (MAT_ISOTROPIC, textures.get(m.albedo, build_tex), 0.0)
        raise UnsupportedOnGpuPath("material %s is not supported on the GPU path" % type(m).__name__)

    # a medium's boundary is its own little world: its leaves are appended AFTER the world's primitives (flagged
    # PRIM_BOUNDARY) and the medium refers to them by (first, count)
    n_world = len(leaves)
    boundary_of = {}
    for i in range(n_world):
        o, chain, flip = leaves[i]
        if isinstance(o, hit.ConstantMedium):
            b = []
            _leaves(o.boundary, b, set(), chain, flip, False)
            if any(isinstance(x[0], hit.ConstantMedium) for x in b):
                raise UnsupportedOnGpuPath("a ConstantMedium inside a ConstantMedium's boundary is not supported on the GPU path")
            boundary_of[i] = (len(leaves), len(b))
            leaves.extend(b)
    fs = FlatScene()
    fs.n_world = n_world
    index_of = {k: i for i, k in enumerate(keys)}
    fs.media_calls = np.array([index_of[k] for k, lo in medium_calls], np.int32)
    # per call: the first primitive of the Hitlist items that narrow it (= its own index: none) -- rtmi_scene_set_media_calls_narrowed
    fs.media_narrow_from = np.array([index_of[k] if lo is None else lo for k, lo in medium_calls], np.int32)
    # RTMI_MEDIA_HITLIST (1): the world is a Hitlist, its narrowing reaches every medium; 2 = RTMI_MEDIA_NARROWED: Hitlists holding media below bvh-nodes (round 4)
    fs.media_mode = 1 if modes == {"hitlist"} else (2 if "narrowed" in modes or "hitlist" in modes else 0)
    n = len(leaves)
    fs.prim_kind = np.zeros(n, np.int32)
    fs.prim_geom = np.zeros((n, PRIM_STRIDE), np.float64)
    fs.prim_mat = np.zeros(n, np.int32)
    fs.prim_flip = np.zeros(n, np.int32)
    fs.prim_xform = np.zeros((n, 2), np.int32)
    chains, xk, xp = {}, [], []
    for i, (o, chain, flip) in enumerate(leaves):
        g = fs.prim_geom[i]
        if isinstance(o, hit.ConstantMedium):
            fs.prim_kind[i] = PRIM_MEDIUM
            g[0], g[1], g[2] = o.density, boundary_of[i][0], boundary_of[i][1]
            fs.prim_mat[i] = materials.get(o.phase_fn, build_mat)
            fs.prim_flip[i] = flip
            if chain:
                if chain not in chains:
                    chains[chain] = (len(xk), len(chain))
                    for kind, params in chain:
                        xk.append(kind)
                        xp.append(params)
                fs.prim_xform[i] = chains[chain]
            continue
        if isinstance(o, hit.MovingSphere):
            fs.prim_kind[i] = PRIM_MOVING
            g[0:3], g[3], g[4:7], g[7], g[8] = o.center0, o.radius, o.center1, o.t0, o.t1
        elif isinstance(o, (hit.Sphere, hit.UVSphere)):
            fs.prim_kind[i] = PRIM_UVSPHERE if isinstance(o, hit.UVSphere) else PRIM_SPHERE
            g[0:3], g[3], g[4:7], g[7], g[8] = o.center, o.radius, o.center, 0.0, 1.0
        elif isinstance(o, hit.RectXY):
            fs.prim_kind[i] = PRIM_RECT_XY
            g[0:5] = o.x0, o.y0, o.x1, o.y1, o.k
        elif isinstance(o, hit.RectXZ):
            fs.prim_kind[i] = PRIM_RECT_XZ
            g[0:5] = o.x0, o.z0, o.x1, o.z1, o.k
        elif isinstance(o, hit.RectYZ):
            fs.prim_kind[i] = PRIM_RECT_YZ
            g[0:5] = o.y0, o.z0, o.y1, o.z1, o.k
        else:
            fs.prim_kind[i] = PRIM_TRIANGLE
            g[0:3], g[3:6], g[6:9] = o.v0, o.v1, o.v2
        fs.prim_mat[i] = materials.get(o.material, build_mat)
        fs.prim_flip[i] = flip
        if chain:
            if chain not in chains:
                chains[chain] = (len(xk), len(chain))
                for kind, params in chain:
                    xk.append(kind)
                    xp.append(params)
            fs.prim_xform[i] = chains[chain]
        if i >= n_world:
            fs.prim_kind[i] |= PRIM_BOUNDARY
    fs.xform_kind = np.array(xk, np.int32)
    fs.xform_param = np.array(xp, np.float64).reshape(-1, 3)
    fs.mat_kind = np.array([r[0] for r in materials.rows], np.int32)
    fs.mat_tex = np.array([r[1] for r in materials.rows], np.int32)
    fs.mat_param = np.array([r[2] for r in materials.rows], np.float64)
    fs.tex_kind = np.array([r[0] for r in textures.rows], np.int32)
    fs.tex_param = np.array([r[1] for r in textures.rows], np.float64).reshape(-1, TEX_STRIDE)
    fs.tex_child = np.array([r[2] for r in textures.rows], np.int32).reshape(-1, 2)

    fs.images = images
    if uses["perlin"]:
        from . import perlin
        fs.perlin_vectors, fs.perlin_perm = perlin.make_tables(perlin.PERLIN_SEED if perlin_seed is None else perlin_seed)
    if camera is not None:
        fs.cam_kind, fs.cam = flatten_camera(camera)
    return fs


def flatten_camera(camera):
    c = np.zeros(24, np.float64)
    if isinstance(camera, cam.ThinLensCamera):
        c[0:3], c[3:6], c[6:9], c[9:12] = camera.origin, camera.lleft, camera.horiz, camera.vert
        c[12:15], c[15:18], c[18:21] = camera.u, camera.v, camera.w
        c[21], c[22], c[23] = camera.aperture, camera.t0, camera.t1
        return CAM_THINLENS, c
    if isinstance(camera, cam.PinholeCamera):
        c[0:3], c[3:6], c[6:9], c[9:12] = camera.origin, camera.lleft, camera.horiz, camera.vert
        return CAM_PINHOLE, c
    raise UnsupportedOnGpuPath("camera %s is not supported on the GPU path" % type(camera).__name__)

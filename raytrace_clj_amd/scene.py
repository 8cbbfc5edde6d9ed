"""Host mirror of the in-scope part of raytrace-clj.scene (src/raytrace_clj/scene.clj): scene
functions return {"camera": c, "world": w} exactly like the reference's {:camera c :world w}.

make_random_scene = the Shirley cover scene (scene.clj:318-412); make_two_spheres = scene.clj:9-48.
The reference draws scene randomness from the unseeded clojure.core/rand; here every (rand) is the
next value of a seeded SplitMix64 stream, consumed in the reference's evaluation order (per grid cell:
centre x, centre z, choose-mat, then the material's own draws; cells failing the :when filter still
consume their three draws).  make_two_triangles = scene.clj:80-114 and make_cornell_box (classic) = scene.clj:230-316 use the section-8(f3) records
(rectangles, boxes, instances, triangles); make_two_perlin_spheres = scene.clj:50-78, make_textured_sphere = scene.clj:116-150
(its earth.png is not in the reference repository: a synthetic image stands in) and make_example_light = scene.clj:191-228
use the section-8(f4) textures; make_subsurface_sphere = scene.clj:152-189, the foggy Cornell box (classic=False) and
make_final = scene.clj:415-489 use ConstantMedium / Isotropic.  That is every scene function of scene.clj."""
import math

import numpy as np

from . import camera as cam
from . import hitable as hit
from . import shader as shad
from . import texture as tex
from .util import SplitMix64, vec3

SCENE_SEED = 0x5EED0001


def _aspect(nx, ny):
    # (/ (float nx) (float ny)): both operands exact floats, the division is done in double
    return float(np.float32(nx)) / float(np.float32(ny))


def make_two_spheres(nx, ny, seed=SCENE_SEED):
    """two touching spheres -- scene.clj:9-48"""
    rng = SplitMix64(seed)
    checker = tex.checkerboard(tex0=tex.constant(color=vec3(0.2, 0.3, 0.1)),
                               tex1=tex.constant(color=vec3(0.9, 0.9, 0.9)), scale=10)
    return {
        "camera": cam.thin_lens_camera(lookfrom=vec3(13, 2, 3), lookat=vec3(0, 1, 0), vup=vec3(0, 1, 0), vfov=40,
                                       aspect=_aspect(nx, ny), aperture=0.0, focus_dist=10.0, t0=0.0, t1=1.0),
        "world": hit.make_bvh([
            hit.uv_sphere(center=vec3(0, 0, 0), radius=1000,
                          material=shad.diffuse_light(tex=tex.uv_gradient(co=vec3(1, 1, 1), cu=vec3(1, 1, 1),
                                                                          cv=vec3(0.5, 0.7, 1.0), cuv=vec3(0.5, 0.7, 1.0)))),
            hit.sphere(center=vec3(0, -10, 0), radius=10, material=shad.lambertian(albedo=checker)),
            hit.uv_sphere(center=vec3(0, 2, 0), radius=2,
                          material=shad.lambertian(albedo=tex.uv_gradient(co=vec3(0, 1, 0), cu=vec3(0, 1, 1),
                                                                          cv=vec3(1, 0, 1), cuv=vec3(1, 0, 0)))),
        ], 0.0, 1.0, rng),
    }


def make_two_perlin_spheres(nx, ny, seed=SCENE_SEED):
    """two perlin noise spheres -- scene.clj:50-78"""
    rng = SplitMix64(seed)
    turb = tex.perlin_turbulence(scale=4, depth=7)
    return {
        "camera": cam.thin_lens_camera(lookfrom=vec3(13, 2, 3), lookat=vec3(0, 1, 0), vup=vec3(0, 1, 0), vfov=40,
                                       aspect=_aspect(nx, ny), aperture=0.0, focus_dist=10.0, t0=0.0, t1=1.0),
        "world": hit.make_bvh([
            hit.sphere(center=vec3(0, 0, 0), radius=1000, material=shad.diffuse_light(tex=tex.constant(color=0.8 * vec3(0.3, 0.5, 0.8)))),
            hit.sphere(center=vec3(0, -1000, 0), radius=1000, material=shad.lambertian(albedo=turb)),
            hit.sphere(center=vec3(0, 2, 0), radius=2, material=shad.lambertian(albedo=turb)),
        ], 0.0, 1.0, rng),
    }


def _earth(image):
    """the reference reads "earth.png" from the working directory (scene.clj:125, 426); the file is not in its repository, so: the
    caller's pixels, else ./earth.png when there is one (PNG decoder in texture.py), else the synthetic stand-in"""
    import os
    if image is not None:
        return tex.image_map(image=image)
    if os.path.exists("earth.png"):
        return tex.image_map(filename="earth.png")
    return tex.image_map(image=synthetic_earth())


def synthetic_earth(w=256, h=128):
    """stand-in for the reference's earth.png (scene.clj:125-126: not in the repository, *.png is git-ignored): a
    deterministic land/ocean pattern, [h, w, 3] uint8"""
    y, x = np.mgrid[0:h, 0:w]
    lon, lat = x / w * 2 * np.pi, (y / h - 0.5) * np.pi
    land = (np.sin(3 * lon) * np.cos(2 * lat) + 0.5 * np.sin(7 * lon + 1.3) * np.sin(5 * lat) + 0.3 * np.cos(11 * lon - 2 * lat)) > 0.25
    img = np.zeros((h, w, 3), np.uint8)
    img[...] = (20, 60, 140)
    img[land] = (40, 130, 50)
    img[np.abs(lat) > 1.25] = (235, 240, 245)
    return img


def make_textured_sphere(nx, ny, image=None, seed=SCENE_SEED):
    """texture mapped sphere -- scene.clj:116-150 (earth.png replaced by `image`, default synthetic_earth())"""
    rng = SplitMix64(seed)
    checker = tex.checkerboard(tex0=tex.constant(color=vec3(0.2, 0.3, 0.1)), tex1=tex.constant(color=vec3(0.9, 0.9, 0.9)), scale=10)
    earth = tex.flip_texture_v(tex=_earth(image))
    return {
        "camera": cam.thin_lens_camera(lookfrom=vec3(13, 2, 3), lookat=vec3(0, 1, 0), vup=vec3(0, 1, 0), vfov=15,
                                       aspect=_aspect(nx, ny), aperture=0.0, focus_dist=10.0, t0=0.0, t1=1.0),
        "world": hit.make_bvh([
            hit.sphere(center=vec3(0, 0, 0), radius=1000, material=shad.diffuse_light(tex=tex.constant(color=0.8 * vec3(0.3, 0.5, 0.8)))),
            hit.sphere(center=vec3(0, -10, 0), radius=10, material=shad.lambertian(albedo=checker)),
            hit.uv_sphere(center=vec3(0, 1, 0), radius=1, material=shad.lambertian(albedo=earth)),
        ], 0.0, 1.0, rng),
    }


def make_example_light(nx, ny, seed=SCENE_SEED):
    """scene with rectangular area light -- scene.clj:191-228"""
    rng = SplitMix64(seed)
    gray = shad.lambertian(albedo=tex.constant(color=vec3(0.6, 0.6, 0.6)))
    light = shad.diffuse_light(tex=tex.constant(color=vec3(4, 4, 4)))
    return {
        "camera": cam.thin_lens_camera(lookfrom=vec3(13, 2, 3), lookat=vec3(0, 1, 0), vup=vec3(0, 1, 0), vfov=40,
                                       aspect=_aspect(nx, ny), aperture=0.0, focus_dist=10.0, t0=0.0, t1=1.0),
        "world": hit.make_bvh([
            hit.sphere(center=vec3(0, -1000, 0), radius=1000, material=gray),
            hit.sphere(center=vec3(0, 2, 0), radius=2, material=gray),
            hit.sphere(center=vec3(0, 7, 0), radius=2, material=light),
            hit.rect_xy(x0=3, y0=1, x1=5, y1=3, k=-2, material=light),
        ], 0.0, 1.0, rng),
    }


def make_two_triangles(nx, ny, seed=SCENE_SEED):
    """two triangles, view down the -z axis -- scene.clj:80-114"""
    rng = SplitMix64(seed)
    white = tex.constant(color=vec3(0.9, 0.9, 0.9))
    red = tex.constant(color=vec3(0.9, 0, 0))
    return {
        "camera": cam.thin_lens_camera(lookfrom=vec3(1, 1, -10), lookat=vec3(1, 1, 0), vup=vec3(0, 1, 0), vfov=20,
                                       aspect=_aspect(nx, ny), aperture=0.0, focus_dist=10.0, t0=0.0, t1=1.0),
        "world": hit.make_bvh([
            hit.sphere(center=vec3(0, 0, 0), radius=1000, material=shad.diffuse_light(tex=tex.constant(color=0.8 * vec3(0.3, 0.5, 0.8)))),
            hit.triangle(v0=vec3(0, 0, 0), v1=vec3(0, 1, 0), v2=vec3(1, 0, 0), material=shad.lambertian(albedo=red)),
            hit.triangle(v0=vec3(1, 1, 0), v1=vec3(1, 2, 0), v2=vec3(2, 1, 0), material=shad.lambertian(albedo=white)),
        ], 0.0, 1.0, rng),
    }


def make_cornell_box(nx, ny, classic=True, seed=SCENE_SEED):
    """classic cornell box -- scene.clj:230-316; classic=False: bigger light and the two boxes filled with fog (ConstantMedium)"""
    rng = SplitMix64(seed)
    red = shad.lambertian(albedo=tex.constant(color=vec3(0.65, 0.05, 0.05)))
    white = shad.lambertian(albedo=tex.constant(color=vec3(0.73, 0.73, 0.73)))
    green = shad.lambertian(albedo=tex.constant(color=vec3(0.12, 0.45, 0.15)))
    light = shad.diffuse_light(tex=tex.constant(color=vec3(7, 7, 7)))
    box1 = hit.translate(item=hit.rotate_y(item=hit.box(p0=vec3(0, 0, 0), p1=vec3(165, 165, 165), material=white), theta=-18.0),
                         offset=vec3(130, 0, 65))
    box2 = hit.translate(item=hit.rotate_y(item=hit.box(p0=vec3(0, 0, 0), p1=vec3(165, 330, 165), material=white), theta=15.0),
                         offset=vec3(265, 0, 295))
    return {
        "camera": cam.thin_lens_camera(lookfrom=vec3(278, 278, -800), lookat=vec3(278, 278, 0), vup=vec3(0, 1, 0), vfov=40,
                                       aspect=_aspect(nx, ny), aperture=0.0, focus_dist=10.0, t0=0.0, t1=1.0),
        "world": hit.make_bvh([
            hit.flip_normals(item=hit.rect_yz(y0=0, z0=0, y1=555, z1=555, k=555, material=green)),
            hit.rect_yz(y0=0, z0=0, y1=555, z1=555, k=0, material=red),
            hit.rect_xz(x0=213, z0=227, x1=343, z1=332, k=554, material=light) if classic else
            hit.rect_xz(x0=113, z0=127, x1=443, z1=432, k=554, material=light),
            hit.flip_normals(item=hit.rect_xz(x0=0, z0=0, x1=555, z1=555, k=555, material=white)),
            hit.rect_xz(x0=0, z0=0, x1=555, z1=555, k=0, material=white),
            hit.flip_normals(item=hit.rect_xy(x0=0, y0=0, x1=555, y1=555, k=555, material=white)),
            box1 if classic else hit.constant_medium(boundary=box1, density=0.01, albedo=tex.constant(color=vec3(1, 1, 1))),
            box2 if classic else hit.constant_medium(boundary=box2, density=0.01, albedo=tex.constant(color=vec3(0, 0, 0))),
        ], 0.0, 1.0, rng),
    }


def make_subsurface_sphere(nx, ny, seed=SCENE_SEED):
    """subsurface reflection sphere -- scene.clj:152-189: a glass ball that is also the boundary of a blue medium"""
    rng = SplitMix64(seed)
    checker = tex.checkerboard(tex0=tex.constant(color=vec3(0.2, 0.3, 0.1)), tex1=tex.constant(color=vec3(0.9, 0.9, 0.9)), scale=10)
    ball = hit.sphere(center=vec3(0, 2, 0), radius=2, material=shad.dielectric(ri=1.5))
    medium = hit.constant_medium(boundary=ball, density=0.9, albedo=tex.constant(color=vec3(0.2, 0.4, 0.9)))
    return {
        "camera": cam.thin_lens_camera(lookfrom=vec3(13, 2, 3), lookat=vec3(0, 1, 0), vup=vec3(0, 1, 0), vfov=40,
                                       aspect=_aspect(nx, ny), aperture=0.0, focus_dist=10.0, t0=0.0, t1=1.0),
        "world": hit.make_bvh([
            hit.sphere(center=vec3(0, 0, 0), radius=1000, material=shad.diffuse_light(tex=tex.constant(color=0.8 * vec3(0.3, 0.5, 0.8)))),
            hit.sphere(center=vec3(0, -10, 0), radius=10, material=shad.lambertian(albedo=checker)),
            medium,
            ball,
        ], 0.0, 1.0, rng),
    }


def make_final(nx, ny, image=None, seed=SCENE_SEED, parts=None):
    """book 2 final example -- scene.clj:415-489, the scene raytrace-clj.core/-main renders as shipped (core.clj:90).
    earth.png is not in the reference repository: `image` (default synthetic_earth()) stands in.
    parts (experiments only, scripts/gpu_final_variants.py): the names of the world's items to keep (default: all eleven)."""
    rng = SplitMix64(seed)
    rand = rng.rand
    white = shad.lambertian(albedo=tex.constant(color=vec3(0.73, 0.73, 0.73)))
    ground = shad.lambertian(albedo=tex.constant(color=vec3(0.48, 0.83, 0.53)))
    orange = shad.lambertian(albedo=tex.constant(color=vec3(0.7, 0.3, 0.1)))
    light = shad.diffuse_light(tex=tex.constant(color=vec3(7, 7, 7)))
    glass = shad.dielectric(ri=1.5)
    metal = shad.metal(albedo=tex.constant(color=vec3(0.8, 0.8, 0.9)), fuzz=10)
    bndry = hit.sphere(center=vec3(360, 150, 145), radius=70, material=glass)
    earth = shad.lambertian(albedo=tex.flip_texture_v(tex=_earth(image)))
    marble = shad.lambertian(albedo=tex.marble(scale=0.1, depth=4))
    nb, ns = 20, 1000
    boxes = []
    for i in range(nb):
        for j in range(nb):
            w = 100
            p0 = vec3(-1000 + i * w, 0, -1000 + j * w)
            p1 = p0 + vec3(w, 100 * (rand() + 0.01), w)
            boxes.append(hit.box(p0=p0, p1=p1, material=ground))
    ground_bvh = hit.make_bvh(boxes, 0.0, 1.0, rng)
    packed = hit.make_bvh([hit.sphere(center=165.0 * vec3(rand(), rand(), rand()), radius=10, material=white) for _ in range(ns)], 0.0, 1.0, rng)
    items = [
        ("ground", ground_bvh),
        ("light", hit.rect_xz(x0=123, z0=147, x1=423, z1=412, k=554, material=light)),
        ("moving", hit.moving_sphere(center0=vec3(400, 400, 200), t0=0, center1=vec3(430, 400, 200), t1=1, radius=50, material=orange)),
        ("glass", hit.sphere(center=vec3(260, 150, 45), radius=50, material=glass)),
        ("metal", hit.sphere(center=vec3(0, 150, 145), radius=50, material=metal)),
        ("bndry", bndry),
        ("medium", hit.constant_medium(boundary=bndry, density=0.2, albedo=tex.constant(color=vec3(0.2, 0.4, 0.9)))),
        ("haze", hit.constant_medium(boundary=hit.sphere(center=vec3(0, 0, 0), radius=5000, material=glass), density=0.0001,
                                     albedo=tex.constant(color=vec3(1, 1, 1)))),
        ("earth", hit.uv_sphere(center=vec3(400, 200, 400), radius=100, material=earth)),
        ("marble", hit.sphere(center=vec3(220, 280, 300), radius=80, material=marble)),
        ("cube", hit.translate(item=hit.rotate_y(item=packed, theta=15), offset=vec3(-100, 270, 395))),
    ]
    return {
        "camera": cam.thin_lens_camera(lookfrom=vec3(478, 278, -600), lookat=vec3(278, 278, 0), vup=vec3(0, 1, 0), vfov=40,
                                       aspect=_aspect(nx, ny), aperture=0.0, focus_dist=10.0, t0=0.0, t1=1.0),
        "world": hit.make_bvh([it for name, it in items if parts is None or name in parts], 0.0, 1.0, rng),
    }


def make_random_scene(nx, ny, n, moving, seed=SCENE_SEED, mix=(0.8, 0.95), bvh=True):
    """make a random scene -- scene.clj:318-412 (the cover scene is n=11).

    mix = (diffuse threshold, metal threshold) of choose-mat, scene.clj:386,406: the reference's
    (0.8, 0.95); BASELINE config 5 ("dielectric-heavy") uses (0.1, 0.2).
    bvh=False returns a Hitlist instead of the make-bvh tree (same closest hits)."""
    rng = SplitMix64(seed)
    rand = rng.rand
    camera = cam.thin_lens_camera(lookfrom=vec3(13, 2, 3), lookat=vec3(0, 0, 0), vup=vec3(0, 1, 0), vfov=20,
                                  aspect=_aspect(nx, ny), aperture=0.0, focus_dist=10.0, t0=0.0, t1=1.0)
    items = [
        hit.uv_sphere(center=vec3(0, 0, 0), radius=1000,  # sky dome
                      material=shad.diffuse_light(tex=tex.uv_gradient(co=vec3(1, 1, 1), cu=vec3(1, 1, 1),
                                                                      cv=vec3(0.5, 0.7, 1.0), cuv=vec3(0.5, 0.7, 1.0)))),
        hit.sphere(center=vec3(0, -1000, 0), radius=1000,  # ground
                   material=shad.lambertian(albedo=tex.checkerboard(tex0=tex.constant(color=vec3(0.2, 0.3, 0.1)),
                                                                    tex1=tex.constant(color=vec3(0.9, 0.9, 0.9)), scale=10))),
        hit.sphere(center=vec3(0, 1, 0), radius=1, material=shad.dielectric(ri=1.5)),  # glass
        hit.sphere(center=vec3(-4, 1, 0), radius=1,  # plastic
                   material=shad.lambertian(albedo=tex.constant(color=vec3(0.4, 0.2, 0.1)))),
        hit.sphere(center=vec3(4, 1, 0), radius=1,  # metal
                   material=shad.metal(albedo=tex.constant(color=vec3(0.7, 0.6, 0.5)), fuzz=0.0)),
    ]
    for a in range(-n, n):
        for b in range(-n, n):
            center = vec3(a + 0.9 * rand(), 0.2, b + 0.9 * rand())
            choose_mat = rand()
            d = center - vec3(4, 0.2, 0)
            if not math.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) > 0.9:
                continue
            if choose_mat < mix[0]:  # diffuse
                if moving:
                    center1 = center + vec3(0, 0.5 * rand(), 0)
                    albedo = tex.constant(color=vec3(rand() * rand(), rand() * rand(), rand() * rand()))
                    items.append(hit.moving_sphere(center0=center, t0=0.0, center1=center1, t1=1.0, radius=0.2,
                                                   material=shad.lambertian(albedo=albedo)))
                else:
                    albedo = tex.constant(color=vec3(rand() * rand(), rand() * rand(), rand() * rand()))
                    items.append(hit.sphere(center=center, radius=0.2, material=shad.lambertian(albedo=albedo)))
            elif choose_mat < mix[1]:  # metal
                albedo = tex.constant(color=vec3(0.5 * (rand() + 1), 0.5 * (rand() + 1), 0.5 * (rand() + 1)))
                items.append(hit.sphere(center=center, radius=0.2, material=shad.metal(albedo=albedo, fuzz=0.5 * rand())))
            else:  # glass
                items.append(hit.sphere(center=center, radius=0.2, material=shad.dielectric(ri=1.5)))
    world = hit.make_bvh(items, 0.0, 1.0, rng) if bvh else hit.hitlist(items=items)
    return {"camera": camera, "world": world}

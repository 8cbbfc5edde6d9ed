/* host_smoke.c -- a torch-free, Python-free host of librtmi.so: what the reference-side binding (JNA from the JVM, see
 * INTEGRATION.md) does, in plain C.  dlopen()s the library (so the HIP runtime it binds to is the system one, /opt/rocm),
 * builds a scene from flat arrays through rtmi_scene_create, renders through rtmi_render (core.clj:100-108 replaced), then
 * clones the scene onto a second context and renders again through the single-process multi-device entry
 * rtmi_render_multi.  With RTMI_HOST_SMOKE_RCCL=1 in the environment a third frame goes through the in-library RCCL gather on a
 * one-rank communicator (RTMI_MULTI_GATHER=rccl, one replica): this process opens /opt/rocm's librccl.so.1 by soname, as a JVM would.
 * Test infrastructure: built and run by tests/test_gpu_round2.py and tests/test_gpu_round3.py.
 *
 *   host_smoke <librtmi.so> <scene.bin> <out.bin>
 *
 * scene.bin: int32 n_prims, n_mats, n_tex, cam_kind, nx, ny, ns, depth; uint64 seed; then the arrays of rtmi_scene_create
 * in argument order.  out.bin: twice (three times with RTMI_HOST_SMOKE_RCCL) { double linear[ny*nx*3]; uint8 rgb8[ny*nx*3]; uint64 counters[2] }. */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "rtmi.h"

#define LOAD(name) __typeof__(&name) p_##name = (__typeof__(&name))dlsym(h, #name); if (!p_##name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; }
#define CHECK(expr) do { int rc_ = (expr); if (rc_ != RTMI_OK) { fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, p_rtmi_last_error()); return 3; } } while (0)

static void *rd(FILE *f, size_t bytes) {
    void *p = malloc(bytes ? bytes : 1);
    if (!p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read\n"); exit(4); }
    return p;
}

int main(int argc, char **argv) {
    if (argc != 4) { fprintf(stderr, "usage: host_smoke librtmi.so scene.bin out.bin\n"); return 1; }
    void *h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    LOAD(rtmi_last_error) LOAD(rtmi_backend_name) LOAD(rtmi_init) LOAD(rtmi_shutdown) LOAD(rtmi_scene_create) LOAD(rtmi_scene_clone)
    LOAD(rtmi_scene_destroy) LOAD(rtmi_render) LOAD(rtmi_render_multi) LOAD(rtmi_device_info) LOAD(rtmi_rccl_probe) LOAD(rtmi_last_gather_path)
    FILE *f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); return 4; }
    int32_t *hd = (int32_t *)rd(f, 8 * sizeof(int32_t));
    const int n_prims = hd[0], n_mats = hd[1], n_tex = hd[2], cam_kind = hd[3], nx = hd[4], ny = hd[5], ns = hd[6], depth = hd[7];
    uint64_t *seed = (uint64_t *)rd(f, sizeof(uint64_t));
    int32_t *prim_kind = (int32_t *)rd(f, sizeof(int32_t) * n_prims);
    double *prim_geom = (double *)rd(f, sizeof(double) * n_prims * RTMI_PRIM_STRIDE);
    int32_t *prim_mat = (int32_t *)rd(f, sizeof(int32_t) * n_prims);
    int32_t *mat_kind = (int32_t *)rd(f, sizeof(int32_t) * n_mats);
    int32_t *mat_tex = (int32_t *)rd(f, sizeof(int32_t) * n_mats);
    double *mat_param = (double *)rd(f, sizeof(double) * n_mats);
    int32_t *tex_kind = (int32_t *)rd(f, sizeof(int32_t) * n_tex);
    double *tex_param = (double *)rd(f, sizeof(double) * n_tex * RTMI_TEX_STRIDE);
    int32_t *tex_child = (int32_t *)rd(f, sizeof(int32_t) * n_tex * 2);
    double *cam = (double *)rd(f, sizeof(double) * 24);
    fclose(f);

    rtmi_ctx *ctx = NULL, *ctx2 = NULL;
    rtmi_scene *scene = NULL, *clone = NULL;
    CHECK(p_rtmi_init(0, 0, &ctx));
    char arch[64];
    int32_t cus = 0;
    CHECK(p_rtmi_device_info(ctx, &cus, NULL, NULL, arch, 64));
    CHECK(p_rtmi_scene_create(ctx, n_prims, prim_kind, prim_geom, prim_mat, n_mats, mat_kind, mat_tex, mat_param, n_tex, tex_kind, tex_param, tex_child,
                              cam_kind, cam, &scene));
    const size_t npx = (size_t)nx * ny;
    double *lin = (double *)malloc(npx * 3 * sizeof(double));
    uint8_t *q = (uint8_t *)malloc(npx * 3);
    uint64_t cnt[2] = {0, 0};
    FILE *o = fopen(argv[3], "wb");
    if (!o) { perror(argv[3]); return 4; }
    CHECK(p_rtmi_render(scene, nx, ny, ns, depth, *seed, RTMI_F64, 0, 0, nx, ny, lin, q, cnt));
    fwrite(lin, sizeof(double), npx * 3, o); fwrite(q, 1, npx * 3, o); fwrite(cnt, sizeof(uint64_t), 2, o);
    /* one host process, two replicas (here on the same device: the gather is a device copy; on distinct devices ONE ncclGather) */
    CHECK(p_rtmi_init(0, 0, &ctx2));
    CHECK(p_rtmi_scene_clone(scene, ctx2, &clone));
    rtmi_scene *both[2] = {scene, clone};
    cnt[0] = cnt[1] = 0;
    CHECK(p_rtmi_render_multi(2, both, nx, ny, ns, depth, *seed, RTMI_F64, lin, q, cnt));
    fwrite(lin, sizeof(double), npx * 3, o); fwrite(q, 1, npx * 3, o); fwrite(cnt, sizeof(uint64_t), 2, o);
    if (getenv("RTMI_HOST_SMOKE_RCCL")) { /* the RCCL branch on a one-GPU host: a one-rank communicator */
        CHECK(p_rtmi_rccl_probe(NULL));
        printf("rccl probe ok\n");
        if (p_rtmi_rccl_probe("librccl_does_not_exist.so.9") == RTMI_OK) { fprintf(stderr, "probe of a missing library succeeded\n"); return 5; }
        printf("missing library: %s\n", p_rtmi_last_error());
        setenv("RTMI_MULTI_GATHER", "rccl", 1);
        rtmi_scene *one[1] = {scene};
        cnt[0] = cnt[1] = 0;
        CHECK(p_rtmi_render_multi(1, one, nx, ny, ns, depth, *seed, RTMI_F64, lin, q, cnt));
        int32_t path = -1;
        CHECK(p_rtmi_last_gather_path(ctx, &path));
        printf("gather path of the one-rank frame: %d\n", (int)path);
        fwrite(lin, sizeof(double), npx * 3, o); fwrite(q, 1, npx * 3, o); fwrite(cnt, sizeof(uint64_t), 2, o);
        unsetenv("RTMI_MULTI_GATHER");
    }
    fclose(o);
    CHECK(p_rtmi_scene_destroy(clone));
    CHECK(p_rtmi_scene_destroy(scene));
    CHECK(p_rtmi_shutdown(ctx2));
    CHECK(p_rtmi_shutdown(ctx));
    printf("host_smoke ok: backend=%s arch=%s cus=%d rays=%llu\n", p_rtmi_backend_name(), arch, cus, (unsigned long long)cnt[0]);
    return 0;
}

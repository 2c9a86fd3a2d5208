/* A host written in plain C99 against include/ptgpu.h: what a cgo / Rust-bindgen / ctypes consumer sees.
 * Renders a few samples of a two-primitive scene and prints the accumulator's checksum; the Python test compares it
 * with the same scene rendered through the ctypes binding.  Then the same frames through a one-device PtGroup (the
 * in-library multi-GPU path, RCCL exchange included) in the window loop's pattern - ptgpu_group_frame: one sample per
 * frame, each call handing out the newest frame that has arrived, a flush at the end - which must print the same sums. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ptgpu.h"

static PtVec3 v(float x, float y, float z) { PtVec3 r; r.x = x; r.y = y; r.z = z; return r; }

int main(int argc, char **argv) {
    uint32_t spp = argc > 1 ? (uint32_t)atoi(argv[1]) : 3u;
    PtConfig cfg;
    ptgpu_default_config(&cfg);
    PtHandle *h = NULL;
    int rc = ptgpu_create(&cfg, &h);
    if (rc != PT_OK) { fprintf(stderr, "ptgpu_create: %d\n", rc); return 2; }

    PtSphere sphere;
    memset(&sphere, 0, sizeof sphere);
    sphere.origin = v(0.0f, 1.0f, 0.0f);
    sphere.radius = 1.0f;
    sphere.material.kind = 3;  /* Translucent */
    sphere.material.ior = 1.5f;
    PtPlane floor_;
    memset(&floor_, 0, sizeof floor_);
    floor_.origin = v(-4.0f, 0.0f, 4.0f);
    floor_.u = v(8.0f, 0.0f, 0.0f);
    floor_.v = v(0.0f, 0.0f, -8.0f);
    floor_.material.kind = 4;  /* Physically */
    floor_.material.color = v(0.8f, 0.7f, 0.6f);
    floor_.material.roughness = 0.5f;
    ptgpu_plane_normals(&floor_, 1);

    PtSceneDesc sc;
    memset(&sc, 0, sizeof sc);
    sc.width = 32; sc.height = 24; sc.sub_pixel_sampling = 1;
    ptgpu_camera_new(v(0.0f, 2.0f, 6.0f), v(0.0f, 1.0f, 0.0f), v(0.0f, 1.0f, 0.0f), 4.0f, 3.0f, 2.0f, 0, 1.0f, 100.0f, &sc.camera);
    sc.sky.kind = 0;
    sc.sky.radiance = v(0.5f, 0.6f, 0.7f);
    sc.spheres = &sphere; sc.n_spheres = 1;
    sc.planes = &floor_; sc.n_planes = 1;
    if ((rc = ptgpu_upload_scene(h, &sc)) != PT_OK || (rc = ptgpu_render(h, spp)) != PT_OK) {
        fprintf(stderr, "error %d: %s\n", rc, ptgpu_last_error(h));
        return 3;
    }
    uint32_t *p32 = (uint32_t *)malloc(sizeof(uint32_t) * 32 * 24 * 3);
    uint64_t n = 0;
    if ((rc = ptgpu_read_pixels32(h, p32, &n)) != PT_OK) { fprintf(stderr, "error %d: %s\n", rc, ptgpu_last_error(h)); return 4; }
    uint64_t sum = 0, mix = 1469598103934665603ull;
    for (int i = 0; i < 32 * 24 * 3; ++i) { sum += p32[i]; mix = (mix ^ p32[i]) * 1099511628211ull; }
    const uint8_t *shown = NULL;
    if ((rc = ptgpu_present_pixels8(h, &shown)) != PT_OK) return 5;
    printf("samples %llu sum %llu fnv %llu first_byte %u\n", (unsigned long long)n, (unsigned long long)sum, (unsigned long long)mix, (unsigned)shown[0]);
    ptgpu_destroy(h);

    {
        PtGroup *g = NULL;
        int32_t device = 0;
        uint32_t frame;
        uint64_t gn = 0, gsum = 0, gmix = 1469598103934665603ull, bytes = 0;
        double ms = 0.0;
        const uint8_t *gshown = NULL;
        if ((rc = ptgpu_group_create(&cfg, &device, 1, &g)) != PT_OK) { fprintf(stderr, "ptgpu_group_create: %d\n", rc); return 6; }
        if ((rc = ptgpu_group_upload_scene(g, &sc)) != PT_OK) { fprintf(stderr, "error %d: %s\n", rc, ptgpu_group_last_error(g)); return 7; }
        for (frame = 0; frame <= spp; ++frame) {   /* src/window.rs:127-213: one sample per frame, shown one frame behind; then a flush */
            uint64_t showing = 0;
            if ((rc = ptgpu_group_frame(g, frame < spp ? 1u : 0u, &gshown, &showing)) != PT_OK) {
                fprintf(stderr, "error %d: %s\n", rc, ptgpu_group_last_error(g));
                return 8;
            }
            /* the newest frame that has arrived: at the oldest the one two before this call's; the flush: the last */
            if (showing > (frame < spp ? frame + 1u : spp) || showing + 2u < frame + 1u || showing == 0u || (frame == spp && showing != spp)) { fprintf(stderr, "frame %u shows %llu samples\n", (unsigned)frame, (unsigned long long)showing); return 8; }
        }
        if ((rc = ptgpu_group_read_pixels32(g, p32, &gn)) != PT_OK) { fprintf(stderr, "error %d: %s\n", rc, ptgpu_group_last_error(g)); return 9; }
        for (int i = 0; i < 32 * 24 * 3; ++i) { gsum += p32[i]; gmix = (gmix ^ p32[i]) * 1099511628211ull; }
        ptgpu_group_exchange_stats(g, &bytes, &ms);
        printf("group devices %u samples %llu sum %llu fnv %llu first_byte %u exchange_bytes %llu\n", (unsigned)ptgpu_group_size(g),
               (unsigned long long)gn, (unsigned long long)gsum, (unsigned long long)gmix, (unsigned)gshown[0], (unsigned long long)bytes);
        ptgpu_group_destroy(g);
    }
    free(p32);
    return 0;
}

// ptcli -- headless stand-in for the reference's window loop (reference src/window.rs:38-229).
//
// Same start-up behaviour: `--path=<scene>` (default scenes/sample/sample.scene, :45-53); a scene that fails to
// load prints the error and falls back to Scene::default() (:58-63); an HDRI sky is loaded and the accumulation
// restarted (:130-141).  Instead of presenting 1 spp per frame forever it renders `--spp` samples per pixel
// through libptgpu, prints the frame time the way the reference does (:227-228) and writes a PPM.
//
//   ptcli --path=scene.scene --spp=64 --out=image.ppm [--frames=N] [--depth=4] [--device=0] [--reference-comments] [--thin-lens]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "pt_host.hpp"

using namespace pthost;

static bool arg_value(int argc, char **argv, const char *key, std::string &out) {
    size_t n = std::strlen(key);
    for (int i = 1; i < argc; ++i)
        if (std::strncmp(argv[i], key, n) == 0) { out = argv[i] + n; return true; }
    return false;
}
static bool arg_flag(int argc, char **argv, const char *key) {
    for (int i = 1; i < argc; ++i) if (std::strcmp(argv[i], key) == 0) return true;
    return false;
}

int main(int argc, char **argv) {
    std::string path = "scenes/sample/sample.scene", v, out_path = "out.ppm";
    arg_value(argc, argv, "--path=", path);
    arg_value(argc, argv, "--out=", out_path);
    uint32_t spp = 16, frames = 0;
    int depth = 4, device = 0;
    if (arg_value(argc, argv, "--spp=", v)) spp = (uint32_t)std::strtoul(v.c_str(), nullptr, 10);
    if (arg_value(argc, argv, "--frames=", v)) frames = (uint32_t)std::strtoul(v.c_str(), nullptr, 10);
    if (arg_value(argc, argv, "--depth=", v)) depth = std::atoi(v.c_str());
    if (arg_value(argc, argv, "--device=", v)) device = std::atoi(v.c_str());
    uint32_t flags = arg_flag(argc, argv, "--reference-comments") ? PARSE_REFERENCE_COMMENT_TAIL : PARSE_DEFAULT;

    Scene scene;
    std::string err;
    if (!load_scene(path, scene, err, flags)) {
        std::printf("%s\n", err.c_str());
        std::printf("Using the default scene.\n");
        scene = Scene::default_scene();
    }
    if (scene.sky.kind == PT_SKY_HDRI) {
        try {
            scene.sky.texture = load_hdri_texture(scene.sky.path);
        } catch (const std::exception &e) {
            std::printf("%s\n", e.what());  // the reference's loader thread panics; the sky then stays white forever
        }
    }

    PtConfig cfg;
    ptgpu_default_config(&cfg);
    cfg.device = device;
    cfg.max_depth = depth;
    if (arg_flag(argc, argv, "--thin-lens")) cfg.camera_model = 1;  // CameraSampler::thin_lens_ray (reference src/tracer.rs:144-197)
    Renderer r;
    int rc = r.create(&cfg);
    if (rc) { std::fprintf(stderr, "ptgpu_create failed (%d): an MI355X (gfx950) is required, there is no CPU path\n", rc); return 2; }
    if ((rc = r.set_scene(scene))) { std::fprintf(stderr, "set_scene failed (%d): %s\n", rc, r.last_error()); return 3; }

    // `--frames=N` splits the work like the reference's loop (one render call per frame); default is one call
    uint32_t per_frame = frames ? (spp + frames - 1) / frames : spp;
    uint32_t done = 0;
    Backbuffer bb;
    const uint8_t *display = nullptr;
    while (done < spp) {
        uint32_t n = std::min(per_frame, spp - done);
        auto t0 = std::chrono::steady_clock::now();
        // a frame = the render call plus the display bytes, like process(tile) x tiles, wait(), glDrawPixels (src/window.rs:166-210)
        if ((rc = r.render(n)) || (rc = r.present(&display))) { std::fprintf(stderr, "render failed (%d): %s\n", rc, r.last_error()); return 4; }
        auto t1 = std::chrono::steady_clock::now();
        std::printf("frame_time = %f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count());
        done += n;
    }
    if ((rc = r.read(bb))) { std::fprintf(stderr, "readback failed (%d): %s\n", rc, r.last_error()); return 4; }
    PtStats st;
    r.stats(st);
    std::printf("%ux%u, %llu spp, %llu rays\n", bb.width, bb.height, (unsigned long long)bb.num_samples, (unsigned long long)st.rays);
    const bool pfm = out_path.size() > 4 && out_path.compare(out_path.size() - 4, 4, ".pfm") == 0;
    if (!(pfm ? write_pfm(out_path, bb) : write_ppm(out_path, bb))) { std::fprintf(stderr, "could not write %s\n", out_path.c_str()); return 5; }
    return 0;
}

// pt_host.cpp -- Scene/Renderer/Backbuffer plumbing above the C ABI, the Radiance .hdr loader and image export.
#include "pt_host.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace pthost {

// reference src/scene.rs:162-167
Scene Scene::default_scene() {
    Scene s;
    s.camera = Camera(vec3(0.0f, 2.0f, 20.0f), vec3(0, 0, 0), vec3(0.0f, 1.0f, 0.0f), 4.0f, 4.0f, 10.0f, ToneMapping::exposure_of(1.0f), 100.0f);
    s.image_settings = ImageSettings{256, 256, true};
    s.sky = Sky::constant(vec3(1.0f, 1.0f, 1.0f));
    return s;
}

PtSceneDesc Scene::desc() const {
    PtSceneDesc d;
    std::memset(&d, 0, sizeof d);
    d.width = (uint32_t)image_settings.width;
    d.height = (uint32_t)image_settings.height;
    d.sub_pixel_sampling = image_settings.sub_pixel_sampling ? 1 : 0;
    d.camera = camera.c;
    d.sky.kind = sky.kind;
    d.sky.radiance = sky.radiance;
    if (sky.kind == PT_SKY_HDRI && sky.texture) {
        d.sky.pixels = sky.texture->pixels.data();
        d.sky.width = sky.texture->width;
        d.sky.height = sky.texture->height;
    }
    d.spheres = spheres.data(); d.n_spheres = spheres.size();
    d.planes = planes.data(); d.n_planes = planes.size();
    d.emissive_spheres = emissive_spheres.data(); d.n_emissive_spheres = emissive_spheres.size();
    d.emissive_planes = emissive_planes.data(); d.n_emissive_planes = emissive_planes.size();
    return d;
}

// ---- Renderer -------------------------------------------------------------------------------------------
Renderer::~Renderer() { if (h_) ptgpu_destroy(h_); }

int Renderer::create(const PtConfig *cfg) {
    PtConfig c;
    if (cfg) c = *cfg; else ptgpu_default_config(&c);
    if (h_) { ptgpu_destroy(h_); h_ = nullptr; }
    return ptgpu_create(&c, &h_);
}

int Renderer::set_scene(const Scene &scene) {
    if (!h_) return PT_ERR_INVALID_ARGUMENT;
    // the reference's Backbuffer is sized from image_settings as u32 (src/window.rs:65-66,104)
    if (scene.image_settings.width == 0 || scene.image_settings.height == 0 || scene.image_settings.width > 0xFFFFFFFFull ||
        scene.image_settings.height > 0xFFFFFFFFull)
        return PT_ERR_INVALID_ARGUMENT;
    PtSceneDesc d = scene.desc();
    int rc = ptgpu_upload_scene(h_, &d);
    if (rc == PT_OK) { w_ = d.width; h_px_ = d.height; }
    return rc;
}

int Renderer::clear() { return h_ ? ptgpu_clear(h_) : PT_ERR_INVALID_ARGUMENT; }
int Renderer::render(uint32_t n) { return h_ ? ptgpu_render(h_, n) : PT_ERR_INVALID_ARGUMENT; }
int Renderer::stats(PtStats &st) { return h_ ? ptgpu_get_stats(h_, &st) : PT_ERR_INVALID_ARGUMENT; }
const char *Renderer::last_error() const { return h_ ? ptgpu_last_error(h_) : "renderer not created"; }

int Renderer::read(Backbuffer &bb) {
    if (!h_) return PT_ERR_INVALID_ARGUMENT;
    bb.width = w_; bb.height = h_px_;
    bb.pixels32.resize((size_t)w_ * h_px_ * 3);
    bb.pixels8.resize((size_t)w_ * h_px_ * 3);
    int rc = ptgpu_read_pixels32(h_, bb.pixels32.data(), &bb.num_samples);
    if (rc) return rc;
    return ptgpu_read_pixels8(h_, bb.pixels8.data());
}

int Renderer::present(const uint8_t **pixels8) { return h_ ? ptgpu_present_pixels8(h_, pixels8) : PT_ERR_INVALID_ARGUMENT; }

// ---- Radiance .hdr (RGBE) --------------------------------------------------------------------------------
// What stb_image's HDR path does for a 3-component float load (the reference calls stb_image::image::load,
// src/content.rs:17): header "#?RADIANCE"/"#?RGBE", FORMAT=32-bit_rle_rgbe, "-Y h +X w", scanlines either flat
// or new-style RLE; texel = mantissa * 2^(e - 136), black when e == 0; rows top to bottom.
namespace {
void rgbe_to_float(const unsigned char *rgbe, float *out) {
    if (rgbe[3] != 0) {
        float f = std::ldexp(1.0f, (int)rgbe[3] - (128 + 8));
        out[0] = rgbe[0] * f; out[1] = rgbe[1] * f; out[2] = rgbe[2] * f;
    } else {
        out[0] = out[1] = out[2] = 0.0f;
    }
}
}  // namespace

std::shared_ptr<HDRITexture> load_hdri_texture(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("Could not load the image \"" + path + "\"! Error: can't fopen");
    std::string line;
    std::getline(f, line);
    if (line != "#?RADIANCE" && line != "#?RGBE") throw std::runtime_error("Image \"" + path + "\" is not an HDR image.");
    bool format_ok = false;
    while (std::getline(f, line)) {
        if (line.empty()) break;
        if (line == "FORMAT=32-bit_rle_rgbe") format_ok = true;
    }
    if (!format_ok) throw std::runtime_error("Could not load the image \"" + path + "\"! Error: unsupported format");
    std::getline(f, line);
    int h = 0, w = 0;
    if (std::sscanf(line.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0)
        throw std::runtime_error("Could not load the image \"" + path + "\"! Error: unsupported data layout");
    auto tex = std::make_shared<HDRITexture>();
    tex->width = (size_t)w; tex->height = (size_t)h;
    tex->pixels.resize((size_t)w * h * 3);
    std::vector<unsigned char> scan((size_t)w * 4);
    auto need = [&](void *dst, size_t n) {
        f.read((char *)dst, (std::streamsize)n);
        if ((size_t)f.gcount() != n) throw std::runtime_error("Could not load the image \"" + path + "\"! Error: truncated file");
    };
    if (w < 8 || w >= 32768) {  // flat data
        for (size_t i = 0; i < (size_t)w * h; ++i) { unsigned char px[4]; need(px, 4); rgbe_to_float(px, &tex->pixels[3 * i]); }
        return tex;
    }
    for (int y = 0; y < h; ++y) {
        unsigned char hd[4];
        need(hd, 4);
        if (hd[0] != 2 || hd[1] != 2 || (hd[2] & 0x80)) {
            // not run-length encoded: this pixel is plain data and so is the rest of the file
            if (y != 0) throw std::runtime_error("Could not load the image \"" + path + "\"! Error: invalid scanline");
            rgbe_to_float(hd, &tex->pixels[0]);
            for (size_t i = 1; i < (size_t)w * h; ++i) { unsigned char px[4]; need(px, 4); rgbe_to_float(px, &tex->pixels[3 * i]); }
            return tex;
        }
        if ((((int)hd[2]) << 8 | hd[3]) != w) throw std::runtime_error("Could not load the image \"" + path + "\"! Error: invalid decoded scanline length");
        for (int k = 0; k < 4; ++k) {
            int i = 0;
            while (i < w) {
                unsigned char count, value;
                need(&count, 1);
                if (count > 128) {
                    count -= 128;
                    if (i + count > w) throw std::runtime_error("Could not load the image \"" + path + "\"! Error: corrupt RLE");
                    need(&value, 1);
                    for (int z = 0; z < count; ++z) scan[(size_t)(i++) * 4 + k] = value;
                } else {
                    if (count == 0 || i + count > w) throw std::runtime_error("Could not load the image \"" + path + "\"! Error: corrupt RLE");
                    for (int z = 0; z < count; ++z) { need(&value, 1); scan[(size_t)(i++) * 4 + k] = value; }
                }
            }
        }
        for (int x = 0; x < w; ++x) rgbe_to_float(&scan[(size_t)x * 4], &tex->pixels[((size_t)y * w + x) * 3]);
    }
    return tex;
}

// ---- export ----------------------------------------------------------------------------------------------
bool write_ppm(const std::string &path, const Backbuffer &bb) {
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    f << "P6\n" << bb.width << " " << bb.height << "\n255\n";
    for (uint32_t y = 0; y < bb.height; ++y) {
        const uint8_t *row = &bb.pixels8[(size_t)(bb.height - 1 - y) * bb.width * 3];  // row 0 of the backbuffer is the bottom
        f.write((const char *)row, (std::streamsize)bb.width * 3);
    }
    return (bool)f;
}

// Portable float map of the accumulated mean (the value assign_pixel8 scales by iso and clamps, reference
// src/tracer.rs:271-281: sum / (num_samples * 255)), for offline comparison of renders without the 8-bit step.
// Little-endian (negative scale), rows bottom to top as the format defines - the backbuffer's own order.
bool write_pfm(const std::string &path, const Backbuffer &bb) {
    std::ofstream f(path, std::ios::binary);
    if (!f || bb.num_samples == 0) return false;
    f << "PF\n" << bb.width << " " << bb.height << "\n-1.0\n";
    std::vector<float> row((size_t)bb.width * 3);
    const float denom = (float)bb.num_samples * 255.0f;
    for (uint32_t y = 0; y < bb.height; ++y) {
        for (size_t k = 0; k < row.size(); ++k) row[k] = (float)bb.pixels32[(size_t)y * bb.width * 3 + k] / denom;
        f.write((const char *)row.data(), (std::streamsize)(row.size() * sizeof(float)));
    }
    return (bool)f;
}

}  // namespace pthost

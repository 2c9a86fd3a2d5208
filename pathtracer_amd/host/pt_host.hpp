// pt_host.hpp -- C++ host side above the C ABI (include/ptgpu.h).
//
// Mirrors the reference's host-visible types and entry points for the hot path so that code written against
// the reference reads the same here (paths relative to the reference checkout):
//   scene::{Material, Sphere, Plane, Sky, Scene}         src/scene.rs:71-167
//   tracer::{ImageSettings, ToneMapping, Camera, Backbuffer}   src/tracer.rs:8-13,43-66,200-283
//   parser::parse_scene                                   src/parser.rs:50-112 (grammar :114-649)
//   content::load_hdri_texture                            src/content.rs:13-26 (stb_image's Radiance .hdr decode)
//   the WorkerPool + tracer::render pair                  src/window.rs:109-116,166-196 -> Renderer
// The reference is Rust; no Rust toolchain exists in this environment, hence C++ here.  All rendering goes
// through libptgpu (HIP); nothing in this directory traces rays on the CPU.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/ptgpu.h"

namespace pthost {

using Vec3 = PtVec3;
inline Vec3 vec3(float x, float y, float z) { return Vec3{x, y, z}; }

// src/scene.rs:84-92
struct Material {
    enum Kind : int32_t { None = PT_MAT_NONE, Emissive = PT_MAT_EMISSIVE, Mirror = PT_MAT_MIRROR, Translucent = PT_MAT_TRANSLUCENT,
                          Physically = PT_MAT_PHYSICALLY };
    PtMaterial m{PT_MAT_NONE, {0, 0, 0}, 0.0f, 0.0f, 1.0f};
    static Material none() { return Material{}; }
    static Material emissive(Vec3 radiance) { Material r; r.m.kind = Emissive; r.m.color = radiance; return r; }
    static Material mirror() { Material r; r.m.kind = Mirror; return r; }
    static Material translucent(float ior) { Material r; r.m.kind = Translucent; r.m.ior = ior; return r; }
    static Material physically(Vec3 reflectivity, float roughness, float metalness) {
        Material r; r.m.kind = Physically; r.m.color = reflectivity; r.m.roughness = roughness; r.m.metalness = metalness; return r;
    }
    bool is_emissive() const { return m.kind == Emissive; }
};

// src/scene.rs:128-133
struct Sphere {
    PtSphere s;
    Sphere(Vec3 origin, float radius, const Material &material) : s{origin, radius, material.m} {}
};

// src/scene.rs:95-114: Plane::new computes the normal
struct Plane {
    PtPlane p;
    Plane(Vec3 origin, Vec3 u, Vec3 v, const Material &material) : p{origin, u, v, material.m, {0, 0, 0}} {
        ptgpu_plane_normal(u, v, &p.normal);
    }
};

// src/tracer.rs:43-48
struct ToneMapping {
    int32_t kind = PT_TONE_CLAMP;
    float exposure = 1.0f;
    static ToneMapping clamp() { return {PT_TONE_CLAMP, 1.0f}; }
    static ToneMapping reinhard() { return {PT_TONE_REINHARD, 1.0f}; }
    static ToneMapping exposure_of(float value) { return {PT_TONE_EXPOSURE, value}; }
};

// src/tracer.rs:57-96
struct Camera {
    PtCamera c{};
    Vec3 target{0, 0, 0}, up{0, 1, 0};  // Camera::new consumes these; kept only so a scene can be written back out
    Camera() : Camera(vec3(0.0f, 2.0f, 20.0f), vec3(0, 0, 0), vec3(0.0f, 1.0f, 0.0f), 4.0f, 4.0f, 10.0f, ToneMapping::clamp(), 100.0f) {}
    Camera(Vec3 position, Vec3 target_, Vec3 up_, float width, float height, float z_near, ToneMapping tm, float iso)
        : target(target_), up(up_) {
        ptgpu_camera_new(position, target_, up_, width, height, z_near, tm.kind, tm.exposure, iso, &c);
    }
};

// src/scene.rs:7-30
struct HDRITexture {
    std::vector<float> pixels;  // width*height*3, rows top to bottom as stb_image delivers them
    size_t width = 0, height = 0;
};
// src/content.rs:13-26; throws std::runtime_error like the reference panics
std::shared_ptr<HDRITexture> load_hdri_texture(const std::string &path);

// src/scene.rs:71-75
struct Sky {
    int32_t kind = PT_SKY_CONSTANT;
    Vec3 radiance{1.0f, 1.0f, 1.0f};
    std::string path;
    std::shared_ptr<HDRITexture> texture;  // None until the content loader delivers it
    static Sky constant(Vec3 r) { Sky s; s.kind = PT_SKY_CONSTANT; s.radiance = r; return s; }
    static Sky hdri(const std::string &p) { Sky s; s.kind = PT_SKY_HDRI; s.path = p; return s; }
};

// src/tracer.rs:8-13
struct ImageSettings { size_t width = 256, height = 256; bool sub_pixel_sampling = true; };

// src/scene.rs:151-167
struct Scene {
    ImageSettings image_settings;
    Camera camera;
    Sky sky;
    std::vector<PtSphere> spheres;
    std::vector<PtPlane> planes;
    std::vector<PtSphere> emissive_spheres;
    std::vector<PtPlane> emissive_planes;
    static Scene default_scene();  // Scene::default(): empty, white sky, 256x256, Exposure(1.0), iso 100
    PtSceneDesc desc() const;      // view for ptgpu_upload_scene (valid while *this is alive and unchanged)
};

// src/parser.rs:6-21
struct ParseError { std::string message; size_t line = 0, column = 0; };

enum ParseFlags : uint32_t {
    PARSE_DEFAULT = 0,
    // src/parser.rs:467-476 (parse_until_newline) consumes all but the LAST character of a `//` comment, so in the
    // reference any comment that is not empty leaves one stray character behind and the scene fails to parse.
    // Off by default (comments run to the end of the line); set it to reproduce the reference exactly.
    PARSE_REFERENCE_COMMENT_TAIL = 1u,
};
// src/parser.rs:50-112.  Returns false and fills `err` on failure (the reference returns Err(ParseError)).
bool parse_scene(const std::string &content, Scene &out, ParseError &err, uint32_t flags = PARSE_DEFAULT);
// src/window.rs:28-36
bool load_scene(const std::string &filename, Scene &out, std::string &err, uint32_t flags = PARSE_DEFAULT);
// Inverse of parse_scene in the same dialect (one field per line, floats with a decimal point and no exponent).
std::string write_scene(const Scene &scene);

// src/tracer.rs:200-283: host copy of what the reference keeps in Backbuffer
struct Backbuffer {
    uint32_t width = 0, height = 0;
    uint64_t num_samples = 0;
    std::vector<uint32_t> pixels32;  // 3 per pixel
    std::vector<uint8_t> pixels8;    // 3 per pixel, row 0 = bottom (src/tracer.rs:91, src/window.rs:205-210)
};

// Replaces `WorkerPool::new(8, |tile| tracer::render(tile, &backbuffer, scene))` + process()/wait()
// (src/window.rs:109-116,166-196).  Every method returns a PtStatus (0 = ok); last_error() has the text.
class Renderer {
public:
    Renderer() = default;
    ~Renderer();
    Renderer(const Renderer &) = delete;
    Renderer &operator=(const Renderer &) = delete;
    int create(const PtConfig *cfg = nullptr);     // nullptr: ptgpu_default_config
    int set_scene(const Scene &scene);             // wherever the reference does `*scene = ...; backbuffer.clear()`
    int clear();                                   // Backbuffer::clear
    int render(uint32_t n_samples = 1);            // n frames of process(tile) x tiles + wait()
    int read(Backbuffer &bb);                      // pixels32, pixels8, num_samples
    int present(const uint8_t **pixels8);          // what the window loop hands to glDrawPixels every frame (src/window.rs:205-210):
                                                   // the display bytes in the library's page-locked buffer, valid until the next call
    int stats(PtStats &st);
    const char *last_error() const;
    PtHandle *handle() { return h_; }
private:
    PtHandle *h_ = nullptr;
    uint32_t w_ = 0, h_px_ = 0;
};

// Image export (the reference has none: it only presents to a window).  Rows are flipped so row 0 is the top.
bool write_ppm(const std::string &path, const Backbuffer &bb);
bool write_pfm(const std::string &path, const Backbuffer &bb);  // accumulated mean as floats (.pfm)

}  // namespace pthost

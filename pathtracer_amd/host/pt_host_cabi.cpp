// pt_host_cabi.cpp -- plain-C window onto the C++ host layer (parser, scene writer, .hdr loader) so it can be
// driven from tests and scripts.  No rendering here; see include/ptgpu.h for that.
#include <cstring>
#include <stdexcept>
#include <string>

#include "pt_host.hpp"

using namespace pthost;

struct PtHostScene {
    Scene scene;
    std::string error;
    size_t err_line = 0, err_column = 0;
    std::string text;  // write_scene output
};

extern "C" {

// Parses `.scene` text (reference src/parser.rs:50).  Always returns an object; *ok says whether parsing succeeded.
PtHostScene *pthost_parse_scene(const char *text, uint64_t len, uint32_t flags, int32_t *ok) {
    auto *s = new PtHostScene();
    ParseError err;
    bool good = parse_scene(std::string(text, (size_t)len), s->scene, err, flags);
    if (!good) { s->error = err.message; s->err_line = err.line; s->err_column = err.column; }
    if (ok) *ok = good ? 1 : 0;
    return s;
}
void pthost_scene_free(PtHostScene *s) { delete s; }
const char *pthost_scene_error(const PtHostScene *s, uint64_t *line, uint64_t *column) {
    if (line) *line = s->err_line;
    if (column) *column = s->err_column;
    return s->error.c_str();
}
// The parsed scene as the PtSceneDesc the renderer takes (valid until pthost_scene_free).
void pthost_scene_desc(const PtHostScene *s, PtSceneDesc *out) { *out = s->scene.desc(); }
const char *pthost_scene_sky_path(const PtHostScene *s) { return s->scene.sky.path.c_str(); }
void pthost_scene_lookat(const PtHostScene *s, PtVec3 *target, PtVec3 *up) { *target = s->scene.camera.target; *up = s->scene.camera.up; }
// Scene text in the same dialect.
const char *pthost_scene_write(PtHostScene *s) { s->text = write_scene(s->scene); return s->text.c_str(); }
// Loads the sky texture named by the scene (reference src/content.rs:13-26 + src/window.rs:130-141); 0 on success.
int32_t pthost_scene_load_sky(PtHostScene *s, const char *base_dir) {
    if (s->scene.sky.kind != PT_SKY_HDRI) return 0;
    try {
        std::string p = s->scene.sky.path;
        if (base_dir && *base_dir && !p.empty() && p[0] != '/') p = std::string(base_dir) + "/" + p;
        s->scene.sky.texture = load_hdri_texture(p);
        return 0;
    } catch (const std::exception &e) {
        s->error = e.what();
        return -1;
    }
}

// Standalone .hdr decode: returns a malloc'ed width*height*3 float array (free with pthost_free) or NULL.
float *pthost_load_hdr(const char *path, uint64_t *width, uint64_t *height) {
    try {
        auto t = load_hdri_texture(path);
        float *out = (float *)std::malloc(t->pixels.size() * sizeof(float));
        if (!out) return nullptr;
        std::memcpy(out, t->pixels.data(), t->pixels.size() * sizeof(float));
        *width = t->width; *height = t->height;
        return out;
    } catch (const std::exception &) {
        return nullptr;
    }
}
void pthost_free(void *p) { std::free(p); }

}  // extern "C"

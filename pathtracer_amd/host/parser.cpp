// parser.cpp -- reader and writer for the reference's `.scene` dialect.
//
// Follows the reference's hand-written recursive-descent parser rule by rule (reference src/parser.rs:50-649),
// including its observable quirks:
//   * fixed field order inside every block; `sphere`, `plane` and `physically` need at least one newline between
//     fields (parse_free_and_newline, :641-649); the other blocks do not;
//   * floats are [sign][sign]digits "." digits -- no exponent, no bare integers, no ".5" (:615-633); a doubled sign
//     is accepted by the grammar and then fails the numeric conversion (the reference panics; an error here);
//   * `sub_pixel_sampling = false` parses as TRUE (:591-593);
//   * keywords are prefix matches (:534-546); emissive primitives are routed to their own lists (:76-94);
//   * any failure inside a block surfaces as the one generic error at the block's start (:108).
// Only ASCII white space and digits are recognised (Rust's char::is_whitespace / is_numeric also accept other
// Unicode classes).
#include <cerrno>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "pt_host.hpp"

namespace pthost {
namespace {

struct Ctx {
    const char *p;
    size_t n;          // bytes left
    size_t line, column;
    uint32_t flags;
};

bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; }
bool is_digit(char c) { return c >= '0' && c <= '9'; }

void advance_cols(Ctx &c, size_t k) { c.p += k; c.n -= k; c.column += k; }

// :484-503
bool parse_whitespace(Ctx &c) {
    size_t k = 0;
    while (k < c.n && is_ws(c.p[k])) {
        if (c.p[k] == '\n') { c.line += 1; c.column = 1; } else c.column += 1;
        ++k;
    }
    c.p += k; c.n -= k;
    return k > 0;
}

// :534-546
bool parse_string(Ctx &c, const char *s) {
    size_t len = std::strlen(s);
    if (c.n < len || std::memcmp(c.p, s, len) != 0) return false;
    advance_cols(c, len);
    return true;
}

// :478-482 with :467-476
bool parse_one_line_comment(Ctx &c) {
    Ctx t = c;
    if (!parse_string(t, "//")) return false;
    size_t k = 0;
    while (k < t.n && t.p[k] != '\n') ++k;
    if ((t.flags & PARSE_REFERENCE_COMMENT_TAIL) && k > 0) k -= 1;  // `comment.next()` before `count()`
    advance_cols(t, k);
    c = t;
    return true;
}

// :505-526
void parse_free(Ctx &c) {
    for (;;) {
        if (parse_whitespace(c)) continue;
        if (parse_one_line_comment(c)) continue;
        break;
    }
}

bool parse_free_and_string(Ctx &c, const char *s) {  // :528-532
    Ctx t = c;
    parse_free(t);
    if (!parse_string(t, s)) return false;
    c = t;
    return true;
}

// :641-649
bool parse_free_and_newline(Ctx &c) {
    Ctx t = c;
    parse_free(t);
    if (t.line == c.line) return false;
    c = t;
    return true;
}

// :548-561
void parse_sign(Ctx &c, int &sign) {
    sign = 1;
    if (parse_string(c, "-")) { sign = -1; return; }
    if (parse_string(c, "+")) return;
}

// :598-613; the reference unwraps `parse::<i32>()`, i.e. panics on overflow -> failure here
bool parse_i32(Ctx &c, long long &value) {
    Ctx t = c;
    int sign;
    parse_sign(t, sign);
    size_t k = 0;
    while (k < t.n && is_digit(t.p[k])) ++k;
    if (k == 0) return false;
    long long v = 0;
    for (size_t i = 0; i < k; ++i) {
        v = v * 10 + (t.p[i] - '0');
        if (v > (long long)INT_MAX) return false;
    }
    advance_cols(t, k);
    value = sign * v;
    c = t;
    return true;
}

// :563-576
bool parse_u32(Ctx &c) {
    size_t k = 0;
    unsigned long long v = 0;
    while (k < c.n && is_digit(c.p[k])) {
        v = v * 10 + (unsigned)(c.p[k] - '0');
        if (v > 0xFFFFFFFFull) return false;
        ++k;
    }
    if (k == 0) return false;
    advance_cols(c, k);
    return true;
}

// :615-633
bool parse_f32(Ctx &c, float &value) {
    Ctx t = c;
    int sign;
    long long ipart;
    parse_sign(t, sign);
    if (!parse_i32(t, ipart)) return false;  // which accepts a second sign
    if (!parse_string(t, ".")) return false;
    if (!parse_u32(t)) return false;
    size_t len = t.column - c.column;
    std::string chars(c.p, len);
    // Rust's f32::from_str takes one optional sign; "--1.0" or "+-1.0" is an Err the reference unwraps
    size_t digits_at = (chars[0] == '+' || chars[0] == '-') ? 1 : 0;
    if (!is_digit(chars[digits_at])) return false;
    errno = 0;
    char *end = nullptr;
    value = std::strtof(chars.c_str(), &end);  // correctly rounded, like Rust's parse
    if (end != chars.c_str() + len) return false;
    c = t;
    return true;
}

bool parse_free_and_f32(Ctx &c, float &v) { Ctx t = c; parse_free(t); if (!parse_f32(t, v)) return false; c = t; return true; }
bool parse_free_and_i32(Ctx &c, long long &v) { Ctx t = c; parse_free(t); if (!parse_i32(t, v)) return false; c = t; return true; }

// :584-596 -- "false" yields true as well
bool parse_free_and_bool(Ctx &c, bool &v) {
    Ctx t = c;
    parse_free(t);
    if (parse_free_and_string(t, "true")) { v = true; c = t; return true; }
    if (parse_free_and_string(t, "false")) { v = true; c = t; return true; }
    return false;
}

// :259-273
bool parse_free_and_vec3(Ctx &c, Vec3 &v) {
    Ctx t = c;
    if (!parse_free_and_string(t, "(")) return false;
    if (!parse_free_and_f32(t, v.x)) return false;
    if (!parse_free_and_string(t, ",")) return false;
    if (!parse_free_and_f32(t, v.y)) return false;
    if (!parse_free_and_string(t, ",")) return false;
    if (!parse_free_and_f32(t, v.z)) return false;
    if (!parse_free_and_string(t, ")")) return false;
    c = t;
    return true;
}

bool key_eq(Ctx &c, const char *key) { return parse_free_and_string(c, key) && parse_free_and_string(c, "="); }

// :428-443
bool parse_free_and_path(Ctx &c, std::string &path) {
    Ctx t = c;
    parse_free(t);
    size_t k = 0;
    auto ok = [](char ch) {
        return (ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z') || is_digit(ch) || ch == '\\' || ch == '/' || ch == '_' || ch == '.';
    };
    while (k < t.n && ok(t.p[k])) ++k;
    if (k == 0) return false;
    path.assign(t.p, k);
    advance_cols(t, k);
    c = t;
    return true;
}

// :275-357
bool parse_free_and_material(Ctx &c, Material &m) {
    Ctx t = c;
    if (parse_free_and_string(t, "physically")) {  // :294-323
        Vec3 refl; float rough, metal;
        if (!parse_free_and_string(t, "{")) return false;
        if (!key_eq(t, "reflectivity") || !parse_free_and_vec3(t, refl)) return false;
        if (!parse_free_and_newline(t)) return false;
        if (!key_eq(t, "roughness") || !parse_free_and_f32(t, rough)) return false;
        if (!parse_free_and_newline(t)) return false;
        if (!key_eq(t, "metalness") || !parse_free_and_f32(t, metal)) return false;
        if (!parse_free_and_string(t, "}")) return false;
        m = Material::physically(refl, rough, metal);
    } else if (parse_free_and_string(t, "emissive")) {  // :325-335
        Vec3 rad;
        if (!parse_free_and_string(t, "{")) return false;
        if (!key_eq(t, "radiance") || !parse_free_and_vec3(t, rad)) return false;
        if (!parse_free_and_string(t, "}")) return false;
        m = Material::emissive(rad);
    } else if (parse_free_and_string(t, "translucent")) {  // :337-347
        float ior;
        if (!parse_free_and_string(t, "{")) return false;
        if (!key_eq(t, "ior") || !parse_free_and_f32(t, ior)) return false;
        if (!parse_free_and_string(t, "}")) return false;
        m = Material::translucent(ior);
    } else {
        return false;
    }
    c = t;
    return true;
}

// :114-135
bool parse_image(Ctx &c, ImageSettings &img) {
    Ctx t = c;
    parse_free(t);
    long long w, h; bool sps;
    if (!parse_free_and_string(t, "image") || !parse_free_and_string(t, "{")) return false;
    if (!key_eq(t, "width") || !parse_free_and_i32(t, w)) return false;
    if (!key_eq(t, "height") || !parse_free_and_i32(t, h)) return false;
    if (!key_eq(t, "sub_pixel_sampling") || !parse_free_and_bool(t, sps)) return false;
    if (!parse_free_and_string(t, "}")) return false;
    img.width = (size_t)w; img.height = (size_t)h; img.sub_pixel_sampling = sps;  // `as usize`
    c = t;
    return true;
}

// :359-394
bool parse_sphere(Ctx &c, std::vector<PtSphere> &plain, std::vector<PtSphere> &emissive) {
    Ctx t = c;
    Vec3 origin; float radius; Material m;
    if (!parse_free_and_string(t, "sphere") || !parse_free_and_string(t, "{")) return false;
    if (!key_eq(t, "origin") || !parse_free_and_vec3(t, origin)) return false;
    if (!parse_free_and_newline(t)) return false;
    if (!key_eq(t, "radius") || !parse_free_and_f32(t, radius)) return false;
    if (!parse_free_and_newline(t)) return false;
    if (!key_eq(t, "material") || !parse_free_and_material(t, m)) return false;
    if (!parse_free_and_string(t, "}")) return false;
    (m.is_emissive() ? emissive : plain).push_back(Sphere(origin, radius, m).s);  // :76-84
    c = t;
    return true;
}

// :396-426
bool parse_plane(Ctx &c, std::vector<PtPlane> &plain, std::vector<PtPlane> &emissive) {
    Ctx t = c;
    Vec3 origin, u, v; Material m;
    if (!parse_free_and_string(t, "plane") || !parse_free_and_string(t, "{")) return false;
    if (!key_eq(t, "origin") || !parse_free_and_vec3(t, origin)) return false;
    if (!parse_free_and_newline(t)) return false;
    if (!key_eq(t, "u") || !parse_free_and_vec3(t, u)) return false;
    if (!parse_free_and_newline(t)) return false;
    if (!key_eq(t, "v") || !parse_free_and_vec3(t, v)) return false;
    if (!parse_free_and_newline(t)) return false;
    if (!key_eq(t, "material") || !parse_free_and_material(t, m)) return false;
    if (!parse_free_and_string(t, "}")) return false;
    (m.is_emissive() ? emissive : plain).push_back(Plane(origin, u, v, m).p);  // :86-94
    c = t;
    return true;
}

// :209-257
bool parse_sky(Ctx &c, Sky &sky) {
    Ctx t = c;
    if (!parse_free_and_string(t, "sky") || !parse_free_and_string(t, "{")) return false;
    parse_free(t);
    if (parse_free_and_string(t, "constant")) {
        Vec3 r;
        if (!parse_free_and_string(t, "{") || !key_eq(t, "radiance") || !parse_free_and_vec3(t, r) || !parse_free_and_string(t, "}")) return false;
        sky = Sky::constant(r);
    } else if (parse_free_and_string(t, "hdri")) {
        std::string path;
        if (!parse_free_and_string(t, "{") || !key_eq(t, "path") || !parse_free_and_path(t, path) || !parse_free_and_string(t, "}")) return false;
        sky = Sky::hdri(path);
    } else {
        return false;
    }
    if (!parse_free_and_string(t, "}")) return false;
    c = t;
    return true;
}

// :137-207
bool parse_camera(Ctx &c, Camera &cam) {
    Ctx t = c;
    parse_free(t);
    Vec3 position, target, up; float width, height, z_near, iso;
    ToneMapping tm;
    if (!parse_free_and_string(t, "camera") || !parse_free_and_string(t, "{")) return false;
    if (!key_eq(t, "position") || !parse_free_and_vec3(t, position)) return false;
    if (!key_eq(t, "target") || !parse_free_and_vec3(t, target)) return false;
    if (!key_eq(t, "up") || !parse_free_and_vec3(t, up)) return false;
    if (!key_eq(t, "width") || !parse_free_and_f32(t, width)) return false;
    if (!key_eq(t, "height") || !parse_free_and_f32(t, height)) return false;
    if (!key_eq(t, "z_near") || !parse_free_and_f32(t, z_near)) return false;
    if (!key_eq(t, "tone_mapping")) return false;
    parse_free(t);
    if (parse_free_and_string(t, "clamp")) tm = ToneMapping::clamp();
    else if (parse_free_and_string(t, "reinhard")) tm = ToneMapping::reinhard();
    else {
        float value;
        if (!parse_free_and_string(t, "exposure") || !parse_free_and_string(t, "{") || !key_eq(t, "value") ||
            !parse_free_and_f32(t, value) || !parse_free_and_string(t, "}"))
            return false;
        tm = ToneMapping::exposure_of(value);
    }
    if (!key_eq(t, "iso") || !parse_free_and_f32(t, iso)) return false;
    if (!parse_free_and_string(t, "}")) return false;
    cam = Camera(position, target, up, width, height, z_near, tm, iso);
    c = t;
    return true;
}

std::string fmt_f32(float f) {
    // fixed notation with a decimal point and enough digits to round-trip; the grammar has no exponent form
    char buf[128];
    std::snprintf(buf, sizeof buf, "%.50f", (double)f);
    std::string s(buf);
    // shortest prefix that still parses back to the same float
    size_t dot = s.find('.');
    for (size_t digits = 1; dot + 1 + digits <= s.size(); ++digits) {
        std::string cand = s.substr(0, dot + 1 + digits);
        if (std::strtof(cand.c_str(), nullptr) == f) return cand;
    }
    return s;
}
std::string fmt_v3(Vec3 v) { return "(" + fmt_f32(v.x) + ", " + fmt_f32(v.y) + ", " + fmt_f32(v.z) + ")"; }

std::string fmt_material(const PtMaterial &m) {
    std::ostringstream o;
    if (m.kind == PT_MAT_EMISSIVE) o << "emissive {\n        radiance = " << fmt_v3(m.color) << "\n    }";
    else if (m.kind == PT_MAT_TRANSLUCENT) o << "translucent {\n        ior = " << fmt_f32(m.ior) << "\n    }";
    else o << "physically {\n        reflectivity = " << fmt_v3(m.color) << "\n        roughness = " << fmt_f32(m.roughness)
           << "\n        metalness = " << fmt_f32(m.metalness) << "\n    }";
    return o.str();
}

}  // namespace

bool parse_scene(const std::string &content, Scene &out, ParseError &err, uint32_t flags) {
    Ctx c{content.data(), content.size(), 1, 1, flags};
    Scene sc;  // parse_scene's defaults (:55-61): 256x256 jittered, white constant sky, default camera with Clamp
    for (;;) {
        parse_free(c);
        if (c.n == 0) break;
        if (parse_image(c, sc.image_settings)) continue;
        if (parse_sphere(c, sc.spheres, sc.emissive_spheres)) continue;
        if (parse_plane(c, sc.planes, sc.emissive_planes)) continue;
        if (parse_sky(c, sc.sky)) continue;
        if (parse_camera(c, sc.camera)) continue;
        err.message = "Expected \"camera\", \"sphere\", \"plane\" or \"sky\".";
        err.line = c.line; err.column = c.column;
        return false;
    }
    out = std::move(sc);
    return true;
}

bool load_scene(const std::string &filename, Scene &out, std::string &err, uint32_t flags) {
    std::ifstream f(filename, std::ios::binary);
    if (!f) { err = "Could not read the file \"" + filename + "\"."; return false; }
    std::stringstream ss;
    ss << f.rdbuf();
    ParseError pe;
    if (!parse_scene(ss.str(), out, pe, flags)) {
        err = "Could not parse the file \"" + filename + "\". Error: \"" + pe.message + "\" at line " + std::to_string(pe.line) +
              ", column " + std::to_string(pe.column) + ".";
        return false;
    }
    return true;
}

std::string write_scene(const Scene &sc) {
    std::ostringstream o;
    o << "image {\n    width = " << sc.image_settings.width << "\n    height = " << sc.image_settings.height
      << "\n    sub_pixel_sampling = true\n}\n\n";
    if (sc.sky.kind == PT_SKY_HDRI) o << "sky {\n    hdri {\n        path = " << sc.sky.path << "\n    }\n}\n\n";
    else o << "sky {\n    constant {\n        radiance = " << fmt_v3(sc.sky.radiance) << "\n    }\n}\n\n";
    const PtCamera &c = sc.camera.c;
    o << "camera {\n    position = " << fmt_v3(c.position) << "\n    target = " << fmt_v3(sc.camera.target) << "\n    up = " << fmt_v3(sc.camera.up)
      << "\n    width = " << fmt_f32(c.width) << "\n    height = " << fmt_f32(c.height) << "\n    z_near = " << fmt_f32(c.z_near)
      << "\n    tone_mapping = ";
    if (c.tone_mapping == PT_TONE_CLAMP) o << "clamp";
    else if (c.tone_mapping == PT_TONE_REINHARD) o << "reinhard";
    else o << "exposure { value = " << fmt_f32(c.exposure) << " }";
    o << "\n    iso = " << fmt_f32(c.iso) << "\n}\n\n";
    auto plane = [&](const PtPlane &p) {
        o << "plane {\n    origin = " << fmt_v3(p.origin) << "\n    u = " << fmt_v3(p.u) << "\n    v = " << fmt_v3(p.v)
          << "\n    material = " << fmt_material(p.material) << "\n}\n\n";
    };
    auto sphere = [&](const PtSphere &s) {
        o << "sphere {\n    origin = " << fmt_v3(s.origin) << "\n    radius = " << fmt_f32(s.radius) << "\n    material = "
          << fmt_material(s.material) << "\n}\n\n";
    };
    for (auto &p : sc.planes) plane(p);
    for (auto &p : sc.emissive_planes) plane(p);
    for (auto &s : sc.spheres) sphere(s);
    for (auto &s : sc.emissive_spheres) sphere(s);
    return o.str();
}

}  // namespace pthost

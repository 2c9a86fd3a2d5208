"""ctypes binding of the CPU oracle (oracle/libptref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never
by the pathtracer_amd package.  It consumes the same scene description objects (pathtracer_amd.scene) but
derives everything (plane normals, camera plane) with the oracle's own code.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libptref.so")

RNG_XORSHIFT, RNG_COUNTER = 0, 1
MATH_LIBM, MATH_DET = 0, 1
FN_SIN, FN_COS, FN_POW, FN_EXP, FN_ACOS, FN_ATAN2 = range(6)
MISS = 0xFFFFFFFF


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def t(self):
        return (self.x, self.y, self.z)


class Camera(C.Structure):
    _fields_ = [("plane_origin", Vec3), ("plane_u", Vec3), ("plane_v", Vec3), ("position", Vec3),
                ("width", C.c_float), ("height", C.c_float), ("z_near", C.c_float), ("tone_mapping", C.c_int32),
                ("exposure", C.c_float), ("iso", C.c_float)]


class Sky(C.Structure):
    _fields_ = [("kind", C.c_int32), ("radiance", Vec3), ("pixels", C.c_void_p), ("width", C.c_uint64), ("height", C.c_uint64)]


class Scene(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("sub_pixel_sampling", C.c_int32), ("camera", Camera),
                ("sky", Sky),
                ("spheres", C.c_void_p), ("n_spheres", C.c_uint64),
                ("planes", C.c_void_p), ("n_planes", C.c_uint64),
                ("emissive_spheres", C.c_void_p), ("n_emissive_spheres", C.c_uint64),
                ("emissive_planes", C.c_void_p), ("n_emissive_planes", C.c_uint64)]


class Options(C.Structure):
    _fields_ = [("rng_mode", C.c_int32), ("math_mode", C.c_int32), ("seed", C.c_uint32), ("max_depth", C.c_int32),
                ("sub_pixel_override", C.c_int32), ("camera_model", C.c_int32), ("focus_distance", C.c_float),
                ("lens_aperture", C.c_float), ("hoist_plane_axes", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("prim_tests", C.c_uint64)]


class Backbuffer(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("num_samples", C.c_uint64), ("pixels32", C.c_void_p),
                ("pixels8", C.c_void_p)]


class Hit(C.Structure):
    _fields_ = [("t", C.c_float), ("position", Vec3), ("normal", Vec3), ("transition", C.c_int32), ("prim_id", C.c_uint32)]


_lib = None


def build() -> None:
    subprocess.check_call(["make", "-C", _HERE, "libptref.so"], stdout=subprocess.DEVNULL)


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(LIB_PATH)
    L.ptref_xorshift32_next.argtypes = [C.POINTER(C.c_uint32)]
    L.ptref_xorshift32_next.restype = C.c_uint32
    L.ptref_random32_from_u32.argtypes = [C.c_uint32]
    L.ptref_random32_from_u32.restype = C.c_float
    L.ptref_counter_u32.argtypes = [C.c_uint32] * 5
    L.ptref_counter_u32.restype = C.c_uint32
    L.ptref_math_eval.argtypes = [C.c_int32, C.c_int32, C.c_float, C.c_float]
    L.ptref_math_eval.restype = C.c_float
    L.ptref_math_eval_array.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    L.ptref_math_eval_array.restype = None
    L.ptref_camera_new.argtypes = [Vec3, Vec3, Vec3, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_float, C.c_float,
                                   C.POINTER(Camera)]
    L.ptref_camera_new.restype = None
    L.ptref_plane_normal.argtypes = [Vec3, Vec3, C.POINTER(Vec3)]
    L.ptref_plane_normal.restype = None
    L.ptref_plane_normals.argtypes = [C.c_void_p, C.c_uint64]
    L.ptref_plane_normals.restype = None
    L.ptref_find_scene_hit.argtypes = [C.POINTER(Scene), Vec3, Vec3, C.POINTER(Hit)]
    L.ptref_find_scene_hit.restype = C.c_int32
    L.ptref_pinhole_ray.argtypes = [C.POINTER(Scene), C.POINTER(Options), C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_int32,
                                    C.POINTER(Vec3), C.POINTER(Vec3)]
    L.ptref_pinhole_ray.restype = None
    L.ptref_thin_lens_ray.argtypes = [C.POINTER(Scene), C.POINTER(Options), C.c_float, C.c_float, C.POINTER(Vec3), C.POINTER(Vec3)]
    L.ptref_thin_lens_ray.restype = None
    L.ptref_trace_radiance.argtypes = [C.POINTER(Scene), C.POINTER(Options), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, Vec3,
                                       Vec3, C.c_int32, C.POINTER(Stats)]
    L.ptref_trace_radiance.restype = Vec3
    L.ptref_tone_map_gamma.argtypes = [C.POINTER(Scene), C.POINTER(Options), Vec3]
    L.ptref_tone_map_gamma.restype = Vec3
    L.ptref_pixel32_from_unit.argtypes = [Vec3, C.POINTER(C.c_uint32 * 3)]
    L.ptref_pixel32_from_unit.restype = None
    L.ptref_render_pass.argtypes = [C.POINTER(Scene), C.POINTER(Options), C.POINTER(Backbuffer), C.POINTER(C.c_uint32), C.c_uint32,
                                    C.POINTER(Stats)]
    L.ptref_render_pass.restype = None
    L.ptref_render_pass_pool.argtypes = [C.POINTER(Scene), C.POINTER(Options), C.POINTER(Backbuffer), C.c_int32, C.c_uint32,
                                         C.POINTER(Stats)]
    L.ptref_render_pass_pool.restype = None
    L.ptref_primary_ids.argtypes = [C.POINTER(Scene), C.POINTER(Options), C.c_uint32, C.c_void_p]
    L.ptref_primary_ids.restype = None
    L.ptref_assign_pixels8.argtypes = [C.POINTER(Scene), C.c_void_p, C.c_uint64, C.c_void_p]
    L.ptref_assign_pixels8.restype = None
    _lib = L
    return L


def _v(t) -> Vec3:
    return Vec3(float(t[0]), float(t[1]), float(t[2]))


def camera_new(cam) -> Camera:
    out = Camera()
    lib().ptref_camera_new(_v(cam.position), _v(cam.target), _v(cam.up), cam.width, cam.height, cam.z_near, cam.tone_mapping,
                           cam.exposure, cam.iso, C.byref(out))
    return out


def options(rng_mode=RNG_COUNTER, math_mode=MATH_DET, seed=314159265, max_depth=4, sub_pixel_override=-1, camera_model=0,
            focus_distance=10.0, lens_aperture=4.0, hoist_plane_axes=0) -> Options:
    return Options(rng_mode, math_mode, seed, max_depth, sub_pixel_override, camera_model, focus_distance, lens_aperture,
                   hoist_plane_axes)


class OracleScene:
    """pr_scene built from a pathtracer_amd.scene.Scene; keeps the numpy buffers alive."""

    def __init__(self, scene):
        L = lib()
        self.arrays = [a.copy() for a in scene.arrays()]
        for k in (1, 3):
            if len(self.arrays[k]):
                L.ptref_plane_normals(self.arrays[k].ctypes.data, len(self.arrays[k]))
        d = Scene()
        d.width, d.height = scene.image_settings.width, scene.image_settings.height
        d.sub_pixel_sampling = 1 if scene.image_settings.sub_pixel_sampling else 0
        d.camera = camera_new(scene.camera)
        d.sky.kind = scene.sky.kind
        d.sky.radiance = _v(scene.sky.radiance)
        self.sky_pixels = None
        if scene.sky.kind == 1 and scene.sky.pixels is not None:
            self.sky_pixels = np.ascontiguousarray(scene.sky.pixels, dtype=np.float32)
            d.sky.pixels = self.sky_pixels.ctypes.data
            d.sky.height, d.sky.width = self.sky_pixels.shape[0], self.sky_pixels.shape[1]
        sp, pl, es, ep = self.arrays
        d.spheres, d.n_spheres = (sp.ctypes.data if len(sp) else None), len(sp)
        d.planes, d.n_planes = (pl.ctypes.data if len(pl) else None), len(pl)
        d.emissive_spheres, d.n_emissive_spheres = (es.ctypes.data if len(es) else None), len(es)
        d.emissive_planes, d.n_emissive_planes = (ep.ctypes.data if len(ep) else None), len(ep)
        self.desc = d
        self.width, self.height = d.width, d.height

    # -- single queries ------------------------------------------------------------------------------
    def find_scene_hit(self, origin, direction):
        h = Hit()
        found = lib().ptref_find_scene_hit(C.byref(self.desc), _v(origin), _v(direction), C.byref(h))
        return bool(found), h

    def pinhole_ray(self, x, y, jitter=None):
        o, d = Vec3(), Vec3()
        opt = options()
        ju, jv = (jitter if jitter is not None else (0.0, 0.0))
        lib().ptref_pinhole_ray(C.byref(self.desc), C.byref(opt), x, y, ju, jv, 1 if jitter is not None else 0, C.byref(o), C.byref(d))
        return o.t(), d.t()

    def thin_lens_ray(self, opt: Options, origin, direction, r1, r2):
        o, d = _v(origin), _v(direction)
        lib().ptref_thin_lens_ray(C.byref(self.desc), C.byref(opt), r1, r2, C.byref(o), C.byref(d))
        return o.t(), d.t()

    def trace_radiance(self, opt: Options, origin, direction, depth=None, pixel=0, sample=0, xorshift_state=314159265):
        st = C.c_uint32(xorshift_state)
        stats = Stats()
        r = lib().ptref_trace_radiance(C.byref(self.desc), C.byref(opt), C.byref(st), pixel, sample, _v(origin), _v(direction),
                                       opt.max_depth if depth is None else depth, C.byref(stats))
        return r.t(), stats.rays, st.value

    def primary_ids(self, opt: Options, sample_index=0) -> np.ndarray:
        out = np.empty((self.height, self.width), dtype=np.uint32)
        lib().ptref_primary_ids(C.byref(self.desc), C.byref(opt), sample_index, out.ctypes.data)
        return out

    def render_logging_rays(self, opt: Options, n_samples: int, capacity: int = 1 << 22):
        """render() on one thread with the ray log on: (pixels32, pixels8, rays, log) where log[k] = (origin.xyz,
        direction.xyz, bits of the primitive id found, t) of the k-th find_scene_hit call."""
        L = lib()
        L.ptref_ray_log.argtypes = [C.c_void_p, C.c_uint64]
        L.ptref_ray_log.restype = None
        L.ptref_ray_log_count.restype = C.c_uint64
        log = np.zeros((capacity, 8), dtype=np.float32)
        L.ptref_ray_log(log.ctypes.data, capacity)
        try:
            p32, p8, rays = self.render(opt, n_samples)
            n = int(L.ptref_ray_log_count())
        finally:
            L.ptref_ray_log(None, 0)
        if n > capacity:
            raise ValueError(f"ray log too small: {n} calls")
        return p32, p8, rays, log[:n]

    # -- accumulation ----------------------------------------------------------------------------------
    def render(self, opt: Options, n_samples: int, sample_indices=None, threads: int = 0):
        """n_samples passes over the image.  Returns (pixels32, pixels8, rays).  sample_indices defaults to
        0..n-1 (they key the counter RNG).  threads > 0 uses the worker pool."""
        W, H = self.width, self.height
        p32 = np.zeros((H, W, 3), dtype=np.uint32)
        p8 = np.zeros((H, W, 3), dtype=np.uint8)
        bb = Backbuffer(W, H, 1, p32.ctypes.data, p8.ctypes.data)
        xs = C.c_uint32(opt.seed if opt.seed else 1)
        stats = Stats()
        idx = list(range(n_samples)) if sample_indices is None else list(sample_indices)
        for s in idx:
            if threads > 0:
                lib().ptref_render_pass_pool(C.byref(self.desc), C.byref(opt), C.byref(bb), threads, s, C.byref(stats))
            else:
                lib().ptref_render_pass(C.byref(self.desc), C.byref(opt), C.byref(bb), C.byref(xs), s, C.byref(stats))
            bb.num_samples += 1  # window.rs:211
        return p32, p8, int(stats.rays)

    def assign_pixels8(self, pixels32: np.ndarray, num_samples: int) -> np.ndarray:
        p32 = np.ascontiguousarray(pixels32, dtype=np.uint32)
        out = np.empty(p32.shape, dtype=np.uint8)
        lib().ptref_assign_pixels8(C.byref(self.desc), p32.ctypes.data, num_samples, out.ctypes.data)
        return out


def math_eval(mode: int, fn: int, a, b=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    out = np.empty_like(a)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(np.broadcast_to(np.asarray(b, dtype=np.float32), a.shape))
        bp = b.ctypes.data
    lib().ptref_math_eval_array(mode, fn, a.ctypes.data, bp, out.ctypes.data, a.size)
    return out

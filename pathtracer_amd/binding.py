"""ctypes binding of libptgpu's C ABI (include/ptgpu.h) and a small object wrapper.

The wrapper mirrors how the reference's window loop drives the path (reference src/window.rs:104-213):
create once, upload a scene whenever it changes, render n frames' worth of samples, read the backbuffer.
There is no CPU path behind it: every call ends in the HIP library and fails loudly without it.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import scene as S

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PTGPU_LIB", os.path.join(_HERE, "libptgpu.so"))  # override: A/B builds only

PT_KERNEL_COUNT = 8
KERNEL_NAMES = ["raygen", "trace", "shade", "trace_shade", "terminal", "gather", "resolve", "unused"]
STATUS = {0: "PT_OK", -1: "PT_ERR_INVALID_ARGUMENT", -2: "PT_ERR_NO_DEVICE", -3: "PT_ERR_HIP", -4: "PT_ERR_NO_SCENE",
          -5: "PT_ERR_QUEUE_OVERFLOW", -6: "PT_ERR_OUT_OF_MEMORY", -7: "PT_ERR_INTERNAL"}
ACCEL_AUTO, ACCEL_BRUTE_FORCE, ACCEL_BVH = 0, 1, 2


class PtVec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class PtCamera(C.Structure):
    _fields_ = [("plane_origin", PtVec3), ("plane_u", PtVec3), ("plane_v", PtVec3), ("position", PtVec3),
                ("width", C.c_float), ("height", C.c_float), ("z_near", C.c_float), ("tone_mapping", C.c_int32),
                ("exposure", C.c_float), ("iso", C.c_float)]


class PtSky(C.Structure):
    _fields_ = [("kind", C.c_int32), ("radiance", PtVec3), ("pixels", C.c_void_p), ("width", C.c_uint64), ("height", C.c_uint64)]


class PtSceneDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("sub_pixel_sampling", C.c_int32), ("camera", PtCamera),
                ("sky", PtSky),
                ("spheres", C.c_void_p), ("n_spheres", C.c_uint64),
                ("planes", C.c_void_p), ("n_planes", C.c_uint64),
                ("emissive_spheres", C.c_void_p), ("n_emissive_spheres", C.c_uint64),
                ("emissive_planes", C.c_void_p), ("n_emissive_planes", C.c_uint64)]


class PtConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("seed", C.c_uint32), ("max_depth", C.c_int32), ("sub_pixel_override", C.c_int32),
                ("shard_index", C.c_uint32), ("shard_count", C.c_uint32), ("samples_per_batch", C.c_uint32),
                ("branch_capacity", C.c_uint32), ("accel", C.c_int32), ("fuse_trace_shade", C.c_int32),
                ("profile_events", C.c_int32), ("concurrent_passes", C.c_uint32), ("camera_model", C.c_int32),
                ("focus_distance", C.c_float), ("lens_aperture", C.c_float)]


class PtStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("records", C.c_uint64), ("terminal_records", C.c_uint64), ("bvh_nodes", C.c_uint64),
                ("bvh_leaf_prims", C.c_uint64), ("bvh_steps", C.c_uint64), ("kernel_launches", C.c_uint64 * PT_KERNEL_COUNT),
                ("kernel_ms", C.c_double * PT_KERNEL_COUNT), ("device_bytes", C.c_uint64), ("queue_growth", C.c_uint64), ("debug_cycles", C.c_uint64 * 8),
                ("bvh_depth", C.c_uint64), ("bvh_binary_depth", C.c_uint64), ("bvh_forced_median_splits", C.c_uint64),
                ("bvh_level_fold", C.c_uint64), ("queue_bytes", C.c_uint64), ("level_records", C.c_uint64 * 25),
                ("samples_per_pass", C.c_uint64), ("passes_in_flight", C.c_uint64)]


class PtGpuError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{STATUS.get(code, code)}: {message}")
        self.code = code


_lib = None


def load_library() -> C.CDLL:
    """dlopen libptgpu.so (built in-tree by __graft_entry__.build()).  Raises if it is missing: there is no
    fallback implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} not built; run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(LIB_PATH)
    H = C.c_void_p
    lib.ptgpu_default_config.argtypes = [C.POINTER(PtConfig)]
    lib.ptgpu_default_config.restype = None
    lib.ptgpu_create.argtypes = [C.POINTER(PtConfig), C.POINTER(H)]
    lib.ptgpu_destroy.argtypes = [H]
    lib.ptgpu_destroy.restype = None
    lib.ptgpu_upload_scene.argtypes = [H, C.POINTER(PtSceneDesc)]
    lib.ptgpu_clear.argtypes = [H]
    lib.ptgpu_render.argtypes = [H, C.c_uint32]
    lib.ptgpu_sync.argtypes = [H]
    lib.ptgpu_render_noalloc.argtypes = [H, C.c_uint32]
    lib.ptgpu_frame.argtypes = [H, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.ptgpu_render_tiles.argtypes = [H, C.c_uint64, C.c_uint32]
    lib.ptgpu_tile_shard.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.ptgpu_tile_shard.restype = C.c_uint32
    lib.ptgpu_read_pixels32.argtypes = [H, C.c_void_p, C.POINTER(C.c_uint64)]
    lib.ptgpu_read_pixels8.argtypes = [H, C.c_void_p]
    lib.ptgpu_present_pixels8.argtypes = [H, C.POINTER(C.c_void_p)]
    lib.ptgpu_render_primary_ids.argtypes = [H, C.c_uint32, C.c_void_p]
    lib.ptgpu_device_pixels32.argtypes = [H, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.ptgpu_set_num_samples.argtypes = [H, C.c_uint64]
    lib.ptgpu_stream.argtypes = [H, C.POINTER(C.c_void_p)]
    lib.ptgpu_export_pixels32.argtypes = [H, C.c_void_p]
    lib.ptgpu_import_pixels32.argtypes = [H, C.c_void_p, C.c_uint64]
    lib.ptgpu_get_stats.argtypes = [H, C.POINTER(PtStats)]
    G = C.c_void_p
    lib.ptgpu_group_create.argtypes = [C.POINTER(PtConfig), C.POINTER(C.c_int32), C.c_uint32, C.POINTER(G)]
    lib.ptgpu_group_destroy.argtypes = [G]
    lib.ptgpu_group_destroy.restype = None
    lib.ptgpu_group_upload_scene.argtypes = [G, C.POINTER(PtSceneDesc)]
    lib.ptgpu_group_clear.argtypes = [G]
    lib.ptgpu_group_render.argtypes = [G, C.c_uint32]
    lib.ptgpu_group_frame.argtypes = [G, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.ptgpu_group_sync.argtypes = [G]
    lib.ptgpu_group_read_pixels32.argtypes = [G, C.c_void_p, C.POINTER(C.c_uint64)]
    lib.ptgpu_group_read_pixels8.argtypes = [G, C.c_void_p]
    lib.ptgpu_group_present_pixels8.argtypes = [G, C.POINTER(C.c_void_p)]
    lib.ptgpu_group_get_stats.argtypes = [G, C.POINTER(PtStats)]
    lib.ptgpu_group_exchange_stats.argtypes = [G, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    lib.ptgpu_group_shard_samples.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
    lib.ptgpu_group_shard_samples.restype = C.c_uint64
    lib.ptgpu_group_size.argtypes = [G]
    lib.ptgpu_group_size.restype = C.c_uint32
    lib.ptgpu_group_member.argtypes = [G, C.c_uint32]
    lib.ptgpu_group_member.restype = C.c_void_p
    lib.ptgpu_group_last_error.argtypes = [G]
    lib.ptgpu_group_last_error.restype = C.c_char_p
    for name in ("create", "upload_scene", "clear", "render", "frame", "sync", "read_pixels32", "read_pixels8", "present_pixels8", "get_stats",
                 "exchange_stats"):
        getattr(lib, "ptgpu_group_" + name).restype = C.c_int
    lib.ptgpu_bvh_build_stats.argtypes = [C.POINTER(PtSceneDesc), C.POINTER(PtStats)]
    lib.ptgpu_bvh_build_stats.restype = C.c_int
    lib.ptgpu_bvh_host_trace.argtypes = [C.POINTER(PtSceneDesc), C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    lib.ptgpu_bvh_host_trace.restype = C.c_int
    lib.ptgpu_bvh_host_trace_flags.argtypes = [C.POINTER(PtSceneDesc), C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64),
                                               C.c_uint32]
    lib.ptgpu_bvh_host_trace_flags.restype = C.c_int
    lib.ptgpu_last_error.argtypes = [H]
    lib.ptgpu_last_error.restype = C.c_char_p
    lib.ptgpu_camera_new.argtypes = [PtVec3, PtVec3, PtVec3, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_float, C.c_float,
                                     C.POINTER(PtCamera)]
    lib.ptgpu_camera_new.restype = None
    lib.ptgpu_plane_normal.argtypes = [PtVec3, PtVec3, C.POINTER(PtVec3)]
    lib.ptgpu_plane_normal.restype = None
    lib.ptgpu_plane_normals.argtypes = [C.c_void_p, C.c_uint64]
    lib.ptgpu_plane_normals.restype = None
    lib.ptgpu_eval_math.argtypes = [H, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    lib.ptgpu_selftest_gamma.argtypes = [H, C.POINTER(C.c_uint64)]
    for name in ("ptgpu_create", "ptgpu_upload_scene", "ptgpu_clear", "ptgpu_render", "ptgpu_sync", "ptgpu_read_pixels32",
                 "ptgpu_read_pixels8", "ptgpu_render_primary_ids", "ptgpu_device_pixels32", "ptgpu_set_num_samples",
                 "ptgpu_stream", "ptgpu_get_stats", "ptgpu_eval_math", "ptgpu_export_pixels32", "ptgpu_import_pixels32", "ptgpu_selftest_gamma",
                 "ptgpu_present_pixels8", "ptgpu_render_noalloc", "ptgpu_frame", "ptgpu_render_tiles"):
        getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def _v(t) -> PtVec3:
    return PtVec3(float(t[0]), float(t[1]), float(t[2]))


def default_config(**overrides) -> PtConfig:
    cfg = PtConfig()
    load_library().ptgpu_default_config(C.byref(cfg))
    for k, val in overrides.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, val)
    return cfg


def camera_new(cam: S.Camera) -> PtCamera:
    """Camera::new (reference src/tracer.rs:69-96) as computed by the library's host code."""
    out = PtCamera()
    load_library().ptgpu_camera_new(_v(cam.position), _v(cam.target), _v(cam.up), cam.width, cam.height, cam.z_near,
                                    cam.tone_mapping, cam.exposure, cam.iso, C.byref(out))
    return out


class SceneDesc:
    """A PtSceneDesc plus the numpy buffers it points into (kept alive with it)."""

    def __init__(self, scene: S.Scene):
        lib = load_library()
        self.arrays = list(scene.arrays())
        for k in (1, 3):  # Plane::new's normal (reference src/scene.rs:106)
            if len(self.arrays[k]):
                self.arrays[k] = self.arrays[k].copy()
                lib.ptgpu_plane_normals(self.arrays[k].ctypes.data, len(self.arrays[k]))
        d = PtSceneDesc()
        d.width, d.height = scene.image_settings.width, scene.image_settings.height
        d.sub_pixel_sampling = 1 if scene.image_settings.sub_pixel_sampling else 0
        d.camera = camera_new(scene.camera)
        d.sky.kind = scene.sky.kind
        d.sky.radiance = _v(scene.sky.radiance)
        self.sky_pixels = None
        if scene.sky.kind == S.SKY_HDRI and scene.sky.pixels is not None:
            self.sky_pixels = np.ascontiguousarray(scene.sky.pixels, dtype=np.float32)
            d.sky.pixels = self.sky_pixels.ctypes.data
            d.sky.height, d.sky.width = self.sky_pixels.shape[0], self.sky_pixels.shape[1]
        sp, pl, es, ep = self.arrays
        d.spheres, d.n_spheres = (sp.ctypes.data if len(sp) else None), len(sp)
        d.planes, d.n_planes = (pl.ctypes.data if len(pl) else None), len(pl)
        d.emissive_spheres, d.n_emissive_spheres = (es.ctypes.data if len(es) else None), len(es)
        d.emissive_planes, d.n_emissive_planes = (ep.ctypes.data if len(ep) else None), len(ep)
        self.desc = d


def bvh_build_stats(scene: S.Scene) -> dict:
    """Size and depth of the BVH the library would build for `scene` (host only: works without a GPU)."""
    sd = SceneDesc(scene)
    st = PtStats()
    rc = load_library().ptgpu_bvh_build_stats(C.byref(sd.desc), C.byref(st))
    if rc != 0:
        raise PtGpuError(rc, "ptgpu_bvh_build_stats failed")
    return {k: int(getattr(st, k)) for k in ("bvh_nodes", "bvh_leaf_prims", "bvh_depth", "bvh_binary_depth",
                                             "bvh_forced_median_splits", "bvh_level_fold")}


HOST_TRACE_UNGATED = 1


def bvh_host_trace(scene: S.Scene, rays: np.ndarray, flags: int = 0):
    """Nearest primitive and distance for each ray (rows of origin.xyz, direction.xyz), found by the kernels' traversal
    functions compiled for the CPU (host only: works without a GPU), and the checker's counts: (prims, ts, faults) with
    faults = dict(bad_units, bad_kinds, bad_slots, steps).  flags = HOST_TRACE_UNGATED walks every ray through the tree,
    also those the traversal normally leaves to the linear scan."""
    sd = SceneDesc(scene)
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
    n = rays.shape[0]
    prims = np.empty(n, dtype=np.uint32)
    ts = np.empty(n, dtype=np.float32)
    faults = (C.c_uint64 * 4)()
    rc = load_library().ptgpu_bvh_host_trace_flags(C.byref(sd.desc), n, rays.ctypes.data, prims.ctypes.data, ts.ctypes.data, faults, flags)
    if rc != 0:
        raise PtGpuError(rc, "ptgpu_bvh_host_trace failed")
    return prims, ts, dict(bad_units=int(faults[0]), bad_kinds=int(faults[1]), bad_slots=int(faults[2]), steps=int(faults[3]))


def tile_shard(tile_x: int, tile_y: int, tiles_x: int, n: int) -> int:
    """Which of n tile shards owns 32x32 tile (tile_x, tile_y) of an image tiles_x tiles wide (pure arithmetic, no GPU)."""
    return int(load_library().ptgpu_tile_shard(tile_x, tile_y, tiles_x, n))


def _stats_dict(st: PtStats) -> dict:
    return {
        "samples": st.samples, "rays": st.rays, "shadow_rays": st.shadow_rays, "records": st.records, "terminal_records": st.terminal_records, "bvh_nodes": st.bvh_nodes,
        "bvh_leaf_prims": st.bvh_leaf_prims, "bvh_steps": st.bvh_steps, "device_bytes": st.device_bytes, "queue_growth": st.queue_growth,
        "kernel_launches": {KERNEL_NAMES[k]: st.kernel_launches[k] for k in range(PT_KERNEL_COUNT)},
        "kernel_ms": {KERNEL_NAMES[k]: st.kernel_ms[k] for k in range(PT_KERNEL_COUNT)},
        "debug_cycles": [int(st.debug_cycles[k]) for k in range(8)],
        "bvh_depth": st.bvh_depth, "bvh_binary_depth": st.bvh_binary_depth,
        "bvh_forced_median_splits": st.bvh_forced_median_splits, "bvh_level_fold": st.bvh_level_fold,
        "queue_bytes": st.queue_bytes, "level_records": [int(st.level_records[k]) for k in range(25)],
        "samples_per_pass": int(st.samples_per_pass), "passes_in_flight": int(st.passes_in_flight),
    }


class PtGroup:
    """Several GPUs of one node behind one handle, driven from this one process (ptgpu_group_*): what a single-process host
    like the reference's window loop (reference src/window.rs:104-213) uses in place of PtGpu when it has more than one
    GPU.  `devices` = list of HIP ordinals, or None with n_devices = 0 for all visible ones."""

    def __init__(self, devices=None, n_devices: int = 0, cfg: Optional[PtConfig] = None, **overrides):
        self.lib = load_library()
        self.cfg = cfg if cfg is not None else default_config(**overrides)
        self.g = C.c_void_p()
        arr = None
        if devices is not None:
            n_devices = len(devices)
            arr = (C.c_int32 * n_devices)(*devices)
        rc = self.lib.ptgpu_group_create(C.byref(self.cfg), arr, n_devices, C.byref(self.g))
        if rc != 0:
            self.g = C.c_void_p()
            raise PtGpuError(rc, "ptgpu_group_create failed (gfx950 GPUs are required)")
        self.width = self.height = 0

    def _chk(self, rc: int) -> None:
        if rc != 0:
            raise PtGpuError(rc, (self.lib.ptgpu_group_last_error(self.g) or b"").decode())

    def close(self) -> None:
        if self.g:
            self.lib.ptgpu_group_destroy(self.g)
            self.g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def size(self) -> int:
        return int(self.lib.ptgpu_group_size(self.g))

    def upload_scene(self, scene: S.Scene) -> None:
        sd = SceneDesc(scene)
        self._chk(self.lib.ptgpu_group_upload_scene(self.g, C.byref(sd.desc)))
        self.width, self.height = scene.image_settings.width, scene.image_settings.height

    def clear(self) -> None:
        self._chk(self.lib.ptgpu_group_clear(self.g))

    def render(self, n_samples: int) -> None:
        self._chk(self.lib.ptgpu_group_render(self.g, int(n_samples)))

    def frame(self, n_samples: int = 1):
        """One iteration of the window loop on all GPUs (ptgpu_group_frame): enqueues n_samples, returns (display bytes of the
        newest frame that has arrived - a view of page-locked memory valid until the next call -, samples per pixel they show)."""
        p, n = C.c_void_p(), C.c_uint64()
        self._chk(self.lib.ptgpu_group_frame(self.g, int(n_samples), C.byref(p), C.byref(n)))
        buf = (C.c_uint8 * (self.height * self.width * 3)).from_address(p.value)
        return np.frombuffer(buf, dtype=np.uint8).reshape(self.height, self.width, 3), int(n.value)

    def sync(self) -> None:
        self._chk(self.lib.ptgpu_group_sync(self.g))

    def read_pixels32(self):
        out = np.empty((self.height, self.width, 3), dtype=np.uint32)
        n = C.c_uint64()
        self._chk(self.lib.ptgpu_group_read_pixels32(self.g, out.ctypes.data, C.byref(n)))
        return out, int(n.value)

    def read_pixels8(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 3), dtype=np.uint8)
        self._chk(self.lib.ptgpu_group_read_pixels8(self.g, out.ctypes.data))
        return out

    def present_pixels8(self) -> np.ndarray:
        p = C.c_void_p()
        self._chk(self.lib.ptgpu_group_present_pixels8(self.g, C.byref(p)))
        buf = (C.c_uint8 * (self.height * self.width * 3)).from_address(p.value)
        return np.frombuffer(buf, dtype=np.uint8).reshape(self.height, self.width, 3)

    def stats(self) -> dict:
        st = PtStats()
        self._chk(self.lib.ptgpu_group_get_stats(self.g, C.byref(st)))
        return _stats_dict(st)

    def exchange_stats(self) -> dict:
        b, ms = C.c_uint64(), C.c_double()
        self._chk(self.lib.ptgpu_group_exchange_stats(self.g, C.byref(b), C.byref(ms)))
        return {"bytes": int(b.value), "ms": float(ms.value)}


class PtGpu:
    """One handle = one GPU's share of the render (reference: the WorkerPool + Backbuffer pair of
    src/window.rs:104-116)."""

    def __init__(self, cfg: Optional[PtConfig] = None, **overrides):
        self.lib = load_library()
        self.cfg = cfg if cfg is not None else default_config(**overrides)
        self.h = C.c_void_p()
        rc = self.lib.ptgpu_create(C.byref(self.cfg), C.byref(self.h))
        if rc != 0:
            self.h = C.c_void_p()
            raise PtGpuError(rc, "ptgpu_create failed (a gfx950 GPU is required)")
        self.width = self.height = 0

    def _chk(self, rc: int) -> None:
        if rc != 0:
            raise PtGpuError(rc, (self.lib.ptgpu_last_error(self.h) or b"").decode())

    def close(self) -> None:
        if self.h:
            self.lib.ptgpu_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def upload_scene(self, scene: S.Scene) -> None:
        sd = SceneDesc(scene)
        self._chk(self.lib.ptgpu_upload_scene(self.h, C.byref(sd.desc)))
        self.width, self.height = scene.image_settings.width, scene.image_settings.height

    def clear(self) -> None:
        self._chk(self.lib.ptgpu_clear(self.h))

    def render(self, n_samples: int) -> None:
        self._chk(self.lib.ptgpu_render(self.h, int(n_samples)))

    def render_tiles(self, first_sample: int, n_samples: int) -> None:
        """Global sample indices first_sample .. first_sample + n_samples - 1 for the pixels of this handle's 32x32 tiles
        (tile shards: shard_index of shard_count by ptgpu_tile_shard)."""
        self._chk(self.lib.ptgpu_render_tiles(self.h, int(first_sample), int(n_samples)))

    def frame(self, n_samples: int = 1):
        """One iteration of the window loop (ptgpu_frame): enqueues n_samples, returns (display bytes of the newest frame that
        has arrived - a view of page-locked memory valid until the next call -, samples per pixel they show)."""
        p, n = C.c_void_p(), C.c_uint64()
        self._chk(self.lib.ptgpu_frame(self.h, int(n_samples), C.byref(p), C.byref(n)))
        buf = (C.c_uint8 * (self.height * self.width * 3)).from_address(p.value)
        return np.frombuffer(buf, dtype=np.uint8).reshape(self.height, self.width, 3), int(n.value)

    def render_noalloc(self, n_samples: int) -> None:
        """ptgpu_render with its ray queues in the library's static device array: no allocation during the call."""
        self._chk(self.lib.ptgpu_render_noalloc(self.h, int(n_samples)))

    def sync(self) -> None:
        self._chk(self.lib.ptgpu_sync(self.h))

    def read_pixels32(self):
        out = np.empty((self.height, self.width, 3), dtype=np.uint32)
        n = C.c_uint64()
        self._chk(self.lib.ptgpu_read_pixels32(self.h, out.ctypes.data, C.byref(n)))
        return out, int(n.value)

    def read_pixels8(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 3), dtype=np.uint8)
        self._chk(self.lib.ptgpu_read_pixels8(self.h, out.ctypes.data))
        return out

    def present_pixels8(self) -> np.ndarray:
        """The display bytes as a view of the library's own page-locked buffer (valid until the next present / upload /
        close): what a window loop hands to glDrawPixels every frame."""
        p = C.c_void_p()
        self._chk(self.lib.ptgpu_present_pixels8(self.h, C.byref(p)))
        buf = (C.c_uint8 * (self.height * self.width * 3)).from_address(p.value)
        return np.frombuffer(buf, dtype=np.uint8).reshape(self.height, self.width, 3)

    def primary_ids(self, sample_index: int = 0) -> np.ndarray:
        out = np.empty((self.height, self.width), dtype=np.uint32)
        self._chk(self.lib.ptgpu_render_primary_ids(self.h, int(sample_index), out.ctypes.data))
        return out

    def device_pixels32(self):
        p, n = C.c_void_p(), C.c_uint64()
        self._chk(self.lib.ptgpu_device_pixels32(self.h, C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)

    def set_num_samples(self, n: int) -> None:
        self._chk(self.lib.ptgpu_set_num_samples(self.h, int(n)))

    def export_pixels32(self, dst_device_ptr: int) -> None:
        self._chk(self.lib.ptgpu_export_pixels32(self.h, C.c_void_p(dst_device_ptr)))

    def import_pixels32(self, src_device_ptr: int, num_samples: int) -> None:
        self._chk(self.lib.ptgpu_import_pixels32(self.h, C.c_void_p(src_device_ptr), int(num_samples)))

    def stream(self) -> int:
        p = C.c_void_p()
        self._chk(self.lib.ptgpu_stream(self.h, C.byref(p)))
        return int(p.value or 0)

    def stats(self) -> dict:
        st = PtStats()
        self._chk(self.lib.ptgpu_get_stats(self.h, C.byref(st)))
        return _stats_dict(st)

    def selftest_gamma(self) -> int:
        n = C.c_uint64()
        self._chk(self.lib.ptgpu_selftest_gamma(self.h, C.byref(n)))
        return int(n.value)

    def eval_math(self, fn: int, a: np.ndarray, b: Optional[np.ndarray] = None) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.float32)
        out = np.empty_like(a)
        bp = None
        if b is not None:
            b = np.ascontiguousarray(b, dtype=np.float32)
            bp = b.ctypes.data
        self._chk(self.lib.ptgpu_eval_math(self.h, fn, a.ctypes.data, bp, out.ctypes.data, a.size))
        return out

"""ctypes window onto the C++ host layer (pathtracer_amd/host): `.scene` parser and writer, `.hdr` loader.

The parser is the C++ restatement of the reference's src/parser.rs; it is host-side code (no GPU needed).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import binding as B
from . import scene as S

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpthost.so")
PARSE_DEFAULT, PARSE_REFERENCE_COMMENT_TAIL = 0, 1

_lib = None


class ParseError(ValueError):
    def __init__(self, message, line, column):
        super().__init__(f"{message} (line {line}, column {column})")
        self.message, self.line, self.column = message, line, column


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        B.load_library()  # libpthost links against libptgpu
        L = C.CDLL(LIB_PATH)
        L.pthost_parse_scene.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_int32)]
        L.pthost_parse_scene.restype = C.c_void_p
        L.pthost_scene_free.argtypes = [C.c_void_p]
        L.pthost_scene_free.restype = None
        L.pthost_scene_error.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.pthost_scene_error.restype = C.c_char_p
        L.pthost_scene_desc.argtypes = [C.c_void_p, C.POINTER(B.PtSceneDesc)]
        L.pthost_scene_desc.restype = None
        L.pthost_scene_sky_path.argtypes = [C.c_void_p]
        L.pthost_scene_sky_path.restype = C.c_char_p
        L.pthost_scene_lookat.argtypes = [C.c_void_p, C.POINTER(B.PtVec3), C.POINTER(B.PtVec3)]
        L.pthost_scene_lookat.restype = None
        L.pthost_scene_write.argtypes = [C.c_void_p]
        L.pthost_scene_write.restype = C.c_char_p
        L.pthost_scene_load_sky.argtypes = [C.c_void_p, C.c_char_p]
        L.pthost_scene_load_sky.restype = C.c_int32
        L.pthost_load_hdr.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.pthost_load_hdr.restype = C.POINTER(C.c_float)
        L.pthost_free.argtypes = [C.c_void_p]
        L.pthost_free.restype = None
        _lib = L
    return _lib


def _arr(ptr, n, dt):
    if not ptr or n == 0:
        return np.zeros(0, dtype=dt)
    buf = (C.c_char * (n * dt.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dt).copy()


class ParsedScene:
    """Result of parse_scene: the flattened arrays (SPHERE_DT / PLANE_DT, normals already computed by Plane::new),
    camera, sky and image settings, plus the look-at arguments the camera block held."""

    def __init__(self, text: str, flags: int = PARSE_DEFAULT):
        L = lib()
        raw = text.encode("utf-8")
        ok = C.c_int32()
        self._h = L.pthost_parse_scene(raw, len(raw), flags, C.byref(ok))
        if not ok.value:
            line, col = C.c_uint64(), C.c_uint64()
            msg = L.pthost_scene_error(self._h, C.byref(line), C.byref(col)).decode()
            L.pthost_scene_free(self._h)
            self._h = None
            raise ParseError(msg, line.value, col.value)
        d = B.PtSceneDesc()
        L.pthost_scene_desc(self._h, C.byref(d))
        self.desc = d
        self.spheres = _arr(d.spheres, d.n_spheres, S.SPHERE_DT)
        self.planes = _arr(d.planes, d.n_planes, S.PLANE_DT)
        self.emissive_spheres = _arr(d.emissive_spheres, d.n_emissive_spheres, S.SPHERE_DT)
        self.emissive_planes = _arr(d.emissive_planes, d.n_emissive_planes, S.PLANE_DT)
        self.sky_path = L.pthost_scene_sky_path(self._h).decode()
        t, u = B.PtVec3(), B.PtVec3()
        L.pthost_scene_lookat(self._h, C.byref(t), C.byref(u))
        self.target, self.up = (t.x, t.y, t.z), (u.x, u.y, u.z)

    def write(self) -> str:
        return lib().pthost_scene_write(self._h).decode()

    def to_scene(self) -> S.Scene:
        """The equivalent pathtracer_amd.scene.Scene (look-at camera arguments, primitive arrays)."""
        d = self.desc
        cam = d.camera
        sc = S.Scene(
            image_settings=S.ImageSettings(d.width, d.height, bool(d.sub_pixel_sampling)),
            camera=S.Camera((cam.position.x, cam.position.y, cam.position.z), self.target, self.up, cam.width, cam.height, cam.z_near,
                            cam.tone_mapping, cam.exposure, cam.iso),
            sky=S.Sky.constant((d.sky.radiance.x, d.sky.radiance.y, d.sky.radiance.z)) if d.sky.kind == S.SKY_CONSTANT
            else S.Sky.hdri(self.sky_path, None),
        )
        sc.spheres, sc.planes = self.spheres.copy(), self.planes.copy()
        sc.emissive_spheres, sc.emissive_planes = self.emissive_spheres.copy(), self.emissive_planes.copy()
        for a in (sc.planes, sc.emissive_planes):
            if len(a):
                a["normal"] = 0.0  # recomputed by whichever library consumes the scene
        return sc

    def __del__(self):
        if getattr(self, "_h", None):
            lib().pthost_scene_free(self._h)
            self._h = None


def parse_scene(text: str, flags: int = PARSE_DEFAULT) -> ParsedScene:
    return ParsedScene(text, flags)


def load_hdr(path: str) -> Optional[np.ndarray]:
    """(height, width, 3) float32, rows top to bottom, or None if the file cannot be decoded."""
    w, h = C.c_uint64(), C.c_uint64()
    p = lib().pthost_load_hdr(path.encode(), C.byref(w), C.byref(h))
    if not p:
        return None
    out = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    lib().pthost_free(p)
    return out

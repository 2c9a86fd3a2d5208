"""Scene description objects, mirroring the reference's data model.

These are plain containers for what the reference's parser produces (reference src/scene.rs:71-167,
src/tracer.rs:8-13,43-66; built by src/parser.rs:50-112).  Nothing is computed here: the derived values
(plane normal, camera projection plane) are produced by the library that consumes the scene, the way the
Rust host computes them in Plane::new / Camera::new.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

Vec3 = Tuple[float, float, float]

# Binary layouts of the C ABI's primitive structs (all fields 4 bytes wide, no padding).
MATERIAL_DT = np.dtype([("kind", "<i4"), ("color", "<f4", 3), ("roughness", "<f4"), ("metalness", "<f4"), ("ior", "<f4")])
SPHERE_DT = np.dtype([("origin", "<f4", 3), ("radius", "<f4"), ("material", MATERIAL_DT)])
PLANE_DT = np.dtype([("origin", "<f4", 3), ("u", "<f4", 3), ("v", "<f4", 3), ("material", MATERIAL_DT), ("normal", "<f4", 3)])

# reference src/scene.rs:84-92
MAT_NONE, MAT_EMISSIVE, MAT_MIRROR, MAT_TRANSLUCENT, MAT_PHYSICALLY = 0, 1, 2, 3, 4
# reference src/tracer.rs:43-48
TONE_CLAMP, TONE_REINHARD, TONE_EXPOSURE = 0, 1, 2
# reference src/scene.rs:71-75
SKY_CONSTANT, SKY_HDRI = 0, 1


@dataclass
class Material:
    kind: int = MAT_NONE
    color: Vec3 = (0.0, 0.0, 0.0)  # Emissive radiance / Physically reflectivity
    roughness: float = 0.0
    metalness: float = 0.0
    ior: float = 1.0

    @staticmethod
    def none() -> "Material":
        return Material(MAT_NONE)

    @staticmethod
    def emissive(radiance: Vec3) -> "Material":
        return Material(MAT_EMISSIVE, tuple(radiance))

    @staticmethod
    def mirror() -> "Material":
        return Material(MAT_MIRROR)

    @staticmethod
    def translucent(ior: float) -> "Material":
        return Material(MAT_TRANSLUCENT, ior=ior)

    @staticmethod
    def physically(reflectivity: Vec3, roughness: float, metalness: float) -> "Material":
        return Material(MAT_PHYSICALLY, tuple(reflectivity), roughness, metalness)


@dataclass
class Sphere:
    origin: Vec3
    radius: float
    material: Material


@dataclass
class Plane:
    """A finite one-sided parallelogram origin + a*u + b*v (reference src/scene.rs:95-114)."""
    origin: Vec3
    u: Vec3
    v: Vec3
    material: Material


@dataclass
class Camera:
    """Arguments of Camera::new (reference src/tracer.rs:69)."""
    position: Vec3 = (0.0, 2.0, 20.0)
    target: Vec3 = (0.0, 0.0, 0.0)
    up: Vec3 = (0.0, 1.0, 0.0)
    width: float = 4.0
    height: float = 4.0
    z_near: float = 10.0
    tone_mapping: int = TONE_CLAMP
    exposure: float = 1.0
    iso: float = 100.0


@dataclass
class Sky:
    kind: int = SKY_CONSTANT
    radiance: Vec3 = (1.0, 1.0, 1.0)
    path: str = ""
    pixels: Optional[np.ndarray] = None  # (height, width, 3) float32 once loaded

    @staticmethod
    def constant(radiance: Vec3) -> "Sky":
        return Sky(SKY_CONSTANT, tuple(radiance))

    @staticmethod
    def hdri(path: str, pixels: Optional[np.ndarray] = None) -> "Sky":
        return Sky(SKY_HDRI, (1.0, 1.0, 1.0), path, pixels)


@dataclass
class ImageSettings:
    width: int = 256
    height: int = 256
    sub_pixel_sampling: bool = True


@dataclass
class Scene:
    """reference src/scene.rs:151-160; defaults as parse_scene's (src/parser.rs:55-61)."""
    image_settings: ImageSettings = field(default_factory=ImageSettings)
    camera: Camera = field(default_factory=Camera)
    sky: Sky = field(default_factory=Sky)
    spheres: List[Sphere] = field(default_factory=list)
    planes: List[Plane] = field(default_factory=list)
    emissive_spheres: List[Sphere] = field(default_factory=list)
    emissive_planes: List[Plane] = field(default_factory=list)

    def add(self, prim) -> None:
        """Route a primitive the way parse_scene does (reference src/parser.rs:76-94)."""
        emissive = prim.material.kind == MAT_EMISSIVE
        if isinstance(prim, Sphere):
            (self.emissive_spheres if emissive else self.spheres).append(prim)
        else:
            (self.emissive_planes if emissive else self.planes).append(prim)

    def arrays(self):
        """(spheres, planes, emissive_spheres, emissive_planes) as SPHERE_DT / PLANE_DT arrays; plane normals
        are left zero for the consuming library to fill (Plane::new, reference src/scene.rs:106)."""
        return (spheres_to_array(self.spheres), planes_to_array(self.planes),
                spheres_to_array(self.emissive_spheres), planes_to_array(self.emissive_planes))

    @property
    def n_primitives(self) -> int:
        return len(self.spheres) + len(self.planes) + len(self.emissive_spheres) + len(self.emissive_planes)


def _fill_material(dst, mats) -> None:
    dst["kind"] = [m.kind for m in mats]
    dst["color"] = [m.color for m in mats]
    dst["roughness"] = [m.roughness for m in mats]
    dst["metalness"] = [m.metalness for m in mats]
    dst["ior"] = [m.ior for m in mats]


def spheres_to_array(spheres) -> np.ndarray:
    if isinstance(spheres, np.ndarray):
        assert spheres.dtype == SPHERE_DT
        return np.ascontiguousarray(spheres)
    out = np.zeros(len(spheres), dtype=SPHERE_DT)
    if len(spheres):
        out["origin"] = [s.origin for s in spheres]
        out["radius"] = [s.radius for s in spheres]
        _fill_material(out["material"], [s.material for s in spheres])
    return out


def planes_to_array(planes) -> np.ndarray:
    if isinstance(planes, np.ndarray):
        assert planes.dtype == PLANE_DT
        return np.ascontiguousarray(planes)
    out = np.zeros(len(planes), dtype=PLANE_DT)
    if len(planes):
        out["origin"] = [p.origin for p in planes]
        out["u"] = [p.u for p in planes]
        out["v"] = [p.v for p in planes]
        _fill_material(out["material"], [p.material for p in planes])
    return out


def sample_scene(width: int = 1024, height: int = 1024, camera_width: float = 5.0) -> Scene:
    """The reference's one shipped scene (values of reference scenes/sample/sample.scene:1-145): a closed
    10x10x10 box with red/green side walls, a ceiling light, a glass, a glossy and a metal sphere.
    `camera_width` = 8.8888889 with 1920x1080 gives the 16:9 variant with square pixels."""
    white = Material.physically((1.0, 1.0, 1.0), 0.8, 0.0)
    sc = Scene(
        image_settings=ImageSettings(width, height, True),
        camera=Camera((0.0, 5.0, 15.0), (0.0, 5.0, 0.0), (0.0, 1.0, 0.0), camera_width, 5.0, 5.0, TONE_CLAMP, 1.0, 200.0),
        sky=Sky.constant((0.1, 0.1, 0.1)),
    )
    # file order matters: it defines primitive ids
    sc.add(Plane((-5.0, 10.0, -5.0), (10.0, 0.0, 0.0), (0.0, 0.0, 10.0), white))                       # ceiling
    sc.add(Plane((-2.5, 9.9, -2.5), (5.0, 0.0, 0.0), (0.0, 0.0, 5.0), Material.emissive((10.0, 10.0, 10.0))))  # light
    sc.add(Plane((-5.0, 0.0, 5.0), (10.0, 0.0, 0.0), (0.0, 0.0, -10.0), white))                        # floor
    sc.add(Plane((-5.0, 0.0, -5.0), (10.0, 0.0, 0.0), (0.0, 10.0, 0.0), white))                        # back
    sc.add(Plane((-5.0, 0.0, -5.0), (0.0, 10.0, 0.0), (0.0, 0.0, 10.0), Material.physically((1.0, 0.0, 0.0), 0.8, 0.0)))  # left
    sc.add(Plane((5.0, 0.0, 5.0), (0.0, 10.0, 0.0), (0.0, 0.0, -10.0), Material.physically((0.0, 1.0, 0.0), 0.8, 0.0)))   # right
    sc.add(Plane((5.0, 0.0, 5.0), (-10.0, 0.0, 0.0), (0.0, 10.0, 0.0), white))                         # front
    sc.add(Sphere((2.0, 2.0, 0.0), 2.0, Material.translucent(1.5)))                                     # glass
    sc.add(Sphere((-2.2, 6.2, -0.5), 2.0, Material.physically((1.0, 0.79, 0.74), 0.0, 0.0)))            # glossy
    sc.add(Sphere((-1.8, 1.5, 3.0), 1.5, Material.physically((0.95, 0.93, 0.88), 0.0, 1.0)))            # metal
    return sc


def diffuse_room_scene(width: int = 1920, height: int = 1080, n_spheres: int = 20, seed: int = 777) -> Scene:
    """Closed diffuse room without glass or roughness-0 metals: six inward rectangles (roughness 1), a small
    ceiling light of radiance 20 and `n_spheres` diffuse spheres.  Used with max_depth = 16 to stress queue
    compaction: most paths live all 16 levels, a shrinking fraction ends on the light."""
    rng = np.random.RandomState(seed)
    grey = Material.physically((0.8, 0.8, 0.8), 1.0, 0.0)
    sc = Scene(
        image_settings=ImageSettings(width, height, True),
        camera=Camera((0.0, 5.0, 14.5), (0.0, 5.0, 0.0), (0.0, 1.0, 0.0), 5.0 * width / height, 5.0, 5.0, TONE_CLAMP, 1.0, 200.0),
        sky=Sky.constant((0.0, 0.0, 0.0)),
    )
    sc.add(Plane((-5.0, 10.0, -5.0), (10.0, 0.0, 0.0), (0.0, 0.0, 20.0), grey))
    sc.add(Plane((-1.5, 9.9, -1.5), (3.0, 0.0, 0.0), (0.0, 0.0, 3.0), Material.emissive((20.0, 20.0, 20.0))))
    sc.add(Plane((-5.0, 0.0, 15.0), (10.0, 0.0, 0.0), (0.0, 0.0, -20.0), grey))
    sc.add(Plane((-5.0, 0.0, -5.0), (10.0, 0.0, 0.0), (0.0, 10.0, 0.0), grey))
    sc.add(Plane((-5.0, 0.0, -5.0), (0.0, 10.0, 0.0), (0.0, 0.0, 20.0), Material.physically((0.8, 0.2, 0.2), 1.0, 0.0)))
    sc.add(Plane((5.0, 0.0, 15.0), (0.0, 10.0, 0.0), (0.0, 0.0, -20.0), Material.physically((0.2, 0.8, 0.2), 1.0, 0.0)))
    sc.add(Plane((5.0, 0.0, 15.0), (-10.0, 0.0, 0.0), (0.0, 10.0, 0.0), grey))
    for _ in range(n_spheres):
        r = float(rng.uniform(0.4, 1.2))
        o = (float(rng.uniform(-4.0, 4.0)), float(rng.uniform(r, 6.0)), float(rng.uniform(-4.0, 6.0)))
        col = tuple(float(c) for c in rng.uniform(0.3, 0.9, 3))
        sc.add(Sphere(o, r, Material.physically(col, 1.0, 0.0)))
    return sc


def quad_field_scene(width: int = 1920, height: int = 1080, grid: int = 1000, n_spheres: int = 1000, seed: int = 12345) -> Scene:
    """Procedural stress scene for BVH traversal: a grid x grid height field of small, randomly tilted
    rectangles (u perpendicular to v) over [-50,50]^2, `n_spheres` small spheres above it, one large emissive
    rectangle overhead, constant sky 0.5.  grid = 1000 gives ~1M planes.  Primitives are held as numpy
    structured arrays (PLANE_DT / SPHERE_DT) instead of object lists."""
    rng = np.random.RandomState(seed)
    sc = Scene(
        image_settings=ImageSettings(width, height, True),
        camera=Camera((0.0, 26.0, 62.0), (0.0, 0.0, 8.0), (0.0, 1.0, 0.0), 8.0 * width / height, 8.0, 10.0, TONE_CLAMP, 1.0, 100.0),
        sky=Sky.constant((0.5, 0.5, 0.5)),
    )
    cell = 100.0 / grid
    n = grid * grid
    gx, gz = np.meshgrid(np.arange(grid), np.arange(grid), indexing="ij")
    gx = gx.reshape(-1); gz = gz.reshape(-1)
    cx = -50.0 + (gx + 0.5) * cell + rng.uniform(-0.2, 0.2, n) * cell
    cz = -50.0 + (gz + 0.5) * cell + rng.uniform(-0.2, 0.2, n) * cell
    cy = 2.0 * np.sin(cx * 0.15) * np.cos(cz * 0.11) + rng.uniform(-0.05, 0.05, n)
    yaw = rng.uniform(0.0, 2.0 * np.pi, n)
    tilt = rng.uniform(-0.35, 0.35, n)
    c, s_ = np.cos(yaw), np.sin(yaw)
    ct, st = np.cos(tilt), np.sin(tilt)
    # u along the yawed x axis, v perpendicular to it and tilted out of the ground plane; u x v = (s st, ct, c st)
    # points up, so the one-sided rectangles face the camera and the light
    u = np.stack([c, np.zeros(n), -s_], axis=1) * cell
    v = np.stack([-s_ * ct, st, -c * ct], axis=1) * cell
    centre = np.stack([cx, cy, cz], axis=1)
    planes = np.zeros(n, dtype=PLANE_DT)
    planes["u"] = u.astype(np.float32)
    planes["v"] = v.astype(np.float32)
    planes["origin"] = (centre - 0.5 * (u + v)).astype(np.float32)
    planes["material"]["kind"] = MAT_PHYSICALLY
    planes["material"]["color"] = rng.uniform(0.2, 1.0, (n, 3)).astype(np.float32)
    planes["material"]["roughness"] = rng.uniform(0.2, 1.0, n).astype(np.float32)
    planes["material"]["metalness"] = (rng.uniform(0.0, 1.0, n) < 0.1).astype(np.float32)
    planes["material"]["ior"] = 1.0
    sc.planes = planes
    spheres = np.zeros(n_spheres, dtype=SPHERE_DT)
    spheres["radius"] = rng.uniform(0.05, 0.5, n_spheres).astype(np.float32)
    spheres["origin"] = np.stack([rng.uniform(-45.0, 45.0, n_spheres), rng.uniform(2.5, 6.0, n_spheres),
                                  rng.uniform(-45.0, 45.0, n_spheres)], axis=1).astype(np.float32)
    spheres["material"]["kind"] = MAT_PHYSICALLY
    spheres["material"]["color"] = rng.uniform(0.2, 1.0, (n_spheres, 3)).astype(np.float32)
    spheres["material"]["roughness"] = rng.uniform(0.2, 1.0, n_spheres).astype(np.float32)
    spheres["material"]["ior"] = 1.0
    sc.spheres = spheres
    # overhead light, normal u x v = (0,-1,0) facing the field
    sc.add(Plane((-30.0, 40.0, -30.0), (60.0, 0.0, 0.0), (0.0, 0.0, 60.0), Material.emissive((6.0, 6.0, 6.0))))
    return sc


def _f(x) -> str:
    """A float in the `.scene` dialect: digits '.' digits, no exponent (reference src/parser.rs:615-633); shortest
    text that reads back to the same f32."""
    return np.format_float_positional(np.float32(x), unique=True, trim="0")


def _v(v) -> str:
    return "(" + ", ".join(_f(c) for c in v) + ")"


def _material_text(kind, color, roughness, metalness, ior) -> str:
    if kind == MAT_EMISSIVE:
        return "emissive {\n        radiance = %s\n    }" % _v(color)
    if kind == MAT_TRANSLUCENT:
        return "translucent {\n        ior = %s\n    }" % _f(ior)
    if kind == MAT_PHYSICALLY:
        return ("physically {\n        reflectivity = %s\n        roughness = %s\n        metalness = %s\n    }"
                % (_v(color), _f(roughness), _f(metalness)))
    raise ValueError("the .scene grammar has no syntax for Mirror / None materials (reference src/parser.rs:275-292)")


def to_text(scene: Scene) -> str:
    """The scene in the reference's `.scene` dialect (field order and newline rules of reference
    src/parser.rs:114-426), loadable by the unchanged Rust parser and by pathtracer_amd/host/parser.cpp."""
    out = []
    im, cam = scene.image_settings, scene.camera
    out.append("image {\n    width = %d\n    height = %d\n    sub_pixel_sampling = true\n}\n" % (im.width, im.height))
    if scene.sky.kind == SKY_HDRI:
        out.append("sky {\n    hdri {\n        path = %s\n    }\n}\n" % scene.sky.path)
    else:
        out.append("sky {\n    constant {\n        radiance = %s\n    }\n}\n" % _v(scene.sky.radiance))
    tm = {TONE_CLAMP: "clamp", TONE_REINHARD: "reinhard"}.get(cam.tone_mapping, "exposure { value = %s }" % _f(cam.exposure))
    out.append("camera {\n    position = %s\n    target = %s\n    up = %s\n    width = %s\n    height = %s\n    z_near = %s\n"
               "    tone_mapping = %s\n    iso = %s\n}\n" % (_v(cam.position), _v(cam.target), _v(cam.up), _f(cam.width), _f(cam.height),
                                                              _f(cam.z_near), tm, _f(cam.iso)))
    sp, pl, es, ep = scene.arrays()
    for arr in (pl, ep):
        for p in arr:
            m = p["material"]
            out.append("plane {\n    origin = %s\n    u = %s\n    v = %s\n    material = %s\n}\n" % (
                _v(p["origin"]), _v(p["u"]), _v(p["v"]),
                _material_text(int(m["kind"]), m["color"], m["roughness"], m["metalness"], m["ior"])))
    for arr in (sp, es):
        for s in arr:
            m = s["material"]
            out.append("sphere {\n    origin = %s\n    radius = %s\n    material = %s\n}\n" % (
                _v(s["origin"]), _f(s["radius"]),
                _material_text(int(m["kind"]), m["color"], m["roughness"], m["metalness"], m["ior"])))
    return "\n".join(out)

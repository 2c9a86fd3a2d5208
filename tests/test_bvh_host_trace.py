"""The BVH traversal of the kernels, compiled for the CPU (ptgpu_bvh_host_trace: the same functions, every reference to a
unit of the tree and every stack slot checked), against the oracle's linear scan (reference src/scene.rs:263-297):
  * over every ray the oracle traces while it renders randomised scenes - primary, bounce and light rays, bit for bit
    the rays the GPU traces, among them the NaN rays that follow a hit in the centre of a sphere of radius zero;
  * over rays no render produces: NaN, infinite and huge origins and directions, origins far outside the scene aimed
    at it, against trees with partly filled nodes, an empty tree and primitives that cannot be bounded.
A reference that leaves the tree is what faults a GPU; here it is a count that must be zero.  No GPU needed.
"""
import numpy as np
import pytest

from pathtracer_amd import binding, scene as S
from test_gpu_fuzz import random_scene, _material

NO_FAULTS = dict(bad_units=0, bad_kinds=0, bad_slots=0)


def _check_against_linear_scan(sc, osc, rays, want_prims=None, want_ts=None):
    prims, ts, f = binding.bvh_host_trace(sc, rays)
    assert {k: f[k] for k in NO_FAULTS} == NO_FAULTS
    if want_prims is None:
        want_prims = np.empty(len(rays), dtype=np.uint32)
        want_ts = np.zeros(len(rays), dtype=np.float32)
        for k, r in enumerate(rays):
            found, h = osc.find_scene_hit(tuple(float(x) for x in r[:3]), tuple(float(x) for x in r[3:6]))
            want_prims[k] = h.prim_id if found else 0xFFFFFFFF
            want_ts[k] = h.t if found else 0.0
    assert np.array_equal(prims, want_prims)
    hit = want_prims != 0xFFFFFFFF
    assert np.array_equal(ts[hit], want_ts[hit], equal_nan=True)
    return f


@pytest.mark.parametrize("seed", range(64))
def test_every_ray_of_a_render(oracle, seed):
    sc, depth, thin = random_scene(seed)
    osc = oracle.OracleScene(sc)
    opt = oracle.options(oracle.RNG_COUNTER, oracle.MATH_DET, max_depth=depth, camera_model=thin)
    _, _, n_rays, log = osc.render_logging_rays(opt, 2)
    assert len(log) == n_rays
    _check_against_linear_scan(sc, osc, log[:, :6], log[:, 6].view(np.uint32), log[:, 7])


def point_sphere_scene():
    sc = S.Scene(image_settings=S.ImageSettings(32, 32, False),   # (un-jittered samples sit on pixel corners: pixel (16, 16) looks down the axis)
                 camera=S.Camera((1.0, 3.0, 10.0), (1.0, 3.0, 0.0), (0.0, 1.0, 0.0), 8.0, 8.0, 5.0, S.TONE_CLAMP, 1.0, 100.0),
                 sky=S.Sky.constant((0.5, 0.6, 0.7)))
    grey = S.Material.physically((0.8, 0.8, 0.8), 0.5, 0.0)
    sc.add(S.Plane((-8.0, 0.0, 8.0), (16.0, 0.0, 0.0), (0.0, 0.0, -16.0), grey))
    sc.add(S.Sphere((1.0, 3.0, 0.0), 0.0, grey))
    sc.add(S.Sphere((-3.0, 4.0, -1.0), 0.8, S.Material.mirror()))
    sc.add(S.Sphere((3.4, 7.2, -3.2), 0.8, grey))
    sc.add(S.Plane((-2.0, 9.5, -2.0), (4.0, 0.0, 0.0), (0.0, 0.0, 4.0), S.Material.emissive((10.0, 9.0, 8.0))))
    return sc


def test_bounce_off_the_centre_of_a_point_sphere(oracle):
    """The ray through the image centre hits a sphere of radius zero exactly in its centre: the normal is 0/0 and the
    next ray starts at NaN.  Its slab distances would all be NaN; it must not enter the tree."""
    sc = point_sphere_scene()
    osc = oracle.OracleScene(sc)
    opt = oracle.options(oracle.RNG_COUNTER, oracle.MATH_DET, max_depth=2)
    _, _, _, log = osc.render_logging_rays(opt, 1)
    assert np.isnan(log[:, :3]).any(), "the construction no longer produces a NaN origin"
    _check_against_linear_scan(sc, osc, log[:, :6], log[:, 6].view(np.uint32), log[:, 7])


def _scenes():
    yield "fuzz", random_scene(3)[0]
    yield "fuzz with a point sphere", random_scene(205928)[0]
    # a few dozen small primitives: nodes at every fill from one child to four
    rng = np.random.RandomState(5)
    sc = S.Scene(image_settings=S.ImageSettings(16, 16, False))
    for _ in range(37):
        sc.add(S.Sphere(tuple(float(x) for x in rng.uniform(-6, 6, 3)), float(rng.uniform(0.05, 0.9)), _material(rng)))
    for _ in range(22):
        o = rng.uniform(-6, 6, 3); u = rng.uniform(-2, 2, 3); v = np.cross(u, rng.uniform(-2, 2, 3))
        sc.add(S.Plane(tuple(map(float, o)), tuple(map(float, u)), tuple(map(float, v / 3.0)), _material(rng)))
    yield "clutter", sc
    one = S.Scene(image_settings=S.ImageSettings(16, 16, False))
    one.add(S.Sphere((0.0, 1.0, 0.0), 1.0, S.Material.mirror()))
    yield "one sphere", one
    # nothing that can be bounded (an empty tree) next to primitives that are tested for every ray
    far = S.Scene(image_settings=S.ImageSettings(16, 16, False))
    far.add(S.Sphere((float("nan"), 0.0, 0.0), 1.0, S.Material.mirror()))
    far.add(S.Plane((-3e18, 0.0, 1.5), (6e18, 0.0, 0.0), (0.0, 0.0, -3.0), S.Material.none()))
    yield "unboundable only", far
    # a small scene over a strip that runs far beyond the tree (so it is tested for every ray, and hits on it start rays
    # far outside the tree); a square plate of that length has no finite normal in f32
    mix = S.Scene(image_settings=S.ImageSettings(64, 48, True),
                  camera=S.Camera((0.0, 1.5, 12.0), (0.0, 1.45, 0.0), (0.0, 1.0, 0.0), 8.0, 6.0, 5.0, S.TONE_REINHARD, 1.0, 100.0),
                  sky=S.Sky.constant((0.6, 0.7, 0.9)))
    rng = np.random.RandomState(8)
    mix.add(S.Plane((-3e18, 0.0, 1.5), (6e18, 0.0, 0.0), (0.0, 0.0, -3.0), S.Material.physically((0.5, 0.5, 0.5), 0.5, 0.0)))
    for _ in range(14):
        mix.add(S.Sphere((float(rng.uniform(-5, 5)), float(rng.uniform(0.3, 3)), float(rng.uniform(-5, 5))), float(rng.uniform(0.3, 1.2)), _material(rng)))
    for _ in range(8):
        o = rng.uniform(-5, 5, 3); o[1] = abs(o[1]) * 0.6 + 0.1; u = rng.uniform(-2, 2, 3); v = np.cross(u, rng.uniform(-2, 2, 3))
        mix.add(S.Plane(tuple(map(float, o)), tuple(map(float, u)), tuple(map(float, v / 3.0)), _material(rng)))
    yield "small scene over a long strip", mix
    # a rectangle whose origin is NaN: the reference's plane test accepts it, at distance NaN, for every ray that faces it, and
    # its scan then keeps whatever it found first for good
    for where in ("first", "last"):
        sc = point_sphere_scene()
        nan_plane = S.Plane((float("nan"), 6.0, 0.0), (1.0, 0.0, 0.0), (0.0, 0.0, 1.0), S.Material.physically((0.9, 0.2, 0.2), 0.5, 0.0))
        sc.planes.insert(0 if where == "first" else len(sc.planes), nan_plane)
        yield f"a plane at NaN ({where})", sc


@pytest.mark.parametrize("name,sc", list(_scenes()), ids=[n for n, _ in _scenes()])
def test_every_ray_of_a_render_of_the_special_scenes(oracle, name, sc):
    osc = oracle.OracleScene(sc)
    _, _, _, log = osc.render_logging_rays(oracle.options(oracle.RNG_COUNTER, oracle.MATH_DET, max_depth=3), 2)
    _check_against_linear_scan(sc, osc, log[:, :6], log[:, 6].view(np.uint32), log[:, 7])


@pytest.mark.parametrize("name,sc", list(_scenes()), ids=[n for n, _ in _scenes()])
def test_rays_no_render_produces(oracle, name, sc):
    osc = oracle.OracleScene(sc)
    rng = np.random.RandomState(99)
    n = 6000
    special = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-45, 3e38, -3e38, 1e30, -1e20, 1e10, -1e6, 1e3], dtype=np.float32)
    o = rng.uniform(-10, 10, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    for arr, p in ((o, 0.25), (d, 0.15)):
        m = rng.rand(n, 3) < p
        arr[m] = special[rng.randint(0, len(special), int(m.sum()))]
    # and rays from far away, aimed at points inside the scene (the distances differ by less than their rounding)
    k = n // 3
    target = rng.uniform(-5, 5, (k, 3))
    away = rng.normal(size=(k, 3)); away /= np.linalg.norm(away, axis=1, keepdims=True)
    dist = 10.0 ** rng.uniform(1.5, 14.0, (k, 1))
    o[:k] = (target + away * dist).astype(np.float32)
    back = (target - o[:k].astype(np.float64)); back /= np.linalg.norm(back, axis=1, keepdims=True)
    d[:k] = back.astype(np.float32)
    axis = rng.rand(k) < 0.3   # some of them exactly along an axis
    a = rng.randint(0, 3, k)
    for i in np.where(axis)[0]:
        s = np.sign(d[i, a[i]]) or 1.0
        d[i] = 0.0; d[i, a[i]] = s
        o[i] = target[i]; o[i, a[i]] = -s * dist[i, 0]
    f = _check_against_linear_scan(sc, osc, np.concatenate([o, d], axis=1))
    tiny = name in ("one sphere", "unboundable only")   # a tree of fewer than three units is not walked: its one primitive is scanned
    assert (f["steps"] == 0) if tiny else (f["steps"] > 0)
    # Safety by construction: with the gate off - every ray, NaN and infinite ones included, is walked through the tree, so
    # the answers may differ from the scan's - the walk still follows only references the builder wrote (each node names
    # how many children it has), ends, and keeps its stack in bounds.
    _, _, g = binding.bvh_host_trace(sc, np.concatenate([o, d], axis=1), flags=binding.HOST_TRACE_UNGATED)
    assert g["bad_units"] == 0 and g["bad_slots"] == 0 and g["steps"] >= f["steps"]

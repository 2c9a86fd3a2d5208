"""The BVH must return exactly what the reference's linear scan returns (reference src/scene.rs:263-297):
same primitive id for every ray, hence the same accumulated image.  Checked against the oracle where the
oracle finishes in seconds, and against the GPU's own linear scan (itself oracle-checked in
test_gpu_parity.py) at sizes the oracle cannot reach."""
import numpy as np
import pytest

from conftest import materials_scene
from pathtracer_amd import scene as S

pytestmark = pytest.mark.gpu


def _ids_and_pixels(binding, sc, accel, spp, **cfg):
    with binding.PtGpu(accel=accel, **cfg) as gpu:
        gpu.upload_scene(sc)
        ids = [gpu.primary_ids(s) for s in range(2)]
        gpu.render(spp)
        p32, _ = gpu.read_pixels32()
        st = gpu.stats()
    return ids, p32, st


@pytest.mark.parametrize("make", [lambda: S.sample_scene(128, 96, camera_width=6.6666667), lambda: materials_scene(96, 72),
                                  lambda: S.diffuse_room_scene(96, 54, n_spheres=20)])
def test_small_scenes_bvh_equals_linear_scan_and_oracle(binding, oracle, make):
    sc = make()
    ids_b, p_b, st_b = _ids_and_pixels(binding, sc, binding.ACCEL_BVH, 4)
    ids_l, p_l, st_l = _ids_and_pixels(binding, sc, binding.ACCEL_BRUTE_FORCE, 4)
    assert st_b["bvh_nodes"] > 0 and st_l["bvh_nodes"] == 0
    for a, b in zip(ids_b, ids_l):
        assert np.array_equal(a, b)
    assert np.array_equal(p_b, p_l)
    assert st_b["rays"] == st_l["rays"]
    osc = oracle.OracleScene(sc)
    opt = oracle.options()
    assert np.array_equal(ids_b[1], osc.primary_ids(opt, 1))
    r32, _, rays = osc.render(opt, 4)
    assert np.array_equal(p_b, r32) and st_b["rays"] == rays


def test_quad_field_10k_against_oracle(binding, oracle):
    sc = S.quad_field_scene(96, 54, grid=100, n_spheres=100)
    ids_b, p_b, st_b = _ids_and_pixels(binding, sc, binding.ACCEL_AUTO, 2)
    assert st_b["bvh_nodes"] > 1000  # auto picks the BVH for 10k primitives
    osc = oracle.OracleScene(sc)
    opt = oracle.options()
    for s in range(2):
        assert np.array_equal(ids_b[s], osc.primary_ids(opt, s))
    r32, _, rays = osc.render(opt, 2)  # bounce and shadow rays go through the BVH too
    assert np.array_equal(p_b, r32)
    assert st_b["rays"] == rays


def test_quad_field_250k_bvh_equals_gpu_linear_scan(binding):
    sc = S.quad_field_scene(160, 90, grid=500, n_spheres=500)
    ids_b, p_b, st_b = _ids_and_pixels(binding, sc, binding.ACCEL_BVH, 2)
    ids_l, p_l, st_l = _ids_and_pixels(binding, sc, binding.ACCEL_BRUTE_FORCE, 2)
    for a, b in zip(ids_b, ids_l):
        assert np.array_equal(a, b)
    assert np.array_equal(p_b, p_l)
    assert st_b["rays"] == st_l["rays"]
    hit = ids_b[0] != 0xFFFFFFFF
    assert hit.mean() > 0.4 and len(np.unique(ids_b[0])) > 1000  # the camera really looks at the field


def test_unboundable_and_dead_planes(binding, oracle):
    """A plane with parallel u, v has a NaN normal and can never be hit; one with a usable normal but a zero-length
    v cannot be bounded and must still be tested for every ray."""
    sc = S.sample_scene(64, 48, camera_width=6.6666667)
    sc.add(S.Plane((0.0, 1.0, 0.0), (1.0, 0.0, 0.0), (2.0, 0.0, 0.0), S.Material.physically((1.0, 1.0, 1.0), 0.5, 0.0)))
    ids_b, p_b, _ = _ids_and_pixels(binding, sc, binding.ACCEL_BVH, 2)
    ids_l, p_l, _ = _ids_and_pixels(binding, sc, binding.ACCEL_BRUTE_FORCE, 2)
    assert np.array_equal(ids_b[0], ids_l[0]) and np.array_equal(p_b, p_l)
    assert np.array_equal(ids_b[0], oracle.OracleScene(sc).primary_ids(oracle.options(), 0))


def test_deep_traversal_stack_spills_correctly(binding):
    """16384 small spheres zig-zag along the view axis: a central ray misses every sphere but pierces every box,
    so the traversal stack grows to the depth of the tree (about 12), beyond the part that is kept in LDS, and the
    entries that spill to global memory must come back intact.  Several passes run concurrently on their own
    streams, each with its own spill area."""
    n, r = 16384, 0.05
    sc = S.Scene(image_settings=S.ImageSettings(48, 36, True),
                 camera=S.Camera((0.0, 0.0, 40.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 0.012, 0.009, 20.0, S.TONE_CLAMP, 1.0, 100.0),
                 sky=S.Sky.constant((0.6, 0.7, 0.8)))
    k = np.arange(n)
    spheres = np.zeros(n, dtype=S.SPHERE_DT)
    spheres["radius"] = r
    spheres["origin"] = np.stack([np.where(k % 2 == 0, 2.0 * r, -2.0 * r), np.zeros(n), 10.0 - k * 0.01], axis=1).astype(np.float32)
    spheres["material"]["kind"] = S.MAT_PHYSICALLY
    spheres["material"]["color"] = (0.8, 0.5, 0.3)
    spheres["material"]["roughness"] = 0.6
    spheres["material"]["ior"] = 1.0
    sc.spheres = spheres
    sc.add(S.Plane((-30.0, -30.0, -160.0), (60.0, 0.0, 0.0), (0.0, 60.0, 0.0), S.Material.physically((0.7, 0.7, 0.7), 0.8, 0.0)))
    sc.add(S.Plane((-4.0, 6.0, -60.0), (8.0, 0.0, 0.0), (0.0, 0.0, 40.0), S.Material.emissive((9.0, 9.0, 9.0))))
    ids_b, p_b, st_b = _ids_and_pixels(binding, sc, binding.ACCEL_BVH, 6, samples_per_batch=1, concurrent_passes=3)
    ids_l, p_l, st_l = _ids_and_pixels(binding, sc, binding.ACCEL_BRUTE_FORCE, 6)
    for a, b in zip(ids_b, ids_l):
        assert np.array_equal(a, b)
    assert np.array_equal(p_b, p_l) and st_b["rays"] == st_l["rays"]
    centre = ids_b[0].reshape(36, 48)[16:20, 22:26]
    assert np.all(centre == n)  # the central rays thread the whole row and end on the back plane
    assert len(np.unique(ids_b[0])) > 50


@pytest.mark.parametrize("pipes", [1, 2, 4])
def test_concurrent_passes_sum_to_the_same_image(binding, oracle, pipes):
    """Passes in flight on separate streams add into one integer accumulator: any number of them, any pass size,
    gives the sums of a single pass (and of the oracle)."""
    sc = materials_scene(96, 72)
    with binding.PtGpu(samples_per_batch=2, concurrent_passes=pipes) as gpu:
        gpu.upload_scene(sc)
        gpu.render(5)
        gpu.render(4)
        p32, n = gpu.read_pixels32()
        rays = gpu.stats()["rays"]
        gpu.clear()
        gpu.render(3)
        q32, m = gpu.read_pixels32()
    assert n == 9 and m == 3
    osc = oracle.OracleScene(sc)
    r32, _, r_rays = osc.render(oracle.options(), 9)
    assert np.array_equal(p32, r32) and rays == r_rays
    assert np.array_equal(q32, osc.render(oracle.options(), 3)[0])


def test_rays_parallel_to_an_axis(binding, oracle):
    """Without jitter the centre column and row of a symmetric camera have a direction component that is exactly zero:
    the slab test must not turn the infinite reciprocal into a culled box."""
    sc = S.sample_scene(64, 64)
    osc = oracle.OracleScene(sc)
    ref = osc.primary_ids(oracle.options(sub_pixel_override=0), 0)
    assert (ref[:, 32] != 0xFFFFFFFF).all()
    with binding.PtGpu(accel=2, sub_pixel_override=0) as gpu:
        gpu.upload_scene(sc)
        ids = gpu.primary_ids(0)
        gpu.render(2)
        p32, _ = gpu.read_pixels32()
    assert np.array_equal(ids, ref)
    r32, _, _ = osc.render(oracle.options(sub_pixel_override=0), 2)
    assert np.array_equal(p32, r32)


# ---- the ~1M-primitive procedural scene at its full size ------------------------------------------------------------
# The oracle needs ~27 ms per ray here (it scans a million primitives the way find_scene_hit does), so it checks small
# images of the same camera; the GPU's own linear scan, itself oracle-checked everywhere else, covers the larger ones.
def _full_field(width, height):
    return S.quad_field_scene(width, height, grid=1000, n_spheres=1000)


def test_full_field_bvh_shape(binding):
    with binding.PtGpu(accel=binding.ACCEL_AUTO) as gpu:
        gpu.upload_scene(_full_field(64, 36))
        st = gpu.stats()
    assert st["bvh_leaf_prims"] == 1_001_001
    assert 333_667 <= st["bvh_nodes"] < 560_000      # 4-wide nodes over 1M leaves: at least (n - 1) / 3, were every node full
    # the field is regular enough that binned SAH stays balanced: no forced median split, folded by box area
    assert st["bvh_binary_depth"] <= 30 and st["bvh_forced_median_splits"] == 0 and st["bvh_level_fold"] == 0
    assert 3 * st["bvh_depth"] <= 64                 # the traversal stack's bound


def test_full_field_primary_ids_bvh_equals_gpu_linear_scan(binding):
    sc = _full_field(256, 144)
    with binding.PtGpu(accel=binding.ACCEL_BVH) as gpu:
        gpu.upload_scene(sc)
        ids_b = [gpu.primary_ids(s) for s in range(2)]
    with binding.PtGpu(accel=binding.ACCEL_BRUTE_FORCE) as gpu:
        gpu.upload_scene(sc)
        ids_l = [gpu.primary_ids(s) for s in range(2)]
    for a, b in zip(ids_b, ids_l):
        assert np.array_equal(a, b)
    hit = ids_b[0] != 0xFFFFFFFF
    assert hit.mean() > 0.4 and len(np.unique(ids_b[0])) > 5000


def test_full_field_pixels_bvh_equals_gpu_linear_scan(binding):
    sc = _full_field(64, 36)
    _, p_b, st_b = _ids_and_pixels(binding, sc, binding.ACCEL_BVH, 1)
    _, p_l, st_l = _ids_and_pixels(binding, sc, binding.ACCEL_BRUTE_FORCE, 1)
    assert np.array_equal(p_b, p_l)
    # (no shadow rays here: the overhead light faces down, so Plane::sample says None for every point below it,
    #  reference src/scene.rs:117)
    assert st_b["rays"] == st_l["rays"] > 64 * 36 and st_b["records"] == st_l["records"]


def test_full_field_against_oracle(binding, oracle):
    sc = _full_field(40, 22)
    with binding.PtGpu(accel=binding.ACCEL_BVH) as gpu:
        gpu.upload_scene(sc)
        ids = gpu.primary_ids(0)
    assert np.array_equal(ids, oracle.OracleScene(sc).primary_ids(oracle.options(), 0))
    small = _full_field(24, 14)
    _, p_b, st_b = _ids_and_pixels(binding, small, binding.ACCEL_BVH, 1)
    r32, _, rays = oracle.OracleScene(small).render(oracle.options(), 1)   # bounce and shadow rays through the BVH too
    assert np.array_equal(p_b, r32) and st_b["rays"] == rays


# ---- a tree as deep as the builder allows ------------------------------------------------------------------------------
def skewed_spheres_scene(n=300, width=48, height=36):
    """Spheres on the x axis whose positions and radii grow geometrically over 36 decades: binned SAH peels a few of
    the largest off at every split, so the binary tree is a spine that runs into the builder's depth cap (forced median
    splits) and, folded by box area, a 4-wide tree far deeper than the traversal stack could follow.  The camera looks
    along the row from beside it: its rays pierce most boxes (all of them once the boxes are padded for 1e18-sized
    coordinates) and miss most spheres."""
    ratio = 1e36 ** (1.0 / n)
    x = 1e-18 * ratio ** np.arange(n)
    sc = S.Scene(image_settings=S.ImageSettings(width, height, True),
                 camera=S.Camera((-1.0, 0.45, 0.0), (1.0, 0.9, 0.0), (0.0, 1.0, 0.0), 0.4, 0.3, 0.5, S.TONE_CLAMP, 1.0, 100.0),
                 sky=S.Sky.constant((0.6, 0.7, 0.8)))
    sp = np.zeros(n, dtype=S.SPHERE_DT)
    sp["radius"] = (0.3 * x).astype(np.float32)
    sp["origin"] = np.stack([x, np.zeros(n), np.zeros(n)], axis=1).astype(np.float32)
    sp["material"]["kind"] = S.MAT_PHYSICALLY
    sp["material"]["color"] = (0.8, 0.5, 0.3)
    sp["material"]["roughness"] = 0.6
    sp["material"]["ior"] = 1.0
    sc.spheres = sp
    sc.add(S.Plane((-4.0, 6.0, -4.0), (8.0, 0.0, 0.0), (0.0, 0.0, 8.0), S.Material.emissive((9.0, 9.0, 9.0))))
    return sc


def test_skewed_tree_is_refolded_and_traversed_exactly(binding, oracle):
    sc = skewed_spheres_scene()
    ids_b, p_b, st_b = _ids_and_pixels(binding, sc, binding.ACCEL_BVH, 3, samples_per_batch=1, concurrent_passes=2)
    assert st_b["bvh_binary_depth"] == 38 and st_b["bvh_forced_median_splits"] > 0
    assert st_b["bvh_level_fold"] == 1 and 3 * st_b["bvh_depth"] <= 64
    ids_l, p_l, st_l = _ids_and_pixels(binding, sc, binding.ACCEL_BRUTE_FORCE, 3)
    for a, b in zip(ids_b, ids_l):
        assert np.array_equal(a, b)
    assert np.array_equal(p_b, p_l) and st_b["rays"] == st_l["rays"]
    osc = oracle.OracleScene(sc)
    assert np.array_equal(ids_b[0], osc.primary_ids(oracle.options(), 0))
    assert np.array_equal(p_b, osc.render(oracle.options(), 3)[0])
    assert len(np.unique(ids_b[0])) > 5   # spheres of several sizes, the light or the sky are seen


# ---- boxes on an 8-bit grid per node: sizes over five decades, far from the coordinate origin -------------------------
def clutter_scene(seed, n_planes=2500, n_spheres=1500, width=160, height=90, centre=(5000.0, -3000.0, 2000.0)):
    """Randomly oriented rectangles (u, v NOT perpendicular in general, so their acceptance region is the slab
    intersection, not the parallelogram) and spheres with sizes log-uniform over 1e-2 .. 1e3 in a 400-unit cloud whose
    centre lies thousands of units from the origin: the grid origins of the tree's nodes are large numbers and a node's
    cell size varies by orders of magnitude between siblings' subtrees.  The camera sits inside the cloud; a light faces it."""
    rng = np.random.RandomState(seed)
    c = np.asarray(centre)
    sc = S.Scene(image_settings=S.ImageSettings(width, height, True),
                 camera=S.Camera(tuple(c + (0.0, 0.0, 120.0)), tuple(c), (0.0, 1.0, 0.0), 9.0 * width / height, 9.0, 10.0,
                                 S.TONE_CLAMP, 1.0, 100.0),
                 sky=S.Sky.constant((0.4, 0.5, 0.6)))
    size = 10.0 ** rng.uniform(-2.0, 3.0, n_planes)
    u = rng.normal(size=(n_planes, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    v = rng.normal(size=(n_planes, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    planes = np.zeros(n_planes, dtype=S.PLANE_DT)
    planes["u"] = (u * size[:, None]).astype(np.float32)
    planes["v"] = (v * (size * rng.uniform(0.2, 1.0, n_planes))[:, None]).astype(np.float32)
    planes["origin"] = (c + rng.uniform(-200.0, 200.0, (n_planes, 3))).astype(np.float32)
    planes["material"]["kind"] = S.MAT_PHYSICALLY
    planes["material"]["color"] = rng.uniform(0.2, 1.0, (n_planes, 3)).astype(np.float32)
    planes["material"]["roughness"] = rng.uniform(0.0, 1.0, n_planes).astype(np.float32)
    planes["material"]["metalness"] = (rng.uniform(0.0, 1.0, n_planes) < 0.2).astype(np.float32)
    planes["material"]["ior"] = 1.0
    sc.planes = planes
    spheres = np.zeros(n_spheres, dtype=S.SPHERE_DT)
    spheres["radius"] = (10.0 ** rng.uniform(-2.0, 1.5, n_spheres)).astype(np.float32)
    spheres["origin"] = (c + rng.uniform(-200.0, 200.0, (n_spheres, 3))).astype(np.float32)
    spheres["material"]["kind"] = np.where(rng.uniform(0.0, 1.0, n_spheres) < 0.1, S.MAT_TRANSLUCENT, S.MAT_PHYSICALLY)
    spheres["material"]["color"] = rng.uniform(0.2, 1.0, (n_spheres, 3)).astype(np.float32)
    spheres["material"]["roughness"] = rng.uniform(0.0, 1.0, n_spheres).astype(np.float32)
    spheres["material"]["ior"] = 1.5
    sc.spheres = spheres
    sc.add(S.Plane(tuple(c + (-150.0, 250.0, -150.0)), (300.0, 0.0, 0.0), (0.0, 0.0, 300.0), S.Material.emissive((8.0, 8.0, 8.0))))
    return sc


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_clutter_bvh_equals_linear_scan_and_oracle(binding, oracle, seed):
    sc = clutter_scene(seed)
    ids_b, p_b, st_b = _ids_and_pixels(binding, sc, binding.ACCEL_BVH, 3, max_depth=5)
    ids_l, p_l, st_l = _ids_and_pixels(binding, sc, binding.ACCEL_BRUTE_FORCE, 3, max_depth=5)
    # (a few rectangles with nearly parallel edges cannot be bounded and are tested for every ray)
    assert 3980 <= st_b["bvh_leaf_prims"] <= 4001 and (ids_b[0] != 0xFFFFFFFF).mean() > 0.5
    for a, b in zip(ids_b, ids_l):
        assert np.array_equal(a, b)
    assert np.array_equal(p_b, p_l) and st_b["rays"] == st_l["rays"]
    small = clutter_scene(seed, width=48, height=27)
    osc = oracle.OracleScene(small)
    with binding.PtGpu(accel=binding.ACCEL_BVH, max_depth=5) as gpu:
        gpu.upload_scene(small)
        ids = gpu.primary_ids(1)
        gpu.render(2)
        p32, _ = gpu.read_pixels32()
        rays = gpu.stats()["rays"]
    assert np.array_equal(ids, osc.primary_ids(oracle.options(max_depth=5), 1))
    r32, _, r_rays = osc.render(oracle.options(max_depth=5), 2)
    assert np.array_equal(p32, r32) and rays == r_rays


def _degenerate_scenes():
    from test_bvh_host_trace import _scenes, point_sphere_scene
    yield "point sphere hit in its centre", point_sphere_scene(), 3
    by_name = dict(_scenes())
    yield "small scene over a long strip", by_name["small scene over a long strip"], 3
    yield "unboundable only", by_name["unboundable only"], 2
    for where in ("first", "last"):   # (every ray that faces it "hits" it at distance NaN)
        yield f"a plane at NaN ({where})", by_name[f"a plane at NaN ({where})"], 3


@pytest.mark.parametrize("name,sc,depth", list(_degenerate_scenes()), ids=[n for n, *_ in _degenerate_scenes()])
def test_degenerate_rays_and_primitives(binding, oracle, name, sc, depth):
    """Rays and primitives the tree cannot answer for - a NaN origin after a hit in the centre of a sphere of radius zero,
    origins far outside the tree, distances that are NaN - are repeated by the linear scan in a second kernel; the image
    must be the linear-scan pipeline's, bit for bit, and nothing may leave the tree on the way (that is a memory fault)."""
    ids_b, p_b, st_b = _ids_and_pixels(binding, sc, binding.ACCEL_BVH, 3, max_depth=depth)
    ids_l, p_l, st_l = _ids_and_pixels(binding, sc, binding.ACCEL_BRUTE_FORCE, 3, max_depth=depth)
    for a, b in zip(ids_b, ids_l):
        assert np.array_equal(a, b)
    assert np.array_equal(p_b, p_l) and st_b["rays"] == st_l["rays"]
    # the fused BVH kernels repeat such rays in place
    _, p_f, st_f = _ids_and_pixels(binding, sc, binding.ACCEL_BVH, 3, max_depth=depth, fuse_trace_shade=1)
    assert np.array_equal(p_f, p_l) and st_f["rays"] == st_l["rays"]
    osc = oracle.OracleScene(sc)
    opt = oracle.options(max_depth=depth)
    assert np.array_equal(ids_b[1], osc.primary_ids(opt, 1))
    r32, _, rays = osc.render(opt, 3)
    assert np.array_equal(p_b, r32), f"{(p_b != r32).any(axis=2).sum()} pixels differ"
    assert st_b["rays"] == rays, (st_b["rays"], rays)

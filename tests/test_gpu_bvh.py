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

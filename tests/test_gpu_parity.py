"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar: bit-exact.  Primitive ids, u32 accumulators and display bytes are integers; the f32 radiance behind them
is reproduced operation for operation (no FMA contraction, correctly rounded div/sqrt, deterministic
transcendental functions on both sides), so even the quantised sums must be identical.
"""
import numpy as np
import pytest

from conftest import hdri_scene, materials_scene
from pathtracer_amd import scene as S

pytestmark = pytest.mark.gpu


def _oracle_render(oracle, sc, n, max_depth=4, seed=314159265, sub_pixel_override=-1, sample_indices=None):
    osc = oracle.OracleScene(sc)
    opt = oracle.options(oracle.RNG_COUNTER, oracle.MATH_DET, seed=seed, max_depth=max_depth, sub_pixel_override=sub_pixel_override)
    return osc, opt, osc.render(opt, n, sample_indices=sample_indices)


def _gpu_render(binding, sc, n, **cfg):
    with binding.PtGpu(**cfg) as gpu:
        gpu.upload_scene(sc)
        gpu.render(n)
        p32, ns = gpu.read_pixels32()
        p8 = gpu.read_pixels8()
        st = gpu.stats()
    return p32, p8, ns, st


# ---- deterministic math on the device -------------------------------------------------------------------
@pytest.mark.parametrize("fn,lo,hi", [(0, -10.0, 10.0), (1, -10.0, 10.0), (3, -110.0, 95.0), (4, -1.0, 1.0)])
def test_det_math_unary_bit_exact(binding, oracle, fn, lo, hi):
    rng = np.random.RandomState(fn + 11)
    a = rng.uniform(lo, hi, 400000).astype(np.float32)
    special = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, np.inf, -np.inf, np.nan, 1e-45, 6.2831855, 3.1415927, 1e6, -1e6,
                        1e20, 0.99999994, -0.99999994], dtype=np.float32)
    a = np.concatenate([a, special])
    with binding.PtGpu() as gpu:
        got = gpu.eval_math(fn, a)
    ref = oracle.math_eval(oracle.MATH_DET, fn, a)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_det_math_pow_atan2_bit_exact(binding, oracle):
    rng = np.random.RandomState(3)
    x = np.concatenate([rng.uniform(0.0, 1.5, 300000), rng.uniform(-2.0, 50.0, 100000)]).astype(np.float32)
    with binding.PtGpu() as gpu:
        for y in (5.0, 1.0 / 2.2, 2.0, -1.5, 3.0, 0.0):
            yy = np.full_like(x, np.float32(y))
            got = gpu.eval_math(2, x, yy)
            ref = oracle.math_eval(oracle.MATH_DET, 2, x, yy)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), y
        sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 2.0, -2.0, 0.5, 3.0, -3.0], dtype=np.float32)
        X, Y = np.meshgrid(sp, sp)
        got = gpu.eval_math(2, X.ravel(), Y.ravel())
        ref = oracle.math_eval(oracle.MATH_DET, 2, X.ravel(), Y.ravel())
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        yv = rng.uniform(-3, 3, 300000).astype(np.float32)
        xv = rng.uniform(-3, 3, 300000).astype(np.float32)
        got = gpu.eval_math(5, np.concatenate([yv, Y.ravel()]), np.concatenate([xv, X.ravel()]))
        ref = oracle.math_eval(oracle.MATH_DET, 5, np.concatenate([yv, Y.ravel()]), np.concatenate([xv, X.ravel()]))
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_gamma_table_is_exact_for_every_float_in_unit_interval(binding):
    """The resolve kernel looks trunc(255*x^(1/2.2)) up in a 255-entry step table; the device checks the lookup
    against the direct deterministic powf for all 2^30 floats in [0,1] (and NaN/negative/>1 edge ranges)."""
    with binding.PtGpu() as gpu:
        gpu.upload_scene(S.sample_scene(8, 8))
        assert gpu.selftest_gamma() == 0


# ---- primary-hit ids ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("jitter", [0, 1])
def test_primary_ids_sample_scene_1024(binding, oracle, jitter):
    sc = S.sample_scene(1024, 1024)
    with binding.PtGpu(sub_pixel_override=jitter) as gpu:
        gpu.upload_scene(sc)
        ids = gpu.primary_ids(3)
    osc = oracle.OracleScene(sc)
    ref = osc.primary_ids(oracle.options(sub_pixel_override=jitter), 3)
    assert np.array_equal(ids, ref)
    if not jitter:
        assert ids[512, 512] == 5  # centre ray: back wall (front wall is one-sided and faces away)
    assert set(np.unique(ids)) <= {0, 1, 2, 3, 4, 5, 6, 7, 9}  # id 8 (front wall) is never visible


def test_primary_ids_1080p_variant(binding, oracle):
    sc = S.sample_scene(1920, 1080, camera_width=8.8888889)
    with binding.PtGpu() as gpu:
        gpu.upload_scene(sc)
        ids = gpu.primary_ids(0)
    ref = oracle.OracleScene(sc).primary_ids(oracle.options(), 0)
    assert np.array_equal(ids, ref)


# ---- accumulated radiance ----------------------------------------------------------------------------------------
def test_sample_scene_pixels_bit_exact(binding, oracle):
    sc = S.sample_scene(160, 120, camera_width=6.6666667)
    p32, p8, ns, st = _gpu_render(binding, sc, 6)
    osc, opt, (r32, r8, rays) = _oracle_render(oracle, sc, 6)
    assert ns == 6
    assert np.array_equal(p32, r32)
    assert np.array_equal(p8, r8)
    assert st["rays"] == rays
    assert st["samples"] == 160 * 120 * 6


def test_unfused_pipeline_matches(binding, oracle):
    sc = S.sample_scene(96, 96)
    a32, a8, _, sa = _gpu_render(binding, sc, 3, fuse_trace_shade=0)
    b32, b8, _, sb = _gpu_render(binding, sc, 3, fuse_trace_shade=1)
    _, _, (r32, r8, rays) = _oracle_render(oracle, sc, 3)
    assert np.array_equal(a32, r32) and np.array_equal(b32, r32)
    assert np.array_equal(a8, r8) and np.array_equal(b8, r8)
    assert sa["rays"] == rays and sb["rays"] == rays


@pytest.mark.parametrize("tone", [S.TONE_CLAMP, S.TONE_REINHARD, S.TONE_EXPOSURE])
def test_all_materials_and_lights(binding, oracle, tone):
    sc = materials_scene(80, 60, tone=tone, exposure=0.7)
    p32, p8, ns, st = _gpu_render(binding, sc, 8)
    _, _, (r32, r8, rays) = _oracle_render(oracle, sc, 8)
    assert np.array_equal(p32, r32)
    assert np.array_equal(p8, r8)
    assert st["rays"] == rays
    assert st["shadow_rays"] > 0  # the upward-facing light and the sphere light do get sampled


def test_hdri_sky_exposure(binding, oracle):
    sc = hdri_scene()
    p32, p8, _, st = _gpu_render(binding, sc, 8)
    _, _, (r32, r8, rays) = _oracle_render(oracle, sc, 8)
    assert np.array_equal(p32, r32)
    assert np.array_equal(p8, r8)
    assert st["rays"] == rays


def test_hdri_not_loaded_is_white(binding, oracle):
    sc = hdri_scene()
    sc.sky = S.Sky.hdri("pending.hdr", None)
    p32, _, _, _ = _gpu_render(binding, sc, 2)
    _, _, (r32, _, _) = _oracle_render(oracle, sc, 2)
    assert np.array_equal(p32, r32)


def test_depth_16_diffuse_room(binding, oracle):
    sc = S.diffuse_room_scene(96, 54, n_spheres=12)
    p32, p8, _, st = _gpu_render(binding, sc, 4, max_depth=16)
    _, _, (r32, r8, rays) = _oracle_render(oracle, sc, 4, max_depth=16)
    assert np.array_equal(p32, r32)
    assert np.array_equal(p8, r8)
    assert st["rays"] == rays


@pytest.mark.parametrize("depth", [1, 2, 7])
def test_other_depths(binding, oracle, depth):
    sc = S.sample_scene(64, 64)
    p32, _, _, st = _gpu_render(binding, sc, 3, max_depth=depth)
    _, _, (r32, _, rays) = _oracle_render(oracle, sc, 3, max_depth=depth)
    assert np.array_equal(p32, r32)
    assert st["rays"] == rays


def test_empty_scene_and_no_lights(binding, oracle):
    sc = S.Scene()  # Scene::default: no primitives, white sky, 256x256
    sc.camera.tone_mapping, sc.camera.exposure = S.TONE_EXPOSURE, 1.0
    p32, p8, _, _ = _gpu_render(binding, sc, 2)
    _, _, (r32, r8, _) = _oracle_render(oracle, sc, 2)
    assert np.array_equal(p32, r32) and np.array_equal(p8, r8)
    sc2 = S.sample_scene(48, 48)
    sc2.emissive_planes.clear()  # no lights: the terminal sample is black instead of the reference's `% 0` panic
    p32, _, _, _ = _gpu_render(binding, sc2, 2)
    _, _, (r32, _, _) = _oracle_render(oracle, sc2, 2)
    assert np.array_equal(p32, r32)


def test_ragged_image_sizes(binding, oracle):
    for w, h in [(1, 1), (33, 17), (65, 31)]:  # not multiples of the 32x32 tile nor of the 64-lane wave
        sc = S.sample_scene(w, h, camera_width=5.0 * w / h)
        p32, p8, _, _ = _gpu_render(binding, sc, 3)
        _, _, (r32, r8, _) = _oracle_render(oracle, sc, 3)
        assert np.array_equal(p32, r32) and np.array_equal(p8, r8), (w, h)


# ---- host-visible state machine -----------------------------------------------------------------------------------
def test_batching_and_incremental_render_are_equivalent(binding):
    sc = S.sample_scene(64, 48, camera_width=6.6666667)
    a, _, _, _ = _gpu_render(binding, sc, 7, samples_per_batch=1)
    b, _, _, _ = _gpu_render(binding, sc, 7, samples_per_batch=4)
    with binding.PtGpu(samples_per_batch=3) as gpu:
        gpu.upload_scene(sc)
        for n in (1, 2, 4):
            gpu.render(n)
        c, ns = gpu.read_pixels32()
        assert ns == 7
        gpu.clear()
        z, ns0 = gpu.read_pixels32()
        assert ns0 == 0 and not z.any() and not gpu.read_pixels8().any()
        gpu.render(7)
        d, _ = gpu.read_pixels32()
    assert np.array_equal(a, b) and np.array_equal(a, c) and np.array_equal(a, d)


def test_sample_sharding_sums_to_single_gpu(binding):
    """Two shards (as two ranks would render them) add up to exactly the unsharded accumulator."""
    sc = S.sample_scene(72, 40, camera_width=9.0)
    full, _, _, _ = _gpu_render(binding, sc, 6)
    parts = []
    for r in range(2):
        p, _, ns, _ = _gpu_render(binding, sc, 3, shard_index=r, shard_count=2)
        assert ns == 3
        parts.append(p)
    assert np.array_equal(parts[0] + parts[1], full)


def test_export_import_accumulator(binding):
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    sc = S.sample_scene(40, 24)
    with binding.PtGpu() as g1, binding.PtGpu() as g2:
        g1.upload_scene(sc); g2.upload_scene(sc)
        g1.render(2)
        ref, _ = g1.read_pixels32()
        buf = C.c_void_p()
        assert hip.hipMalloc(C.byref(buf), 40 * 24 * 3 * 4) == 0
        g1.export_pixels32(buf.value)
        g2.import_pixels32(buf.value, 2)
        got, ns = g2.read_pixels32()
        assert ns == 2 and np.array_equal(got, ref)
        assert np.array_equal(g2.read_pixels8(), g1.read_pixels8())
        hip.hipFree(buf)


def test_errors_are_reported_not_thrown(binding):
    with binding.PtGpu() as gpu:
        with pytest.raises(binding.PtGpuError) as e:
            gpu.render(1)
        assert e.value.code == -4  # PT_ERR_NO_SCENE
    with pytest.raises(binding.PtGpuError) as e:
        binding.PtGpu(max_depth=0)
    assert e.value.code == -1
    with pytest.raises(binding.PtGpuError):
        binding.PtGpu(shard_index=2, shard_count=2)


def test_glass_queue_overflow_is_detected(binding):
    """A scene that is all glass doubles the queue at every level; with branch_capacity 1 the level after the
    first cannot hold it and the library must say so instead of dropping rays silently."""
    sc = S.Scene(image_settings=S.ImageSettings(128, 128, True),
                 camera=S.Camera((0.0, 0.0, 6.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 1.0, 1.0, 3.0, S.TONE_CLAMP, 1.0, 100.0),
                 sky=S.Sky.constant((1.0, 1.0, 1.0)))
    sc.add(S.Sphere((0.0, 0.0, 0.0), 2.0, S.Material.translucent(1.5)))
    with binding.PtGpu(branch_capacity=1, samples_per_batch=1) as gpu:
        gpu.upload_scene(sc)
        gpu.render(1)  # 16384 primary rays, all on glass -> 32768 children for a level sized for ~1.25 x 16384
        with pytest.raises(binding.PtGpuError) as e:
            gpu.read_pixels32()
        assert e.value.code == -5


def test_thin_lens_camera_bit_exact(binding, oracle):
    """CameraSampler::thin_lens_ray (reference src/tracer.rs:144-197; dead code there) as an optional camera model:
    primary-hit ids and accumulated pixels against the oracle, both pipelines."""
    sc = S.sample_scene(128, 96, camera_width=6.6666667)
    opt = oracle.options(camera_model=1, focus_distance=12.0, lens_aperture=8.0)
    osc = oracle.OracleScene(sc)
    r32, r8, rays = osc.render(opt, 5)
    for fused in (1, 0):
        with binding.PtGpu(camera_model=1, focus_distance=12.0, lens_aperture=8.0, fuse_trace_shade=fused) as gpu:
            gpu.upload_scene(sc)
            ids = gpu.primary_ids(3)
            gpu.render(5)
            p32, _ = gpu.read_pixels32()
            assert np.array_equal(ids, osc.primary_ids(opt, 3))
            assert np.array_equal(p32, r32) and gpu.stats()["rays"] == rays
            assert np.array_equal(gpu.read_pixels8(), r8)
    with binding.PtGpu() as gpu:  # and it is not the pinhole image
        gpu.upload_scene(sc)
        gpu.render(5)
        assert not np.array_equal(gpu.read_pixels32()[0], r32)


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("depth", [1, 2, 5])
def test_every_pipeline_variant_is_bit_exact(binding, oracle, mode, depth):
    """fuse_trace_shade 0 (trace + shade kernels), 1 (fused), 2 (fused, block regroup by material arm) and
    3 (trace-ahead: children traced by the parent's kernel, leaf children never queued) all evaluate the same
    recursion: same sums, same ray counts as the oracle, on the scene with every material, both light kinds and
    an open sky (so that leaves are common)."""
    sc = materials_scene(112, 80)
    p32, p8, _, st = _gpu_render(binding, sc, 4, fuse_trace_shade=mode, max_depth=depth)
    osc = oracle.OracleScene(sc)
    r32, r8, rays = osc.render(oracle.options(max_depth=depth), 4)
    assert np.array_equal(p32, r32) and np.array_equal(p8, r8)
    assert st["rays"] == rays
    if mode == 3 and depth > 1:
        assert st["records"] < rays  # leaf children were not queued


def _glass_scene(w, h):
    """Nested glass: every primary ray meets glass, and so do most of its children - the queue doubles per level."""
    sc = S.Scene(image_settings=S.ImageSettings(w, h, True),
                 camera=S.Camera((0.0, 0.0, 6.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 1.0, 1.0 * h / w, 3.0, S.TONE_REINHARD, 1.0, 100.0),
                 sky=S.Sky.constant((0.8, 0.9, 1.0)))
    sc.add(S.Sphere((0.0, 0.0, 0.0), 2.0, S.Material.translucent(1.5)))
    sc.add(S.Sphere((0.3, 0.1, 0.0), 1.0, S.Material.translucent(1.3)))
    sc.add(S.Sphere((-0.2, -0.1, 0.2), 0.4, S.Material.physically((0.9, 0.4, 0.2), 0.5, 0.0)))
    sc.add(S.Plane((-1.0, 2.5, -1.0), (2.0, 0.0, 0.0), (0.0, 0.0, 2.0), S.Material.emissive((6.0, 6.0, 6.0))))
    return sc


@pytest.mark.parametrize("pipes", [1, 2])
def test_levels_grow_and_voided_passes_are_rendered_again(binding, oracle, pipes):
    """With branch_capacity left at 0 the levels start small (one record per sample and level).  Passes that overflow add nothing,
    the levels are re-allocated larger and exactly those sample ranges are rendered again: the sums, the 8-bit
    image and the ray count are the oracle's, whichever passes had to be repeated and in whatever order."""
    sc = _glass_scene(96, 80)
    depth = 6
    with binding.PtGpu(max_depth=depth, samples_per_batch=2, concurrent_passes=pipes) as gpu:
        gpu.upload_scene(sc)
        before = gpu.stats()
        assert before["queue_growth"] == 1
        gpu.render(3)
        gpu.render(4)
        p32, n = gpu.read_pixels32()
        p8 = gpu.read_pixels8()
        st = gpu.stats()
        assert n == 7 and st["queue_growth"] > 2 and st["device_bytes"] > before["device_bytes"]
        gpu.render(2)  # and it keeps working with the grown levels
        q32, m = gpu.read_pixels32()
    osc = oracle.OracleScene(sc)
    r32, r8, rays = osc.render(oracle.options(max_depth=depth), 7)
    assert np.array_equal(p32, r32) and np.array_equal(p8, r8) and st["rays"] == rays
    assert m == 9 and np.array_equal(q32, osc.render(oracle.options(max_depth=depth), 9)[0])


def test_clear_discards_voided_passes(binding, oracle):
    sc = _glass_scene(64, 64)
    with binding.PtGpu(max_depth=6, samples_per_batch=1) as gpu:
        gpu.upload_scene(sc)
        gpu.render(2)   # overflows at growth 2 (not looked at yet)
        gpu.clear()     # ... and is thrown away
        gpu.render(3)
        p32, n = gpu.read_pixels32()
    assert n == 3
    assert np.array_equal(p32, oracle.OracleScene(sc).render(oracle.options(max_depth=6), 3)[0])


def test_more_passes_than_flag_slots_between_synchronisations(binding, oracle):
    """1100 one-sample passes in a single render call: the ring of per-pass overflow flags (1024) is read and
    reused on the way."""
    sc = _glass_scene(32, 24)
    with binding.PtGpu(max_depth=5, samples_per_batch=1, concurrent_passes=2) as gpu:
        gpu.upload_scene(sc)
        gpu.render(1100)
        p32, n = gpu.read_pixels32()
        st = gpu.stats()
    r32, _, rays = oracle.OracleScene(sc).render(oracle.options(max_depth=5), 1100, threads=8)
    assert n == 1100 and np.array_equal(p32, r32) and st["rays"] == rays


# ---- axis-aligned rectangles take a shorter plane test on the device ------------------------------------------------
def _axis_box_scene(width=96, height=72):
    """Six inward walls whose normals run along +-x, +-y, +-z, with the edge that runs along the 'next' axis given first
    on some walls and second on others and with negative edge directions; one tilted rectangle (general test) between
    them; a wall lying exactly in the camera's projection plane, so that the numerator of its plane equation is a sum
    of zeros for every primary ray; a glass sphere (total internal reflection makes NaN rays, which must take the
    general test) and a light."""
    grey = S.Material.physically((0.8, 0.8, 0.8), 0.9, 0.0)
    red = S.Material.physically((0.9, 0.2, 0.2), 0.5, 0.0)
    gold = S.Material.physically((1.0, 0.8, 0.4), 0.2, 1.0)
    sc = S.Scene(
        image_settings=S.ImageSettings(width, height, True),
        camera=S.Camera((0.0, 5.0, 15.0), (0.0, 5.0, 0.0), (0.0, 1.0, 0.0), 8.0, 6.0, 10.0, S.TONE_CLAMP, 1.0, 200.0),
        sky=S.Sky.constant((0.2, 0.2, 0.2)),
    )
    sc.add(S.Plane((5.0, 0.0, 5.0), (0.0, 0.0, -10.0), (-10.0, 0.0, 0.0), grey))     # floor, n = +y, negative edges
    sc.add(S.Plane((-5.0, 10.0, -5.0), (10.0, 0.0, 0.0), (0.0, 0.0, 10.0), grey))    # ceiling, n = -y, edges the other way round
    sc.add(S.Plane((-5.0, 0.0, -5.0), (10.0, 0.0, 0.0), (0.0, 10.0, 0.0), gold))     # back wall, n = +z
    sc.add(S.Plane((-5.0, 0.0, -5.0), (0.0, 10.0, 0.0), (0.0, 0.0, 10.0), red))      # left wall, n = +x
    sc.add(S.Plane((5.0, 0.0, -5.0), (0.0, 0.0, 10.0), (0.0, 10.0, 0.0), grey))      # right wall, n = -x, edges the other way round
    sc.add(S.Plane((2.0, 1.0, 5.0), (2.0, 0.0, 0.0), (0.0, 2.0, 0.0), gold))         # in the projection plane z = 5, n = +z
    sc.add(S.Plane((5.0, 0.0, 16.0), (-10.0, 0.0, 0.0), (0.0, 10.0, 0.0), grey))     # behind the camera, n = -z
    sc.add(S.Plane((-1.0, 0.5, 1.0), (3.0, 0.5, 0.0), (0.0, 0.4, -3.0), red))        # tilted: general test
    sc.add(S.Plane((-2.0, 9.9, -2.0), (4.0, 0.0, 0.0), (0.0, 0.0, 4.0), S.Material.emissive((12.0, 12.0, 12.0))))
    sc.add(S.Sphere((2.0, 2.0, 0.0), 2.0, S.Material.translucent(1.5)))
    sc.add(S.Sphere((-2.5, 1.5, 2.0), 1.5, gold))
    return sc


def test_axis_aligned_rectangles_short_test(binding, oracle):
    sc = _axis_box_scene()
    p32, p8, ns, st = _gpu_render(binding, sc, 5, max_depth=5)
    osc, opt, (r32, r8, rays) = _oracle_render(oracle, sc, 5, max_depth=5)
    assert np.array_equal(p32, r32) and np.array_equal(p8, r8) and st["rays"] == rays
    for jitter in (0, 1):
        with binding.PtGpu(sub_pixel_override=jitter) as gpu:
            gpu.upload_scene(sc)
            ids = gpu.primary_ids(1)
        assert np.array_equal(ids, osc.primary_ids(oracle.options(sub_pixel_override=jitter), 1))


# ---- primary rays that cannot reach the scene's bounding box skip the scan ------------------------------------------
@pytest.mark.parametrize("jitter", [0, 1])
def test_silhouettes_on_the_faces_of_the_scene_box(binding, oracle, jitter):
    """Open scenes whose outline IS their bounding box: a lone sphere (every silhouette ray grazes a face of the box), a
    rectangle seen edge-on through the image centre, and a camera looking along an axis, so that with jitter off a whole
    image column and row have an exactly zero direction component.  Most of the image sees the sky."""
    gold = S.Material.physically((1.0, 0.8, 0.4), 0.2, 1.0)
    grey = S.Material.physically((0.8, 0.8, 0.8), 0.9, 0.0)
    for prims in ([S.Sphere((0.0, 0.0, 0.0), 1.0, gold)],
                  [S.Plane((-1.0, 0.0, -1.0), (2.0, 0.0, 0.0), (0.0, 0.0, 2.0), grey)],   # y = 0: edge-on for the centre row
                  [S.Sphere((-2.0, 0.5, 0.0), 0.5, S.Material.translucent(1.5)), S.Sphere((1.5, -0.25, 1.0), 0.75, gold),
                   S.Plane((-3.0, -1.0, -1.0), (6.0, 0.0, 0.0), (0.0, 0.0, 3.0), grey),
                   S.Plane((-0.5, 2.0, -0.5), (1.0, 0.0, 0.0), (0.0, 0.0, 1.0), S.Material.emissive((9.0, 9.0, 9.0)))]):
        sc = S.Scene(
            image_settings=S.ImageSettings(129, 97, bool(jitter)),
            camera=S.Camera((0.0, 0.0, 8.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 8.0, 6.0, 4.0, S.TONE_CLAMP, 1.0, 100.0),
            sky=S.Sky.constant((0.3, 0.4, 0.5)),
        )
        for p in prims:
            sc.add(p)
        p32, p8, ns, st = _gpu_render(binding, sc, 6)
        _, _, (r32, r8, rays) = _oracle_render(oracle, sc, 6)
        assert np.array_equal(p32, r32) and np.array_equal(p8, r8) and st["rays"] == rays, len(prims)


# ---- linear scan over many spheres: the full tests run lane by lane on each lane's own candidates -------------------
@pytest.mark.parametrize("n_spheres", [9, 33, 80])
def test_many_spheres_linear_scan(binding, oracle, n_spheres):
    """More than eight spheres switch the fused kernels to the deferred sphere tests (candidates kept in a 32-bit mask per
    lane, so 33 and 80 spheres also cross a chunk boundary); with at most 100 primitives the scene stays on the linear scan.
    Overlapping spheres give lanes several candidates, glass ones rays that start inside a sphere."""
    sc = S.diffuse_room_scene(96, 54, n_spheres=n_spheres, seed=n_spheres)
    for k, sp in enumerate(sc.spheres):
        if k % 3 == 0:
            sp.radius *= 2.5                               # overlaps
        if k % 4 == 1:
            sp.material = S.Material.translucent(1.5)
    p32, p8, ns, st = _gpu_render(binding, sc, 3, max_depth=6)
    assert st["bvh_nodes"] == 0
    _, _, (r32, r8, rays) = _oracle_render(oracle, sc, 3, max_depth=6)
    assert np.array_equal(p32, r32) and np.array_equal(p8, r8) and st["rays"] == rays


def test_present_gives_the_display_bytes(binding):
    """The window loop's per-frame readback (Backbuffer.pixels8, src/window.rs:205-210) through the library's own
    page-locked buffer: the same bytes as ptgpu_read_pixels8, frame after frame and after a new scene of another size."""
    with binding.PtGpu() as gpu:
        for w, h_ in ((96, 64), (40, 56)):
            gpu.upload_scene(S.sample_scene(w, h_, camera_width=5.0 * w / h_))
            for _ in range(3):
                gpu.render(1)
                shown = gpu.present_pixels8().copy()
                assert shown.shape == (h_, w, 3) and np.array_equal(shown, gpu.read_pixels8())


def test_basic_operations_match_the_host(binding, oracle):
    """f32 square root and division compile to instruction sequences on the device (correctly rounded by default) and
    the u32 -> (0,1] mapping of random32 is a division by 2^32: random bit patterns of every class against the host's
    IEEE results, bit for bit (tests/tools/math_exhaustive.py runs the same over all 2^32 arguments)."""
    rng = np.random.RandomState(9)
    a = rng.randint(0, 1 << 32, 1 << 20, dtype=np.uint64).astype(np.uint32)
    b = rng.randint(0, 1 << 32, 1 << 20, dtype=np.uint64).astype(np.uint32)
    near = (b & np.uint32(0x807FFFFF)) | (a & np.uint32(0x7F800000))   # same exponent as the dividend
    fa = a.view(np.float32)
    with binding.PtGpu() as gpu:
        for fn, x, y in ((6, fa, None), (8, fa, None), (7, fa, b.view(np.float32)), (7, fa, near.view(np.float32))):
            got = gpu.eval_math(fn, x, y)
            ref = oracle.math_eval(oracle.MATH_DET, fn, x, y)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), fn
        # The device's square root has a short path for waves without an argument below 2^-96 (random bit patterns never
        # take it: half of them are negative): whole blocks of arguments from 2^-96 up, infinities and NaNs included, next
        # to blocks that hold a tiny, zero or negative argument among them.
        big = (rng.randint(0x0F800000, 0x7FC00001, 1 << 20, dtype=np.int64).astype(np.uint32))
        mixed = big.copy()
        mixed[::257] = a[::257]
        for x in (big, mixed):
            got = gpu.eval_math(6, x.view(np.float32), None)
            ref = oracle.math_eval(oracle.MATH_DET, 6, x.view(np.float32), None)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_normalize_with_a_shared_reciprocal_matches_three_divisions(binding):
    """normalize is three correctly rounded divisions by the same length; waves whose components are all of ordinary size
    compute the refined reciprocal once (pt_math.h).  Against numpy's IEEE float32 arithmetic, bit for bit: vectors of
    ordinary size (whole waves on the shared path), vectors with tiny, zero, huge, infinite and NaN components (the
    divisions as written), and blocks that mix the two."""
    rng = np.random.RandomState(21)
    n = 1 << 21

    def reference(x, y):
        with np.errstate(all="ignore"):
            z = (x * np.float32(0.37) + y).astype(np.float32)
            length = np.sqrt(((x * x + y * y) + z * z).astype(np.float32)).astype(np.float32)
            return [(c / length).astype(np.float32) for c in (x, y, z)]

    ordinary = lambda: (rng.uniform(-1, 1, n) * 10.0 ** rng.uniform(-12, 12, n)).astype(np.float32)
    bits = lambda: rng.randint(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    special = np.array([0.0, -0.0, 1e-45, -1e-39, 1e-31, 3e38, -3e38, np.inf, -np.inf, np.nan, 1.0, -2.5], dtype=np.float32)
    cases = [(ordinary(), ordinary()), (bits(), bits())]
    x, y = ordinary(), ordinary()
    x[::193] = special[rng.randint(0, len(special), len(x[::193]))]
    y[::389] = special[rng.randint(0, len(special), len(y[::389]))]
    cases.append((x, y))
    x, y = ordinary(), ordinary()                                        # zeros of both signs stay on the shared path
    x[::3] = np.where(rng.randint(0, 2, len(x[::3])) == 1, np.float32(0.0), np.float32(-0.0))
    y[::7] = np.where(rng.randint(0, 2, len(y[::7])) == 1, np.float32(0.0), np.float32(-0.0))
    cases.append((x, y))
    e = rng.randint(1, 255, n).astype(np.uint32) << 23                   # every exponent, components of like magnitude
    cases.append(((e | rng.randint(0, 1 << 23, n).astype(np.uint32)).view(np.float32),
                  (e | rng.randint(0, 1 << 23, n).astype(np.uint32) | np.uint32(0x80000000)).view(np.float32)))
    with binding.PtGpu() as gpu:
        for x, y in cases:
            want = reference(x, y)
            for fn, w in zip((9, 10, 11), want):
                got = gpu.eval_math(fn, x, y)
                same = (got.view(np.uint32) == w.view(np.uint32)) | (np.isnan(got) & np.isnan(w))
                assert same.all(), (fn, int((~same).sum()), x[~same][:4], y[~same][:4], got[~same][:4], w[~same][:4])

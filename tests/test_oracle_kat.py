"""Known-answer tests that pin the CPU oracle.

The reference ships no tests or golden vectors, so these values were derived BY HAND from its source
(file:line given per case) and from IEEE-754 arithmetic; they pin the oracle's reading of that source.
"""
import ctypes as C
import math

import numpy as np

from pathtracer_amd import scene as S


def f32(x):
    return float(np.float32(x))


def test_k5_xorshift32_stream_and_random32(oracle):
    # reference src/common.rs:28-41, seed 314159265
    L = oracle.lib()
    st = C.c_uint32(314159265)
    got = [L.ptref_xorshift32_next(C.byref(st)) for _ in range(6)]
    assert got == [2971524119, 1501041240, 1028966369, 280892309, 2337768595, 58652497]
    # r as f32 / u32::MAX as f32; u32::MAX as f32 == 2^32
    assert L.ptref_random32_from_u32(2971524119) == f32(0.69186187)
    assert L.ptref_random32_from_u32(1501041240) == f32(0.3494884)
    assert L.ptref_random32_from_u32(0xFFFFFFFF) == 1.0  # can return exactly 1.0 ...
    assert L.ptref_random32_from_u32(1) > 0.0            # ... never 0 (xorshift never yields 0)


def test_counter_rng_never_zero_and_is_a_pure_function(oracle):
    L = oracle.lib()
    a = L.ptref_counter_u32(1, 2, 3, 4, 5)
    assert a == L.ptref_counter_u32(1, 2, 3, 4, 5) and a != 0
    assert len({L.ptref_counter_u32(1, 2, 3, 4, d) for d in range(64)}) == 64
    vals = np.array([L.ptref_counter_u32(7, p, 0, 1, 0) for p in range(20000)], dtype=np.uint64)
    assert vals.min() > 0
    assert abs(vals.mean() / 2**32 - 0.5) < 0.01


def test_k1_camera_plane_of_sample_scene(oracle):
    # reference src/tracer.rs:81-96 with sample.scene:18-27
    cam = oracle.camera_new(S.sample_scene().camera)
    assert cam.plane_origin.t() == (-2.5, 2.5, 10.0)
    assert cam.plane_u.t() == (5.0, 0.0, 0.0)
    assert cam.plane_v.t() == (0.0, 5.0, 0.0)


def test_k4_plane_normals_of_sample_scene(oracle):
    # reference src/scene.rs:106; ids 3..8 then the light (id 9)
    osc = oracle.OracleScene(S.sample_scene())
    want = [(0, -1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 0), (-1, 0, 0), (0, 0, -1)]
    assert [tuple(map(float, n)) for n in osc.arrays[1]["normal"]] == [tuple(map(float, w)) for w in want]
    assert tuple(map(float, osc.arrays[3]["normal"][0])) == (0.0, -1.0, 0.0)


def test_k2_centre_primary_ray(oracle):
    # reference src/tracer.rs:125-142, src/scene.rs:263-297
    osc = oracle.OracleScene(S.sample_scene())
    o, d = osc.pinhole_ray(512, 512)
    assert o == (0.0, 5.0, 10.0) and d == (0.0, 0.0, -1.0)
    found, h = osc.find_scene_hit(o, d)
    # the front wall (z=5, id 8) is crossed first at t=5 but from behind (Transition::Out) -> discarded;
    # ceiling/floor/side walls are parallel (denum 0); spheres are missed -> back wall, exactly t=15
    assert found and h.prim_id == 5 and h.t == 15.0 and h.transition == 0


def test_k3_ray_into_the_light(oracle):
    osc = oracle.OracleScene(S.sample_scene())
    found, h = osc.find_scene_hit((0.0, 5.0, 0.0), (0.0, 1.0, 0.0))
    assert found and h.prim_id == 9
    assert h.t == f32(f32(9.9) - 5.0)  # 4.8999996
    # radiance (10,10,10) -> clamp -> 1.0 -> gamma -> 255; pixel8 = min(255, 2.0*255/1)
    opt = oracle.options(oracle.RNG_XORSHIFT, oracle.MATH_LIBM)
    rad, rays, _ = osc.trace_radiance(opt, (0.0, 5.0, 0.0), (0.0, 1.0, 0.0))
    assert rad == (10.0, 10.0, 10.0) and rays == 1
    g = oracle.lib().ptref_tone_map_gamma(C.byref(osc.desc), C.byref(opt), oracle.Vec3(*rad))
    q = (C.c_uint32 * 3)()
    oracle.lib().ptref_pixel32_from_unit(g, C.byref(q))
    assert list(q) == [255, 255, 255]
    one = oracle.OracleScene(S.sample_scene(1, 1))
    assert one.assign_pixels8(np.array([[[255, 255, 255]]], dtype=np.uint32), 1).tolist() == [[[255, 255, 255]]]


def test_k6_terminal_light_sample_is_black_in_sample_scene(oracle):
    # reference src/scene.rs:117: the light's normal points down, so every point below it fails the facing test
    # BEFORE any random draw; no shadow ray is traced and the xorshift stream advances by exactly one
    # (the light index, src/tracer.rs:454)
    osc = oracle.OracleScene(S.sample_scene())
    opt = oracle.options(oracle.RNG_XORSHIFT, oracle.MATH_LIBM)
    rad, rays, state = osc.trace_radiance(opt, (1.0, 3.0, 1.0), (0.0, 1.0, 0.0), depth=0, xorshift_state=314159265)
    assert rad == (0.0, 0.0, 0.0) and rays == 0 and state == 2971524119


def test_k7_nan_ray_misses_everything(oracle):
    # reference src/scene.rs:169-261: every sphere test fails on NaN; every plane comes out Transition::Out
    osc = oracle.OracleScene(S.sample_scene())
    nan = float("nan")
    found, h = osc.find_scene_hit((0.0, 5.0, 0.0), (nan, nan, nan))
    assert not found and h.prim_id == oracle.MISS
    rad, _, _ = osc.trace_radiance(oracle.options(), (0.0, 5.0, 0.0), (nan, nan, nan))
    assert rad == (f32(0.1),) * 3  # the sky


def test_plane_is_one_sided_and_finite(oracle):
    osc = oracle.OracleScene(S.sample_scene())
    # from outside the box, looking in through the left wall's back face: the wall is ignored, the right wall is hit
    found, h = osc.find_scene_hit((-8.0, 5.0, 0.0), (1.0, 0.0, 0.0))
    assert found and h.prim_id == 1  # ... here the glossy sphere (centre (-2.2,6.2,-0.5), r=2) is the first thing behind it
    found, h = osc.find_scene_hit((-8.0, 9.5, 4.0), (1.0, 0.0, 0.0))
    assert found and h.prim_id == 7 and h.t == 13.0  # clear path: the right wall, seen from its front
    found, h = osc.find_scene_hit((0.0, 5.0, 0.0), (0.0, 0.0, 1.0))
    assert found and h.prim_id == 8 and h.t == 5.0  # front wall seen from inside
    # beyond the rectangle's extent the plane does not exist
    found, h = osc.find_scene_hit((0.0, 20.0, 0.0), (0.0, 0.0, -1.0))
    assert not found


def test_sphere_transition_out_from_inside(oracle):
    osc = oracle.OracleScene(S.sample_scene())
    found, h = osc.find_scene_hit((2.0, 2.0, 0.0), (0.0, 0.0, 1.0))  # centre of the glass sphere
    assert found and h.prim_id == 0 and h.transition == 1 and h.t == 2.0
    assert h.normal.t() == (0.0, 0.0, -1.0)  # normal flipped towards the inside (src/scene.rs:216)


def test_saturating_casts_and_pixels8(oracle):
    q = (C.c_uint32 * 3)()
    L = oracle.lib()
    L.ptref_pixel32_from_unit(oracle.Vec3(float("nan"), -1.0, float("inf")), C.byref(q))
    assert list(q) == [0, 0, 0xFFFFFFFF]  # Rust `as u32`: NaN/negative -> 0, too large -> u32::MAX
    L.ptref_pixel32_from_unit(oracle.Vec3(0.999999, 0.5, 1.0), C.byref(q))
    assert list(q) == [254, 127, 255]     # truncation, not rounding (src/common.rs:80)
    osc = oracle.OracleScene(S.sample_scene(2, 1))  # iso 200 -> factor 2
    p8 = osc.assign_pixels8(np.array([[[100, 300, 0], [127, 128, 255]]], dtype=np.uint32), 2)
    assert p8.tolist() == [[[100, 255, 0], [127, 128, 255]]]


def test_det_math_agrees_with_libm_within_one_ulp(oracle):
    rng = np.random.RandomState(0)
    cases = [(oracle.FN_SIN, rng.uniform(0, 6.3, 100000), None), (oracle.FN_COS, rng.uniform(0, 6.3, 100000), None),
             (oracle.FN_EXP, rng.uniform(-30, 10, 100000), None), (oracle.FN_ACOS, rng.uniform(-1, 1, 100000), None),
             (oracle.FN_POW, rng.uniform(0, 1.5, 100000), 5.0), (oracle.FN_POW, rng.uniform(0, 1.0, 100000), 1.0 / 2.2),
             (oracle.FN_ATAN2, rng.uniform(-2, 2, 100000), rng.uniform(-2, 2, 100000))]
    for fn, a, b in cases:
        a = a.astype(np.float32)
        bb = None if b is None else np.broadcast_to(np.asarray(b, dtype=np.float32), a.shape)
        d = oracle.math_eval(oracle.MATH_DET, fn, a, bb)
        l = oracle.math_eval(oracle.MATH_LIBM, fn, a, bb)
        ulp = np.spacing(np.maximum(np.abs(l), np.float32(1e-30)))
        assert np.all(np.abs(d.astype(np.float64) - l.astype(np.float64)) <= 1.5 * ulp), fn


def test_det_pow_special_cases(oracle):
    P = lambda x, y: float(oracle.math_eval(oracle.MATH_DET, oracle.FN_POW, np.array([x], np.float32), np.array([y], np.float32))[0])
    assert P(0.0, 0.45) == 0.0 and P(1.0, 0.45) == 1.0 and P(2.0, 3.0) == 8.0 and P(0.5, 5.0) == 0.03125
    assert math.isnan(P(-0.5, 0.45)) and math.isnan(P(float("nan"), 5.0))
    assert P(-0.5, 5.0) == -0.03125                    # negative base, odd integer exponent (1 - cos_theta can be < 0)
    assert P(float("inf"), 0.45) == float("inf") and P(-float("inf"), 0.45) == float("inf")
    assert P(-0.0, 0.45) == 0.0 and P(7.0, 0.0) == 1.0


def test_thin_lens_ray_follows_the_reference_formula(oracle):
    # reference src/tracer.rs:144-197, redone here step by step in numpy float32
    sc = S.sample_scene()
    osc = oracle.OracleScene(sc)
    o, d = osc.pinhole_ray(300, 700, jitter=(0.25, 0.75))
    F = np.float32
    o32, d32 = np.array(o, dtype=F), np.array(d, dtype=F)
    cam = osc.desc.camera
    su = np.array(cam.plane_u.t(), dtype=F) / F(sc.image_settings.width)
    sv = np.array(cam.plane_v.t(), dtype=F) / F(sc.image_settings.height)
    for r1, r2 in ((0.5, 1.0), (0.125, 0.3), (1.0, 0.01)):
        focus = o32 + F(10.0) * d32
        coc = F(22.0) / F(4.0)
        alpha = F(2.0) * F(np.pi) * F(r1)
        r = coc * F(r2)
        l_u = r * F(math.sin(float(alpha)))
        l_v = r * F(math.cos(float(alpha)))
        lens_o = (o32 + l_u * su) + l_v * sv
        diff = focus - lens_o
        length = np.sqrt(F((diff[0] * diff[0] + diff[1] * diff[1]) + diff[2] * diff[2]))
        lens_d = diff / length
        for mode in (oracle.MATH_LIBM, oracle.MATH_DET):
            oo, dd = osc.thin_lens_ray(oracle.options(math_mode=mode, camera_model=1), o, d, r1, r2)
            # sin/cos differ from the double-rounded value by at most an ulp; everything else is exact f32 arithmetic
            assert np.allclose(oo, lens_o, rtol=0, atol=2e-6) and np.allclose(dd, lens_d, rtol=0, atol=2e-6)
    # r2 -> 0 collapses the lens to the pinhole origin: the ray is the pinhole ray again
    oo, dd = osc.thin_lens_ray(oracle.options(camera_model=1), o, d, 0.3, 0.0)
    assert oo == o and np.allclose(dd, d, atol=1e-7)


def test_thin_lens_draws_follow_the_pinhole_draws(oracle):
    # xorshift stream: u, v jitter first, then r1, r2 (src/tracer.rs:132-133,183-184); with the counter RNG they
    # are dimensions 2 and 3 of the camera node.  Same scene, same seed: the pinhole image differs from the
    # thin-lens image, and both are reproducible.
    sc = S.sample_scene(64, 48, camera_width=6.6666667)
    osc = oracle.OracleScene(sc)
    a = osc.primary_ids(oracle.options(camera_model=0), 0)
    b = osc.primary_ids(oracle.options(camera_model=1), 0)
    c = osc.primary_ids(oracle.options(camera_model=1), 0)
    assert np.array_equal(b, c) and not np.array_equal(a, b)


def test_hoisted_plane_axes_is_a_timing_variant_only(oracle):
    """The CPU baseline's second figure evaluates normalize(u), length(u), normalize(v), length(v) of each plane once per
    pass instead of per ray and plane (scene.rs:244-253): the same operations on the same inputs, so every accumulator
    and the ray count must come out identical, serial and through the worker pool."""
    R = oracle
    sc = R.OracleScene(S.sample_scene(48, 40, camera_width=6.0))
    for rng, math, threads in ((R.RNG_XORSHIFT, R.MATH_LIBM, 0), (R.RNG_COUNTER, R.MATH_DET, 3)):
        a = sc.render(R.options(rng, math), 3, threads=threads)
        b = sc.render(R.options(rng, math, hoist_plane_axes=1), 3, threads=threads)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]

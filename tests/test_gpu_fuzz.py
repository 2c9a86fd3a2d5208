"""Randomised parity: small random scenes, every one rendered by the linear-scan kernels, by the BVH kernels and by the
oracle; all three must agree bit for bit (ids, accumulators, display bytes, ray counts).

The generator leans on the cases the device code treats specially: axis-aligned rectangles in every orientation, edge
order and edge sign (the short plane test), rectangles through the camera's projection plane and through other
primitives' surfaces (zero numerators, equal distances), tilted rectangles, glass inside glass, lights of both kinds,
all three tone maps, constant and textured sky, jitter on and off, depths 1..6, the thin-lens camera, a sphere around the
camera, degenerate primitives; the linear scan runs in each of its pipeline variants in turn.
"""
import numpy as np
import pytest

from pathtracer_amd import scene as S

pytestmark = pytest.mark.gpu


def _material(rng):
    k = rng.randint(0, 10)
    if k == 0:
        return S.Material.mirror()
    if k == 1:
        return S.Material.none()
    if k <= 3:
        return S.Material.translucent(float(rng.choice([1.1, 1.33, 1.5, 2.4])))
    if k == 4:
        return S.Material.emissive(tuple(float(x) for x in rng.uniform(0.5, 12.0, 3)))
    return S.Material.physically(tuple(float(x) for x in rng.uniform(0.05, 1.0, 3)), float(rng.choice([0.0, 0.2, 0.5, 0.8, 1.0])),
                                 float(rng.choice([0.0, 0.0, 1.0, 0.5])))


def _axis_rectangle(rng, lo, hi):
    """A rectangle normal to a random axis at a grid coordinate, edges along the two other axes in random order and
    direction (grid coordinates make coplanar and touching rectangles likely)."""
    a = rng.randint(0, 3)
    b, c = [(1, 2), (2, 0), (0, 1)][a] if rng.rand() < 0.5 else [(2, 1), (0, 2), (1, 0)][a]
    origin = np.array([float(rng.randint(lo, hi + 1)) for _ in range(3)])
    u = np.zeros(3)
    v = np.zeros(3)
    u[b] = float(rng.randint(1, hi - lo + 1)) * (1.0 if rng.rand() < 0.5 else -1.0)
    v[c] = float(rng.randint(1, hi - lo + 1)) * (1.0 if rng.rand() < 0.5 else -1.0)
    return tuple(origin), tuple(u), tuple(v)


def random_scene(seed, width=48, height=36):
    rng = np.random.RandomState(1000 + seed)
    tone = int(rng.choice([S.TONE_CLAMP, S.TONE_REINHARD, S.TONE_EXPOSURE]))
    z_near = float(rng.choice([2.0, 5.0, 10.0]))
    cam_z = float(rng.choice([10.0, 12.0, 15.0]))
    sc = S.Scene(
        image_settings=S.ImageSettings(width, height, bool(rng.rand() < 0.7)),
        camera=S.Camera((float(rng.randint(-2, 3)), float(rng.randint(2, 7)), cam_z), (0.0, float(rng.randint(2, 6)), 0.0), (0.0, 1.0, 0.0),
                        8.0, 6.0, z_near, tone, float(rng.uniform(0.3, 2.0)), float(rng.choice([100.0, 200.0, 400.0]))),
        sky=S.Sky.constant(tuple(float(x) for x in rng.uniform(0.0, 1.0, 3))),
    )
    grey = S.Material.physically((0.8, 0.8, 0.8), float(rng.choice([0.3, 0.9])), 0.0)
    if rng.rand() < 0.5:   # floor and back wall, edges in either order
        sc.add(S.Plane((-8.0, 0.0, 8.0), (16.0, 0.0, 0.0), (0.0, 0.0, -16.0), grey))
        sc.add(S.Plane((-8.0, 0.0, -6.0), (16.0, 0.0, 0.0), (0.0, 12.0, 0.0), _material(rng)))
    else:
        sc.add(S.Plane((-8.0, 0.0, -8.0), (0.0, 0.0, 16.0), (16.0, 0.0, 0.0), grey))
        sc.add(S.Plane((8.0, 12.0, -6.0), (-16.0, 0.0, 0.0), (0.0, -12.0, 0.0), _material(rng)))
    if rng.rand() < 0.5:   # close the box, so that paths live to the depth limit
        sc.add(S.Plane((-8.0, 12.0, -6.0), (16.0, 0.0, 0.0), (0.0, 0.0, 22.0), _material(rng)))   # ceiling, faces down
        sc.add(S.Plane((-8.0, 0.0, -6.0), (0.0, 12.0, 0.0), (0.0, 0.0, 22.0), _material(rng)))    # left, faces +x
        sc.add(S.Plane((8.0, 0.0, 16.0), (0.0, 12.0, 0.0), (0.0, 0.0, -22.0), _material(rng)))    # right, faces -x
        sc.add(S.Plane((8.0, 0.0, 16.0), (-16.0, 0.0, 0.0), (0.0, 12.0, 0.0), grey))              # behind the camera, faces -z
    if rng.rand() < 0.6:   # a light overhead, facing down
        sc.add(S.Plane((-2.0, 9.5, -2.0), (4.0, 0.0, 0.0), (0.0, 0.0, 4.0), S.Material.emissive((10.0, 9.0, 8.0))))
    for _ in range(rng.randint(0, 7)):
        sc.add(S.Sphere(tuple(float(x) for x in rng.uniform(-4.0, 4.0, 3) + np.array([0.0, 4.0, 0.0])), float(rng.uniform(0.4, 2.5)), _material(rng)))
    for _ in range(rng.randint(1, 10)):
        if rng.rand() < 0.7:
            o, u, v = _axis_rectangle(rng, -5, 5)
            o = (o[0], o[1] + 5.0, o[2])
        else:
            o = tuple(float(x) for x in rng.uniform(-5.0, 5.0, 3) + np.array([0.0, 5.0, 0.0]))
            u = tuple(float(x) for x in rng.uniform(-6.0, 6.0, 3))
            w = rng.uniform(-6.0, 6.0, 3)
            v = tuple(float(x) for x in np.cross(np.array(u), w) / 6.0)   # perpendicular to u
        sc.add(S.Plane(o, u, v, _material(rng)))
    if rng.rand() < 0.5:   # a rectangle in the projection plane: every primary ray starts on it
        zc = cam_z - z_near
        sc.add(S.Plane((float(rng.randint(-4, 1)), float(rng.randint(1, 4)), zc), (float(rng.randint(1, 5)), 0.0, 0.0), (0.0, float(rng.randint(1, 4)), 0.0),
                       _material(rng)))
    for _ in range(rng.randint(0, 3)):   # exact copies under another material: equal distances, the earlier id must win
        if rng.rand() < 0.5 and sc.spheres:
            src = sc.spheres[rng.randint(0, len(sc.spheres))]
            sc.add(S.Sphere(src.origin, src.radius, _material(rng)))
        elif sc.planes:
            src = sc.planes[rng.randint(0, len(sc.planes))]
            sc.add(S.Plane(src.origin, src.u, src.v, _material(rng)))
    if rng.rand() < 0.15:  # the camera sits inside this one: primary rays leave it (Transition::Out)
        sc.add(S.Sphere((0.0, 5.0, cam_z - 1.0), float(rng.uniform(12.0, 30.0)), _material(rng)))
    if rng.rand() < 0.1:   # degenerate primitives: a point sphere, a rectangle without area (its normal is NaN)
        sc.add(S.Sphere((1.0, 3.0, 0.0), 0.0, _material(rng)))
        sc.add(S.Plane((0.0, 4.0, 0.0), (0.0, 0.0, 0.0), (0.0, 2.0, 0.0), _material(rng)))
    if rng.rand() < 0.25:
        tex = rng.uniform(0.0, 4.0, (8, 16, 3)).astype(np.float32)
        sc.sky = S.Sky.hdri("random.hdr", tex)
    if rng.rand() < 0.25:  # a crowd of small spheres, some overlapping: more than eight spheres send the linear scan's full sphere
        #                    tests through its per-lane candidate masks (more than 32: across the masks' chunk boundary)
        for _ in range(int(rng.choice([6, 12, 30, 45]))):
            sc.add(S.Sphere(tuple(float(x) for x in rng.uniform(-5.0, 5.0, 3) + np.array([0.0, 5.0, 0.0])), float(rng.uniform(0.1, 1.2)), _material(rng)))
    return sc, int(rng.randint(1, 7)), int(rng.randint(0, 2))


@pytest.mark.parametrize("seed", range(384))
def test_random_scene_three_ways(binding, oracle, seed):
    sc, depth, thin = random_scene(seed)
    spp = 3
    osc = oracle.OracleScene(sc)
    opt = oracle.options(oracle.RNG_COUNTER, oracle.MATH_DET, max_depth=depth, camera_model=thin)
    r32, r8, rays = osc.render(opt, spp)
    ids = osc.primary_ids(opt, 1)
    # linear scan in one of its four pipeline variants, then the BVH; one or two passes in flight, several passes per call
    for accel, cfg in ((1, dict(fuse_trace_shade=seed % 4, concurrent_passes=1 + seed % 2, samples_per_batch=1 + seed % 3)),
                       (2, dict(concurrent_passes=1 + (seed // 2) % 2, samples_per_batch=1 + (seed // 3) % 3))):
        with binding.PtGpu(accel=accel, max_depth=depth, camera_model=thin, **cfg) as gpu:
            gpu.upload_scene(sc)
            gpu.render(spp)
            p32, n = gpu.read_pixels32()
            p8 = gpu.read_pixels8()
            st = gpu.stats()
            got_ids = gpu.primary_ids(1)
        assert n == spp
        assert np.array_equal(got_ids, ids), (seed, accel)
        assert st["rays"] == rays, (seed, accel)
        assert np.array_equal(p32, r32) and np.array_equal(p8, r8), (seed, accel)


@pytest.mark.parametrize("seed", range(6))
def test_random_soup_bvh_equals_linear_scan(binding, oracle, seed):
    """A few thousand small primitives on a coarse grid (many coplanar, touching and duplicated ones, so equal distances
    are common): the BVH must pick the primitive the linear scan picks, for primary ids against the oracle and for the
    accumulated image against the GPU's own linear scan."""
    rng = np.random.RandomState(77 + seed)
    sc = S.Scene(
        image_settings=S.ImageSettings(96, 72, bool(seed % 2)),
        camera=S.Camera((0.0, 6.0, 22.0), (0.0, 5.0, 0.0), (0.0, 1.0, 0.0), 8.0, 6.0, 6.0, S.TONE_REINHARD, 1.0, 100.0),
        sky=S.Sky.constant((0.6, 0.7, 0.9)),
    )
    n = 1500 + 500 * (seed % 6)
    for i in range(n):
        k = rng.rand()
        if k < 0.3:
            sc.add(S.Sphere(tuple(float(x) for x in rng.randint(-8, 9, 3) + np.array([0, 8, 0])), float(rng.choice([0.25, 0.5, 1.0])), _material(rng)))
        elif k < 0.8:
            o, u, v = _axis_rectangle(rng, -8, 8)
            u = tuple(x / 4.0 for x in u)
            v = tuple(x / 4.0 for x in v)
            sc.add(S.Plane((o[0], o[1] + 8.0, o[2]), u, v, _material(rng)))
        else:
            o = tuple(float(x) for x in rng.uniform(-8.0, 8.0, 3) + np.array([0.0, 8.0, 0.0]))
            u = rng.uniform(-1.5, 1.5, 3)
            v = np.cross(u, rng.uniform(-1.5, 1.5, 3))
            sc.add(S.Plane(o, tuple(float(x) for x in u), tuple(float(x) for x in v), _material(rng)))
        if i % 50 == 49:   # a copy of a recent primitive
            if sc.planes and rng.rand() < 0.5:
                src = sc.planes[-1]
                sc.add(S.Plane(src.origin, src.u, src.v, _material(rng)))
            elif sc.spheres:
                src = sc.spheres[-1]
                sc.add(S.Sphere(src.origin, src.radius, _material(rng)))
    osc = oracle.OracleScene(sc)
    ref_ids = osc.primary_ids(oracle.options(), 2)
    out = {}
    for accel in (1, 2):
        with binding.PtGpu(accel=accel, max_depth=4) as gpu:
            gpu.upload_scene(sc)
            ids = gpu.primary_ids(2)
            gpu.render(2)
            out[accel] = (ids, gpu.read_pixels32()[0], gpu.stats()["rays"])
    assert np.array_equal(out[1][0], ref_ids) and np.array_equal(out[2][0], ref_ids)
    assert out[1][2] == out[2][2] and np.array_equal(out[1][1], out[2][1])


@pytest.mark.parametrize("depth", [8, 11])
def test_glass_everywhere_deep_tree(binding, oracle, depth):
    """Nothing but glass: every hit branches in two, so the levels hold many records per sample (dozens at the deepest) and the queue
    levels have to grow; the deepest levels see the NaN directions of total internal reflection."""
    sc = S.Scene(
        image_settings=S.ImageSettings(24, 18, True),
        camera=S.Camera((0.0, 3.0, 12.0), (0.0, 3.0, 0.0), (0.0, 1.0, 0.0), 8.0, 6.0, 4.0, S.TONE_EXPOSURE, 0.7, 100.0),
        sky=S.Sky.constant((0.9, 0.8, 0.7)),
    )
    sc.add(S.Plane((-8.0, 0.0, 8.0), (16.0, 0.0, 0.0), (0.0, 0.0, -16.0), S.Material.translucent(1.3)))
    sc.add(S.Plane((-8.0, 0.0, -6.0), (16.0, 0.0, 0.0), (0.0, 12.0, 0.0), S.Material.translucent(1.7)))
    sc.add(S.Sphere((0.0, 3.0, 0.0), 3.0, S.Material.translucent(1.5)))
    sc.add(S.Sphere((0.0, 3.0, 0.0), 2.0, S.Material.translucent(1.2)))
    sc.add(S.Sphere((0.5, 3.0, 0.5), 0.8, S.Material.translucent(2.0)))
    sc.add(S.Sphere((-3.5, 2.0, 2.0), 1.5, S.Material.translucent(1.5)))
    sc.add(S.Sphere((3.5, 2.0, 2.0), 1.5, S.Material.emissive((3.0, 2.0, 1.0))))
    osc = oracle.OracleScene(sc)
    r32, r8, rays = osc.render(oracle.options(max_depth=depth), 2)
    for accel in (1, 2):
        with binding.PtGpu(accel=accel, max_depth=depth) as gpu:
            gpu.upload_scene(sc)
            gpu.render(2)
            p32, _ = gpu.read_pixels32()
            st = gpu.stats()
        assert st["rays"] == rays and np.array_equal(p32, r32), accel
    assert rays > 24 * 18 * 2 * 2 * depth   # (far more than one ray per level: the tree does branch)


@pytest.mark.parametrize("seed", range(48))
def test_random_call_sequences(binding, seed):
    """The handle as a state machine: random sequences of upload / clear / render(n) / reads / sync on one handle must
    leave exactly what a fresh handle renders in one call with the same number of samples since the last clear or
    upload, whatever the pass size, the number of passes in flight and the scene sizes uploaded in between."""
    rng = np.random.RandomState(4000 + seed)
    scenes = [S.sample_scene(40, 30, camera_width=20.0 / 3.0), S.sample_scene(64, 24, camera_width=40.0 / 3.0), random_scene(seed)[0]]
    cfg = dict(samples_per_batch=int(rng.randint(0, 4)), concurrent_passes=int(rng.randint(1, 4)), max_depth=int(rng.randint(2, 6)),
               accel=int(rng.choice([0, 2])))
    with binding.PtGpu(**cfg) as gpu:
        cur, n = None, 0
        frames = []      # (display bytes handed out by a frame call, samples they show) since the last upload or clear
        for _ in range(30):
            op = rng.randint(0, 11)
            if cur is None or op == 0:
                cur = scenes[rng.randint(0, len(scenes))]
                gpu.upload_scene(cur)
                n, frames = 0, []
            elif op == 1:
                gpu.clear()
                n, frames = 0, []
            elif op <= 4:
                k = int(rng.randint(1, 6))
                gpu.render(k)
                n += k
            elif op == 5:
                p32, ns = gpu.read_pixels32()
                assert ns == n
            elif op == 6:
                assert gpu.present_pixels8().shape == (cur.image_settings.height, cur.image_settings.width, 3)
            elif op == 7:
                gpu.sync()
            else:    # the window loop's call, between everything else: k samples and the newest frame that has arrived
                k = int(rng.randint(0, 4))
                img, shown = gpu.frame(k)
                n += k
                assert shown <= n
                frames.append((img.copy(), shown))
        got, ns = gpu.read_pixels32()
        got8 = gpu.read_pixels8()
    assert ns == n
    with binding.PtGpu(max_depth=cfg["max_depth"]) as fresh:
        fresh.upload_scene(cur)
        done = 0
        for img, shown in sorted(frames, key=lambda f: f[1]):   # every frame handed out: the display bytes after `shown` samples
            if shown > done:
                fresh.render(shown - done)
                done = shown
            assert np.array_equal(img, fresh.read_pixels8()), (shown, n)
        if n > done:
            fresh.render(n - done)
        want, _ = fresh.read_pixels32()
        want8 = fresh.read_pixels8()
    assert np.array_equal(got, want) and np.array_equal(got8, want8)

"""Statistical parity with the reference-faithful oracle, and size-independent checks at the full benchmark size.

The reference's RNG is one racy global xorshift32 and its transcendental functions are the platform libm, so
against THAT configuration (oracle: xorshift stream + libm) only the converged image can be compared: two
independent Monte-Carlo estimates of the same per-pixel expectation E[quantise(gamma(tonemap(L)))].
Tolerance (stated here, in 8-bit units of sum/n): RMSE <= 3.0 at 1024 spp, and the error must fall like
1/sqrt(spp) (x16 samples -> x4 +- 35 %).  The estimator is the same on both sides (no Russian roulette, both glass
lobes per sample, same light sampling), so there is no bias term to budget for.
"""
import numpy as np
import pytest

from pathtracer_amd import scene as S

pytestmark = pytest.mark.gpu


def _rmse(a, na, b, nb):
    return float(np.sqrt(np.mean((a.astype(np.float64) / na - b.astype(np.float64) / nb) ** 2)))


def test_converged_image_matches_reference_rng_and_libm(binding, oracle):
    sc = S.sample_scene(128, 72, camera_width=8.8888889)
    osc = oracle.OracleScene(sc)
    ref = {}
    for n in (64, 1024):
        ref[n], _, _ = osc.render(oracle.options(oracle.RNG_XORSHIFT, oracle.MATH_LIBM), n, threads=8)
    got = {}
    with binding.PtGpu() as gpu:
        gpu.upload_scene(sc)
        gpu.render(64)
        got[64], _ = gpu.read_pixels32()
        gpu.render(1024 - 64)
        got[1024], n = gpu.read_pixels32()
        assert n == 1024
    e64, e1024 = _rmse(got[64], 64, ref[64], 64), _rmse(got[1024], 1024, ref[1024], 1024)
    assert e1024 <= 3.0, e1024
    assert 4.0 * 0.65 <= e64 / e1024 <= 4.0 * 1.35, (e64, e1024)
    # no systematic offset: the mean image levels agree far better than the per-pixel noise
    for c in range(3):
        assert abs(got[1024][..., c].mean() - ref[1024][..., c].mean()) / 1024 < 0.25


def test_deterministic_math_vs_libm_same_rng(binding, oracle):
    """Same counter RNG, libm instead of the deterministic functions: identical for all but a handful of samples
    (a 1-ulp difference has to land exactly on a quantisation step or flip a specular/diffuse pick)."""
    sc = S.sample_scene(96, 54, camera_width=8.8888889)
    osc = oracle.OracleScene(sc)
    det, _, _ = osc.render(oracle.options(oracle.RNG_COUNTER, oracle.MATH_DET), 16)
    lib, _, _ = osc.render(oracle.options(oracle.RNG_COUNTER, oracle.MATH_LIBM), 16)
    with binding.PtGpu() as gpu:
        gpu.upload_scene(sc)
        gpu.render(16)
        got, _ = gpu.read_pixels32()
    assert np.array_equal(got, det)
    assert (got != lib).mean() < 0.02
    assert np.abs(got.astype(np.int64) - lib.astype(np.int64)).mean() < 0.5


def test_full_size_1080p_properties(binding, oracle):
    """At the benchmark's size (1920x1080) the oracle affords only a few samples per pixel (the test after this one), so
    also check what does not depend on size: sharded + combined == unsharded (linearity of the integer accumulator), incremental ==
    one-shot, pixels8 is the documented function of pixels32, accumulator bounds, ray bookkeeping."""
    sc = S.sample_scene(1920, 1080, camera_width=8.8888889)
    with binding.PtGpu() as gpu:
        gpu.upload_scene(sc)
        gpu.render(6)
        full, n = gpu.read_pixels32()
        p8 = gpu.read_pixels8()
        st = gpu.stats()
        gpu.clear()
        gpu.render(2); gpu.render(1); gpu.render(3)
        again, _ = gpu.read_pixels32()
    assert n == 6 and np.array_equal(full, again)
    parts = []
    for r in range(3):
        with binding.PtGpu(shard_index=r, shard_count=3) as g:
            g.upload_scene(sc)
            g.render(2)
            parts.append(g.read_pixels32()[0])
    assert np.array_equal(parts[0] + parts[1] + parts[2], full)
    assert full.max() <= 6 * 255 and full.min() >= 0
    assert np.array_equal(p8, oracle.OracleScene(sc).assign_pixels8(full, 6))
    P = 1920 * 1080
    assert st["samples"] == 6 * P
    assert 6 * P <= st["rays"] <= 6 * P * 31 + st["shadow_rays"]  # >= one ray per sample, <= a full glass tree
    assert st["records"] >= st["rays"] - st["shadow_rays"]
    # the light (id 9) region saturates: every sample there quantises to 255
    ids = None
    with binding.PtGpu(sub_pixel_override=0) as g:
        g.upload_scene(sc)
        ids = g.primary_ids(0)
    inner = np.zeros_like(ids, dtype=bool)
    # the jittered sample lies in [x,x+1]x[y,y+1]; its four corners are un-jittered sample points of this 3x3 block
    inner[1:-1, 1:-1] = True
    for dy in (0, 1, 2):
        for dx in (0, 1, 2):
            inner[1:-1, 1:-1] &= ids[dy:ids.shape[0] - 2 + dy, dx:ids.shape[1] - 2 + dx] == 9
    assert inner.sum() > 1000 and np.all(full[inner] == 6 * 255)


@pytest.mark.parametrize("width,height,spp", [(1920, 1080, 4), (3840, 2160, 1), (7680, 4320, 1)])
def test_full_size_pixels_bit_exact(binding, oracle, width, height, spp):
    """The benchmark's own image sizes, pixel for pixel: the oracle's worker pool (counter RNG, deterministic math: the
    result does not depend on which thread renders which tile) finishes a few samples per pixel in seconds."""
    sc = S.sample_scene(width, height, camera_width=8.8888889)
    with binding.PtGpu() as gpu:
        gpu.upload_scene(sc)
        gpu.render(spp)
        p32, n = gpu.read_pixels32()
        p8 = gpu.read_pixels8()
        st = gpu.stats()
    r32, r8, rays = oracle.OracleScene(sc).render(oracle.options(oracle.RNG_COUNTER, oracle.MATH_DET), spp, threads=16)
    assert n == spp and st["rays"] == rays
    assert np.array_equal(p32, r32) and np.array_equal(p8, r8)


def test_full_size_room_depth_16_bit_exact(binding, oracle):
    """The queue-compaction configuration at its own size: the closed diffuse room, 1920x1080, max_depth 16, against the
    oracle's worker pool pixel for pixel (one sample per pixel: 17 levels of records, a shrinking share of the paths
    ending on the light at every level)."""
    sc = S.diffuse_room_scene(1920, 1080)
    with binding.PtGpu(max_depth=16) as gpu:
        gpu.upload_scene(sc)
        gpu.render(1)
        p32, n = gpu.read_pixels32()
        st = gpu.stats()
    r32, _, rays = oracle.OracleScene(sc).render(oracle.options(max_depth=16), 1, threads=16)
    assert n == 1 and st["rays"] == rays and np.array_equal(p32, r32)
    occ = [x / (1920 * 1080) for x in st["level_records"][:17]]
    assert occ[0] == 1.0 and all(occ[k + 1] <= occ[k] for k in range(15)) and occ[15] > 0.8   # most paths live all 16 levels


def test_full_size_2160p_eight_shards_sum_to_one_gpu(binding):
    """The 3840x2160 configuration's gate, with the eight GPUs replayed one after the other on this one: shard k of 8
    renders the sample indices the group's split rule gives it, and the sum of the eight accumulators (what the RCCL
    reduction computes) is the single-GPU accumulator for the same 16 samples, bit for bit."""
    lib = binding.load_library()
    sc = S.sample_scene(3840, 2160, camera_width=8.8888889)
    with binding.PtGpu() as one:
        one.upload_scene(sc)
        one.render(16)
        ref, n_ref = one.read_pixels32()
    total = np.zeros_like(ref)
    n_total = 0
    for k in range(8):
        with binding.PtGpu(shard_index=k, shard_count=8) as gpu:
            gpu.upload_scene(sc)
            first = 0
            for call in (5, 11):
                mine = lib.ptgpu_group_shard_samples(first, call, k, 8)
                if mine:
                    gpu.render(mine)
                first += call
            p32, n = gpu.read_pixels32()
        total += p32
        n_total += n
    assert n_total == n_ref == 16 and np.array_equal(total, ref)

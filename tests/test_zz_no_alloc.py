"""ptgpu_render_noalloc (ray queues in the library's static device array, no allocation during the call) against
ptgpu_render on the same inputs: the accumulators must be identical bit for bit, on two calls in a row - the second on a
smaller input, so that whatever the first call left in the shared array would show."""
import numpy as np
import pytest

from conftest import materials_scene
from pathtracer_amd import scene as S

pytestmark = pytest.mark.gpu


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


def _both(binding, sc, spp, **cfg):
    """(pixels32, pixels8, stats) of spp samples through ptgpu_render and through ptgpu_render_noalloc."""
    out = []
    for entry in ("render", "render_noalloc"):
        with binding.PtGpu(**cfg) as gpu:
            gpu.upload_scene(sc)
            getattr(gpu, entry)(spp)
            p32, n = gpu.read_pixels32()
            out.append((p32, gpu.read_pixels8(), n, gpu.stats()))
    return out


def test_two_calls_in_a_row_second_smaller(binding, oracle):
    big = S.sample_scene(96, 64, camera_width=7.5)          # glass: queue levels sized for branching
    small = materials_scene(40, 30)                         # every material, sphere and plane lights (shadow rays)
    with binding.PtGpu() as ref, binding.PtGpu() as gpu:
        for sc, spp in ((big, 5), (small, 3)):
            ref.upload_scene(sc)
            gpu.upload_scene(sc)
            ref.render(spp)
            gpu.render_noalloc(spp)
            a32, an = ref.read_pixels32()
            b32, bn = gpu.read_pixels32()
            assert an == bn == spp
            assert np.array_equal(a32, b32)
            assert np.array_equal(ref.read_pixels8(), gpu.read_pixels8())
            sa, sb = ref.stats(), gpu.stats()
            for key in ("samples", "rays", "shadow_rays", "records", "terminal_records"):
                assert sa[key] == sb[key], key
            o32, o8, rays = oracle.OracleScene(sc).render(oracle.options(), spp)
            assert np.array_equal(b32, o32) and sb["rays"] == rays
        assert sb["shadow_rays"] > 0


def test_mixed_with_render_and_many_passes(binding):
    """Both entry points draw sample indices from the same counter, and a call larger than one pass (64 spp at most)
    or one round (64 passes) is cut up inside the call."""
    sc = materials_scene(32, 24)
    with binding.PtGpu(samples_per_batch=2) as ref, binding.PtGpu(samples_per_batch=2) as gpu:
        ref.upload_scene(sc)
        gpu.upload_scene(sc)
        ref.render(150)
        gpu.render(3)
        gpu.render_noalloc(140)
        gpu.render(7)
        a32, an = ref.read_pixels32()
        b32, bn = gpu.read_pixels32()
    assert an == bn == 150 and np.array_equal(a32, b32)


def test_glass_tree_grows_inside_the_scratch(binding):
    """Nested glass branches up to 2^depth-fold: the levels start sized for twofold branching, the voided passes are
    repeated with the scratch carved for more, and the sums still equal ptgpu_render's (whose levels grew as well)."""
    sc = _glass_scene(96, 80)
    (a32, a8, an, sa), (b32, b8, bn, sb) = _both(binding, sc, 7, max_depth=6, samples_per_batch=2)
    assert sa["queue_growth"] > 2                      # the scene does overflow twofold levels
    assert an == bn == 7 and np.array_equal(a32, b32) and np.array_equal(a8, b8)
    assert sa["rays"] == sb["rays"] and sa["records"] == sb["records"]


def test_limits_are_errors(binding):
    with binding.PtGpu() as gpu:
        gpu.upload_scene(S.sample_scene(640, 480, camera_width=6.6666667))   # one sample per pixel does not fit 64 MiB
        with pytest.raises(binding.PtGpuError) as e:
            gpu.render_noalloc(1)
        assert e.value.code == -1 and "static scratch" in str(e.value)
        gpu.render(1)                                                        # the handle is still good
        assert gpu.read_pixels32()[1] == 1
    with binding.PtGpu(accel=binding.ACCEL_BVH) as gpu:
        gpu.upload_scene(S.sample_scene(32, 32))
        with pytest.raises(binding.PtGpuError) as e:
            gpu.render_noalloc(1)
        assert e.value.code == -1 and "BVH" in str(e.value)
    with binding.PtGpu(branch_capacity=2, max_depth=6) as gpu:               # fixed capacity: an overflow is reported, not grown away
        gpu.upload_scene(_glass_scene(96, 80))
        with pytest.raises(binding.PtGpuError) as e:
            gpu.render_noalloc(4)
        assert e.value.code == -5
        assert gpu.read_pixels32()[1] == 0                                   # nothing was accumulated, nothing is counted


def test_largest_image_that_fits(binding):
    """424x424 pixels at max_depth 4 use 63.8 of the array's 64 MiB with one sample per pass: the carve reaches the far end
    of the array and the sums still equal ptgpu_render's.  (A closed room without glass: levels sized for one record
    per sample suffice.  The sample scene at this size overflows them through its glass sphere, would need levels twice
    as large, and is refused with PT_ERR_QUEUE_OVERFLOW.)"""
    sc = S.diffuse_room_scene(424, 424)
    (a32, a8, an, sa), (b32, b8, bn, sb) = _both(binding, sc, 3, max_depth=4)
    assert an == bn == 3 and np.array_equal(a32, b32) and np.array_equal(a8, b8) and sa["rays"] == sb["rays"]
    with binding.PtGpu() as gpu:
        gpu.upload_scene(S.sample_scene(424, 424))
        with pytest.raises(binding.PtGpuError) as e:
            gpu.render_noalloc(1)
        assert e.value.code == -5


def _device_free_bytes():
    """Free device memory as the HIP runtime reports it (hipMemGetInfo through the runtime libptgpu already loaded)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_the_call_takes_no_device_memory(binding):
    """What the handle holds and what the device has free are the same before and after a call (the first call on a fresh
    process is left out of the comparison: the runtime loads kernels and sets up their spill memory when they first run)."""
    sc = materials_scene(48, 36)
    with binding.PtGpu() as gpu:
        gpu.upload_scene(sc)
        gpu.render_noalloc(2)
        gpu.sync()
        held, free = gpu.stats()["device_bytes"], _device_free_bytes()
        for spp in (70, 3):            # more than one pass, then a small call
            gpu.render_noalloc(spp)
            gpu.sync()
            assert gpu.stats()["device_bytes"] == held
            assert _device_free_bytes() == free
        assert gpu.read_pixels32()[1] == 75

"""The in-library multi-GPU path (ptgpu_group_*: one handle per device, sample-index shards, RCCL reduction at
readback) on the one GPU a test box has: a one-device group must equal a plain handle bit for bit - through the real
RCCL communicator - and the group's split rule, applied to several handles on the same GPU, must reproduce the
single-GPU image.  More than one device is covered by the driver's scaling run only."""
import numpy as np
import pytest

from conftest import materials_scene
from pathtracer_amd import scene as S

pytestmark = pytest.mark.gpu


def test_one_device_group_equals_plain_handle(binding, oracle):
    sc = S.sample_scene(96, 64, camera_width=7.5)
    with binding.PtGroup(devices=[0]) as grp, binding.PtGpu() as gpu:
        assert grp.size == 1
        grp.upload_scene(sc)
        gpu.upload_scene(sc)
        for n in (1, 3, 2):                      # progressive: read between render calls, the devices keep their sums
            grp.render(n)
            gpu.render(n)
            a32, an = grp.read_pixels32()
            b32, bn = gpu.read_pixels32()
            assert an == bn and np.array_equal(a32, b32)
            assert np.array_equal(grp.read_pixels8(), gpu.read_pixels8())
            assert np.array_equal(grp.present_pixels8(), gpu.present_pixels8())
        ex = grp.exchange_stats()
        assert ex["bytes"] == 96 * 64 * 12 and ex["ms"] >= 0.0
        sa, sb = grp.stats(), gpu.stats()
        for key in ("samples", "rays", "shadow_rays", "records"):
            assert sa[key] == sb[key], key
        r32, r8, rays = oracle.OracleScene(sc).render(oracle.options(), 6)
        assert np.array_equal(a32, r32) and sa["rays"] == rays
        grp.clear()
        grp.render(2)
        c32, cn = grp.read_pixels32()
        assert cn == 2 and np.array_equal(c32, oracle.OracleScene(sc).render(oracle.options(), 2)[0])
        # a new scene of another size through the same group
        small = materials_scene(40, 30)
        grp.upload_scene(small)
        grp.render(3)
        d32, dn = grp.read_pixels32()
        assert dn == 3 and np.array_equal(d32, oracle.OracleScene(small).render(oracle.options(), 3)[0])


def test_group_errors(binding):
    with pytest.raises(binding.PtGpuError) as e:
        binding.PtGroup(devices=[0, 0])                      # one handle per device
    assert e.value.code == -1
    with pytest.raises(binding.PtGpuError) as e:
        binding.PtGroup(devices=[4096])
    assert e.value.code == -2                                # no such device
    with binding.PtGroup(devices=[0]) as grp:
        with pytest.raises(binding.PtGpuError) as e:
            grp.render(1)
        assert e.value.code == -4 and "no scene" in str(e.value)


@pytest.mark.parametrize("n_dev", [2, 3, 8])
def test_split_rule_on_one_gpu_reproduces_the_single_gpu_image(binding, n_dev):
    """What a group of n devices does, replayed with n plain handles on one GPU: handle k is shard k of n and renders,
    per call, the number of samples ptgpu_group_shard_samples gives it.  The sum of the accumulators (what ncclReduce
    computes) must be the image of one handle rendering all the samples."""
    lib = binding.load_library()
    sc = materials_scene(64, 48)
    calls = (1, 5, 2, 7, 1)
    with binding.PtGpu() as one:
        one.upload_scene(sc)
        one.render(sum(calls))
        ref, n_ref = one.read_pixels32()
    total = np.zeros_like(ref)
    n_total = 0
    for k in range(n_dev):
        with binding.PtGpu(shard_index=k, shard_count=n_dev) as gpu:
            gpu.upload_scene(sc)
            first = 0
            for c in calls:
                mine = lib.ptgpu_group_shard_samples(first, c, k, n_dev)
                if mine:
                    gpu.render(mine)
                first += c
            p32, n = gpu.read_pixels32()
        total += p32
        n_total += n
    assert n_total == n_ref == sum(calls)
    assert np.array_equal(total, ref)

"""The window loop on the GPU (reference src/window.rs:127-229: one sample per pixel per frame, then glDrawPixels of
Backbuffer.pixels8): ptgpu_frame / ptgpu_group_frame pipeline the two - a call enqueues its frame and hands out the newest
frame that has arrived on the host - and ptgpu_render_tiles splits a frame over several handles by 32x32 tile.  Every image handed out must be the
bytes the synchronous loop (ptgpu_render + ptgpu_read_pixels8) shows after the same number of samples, and the sums of
tile shards must add up to one handle's, bit for bit."""
import numpy as np
import pytest

from conftest import materials_scene
from pathtracer_amd import scene as S

pytestmark = pytest.mark.gpu


def _sync_loop(binding, sc, frames, spp=1, **cfg):
    """[(pixels8, pixels32) after k frames of spp samples] the synchronous way."""
    out = []
    with binding.PtGpu(**cfg) as gpu:
        gpu.upload_scene(sc)
        for _ in range(frames):
            gpu.render(spp)
            out.append((gpu.read_pixels8(), gpu.read_pixels32()[0]))
    return out


def _frame_loop(obj, frames, spp=1):
    """The images a loop of `frames` calls + one flush hands out: [(copy of pixels8, samples it shows)]."""
    got = []
    for _ in range(frames):
        img, n = obj.frame(spp)
        got.append((img.copy(), n))
    img, n = obj.frame(0)
    got.append((img.copy(), n))
    return got


def _check_sequence(got, want, spp, before=0):
    """Call k hands out the newest frame that has arrived - its own, or one or two before it (never older: that one is
    waited for) -, the first call its own, the flush the last; every image is the synchronous loop's after that many
    samples.  `before`: samples rendered before the first frame."""
    frames = len(got) - 1
    shown = [(n - before) // spp for _, n in got]
    assert all((n - before) % spp == 0 for _, n in got)
    assert shown[0] == 1 and shown[-1] == frames
    for k, f in enumerate(shown[:-1], start=1):
        assert max(1, k - 2) <= f <= k, (k, f)
    assert shown == sorted(shown)
    for img, n in got:
        assert np.array_equal(img, want[n // spp - 1][0]), f"image after {n} samples"


@pytest.mark.parametrize("spp,batch", [(1, 0), (3, 2)])
def test_frame_loop_shows_what_the_synchronous_loop_shows(binding, oracle, spp, batch):
    sc = S.sample_scene(96, 64, camera_width=7.5)
    want = _sync_loop(binding, sc, 7, spp, samples_per_batch=batch)
    with binding.PtGpu(samples_per_batch=batch) as gpu:
        gpu.upload_scene(sc)
        got = _frame_loop(gpu, 7, spp)
        _check_sequence(got, want, spp)
        p32, n = gpu.read_pixels32()
        assert n == 7 * spp and np.array_equal(p32, want[-1][1])
        assert np.array_equal(gpu.read_pixels8(), want[-1][0])
        # after a clear the loop starts afresh: the first call waits for its own frame again
        gpu.clear()
        _check_sequence(_frame_loop(gpu, 3, spp), want, spp)
    # and the whole thing is the reference's Backbuffer after that many frames
    o32, o8, _ = oracle.OracleScene(sc).render(oracle.options(), 7 * spp)
    assert np.array_equal(want[-1][1], o32) and np.array_equal(want[-1][0], o8)


def test_frames_mixed_with_render_and_reads(binding):
    sc = materials_scene(64, 48)
    want = _sync_loop(binding, sc, 9)
    with binding.PtGpu() as gpu:
        gpu.upload_scene(sc)
        img, n = gpu.frame(0)                       # a flush before anything was rendered: the empty backbuffer
        assert n == 0 and not img.any()
        gpu.render(2)                               # plain samples first: the frames count them
        img, n = gpu.frame(1)                       # the first frame: waited for
        assert n == 3 and np.array_equal(img, want[2][0])
        assert np.array_equal(gpu.read_pixels8(), want[2][0])     # a synchronous read in the middle
        img, n = gpu.frame(1)                       # its own frame if that has arrived already, else the one before
        assert n in (3, 4) and np.array_equal(img, want[n - 1][0])
        gpu.render(4)
        img, n = gpu.frame(1)
        assert n in (3, 4, 9) and np.array_equal(img, want[n - 1][0])
        img, n = gpu.frame(0)
        assert n == 9 and np.array_equal(img, want[8][0])


def test_frame_that_overflows_its_queues_is_repeated(binding):
    """Nested glass branches beyond the levels' first size: the passes of the first frames are voided, rendered again with
    larger levels, and no image that misses samples is ever handed out."""
    from test_zz_no_alloc import _glass_scene
    sc = _glass_scene(96, 80)
    want = _sync_loop(binding, sc, 6, max_depth=6, samples_per_batch=1)
    with binding.PtGpu(max_depth=6, samples_per_batch=1) as gpu:   # (levels sized for one sample per pass: glass overflows them)
        gpu.upload_scene(sc)
        got = _frame_loop(gpu, 6)
        assert gpu.stats()["queue_growth"] > 1       # the scene did overflow
        for img, n in got:
            assert 1 <= n <= 6 and np.array_equal(img, want[n - 1][0]), n
        assert got[-1][1] == 6
        assert np.array_equal(gpu.read_pixels32()[0], want[-1][1])


@pytest.mark.parametrize("n_shards,size", [(2, (96, 64)), (3, (100, 70)), (8, (256, 144)), (5, (40, 30))])
def test_tile_shards_add_up_to_one_handle(binding, n_shards, size):
    """One handle per shard on the one GPU (what the devices of a node do): sums disjoint by tile, their total the image."""
    sc = S.sample_scene(*size, camera_width=5.0 * size[0] / size[1])
    with binding.PtGpu() as one:
        one.upload_scene(sc)
        one.render(3)
        want32, _ = one.read_pixels32()
        want8 = one.read_pixels8()
    total = np.zeros_like(want32)
    shown = np.zeros_like(want8)
    owned = np.zeros(want32.shape[:2], dtype=np.int32)
    tiles_x = (size[0] + 31) // 32
    for k in range(n_shards):
        with binding.PtGpu(shard_index=k, shard_count=n_shards) as gpu:
            gpu.upload_scene(sc)
            gpu.render_tiles(0, 1)
            gpu.render_tiles(1, 2)
            p32, n = gpu.read_pixels32()
            assert n == 3
            p8 = gpu.read_pixels8()
        mine = np.zeros(want32.shape[:2], dtype=bool)
        for ty in range((size[1] + 31) // 32):
            for tx in range(tiles_x):
                if binding.tile_shard(tx, ty, tiles_x, n_shards) == k:
                    mine[ty * 32:(ty + 1) * 32, tx * 32:(tx + 1) * 32] = True
        assert not p32[~mine].any()                  # nothing outside its own tiles
        assert np.array_equal(p32[mine], want32[mine]) and np.array_equal(p8[mine], want8[mine])
        total += p32
        shown += p8
        owned += mine
    assert (owned == 1).all()
    assert np.array_equal(total, want32) and np.array_equal(shown, want8)


def test_group_of_one_frame_loop(binding):
    """The group's frame loop through the real RCCL communicator (one device: the exchange is a copy)."""
    sc = S.sample_scene(96, 64, camera_width=7.5)
    want = _sync_loop(binding, sc, 5)
    with binding.PtGroup(devices=[0]) as group:
        group.upload_scene(sc)
        _check_sequence(_frame_loop(group, 5), want, 1)
        p32, n = group.read_pixels32()
        assert n == 5 and np.array_equal(p32, want[-1][1])
        group.render(3)                              # (a group of one: still "tiles", the loop stays pipelined)
        img, n = group.frame(0)
        assert n == 5                                # the newest FRAME is still the fifth
        img, n = group.frame(1)
        assert n in (5, 9)
        img, n = group.frame(0)
        assert n == 9


def test_group_of_four_on_one_device(binding, monkeypatch):
    """Four members - tile split, per-member frames, flags, the exchange of display bytes - on the one GPU at hand
    (PTGPU_GROUP_SHARED_DEVICE: the exchange is a kernel there, RCCL cannot connect two ranks on one device)."""
    monkeypatch.setenv("PTGPU_GROUP_SHARED_DEVICE", "1")
    sc = materials_scene(160, 100)
    want = _sync_loop(binding, sc, 17)
    with binding.PtGroup(devices=[0, 0, 0, 0]) as group:
        assert group.size == 4
        group.upload_scene(sc)
        _check_sequence(_frame_loop(group, 6), want, 1)
        group.render(2)                              # fewer samples than members: by tile
        p32, n = group.read_pixels32()
        assert n == 8 and np.array_equal(p32, want[7][1])
        assert np.array_equal(group.read_pixels8(), want[7][0])
        img, n = group.frame(1)                      # still only tile shards: pipelined (its own frame, or the sixth)
        assert n in (6, 9) and np.array_equal(img, want[n - 1][0])
        img, n = group.frame(0)
        assert n == 9 and np.array_equal(img, want[8][0])
        group.render(7)                              # by sample index: 7 samples over 4 members
        p32, n = group.read_pixels32()
        assert n == 16 and np.array_equal(p32, want[15][1])
        img, n = group.frame(1)                      # sums are no longer per tile: rendered, reduced and shown at once
        assert n == 17 and np.array_equal(img, want[16][0])
        st = group.stats()
        assert st["samples"] == 17 * 160 * 100
        group.clear()
        _check_sequence(_frame_loop(group, 3), want, 1)

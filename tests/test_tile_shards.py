"""The tile-shard rule (ptgpu_tile_shard: pure arithmetic, no GPU): every 32x32 work tile of the reference's fan-out
(reference src/window.rs:166-179) belongs to exactly one of n handles, the handles' shares differ by at most a few tiles,
and no handle owns whole columns of tiles when a row of tiles holds a whole number of rounds."""
import numpy as np
import pytest


@pytest.mark.parametrize("size", [(1920, 1080), (3840, 2160), (1024, 1024), (96, 64), (33, 31)])
@pytest.mark.parametrize("n", [1, 2, 3, 4, 8, 16])
def test_every_tile_has_one_owner_and_shares_are_even(binding, size, n):
    tiles_x, tiles_y = (size[0] + 31) // 32, (size[1] + 31) // 32
    owner = np.array([[binding.tile_shard(tx, ty, tiles_x, n) for tx in range(tiles_x)] for ty in range(tiles_y)])
    assert owner.min() >= 0 and owner.max() < n
    counts = np.bincount(owner.ravel(), minlength=n)
    assert counts.sum() == tiles_x * tiles_y
    assert counts.max() - counts.min() <= max(1, tiles_y if tiles_x % n == 0 and n > 1 else 1)
    if n > 1 and tiles_x >= n and tiles_y >= 2:
        # neighbours differ along a row, and a column is not all one owner
        assert (owner[:, 1:] != owner[:, :-1]).all()
        assert any(len(set(owner[:, tx])) > 1 for tx in range(tiles_x))


def test_rows_shift_when_a_row_holds_whole_rounds(binding):
    # 1024 pixels = 32 tiles per row over 8 handles: without the shift handle k would own columns k, k + 8, ...
    owner = np.array([[binding.tile_shard(tx, ty, 32, 8) for tx in range(32)] for ty in range(8)])
    for ty in range(1, 8):
        assert np.array_equal(owner[ty], np.roll(owner[0], -ty))

"""N > 1 path on CPU: two processes over gloo render disjoint sample shards (with the oracle standing in for
the GPU renderer), all-reduce the integer accumulators through pathtracer_amd.distributed and must land on
exactly the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from pathtracer_amd import distributed as D
from pathtracer_amd import scene as S


def test_shard_index_sets_partition_the_samples():
    for world in (1, 2, 3, 8):
        for total in (0, 1, 7, 16, 1024):
            counts = D.split_samples(total, world)
            assert sum(counts) == total
            got = sorted(s for r in range(world) for s in D.shard_sample_indices(r, world, counts[r]))
            assert got == list(range(total))
    assert D.shard_sample_indices(1, 4, 3, first_local=2) == [9, 13, 17]
    with pytest.raises(ValueError):
        D.shard_sample_indices(2, 2, 1)


def _worker(rank, world, port, total, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import ptref as R
    sc = S.sample_scene(48, 32, camera_width=7.5)
    osc = R.OracleScene(sc)
    opt = R.options(R.RNG_COUNTER, R.MATH_DET)
    n_local = D.split_samples(total, world)[rank]
    p32, _, _ = osc.render(opt, n_local, sample_indices=D.shard_sample_indices(rank, world, n_local))
    t = torch.from_numpy(p32.view(np.int32).copy())
    D.allreduce_sum_(t)
    n = torch.tensor([n_local], dtype=torch.int64)
    dist.all_reduce(n)
    if rank == 0:
        np.save(out_path, np.concatenate([t.numpy().ravel().view(np.uint32), np.array([int(n.item())], dtype=np.uint32)]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_single_process(tmp_path, oracle):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    total = 5  # odd on purpose: ranks hold 3 and 2 samples
    out = str(tmp_path / "combined.npy")
    mp.spawn(_worker, args=(2, port, total, out), nprocs=2, join=True)
    got = np.load(out)
    sc = S.sample_scene(48, 32, camera_width=7.5)
    ref, _, _ = oracle.OracleScene(sc).render(oracle.options(oracle.RNG_COUNTER, oracle.MATH_DET), total)
    assert got[-1] == total
    assert np.array_equal(got[:-1], ref.ravel())

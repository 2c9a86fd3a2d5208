"""Two PROCESSES render disjoint sample shards with the GPU renderer - both on the one card this box has - and combine
their accumulators through pathtracer_amd.distributed exactly as the ranks of a multi-GPU run do (RANK / WORLD_SIZE from
the environment -> shard_index / shard_count -> export -> all-reduce -> import).  RCCL refuses two ranks on one device,
so the exchange goes over gloo (through host memory); everything before and after it is the code path of an N-GPU run.
The combined image must equal one handle rendering all the samples, and the oracle, bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT
from pathtracer_amd import scene as S

pytestmark = pytest.mark.gpu

WORLD = 2
TOTAL = 7          # odd on purpose: the ranks hold 4 and 3 samples
SIZE = (160, 96)


def _scene():
    return S.sample_scene(*SIZE, camera_width=7.5)


def _worker(rank, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD), LOCAL_RANK="0")
    import torch.distributed as dist
    from pathtracer_amd import binding, distributed as D
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    r, w, _ = D.env_rank()
    n_local = D.split_samples(TOTAL, w)[r]
    with binding.PtGpu(device=0, shard_index=r, shard_count=w) as gpu:
        gpu.upload_scene(_scene())
        gpu.render(n_local)
        gpu.sync()
        total = D.combine_accumulators(gpu, n_local)
        p32, n = gpu.read_pixels32()
        p8 = gpu.read_pixels8()
        rays = gpu.stats()["rays"]
    assert total == TOTAL and n == TOTAL
    np.savez(out_path % rank, p32=p32, p8=p8, rays=np.array([rays], dtype=np.uint64))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_one_handle_and_the_oracle(tmp_path, binding, oracle):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_worker, args=(port, out), nprocs=WORLD, join=True)
    got = [np.load(out % r) for r in range(WORLD)]
    with binding.PtGpu(device=0) as gpu:
        gpu.upload_scene(_scene())
        gpu.render(TOTAL)
        one32, n = gpu.read_pixels32()
        one8 = gpu.read_pixels8()
        one_rays = gpu.stats()["rays"]
    assert n == TOTAL
    ref32, ref8, _ = oracle.OracleScene(_scene()).render(oracle.options(oracle.RNG_COUNTER, oracle.MATH_DET), TOTAL)
    for r in range(WORLD):   # every rank ends with the whole image
        assert np.array_equal(got[r]["p32"], one32) and np.array_equal(got[r]["p8"], one8)
        assert np.array_equal(got[r]["p32"], ref32) and np.array_equal(got[r]["p8"], ref8)
    assert int(got[0]["rays"][0]) + int(got[1]["rays"][0]) == one_rays   # the shards partition the work

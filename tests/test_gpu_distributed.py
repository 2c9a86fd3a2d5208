"""The RCCL combine path of bench.py, as far as one GPU can exercise it: launched by torch.distributed.run the way
the scaling bench is, one rank, with the switch that forces export -> nccl all_reduce -> import on a single rank.
(The N > 1 arithmetic is covered by tests/test_distributed_gloo.py and test_sample_sharding_sums_to_single_gpu.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_under_torchrun_with_nccl_collective():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PTGPU_FORCE_COLLECTIVE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--width", "640", "--height", "360", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["config"]["accumulated_spp"] == 2 * d["config"]["spp_per_step"] and d["value"] > 0
    assert d["roofline"]["kernel"] == "trace_shade" and 0 < d["roofline"]["frac"] < 1
    assert d["exchange"]["bytes"] == 640 * 360 * 12 and d["exchange"]["ms"] > 0


def test_bench_with_two_ranks_sharing_the_gpu():
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, two ranks, RANK / LOCAL_RANK / WORLD_SIZE from
    the environment) - on the one GPU of this box: both ranks render on device 0 (PTGPU_BENCH_ONE_DEVICE) and, RCCL
    refusing two ranks on one device, exchange over gloo.  Everything else - shard indices, barriers, the sum and the maximum
    over the ranks, rank 0's line - is the code of an N-GPU run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PTGPU_BENCH_ONE_DEVICE="1", PTGPU_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--width", "640", "--height", "360", "--spp-per-step", "32", "--batch", "8"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                    # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["accumulated_spp"] == 2 * 2 * 32      # both ranks' samples in every rank's accumulator
    total_msamples = 2 * 2 * 32 * 640 * 360 / 1e6             # the sum over the ranks over the slowest rank's time
    assert abs(d["msamples_per_s"] * d["ms_per_step"] * 2 / 1e3 - total_msamples) < 0.02 * total_msamples
    assert d["roofline"]["kernel"] == "trace_shade" and 0 < d["roofline"]["frac"] < 1   # rank 0's one-pass leg
    assert "cpu_baseline" not in d and "frame_loop" not in d  # (N = 1 only)
    assert d["exchange"]["bytes"] == 640 * 360 * 12


def test_bench_inproc_group():
    """bench.py --inproc: the same JSON shape with the GPUs driven from one process through the library's PtGroup."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--inproc", "--steps", "2", "--warmup", "1", "--spp-per-step", "8",
           "--width", "640", "--height", "360", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["accumulated_spp"] == 16 and d["value"] > 0 and "PtGroup" in d["config"]["parallelism"]
    assert d["roofline"]["kernel"] == "trace_shade" and 0 < d["roofline"]["frac"] < 1
    assert d["exchange"]["bytes"] == 640 * 360 * 12 and d["exchange"]["ms"] > 0 and d["exchange"]["device_ms"] >= 0


@pytest.mark.parametrize("inproc", [False, True])
def test_bench_strong_scaling_mode(inproc):
    """bench.py --scaling strong (a fixed total of samples split over the GPUs by sample index; reports seconds to that
    total and the exchange's share), as one torch.distributed rank and through the library's PtGroup."""
    args = ["--gpus", "1", "--scaling", "strong", "--width", "320", "--height", "180", "--total-spp", "48", "--steps", "3", "--warmup", "1",
            "--no-cpu-baseline"]
    if inproc:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--inproc"] + args
        env = dict(os.environ)
    else:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ, PTGPU_FORCE_COLLECTIVE="1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["config"]["accumulated_spp"] == 48 and d["value"] > 0
    assert d["sec_to_total_spp"]["total_spp"] == 48 and 0 < d["sec_to_total_spp"]["of_which_exchange_s"] < d["sec_to_total_spp"]["seconds"]
    seconds = d["ms_per_step"] * d["steps"] * 1e-3
    assert abs(d["msamples_per_s"] * 1e6 * seconds / (320 * 180 * 48) - 1.0) < 0.02   # every sample of the total, once
    assert d["exchange"]["bytes"] == 320 * 180 * 12

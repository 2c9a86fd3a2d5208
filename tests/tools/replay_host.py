"""One-off soak on the CPU: the oracle renders the randomised scenes of tests/test_gpu_fuzz.py with its ray log on, and
every logged ray is walked through the BVH by the traversal functions of the kernels compiled for the host
(ptgpu_bvh_host_trace), which check every reference into the tree and every stack slot.  Prints the seeds whose rays
leave the tree, differ from the oracle's answer or start at NaN.
python tests/tools/replay_host.py FIRST COUNT"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import test_gpu_fuzz as F
from oracle import ptref
from pathtracer_amd import binding
binding.load_library(); ptref.lib()
first, count = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time(); total = 0
for seed in range(first, first + count):
    sc, depth, thin = F.random_scene(seed)
    osc = ptref.OracleScene(sc)
    opt = ptref.options(ptref.RNG_COUNTER, ptref.MATH_DET, max_depth=depth, camera_model=thin)
    _, _, _, log = osc.render_logging_rays(opt, 3)
    prims, ts, f = binding.bvh_host_trace(sc, log[:, :6])
    total += len(log)
    want = log[:, 6].view(np.uint32)
    nan_o = int(np.isnan(log[:, :3]).any(axis=1).sum())
    bad = (prims != want) | ((want != 0xFFFFFFFF) & (ts != log[:, 7]) & ~(np.isnan(ts) & np.isnan(log[:, 7])))
    if f["bad_units"] or f["bad_kinds"] or f["bad_slots"] or bad.any() or nan_o:
        print("seed", seed, "rays", len(log), f, "mismatches", int(bad.sum()), "rays from NaN", nan_o, flush=True)
print("done", count, "seeds", total, "rays", f"{time.time() - t0:.0f} s")

"""One-off soak of a randomised test over many more seeds than the suite runs:
python tests/tools/soak.py FIRST COUNT [scenes|calls|soups]"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_fuzz as F
from oracle import ptref
from pathtracer_amd import binding
binding.load_library(); ptref.lib()
first, count = int(sys.argv[1]), int(sys.argv[2])
which = sys.argv[3] if len(sys.argv) > 3 else "scenes"
t = time.time(); bad = []
for seed in range(first, first + count):
    try:
        if which == "scenes": F.test_random_scene_three_ways(binding, ptref, seed)
        elif which == "calls": F.test_random_call_sequences(binding, seed)
        else: F.test_random_soup_bvh_equals_linear_scan(binding, ptref, seed)
    except AssertionError as e:
        bad.append(seed); print("MISMATCH seed", seed, str(e)[:200], flush=True)
    if (seed - first) % 500 == 499: print(seed, f"{time.time()-t:.0f} s", flush=True)
print("done", count, "seeds,", len(bad), "mismatches", bad[:20], flush=True)

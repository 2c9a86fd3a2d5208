"""Soak of the shared-reciprocal normalize (pt_math.h) against numpy's IEEE float32 divisions: ROUNDS x 2^22 vectors
of like-magnitude components over every exponent (so every mantissa of the length, the all-ones ones included, turns up).
python profiles/debug/normalize_soak.py [rounds]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathtracer_amd import binding

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = 1 << 22
bad = 0
ones = 0
with binding.PtGpu() as gpu:
    for r in range(rounds):
        rng = np.random.RandomState(1000 + r)
        e = (rng.randint(30, 220, n).astype(np.uint32) + rng.randint(0, 3, n).astype(np.uint32)) << 23
        x = (e | rng.randint(0, 1 << 23, n).astype(np.uint32) | (rng.randint(0, 2, n).astype(np.uint32) << 31)).view(np.float32)
        e2 = (e >> 23) + rng.randint(-2, 3, n).astype(np.uint32)
        y = ((e2 << 23) | rng.randint(0, 1 << 23, n).astype(np.uint32) | (rng.randint(0, 2, n).astype(np.uint32) << 31)).view(np.float32)
        x[rng.randint(0, n, n >> 6)] = 0.0
        y[rng.randint(0, n, n >> 6)] = -0.0
        x[rng.randint(0, n, n >> 8)] = -0.0
        with np.errstate(all="ignore"):
            z = (x * np.float32(0.37) + y).astype(np.float32)
            length = np.sqrt(((x * x + y * y) + z * z).astype(np.float32)).astype(np.float32)
        ones += int(((length.view(np.uint32) & 0x7FFFFF) == 0x7FFFFF).sum())
        for fn, c in zip((9, 10, 11), (x, y, z)):
            with np.errstate(all="ignore"):
                w = (c / length).astype(np.float32)
            got = gpu.eval_math(fn, x, y)
            same = (got.view(np.uint32) == w.view(np.uint32)) | (np.isnan(got) & np.isnan(w))
            if not same.all():
                bad += int((~same).sum())
                print("round", r, "fn", fn, "differs", int((~same).sum()), x[~same][:3], y[~same][:3], got[~same][:3], w[~same][:3], flush=True)
        if r % 8 == 7:
            print("round", r + 1, "of", rounds, "mismatches", bad, "lengths with an all-ones mantissa", ones, flush=True)
print("vectors", rounds * n, "mismatches", bad)
sys.exit(1 if bad else 0)

"""One-off (CPU): the oracle's deterministic sin / cos / exp / acos / pow(x,5) / pow(x,1/2.2) against the platform libm
for EVERY binary32 input: how many results differ and by how many ulp at most (finite results; NaN must meet NaN)."""
import sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, ".")
import numpy as np
from oracle import ptref as R
R.lib()
fns = [int(a) for a in sys.argv[1:]] or [0, 1, 3, 4, 25, 22]
SUB = 1 << 22
import os
LIMIT = float(os.environ.get("SINCOS_LIMIT", "0"))

def ordered(u):  # monotone integer image of the float bit pattern
    u = u.astype(np.int64)
    return np.where(u & 0x80000000, 0x80000000 - u, u)

def task(args):
    fn, c = args
    x = np.arange(c, c + SUB, dtype=np.uint64).astype(np.uint32).view(np.float32)
    if fn >= 20:
        y = np.full(SUB, np.float32(5.0 if fn == 25 else 1.0 / 2.2), dtype=np.float32)
        a, b = R.math_eval(R.MATH_DET, 2, x, y), R.math_eval(R.MATH_LIBM, 2, x, y)
    else:
        a, b = R.math_eval(R.MATH_DET, fn, x), R.math_eval(R.MATH_LIBM, fn, x)
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    nan_bad = int((nan_a != nan_b).sum())
    ok = ~(nan_a | nan_b)
    if fn in (0, 1) and LIMIT > 0: ok &= np.abs(x) <= LIMIT   # the path only takes sin / cos of 2*pi*xi, xi in (0, 1]
    d = np.abs(ordered(a.view(np.uint32)[ok]) - ordered(b.view(np.uint32)[ok]))
    return int((d != 0).sum()), int(d.max()) if d.size else 0, nan_bad

with ThreadPoolExecutor(16) as pool:
    for fn in fns:
        t = time.time(); differ = 0; worst = 0; nan_bad = 0
        for n, (dn, mx, nb) in enumerate(pool.map(task, ((fn, c) for c in range(0, 1 << 32, SUB)))):
            differ += dn; worst = max(worst, mx); nan_bad += nb
            if n % 256 == 255: print(f"fn {fn}: {100.0 * (n + 1) * SUB / 2**32:.0f} %", flush=True)
        print(f"fn {fn}: all 2^32 inputs: {differ} results differ from libm ({100.0 * differ / 2**32:.4f} %), at most {worst} ulp, NaN disagreements {nan_bad}, {time.time() - t:.0f} s", flush=True)

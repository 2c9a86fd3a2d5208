"""One-off: the device's deterministic sin / cos / exp / acos against the oracle's for EVERY binary32 input (2^32 each),
bit patterns compared; powf(x, y) for every x at the two exponents the path uses (5 and 1/2.2, given as fn 25 and 22).
python tests/tools/math_exhaustive.py [fn ...]   (0 sin, 1 cos, 3 exp, 4 acos, 25 pow(x,5), 22 pow(x,1/2.2), 6 sqrt, 8 random32 of the bit pattern;
7 = division: 2^32 pairs of random bit patterns, half of them with the divisor's exponent forced near the dividend's)"""
import sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, ".")
import numpy as np
from oracle import ptref as R
from pathtracer_amd import binding

fns = [int(a) for a in sys.argv[1:]] or [0, 1, 3, 4, 25, 22, 6, 8, 7]
CH = 1 << 26            # inputs per device call
SUB = 1 << 22           # inputs per oracle thread task
R.lib(); binding.load_library()
with binding.PtGpu() as gpu, ThreadPoolExecutor(16) as pool:
    gpu.upload_scene(__import__("pathtracer_amd.scene", fromlist=["x"]).sample_scene(8, 8))
    for fn in fns:
        t = time.time(); bad = 0; first = None
        for c in range(0, 1 << 32, CH):
            x = np.arange(c, c + CH, dtype=np.uint64).astype(np.uint32).view(np.float32)
            if fn == 7:
                rng = np.random.RandomState(c >> 26)
                x = rng.randint(0, 1 << 32, CH, dtype=np.uint64).astype(np.uint32)
                yb = rng.randint(0, 1 << 32, CH, dtype=np.uint64).astype(np.uint32)
                if (c >> 26) & 1:   # divisor exponent within +-3 of the dividend's
                    e = ((x >> 23) & 255).astype(np.int64) + rng.randint(-3, 4, CH)
                    yb = (yb & np.uint32(0x807FFFFF)) | (np.clip(e, 0, 255).astype(np.uint32) << 23)
                x = x.view(np.float32); y = yb.view(np.float32)
                got = gpu.eval_math(7, x, y).view(np.uint32)
                parts = list(pool.map(lambda k: R.math_eval(R.MATH_DET, 7, x[k:k + SUB], y[k:k + SUB]).view(np.uint32), range(0, CH, SUB)))
            elif fn >= 20:
                y = np.full(CH, np.float32(5.0 if fn == 25 else 1.0 / 2.2), dtype=np.float32)
                got = gpu.eval_math(2, x, y).view(np.uint32)
                parts = list(pool.map(lambda k: R.math_eval(R.MATH_DET, 2, x[k:k + SUB], y[k:k + SUB]).view(np.uint32), range(0, CH, SUB)))
            else:
                got = gpu.eval_math(fn, x).view(np.uint32)
                parts = list(pool.map(lambda k: R.math_eval(R.MATH_DET, fn, x[k:k + SUB]).view(np.uint32), range(0, CH, SUB)))
            ref = np.concatenate(parts)
            ne = got != ref
            if ne.any():
                bad += int(ne.sum())
                if first is None:
                    j = int(np.argmax(ne)); first = (hex(c + j), hex(int(got[j])), hex(int(ref[j])))
            if (c // CH) % 8 == 7: print(f"fn {fn}: {100.0 * (c + CH) / 2**32:.0f} %  {time.time() - t:.0f} s  mismatches {bad}", flush=True)
        print(f"fn {fn}: all 2^32 inputs, {bad} mismatches, first {first}, {time.time() - t:.0f} s", flush=True)

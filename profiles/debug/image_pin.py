"""Measures how close the GPU render of the sample scene comes to the reference's published render (docs/5.jpg) at the
two fixture sizes and several sample counts; the bounds of tests/test_reference_image.py are 1.2x these figures."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pathtracer_amd import binding as B, scene as S

def errors(p8, ref):
    d = p8[::-1].astype(np.float64) - ref.astype(np.float64)
    pp = np.sqrt((d ** 2).mean(axis=2))
    return round(float(np.sqrt((d ** 2).mean())), 3), [round(float(x), 3) for x in d.mean(axis=(0, 1))], round(float(np.median(pp)), 3), round(float((pp > 20).mean()), 4)

for n in (256, 512):
    fx = np.load(os.path.join(ROOT, "tests", "golden", f"reference_render_{n}.npz"))
    with B.PtGpu() as gpu:
        gpu.upload_scene(S.sample_scene(n, n))
        done = 0
        for spp in (2048, 8192, 32768):
            gpu.render(spp - done); done = spp
            p8 = gpu.read_pixels8()
            line = {k: errors(p8, fx[k]) for k in fx.files}
            print(n, spp, line, flush=True)

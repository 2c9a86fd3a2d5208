"""Diagnostic: where the BVH pipeline overtakes the fused linear scan.  The closed room (1080p, depth 8) with N spheres,
rendered with the acceleration structure forced off and on.  Usage (GPU box): python profiles/debug/accel_crossover.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathtracer_amd import binding as B, scene as S

for n in (20, 50, 80, 110, 140, 170, 230):
    sc = S.diffuse_room_scene(1920, 1080, n_spheres=n)
    out = []
    for accel in (1, 2):   # PT_ACCEL_LINEAR, PT_ACCEL_BVH
        with B.PtGpu(max_depth=8, accel=accel) as g:
            g.upload_scene(sc)
            g.render(32); g.sync()
            r0 = g.stats()["rays"]
            t = time.perf_counter()
            g.render(128); g.sync()
            dt = time.perf_counter() - t
            out.append((g.stats()["rays"] - r0) / dt / 1e9)
    print(f"{n + 7} primitives: linear scan {out[0]:.2f} Grays/s, BVH {out[1]:.2f} Grays/s", flush=True)

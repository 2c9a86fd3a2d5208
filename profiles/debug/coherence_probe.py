"""Diagnostic: how much of the bounce rays' extra cost in the BVH trace kernel is incoherence.  The 1M-rectangle field as
it is (rough surfaces: bounce directions scatter over the hemisphere) and with every rectangle a perfect mirror (bounce
rays of neighbouring pixels stay neighbours); the level-1 figures are the difference between max_depth 2 and 1.
Usage (GPU box): python profiles/debug/coherence_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathtracer_amd import binding as B, scene as S


def run(sc, depth):
    with B.PtGpu(max_depth=depth, profile_events=1) as g:
        g.upload_scene(sc)
        g.render(64); g.sync()
        st = g.stats()
    return st["kernel_ms"]["trace"], st["rays"] - st["shadow_rays"], st["bvh_steps"]


for name in ("rough (as shipped)", "mirrors"):
    sc = S.quad_field_scene(1920, 1080)
    if name == "mirrors":
        sc.planes["material"]["roughness"] = 0.0
        sc.planes["material"]["metalness"] = 1.0
    t1, r1, s1 = run(sc, 1)
    t2, r2, s2 = run(sc, 2)
    print(f"{name}: level 0: {r1 / 1e6:.1f} M rays, {t1:.2f} ms, {r1 / t1 / 1e6:.2f} Grays/s, {s1 / r1:.1f} steps/ray; "
          f"level 1: {(r2 - r1) / 1e6:.1f} M rays, {t2 - t1:.2f} ms, {(r2 - r1) / (t2 - t1) / 1e6:.2f} Grays/s, {(s2 - s1) / (r2 - r1):.1f} steps/ray", flush=True)

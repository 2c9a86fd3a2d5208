"""Diagnostic: throughput of the quad-field (BVH) workload against samples per pass and passes in flight.
Run from the repo root on the GPU box: python profiles/scan_pass_size.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
from pathtracer_amd import binding as B, scene as S
sc = S.quad_field_scene(1920, 1080)
for b, np_ in ((4, 1), (4, 2), (4, 3), (4, 4), (8, 2), (8, 3), (8, 4), (16, 2), (16, 4)):
    cfg = B.default_config(); cfg.samples_per_batch = b; cfg.profile_events = 0; cfg.concurrent_passes = np_
    with B.PtGpu(cfg) as g:
        g.upload_scene(sc)
        g.render(b); g.sync()
        t0 = time.time(); g.render(64); g.sync(); dt = time.time() - t0
        st = g.stats()
        print(b, "spp/pass", np_, "pipes", round(st["rays"] * 64 / (64 + b) / dt / 1e6, 1), "Mrays/s", round(dt * 1e3, 2), "ms", round(st["device_bytes"] / 2**30, 2), "GiB", flush=True)

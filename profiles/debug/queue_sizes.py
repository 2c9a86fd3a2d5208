"""Diagnostic: queue memory and pass size the library picks by default (1080p).  Usage (GPU box): python profiles/debug/queue_sizes.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathtracer_amd import binding as B, scene as S

for name, sc, cfg in (("sample", S.sample_scene(1920, 1080, 8.8888889), {}), ("sample, one pass in flight", S.sample_scene(1920, 1080, 8.8888889), {"concurrent_passes": 1}),
                      ("room depth 16", S.diffuse_room_scene(1920, 1080), {"max_depth": 16}),
                      ("quad field 100k", S.quad_field_scene(1920, 1080, grid=316, n_spheres=100), {}),
                      ("quad field 100k, two passes in flight", S.quad_field_scene(1920, 1080, grid=316, n_spheres=100), {"concurrent_passes": 2})):
    with B.PtGpu(**cfg) as g:
        g.upload_scene(sc)
        g.render(1); g.sync()
        st = g.stats()
        print(f"{name}: queue {st['queue_bytes'] / 2**30:.2f} GiB, level-0 records of the pass {st['level_records'][0]}, passes {st.get('passes')}", flush=True)

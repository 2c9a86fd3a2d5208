"""Diagnostic: BVH traversal steps per ray and lane utilisation on the quad field.  Needs a library built with
-DPT_TRAV_STATS selected through PTGPU_LIB.  Usage: PTGPU_LIB=... python profiles/traversal_stats.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import binding as B, scene as S
sc = S.quad_field_scene(1920, 1080)
cfg = B.default_config(); cfg.fuse_trace_shade = 0
with B.PtGpu(cfg) as g:
    g.upload_scene(sc)
    g.render(4); g.sync()
    st = g.stats()
    d = st["debug_cycles"]
    print("steps", d[0], "slots", d[1], "rays", d[2], "leaf steps", d[2], "steps/ray", d[0]/st["rays"], "lane util", d[0]/max(d[1],1), "max steps", d[3], "nodes", st["bvh_nodes"], st["rays"], st["kernel_ms"])

"""Diagnostic: records per level of the 1M-rectangle field at 1080p, depth 4 (one pass of 64 spp).
Usage (GPU box): python profiles/debug/level_rays.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathtracer_amd import binding as B, scene as S
with B.PtGpu(max_depth=4) as g:
    g.upload_scene(S.quad_field_scene(1920, 1080))
    g.render(64); g.sync()
    st = g.stats()
    print("records per level:", [r for r in st["level_records"][:6]], "samples", st["samples"], "rays", st["rays"], "shadow", st["shadow_rays"])

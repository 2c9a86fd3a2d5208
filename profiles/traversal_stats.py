"""Diagnostic: BVH traversal steps per ray and lane utilisation on the quad field, primary rays alone (max_depth 1)
and the whole path tree.  Needs a library built with -DPT_TRAV_STATS selected through PTGPU_LIB.
Usage: PTGPU_LIB=... python profiles/traversal_stats.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import binding as B, scene as S
sc = S.quad_field_scene(1920, 1080)
for depth in (1, 4):
    with B.PtGpu(max_depth=depth, fuse_trace_shade=0, concurrent_passes=1, profile_events=1) as g:
        g.upload_scene(sc)
        g.render(32); g.sync()
        st = g.stats()
        d = st["debug_cycles"]
        traced = st["rays"] - st["shadow_rays"]
        print(f"max_depth {depth}: traced rays {traced/1e6:.1f} M, lane-steps {d[0]/1e6:.0f} M ({d[0]/traced:.1f} per ray, {d[2]/traced:.1f} of them in leaves), "
              f"lane utilisation {d[0]/max(d[1],1):.3f}, longest ray {d[3]} steps, 64-byte units per ray {st['bvh_steps']/traced:.1f}, "
              f"trace {st['kernel_ms']['trace']:.2f} ms in {st['kernel_launches']['trace']} launches", flush=True)

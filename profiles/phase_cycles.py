"""Diagnostic: per-phase shader cycles of the fused kernel.  Needs a library built with -DPT_STAMPS
(hipcc ... -DPT_STAMPS -o libptgpu_stamps.so) selected through PTGPU_LIB; a normal build reports zeros.
Usage: PTGPU_LIB=... python profiles/phase_cycles.py [sample|room]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import binding as B, scene as S
wl = sys.argv[1] if len(sys.argv) > 1 else "sample"
sc = S.sample_scene(1920, 1080, camera_width=5.0 * 1920 / 1080) if wl == "sample" else S.diffuse_room_scene(1920, 1080)
cfg = B.default_config(); cfg.fuse_trace_shade = 1; cfg.max_depth = 4 if wl == "sample" else 16
with B.PtGpu(cfg) as g:
    g.upload_scene(sc)
    g.render(8); g.sync()
    st = g.stats()
    d = st["debug_cycles"]; tot = sum(d) or 1
    names = ["tile+fetch", "intersect", "hit+material", "append atomic", "brdf+node stores", "5", "child stores", "7"]
    print({n: round(100.0 * v / tot, 1) for n, v in zip(names, d)}, "cycles/ray", round(tot / st["rays"], 1))

"""How much of the fused kernel's time is material-arm divergence and how much the sphere tests?  Variants of the sample
scene with the same walls: (a) as shipped, (b) the three spheres made diffuse like the walls, (c) no spheres."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pathtracer_amd import binding as B, scene as S

def variant(kind):
    sc = S.sample_scene(1920, 1080, camera_width=5.0 * 1920 / 1080)
    white = S.Material.physically((1.0, 1.0, 1.0), 0.8, 0.0)
    if kind == "diffuse":
        sc.spheres = [S.Sphere(s.origin, s.radius, white) for s in sc.spheres]
    elif kind == "nospheres":
        sc.spheres = []
    elif kind == "noglass":
        sc.spheres = [S.Sphere(s.origin, s.radius, white if s.material.kind == S.MAT_TRANSLUCENT else s.material) for s in sc.spheres]
    return sc

for kind in ("shipped", "noglass", "diffuse", "nospheres"):
    with B.PtGpu(concurrent_passes=1, profile_events=1) as g:
        g.upload_scene(variant(kind))
        g.render(64); g.sync(); g.clear()
        g.render(256); g.sync()
        st = g.stats()
        ms = st["kernel_ms"]["trace_shade"]; recs = st["rays"] - st["shadow_rays"]
        lv = [round(x / st["level_records"][0], 3) for x in st["level_records"][:5]]
        print(f"{kind:10s} trace_shade {ms:8.2f} ms for {recs/1e6:8.1f} M traced records = {1e3*ms/(recs/1e6):6.2f} us/M; levels {lv}; "
              f"gather {st['kernel_ms']['gather']:.1f} resolve {st['kernel_ms']['resolve']:.1f}", flush=True)

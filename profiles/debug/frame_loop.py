"""The reference's window loop against the library (one sample per pixel per frame + the display bytes on the host every
frame, reference src/window.rs:127-229), and bulk rendering, over the handle's two sizing knobs: passes in flight and
samples per pass (= queue memory).  Usage (GPU box): python profiles/debug/frame_loop.py [sample|room|quads] [WxH] [PxB,PxB,...]
(P passes in flight x B samples per pass; 0x0 = the library's choice)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathtracer_amd import binding as B, scene as S  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "sample"
W, H = (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1920x1080").split("x"))
if workload == "sample":
    sc, cfg0 = S.sample_scene(W, H, camera_width=5.0 * W / H), {}
elif workload == "room":
    sc, cfg0 = S.diffuse_room_scene(W, H), {"max_depth": 16}
else:
    sc, cfg0 = S.quad_field_scene(W, H, grid=1000, n_spheres=1000), {}
P = W * H
BULK_SPP = 1024 if workload != "quads" else 256

print(f"{workload} {W}x{H}: passes x spp_per_pass -> queue GiB, upload s, bulk Grays/s (Msamples/s), frame loop ms (Msamples/s) sync | pipelined", flush=True)
combos = [tuple(int(x) for x in c.split("x")) for c in sys.argv[3].split(",")] if len(sys.argv) > 3 else \
    [(0, 0), (2, 16), (2, 11), (2, 8), (2, 4), (3, 8), (3, 5), (4, 8), (4, 5), (4, 3), (4, 2), (1, 32), (1, 16)]
for passes, batch in combos:
    t = time.perf_counter()
    with B.PtGpu(concurrent_passes=passes, samples_per_batch=batch, **cfg0) as gpu:
        gpu.upload_scene(sc)
        gpu.sync()
        t_up = time.perf_counter() - t
        gpu.render(64)
        gpu.sync()
        gpu.clear()
        t = time.perf_counter()
        gpu.render(BULK_SPP)
        gpu.sync()
        dt = time.perf_counter() - t
        st = gpu.stats()
        bulk = st["rays"] / dt / 1e9
        bulk_ms = P * BULK_SPP / dt / 1e6
        res = []
        for mode in ("sync", "pipelined"):
            step = (lambda: (gpu.render(1), gpu.present_pixels8())) if mode == "sync" else (lambda: gpu.frame(1))
            gpu.clear()
            for _ in range(16):
                step()
            gpu.frame(0)
            gpu.sync()
            gpu.clear()
            frames = 400
            t = time.perf_counter()
            for _ in range(frames):
                step()
            if mode == "pipelined":
                gpu.frame(0)
            dt = time.perf_counter() - t
            res.append(f"{1e3 * dt / frames:.3f} ({P * frames / dt / 1e6:.0f})")
        print(f"passes {passes} spp/pass {batch:2d}: {st['queue_bytes'] / 2**30:6.2f} GiB, upload {t_up:5.2f} s, bulk {bulk:6.2f} ({bulk_ms:.0f}), "
              f"frame {res[0]} | {res[1]}", flush=True)

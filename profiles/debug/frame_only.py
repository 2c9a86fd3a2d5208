"""Only the pipelined window loop (ptgpu_frame, 1 spp per frame), for a profiler: python3 profiles/debug/frame_only.py [WxH] [frames] [passes]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathtracer_amd import binding as B, scene as S  # noqa: E402

W, H = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1920x1080").split("x"))
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 400
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 0
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 4
with B.PtGpu(concurrent_passes=passes, samples_per_batch=batch) as gpu:
    gpu.upload_scene(S.sample_scene(W, H, camera_width=5.0 * W / H))
    for _ in range(16):
        gpu.frame(1)
    gpu.frame(0)
    gpu.sync()
    gpu.clear()
    t = time.perf_counter()
    for _ in range(frames):
        gpu.frame(1)
    gpu.frame(0)
    dt = time.perf_counter() - t
    print(f"{W}x{H} passes {passes}: {1e3 * dt / frames:.3f} ms per frame, {W * H * frames / dt / 1e6:.0f} Msamples/s", flush=True)

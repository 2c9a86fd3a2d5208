"""Host time of ptgpu_upload_scene for the 1M-rectangle field (flattening + BVH build + copies)."""
import sys, time
sys.path.insert(0, ".")
from pathtracer_amd import binding, scene as S
t = time.perf_counter(); sc = S.quad_field_scene(1920, 1080); print(f"generate {time.perf_counter()-t:.2f} s", flush=True)
with binding.PtGpu() as gpu:
    for rep in range(2):
        t = time.perf_counter(); gpu.upload_scene(sc); gpu.sync(); print(f"upload {time.perf_counter()-t:.2f} s", flush=True)

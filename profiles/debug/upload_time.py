"""How long ptgpu_upload_scene takes (it sizes and allocates the queue levels) and what the first render call adds."""
import sys, time
sys.path.insert(0, ".")
from pathtracer_amd import binding, scene as S
order = sys.argv[1:] or ["1024", "1080"]
scenes = {"1080": S.sample_scene(1920, 1080, camera_width=8.8888889), "1024": S.sample_scene(1024, 1024)}
for name in order:
    sc = scenes[name]
    t0 = time.perf_counter()
    gpu = binding.PtGpu()
    print(f"{name}: create {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
    for rep in range(4):
        t = time.perf_counter(); gpu.upload_scene(sc); gpu.sync(); t1 = time.perf_counter()
        gpu.render(64 if rep == 1 else 1); gpu.sync(); t2 = time.perf_counter()
        print(f"{name}: upload {1e3*(t1-t):.1f} ms, render {1e3*(t2-t1):.2f} ms, device {gpu.stats()['device_bytes']/2**30:.1f} GiB", flush=True)
    t = time.perf_counter(); gpu.close(); print(f"{name}: close {1e3*(time.perf_counter()-t):.1f} ms", flush=True)

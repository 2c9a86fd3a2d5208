"""Linear scan against BVH on the closed diffuse room with N spheres: where the library's automatic choice should switch."""
import sys, time
sys.path.insert(0, ".")
from pathtracer_amd import binding, scene as S

for n in (20, 40, 60, 80, 120, 160):
    sc = S.diffuse_room_scene(1920, 1080, n_spheres=n)
    row = []
    for accel in (1, 2):
        with binding.PtGpu(accel=accel, max_depth=8) as gpu:
            gpu.upload_scene(sc)
            gpu.render(16)
            gpu.sync()
            gpu.clear()
            t = time.perf_counter()
            gpu.render(64)
            gpu.sync()
            dt = time.perf_counter() - t
            row.append(gpu.stats()["rays"] / dt / 1e9)
    print(f"{n + 7:4d} primitives: linear scan {row[0]:6.2f} Grays/s   BVH {row[1]:6.2f} Grays/s", flush=True)

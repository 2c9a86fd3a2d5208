"""The reference's window loop against the library: one 1-spp render call and one display readback per frame.
Pageable destination against a page-locked one."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from pathtracer_amd import binding, scene as S

sc = S.sample_scene(1024, 1024)
with binding.PtGpu() as gpu:
    gpu.upload_scene(sc)
    n = 1024 * 1024 * 3
    pageable = np.empty(n, dtype=np.uint8)
    pinned = torch.empty(n, dtype=torch.uint8).pin_memory()
    for name, ptr in (("pageable", pageable.ctypes.data), ("page-locked", pinned.data_ptr())):
        for phase in ("render+read", "render only", "read only"):
            for _ in range(8):
                gpu.render(1); gpu._chk(gpu.lib.ptgpu_read_pixels8(gpu.h, ptr))
            gpu.sync()
            t = time.perf_counter()
            for _ in range(200):
                if phase != "read only": gpu.render(1)
                if phase != "render only": gpu._chk(gpu.lib.ptgpu_read_pixels8(gpu.h, ptr))
            gpu.sync()
            print(f"{name:12s} {phase:12s} {1e3 * (time.perf_counter() - t) / 200:.3f} ms per frame", flush=True)

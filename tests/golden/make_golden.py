"""Generates tests/golden/golden_v1.npz from the CPU oracle (oracle/).

The reference ships no golden vectors (it has no tests at all), so these are REGRESSION pins of the oracle's
output, not reference outputs: they catch accidental changes of the oracle and let the GPU tests compare against
committed data.  Deterministic math mode only (the libm mode depends on the machine's libm build).
Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import hdri_scene, materials_scene  # noqa: E402
from oracle import ptref as R  # noqa: E402
from pathtracer_amd import scene as S  # noqa: E402


def cases():
    """name -> (scene, max_depth, spp)"""
    return {
        "sample64": (S.sample_scene(64, 64), 4, 16),
        "materials": (materials_scene(40, 30, tone=S.TONE_REINHARD), 4, 8),
        "materials_exposure": (materials_scene(40, 30, tone=S.TONE_EXPOSURE, exposure=0.7), 4, 4),
        "hdri": (hdri_scene(32, 24), 4, 8),
        "room_depth16": (S.diffuse_room_scene(48, 27, n_spheres=12), 16, 4),
    }


def main():
    out = {}
    import ctypes as C
    st = C.c_uint32(314159265)
    out["xorshift32_first16"] = np.array([R.lib().ptref_xorshift32_next(C.byref(st)) for _ in range(16)], dtype=np.uint32)
    for name, (sc, depth, spp) in cases().items():
        osc = R.OracleScene(sc)
        opt = R.options(R.RNG_COUNTER, R.MATH_DET, max_depth=depth)
        out[f"{name}.ids_nojitter"] = osc.primary_ids(R.options(R.RNG_COUNTER, R.MATH_DET, sub_pixel_override=0), 0)
        out[f"{name}.ids_jitter_s1"] = osc.primary_ids(R.options(R.RNG_COUNTER, R.MATH_DET, sub_pixel_override=1), 1)
        p32, p8, rays = osc.render(opt, spp)
        out[f"{name}.pixels32"] = p32
        out[f"{name}.pixels8"] = p8
        out[f"{name}.rays"] = np.array([rays], dtype=np.uint64)
    # the reference's own RNG stream in serial draw order (deterministic math): pins the recursion/draw order
    sc = S.sample_scene(64, 64)
    p32, _, rays = R.OracleScene(sc).render(R.options(R.RNG_XORSHIFT, R.MATH_DET), 4)
    out["sample64.xorshift.pixels32"] = p32
    out["sample64.xorshift.rays"] = np.array([rays], dtype=np.uint64)
    path = os.path.join(ROOT, "tests", "golden", "golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()

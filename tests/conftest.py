import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from pathtracer_amd import scene as S  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import ptref
    ptref.lib()
    return ptref


@pytest.fixture(scope="session")
def binding():
    from pathtracer_amd import binding as b
    b.load_library()
    return b


# ---- scenes shared by the CPU and GPU tests -------------------------------------------------------------

def materials_scene(width=80, height=60, tone=S.TONE_REINHARD, exposure=1.0):
    """Every material kind (incl. Mirror and None, which the reference's parser cannot produce), an emissive
    sphere and two emissive planes, one of them facing up so terminal light samples do trace shadow rays."""
    sc = S.Scene(
        image_settings=S.ImageSettings(width, height, True),
        camera=S.Camera((0.5, 3.0, 12.0), (0.0, 2.0, 0.0), (0.0, 1.0, 0.0), 6.0, 4.5, 4.0, tone, exposure, 150.0),
        sky=S.Sky.constant((0.3, 0.4, 0.6)),
    )
    sc.add(S.Plane((-8.0, 0.0, 8.0), (16.0, 0.0, 0.0), (0.0, 0.0, -16.0), S.Material.physically((0.7, 0.7, 0.6), 0.6, 0.0)))
    sc.add(S.Plane((-3.0, 0.01, 3.0), (2.0, 0.0, 0.0), (0.0, 0.0, -2.0), S.Material.emissive((4.0, 3.0, 2.0))))   # faces up
    sc.add(S.Plane((-2.0, 7.0, -2.0), (4.0, 0.0, 0.0), (0.0, 0.0, 4.0), S.Material.emissive((6.0, 6.0, 6.0))))    # faces down
    sc.add(S.Plane((-8.0, 0.0, -6.0), (16.0, 0.0, 0.0), (0.0, 9.0, 1.0), S.Material.physically((0.9, 0.3, 0.2), 0.3, 1.0)))
    sc.add(S.Sphere((-2.5, 1.5, 0.0), 1.5, S.Material.mirror()))
    sc.add(S.Sphere((0.8, 1.2, 1.5), 1.2, S.Material.translucent(1.33)))
    sc.add(S.Sphere((3.2, 1.0, -0.5), 1.0, S.Material.none()))
    sc.add(S.Sphere((0.0, 4.2, -1.0), 0.9, S.Material.physically((0.2, 0.5, 0.9), 0.45, 0.5)))
    sc.add(S.Sphere((2.0, 3.5, 2.0), 0.5, S.Material.emissive((2.0, 2.0, 1.0))))
    return sc


def hdri_scene(width=64, height=48):
    """Exposure tone mapping and an equirectangular sky texture (synthetic 32x16)."""
    rng = np.random.RandomState(5)
    tex = rng.uniform(0.0, 3.0, (16, 32, 3)).astype(np.float32)
    sc = S.Scene(
        image_settings=S.ImageSettings(width, height, True),
        camera=S.Camera((0.0, 2.0, 9.0), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0), 4.0, 3.0, 3.0, S.TONE_EXPOSURE, 0.8, 100.0),
        sky=S.Sky.hdri("synthetic.hdr", tex),
    )
    sc.add(S.Plane((-6.0, 0.0, 6.0), (12.0, 0.0, 0.0), (0.0, 0.0, -12.0), S.Material.physically((0.8, 0.8, 0.8), 0.9, 0.0)))
    sc.add(S.Sphere((0.0, 1.0, 0.0), 1.0, S.Material.physically((0.95, 0.9, 0.8), 0.1, 1.0)))
    sc.add(S.Sphere((2.2, 0.8, 1.0), 0.8, S.Material.translucent(1.5)))
    return sc

"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol the header
declares, and its host-side helpers agree with the oracle.  No GPU compute is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, materials_scene
from pathtracer_amd import scene as S


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ptgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptgpu_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(binding):
    lib = binding.load_library()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ptgpu.h but not exported"


def test_struct_layouts_match_the_header(binding):
    # sizes follow from the header's field lists (all 4- or 8-byte members, natural alignment)
    assert C.sizeof(binding.PtVec3) == 12
    assert C.sizeof(binding.PtCamera) == 4 * 12 + 6 * 4
    assert C.sizeof(binding.PtSky) == 40
    assert C.sizeof(binding.PtConfig) == 15 * 4
    assert S.MATERIAL_DT.itemsize == 28 and S.SPHERE_DT.itemsize == 44 and S.PLANE_DT.itemsize == 76
    assert C.sizeof(binding.PtSceneDesc) == 12 + C.sizeof(binding.PtCamera) + 4 + C.sizeof(binding.PtSky) + 8 * 8


def test_default_config_matches_reference_constants(binding):
    cfg = binding.default_config()
    assert cfg.max_depth == 4            # reference src/tracer.rs:633
    assert cfg.seed == 314159265         # reference src/common.rs:28
    assert cfg.shard_count == 1 and cfg.shard_index == 0 and cfg.sub_pixel_override == -1


def test_no_gpu_means_loud_failure_not_fallback(binding):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(binding.PtGpuError) as e:
        binding.PtGpu()
    assert e.value.code == -2  # PT_ERR_NO_DEVICE


def test_host_helpers_agree_with_oracle(binding, oracle):
    """Camera::new and Plane::new as computed by the library's host code vs the oracle's, bit for bit."""
    rng = np.random.RandomState(9)
    for _ in range(200):
        cam = S.Camera(tuple(rng.uniform(-20, 20, 3)), tuple(rng.uniform(-5, 5, 3)), tuple(rng.uniform(-1, 1, 3)),
                       float(rng.uniform(1, 9)), float(rng.uniform(1, 9)), float(rng.uniform(0.5, 12)), 2, 0.5, 320.0)
        a, b = binding.camera_new(cam), oracle.camera_new(cam)
        assert bytes(a) == bytes(b)
    sc = materials_scene()
    sd, osc = binding.SceneDesc(sc), oracle.OracleScene(sc)
    for k in range(4):
        assert sd.arrays[k].tobytes() == osc.arrays[k].tobytes()
    q = S.quad_field_scene(grid=40, n_spheres=10)
    sd, osc = binding.SceneDesc(q), oracle.OracleScene(q)
    assert sd.arrays[1].tobytes() == osc.arrays[1].tobytes()
    n = sd.arrays[1]["normal"]
    assert np.all(n[:, 1] > 0.9)  # the height-field rectangles face up


def test_scene_routing_like_the_parser():
    sc = S.sample_scene()
    assert (len(sc.spheres), len(sc.planes), len(sc.emissive_spheres), len(sc.emissive_planes)) == (3, 6, 0, 1)
    # ids: spheres, emissive spheres, planes, emissive planes (reference src/scene.rs:266,280)
    assert sc.spheres[0].material.kind == S.MAT_TRANSLUCENT and sc.planes[2].origin == (-5.0, 0.0, -5.0)


def test_header_is_plain_c99():
    """The public header is consumed by C, cgo, bindgen: it must compile as strict C99 with nothing but <stdint.h>,
    and so must the C host that the GPU test runs."""
    import subprocess
    src = os.path.join(ROOT, "tests", "c", "abi_smoke.c")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), src])


def test_group_split_rule_partitions_the_sample_indices(binding):
    """ptgpu_group_shard_samples(first, count, k, n) = |{s in [first, first+count) : s % n == k}|: over k the counts
    partition every call, and over calls device k receives exactly its congruence class, in order."""
    lib = binding.load_library()
    rng = np.random.RandomState(3)
    for n in (1, 2, 3, 4, 7, 8):
        first = 0
        seen = [0] * n
        for _ in range(40):
            count = int(rng.randint(0, 20))
            got = [lib.ptgpu_group_shard_samples(first, count, k, n) for k in range(n)]
            want = [sum(1 for s in range(first, first + count) if s % n == k) for k in range(n)]
            assert got == want and sum(got) == count
            for k in range(n):
                seen[k] += got[k]
            first += count
        assert seen == [sum(1 for s in range(first) if s % n == k) for k in range(n)]
    assert lib.ptgpu_group_shard_samples(0, 10, 3, 3) == 0 and lib.ptgpu_group_shard_samples(0, 10, 0, 0) == 0
    assert lib.ptgpu_group_shard_samples(2**40, 2**33, 5, 8) == 2**30


def test_library_links_rccl_itself():
    """The multi-GPU reduction is the library's own (ncclReduce from pt_group.h), not a Python-side collective."""
    import subprocess
    out = subprocess.check_output(["readelf", "-d", os.path.join(ROOT, "pathtracer_amd", "libptgpu.so")], text=True)
    assert "librccl.so" in out

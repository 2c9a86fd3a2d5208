"""End to end through the C++ host: ptcli loads a `.scene` file the way the reference's window loop does
(reference src/window.rs:45-63,130-141), renders through libptgpu and writes a PPM; the image must equal what the
Python binding renders for the same scene (both sit on the same C ABI), and the oracle's result."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from pathtracer_amd import host_binding as H
from pathtracer_amd import scene as S

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "pathtracer_amd", "ptcli")


def _read_ppm(path):
    data = open(path, "rb").read()
    assert data[:2] == b"P6"
    parts = data.split(b"\n", 3)
    w, h = map(int, parts[1].split())
    return np.frombuffer(parts[3], dtype=np.uint8).reshape(h, w, 3)[::-1]  # back to bottom-up rows


def test_ptcli_renders_scene_file(binding, oracle, tmp_path):
    sc = S.sample_scene(96, 64, camera_width=7.5)
    path = str(tmp_path / "s.scene")
    open(path, "w").write("// generated\n" + S.to_text(sc))
    out = str(tmp_path / "o.ppm")
    r = subprocess.run([CLI, f"--path={path}", "--spp=6", f"--out={out}", "--frames=3"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("frame_time = ") == 3 and "96x64, 6 spp" in r.stdout
    img = _read_ppm(out)
    with binding.PtGpu() as gpu:
        gpu.upload_scene(sc)
        gpu.render(6)
        want = gpu.read_pixels8()
    assert np.array_equal(img, want)
    _, ref8, _ = oracle.OracleScene(sc).render(oracle.options(), 6)
    assert np.array_equal(img, ref8)


def test_ptcli_falls_back_to_default_scene(tmp_path):
    out = str(tmp_path / "d.ppm")
    r = subprocess.run([CLI, "--path=/no/such/file.scene", "--spp=1", f"--out={out}"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    assert "Could not read the file" in r.stdout and "Using the default scene." in r.stdout
    img = _read_ppm(out)
    # Scene::default(): no geometry, white sky, Exposure(1.0), iso 100 -> 1-exp(-1) -> gamma -> 207
    assert img.shape == (256, 256, 3) and np.all(img == img[0, 0]) and img[0, 0, 0] == 207


def test_hdri_sky_through_file_loader(binding, oracle, tmp_path):
    """scene file -> C++ parser -> .hdr decoded by the host loader -> GPU, against oracle with the same texture."""
    from test_scene_parser import _write_rle_hdr
    rng = np.random.RandomState(8)
    rgbe = np.concatenate([rng.randint(0, 256, (16, 32, 3)), rng.randint(126, 131, (16, 32, 1))], axis=2).astype(np.uint8)
    hdr = str(tmp_path / "sky.hdr")
    _write_rle_hdr(hdr, rgbe)
    tex = H.load_hdr(hdr)
    sc = S.Scene(image_settings=S.ImageSettings(64, 48, True),
                 camera=S.Camera((0.0, 2.0, 9.0), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0), 4.0, 3.0, 3.0, S.TONE_REINHARD, 1.0, 100.0),
                 sky=S.Sky.hdri("sky.hdr", None))  # the path grammar has no '-': keep it relative to the cwd
    sc.add(S.Plane((-6.0, 0.0, 6.0), (12.0, 0.0, 0.0), (0.0, 0.0, -12.0), S.Material.physically((0.8, 0.8, 0.8), 0.9, 0.0)))
    sc.add(S.Sphere((0.0, 1.0, 0.0), 1.0, S.Material.physically((0.95, 0.9, 0.8), 0.1, 1.0)))
    path = str(tmp_path / "h.scene")
    open(path, "w").write(S.to_text(sc))
    out = str(tmp_path / "h.ppm")
    r = subprocess.run([CLI, f"--path={path}", "--spp=4", f"--out={out}"], capture_output=True, text=True, timeout=120, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr + r.stdout
    img = _read_ppm(out)
    sc.sky.pixels = tex
    _, ref8, _ = oracle.OracleScene(sc).render(oracle.options(), 4)
    assert np.array_equal(img, ref8)


def test_ptcli_thin_lens_and_pfm_export(binding, tmp_path):
    """--thin-lens selects CameraSampler::thin_lens_ray; a .pfm output holds the accumulated mean as floats."""
    sc = S.sample_scene(64, 48, camera_width=6.6666667)
    path = str(tmp_path / "s.scene")
    open(path, "w").write(S.to_text(sc))
    out = str(tmp_path / "o.pfm")
    r = subprocess.run([CLI, f"--path={path}", "--spp=3", f"--out={out}", "--thin-lens"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    data = open(out, "rb").read()
    head, dims, scale, body = data.split(b"\n", 3)
    assert head == b"PF" and dims == b"64 48" and float(scale) < 0
    img = np.frombuffer(body, dtype="<f4").reshape(48, 64, 3)
    with binding.PtGpu(camera_model=1) as gpu:
        gpu.upload_scene(sc)
        gpu.render(3)
        p32, n = gpu.read_pixels32()
    assert np.array_equal(img, p32.reshape(48, 64, 3).astype(np.float32) / np.float32(3 * 255.0))


def test_plain_c_host_renders_the_same(binding):
    """tests/c/abi_smoke.c (C99, -pedantic, built by __graft_entry__.build()) drives the library through the public
    header alone; its accumulator checksum must equal the same scene rendered through the ctypes binding."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "c", "abi_smoke")
    r = subprocess.run([exe, "3"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    words = lines[0].split()
    got = dict(zip(words[0::2], (int(w) for w in words[1::2])))
    sc = S.Scene(image_settings=S.ImageSettings(32, 24, True),
                 camera=S.Camera((0.0, 2.0, 6.0), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0), 4.0, 3.0, 2.0, S.TONE_CLAMP, 1.0, 100.0),
                 sky=S.Sky.constant((0.5, 0.6, 0.7)))
    sc.add(S.Sphere((0.0, 1.0, 0.0), 1.0, S.Material.translucent(1.5)))
    sc.add(S.Plane((-4.0, 0.0, 4.0), (8.0, 0.0, 0.0), (0.0, 0.0, -8.0), S.Material.physically((0.8, 0.7, 0.6), 0.5, 0.0)))
    with binding.PtGpu() as gpu:
        gpu.upload_scene(sc)
        gpu.render(3)
        p32, n = gpu.read_pixels32()
        p8 = gpu.read_pixels8()
    assert got["samples"] == n == 3 and got["sum"] == int(p32.astype(np.uint64).sum()) and got["first_byte"] == int(p8.reshape(-1)[0])
    # the same frames through a one-device PtGroup (one sample + one present per frame, RCCL reduction at every present)
    gw = [ln for ln in lines if ln.startswith("group ")][0].split()   # (RCCL prints a version banner of its own)
    grp = dict(zip(gw[1::2], (int(w) for w in gw[2::2])))
    assert grp["devices"] == 1 and grp["exchange_bytes"] == 32 * 24 * 12
    for key in ("samples", "sum", "fnv", "first_byte"):
        assert grp[key] == got[key], key

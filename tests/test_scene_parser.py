"""The C++ `.scene` reader (pathtracer_amd/host/parser.cpp) against the reference's grammar (reference
src/parser.rs:50-649), its quirks included, and the `.hdr` loader against an independent decoder."""
import os
import struct

import numpy as np
import pytest

from conftest import ROOT
from pathtracer_amd import host_binding as H
from pathtracer_amd import scene as S

REF_SCENE = "/root/reference/scenes/sample/sample.scene"
REF_HDR = "/root/reference/scenes/sample/venice_sunset_1k.hdr"

MINI = """image {{ width = 64 height = 32 sub_pixel_sampling = {sps} }}
sky {{ constant {{ radiance = (0.5, 0.25, 1.0) }} }}
camera {{ position = (0.0, 1.0, 5.0) target = (0.0, 1.0, 0.0) up = (0.0, 1.0, 0.0) width = 2.0 height = 1.0 z_near = 1.5
  tone_mapping = {tm} iso = 100.0 }}
sphere {{
  origin = (0.0, 1.0, 0.0)
  radius = 1.0
  material = translucent {{ ior = 1.5 }}
}}
"""


def mini(sps="true", tm="clamp"):
    return MINI.format(sps=sps, tm=tm)


def test_generated_scene_files_parse_to_the_python_scenes(binding):
    for name, sc in (("sample.scene", S.sample_scene()), ("sample_1080p.scene", S.sample_scene(1920, 1080, 8.8888889)),
                     ("diffuse_room.scene", S.diffuse_room_scene())):
        text = open(os.path.join(ROOT, "scenes", name)).read()
        p = H.parse_scene(text)
        want = binding.SceneDesc(sc)
        assert (p.desc.width, p.desc.height, p.desc.sub_pixel_sampling) == (sc.image_settings.width, sc.image_settings.height, 1)
        assert bytes(p.desc.camera) == bytes(want.desc.camera)
        for got, ref in zip((p.spheres, p.planes, p.emissive_spheres, p.emissive_planes), want.arrays):
            assert got.tobytes() == ref.tobytes()
        # writer -> reader is the identity
        q = H.parse_scene(p.write())
        assert q.planes.tobytes() == p.planes.tobytes() and q.spheres.tobytes() == p.spheres.tobytes()
        assert bytes(q.desc.camera) == bytes(p.desc.camera)


@pytest.mark.skipif(not os.path.exists(REF_SCENE), reason="reference checkout not present")
def test_reference_sample_scene_is_table_s(binding):
    """The shipped sample.scene (read here as DATA, where the reference checkout exists) parses to exactly the
    scene the rest of this repository builds programmatically -- in comment-tolerant mode.  With the reference's own
    comment rule it does not parse at all (next test)."""
    p = H.parse_scene(open(REF_SCENE).read())
    want = binding.SceneDesc(S.sample_scene())
    assert (p.desc.width, p.desc.height) == (1024, 1024)
    assert bytes(p.desc.camera) == bytes(want.desc.camera)
    for got, ref in zip((p.spheres, p.planes, p.emissive_spheres, p.emissive_planes), want.arrays):
        assert got.tobytes() == ref.tobytes()
    assert tuple(p.desc.sky.radiance.__getattribute__(k) for k in "xyz") == tuple(np.float32(0.1) for _ in range(3))


def test_reference_comment_rule_leaves_the_last_character_behind():
    # reference src/parser.rs:467-476: `comment.next()` before `count()` consumes one character too few
    text = "// hello\n" + mini()
    assert H.parse_scene(text).desc.width == 64
    with pytest.raises(H.ParseError) as e:
        H.parse_scene(text, H.PARSE_REFERENCE_COMMENT_TAIL)
    assert (e.value.line, e.value.column) == (1, 8)  # stuck on the 'o' of "hello"
    assert "Expected \"camera\", \"sphere\", \"plane\" or \"sky\"." in e.value.message
    # an EMPTY comment is the one form the reference's rule survives
    assert H.parse_scene("//\n" + mini(), H.PARSE_REFERENCE_COMMENT_TAIL).desc.height == 32
    if os.path.exists(REF_SCENE):
        with pytest.raises(H.ParseError):
            H.parse_scene(open(REF_SCENE).read(), H.PARSE_REFERENCE_COMMENT_TAIL)


def test_bool_bug_false_parses_as_true():
    # reference src/parser.rs:591-593
    assert H.parse_scene(mini(sps="false")).desc.sub_pixel_sampling == 1
    assert H.parse_scene(mini(sps="true")).desc.sub_pixel_sampling == 1
    with pytest.raises(H.ParseError):
        H.parse_scene(mini(sps="maybe"))


@pytest.mark.parametrize("bad", ["1", ".5", "1e3", "1.", "--1.0", "+-1.0", "1.0e2", "0x1.0", "99999999999.0"])
def test_float_syntax_rejections(bad):
    # reference src/parser.rs:615-633: [sign]digits '.' digits only (a doubled sign panics there, errors here)
    with pytest.raises(H.ParseError):
        H.parse_scene(mini().replace("radius = 1.0", "radius = " + bad))


@pytest.mark.parametrize("good,val", [("+2.5", 2.5), ("-0.25", -0.25), ("007.500", 7.5), ("0.1", np.float32(0.1))])
def test_float_syntax_accepted(good, val):
    p = H.parse_scene(mini().replace("radius = 1.0", "radius = " + good))
    assert p.spheres["radius"][0] == np.float32(val)


def test_newline_rules():
    # sphere / plane / physically fields need a newline between them (reference src/parser.rs:302,308,387,389,405-417)
    one_line = mini().replace("origin = (0.0, 1.0, 0.0)\n  radius", "origin = (0.0, 1.0, 0.0) radius")
    with pytest.raises(H.ParseError):
        H.parse_scene(one_line)
    phys = "material = physically { reflectivity = (1.0, 1.0, 1.0) roughness = 0.5 metalness = 0.0 }"
    with pytest.raises(H.ParseError):
        H.parse_scene(mini().replace("material = translucent { ior = 1.5 }", phys))
    ok = "material = physically { reflectivity = (1.0, 1.0, 1.0)\n roughness = 0.5 // c\n metalness = 0.0 }"
    p = H.parse_scene(mini().replace("material = translucent { ior = 1.5 }", ok))
    assert p.spheres["material"]["kind"][0] == S.MAT_PHYSICALLY and p.spheres["material"]["roughness"][0] == 0.5
    # image / camera / sky blocks have no such rule: the whole MINI header is on one line each


def test_emissive_routing_defaults_and_tone_mapping():
    text = mini(tm="exposure { value = 0.75 }") + """
plane {
  origin = (0.0, 3.0, 0.0)
  u = (1.0, 0.0, 0.0)
  v = (0.0, 0.0, 1.0)
  material = emissive { radiance = (2.0, 3.0, 4.0) }
}
sphere {
  origin = (1.0, 1.0, 1.0)
  radius = 0.5
  material = emissive { radiance = (1.0, 1.0, 1.0) }
}
"""
    p = H.parse_scene(text)
    assert (len(p.spheres), len(p.planes), len(p.emissive_spheres), len(p.emissive_planes)) == (1, 0, 1, 1)
    assert tuple(p.emissive_planes["normal"][0]) == (0.0, -1.0, 0.0)   # Plane::new ran (u x v, normalised)
    assert p.desc.camera.tone_mapping == S.TONE_EXPOSURE and p.desc.camera.exposure == 0.75
    assert H.parse_scene(mini(tm="reinhard")).desc.camera.tone_mapping == S.TONE_REINHARD
    # defaults of parse_scene (reference src/parser.rs:55-61): 256x256, jitter on, white sky
    d = H.parse_scene("").desc
    assert (d.width, d.height, d.sub_pixel_sampling, d.sky.kind) == (256, 256, 1, S.SKY_CONSTANT)
    assert (d.sky.radiance.x, d.camera.iso, d.camera.tone_mapping) == (1.0, 100.0, S.TONE_CLAMP)
    # hdri sky keeps the path, texture pending
    h = H.parse_scene(mini().replace("constant { radiance = (0.5, 0.25, 1.0) }", "hdri { path = scenes/sample/sky_1k.hdr }"))
    assert h.desc.sky.kind == S.SKY_HDRI and h.sky_path == "scenes/sample/sky_1k.hdr" and not h.desc.sky.pixels


def test_unknown_block_reports_position():
    with pytest.raises(H.ParseError) as e:
        H.parse_scene(mini() + "\n\n  triangle { }\n")
    assert e.value.column == 3 and e.value.line == mini().count("\n") + 3


# ---- .hdr ---------------------------------------------------------------------------------------------------
def _rgbe_decode_reference(path):
    """Independent numpy decoder of the Radiance format (flat or new-style RLE scanlines)."""
    data = open(path, "rb").read()
    hdr_end = data.index(b"\n\n") + 2
    res_end = data.index(b"\n", hdr_end)
    parts = data[hdr_end:res_end].split()
    h, w = int(parts[1]), int(parts[3])
    pos = res_end + 1
    out = np.zeros((h, w, 4), dtype=np.uint8)
    for y in range(h):
        if data[pos] == 2 and data[pos + 1] == 2 and not data[pos + 2] & 0x80:
            pos += 4
            for k in range(4):
                x = 0
                while x < w:
                    c = data[pos]; pos += 1
                    if c > 128:
                        out[y, x:x + c - 128, k] = data[pos]; pos += 1; x += c - 128
                    else:
                        out[y, x:x + c, k] = np.frombuffer(data[pos:pos + c], dtype=np.uint8); pos += c; x += c
        else:
            out[y] = np.frombuffer(data[pos:pos + 4 * w], dtype=np.uint8).reshape(w, 4); pos += 4 * w
    e = out[..., 3].astype(np.int32)
    scale = np.where(e > 0, np.ldexp(np.float32(1.0), e - 136), np.float32(0.0)).astype(np.float32)
    return (out[..., :3].astype(np.float32) * scale[..., None]).astype(np.float32)


def _write_rle_hdr(path, rgbe):
    h, w, _ = rgbe.shape
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w))
        for y in range(h):
            f.write(bytes([2, 2, w >> 8, w & 255]))
            for k in range(4):
                row = rgbe[y, :, k]
                x = 0
                while x < w:
                    run = 1
                    while x + run < w and run < 127 and row[x + run] == row[x]:
                        run += 1
                    if run >= 3:
                        f.write(bytes([128 + run, int(row[x])])); x += run
                    else:
                        n = min(100, w - x)
                        f.write(bytes([n]) + row[x:x + n].tobytes()); x += n


def test_hdr_loader_rle_and_flat(tmp_path):
    rng = np.random.RandomState(4)
    rgbe = rng.randint(0, 256, (6, 40, 4)).astype(np.uint8)
    rgbe[2, 5:30, :] = rgbe[2, 5, :]      # long runs
    rgbe[3, :, 3] = 0                      # exponent 0 -> black
    p = str(tmp_path / "t.hdr")
    _write_rle_hdr(p, rgbe)
    got = H.load_hdr(p)
    assert got.shape == (6, 40, 3)
    assert np.array_equal(got, _rgbe_decode_reference(p))
    assert not got[3].any()
    flat = str(tmp_path / "flat.hdr")
    small = rgbe[:, :5]
    with open(flat, "wb") as f:  # width < 8: stb reads flat pixels
        f.write(b"#?RGBE\nFORMAT=32-bit_rle_rgbe\n\n-Y 6 +X 5\n" + small.tobytes())
    assert np.array_equal(H.load_hdr(flat), _rgbe_decode_reference(flat))
    assert H.load_hdr(str(tmp_path / "missing.hdr")) is None
    bad = str(tmp_path / "bad.hdr")
    open(bad, "wb").write(b"P6\n1 1\n255\n\0\0\0")
    assert H.load_hdr(bad) is None


@pytest.mark.skipif(not os.path.exists(REF_HDR), reason="reference checkout not present")
def test_hdr_loader_on_the_reference_sky():
    got = H.load_hdr(REF_HDR)
    assert got.shape == (512, 1024, 3) and np.isfinite(got).all() and got.min() >= 0.0
    assert np.array_equal(got, _rgbe_decode_reference(REF_HDR))


@pytest.mark.parametrize("seed", range(48))
def test_random_scenes_survive_text(binding, seed):
    """Random scenes (the generator of the GPU fuzz test) written in the `.scene` dialect and read back by the C++
    parser give the same primitive tables, camera and sky bit for bit; so does the C++ writer's own output."""
    from test_gpu_fuzz import random_scene
    sc, _, _ = random_scene(seed)
    plain = S.Material.physically((0.5, 0.5, 0.5), 0.5, 0.0)
    for prim in sc.spheres + sc.planes:          # the grammar has no Mirror / None
        if prim.material.kind in (S.MAT_MIRROR, S.MAT_NONE):
            prim.material = plain
    if sc.sky.kind == S.SKY_HDRI:                # a texture in memory has no path to print
        sc.sky = S.Sky.constant((0.25, 0.5, 0.75))
    sc.image_settings.sub_pixel_sampling = True  # the only value the reference parser can produce
    want = binding.SceneDesc(sc)
    p = H.parse_scene(S.to_text(sc))
    assert (p.desc.width, p.desc.height) == (sc.image_settings.width, sc.image_settings.height)
    cam_got, cam_want = bytes(p.desc.camera), bytes(want.desc.camera)
    if sc.camera.tone_mapping != S.TONE_EXPOSURE:   # clamp / reinhard carry no exposure value in the text
        off = type(p.desc.camera).exposure.offset
        cam_got, cam_want = cam_got[:off] + cam_got[off + 4:], cam_want[:off] + cam_want[off + 4:]
    assert cam_got == cam_want
    for got, ref in zip((p.spheres, p.planes, p.emissive_spheres, p.emissive_planes), want.arrays):
        assert got.tobytes() == ref.tobytes()
    q = H.parse_scene(p.write())
    for a, b in zip((q.spheres, q.planes, q.emissive_spheres, q.emissive_planes), (p.spheres, p.planes, p.emissive_spheres, p.emissive_planes)):
        assert a.tobytes() == b.tobytes()
    assert bytes(q.desc.camera) == bytes(p.desc.camera)


def test_procedural_quad_field_survives_text(binding):
    """The 1M-rectangle stress scene of the benchmark is generated from a seed; written as `.scene` text (here at
    100 x 100 rectangles + 100 spheres, ~2 MB) and read back by the C++ parser it gives the same tables bit for bit, so
    the Rust host can load the same workload from a file."""
    import time
    sc = S.quad_field_scene(64, 48, grid=100, n_spheres=100)
    text = S.to_text(sc)
    t = time.perf_counter()
    p = H.parse_scene(text)
    dt = time.perf_counter() - t
    want = binding.SceneDesc(sc)
    for got, ref in zip((p.spheres, p.planes, p.emissive_spheres, p.emissive_planes), want.arrays):
        assert got.tobytes() == ref.tobytes()
    assert len(p.planes) == 100 * 100 and len(text) / dt > 5e6   # (the parser reads tens of MB per second)

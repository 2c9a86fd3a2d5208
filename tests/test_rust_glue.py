"""The Rust side of the boundary (integration/rust) cannot be compiled here (no rustc), so it is checked as text:
`src/ptgpu.rs` must declare every export, struct field and constant of include/ptgpu.h with the same order and the
corresponding type, and the patches must apply to the reference's sources where those are present."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "ptgpu.h")
RUST = os.path.join(ROOT, "integration", "rust")

C_SCALAR = {"int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "float": "f32", "double": "f64", "int": "i32",
            "void": "c_void", "char": "c_char"}


def _strip_c(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def c_type_to_rust(ctype):
    """'const PtSphere *' -> '*const PtSphere', 'PtHandle **' -> '*mut *mut PtHandle', 'const uint8_t **' -> '*mut *const u8'."""
    t = ctype.strip()
    stars = t.count("*")
    t = t.replace("*", " ").split()
    const = "const" in t
    base = [w for w in t if w not in ("const", "struct")][0]
    rust = C_SCALAR.get(base, base)
    for level in range(stars):
        innermost = level == 0
        rust = ("*const " if (const and innermost) else "*mut ") + rust
    return rust


def parse_header():
    text = _strip_c(open(HEADER).read())
    structs, consts, funcs = {}, {}, {}
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            mm = re.match(r"(.*?[\s\*])(\w+(?:\s*\[[^\]]+\])?(?:\s*,\s*\*?\w+)*)\s*$", decl, flags=re.S)
            ctype, names = mm.group(1), mm.group(2)
            for name in names.split(","):
                name = name.strip()
                arr = re.match(r"(\w+)\s*\[([^\]]+)\]", name)
                if arr:
                    fields.append((arr.group(1), "[%s; %s]" % (c_type_to_rust(ctype), arr.group(2).strip())))
                else:
                    fields.append((name, c_type_to_rust(ctype)))
        structs[m.group(2)] = fields
    for m in re.finditer(r"enum\s*\{(.*?)\}", text, flags=re.S):
        for item in m.group(1).split(","):
            if "=" in item:
                k, v = item.split("=")
                consts[k.strip()] = int(v.strip(), 0)
    for m in re.finditer(r"#define\s+(PTGPU_\w+)\s+\((\d+)u\s*<<\s*(\d+)\)", text):
        consts[m.group(1)] = int(m.group(2)) << int(m.group(3))
    opaque = re.findall(r"typedef\s+struct\s+(\w+)\s+\1\s*;", text)
    for m in re.finditer(r"^([\w \*]+?[\s\*])(ptgpu_\w+)\s*\((.*?)\)\s*;", text, flags=re.S | re.M):
        ret = m.group(1).strip()
        args = []
        for a in m.group(3).split(","):
            a = " ".join(a.split())
            mm = re.match(r"(.*?[\s\*])(\w+)$", a)
            args.append((mm.group(2), c_type_to_rust(mm.group(1))))
        funcs[m.group(2)] = (None if ret == "void" else c_type_to_rust(ret), args)
    return structs, consts, funcs, opaque


def parse_rust():
    text = re.sub(r"//[^\n]*", "", open(os.path.join(RUST, "src", "ptgpu.rs")).read())
    structs, consts, funcs = {}, {}, {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^\)]*\)\]\s*)?pub struct (\w+)\s*\{(.*?)\}", text, flags=re.S):
        fields = []
        for f in m.group(2).split(",\n"):
            f = f.strip().rstrip(",")
            if f:
                mm = re.match(r"pub (\w+):\s*(.+)$", f, flags=re.S)
                fields.append((mm.group(1), " ".join(mm.group(2).split())))
        structs[m.group(1)] = fields
    for m in re.finditer(r"pub const (\w+):\s*\w+\s*=\s*([^;]+);", text):
        consts[m.group(1)] = eval(m.group(2).strip(), {"__builtins__": {}})
    opaque = re.findall(r"pub enum (\w+)\s*\{\s*\}", text)
    ext = re.search(r'extern "C"\s*\{(.*)\}', text, flags=re.S).group(1)
    for m in re.finditer(r"pub fn (\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", ext, flags=re.S):
        args = []
        for a in re.split(r",\s*(?=\w+\s*:)", " ".join(m.group(2).split())):
            if a.strip():
                name, typ = a.split(":", 1)
                args.append((name.strip().rstrip("_"), typ.strip()))
        funcs[m.group(1)] = (m.group(3).strip() if m.group(3) else None, args)
    return structs, consts, funcs, opaque


def test_parsers_see_the_whole_header():
    structs, consts, funcs, opaque = parse_header()
    assert set(structs) == {"PtVec3", "PtMaterial", "PtSphere", "PtPlane", "PtCamera", "PtSky", "PtSceneDesc", "PtConfig", "PtStats"}
    assert set(opaque) == {"PtHandle", "PtGroup"}
    assert len(funcs) >= 40 and "ptgpu_group_create" in funcs and "ptgpu_render_noalloc" in funcs
    assert funcs["ptgpu_present_pixels8"] == ("i32", [("h", "*mut PtHandle"), ("pixels", "*mut *const u8")])
    assert funcs["ptgpu_last_error"][0] == "*const c_char"
    assert consts["PT_ERR_INTERNAL"] == -7 and consts["PTGPU_NOALLOC_SCRATCH_BYTES"] == 64 << 20
    # every exported symbol of the header is also what the ctypes test sees: one list, three consumers
    text = _strip_c(open(HEADER).read())
    assert sorted(funcs) == sorted(set(re.findall(r"\b(ptgpu_[a-z0-9_]+)\s*\(", text)))


def test_rust_ffi_mirrors_the_header_exactly():
    hs, hc, hf, ho = parse_header()
    rs, rc, rf, ro = parse_rust()
    assert set(ro) == set(ho)
    assert set(rs) == set(hs), set(rs) ^ set(hs)
    for name, fields in hs.items():
        assert rs[name] == fields, f"struct {name}: header {fields} != rust {rs[name]}"
    for name, value in hc.items():
        assert name in rc, f"constant {name} missing in ptgpu.rs"
        assert rc[name] == value, name
    assert set(rf) == set(hf), f"exports differ: {set(rf) ^ set(hf)}"
    for name, (ret, args) in hf.items():
        assert rf[name][0] == ret, f"{name}: return {rf[name][0]} != {ret}"
        assert rf[name][1] == args, f"{name}: header {args} != rust {rf[name][1]}"


def test_wrapper_calls_only_declared_exports():
    _, _, rf, _ = parse_rust()
    used = set(re.findall(r"\b(ptgpu_\w+)\s*\(", open(os.path.join(RUST, "src", "gpu.rs")).read()))
    assert used and used <= set(rf), used - set(rf)
    build = open(os.path.join(RUST, "build.rs")).read()
    for flag in ("--offload-arch=gfx950", "-ffp-contract=off", "-lrccl", "cargo:rustc-link-lib=dylib=ptgpu"):
        assert flag in build


@pytest.mark.skipif(not os.path.isdir("/root/reference/src") or shutil.which("patch") is None,
                    reason="the reference checkout (or patch) is not present")
def test_patches_apply_to_the_reference(tmp_path):
    work = tmp_path / "crate"
    shutil.copytree("/root/reference/src", work / "src")
    shutil.copy("/root/reference/Cargo.toml", work / "Cargo.toml")
    for name in ("window.rs", "tracer.rs", "main.rs", "Cargo.toml"):
        with open(os.path.join(RUST, name + ".patch")) as f:
            subprocess.run(["patch", "-p1", "--quiet"], stdin=f, cwd=work, check=True)
    window = (work / "src" / "window.rs").read_text()
    code = re.sub(r"//[^\n]*", "", window)
    assert "worker_pool" not in code and "backbuffer" not in code and "worker::" not in code
    assert "renderer.frame(samples_per_frame)" in window and "let raw = shown;" in window and window.count("renderer.upload(") == 3
    assert "pub fn flatten(&self) -> ::ptgpu::PtCamera" in (work / "src" / "tracer.rs").read_text()
    assert "mod ptgpu;" in (work / "src" / "main.rs").read_text() and 'build = "build.rs"' in (work / "Cargo.toml").read_text()

// build.rs -- builds libptgpu.so with hipcc next to the crate and links it.
//
// Expects this repository checked out (or vendored) at <crate root>/ptgpu, i.e. the HIP sources at
// ptgpu/pathtracer_amd/csrc and the C header at ptgpu/include/ptgpu.h.  Set PTGPU_DIR to point elsewhere.
// Add to Cargo.toml:   [package] build = "build.rs"
use std::env;
use std::path::PathBuf;
use std::process::Command;

fn main() {
    let root = PathBuf::from(env::var("PTGPU_DIR").unwrap_or_else(|_| String::from("ptgpu")));
    let csrc = root.join("pathtracer_amd").join("csrc");
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let lib = out.join("libptgpu.so");
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| String::from("/opt/rocm/bin/hipcc"));
    let rocm_lib = env::var("ROCM_LIB").unwrap_or_else(|_| String::from("/opt/rocm/lib"));

    // -ffp-contract=off is not optional: the reference never fuses multiply-add, and the bit-exact parity of hit ids
    // and pixel sums with it depends on the same rounding.  gfx950 = MI355X; there is no other target.
    let status = Command::new(&hipcc)
        .args(&["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o"])
        .arg(&lib)
        .arg(csrc.join("ptgpu.hip"))
        .arg(format!("-L{}", rocm_lib))
        .arg("-lrccl") // the multi-GPU reduction (ptgpu_group_*) calls RCCL from inside the library
        .status()
        .expect("hipcc not found (set HIPCC)");
    assert!(status.success(), "hipcc failed to build libptgpu.so");

    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=ptgpu");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", out.display());
    println!("cargo:rerun-if-changed={}", csrc.display());
    println!("cargo:rerun-if-changed={}", root.join("include").join("ptgpu.h").display());
    println!("cargo:rerun-if-env-changed=PTGPU_DIR");
}

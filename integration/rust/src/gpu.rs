//! What `window.rs` calls instead of the worker pool: the scene flattened into the PODs of `ptgpu.rs` and a
//! renderer that owns either one GPU (PtHandle) or all GPUs of the node (PtGroup).
//!
//! The reference drives `WorkerPool::process(tile)` x tiles + `wait()` once per frame (src/window.rs:166-196) and
//! shows `Backbuffer.pixels8` (:205-210).  `GpuRenderer::frame` is those two steps pipelined (the window shows the
//! newest frame that has reached the host, at most two behind the one it has just enqueued); `render_frame` + `present` are the same two steps one after the other.
#![allow(dead_code)]

use std::ffi::CStr;
use std::ptr;

use common::*;
use ptgpu::*;
use scene::{Material, Plane, Scene, Sky, Sphere};

fn v(a: Vec3) -> PtVec3 {
    PtVec3 { x: a.x, y: a.y, z: a.z }
}

// src/scene.rs:84-92
fn material(m: &Material) -> PtMaterial {
    let zero = PtVec3 { x: 0.0, y: 0.0, z: 0.0 };
    match *m {
        Material::None => PtMaterial { kind: PT_MAT_NONE, color: zero, roughness: 0.0, metalness: 0.0, ior: 1.0 },
        Material::Emissive(c) => PtMaterial { kind: PT_MAT_EMISSIVE, color: v(c), roughness: 0.0, metalness: 0.0, ior: 1.0 },
        Material::Mirror => PtMaterial { kind: PT_MAT_MIRROR, color: zero, roughness: 0.0, metalness: 0.0, ior: 1.0 },
        Material::Translucent(ior) => PtMaterial { kind: PT_MAT_TRANSLUCENT, color: zero, roughness: 0.0, metalness: 0.0, ior: ior },
        Material::Physically(ref p) => PtMaterial {
            kind: PT_MAT_PHYSICALLY,
            color: v(p.reflectivity),
            roughness: p.roughness,
            metalness: p.metalness,
            ior: 1.0,
        },
    }
}

fn sphere(s: &Sphere) -> PtSphere {
    PtSphere { origin: v(s.origin), radius: s.radius, material: material(&s.material) }
}

// `normal` is what Plane::new computed with hmath (src/scene.rs:106): the library takes it as given
fn plane(p: &Plane) -> PtPlane {
    PtPlane { origin: v(p.origin), u: v(p.u), v: v(p.v), material: material(&p.material), normal: v(p.normal) }
}

/// The four primitive lists in the order `find_scene_hit` walks them (src/scene.rs:266,280); primitive ids in any
/// id buffer follow that order.  Keeps the Vecs alive for as long as the PtSceneDesc that points into them.
pub struct FlatScene {
    spheres: Vec<PtSphere>,
    planes: Vec<PtPlane>,
    emissive_spheres: Vec<PtSphere>,
    emissive_planes: Vec<PtPlane>,
    pub desc: PtSceneDesc,
}

pub fn flatten(scene: &Scene) -> FlatScene {
    let spheres: Vec<PtSphere> = scene.spheres.iter().map(sphere).collect();
    let planes: Vec<PtPlane> = scene.planes.iter().map(plane).collect();
    let emissive_spheres: Vec<PtSphere> = scene.emissive_spheres.iter().map(sphere).collect();
    let emissive_planes: Vec<PtPlane> = scene.emissive_planes.iter().map(plane).collect();
    let sky = match scene.sky {
        Sky::Constant(r) => PtSky { kind: PT_SKY_CONSTANT, radiance: v(r), pixels: ptr::null(), width: 0, height: 0 },
        // HDRI(_, None): texture not loaded yet, the miss branch returns white (src/tracer.rs:579)
        Sky::HDRI(_, None) => PtSky { kind: PT_SKY_HDRI, radiance: v(Vec3::zero()), pixels: ptr::null(), width: 0, height: 0 },
        Sky::HDRI(_, Some(ref tex)) => PtSky {
            kind: PT_SKY_HDRI,
            radiance: v(Vec3::zero()),
            pixels: tex.pixels.as_ptr(),
            width: tex.width as u64,
            height: tex.height as u64,
        },
    };
    let desc = PtSceneDesc {
        width: scene.image_settings.width as u32,
        height: scene.image_settings.height as u32,
        sub_pixel_sampling: if scene.image_settings.sub_pixel_sampling { 1 } else { 0 },
        camera: scene.camera.flatten(), // added to `impl Camera` by tracer.rs.patch (its fields are private)
        sky: sky,
        spheres: spheres.as_ptr(),
        n_spheres: spheres.len() as u64,
        planes: planes.as_ptr(),
        n_planes: planes.len() as u64,
        emissive_spheres: emissive_spheres.as_ptr(),
        n_emissive_spheres: emissive_spheres.len() as u64,
        emissive_planes: emissive_planes.as_ptr(),
        n_emissive_planes: emissive_planes.len() as u64,
    };
    FlatScene { spheres: spheres, planes: planes, emissive_spheres: emissive_spheres, emissive_planes: emissive_planes, desc: desc }
}

#[derive(Debug)]
pub struct GpuError {
    pub code: i32,
    pub message: String,
}

enum Backend {
    One(*mut PtHandle),
    Node(*mut PtGroup),
}

/// Replaces `worker::WorkerPool<WorkTile>` + `tracer::Backbuffer` (src/window.rs:104-116).
pub struct GpuRenderer {
    backend: Backend,
    pub width: u32,
    pub height: u32,
}

impl GpuRenderer {
    /// `all_gpus = false`: device 0 through a PtHandle.  `true`: every visible GPU through a PtGroup (a frame is split
    /// by 32x32 tile, larger calls by sample index; one RCCL exchange per image shown).  Without an MI355X this fails with PT_ERR_NO_DEVICE: there is no CPU path.
    pub fn new(all_gpus: bool) -> Result<GpuRenderer, GpuError> {
        unsafe {
            let mut cfg: PtConfig = std::mem::zeroed();
            ptgpu_default_config(&mut cfg); // depth 4 (src/tracer.rs:633), seed 314159265 (src/common.rs:28), one shard
            if all_gpus {
                let mut g: *mut PtGroup = ptr::null_mut();
                let rc = ptgpu_group_create(&cfg, ptr::null(), 0, &mut g);
                if rc != PT_OK {
                    return Err(GpuError { code: rc, message: String::from("ptgpu_group_create failed") });
                }
                Ok(GpuRenderer { backend: Backend::Node(g), width: 0, height: 0 })
            } else {
                let mut h: *mut PtHandle = ptr::null_mut();
                let rc = ptgpu_create(&cfg, &mut h);
                if rc != PT_OK {
                    return Err(GpuError { code: rc, message: String::from("ptgpu_create failed") });
                }
                Ok(GpuRenderer { backend: Backend::One(h), width: 0, height: 0 })
            }
        }
    }

    fn check(&self, rc: i32) -> Result<(), GpuError> {
        if rc == PT_OK {
            return Ok(());
        }
        let text = unsafe {
            let p = match self.backend {
                Backend::One(h) => ptgpu_last_error(h),
                Backend::Node(g) => ptgpu_group_last_error(g),
            };
            CStr::from_ptr(p).to_string_lossy().into_owned()
        };
        Err(GpuError { code: rc, message: text })
    }

    /// Every place the reference does `*scene = loaded; backbuffer.clear()` or patches the sky (src/window.rs:130-163):
    /// re-flattens, rebuilds the BVH, zeroes the accumulators.
    pub fn upload(&mut self, scene: &Scene) -> Result<(), GpuError> {
        let flat = flatten(scene);
        let rc = unsafe {
            match self.backend {
                Backend::One(h) => ptgpu_upload_scene(h, &flat.desc),
                Backend::Node(g) => ptgpu_group_upload_scene(g, &flat.desc),
            }
        };
        self.width = flat.desc.width;
        self.height = flat.desc.height;
        self.check(rc)
    }

    /// `backbuffer.clear()` without a scene change (src/tracer.rs:242-256).
    pub fn clear(&mut self) -> Result<(), GpuError> {
        let rc = unsafe {
            match self.backend {
                Backend::One(h) => ptgpu_clear(h),
                Backend::Node(g) => ptgpu_group_clear(g),
            }
        };
        self.check(rc)
    }

    /// One iteration of the window loop (src/window.rs:166-213): enqueues `samples` more samples per pixel (the
    /// reference's frame is 1) and returns the display bytes - `Backbuffer.pixels8` as glDrawPixels wants it: RGB8,
    /// bottom row first - of the newest frame that has arrived in the library's page-locked memory (at the oldest the
    /// frame two before this one), valid until the next call.  With all GPUs of the node each frame is split over them by 32x32 tile.
    pub fn frame(&mut self, samples: u32) -> Result<*const u8, GpuError> {
        let mut shown: *const u8 = ptr::null();
        let rc = unsafe {
            match self.backend {
                Backend::One(h) => ptgpu_frame(h, samples, &mut shown, ptr::null_mut()),
                Backend::Node(g) => ptgpu_group_frame(g, samples, &mut shown, ptr::null_mut()),
            }
        };
        self.check(rc).map(|_| shown)
    }

    /// One frame of the reference = one sample per pixel over the whole image (src/window.rs:166-196).
    pub fn render_frame(&mut self, samples: u32) -> Result<(), GpuError> {
        let rc = unsafe {
            match self.backend {
                Backend::One(h) => ptgpu_render(h, samples),
                Backend::Node(g) => ptgpu_group_render(g, samples),
            }
        };
        self.check(rc)
    }

    /// `Backbuffer.pixels8` as glDrawPixels wants it (RGB8, bottom row first, src/window.rs:205-210), in the
    /// library's page-locked buffer; valid until the next present / upload.
    pub fn present(&mut self) -> Result<*const u8, GpuError> {
        let mut shown: *const u8 = ptr::null();
        let rc = unsafe {
            match self.backend {
                Backend::One(h) => ptgpu_present_pixels8(h, &mut shown),
                Backend::Node(g) => ptgpu_group_present_pixels8(g, &mut shown),
            }
        };
        self.check(rc).map(|_| shown)
    }
}

impl Drop for GpuRenderer {
    fn drop(&mut self) {
        unsafe {
            match self.backend {
                Backend::One(h) => ptgpu_destroy(h),
                Backend::Node(g) => ptgpu_group_destroy(g),
            }
        }
    }
}

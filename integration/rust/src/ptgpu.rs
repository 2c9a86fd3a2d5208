//! Raw FFI of libptgpu: `include/ptgpu.h` field for field and export for export (2015-edition Rust, like the
//! crate it goes into).  `tests/test_rust_glue.py` of the ptgpu repository parses this file and the header and
//! fails on any missing symbol, field order or type mismatch, so edit both together.
#![allow(dead_code)]
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_void};

// PtStatus
pub const PT_OK: i32 = 0;
pub const PT_ERR_INVALID_ARGUMENT: i32 = -1;
pub const PT_ERR_NO_DEVICE: i32 = -2;
pub const PT_ERR_HIP: i32 = -3;
pub const PT_ERR_NO_SCENE: i32 = -4;
pub const PT_ERR_QUEUE_OVERFLOW: i32 = -5;
pub const PT_ERR_OUT_OF_MEMORY: i32 = -6;
pub const PT_ERR_INTERNAL: i32 = -7;

// src/scene.rs:84-92 `enum Material`
pub const PT_MAT_NONE: i32 = 0;
pub const PT_MAT_EMISSIVE: i32 = 1;
pub const PT_MAT_MIRROR: i32 = 2;
pub const PT_MAT_TRANSLUCENT: i32 = 3;
pub const PT_MAT_PHYSICALLY: i32 = 4;

// src/tracer.rs:43-48 `enum ToneMapping`
pub const PT_TONE_CLAMP: i32 = 0;
pub const PT_TONE_REINHARD: i32 = 1;
pub const PT_TONE_EXPOSURE: i32 = 2;

// src/scene.rs:71-75 `enum Sky`
pub const PT_SKY_CONSTANT: i32 = 0;
pub const PT_SKY_HDRI: i32 = 1;

pub const PT_ACCEL_AUTO: i32 = 0;
pub const PT_ACCEL_BRUTE_FORCE: i32 = 1;
pub const PT_ACCEL_BVH: i32 = 2;

pub const PT_HOST_TRACE_UNGATED: u32 = 1;
pub const PT_TILE: u32 = 32;

pub const PT_KERNEL_RAYGEN: usize = 0;
pub const PT_KERNEL_TRACE: usize = 1;
pub const PT_KERNEL_SHADE: usize = 2;
pub const PT_KERNEL_TRACE_SHADE: usize = 3;
pub const PT_KERNEL_TERMINAL: usize = 4;
pub const PT_KERNEL_GATHER: usize = 5;
pub const PT_KERNEL_RESOLVE: usize = 6;
pub const PT_KERNEL_COUNT: usize = 8;

pub const PTGPU_NOALLOC_SCRATCH_BYTES: u32 = 64 << 20;

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PtVec3 {
    pub x: f32,
    pub y: f32,
    pub z: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PtMaterial {
    pub kind: i32,
    pub color: PtVec3,
    pub roughness: f32,
    pub metalness: f32,
    pub ior: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PtSphere {
    pub origin: PtVec3,
    pub radius: f32,
    pub material: PtMaterial,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PtPlane {
    pub origin: PtVec3,
    pub u: PtVec3,
    pub v: PtVec3,
    pub material: PtMaterial,
    pub normal: PtVec3,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PtCamera {
    pub plane_origin: PtVec3,
    pub plane_u: PtVec3,
    pub plane_v: PtVec3,
    pub position: PtVec3,
    pub width: f32,
    pub height: f32,
    pub z_near: f32,
    pub tone_mapping: i32,
    pub exposure: f32,
    pub iso: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PtSky {
    pub kind: i32,
    pub radiance: PtVec3,
    pub pixels: *const f32,
    pub width: u64,
    pub height: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PtSceneDesc {
    pub width: u32,
    pub height: u32,
    pub sub_pixel_sampling: i32,
    pub camera: PtCamera,
    pub sky: PtSky,
    pub spheres: *const PtSphere,
    pub n_spheres: u64,
    pub planes: *const PtPlane,
    pub n_planes: u64,
    pub emissive_spheres: *const PtSphere,
    pub n_emissive_spheres: u64,
    pub emissive_planes: *const PtPlane,
    pub n_emissive_planes: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PtConfig {
    pub device: i32,
    pub seed: u32,
    pub max_depth: i32,
    pub sub_pixel_override: i32,
    pub shard_index: u32,
    pub shard_count: u32,
    pub samples_per_batch: u32,
    pub branch_capacity: u32,
    pub accel: i32,
    pub fuse_trace_shade: i32,
    pub profile_events: i32,
    pub concurrent_passes: u32,
    pub camera_model: i32,
    pub focus_distance: f32,
    pub lens_aperture: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PtStats {
    pub samples: u64,
    pub rays: u64,
    pub shadow_rays: u64,
    pub records: u64,
    pub terminal_records: u64,
    pub bvh_nodes: u64,
    pub bvh_leaf_prims: u64,
    pub bvh_steps: u64,
    pub kernel_launches: [u64; PT_KERNEL_COUNT],
    pub kernel_ms: [f64; PT_KERNEL_COUNT],
    pub device_bytes: u64,
    pub queue_growth: u64,
    pub debug_cycles: [u64; 8],
    pub bvh_depth: u64,
    pub bvh_binary_depth: u64,
    pub bvh_forced_median_splits: u64,
    pub bvh_level_fold: u64,
    pub queue_bytes: u64,
    pub level_records: [u64; 25],
    pub samples_per_pass: u64,
    pub passes_in_flight: u64,
}

pub enum PtHandle {}
pub enum PtGroup {}

#[link(name = "ptgpu")]
extern "C" {
    pub fn ptgpu_default_config(cfg: *mut PtConfig);
    pub fn ptgpu_create(cfg: *const PtConfig, out: *mut *mut PtHandle) -> i32;
    pub fn ptgpu_destroy(h: *mut PtHandle);
    pub fn ptgpu_upload_scene(h: *mut PtHandle, scene: *const PtSceneDesc) -> i32;
    pub fn ptgpu_clear(h: *mut PtHandle) -> i32;
    pub fn ptgpu_render(h: *mut PtHandle, n_samples: u32) -> i32;
    pub fn ptgpu_sync(h: *mut PtHandle) -> i32;
    pub fn ptgpu_render_noalloc(h: *mut PtHandle, n_samples: u32) -> i32;
    pub fn ptgpu_frame(h: *mut PtHandle, n_samples: u32, pixels: *mut *const u8, num_samples: *mut u64) -> i32;
    pub fn ptgpu_render_tiles(h: *mut PtHandle, first_sample: u64, n_samples: u32) -> i32;
    pub fn ptgpu_tile_shard(tile_x: u32, tile_y: u32, tiles_x: u32, n: u32) -> u32;
    pub fn ptgpu_read_pixels32(h: *mut PtHandle, out: *mut u32, num_samples: *mut u64) -> i32;
    pub fn ptgpu_read_pixels8(h: *mut PtHandle, out: *mut u8) -> i32;
    pub fn ptgpu_present_pixels8(h: *mut PtHandle, pixels: *mut *const u8) -> i32;
    pub fn ptgpu_render_primary_ids(h: *mut PtHandle, sample_index: u32, out: *mut u32) -> i32;
    pub fn ptgpu_device_pixels32(h: *mut PtHandle, device_ptr: *mut *mut c_void, n_words: *mut u64) -> i32;
    pub fn ptgpu_set_num_samples(h: *mut PtHandle, num_samples: u64) -> i32;
    pub fn ptgpu_export_pixels32(h: *mut PtHandle, dst_device: *mut c_void) -> i32;
    pub fn ptgpu_import_pixels32(h: *mut PtHandle, src_device: *const c_void, num_samples: u64) -> i32;
    pub fn ptgpu_stream(h: *mut PtHandle, stream: *mut *mut c_void) -> i32;
    pub fn ptgpu_get_stats(h: *mut PtHandle, out: *mut PtStats) -> i32;
    pub fn ptgpu_bvh_build_stats(scene: *const PtSceneDesc, out: *mut PtStats) -> i32;
    pub fn ptgpu_bvh_host_trace(scene: *const PtSceneDesc, n_rays: u64, rays: *const f32, prims: *mut u32, ts: *mut f32, faults: *mut u64) -> i32;
    pub fn ptgpu_bvh_host_trace_flags(scene: *const PtSceneDesc, n_rays: u64, rays: *const f32, prims: *mut u32, ts: *mut f32, faults: *mut u64, flags: u32) -> i32;
    pub fn ptgpu_last_error(h: *mut PtHandle) -> *const c_char;

    pub fn ptgpu_group_create(cfg: *const PtConfig, devices: *const i32, n_devices: u32, out: *mut *mut PtGroup) -> i32;
    pub fn ptgpu_group_destroy(g: *mut PtGroup);
    pub fn ptgpu_group_upload_scene(g: *mut PtGroup, scene: *const PtSceneDesc) -> i32;
    pub fn ptgpu_group_clear(g: *mut PtGroup) -> i32;
    pub fn ptgpu_group_render(g: *mut PtGroup, n_samples: u32) -> i32;
    pub fn ptgpu_group_frame(g: *mut PtGroup, n_samples: u32, pixels: *mut *const u8, num_samples: *mut u64) -> i32;
    pub fn ptgpu_group_sync(g: *mut PtGroup) -> i32;
    pub fn ptgpu_group_read_pixels32(g: *mut PtGroup, out: *mut u32, num_samples: *mut u64) -> i32;
    pub fn ptgpu_group_read_pixels8(g: *mut PtGroup, out: *mut u8) -> i32;
    pub fn ptgpu_group_present_pixels8(g: *mut PtGroup, pixels: *mut *const u8) -> i32;
    pub fn ptgpu_group_get_stats(g: *mut PtGroup, out: *mut PtStats) -> i32;
    pub fn ptgpu_group_exchange_stats(g: *mut PtGroup, bytes: *mut u64, ms: *mut f64) -> i32;
    pub fn ptgpu_group_shard_samples(first: u64, count: u64, k: u32, n: u32) -> u64;
    pub fn ptgpu_group_size(g: *const PtGroup) -> u32;
    pub fn ptgpu_group_member(g: *mut PtGroup, k: u32) -> *mut PtHandle;
    pub fn ptgpu_group_last_error(g: *mut PtGroup) -> *const c_char;

    pub fn ptgpu_camera_new(position: PtVec3, target: PtVec3, up: PtVec3, width: f32, height: f32, z_near: f32,
                            tone_mapping: i32, exposure: f32, iso: f32, out: *mut PtCamera);
    pub fn ptgpu_plane_normal(u: PtVec3, v: PtVec3, out: *mut PtVec3);
    pub fn ptgpu_plane_normals(planes: *mut PtPlane, n: u64);
    pub fn ptgpu_eval_math(h: *mut PtHandle, fn_: i32, a: *const f32, b: *const f32, out: *mut f32, n: u64) -> i32;
    pub fn ptgpu_selftest_gamma(h: *mut PtHandle, mismatches: *mut u64) -> i32;
}

//! `include/rt_amd.h` for Rust: every struct `#[repr(C)]` with the header's field order, every export
//! declared.  Sizes and offsets are asserted on the C side (static_asserts in csrc/rt_api.cpp) and
//! repeated here as compile-time checks.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

pub const RT_OK: c_int = 0;
pub const RT_ERR_INVALID: c_int = -1;
pub const RT_ERR_UNSUPPORTED: c_int = -2;
pub const RT_ERR_DEVICE: c_int = -3;
pub const RT_ERR_NOMEM: c_int = -4;

// rt_leaf_type
pub const RT_EMPTY: i32 = 0;
pub const RT_SPHERE: i32 = 1;
pub const RT_PLANE: i32 = 2;
pub const RT_CUBE: i32 = 3;
pub const RT_CYLINDER: i32 = 4;
pub const RT_CONE: i32 = 5;
pub const RT_TRIANGLE: i32 = 6;
pub const RT_SMOOTH_TRIANGLE: i32 = 7;
// rt_pattern_type
pub const RT_PAT_NULL: i32 = 0;
pub const RT_PAT_POINT: i32 = 1;
pub const RT_PAT_IDENTITY: i32 = 2;
pub const RT_PAT_STRIPE: i32 = 3;
pub const RT_PAT_RING: i32 = 4;
pub const RT_PAT_CHECKER: i32 = 5;
pub const RT_PAT_GRADIENT: i32 = 6;
pub const RT_PAT_BLEND: i32 = 7;
// rt_csg_op
pub const RT_CSG_UNION: i32 = 0;
pub const RT_CSG_INTERSECTION: i32 = 1;
pub const RT_CSG_DIFFERENCE: i32 = 2;
// flags of rt_render / rt_render_host
pub const RT_FLAG_COUNTERS: u32 = 1;
pub const RT_FLAG_ASYNC: u32 = 2;
pub const RT_FLAG_COMPACT: u32 = 4;
pub const RT_FLAG_BANDS: u32 = 8;
pub const RT_FLAG_BANDS_16: u32 = 16;
pub const RT_FLAG_BANDS_8: u32 = 32;
// build_flags of rt_scene_create_with
pub const RT_BUILD_DEVICE: u32 = 1;
pub const RT_BUILD_HOST: u32 = 2;

#[repr(C)] #[derive(Clone, Copy)]
pub struct rt_xform { pub ray_inv: [f64; 16], pub full_inv: [f64; 16], pub fwd: [f64; 16] }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_triangle { pub p1: [f64; 3], pub e1: [f64; 3], pub e2: [f64; 3], pub normal: [f64; 3],
                         pub n1: [f64; 3], pub n2: [f64; 3], pub n3: [f64; 3] }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_leaf { pub ty: i32, pub closed: i32, pub material: i32, pub xform: i32, pub gate: i32, pub tri: i32,
                     pub csg: i32, pub csg_left: u32, pub minimum: f64, pub maximum: f64, pub plane_normal: [f64; 3] }
#[repr(C)] #[derive(Clone, Copy)]
pub struct rt_gate { pub parent: i32, pub csg: i32, pub to_local: [f64; 16], pub bmin: [f64; 3], pub bmax: [f64; 3] }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_csg { pub op: i32, pub parent: i32, pub gate: i32, pub first_leaf: i32, pub n_leaves: i32, pub reserved: i32 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_material { pub color: [f64; 3], pub ambient: f64, pub diffuse: f64, pub specular: f64, pub shininess: f64,
                         pub reflective: f64, pub refractive_index: f64, pub transparency: f64,
                         pub pattern: i32, pub reserved: i32 }
#[repr(C)] #[derive(Clone, Copy)]
pub struct rt_pattern { pub ty: i32, pub first: i32, pub second: i32, pub reserved: i32,
                        pub inv: [f64; 16], pub a: [f64; 3], pub b: [f64; 3] }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_light { pub intensity: [f64; 3], pub position: [f64; 3] }
#[repr(C)] #[derive(Clone, Copy)]
pub struct rt_camera { pub hsize: u32, pub vsize: u32, pub half_width: f64, pub half_height: f64, pub pixel_size: f64,
                       pub inv: [f64; 16] }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_chain_ref { pub first: i32, pub count: i32 }
#[repr(C)]
pub struct rt_scene_desc {
    pub light: rt_light,
    pub n_leaves: i32, pub n_gates: i32, pub n_xforms: i32, pub n_triangles: i32, pub n_materials: i32, pub n_patterns: i32,
    pub n_csg: i32, pub n_chain_mats: i32,
    pub csg: *const rt_csg,
    pub leaves: *const rt_leaf,
    pub gates: *const rt_gate,
    pub xforms: *const rt_xform,
    pub triangles: *const rt_triangle,
    pub materials: *const rt_material,
    pub patterns: *const rt_pattern,
    pub chain_mats: *const f64,            // n_chain_mats x 16, row-major
    pub xform_chain: *const rt_chain_ref,  // n_xforms entries or null
    pub gate_chain: *const rt_chain_ref,   // n_gates entries or null
}
#[repr(C)] #[derive(Clone, Copy, Default, Debug)]
pub struct rt_stats {
    pub primary: u64, pub shadow: u64, pub reflect: u64, pub refract: u64,
    pub render_ms: f64, pub kernel_ms: f64, pub kernel_launches: u64, pub generations: u64,
    pub node_visits: u64, pub tri_tests: u64, pub quadric_tests: u64, pub gate_tests: u64,
    pub shadow_node_visits: u64, pub shadow_tri_tests: u64, pub shadow_quadric_tests: u64, pub shadow_gate_tests: u64,
}
#[repr(C)] pub struct rt_scene { _private: [u8; 0] }

// the sizes the C side asserts (csrc/rt_api.cpp RT_ABI_SIZE)
const _: [(); 384] = [(); std::mem::size_of::<rt_xform>()];
const _: [(); 168] = [(); std::mem::size_of::<rt_triangle>()];
const _: [(); 72] = [(); std::mem::size_of::<rt_leaf>()];
const _: [(); 184] = [(); std::mem::size_of::<rt_gate>()];
const _: [(); 24] = [(); std::mem::size_of::<rt_csg>()];
const _: [(); 88] = [(); std::mem::size_of::<rt_material>()];
const _: [(); 192] = [(); std::mem::size_of::<rt_pattern>()];
const _: [(); 48] = [(); std::mem::size_of::<rt_light>()];
const _: [(); 160] = [(); std::mem::size_of::<rt_camera>()];
const _: [(); 8] = [(); std::mem::size_of::<rt_chain_ref>()];
const _: [(); 160] = [(); std::mem::size_of::<rt_scene_desc>()];
const _: [(); 128] = [(); std::mem::size_of::<rt_stats>()];

extern "C" {
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_device_count() -> c_int;
    pub fn rt_scene_create(desc: *const rt_scene_desc, device: c_int, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_destroy(scene: *mut rt_scene);
    pub fn rt_scene_create_with(desc: *const rt_scene_desc, device: c_int, build_flags: u32, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_build_info(scene: *const rt_scene, device_builds: *mut u64, device_build_seconds: *mut f64) -> c_int;
    pub fn rt_scene_update(scene: *mut rt_scene, desc: *const rt_scene_desc) -> c_int;
    pub fn rt_scene_rebuilds(scene: *const rt_scene, rebuilds: *mut u64) -> c_int;
    pub fn rt_scene_clone(scene: *const rt_scene, device: c_int, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_footprint(scene: *const rt_scene, scene_bytes: *mut u64, build_seconds: *mut f64) -> c_int;
    pub fn rt_render(scene: *mut rt_scene, camera: *const rt_camera, part: c_int, n_parts: c_int, flags: u32,
                     rgb8_dev: *mut u8, rgb64_dev: *mut f64, hit_leaf_dev: *mut i32, stats: *mut rt_stats) -> c_int;
    pub fn rt_scene_wait(scene: *mut rt_scene, total: *mut rt_stats) -> c_int;
    pub fn rt_render_frames(scene: *mut rt_scene, cameras: *const rt_camera, n_frames: c_int, part: c_int, n_parts: c_int,
                            flags: u32, rgb8_dev: *const *mut u8) -> c_int;
    pub fn rt_scene_stream(scene: *mut rt_scene, hip_stream: *mut *mut c_void) -> c_int;
    pub fn rt_download(scene: *mut rt_scene, src_dev: *const c_void, dst_host: *mut c_void, bytes: u64,
                       after_stream: *mut c_void) -> c_int;
    pub fn rt_scene_copy_stream(scene: *mut rt_scene, hip_stream: *mut *mut c_void) -> c_int;
    pub fn rt_download_2d(scene: *mut rt_scene, src_dev: *const c_void, dst_host: *mut c_void, row_bytes: u64, rows: u64,
                          pitch: u64, after_stream: *mut c_void) -> c_int;
    pub fn rt_copy_fence(scene: *mut rt_scene, slot: u32) -> c_int;
    pub fn rt_copy_wait(scene: *mut rt_scene, slot: u32) -> c_int;
    pub fn rt_host_alloc(bytes: u64) -> *mut c_void;
    pub fn rt_host_free(p: *mut c_void);
    pub fn rt_host_register(p: *mut c_void, bytes: u64) -> c_int;
    pub fn rt_host_unregister(p: *mut c_void) -> c_int;
    pub fn rt_device_alloc(device: c_int, bytes: u64, out: *mut *mut c_void) -> c_int;
    pub fn rt_device_free(device: c_int, p: *mut c_void);
    pub fn rt_render_host(scene: *mut rt_scene, camera: *const rt_camera, part: c_int, n_parts: c_int, flags: u32,
                          rgb8: *mut u8, rgb64: *mut f64, hit_leaf: *mut i32, stats: *mut rt_stats) -> c_int;
    pub fn rt_ray_for_pixel(scene: *mut rt_scene, camera: *const rt_camera, n: u64, xy: *const u32, rays6: *mut f64) -> c_int;
    pub fn rt_color_at(scene: *mut rt_scene, n: u64, rays6: *const f64, remaining: c_int, flags: u32, rgb: *mut f64,
                       leaf: *mut i32, stats: *mut rt_stats) -> c_int;
    pub fn rt_intersect_hit(scene: *mut rt_scene, n: u64, rays6: *const f64, t: *mut f64, leaf: *mut i32) -> c_int;
    pub fn rt_is_shadowed(scene: *mut rt_scene, n: u64, points3: *const f64, shadowed: *mut u8) -> c_int;
    pub fn rt_ppm_encode(device: c_int, rgb8_dev: *const u8, width: u64, height: u64, out_dev: *mut c_char, capacity: u64,
                         length: *mut u64) -> c_int;
    pub fn rt_ppm_encode_host(device: c_int, rgb8_host: *const u8, width: u64, height: u64, out_host: *mut c_char,
                              capacity: u64, length: *mut u64) -> c_int;
    pub fn rt_ppm_encode_static(device: c_int, rgb8_dev: *const u8, width: u64, height: u64, out_dev: *mut c_char,
                                capacity: u64, length: *mut u64) -> c_int;
    pub fn rt_ppm_static_max_pixels() -> u64;
    pub fn rt_ppm_max_bytes(width: u64, height: u64) -> u64;
    pub fn rt_selftest_band_grid(n_bands: c_int, n_dirs: c_int, seed: u64, grid_n: *mut c_int, entries: *mut u64) -> c_int;
    pub fn rt_selftest_band_grid_device(n_bands: c_int, seed: u64, device: c_int, grid_n: *mut c_int, entries: *mut u64,
                                        host_seconds: *mut f64, device_seconds: *mut f64) -> c_int;
    pub fn rt_selftest_bvh_build(n_items: c_int, seed: u64, threads: c_int, hash: *mut u64, n_nodes: *mut u64,
                                 seconds: *mut f64) -> c_int;
    pub fn rt_selftest_world_normal(full_inv16: *const f64, local_normal3: *const f64, out3: *mut f64) -> c_int;
}

/// A negative status becomes a panic with the library's message: the reference panics on every failure
/// of this path (`unwrap` / `expect` in src/parallel.rs).
pub fn check(rc: c_int) {
    if rc != RT_OK {
        let msg = unsafe { std::ffi::CStr::from_ptr(rt_last_error()) }.to_string_lossy().into_owned();
        panic!("rt_amd status {}: {}", rc, msg);
    }
}

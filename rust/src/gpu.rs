//! `parallel_render` on MI355X GPUs: flattens a `World` into the plain arrays of `ffi::rt_scene_desc`,
//! renders bands of 32 pixel rows on `jobs` GPUs and saves the PPM.  Replaces src/parallel.rs:183-204;
//! called from src/main.rs:71,97 with the same arguments.
//!
//! Not compiled in the repository this file ships in (no Rust toolchain there); the C++ host
//! (ray-tracer_amd/host/rt_host.cpp: Flattener, parallel_render) is the tested twin of this file.
use std::io::Write;
use std::path::Path;

use crate::camera::Camera;
use crate::ffi::*;
use crate::light::Material;
use crate::matrix::Matrix4D;
use crate::pattern::{Pattern, PatternType};
use crate::shape::{Shape, ShapeType};
use crate::world::World;

#[derive(Default)]
pub struct FlatWorld {
    pub leaves: Vec<rt_leaf>,
    pub gates: Vec<rt_gate>,
    pub csg: Vec<rt_csg>,
    pub xforms: Vec<rt_xform>,
    pub triangles: Vec<rt_triangle>,
    pub materials: Vec<rt_material>,
    pub patterns: Vec<rt_pattern>,
    pub chain_mats: Vec<f64>,
    pub xform_chain: Vec<rt_chain_ref>,
    pub gate_chain: Vec<rt_chain_ref>,
    pub light: rt_light,
}

impl FlatWorld {
    pub fn desc(&self) -> rt_scene_desc {
        rt_scene_desc {
            light: self.light,
            n_leaves: self.leaves.len() as i32, n_gates: self.gates.len() as i32, n_xforms: self.xforms.len() as i32,
            n_triangles: self.triangles.len() as i32, n_materials: self.materials.len() as i32,
            n_patterns: self.patterns.len() as i32, n_csg: self.csg.len() as i32,
            n_chain_mats: (self.chain_mats.len() / 16) as i32,
            csg: self.csg.as_ptr(), leaves: self.leaves.as_ptr(), gates: self.gates.as_ptr(), xforms: self.xforms.as_ptr(),
            triangles: self.triangles.as_ptr(), materials: self.materials.as_ptr(), patterns: self.patterns.as_ptr(),
            chain_mats: self.chain_mats.as_ptr(), xform_chain: self.xform_chain.as_ptr(), gate_chain: self.gate_chain.as_ptr(),
        }
    }
}

fn m16(m: &Matrix4D) -> [f64; 16] { *m.as_array() }  // pub(crate) accessor over Matrix4D::data
fn v3(t: &crate::tuple::Tuple4D) -> [f64; 3] { [t.x, t.y, t.z] }
fn c3(c: &crate::color::Color) -> [f64; 3] { [c.r, c.g, c.b] }
fn is_identity(m: &[f64; 16]) -> bool { (0..16).all(|i| m[i] == if i % 5 == 0 { 1.0 } else { 0.0 }) }  // -0.0 == 0.0

struct Walk<'a> {
    out: FlatWorld,
    chain: Vec<[f64; 16]>,        // non-identity local inverses from the top level down, outermost first
    left_operands: Vec<&'a Shape>, // left operands of the enclosing CSG operators, innermost last
}

impl<'a> Walk<'a> {
    fn emit_chain(&mut self) -> rt_chain_ref {
        if self.chain.len() < 2 { return rt_chain_ref { first: 0, count: 0 }; }
        let first = (self.out.chain_mats.len() / 16) as i32;
        for m in &self.chain { self.out.chain_mats.extend_from_slice(m); }
        rt_chain_ref { first, count: self.chain.len() as i32 }
    }

    fn add_pattern(&mut self, p: &Pattern) -> i32 {
        let inv = m16(&p.transform.inverse().expect("Pattern transform should be invertible."));  // src/pattern.rs:199
        let mut r = rt_pattern { ty: RT_PAT_NULL, first: -1, second: -1, reserved: 0, inv, a: [1.0; 3], b: [0.0; 3] };
        match &p.ty {
            PatternType::Null => r.ty = RT_PAT_NULL,
            PatternType::Point => r.ty = RT_PAT_POINT,
            PatternType::Identity(c) => { r.ty = RT_PAT_IDENTITY; r.a = c3(c); }
            PatternType::Stripe(a, b) => { r.ty = RT_PAT_STRIPE; r.a = c3(a); r.b = c3(b); }
            PatternType::Ring(a, b) => { r.ty = RT_PAT_RING; r.a = c3(a); r.b = c3(b); }
            PatternType::Checker(a, b) => { r.ty = RT_PAT_CHECKER; r.a = c3(a); r.b = c3(b); }
            PatternType::Gradient(a, b) => { r.ty = RT_PAT_GRADIENT; r.a = c3(a); r.b = c3(b); }
            PatternType::Blend(a, b) => { r.ty = RT_PAT_BLEND; r.first = self.add_pattern(a); r.second = self.add_pattern(b); }
        }
        self.out.patterns.push(r);
        self.out.patterns.len() as i32 - 1
    }

    fn add_material(&mut self, m: &Material) -> i32 {
        let pattern = match &m.pattern { Some(p) => self.add_pattern(p), None => -1 };
        self.out.materials.push(rt_material {
            color: c3(&m.color), ambient: m.ambient, diffuse: m.diffuse, specular: m.specular, shininess: m.shininess,
            reflective: m.reflective, refractive_index: m.refractive_index, transparency: m.transparency, pattern, reserved: 0 });
        self.out.materials.len() as i32 - 1
    }

    /// `ray_acc`: product of the enclosing levels' local inverses in the order shape::intersect applies
    /// them (src/shape.rs:1467-1470).  `gate` / `csg`: innermost enclosing group / operator or -1.
    fn walk(&mut self, s: &'a Shape, ray_acc: Option<Matrix4D>, gate: i32, csg: i32) {
        let local_inv = *s.transform_inverse();
        let ray_inv = match ray_acc { Some(acc) => local_inv * acc, None => local_inv };
        let pushed = !is_identity(&m16(&local_inv));
        if pushed { self.chain.push(m16(&local_inv)); }
        match &s.ty {
            ShapeType::Union(l, r) | ShapeType::Intersection(l, r) | ShapeType::Difference(l, r) => {
                let op = match &s.ty { ShapeType::Union(..) => RT_CSG_UNION, ShapeType::Intersection(..) => RT_CSG_INTERSECTION,
                                       _ => RT_CSG_DIFFERENCE };
                assert!(self.left_operands.len() < 32, "CSG operators nested deeper than 32 levels");
                let ci = self.out.csg.len();
                self.out.csg.push(rt_csg { op, parent: csg, gate, first_leaf: self.out.leaves.len() as i32, n_leaves: 0, reserved: 0 });
                self.left_operands.push(&**l);
                self.walk(&**l, Some(ray_inv), gate, ci as i32);
                self.walk(&**r, Some(ray_inv), gate, ci as i32);
                self.left_operands.pop();
                self.out.csg[ci].n_leaves = self.out.leaves.len() as i32 - self.out.csg[ci].first_leaf;
            }
            ShapeType::Group(children) => {
                if !children.is_empty() {                                  // src/shape.rs:1186-1188
                    let b = s.bounds();                                    // src/shape.rs:470-544, NaN / inf corners included
                    self.out.gates.push(rt_gate { parent: gate, csg, to_local: m16(&ray_inv), bmin: v3(&b.minimum), bmax: v3(&b.maximum) });
                    let chain = self.emit_chain();
                    self.out.gate_chain.push(chain);
                    let gi = self.out.gates.len() as i32 - 1;
                    for c in children { self.walk(c, Some(ray_inv), gi, csg); }
                }
            }
            leaf_ty => {
                let mut leaf = rt_leaf { gate, csg, tri: -1, ..Default::default() };
                match leaf_ty {
                    ShapeType::Empty => leaf.ty = RT_EMPTY,
                    ShapeType::Sphere => leaf.ty = RT_SPHERE,
                    ShapeType::Cube => leaf.ty = RT_CUBE,
                    ShapeType::Plane(n) => { leaf.ty = RT_PLANE; leaf.plane_normal = v3(n); }
                    ShapeType::Cylinder(lo, hi, closed) => { leaf.ty = RT_CYLINDER; leaf.minimum = *lo; leaf.maximum = *hi; leaf.closed = *closed as i32; }
                    ShapeType::Cone(lo, hi, closed) => { leaf.ty = RT_CONE; leaf.minimum = *lo; leaf.maximum = *hi; leaf.closed = *closed as i32; }
                    ShapeType::Triangle(t) => {
                        leaf.ty = RT_TRIANGLE;
                        self.out.triangles.push(rt_triangle { p1: v3(&t.p1), e1: v3(&t.e1), e2: v3(&t.e2), normal: v3(&t.normal), ..Default::default() });
                        leaf.tri = self.out.triangles.len() as i32 - 1;
                    }
                    ShapeType::SmoothTriangle(t) => {
                        leaf.ty = RT_SMOOTH_TRIANGLE;
                        self.out.triangles.push(rt_triangle { p1: v3(&t.triangle_info.p1), e1: v3(&t.triangle_info.e1), e2: v3(&t.triangle_info.e2),
                                                              normal: v3(&t.triangle_info.normal), n1: v3(&t.n1), n2: v3(&t.n2), n3: v3(&t.n3) });
                        leaf.tri = self.out.triangles.len() as i32 - 1;
                    }
                    _ => unreachable!(),
                }
                for (k, left) in self.left_operands.iter().rev().enumerate() {   // bit 0 = innermost operator
                    if left.includes(s) { leaf.csg_left |= 1u32 << k; }           // by value, src/shape.rs:450-467
                }
                leaf.material = self.add_material(&s.material);
                // full_transform_inverse, or transform_inverse for a top-level shape (src/shape.rs:757-784)
                let full = s.full_transform_inverse().copied().unwrap_or(local_inv);
                let fwd = match s.parent_transform() { Some(p) => *p * *s.transform(), None => *s.transform() };
                self.out.xforms.push(rt_xform { ray_inv: m16(&ray_inv), full_inv: m16(&full), fwd: m16(&fwd) });
                let chain = self.emit_chain();
                self.out.xform_chain.push(chain);
                leaf.xform = self.out.xforms.len() as i32 - 1;
                self.out.leaves.push(leaf);
            }
        }
        if pushed { self.chain.pop(); }
    }
}

/// Depth-first over `world.objects`: leaf order is the tie-break of equal distances (src/intersect.rs:71-89).
pub fn flatten(world: &World) -> FlatWorld {
    let mut w = Walk { out: FlatWorld::default(), chain: Vec::new(), left_operands: Vec::new() };
    for s in &world.objects { w.walk(s, None, -1, -1); }
    w.out.light = rt_light { intensity: c3(&world.light_source.intensity), position: v3(&world.light_source.position) };
    w.out
}

/// `Camera` with `transform.inverse()` computed once (the reference recomputes it per pixel, src/camera.rs:87).
pub fn camera_desc(camera: &Camera) -> rt_camera {
    rt_camera { hsize: camera.hsize as u32, vsize: camera.vsize as u32, half_width: camera.half_width,
                half_height: camera.half_height, pixel_size: camera.pixel_size,
                inv: m16(&camera.transform.inverse().expect("Camera transform should be invertible.")) }
}

struct SendPtr<T>(*mut T);
unsafe impl<T> Send for SendPtr<T> {}
unsafe impl<T> Sync for SendPtr<T> {}

/// src/parallel.rs:183-204 with `jobs` = number of GPUs (capped at the GPUs there are).
pub fn parallel_render(world: World, camera: Camera, jobs: usize, out: &Path) {
    assert!(jobs > 0);                                                     // src/parallel.rs:94
    println!("Rendering using {} threads...", jobs);                       // src/parallel.rs:188
    let have = unsafe { rt_device_count() };
    assert!(have > 0, "no GPU visible");
    let n = jobs.min(have as usize);
    let flat = flatten(&world);
    let desc = flat.desc();
    let cam = camera_desc(&camera);
    let (w, h) = (camera.hsize, camera.vsize);

    // one build of the acceleration structure, cloned to the other GPUs
    let mut scenes: Vec<SendPtr<rt_scene>> = Vec::new();
    let mut first = std::ptr::null_mut();
    check(unsafe { rt_scene_create(&desc, 0, &mut first) });
    scenes.push(SendPtr(first));
    for k in 1..n {
        let mut s = std::ptr::null_mut();
        check(unsafe { rt_scene_clone(first, k as i32, &mut s) });
        scenes.push(SendPtr(s));
    }

    // one frame for all GPUs: part k copies exactly its bands of 32 rows into it (page-locked when possible)
    let bytes = w * h * 3;
    let pinned = unsafe { rt_host_alloc(bytes as u64) } as *mut u8;
    let mut fallback = if pinned.is_null() { vec![0u8; bytes] } else { Vec::new() };
    let frame = SendPtr(if pinned.is_null() { fallback.as_mut_ptr() } else { pinned });
    std::thread::scope(|scope| {
        for (k, sc) in scenes.iter().enumerate() {
            let (cam, frame) = (&cam, &frame);
            scope.spawn(move || {
                check(unsafe { rt_render_host(sc.0, cam, k as i32, n as i32, RT_FLAG_BANDS, frame.0,
                                              std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut()) });
            });
        }
    });
    for sc in &scenes { unsafe { rt_scene_destroy(sc.0) }; }

    // Canvas::save (src/canvas.rs:47-94): the same bytes, formatted on GPU 0
    let cap = unsafe { rt_ppm_max_bytes(w as u64, h as u64) } as usize;
    let mut text = vec![0u8; cap];
    let mut len = 0u64;
    check(unsafe { rt_ppm_encode_host(0, frame.0, w as u64, h as u64, text.as_mut_ptr() as *mut _, cap as u64, &mut len) });
    std::fs::File::create(out).unwrap().write_all(&text[..len as usize]).unwrap();
    if !pinned.is_null() { unsafe { rt_host_free(pinned as *mut _) }; }
    println!("...done.\n");                                                // src/parallel.rs:202
    println!("Saved render to {:?}.", out);                                // src/parallel.rs:203
}

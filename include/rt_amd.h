/* C ABI of the MI355X render path: camera ray -> world intersection -> Phong
 * shading with shadow, reflection and refraction rays, all in f64.
 *
 * The reference crate has no FFI layer; its boundary for this path is the Rust
 * API `parallel_render(world, camera, jobs, out)` (src/parallel.rs:183), whose
 * per-pixel body is `Camera::ray_for_pixel` (src/camera.rs:76) followed by
 * `World::color_at` (src/world.rs:249) and `Canvas::write_pixel`
 * (src/canvas.rs:114).  A host (the Rust crate through `extern "C"`, or the C++
 * host in this repository) flattens its `World` into the plain-data
 * description below and calls these entry points.
 *
 * Conventions: every function returns 0 on success or a negative rt_status;
 * rt_last_error() gives the message of the calling thread's last failure.
 * Nothing throws or aborts across this boundary.  Calls block until the work
 * is done.  One caller thread per rt_scene at a time.  Pixels are row-major,
 * index y * width + x (src/canvas.rs:120).  Matrices are row-major 4x4 like
 * `Matrix4D` (src/matrix.rs:101-103).  Pointers named *_dev are device (HBM)
 * addresses on the scene's GPU; all other pointers are host memory.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rt_status {
  RT_OK = 0,
  RT_ERR_INVALID = -1,     /* bad argument or inconsistent description */
  RT_ERR_UNSUPPORTED = -2, /* valid for the reference, not built here: projective transforms, a CSG operator tree whose
                              shapes can yield more than 4096 intersections per ray, operators nested more than 32 deep,
                              a recursion deeper than 4096 */
  RT_ERR_DEVICE = -3,      /* HIP runtime failure */
  RT_ERR_NOMEM = -4
} rt_status;

/* `ShapeType` leaves (src/shape.rs:23-61).  Groups become rt_gate records. */
typedef enum rt_leaf_type {
  RT_EMPTY = 0,
  RT_SPHERE = 1,
  RT_PLANE = 2,
  RT_CUBE = 3,
  RT_CYLINDER = 4,
  RT_CONE = 5,
  RT_TRIANGLE = 6,
  RT_SMOOTH_TRIANGLE = 7
} rt_leaf_type;

/* `PatternType` (src/pattern.rs:8-32). */
typedef enum rt_pattern_type {
  RT_PAT_NULL = 0,
  RT_PAT_POINT = 1,
  RT_PAT_IDENTITY = 2,
  RT_PAT_STRIPE = 3,
  RT_PAT_RING = 4,
  RT_PAT_CHECKER = 5,
  RT_PAT_GRADIENT = 6,
  RT_PAT_BLEND = 7
} rt_pattern_type;

/* The three cached matrices of one `Shape` (src/shape.rs:133-144). */
typedef struct rt_xform {
  double ray_inv[16];  /* world -> object for rays: the product of the local `transform_inverse`
                          of every enclosing group and of the shape, applied in the order
                          `shape::intersect` applies them (src/shape.rs:1467-1470) */
  double full_inv[16]; /* `full_transform_inverse`, or `transform_inverse` for a top-level shape:
                          the matrix of world_to_object / normal_to_world (src/shape.rs:757-784) */
  double fwd[16];      /* object -> world (parent_transform * transform); used for bounds only */
} rt_xform;

/* `TriangleInfo` / `SmoothTriangleInfo` (src/geometry.rs:16-65). */
typedef struct rt_triangle {
  double p1[3], e1[3], e2[3], normal[3];
  double n1[3], n2[3], n3[3]; /* smooth triangles only */
} rt_triangle;

/* One non-container `Shape`.  Leaves are listed in depth-first order of
 * `World.objects`; that order breaks ties between equal intersection
 * distances exactly as the reference's stable sorts do (src/intersect.rs:71-89). */
typedef struct rt_leaf {
  int32_t type;     /* rt_leaf_type */
  int32_t closed;   /* cylinder / cone flag (src/shape.rs:34-37) */
  int32_t material; /* index into materials */
  int32_t xform;    /* index into xforms */
  int32_t gate;     /* innermost enclosing group, index into gates, -1 at top level */
  int32_t tri;      /* index into triangles for the two triangle types, else -1 */
  int32_t csg;      /* innermost enclosing CSG operator, index into csg, -1 when there is none */
  uint32_t csg_left; /* bit k: `left.includes(shape)` (src/shape.rs:450-467, by value) for the k-th
                        enclosing operator counted from the innermost one */
  double minimum, maximum; /* cylinder / cone extent along y */
  double plane_normal[3];  /* `ShapeType::Plane(n)` */
} rt_leaf;

/* One `ShapeType::Group`: the bounding-box gate of `intersect_group`
 * (src/shape.rs:1179-1222).  bmin/bmax are the group's `bounds()` exactly as
 * the reference computes them (src/shape.rs:509-535), tested against the ray
 * in the group's local space. */
typedef struct rt_gate {
  int32_t parent;      /* enclosing group or -1 */
  int32_t csg;         /* innermost CSG operator the group lies below, or -1 */
  double to_local[16]; /* world -> group space for rays (product of local inverses) */
  double bmin[3], bmax[3];
} rt_gate;

/* One `ShapeType::Union / Intersection / Difference` (src/shape.rs:319-364).  Its operands are the
 * shapes below it; the leaves below one operator are contiguous in leaves[] (depth-first order).
 * Operators are listed parents first.  intersect_csg and filter_intersections
 * (src/shape.rs:1339-1354, src/intersect.rs:278-328) run per ray on the device. */
typedef enum rt_csg_op { RT_CSG_UNION = 0, RT_CSG_INTERSECTION = 1, RT_CSG_DIFFERENCE = 2 } rt_csg_op;
typedef struct rt_csg {
  int32_t op;         /* rt_csg_op */
  int32_t parent;     /* enclosing operator (directly or through groups) or -1 */
  int32_t gate;       /* innermost group enclosing the operator itself, or -1 */
  int32_t first_leaf; /* leaves[first_leaf .. first_leaf + n_leaves) lie below this operator */
  int32_t n_leaves;
  int32_t reserved;
} rt_csg;

/* `Material` (src/light.rs:34-63). */
typedef struct rt_material {
  double color[3];
  double ambient, diffuse, specular, shininess;
  double reflective, refractive_index, transparency;
  int32_t pattern; /* index into patterns or -1 */
  int32_t reserved;
} rt_material;

/* `Pattern` (src/pattern.rs:34-38) with its transform already inverted
 * (the reference inverts it on every lookup, src/pattern.rs:199). */
typedef struct rt_pattern {
  int32_t type;     /* rt_pattern_type */
  int32_t first;    /* blend operands: indices into patterns */
  int32_t second;
  int32_t reserved;
  double inv[16];
  double a[3], b[3]; /* primary / secondary colour */
} rt_pattern;

/* `PointLight` (src/light.rs:11-14). */
typedef struct rt_light {
  double intensity[3];
  double position[3];
} rt_light;

/* `Camera` (src/camera.rs:13-30) with `transform.inverse()` precomputed
 * (the reference recomputes it per pixel, src/camera.rs:87). */
typedef struct rt_camera {
  uint32_t hsize, vsize;
  double half_width, half_height, pixel_size;
  double inv[16];
} rt_camera;

/* Transform chains.  `shape::intersect` moves a ray into object space level by level: every enclosing
 * group or CSG operator applies its own `transform_inverse` to the ray its parent handed down
 * (src/shape.rs:1467-1470, 1213-1218, 1339-1354).  Applying the product `ray_inv` / `to_local` once is
 * bit-identical to that only while at most one matrix of the chain differs from the identity.  For a
 * shape or group with two or more, the host lists those non-identity local inverses, outermost
 * first, in chain_mats and points at them from xform_chain / gate_chain; the device then applies
 * them one after the other, as the reference does.  An entry with count 0 (or a null table) means
 * "the product is exact". */
typedef struct rt_chain_ref {
  int32_t first; /* index of the first matrix in chain_mats */
  int32_t count; /* 0: none */
} rt_chain_ref;

typedef struct rt_scene_desc {
  rt_light light;
  int32_t n_leaves, n_gates, n_xforms, n_triangles, n_materials, n_patterns;
  int32_t n_csg, n_chain_mats;
  const rt_csg* csg;
  const rt_leaf* leaves;
  const rt_gate* gates;
  const rt_xform* xforms;
  const rt_triangle* triangles;
  const rt_material* materials;
  const rt_pattern* patterns;
  const double* chain_mats;        /* n_chain_mats x 16, row-major; may be null when n_chain_mats == 0 */
  const rt_chain_ref* xform_chain; /* n_xforms entries, or null */
  const rt_chain_ref* gate_chain;  /* n_gates entries, or null */
} rt_scene_desc;

/* Ray counts use the reference's unit: one ray = one `World::intersect` query
 * (primary src/world.rs:250, shadow :129, reflection :156, refraction :213). */
typedef struct rt_stats {
  uint64_t primary, shadow, reflect, refract;
  double render_ms;         /* wall time of the call */
  double kernel_ms;         /* device time of the generation kernels (HIP events on the library's stream) */
  uint64_t kernel_launches; /* generation kernels launched = ray generations traced */
  uint64_t generations;
  /* traversal work, filled only under RT_FLAG_COUNTERS: closest-hit (and refractive-index) queries... */
  uint64_t node_visits, tri_tests, quadric_tests, gate_tests;
  /* ...and shadow queries */
  uint64_t shadow_node_visits, shadow_tri_tests, shadow_quadric_tests, shadow_gate_tests;
} rt_stats;

typedef struct rt_scene rt_scene;

enum {
  RT_FLAG_COUNTERS = 1, /* run the instrumented kernels and fill the work counters */
  RT_FLAG_ASYNC = 2,    /* rt_render only: enqueue the frame and return; see rt_scene_wait */
  RT_FLAG_COMPACT = 4,  /* rt_render only: rgb8_dev receives just this part's pixels, see rt_render */
  RT_FLAG_BANDS = 8,    /* rt_render only: parts are bands of 32 pixel rows, band b belongs to part b % n_parts */
  RT_FLAG_BANDS_16 = 16, /* with RT_FLAG_BANDS: bands of 16 rows (finer bands spread the expensive rows over the parts */
  RT_FLAG_BANDS_8 = 32   /* more evenly) ... of 8 rows */
};

const char* rt_last_error(void);

/* Number of visible GPUs, or a negative rt_status. */
int rt_device_count(void);

/* Copies the description, builds the acceleration structure over the leaves
 * and uploads everything to `device`.  The caller keeps ownership of desc. */
int rt_scene_create(const rt_scene_desc* desc, int device, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);

/* The same with a choice of builder for the acceleration structure.  Both builders split by the binned
 * surface-area heuristic and their hierarchies traverse equally fast; the one on the GPU builds a million
 * triangles in 20 ms where sixteen host cores take 180 ms, the host one is quicker for a few thousand
 * items.  Without a flag subtrees of 32768 items or more are built on the GPU and the rest on the host;
 * RT_BUILD_DEVICE lowers that to 2048 items, RT_BUILD_HOST keeps everything on the host.  Both are
 * conservative hierarchies over the same shapes, so the images are the same; the reference has neither
 * (it scans every child of a group, src/shape.rs:1213-1219).  RT_AMD_DEVICE_BVH=1 / 0 in the
 * environment overrides the flags, RT_AMD_DEVICE_BVH_MIN the item count. */
enum { RT_BUILD_DEVICE = 1, RT_BUILD_HOST = 2 };
int rt_scene_create_with(const rt_scene_desc* desc, int device, uint32_t build_flags, rt_scene** out);
/* How many subtrees the GPU builder made for this scene and the seconds it took (of the build_seconds
 * rt_scene_footprint reports). */
int rt_scene_build_info(const rt_scene* scene, uint64_t* device_builds, double* device_build_seconds);

/* Shapes moved, the scene stays: `Shape::set_transform` on any shape of the world, which the reference
 * re-propagates to the children (src/shape.rs:592-656), then the same flattening as for rt_scene_create.
 * desc must describe the same shape tree (same counts, same leaf types and group / triangle / transform
 * slots); matrices, group boxes, triangle vertices, materials, patterns and the light are taken from
 * it.  The acceleration structure keeps its shape and gets new boxes (a refit on the GPU, no rebuild);
 * a hierarchy refitted many times over large motions traverses slower than a rebuilt one.  From the
 * first update on every group's bounding-box gate is evaluated exactly as the reference does it (the
 * shortcut for gates that provably contain their subtree needs a rebuild).  Where the hierarchy cannot
 * keep its shape -- a shape crossed between bounded and unbounded (scaled beyond 100, see
 * RT_ERR_UNSUPPORTED above), a transform chain changed its length (an identity level became a transform
 * or the reverse), or the scene has CSG operators -- the acceleration structure is built again with the
 * builder the scene was created with; the handle, its streams and frame buffers stay (rt_scene_rebuilds
 * counts these).  Blocks until frames in flight are done and the update is complete.  An update that fails
 * before any device table was written leaves the scene as it was; one that fails later leaves it unusable
 * (rendering returns RT_ERR_INVALID) until an update succeeds. */
int rt_scene_update(rt_scene* scene, const rt_scene_desc* desc);
/* How many updates of this scene built the acceleration structure again instead of refitting it. */
int rt_scene_rebuilds(const rt_scene* scene, uint64_t* rebuilds);

/* A copy of `scene` on `device` (the same GPU or another one of this process): the acceleration structure
 * is not built again, its arrays are copied device to device.  The copy is independent of the original
 * (own streams, own frames in flight) and is released with rt_scene_destroy.  This is how a host that
 * renders one frame on several GPUs (src/parallel.rs:183-204 hands one Arc<World> to all its workers)
 * pays for one build. */
int rt_scene_clone(const rt_scene* scene, int device, rt_scene** out);

/* Device bytes held by the scene data (acceleration structure included), and
 * the host seconds spent building the acceleration structure. */
int rt_scene_footprint(const rt_scene* scene, uint64_t* scene_bytes, double* build_seconds);

/* `parallel_render` without the file write: renders part `part` of `n_parts`
 * (interleaved 32x32 pixel tiles, tile (tx, ty) belongs to part (tx + ty) % n_parts, or with
 * RT_FLAG_BANDS whole rows of tiles, row ty to part ty % n_parts; part 0 of 1 is the whole image).  Outputs
 * are full-size W*H images in which only this part's pixels are written; any
 * of them may be NULL.  rgb8: the quantised bytes of `Canvas::save`
 * (ceil(clamp(c*255)), src/canvas.rs:60-62); rgb64: the f64 colours;
 * hit_leaf: primary hit leaf index or -1.
 * With RT_FLAG_COMPACT (rgb8 only) the buffer holds only the part's own tiles, 3072 bytes per tile
 * in the order the part owns them (row-major over tiles), and inside a tile sixteen 8x8 pixel
 * blocks (block b at columns 8 (b % 4), rows 8 (b / 4)), each row-major; pixels of a tile that
 * fall outside the image are written as zeros.  This is the unit a multi-GPU host gathers: one
 * contiguous buffer per GPU (the reference gathers per pixel through a channel,
 * src/parallel.rs:44-49). */
int rt_render(rt_scene* scene, const rt_camera* camera, int part, int n_parts, uint32_t flags,
              uint8_t* rgb8_dev, double* rgb64_dev, int32_t* hit_leaf_dev, rt_stats* stats);

/* Frames of a render loop, back to back.  With RT_FLAG_ASYNC, rt_render enqueues the frame on the
 * scene's stream and returns at once (stats is not written); at most eight such calls (rt_render or
 * rt_render_frames) are in flight, a ninth waits for the oldest.  A scene whose materials spawn
 * reflection or refraction rays is a
 * row of ray generations: its first frame for a camera size and depth is rendered one generation at a
 * time (the call blocks, its counts still go to rt_scene_wait); later frames launch every generation at
 * once on the queue lengths of the last frame and are only enqueued, up to three in flight.  Whether such
 * a frame's queues had room is known when its counters are back, and the library looks at them before
 * anything reads the frame: in rt_download / rt_download_2d of its image (which therefore wait for that
 * frame, not for the frames enqueued after it), in rt_scene_wait and in every blocking entry point.  A
 * frame that did not fit is rendered again there, exactly, into the same buffers.  A caller keeps such a
 * GPU busy by enqueuing frame k + 1 before it downloads frame k; work it enqueues itself on the frame's
 * buffers needs rt_scene_wait first.
 * rt_scene_wait blocks until every enqueued frame is complete and returns the sums of their ray
 * counts, kernel time and kernel launches since the previous wait (render_ms is not filled).
 * Asynchronous frames go round-robin to three streams of the scene.  rt_download / rt_download_2d without
 * an after_stream wait for all of them; every blocking entry point of the scene first collects the frames in
 * flight.  Work a caller enqueues on streams of its own (and rt_ppm_encode, which runs on the null
 * stream) is not ordered against frames in flight: call rt_scene_wait first, or order the stream after
 * rt_scene_stream.  The reference renders one frame per process (src/main.rs:71); this pair has no
 * counterpart there. */
int rt_scene_wait(rt_scene* scene, rt_stats* total);

/* Up to RT_MAX_FRAMES_PER_CALL frames of one scene at once: frame k through cameras[k] (all of one canvas size) into
 * rgb8_dev[k], each as rt_render(scene, &cameras[k], part, n_parts, flags, rgb8_dev[k], NULL, NULL, NULL) would
 * render it, and flags must hold RT_FLAG_ASYNC.  For a scene that spawns no reflection or refraction rays the
 * frames are one kernel launch: a part of a frame alone is a short launch that ends with its slowest wave while
 * most of the GPU idles (an eighth of the teapot frame at 1080p: 0.031 ms per frame one at a time, 0.013 four at a
 * time), several frames' worth of it fill the GPU as a whole frame does.  Every other scene gets its frames one
 * after the other.  rt_scene_wait counts one kernel launch and n_frames generations for such a call.  The reference
 * renders its frames one process at a time (src/main.rs:71, src/parallel.rs:183-204); this is the many-frame form
 * of that loop for a moving camera. */
#define RT_MAX_FRAMES_PER_CALL 8
int rt_render_frames(rt_scene* scene, const rt_camera* cameras, int n_frames, int part, int n_parts, uint32_t flags,
                     uint8_t* const* rgb8_dev);

/* The HIP stream (hipStream_t) the most recent frame was enqueued on, so that a caller can order
 * its own streams after it (hipStreamWaitEvent).  Asynchronous frames go round-robin to three
 * streams, so ask again after every rt_render. */
int rt_scene_stream(rt_scene* scene, void** hip_stream);

/* Copies `bytes` from device memory to pinned (page-locked) host memory on the scene's copy
 * stream (a DMA transfer that runs beside the next frame's kernel).  The copy is enqueued; it starts
 * after everything enqueued so far on after_stream (a hipStream_t; NULL = every render stream of the
 * scene) and is complete after rt_scene_wait or after the copy stream
 * (rt_scene_copy_stream) has drained.  The caller keeps src_dev and dst_host untouched until then.
 * Replaces the per-pixel channel receive of the reference's canvas (src/parallel.rs:196-200). */
int rt_download(rt_scene* scene, const void* src_dev, void* dst_host, uint64_t bytes, void* after_stream);
int rt_scene_copy_stream(rt_scene* scene, void** hip_stream);

/* The same for `rows` pieces of row_bytes bytes that lie `pitch` bytes apart in both buffers.  With
 * RT_FLAG_BANDS a part of a frame is exactly that: bands of 32 pixel rows (row_bytes = 32 * width * 3),
 * n_parts bands apart, so every GPU of a node moves its own bands over its own PCIe link straight
 * into one host frame (shared, page-locked memory) and no GPU-to-GPU exchange is needed. */
int rt_download_2d(rt_scene* scene, const void* src_dev, void* dst_host, uint64_t row_bytes, uint64_t rows,
                   uint64_t pitch, void* after_stream);

/* Fences on the copy stream, for a host that reuses frame buffers in a ring: rt_copy_fence marks the point
 * the copy stream has reached (after the downloads enqueued so far) under `slot` (0..15); rt_copy_wait
 * blocks until the copy stream has passed the mark last set under `slot` (returns at once when none was
 * set).  No counterpart in the reference, whose canvas is written under a mutex (src/parallel.rs:48). */
int rt_copy_fence(rt_scene* scene, uint32_t slot);
int rt_copy_wait(rt_scene* scene, uint32_t slot);

/* Page-locked host memory usable from every GPU of the process (NULL when it cannot be had): frames
 * in it are filled by DMA at full PCIe speed.  The reference's canvas is a plain Vec (src/canvas.rs:22-28). */
void* rt_host_alloc(uint64_t bytes);
void rt_host_free(void* p);
/* Page-locks memory the caller owns (the canvas of src/canvas.rs:22-28 where it lies, or a shared-memory
 * mapping in which the processes of several GPUs assemble one frame, each over its own PCIe link) so that
 * rt_download fills it by DMA like memory from rt_host_alloc.  Unregister before the memory is released. */
int rt_host_register(void* p, uint64_t bytes);
int rt_host_unregister(void* p);

/* Device memory on `device` for the frames rt_render writes (zero-filled), for hosts that do not link a
 * GPU runtime of their own.  The reference's canvas is a plain Vec (src/canvas.rs:22-28). */
int rt_device_alloc(int device, uint64_t bytes, void** out);
void rt_device_free(int device, void* p);

/* Same, into host buffers (the library stages through device memory).  With RT_FLAG_BANDS only the
 * part's own bands are copied to the host buffers, so the parts of one frame, rendered by different
 * GPUs from different threads, can be given the same buffers and fill them without overlap. */
int rt_render_host(rt_scene* scene, const rt_camera* camera, int part, int n_parts, uint32_t flags,
                   uint8_t* rgb8, double* rgb64, int32_t* hit_leaf, rt_stats* stats);

/* `Camera::ray_for_pixel` for n pixels: xy = n x (px, py); rays6 = n x
 * (origin xyz, direction xyz).  Host buffers. */
int rt_ray_for_pixel(rt_scene* scene, const rt_camera* camera, uint64_t n, const uint32_t* xy,
                     double* rays6);

/* `World::color_at(ray, remaining)` for n rays built with `Ray4D::new`, for any remaining >= 0 as in the
 * reference (src/world.rs:249; the renderer uses 5): every level of the recursion is one generation of rays
 * whose records stay in HBM until the colours have flowed back, so a deep recursion in a hall of mirrors costs
 * memory in proportion.  More than 4096 levels are refused (RT_ERR_UNSUPPORTED).
 * rgb = n x 3 doubles; leaf (nullable) = hit leaf index or -1.  Host buffers. */
int rt_color_at(rt_scene* scene, uint64_t n, const double* rays6, int remaining, uint32_t flags,
                double* rgb, int32_t* leaf, rt_stats* stats);

/* `World::intersect` followed by `Intersections::hit`: distance (NaN when
 * nothing is hit) and leaf index (-1).  Host buffers. */
int rt_intersect_hit(rt_scene* scene, uint64_t n, const double* rays6, double* t, int32_t* leaf);

/* `World::is_shadowed` for n points (n x 3 doubles) -> n bytes.  Host buffers. */
int rt_is_shadowed(rt_scene* scene, uint64_t n, const double* points3, uint8_t* shadowed);

/* `Canvas::save` (src/canvas.rs:47-94) on the device: the rgb8 frame at rgb8_dev (device memory of
 * `device`, width * height * 3 bytes, row-major, finished - call rt_scene_wait first if it was
 * rendered with RT_FLAG_ASYNC) becomes the text of the reference's PPM file in out_dev (device memory,
 * capacity >= rt_ppm_max_bytes(width, height)): header, numbers, the reference's 70-column line
 * breaks, final newline, byte for byte.  *length receives the size of the file.  Blocking; works in
 * about 2.3 bytes per pixel: the GPU's fixed array of rt_ppm_encode_static when the frame fits it and no
 * other call is using it (then nothing is allocated), else a workspace allocated for the call.  The file then crosses PCIe once (rt_download
 * or hipMemcpy) instead of being formatted from a host copy of the frame. */
int rt_ppm_encode(int device, const uint8_t* rgb8_dev, uint64_t width, uint64_t height, char* out_dev,
                  uint64_t capacity, uint64_t* length);
/* The same for a frame and a file in host memory (staged through `device`; page-locked buffers from
 * rt_host_alloc move fastest).  When capacity is too small, *length still receives the size needed. */
int rt_ppm_encode_host(int device, const uint8_t* rgb8_host, uint64_t width, uint64_t height, char* out_host,
                       uint64_t capacity, uint64_t* length);
/* rt_ppm_encode for callers that cannot allocate device memory (serving loops and harnesses that forbid
 * hipMalloc).  It allocates nothing but it is a blocking call like rt_ppm_encode: it launches on the null
 * stream and waits for the file's length, so it is not for use while a stream is being captured into a
 * graph.  No counterpart in the reference.  Same arguments, same
 * result byte for byte, but the workspace is a fixed array in the library's device image (about 18.4 MB
 * on every GPU the library runs on, held until the process ends) instead of a hipMalloc per call, and the
 * call makes no allocation of its own.  Frames of more than rt_ppm_static_max_pixels() pixels (one
 * 3840x2160 frame) are refused with RT_ERR_UNSUPPORTED.  All calls on one GPU share that GPU's array, which
 * keeps the last call's contents (nothing in it has to start at zero): calls that overlapped in it would race,
 * so the library lets one call at a time into a GPU's array (a host-side lock per GPU: a second call on the
 * same GPU waits for the first to return; rt_ppm_encode, which borrows the array when it is free, never
 * waits and allocates instead).  Calls on different GPUs do not meet. */
int rt_ppm_encode_static(int device, const uint8_t* rgb8_dev, uint64_t width, uint64_t height, char* out_dev,
                         uint64_t capacity, uint64_t* length);
uint64_t rt_ppm_static_max_pixels(void);
/* Upper bound of the file size: 12 bytes per pixel, the header, the final newline. */
uint64_t rt_ppm_max_bytes(uint64_t width, uint64_t height);

/* Diagnostic, host only (no GPU needed): checks the direction index that finds the cones whose
 * parallel-ray branch (src/shape.rs:1089-1103) a ray direction triggers.  Builds the index over
 * n_bands random cones and looks up n_dirs random directions; returns the number of cones the
 * index failed to list for a direction that needs them (0 = conservative) or a negative rt_status.
 * No counterpart in the reference, which scans every shape for every ray. */
int rt_selftest_band_grid(int n_bands, int n_dirs, uint64_t seed, int* grid_n, uint64_t* entries);
/* Diagnostic: the same index built on the host and by the kernels that build it on `device` when a
 * scene is created, over n_bands random cones.  Returns 0 when the two are equal word for word, 1 when
 * they differ, or a negative rt_status.  *grid_n, *entries: resolution and size; *host_seconds,
 * *device_seconds: build times (any of the four may be null). */
int rt_selftest_band_grid_device(int n_bands, uint64_t seed, int device, int* grid_n, uint64_t* entries,
                                 double* host_seconds, double* device_seconds);

/* Diagnostic, host only: builds the acceleration structure over n_items random boxes (70 % triangle
 * slots, 30 % quadric slots) with `threads` build threads (0 = default) and returns a hash of every
 * byte the build produced (nodes, primitive orders, bounds, stack depth), the node count and the time
 * taken.  The hash must not depend on the thread count.  No counterpart in the reference, whose groups
 * are scanned linearly (src/shape.rs:1213-1219). */
int rt_selftest_bvh_build(int n_items, uint64_t seed, int threads, uint64_t* hash, uint64_t* n_nodes,
                          double* seconds);

/* Diagnostic, host only: the world-space normal the library stores for a plane or a flat triangle, whose
 * normal is the same wherever it is hit: shape::normal_at (src/shape.rs:1478-1484) applied to the object-space
 * normal, i.e. transposition(full_inv) * normal with w dropped, normalized, in the reference's operation
 * order.  The kernels load this value instead of repeating the computation per hit. */
int rt_selftest_world_normal(const double* full_inv16, const double* local_normal3, double* out3);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */

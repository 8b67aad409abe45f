// Host side of the MI355X render path: the part of the reference crate that
// stays on the CPU (scene JSON, OBJ models, the Shape tree with its cached
// transforms and group bounds, the PPM writer, parallel_render's driver) and
// the flattening of a World into the plain-data description of include/rt_amd.h.
//
// The reference is Rust and no Rust toolchain is available, so this mirror is
// C++; names follow the reference (World, Camera, Shape, Material, Pattern,
// parallel_render) and citations are reference file:line.
#pragma once
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"

namespace rthost {

// A reference `panic!` / `expect` surfaces as this exception and, through the C
// entry points, as a negative status with the same message.
struct Panic : std::runtime_error {
  using std::runtime_error::runtime_error;
};

struct Mat4 {
  double m[16];
  static Mat4 identity();
  Mat4 operator*(const Mat4& o) const;          // src/matrix.rs:604-621
  bool inverse(Mat4* out) const;                // src/matrix.rs:550-565 (cofactors / determinant)
  Mat4 inverse_or_panic(const char* msg) const;
};
Mat4 view_transform(const double from[3], const double to[3], const double up[4]);  // src/matrix.rs:415-434

struct Pattern {
  int type = RT_PAT_NULL;
  double a[3] = {1, 1, 1}, b[3] = {0, 0, 0};
  Mat4 transform = Mat4::identity();
  std::shared_ptr<Pattern> first, second;  // blend operands
};

struct Material {  // src/light.rs:96-112 defaults
  double color[3] = {1, 1, 1};
  std::optional<Pattern> pattern;
  double ambient = 0.1, diffuse = 0.9, specular = 0.9, shininess = 200.0;
  double reflective = 0.0, refractive_index = 1.0, transparency = 0.0;
};

struct Shape {
  enum Kind { Leaf, Group, Csg } kind = Leaf;
  int type = RT_EMPTY;  // rt_leaf_type for leaves, rt_csg_op for CSG operators
  int closed = 0;
  double minimum = 0, maximum = 0;
  double plane_normal[3] = {0, 1, 0};
  rt_triangle tri{};
  double p2[3] = {0, 0, 0}, p3[3] = {0, 0, 0};  // triangle corners as given (tri keeps p1 and the edges)
  Material material;
  Mat4 transform = Mat4::identity();
  std::vector<Shape> children;
};

struct PointLight {
  double intensity[3] = {0, 0, 0};
  double position[3] = {0, 0, 0};
};

struct World {
  std::vector<Shape> objects;
  PointLight light_source;
};

struct Camera {  // src/camera.rs:13-30
  size_t hsize = 0, vsize = 0;
  double half_width = 0, half_height = 0, pixel_size = 0, field_of_view = 0;
  Mat4 transform = Mat4::identity();
  static Camera make(size_t hsize, size_t vsize, double fov, const Mat4& transform);  // src/camera.rs:43-69
  rt_camera describe() const;  // with transform.inverse() (src/camera.rs:87)
};

struct Scene {
  World world;
  Camera camera;
};

// src/scene.rs:25-366 and src/obj.rs:26-276.  Model paths resolve against
// base_dir (the reference resolves them against the process directory).
Scene scene_from_json(const std::string& text, const std::string& base_dir);
std::map<std::string, Shape> obj_groups_from_text(const std::string& text, size_t* ignored_lines);

// The flattened World: arrays behind an rt_scene_desc.
struct FlatWorld {
  std::vector<rt_leaf> leaves;
  std::vector<rt_gate> gates;
  std::vector<rt_csg> csg;
  std::vector<rt_xform> xforms;
  std::vector<rt_triangle> triangles;
  std::vector<rt_material> materials;
  std::vector<rt_pattern> patterns;
  // transform chains of the shapes and groups that need them (see rt_scene_desc in rt_amd.h)
  std::vector<double> chain_mats;  // 16 per matrix
  std::vector<rt_chain_ref> xform_chain, gate_chain;
  rt_light light;
  rt_scene_desc desc() const;
};
FlatWorld flatten(const World& w);

// Canvas::save (src/canvas.rs:47-94) from already quantised bytes.
std::string ppm_encode(size_t width, size_t height, const uint8_t* rgb8);

struct Color {  // src/color.rs:10-19
  double r = 0, g = 0, b = 0;
};

// Canvas (src/canvas.rs:22-140): the frame as f64 colours, row-major.
class Canvas {
 public:
  size_t width = 0, height = 0;
  std::vector<Color> pixels;
  Canvas(size_t width, size_t height);                          // black (src/canvas.rs:35-41)
  void write_pixel(size_t x, size_t y, const Color& pixel);     // out of bounds: ignored (src/canvas.rs:114-121)
  std::optional<Color> read_pixel(size_t x, size_t y) const;    // out of bounds: nothing (src/canvas.rs:132-139)
  std::vector<uint8_t> quantised() const;                       // ceil(clamp(c * 255, 0, 255)) (src/canvas.rs:60-62)
  void save(const std::string& path) const;                     // src/canvas.rs:47-94, the text formatted on GPU 0
};

// The canvas parallel_render holds when its workers are done (src/parallel.rs:185-200), rendered by
// `jobs` GPUs.
Canvas render_canvas(const World& world, const Camera& camera, size_t jobs);

// parallel_render (src/parallel.rs:183-204): `jobs` is the number of GPUs.
void parallel_render(const World& world, const Camera& camera, size_t jobs, const std::string& out_path);


// parallel_render's workers kept alive over many frames (the reference renders one frame per process,
// src/main.rs:71; a viewer or a camera sweep renders many).  One worker thread and one copy of the scene
// per part; part k renders the bands of 8 pixel rows (RT_AMD_BAND_ROWS: 16, 32) k, k + parts, ... on device devices[k] and copies
// exactly those bands over that GPU's own link into a host frame all workers share, so a frame is
// complete when every worker has passed it and no GPU-to-GPU exchange takes place.  The acceleration
// structure is built once (rt_scene_create on the first device, rt_scene_clone for the others).
class FrameLoop {
 public:
  struct Result {
    double seconds = 0;          // wall time of the timed frames: start barrier -> all workers drained
    std::vector<rt_stats> part;  // per part: sums over its timed frames (rays by kind, kernel time, launches)
  };
  // A loop that renders only some parts of the frame, for one process per GPU: this process's devices
  // take parts first_part, first_part + 1, ... of total_parts, and the frames are assembled in `ring`
  // (depth frames back to back; page-locked or registered with rt_host_register by the caller, who
  // keeps it alive; typically a shared-memory mapping that every process of the job opens).
  struct Share {
    size_t first_part = 0, total_parts = 0;  // total_parts 0: the devices given are all the parts
    uint8_t* ring = nullptr;                 // nullptr: the loop allocates its own ring
  };
  // devices[k]: GPU of part k (the same GPU may serve several parts).  depth: frames in flight per worker
  // and host frames in the ring (1..16).
  FrameLoop(const World& world, const Camera& camera, const std::vector<int>& devices, int depth, const Share& share);
  FrameLoop(const World& world, const Camera& camera, const std::vector<int>& devices, int depth)
      : FrameLoop(world, camera, devices, depth, Share()) {}
  ~FrameLoop();
  FrameLoop(const FrameLoop&) = delete;
  FrameLoop& operator=(const FrameLoop&) = delete;
  // `warmup` untimed frames, then `frames` timed ones.  to_host: every frame is copied into the host ring
  // (frame f into slot f % depth); otherwise frames stay in each GPU's memory.
  Result run(size_t warmup, size_t frames, bool to_host);
  // `frames` frames rendered one at a time (each call blocks, nothing overlaps), optionally with the
  // traversal counters switched on; per part, summed over the frames.
  std::vector<rt_stats> blocking_frames(size_t frames, bool counters);
  const uint8_t* host_frame(int slot) const { return host_[(size_t)slot]; }
  size_t frame_bytes() const { return frame_bytes_; }
  size_t parts() const { return scenes_.size(); }
  rt_scene* scene(size_t part) const { return scenes_[part]; }
  uint64_t scene_bytes() const;
  double build_seconds() const { return build_seconds_; }

 private:
  rt_camera cam_{};
  std::vector<int> devices_;
  std::vector<rt_scene*> scenes_;
  std::vector<std::vector<uint8_t*>> dev_frames_;  // [part][slot], on the part's GPU: one block per part, slots frame_stride_ apart
  std::vector<uint8_t*> host_;                     // [slot], page-locked, shared by all parts
  uint8_t* host_block_ = nullptr;                  // the loop's own ring: one block, slots frame_stride_ apart
  size_t frame_bytes_ = 0;
  // Bytes from one slot to the next, in the device blocks and in the loop's own ring.  A ring of the loop's own is
  // padded to whole rounds of bands (a multiple of parts x band bytes): the bands of one part then lie at one pitch
  // through consecutive slots, and the frames of one launch go to the host with one strided copy (a copy costs about
  // 8 us whatever it moves, as much as an eighth of a 1080p frame takes on the link).  A caller's ring has its
  // frames back to back, and every frame is copied by itself.
  size_t frame_stride_ = 0;
  size_t first_part_ = 0, total_parts_ = 0;
  bool own_ring_ = true;
  int depth_ = 0;
  // rows of a band (8 unless RT_AMD_BAND_ROWS says 16 or 32): fine bands spread the expensive rows of a frame evenly
  // over the parts (teapot 1080p over 8 parts: the fullest part carries 1.08 times the rays of the emptiest with
  // bands of 8 rows, 1.29 times with 32) and one strided copy still moves a part
  size_t band_rows_ = 8;
  bool spawns_ = false;  // some material reflects or refracts: a frame is several generations of rays
  size_t frames_per_launch_ = 1;  // of the last run()
  uint32_t band_flags() const;

 public:
  size_t band_rows() const { return band_rows_; }
  size_t frames_per_launch() const { return frames_per_launch_; }  // what the last run() used (rt_render_frames)

 private:
  double build_seconds_ = 0;
  void release();
};

}  // namespace rthost

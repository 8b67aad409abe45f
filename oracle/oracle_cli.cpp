// TEST INFRASTRUCTURE ONLY -- see rt_oracle.hpp.  Command-line front end of the
// CPU oracle with the reference binary's flags (main.rs:19-56): -s/--scene,
// -o/--output (default ./out.ppm), -j/--jobs (default 1).  Without -s it renders
// the built-in sphere + floor scene of main.rs:72-98.  Extra flag --size WxH
// overrides the canvas size and --stats prints ray counts and timing as JSON.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "rt_oracle.hpp"

using namespace orc;

int main(int argc, char** argv) {
  std::string scene_path, out_path = "./out.ppm";
  int jobs = 1;
  size_t ow = 0, oh = 0;
  bool stats = false;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto next = [&]() -> std::string {
      if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); }
      return argv[++i];
    };
    if (a == "-s" || a == "--scene") scene_path = next();
    else if (a == "-o" || a == "--output") out_path = next();
    else if (a == "-j" || a == "--jobs") {
      std::string v = next();
      char* e = nullptr;
      long j = strtol(v.c_str(), &e, 10);
      if (*e || v.empty() || j < 0) { fprintf(stderr, "-j or --jobs argument must be a positive integer.\n"); return 101; }
      jobs = (int)j;
    } else if (a == "--size") {
      std::string v = next();
      if (sscanf(v.c_str(), "%zux%zu", &ow, &oh) != 2) { fprintf(stderr, "--size wants WxH\n"); return 2; }
    } else if (a == "--stats") stats = true;
    else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
  }
  try {
    Scene scene;
    if (!scene_path.empty()) {
      std::ifstream f(scene_path, std::ios::binary);
      if (!f) { fprintf(stderr, "Failed to read scene JSON file \"%s\".\n", scene_path.c_str()); return 101; }
      std::stringstream ss;
      ss << f.rdbuf();
      scene = scene_from_json(ss.str(), "");
    } else {  // main.rs:72-98
      Shape sphere = Shape::sphere();
      Shape floor = Shape::plane();
      floor.set_transform(mat_translation(0.0, -4.0, 0.0));
      scene.world = World::empty_world();
      scene.world.light = point_light_new({0.85, 0.8, 0.65}, point(-10.0, 10.0, -10.0));
      scene.world.objects.push_back(sphere);
      scene.world.objects.push_back(floor);
      scene.world.finalize();
      scene.camera = Camera::make(1920, 1080, M_PI / 3.0,
                                  mat_view_transform(point(0.0, 1.5, -5.0), point(0.0, 1.0, 0.0),
                                                     vector(0.0, 1.0, 0.0)));
    }
    if (ow && oh) scene.camera = Camera::make(ow, oh, scene.camera.field_of_view, scene.camera.transform);
    printf("Rendering using %d threads...\n", jobs);
    fflush(stdout);
    RenderOut out;
    auto t0 = std::chrono::steady_clock::now();
    parallel_render(scene.world, scene.camera, jobs, 0, 1, 0, 1, &out);
    auto t1 = std::chrono::steady_clock::now();
    std::string ppm = ppm_encode(scene.camera.hsize, scene.camera.vsize, out.rgb.data());
    std::ofstream of(out_path, std::ios::binary);
    if (!of) { fprintf(stderr, "could not create %s\n", out_path.c_str()); return 101; }
    of.write(ppm.data(), (std::streamsize)ppm.size());
    printf("...done.\n\n");
    printf("Saved render to \"%s\".\n", out_path.c_str());
    if (stats) {
      double sec = std::chrono::duration<double>(t1 - t0).count();
      uint64_t rays = out.counts.primary + out.counts.shadow + out.counts.reflect + out.counts.refract;
      printf("{\"render_s\": %.6f, \"rays\": %llu, \"rays_per_s\": %.1f, \"primary\": %llu, \"shadow\": %llu, "
             "\"reflect\": %llu, \"refract\": %llu, \"jobs\": %d}\n",
             sec, (unsigned long long)rays, rays / sec, (unsigned long long)out.counts.primary,
             (unsigned long long)out.counts.shadow, (unsigned long long)out.counts.reflect,
             (unsigned long long)out.counts.refract, jobs);
    }
  } catch (const std::exception& e) {
    fprintf(stderr, "thread 'main' panicked at '%s'\n", e.what());
    return 101;
  }
  return 0;
}

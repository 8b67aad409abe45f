// Command-line renderer with the reference binary's flags (src/main.rs:19-56):
//   -s/--scene FILE   scene JSON (without it: the built-in sphere + floor of src/main.rs:72-98)
//   -o/--output FILE  PPM output, default ./out.ppm (src/consts.rs:4)
//   -j/--jobs N       number of workers, default 1; here a worker is one MI355X
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "rt_host.hpp"

using namespace rthost;

int main(int argc, char** argv) {
  std::string scene_path, out_path = "./out.ppm";
  size_t jobs = 1;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto value = [&]() -> std::string {
      if (i + 1 >= argc) {
        fprintf(stderr, "error: The argument '%s' requires a value but none was supplied\n", a.c_str());
        exit(2);
      }
      return argv[++i];
    };
    if (a == "-s" || a == "--scene") scene_path = value();
    else if (a == "-o" || a == "--output") out_path = value();
    else if (a == "-j" || a == "--jobs") {
      std::string v = value();
      char* e = nullptr;
      unsigned long long n = strtoull(v.c_str(), &e, 10);
      if (v.empty() || *e || v[0] == '-') {
        fprintf(stderr, "thread 'main' panicked at '-j or --jobs argument must be a positive integer.'\n");
        return 101;
      }
      jobs = (size_t)n;
    } else {
      fprintf(stderr, "error: Found argument '%s' which wasn't expected\n", a.c_str());
      return 2;
    }
  }
  try {
    Scene scene;
    if (!scene_path.empty()) {
      std::ifstream f(scene_path, std::ios::binary);
      if (!f) throw Panic("Failed to read scene JSON file \"" + scene_path + "\".");
      std::stringstream ss;
      ss << f.rdbuf();
      scene = scene_from_json(ss.str(), "");
    } else {
      Shape sphere;
      sphere.type = RT_SPHERE;
      Shape floor;
      floor.type = RT_PLANE;
      floor.transform.m[7] = -4.0;  // translation(0, -4, 0)
      scene.world.objects = {sphere, floor};
      double light[3] = {0.85, 0.8, 0.65}, lpos[3] = {-10.0, 10.0, -10.0};
      memcpy(scene.world.light_source.intensity, light, sizeof(light));
      memcpy(scene.world.light_source.position, lpos, sizeof(lpos));
      double from[3] = {0.0, 1.5, -5.0}, to[3] = {0.0, 1.0, 0.0}, up[4] = {0.0, 1.0, 0.0, 0.0};
      scene.camera = Camera::make(1920, 1080, M_PI / 3.0, view_transform(from, to, up));
    }
    parallel_render(scene.world, scene.camera, jobs, out_path);
  } catch (const std::exception& e) {
    fprintf(stderr, "thread 'main' panicked at '%s'\n", e.what());
    return 101;
  }
  return 0;
}

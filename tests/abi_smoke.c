/* The C ABI of include/rt_amd.h from plain C (C99, no C++ and no GPU runtime headers): what a binding in
 * another language sees.  Flattens scenes/example.json by hand (a unit sphere and the plane y = -2,
 * default materials, src/scene.rs:95-207 / src/light.rs:96-112), creates the scene, renders one frame
 * into host memory and writes the bytes to a file; the test compares them with the host library's frame.
 *
 * usage: abi_smoke <camera.bin: one rt_camera> <out.rgb8>
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rt_amd.h"

static void identity(double* m) {
  int i;
  for (i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? 1.0 : 0.0;
}

static int fail(const char* what) {
  fprintf(stderr, "abi_smoke: %s: %s\n", what, rt_last_error());
  return 1;
}

int main(int argc, char** argv) {
  rt_camera cam;
  rt_scene_desc desc;
  rt_leaf leaves[2];
  rt_xform xforms[2];
  rt_material materials[1];
  rt_scene* scene = NULL;
  rt_stats stats;
  uint8_t* rgb8;
  size_t n;
  FILE* f;
  int rc;

  if (argc != 3) {
    fprintf(stderr, "usage: abi_smoke camera.bin out.rgb8\n");
    return 2;
  }
  f = fopen(argv[1], "rb");
  if (!f || fread(&cam, sizeof cam, 1, f) != 1) {
    fprintf(stderr, "abi_smoke: cannot read %s\n", argv[1]);
    return 2;
  }
  fclose(f);

  memset(&desc, 0, sizeof desc);
  memset(leaves, 0, sizeof leaves);
  memset(xforms, 0, sizeof xforms);
  memset(materials, 0, sizeof materials);
  /* light (src/scene.rs:77-84) */
  desc.light.intensity[0] = 0.85; desc.light.intensity[1] = 0.8; desc.light.intensity[2] = 0.65;
  desc.light.position[0] = -10.0; desc.light.position[1] = 10.0; desc.light.position[2] = -10.0;
  /* Material::default (src/light.rs:96-112), shared by both shapes */
  materials[0].color[0] = materials[0].color[1] = materials[0].color[2] = 1.0;
  materials[0].ambient = 0.1; materials[0].diffuse = 0.9; materials[0].specular = 0.9; materials[0].shininess = 200.0;
  materials[0].reflective = 0.0; materials[0].refractive_index = 1.0; materials[0].transparency = 0.0;
  materials[0].pattern = -1;
  /* sphere at the identity */
  identity(xforms[0].ray_inv); identity(xforms[0].full_inv); identity(xforms[0].fwd);
  leaves[0].type = RT_SPHERE; leaves[0].material = 0; leaves[0].xform = 0; leaves[0].gate = -1; leaves[0].tri = -1; leaves[0].csg = -1;
  /* plane moved to y = -2: the inverse moves rays up by 2 */
  identity(xforms[1].ray_inv); identity(xforms[1].full_inv); identity(xforms[1].fwd);
  xforms[1].ray_inv[7] = 2.0; xforms[1].full_inv[7] = 2.0; xforms[1].fwd[7] = -2.0;
  leaves[1].type = RT_PLANE; leaves[1].material = 0; leaves[1].xform = 1; leaves[1].gate = -1; leaves[1].tri = -1; leaves[1].csg = -1;
  leaves[1].plane_normal[1] = 1.0;
  desc.n_leaves = 2; desc.leaves = leaves;
  desc.n_xforms = 2; desc.xforms = xforms;
  desc.n_materials = 1; desc.materials = materials;

  if (rt_device_count() < 1) return fail("rt_device_count");
  if (rt_scene_create(&desc, 0, &scene) != RT_OK) return fail("rt_scene_create");
  /* errors come back as codes with a message, never as aborts */
  rc = rt_render_host(scene, &cam, 0, 0, 0, NULL, NULL, NULL, NULL);
  if (rc != RT_ERR_INVALID || strlen(rt_last_error()) == 0) {
    fprintf(stderr, "abi_smoke: n_parts = 0 returned %d\n", rc);
    return 1;
  }
  n = (size_t)cam.hsize * cam.vsize * 3;
  rgb8 = (uint8_t*)malloc(n);
  if (!rgb8) return 2;
  if (rt_render_host(scene, &cam, 0, 1, 0, rgb8, NULL, NULL, &stats) != RT_OK) return fail("rt_render_host");
  if (stats.primary != (uint64_t)cam.hsize * cam.vsize || stats.shadow == 0 || stats.reflect != 0) {
    fprintf(stderr, "abi_smoke: unexpected ray counts\n");
    return 1;
  }
  rt_scene_destroy(scene);
  f = fopen(argv[2], "wb");
  if (!f || fwrite(rgb8, 1, n, f) != n) return 2;
  fclose(f);
  free(rgb8);
  printf("abi_smoke: %ux%u primary %llu shadow %llu\n", cam.hsize, cam.vsize, (unsigned long long)stats.primary,
         (unsigned long long)stats.shadow);
  return 0;
}

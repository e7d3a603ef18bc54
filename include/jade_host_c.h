/*
 * jade_host_c.h — C entry points of the host scene pipeline (libjade_host.so).
 *
 * This is NOT the drop-in boundary (that is jade_rt.h); it is the repo's own
 * host side — OBJ / render_args.txt loading, procedural stand-ins for the
 * reference's git-ignored assets, the SAH BVH builder, camera and image
 * writers (roles of PathTrace.cu:355-628, 1487-1612, 74-106 and
 * PathTrace.cpp:343-359, 684-687, 883-918) — exposed with C linkage so the
 * CLI, the Python package and the tests can produce a jade_scene_desc.
 * All functions returning int return 0 on success; jadeh_last_error() holds
 * the message otherwise.
 */
#ifndef JADE_HOST_C_H
#define JADE_HOST_C_H

#include <stdint.h>

#include "jade_rt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* == Material, PathTrace.cu:293-301 */
typedef struct jadeh_material {
  float emissive[3];
  float brdf[3];
  int32_t reflex_mode;
  int32_t refract_mode;
  float refract_rate[3];
  float refract_albedo[3];
  float refract_index;
} jadeh_material;

typedef struct jadeh_config {
  int32_t width, height, spp;
  float eye[3];
  float camera[16];
} jadeh_config;

typedef struct jadeh_builder jadeh_builder;
typedef struct jadeh_scene jadeh_scene;

const char* jadeh_last_error(void);

jadeh_builder* jadeh_builder_new(void);
void jadeh_builder_free(jadeh_builder* b);
int jadeh_builder_triangle_count(const jadeh_builder* b);

/* one object = one readObj() call: optional normalise (reference quirk
 * included), 4x4 transform ([col][row] order), flat normals */
int jadeh_builder_add_mesh(jadeh_builder* b, const float* verts, int nv, const int* idx, int nt,
                           const jadeh_material* mat, const float* trans16, int normalize);
int jadeh_builder_add_obj(jadeh_builder* b, const char* path, const jadeh_material* mat,
                          const float* trans16, int normalize);
/* kind: "box" | "quad" | "geodesic" (param = frequency, 20*f^2 triangles) |
 *       "statue" | "dragon" (param = frequency, seed) */
int jadeh_builder_add_proc(jadeh_builder* b, const char* kind, int param, unsigned seed,
                           const jadeh_material* mat, const float* trans16, int normalize);
int jadeh_write_proc_obj(const char* kind, int param, unsigned seed, const char* path);

int jadeh_builder_set_env_constant(jadeh_builder* b, float r, float g, float bl);
int jadeh_builder_set_env_sky(jadeh_builder* b, int w, int h);
int jadeh_builder_set_env_data(jadeh_builder* b, int w, int h, const float* rgb);
int jadeh_builder_set_env_hdr(jadeh_builder* b, const char* path); /* Radiance RGBE */

/* built-in configurations: "tiny", "tinyjade", "C1".."C5"  */
int jadeh_builder_config(jadeh_builder* b, const char* name, jadeh_config* out);
/* render_args.txt (PathTrace.cu:1487-1525); OBJ paths relative to the file */
int jadeh_builder_load_render_args(jadeh_builder* b, const char* path, jadeh_config* out);

/* prefix sums + SAH BVH (leaf_size 8 in the reference) + encode */
jadeh_scene* jadeh_builder_build(jadeh_builder* b, int leaf_size);
/* the same flattening around a BVH built elsewhere (include/jade_bvh.h: the GPU LBVH builder):
 * jadeh_builder_triangles() gives that builder its input (original order), and
 * jadeh_builder_build_with_bvh() takes its output (order[i] = original index of sorted triangle i) */
int jadeh_builder_triangles(const jadeh_builder* b, jade_triangle* out, int capacity);
jadeh_scene* jadeh_builder_build_with_bvh(jadeh_builder* b, const int32_t* order, const jade_bvh_node* nodes, int n_nodes);
void jadeh_scene_free(jadeh_scene* s);
void jadeh_scene_desc(const jadeh_scene* s, jade_scene_desc* out); /* pointers owned by s */
int jadeh_scene_bvh_depth(const jadeh_scene* s);
double jadeh_scene_build_seconds(const jadeh_scene* s);

void jadeh_transform_matrix(const float rot_deg[3], const float trans[3], const float scale[3], float out16[16]);
void jadeh_camera_orbit(float r, float up_deg, float rot_deg, const float center[3], float eye_out[3],
                        float cam_out[16]);

int jadeh_write_bmp(const char* path, const uint8_t* bgr, int w, int h);
int jadeh_write_ppm(const char* path, const uint8_t* bgr, int w, int h);
int jadeh_write_pfm(const char* path, const float* rgb, int w, int h);

#ifdef __cplusplus
}
#endif
#endif /* JADE_HOST_C_H */

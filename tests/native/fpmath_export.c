/* Exposes the inline routines of include/jade_fpmath.h to ctypes (test-only). */
#include "jade_fpmath.h"
#include "jade_rt.h"
#include <stddef.h>

#define EXPORT __attribute__((visibility("default")))
EXPORT int t_selftest(float one) { return jade_fp_selftest(one); }
EXPORT void t_sincos(const float* x, float* s, float* c, int n) { for (int i = 0; i < n; ++i) jade_sincosf(x[i], &s[i], &c[i]); }
EXPORT void t_log2(const float* x, float* y, int n) { for (int i = 0; i < n; ++i) y[i] = jade_log2f(x[i]); }
EXPORT void t_exp2(const float* x, float* y, int n) { for (int i = 0; i < n; ++i) y[i] = jade_exp2f(x[i]); }
EXPORT void t_pow(const float* a, const float* b, float* y, int n) { for (int i = 0; i < n; ++i) y[i] = jade_powf(a[i], b[i]); }
EXPORT void t_atan2(const float* a, const float* b, float* y, int n) { for (int i = 0; i < n; ++i) y[i] = jade_atan2f(a[i], b[i]); }
EXPORT void t_asin(const float* x, float* y, int n) { for (int i = 0; i < n; ++i) y[i] = jade_asinf(x[i]); }
EXPORT void t_floor(const float* x, float* y, int n) { for (int i = 0; i < n; ++i) y[i] = jade_floorf(x[i]); }
EXPORT void t_rand(uint32_t seed, uint32_t* states, float* u, int n) { for (int i = 0; i < n; ++i) { u[i] = jade_rand(&seed); states[i] = seed; } }
EXPORT uint32_t t_seed(uint32_t x, uint32_t y, uint32_t f) { return jade_rng_seed(x, y, f); }
EXPORT float t_dot(const float* a, const float* b) { return jv_dot(jv(a[0], a[1], a[2]), jv(b[0], b[1], b[2])); }
EXPORT float t_mixed(const float* a, const float* b, const float* c) { return jv_mixed(jv(a[0], a[1], a[2]), jv(b[0], b[1], b[2]), jv(c[0], c[1], c[2])); }
EXPORT void t_cross(const float* a, const float* b, float* o) { jvec3 r = jv_cross(jv(a[0], a[1], a[2]), jv(b[0], b[1], b[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
EXPORT void t_normalize(const float* a, float* o) { jvec3 r = jv_normalize(jv(a[0], a[1], a[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
EXPORT void t_transform(const float* v, float f4, const float* m, float* o) { jvec3 r = jade_transform(jv(v[0], v[1], v[2]), f4, m); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
EXPORT float t_fmin(float a, float b) { return jade_fminf(a, b); }
EXPORT float t_fmax(float a, float b) { return jade_fmaxf(a, b); }
/* layout of the boundary structs as the C compiler sees them */
EXPORT void t_layout(int* out) {
  out[0] = (int)sizeof(jade_triangle); out[1] = (int)sizeof(jade_bvh_node); out[2] = (int)sizeof(jade_obj_seg);
  out[3] = (int)sizeof(jade_scene_desc); out[4] = (int)sizeof(jade_render_params); out[5] = (int)sizeof(jade_stats);
  out[6] = (int)offsetof(jade_triangle, norm); out[7] = (int)offsetof(jade_triangle, emissive);
  out[8] = (int)offsetof(jade_triangle, brdf); out[9] = (int)offsetof(jade_triangle, reflex_mode);
  out[10] = (int)offsetof(jade_triangle, refract_rate); out[11] = (int)offsetof(jade_triangle, refract_albedo);
  out[12] = (int)offsetof(jade_triangle, refract_index); out[13] = (int)offsetof(jade_bvh_node, aa);
}

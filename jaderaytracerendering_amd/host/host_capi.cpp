// host_capi.cpp — C entry points over the host scene pipeline, for the CLI,
// the Python bindings (ctypes) and the tests.  Declared in jade_host_c.h.
#include <cstring>
#include <string>

#include "jade_host.hpp"
#include "jade_host_c.h"

using namespace jadehost;

static thread_local std::string g_err;
static int fail(const std::string& m) {
  g_err = m;
  return 1;
}

struct jadeh_builder {
  SceneBuilder b;
};
struct jadeh_scene {
  BuiltScene s;
};

static Material to_mat(const jadeh_material* m) {
  Material r;
  if (!m) return r;
  memcpy(r.emissive, m->emissive, sizeof r.emissive);
  memcpy(r.brdf, m->brdf, sizeof r.brdf);
  r.reflex_mode = m->reflex_mode;
  r.refract_mode = m->refract_mode;
  memcpy(r.refract_rate, m->refract_rate, sizeof r.refract_rate);
  memcpy(r.refract_albedo, m->refract_albedo, sizeof r.refract_albedo);
  r.refract_index = m->refract_index;
  return r;
}
static Mat4 to_mat4(const float* t) {
  Mat4 r = Mat4::identity();
  if (t) memcpy(r.m, t, sizeof r.m);
  return r;
}

extern "C" {

const char* jadeh_last_error(void) { return g_err.c_str(); }

jadeh_builder* jadeh_builder_new(void) { return new jadeh_builder(); }
void jadeh_builder_free(jadeh_builder* b) { delete b; }
int jadeh_builder_triangle_count(const jadeh_builder* b) { return b ? b->b.triangle_count() : 0; }

int jadeh_builder_add_mesh(jadeh_builder* b, const float* verts, int nv, const int* idx, int nt, const jadeh_material* mat,
                           const float* trans16, int normalize) {
  if (!b || !verts || !idx || nv <= 0 || nt <= 0) return fail("add_mesh: bad arguments");
  Mesh m;
  m.vertices.resize(nv);
  for (int i = 0; i < nv; ++i) m.vertices[i] = jv(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
  m.indices.assign(idx, idx + 3 * (size_t)nt);
  for (int i : m.indices)
    if (i < 0 || i >= nv) return fail("add_mesh: index out of range");
  b->b.add_mesh(m, to_mat(mat), to_mat4(trans16), normalize != 0);
  return 0;
}

int jadeh_builder_add_obj(jadeh_builder* b, const char* path, const jadeh_material* mat, const float* trans16, int normalize) {
  if (!b || !path) return fail("add_obj: bad arguments");
  Mesh m;
  std::string err;
  if (!load_obj(path, m, err)) return fail(err);
  b->b.add_mesh(m, to_mat(mat), to_mat4(trans16), normalize != 0);
  return 0;
}

int jadeh_builder_add_proc(jadeh_builder* b, const char* kind, int param, unsigned seed, const jadeh_material* mat,
                           const float* trans16, int normalize) {
  if (!b || !kind) return fail("add_proc: bad arguments");
  std::string k = kind;
  Mesh m;
  if (k == "box") m = make_box();
  else if (k == "quad") m = make_quad(jv(-0.5f, 0, -0.5f), jv(0.5f, 0, -0.5f), jv(0.5f, 0, 0.5f), jv(-0.5f, 0, 0.5f));
  else if (k == "geodesic") m = make_geodesic(param);
  else if (k == "statue") m = make_statue(param, seed, 0);
  else if (k == "dragon") m = make_statue(param, seed, 1);
  else return fail("add_proc: unknown kind " + k);
  b->b.add_mesh(m, to_mat(mat), to_mat4(trans16), normalize != 0);
  return 0;
}

int jadeh_write_proc_obj(const char* kind, int param, unsigned seed, const char* path) {
  std::string k = kind ? kind : "";
  Mesh m;
  if (k == "box") m = make_box();
  else if (k == "quad") m = make_quad(jv(-0.5f, 0, -0.5f), jv(0.5f, 0, -0.5f), jv(0.5f, 0, 0.5f), jv(-0.5f, 0, 0.5f));
  else if (k == "geodesic") m = make_geodesic(param);
  else if (k == "statue") m = make_statue(param, seed, 0);
  else if (k == "dragon") m = make_statue(param, seed, 1);
  else return fail("write_proc_obj: unknown kind " + k);
  return write_obj(path, m) ? 0 : fail(std::string("cannot write ") + path);
}

int jadeh_builder_set_env_constant(jadeh_builder* b, float r, float g, float bl) {
  if (!b) return fail("null builder");
  b->b.set_env(make_env_constant(r, g, bl));
  return 0;
}
int jadeh_builder_set_env_sky(jadeh_builder* b, int w, int h) {
  if (!b || w <= 0 || h <= 0) return fail("set_env_sky: bad arguments");
  b->b.set_env(make_env_sky(w, h));
  return 0;
}
int jadeh_builder_set_env_data(jadeh_builder* b, int w, int h, const float* rgb) {
  if (!b || w <= 0 || h <= 0 || !rgb) return fail("set_env_data: bad arguments");
  EnvMap e;
  e.width = w;
  e.height = h;
  e.rgb.assign(rgb, rgb + (size_t)3 * w * h);
  b->b.set_env(std::move(e));
  return 0;
}
int jadeh_builder_set_env_hdr(jadeh_builder* b, const char* path) {
  if (!b || !path) return fail("set_env_hdr: bad arguments");
  EnvMap e;
  std::string err;
  if (!load_hdr(path, e, err)) return fail(err);
  b->b.set_env(std::move(e));
  return 0;
}

int jadeh_builder_config(jadeh_builder* b, const char* name, jadeh_config* out) {
  if (!b || !name || !out) return fail("config: bad arguments");
  Config c;
  std::string err;
  if (!make_config(name, b->b, c, err)) return fail(err);
  out->width = c.width;
  out->height = c.height;
  out->spp = c.spp;
  memcpy(out->eye, c.eye, sizeof c.eye);
  memcpy(out->camera, c.camera, sizeof c.camera);
  return 0;
}

int jadeh_builder_load_render_args(jadeh_builder* b, const char* path, jadeh_config* out) {
  if (!b || !path || !out) return fail("load_render_args: bad arguments");
  RenderArgs ra;
  std::string err;
  if (!read_render_args(path, ra, err)) return fail(err);
  std::string dir = path;
  size_t slash = dir.find_last_of('/');
  dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
  for (const RenderArgsObject& o : ra.objects) {
    Mesh m;
    std::string file = (!o.file.empty() && o.file[0] == '/') ? o.file : dir + o.file;
    if (!load_obj(file, m, err)) return fail(err);
    b->b.add_mesh(m, o.material, o.trans, o.normalize);
  }
  memset(out, 0, sizeof *out);
  memcpy(out->eye, ra.eye, sizeof ra.eye);
  memcpy(out->camera, ra.camera, sizeof ra.camera);
  return 0;
}

jadeh_scene* jadeh_builder_build(jadeh_builder* b, int leaf_size) {
  if (!b || b->b.triangle_count() == 0) {
    fail("build: empty scene");
    return nullptr;
  }
  jadeh_scene* s = new jadeh_scene();
  s->s = b->b.build(leaf_size > 0 ? leaf_size : 8);
  return s;
}
int jadeh_builder_triangles(const jadeh_builder* b, jade_triangle* out, int capacity) {
  if (!b || !out) return fail("builder_triangles: bad arguments");
  std::vector<jade_triangle> v = b->b.triangles_original();
  if ((int)v.size() > capacity) return fail("builder_triangles: buffer too small");
  memcpy(out, v.data(), v.size() * sizeof(jade_triangle));
  return 0;
}

jadeh_scene* jadeh_builder_build_with_bvh(jadeh_builder* b, const int32_t* order, const jade_bvh_node* nodes, int n_nodes) {
  if (!b || b->b.triangle_count() == 0) {
    fail("build_with_bvh: empty scene");
    return nullptr;
  }
  jadeh_scene* s = new jadeh_scene();
  std::string err;
  if (!b->b.build_with_bvh(order, nodes, n_nodes, s->s, err)) {
    fail(err);
    delete s;
    return nullptr;
  }
  return s;
}

void jadeh_scene_free(jadeh_scene* s) { delete s; }
void jadeh_scene_desc(const jadeh_scene* s, jade_scene_desc* out) {
  if (s && out) *out = s->s.desc();
}
int jadeh_scene_bvh_depth(const jadeh_scene* s) { return s ? s->s.bvh_depth : 0; }
double jadeh_scene_build_seconds(const jadeh_scene* s) { return s ? s->s.build_seconds : 0; }

void jadeh_transform_matrix(const float rot_deg[3], const float trans[3], const float scale[3], float out16[16]) {
  Mat4 m = transform_matrix(rot_deg, trans, scale);
  memcpy(out16, m.m, sizeof m.m);
}
void jadeh_camera_orbit(float r, float up_deg, float rot_deg, const float center[3], float eye_out[3], float cam_out[16]) {
  camera_orbit(r, up_deg, rot_deg, center, eye_out, cam_out);
}

int jadeh_write_bmp(const char* path, const uint8_t* bgr, int w, int h) { return write_bmp(path, bgr, w, h) ? 0 : fail("write_bmp failed"); }
int jadeh_write_ppm(const char* path, const uint8_t* bgr, int w, int h) { return write_ppm(path, bgr, w, h) ? 0 : fail("write_ppm failed"); }
int jadeh_write_pfm(const char* path, const float* rgb, int w, int h) { return write_pfm(path, rgb, w, h) ? 0 : fail("write_pfm failed"); }

}  // extern "C"

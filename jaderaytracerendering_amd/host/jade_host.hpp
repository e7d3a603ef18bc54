// jade_host.hpp — host-side scene pipeline (the repo's own C++).
//
// Everything ABOVE the drop-in boundary of include/jade_rt.h: mesh loading and
// procedural stand-ins for the git-ignored reference assets, the reference's
// SAH BVH conventions, scene flattening into the boundary's arrays, camera,
// environment map, image writers and the render_args.txt hand-off file.
// Mirrors the roles of PathTrace.cu:355-628, 1487-1612 and
// PathTrace.cpp:343-359, 684-687, 883-918 without sharing their code.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "jade_fpmath.h"
#include "jade_rt.h"

namespace jadehost {

// == Material, PathTrace.cu:293-301
struct Material {
  float emissive[3] = {0, 0, 0};
  float brdf[3] = {0.8f, 0.8f, 0.8f};
  int reflex_mode = JADE_DIFFUSE;
  int refract_mode = JADE_NO_REFRACT;
  float refract_rate[3] = {0.8f, 0.8f, 0.8f};
  float refract_albedo[3] = {0.8f, 0.8f, 0.8f};
  float refract_index = 1.0f;
};

struct Mat4 {
  float m[16];  // [col][row] memory order, like glm and camera_transform_dv
  static Mat4 identity();
};
Mat4 mul(const Mat4& a, const Mat4& b);
// getTransformMatrix(rotateCtrl(deg), translateCtrl, scaleCtrl) = T * Rx * Ry * Rz * S
// (PathTrace.cpp:343-359)
Mat4 transform_matrix(const float rot_deg[3], const float trans[3], const float scale[3]);

struct Mesh {
  std::vector<jvec3> vertices;
  std::vector<int> indices;  // 3 per triangle, 0-based
};

// -- mesh sources -----------------------------------------------------------
// readObj's parser (PathTrace.cu:378-408): '#' lines skipped, '/' -> ' ',
// "v x y z", "f a b c" (first three integers only).
bool load_obj(const std::string& path, Mesh& out, std::string& err);
bool write_obj(const std::string& path, const Mesh& mesh);
Mesh make_quad(jvec3 a, jvec3 b, jvec3 c, jvec3 d);
Mesh make_box();                     // unit cube centred at 0, 12 triangles
Mesh make_geodesic(int freq);        // unit sphere, 20*freq^2 triangles, shared vertices
// Closed star-shaped "jade statue" stand-in: geodesic sphere displaced by a
// seeded multi-octave field.  style 0 = upright statue, 1 = elongated "dragon".
Mesh make_statue(int freq, uint32_t seed, int style);
void append(Mesh& dst, const Mesh& src);

// -- scene builder (host triangles, pre-BVH) -------------------------------
struct HostTriangle {  // == Triangle, PathTrace.cu:305-311
  int index;
  int obj_idx;
  jvec3 p1, p2, p3, norm;
  Material material;
};

struct EnvMap {
  int width = 0, height = 0;
  std::vector<float> rgb;  // interleaved, row 0 = top
};
EnvMap make_env_constant(float r, float g, float b);
EnvMap make_env_sky(int width, int height);  // gradient + sun lobe, values in [0, 10]
bool load_hdr(const std::string& path, EnvMap& out, std::string& err);  // Radiance RGBE

struct BuiltScene {
  std::vector<jade_triangle> triangles;  // BVH order
  std::vector<jade_bvh_node> nodes;      // [0] dummy, [1] root
  std::vector<int32_t> emit;
  std::vector<int32_t> mapping;          // original -> sorted
  std::vector<float> prefix;             // original order
  std::vector<jade_obj_seg> segs;
  EnvMap env;
  int bvh_depth = 0;
  double build_seconds = 0;
  jade_scene_desc desc() const;
};

class SceneBuilder {
 public:
  // The body of readObj after parsing (PathTrace.cu:410-456): optional
  // normalise-to-unit (with the reference's max/min quirk, :399-400, :411-423),
  // 4x4 transform, flat normals, one object segment.
  void add_mesh(const Mesh& mesh, const Material& mat, const Mat4& trans, bool normalize);
  void set_env(EnvMap env) { env_ = std::move(env); }
  int triangle_count() const { return (int)tris_.size(); }
  // Area prefix sums (PathTrace.cu:1539-1546), SAH BVH with leaf size 8
  // (:1557-1565), encode (:1570-1612).
  BuiltScene build(int leaf_size = 8) const;
  // The same flattening around a BVH built elsewhere (e.g. jade_bvh_build_lbvh on the GPU):
  // `order[i]` = original index of the triangle at sorted position i, `nodes` in the reference's
  // conventions.  triangles_original() is what such a builder takes as input.
  std::vector<jade_triangle> triangles_original() const;
  bool build_with_bvh(const int32_t* order, const jade_bvh_node* nodes, int n_nodes, BuiltScene& out, std::string& err) const;

 private:
  void finish(std::vector<HostTriangle>& sorted, BuiltScene& out) const;
  std::vector<HostTriangle> tris_;
  std::vector<jade_obj_seg> segs_;
  EnvMap env_;
};

// buildBVHwithSAH conventions (PathTrace.cu:497-628): reorders `tris`.
void build_bvh_sah(std::vector<HostTriangle>& tris, std::vector<jade_bvh_node>& nodes, int leaf_size);
int bvh_depth(const std::vector<jade_bvh_node>& nodes);

// -- camera (PathTrace.cpp:209-211, 684-687) ---------------------------------
// eye on the sphere of radius r around `center`, camera = inverse(lookAt).
void camera_orbit(float r, float up_angle_deg, float rotate_angle_deg, const float center[3], float eye_out[3],
                  float cam_out[16]);

// -- render_args.txt (writer PathTrace.cpp:883-918, reader PathTrace.cu:1487-1525)
struct RenderArgsObject {
  std::string file;
  Mat4 trans;
  Material material;
  bool normalize = false;
};
struct RenderArgs {
  float eye[3];
  float camera[16];
  std::vector<RenderArgsObject> objects;
};
bool read_render_args(const std::string& path, RenderArgs& out, std::string& err);
bool write_render_args(const std::string& path, const RenderArgs& in);

// -- image output -------------------------------------------------------------
// save_image (PathTrace.cu:74-106): 24-bit BMP, BGR, bottom-up, rows unpadded.
bool write_bmp(const std::string& path, const uint8_t* bgr, int width, int height);
bool write_ppm(const std::string& path, const uint8_t* bgr, int width, int height);  // P6, top-down RGB
bool write_pfm(const std::string& path, const float* rgb, int width, int height);    // PF, bottom-up

// -- built-in benchmark configurations (C1..C5, plus "tiny") ----
struct Config {
  std::string name;
  int width = 0, height = 0, spp = 0;
  float eye[3];
  float camera[16];
};
bool make_config(const std::string& name, SceneBuilder& builder, Config& cfg, std::string& err);

}  // namespace jadehost

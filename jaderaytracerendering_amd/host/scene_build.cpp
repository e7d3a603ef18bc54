// scene_build.cpp — meshes, transforms, SAH BVH, flattening to the ABI arrays.
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>
#include <unordered_map>

#include "jade_host.hpp"

namespace jadehost {

// ------------------------------------------------------------------ Mat4 ----

Mat4 Mat4::identity() {
  Mat4 r;
  for (int i = 0; i < 16; ++i) r.m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  return r;
}

Mat4 mul(const Mat4& a, const Mat4& b) {  // column-major a*b
  Mat4 r;
  for (int c = 0; c < 4; ++c)
    for (int row = 0; row < 4; ++row) {
      float s = 0;
      for (int k = 0; k < 4; ++k) s += a.m[4 * k + row] * b.m[4 * c + k];
      r.m[4 * c + row] = s;
    }
  return r;
}

static Mat4 rotation(float deg, int axis) {
  float s, c;
  jade_sincosf(deg * 0.017453292519943295f, &s, &c);
  Mat4 r = Mat4::identity();
  int a = (axis + 1) % 3, b = (axis + 2) % 3;
  r.m[4 * a + a] = c;
  r.m[4 * a + b] = s;
  r.m[4 * b + a] = -s;
  r.m[4 * b + b] = c;
  return r;
}

Mat4 transform_matrix(const float rot_deg[3], const float trans[3], const float scale[3]) {
  Mat4 S = Mat4::identity(), T = Mat4::identity();
  S.m[0] = scale[0]; S.m[5] = scale[1]; S.m[10] = scale[2];
  T.m[12] = trans[0]; T.m[13] = trans[1]; T.m[14] = trans[2];
  Mat4 R = mul(mul(rotation(rot_deg[0], 0), rotation(rot_deg[1], 1)), rotation(rot_deg[2], 2));
  return mul(mul(T, R), S);
}

// ---------------------------------------------------------------- meshes ----

bool load_obj(const std::string& path, Mesh& out, std::string& err) {
  std::ifstream fin(path);
  if (!fin.is_open()) {
    err = "File " + path + " open failed.";
    return false;
  }
  std::string line;
  while (std::getline(fin, line)) {
    if (!line.empty() && line[0] == '#') continue;
    for (char& ch : line)
      if (ch == '/') ch = ' ';
    std::istringstream sin(line);
    std::string type;
    sin >> type;
    if (type == "v") {
      float x = 0, y = 0, z = 0;
      sin >> x >> y >> z;
      out.vertices.push_back(jv(x, y, z));
    } else if (type == "f") {
      int v0 = 0, v1 = 0, v2 = 0;
      sin >> v0 >> v1 >> v2;
      out.indices.push_back(v0 - 1);
      out.indices.push_back(v1 - 1);
      out.indices.push_back(v2 - 1);
    }
  }
  for (int idx : out.indices)
    if (idx < 0 || idx >= (int)out.vertices.size()) {
      err = "File " + path + ": face index out of range";
      return false;
    }
  return true;
}

bool write_obj(const std::string& path, const Mesh& mesh) {
  FILE* f = fopen(path.c_str(), "w");
  if (!f) return false;
  fprintf(f, "# written by jade_host\n");
  for (const jvec3& v : mesh.vertices) fprintf(f, "v %.9g %.9g %.9g\n", v.x, v.y, v.z);
  for (size_t i = 0; i + 2 < mesh.indices.size(); i += 3)
    fprintf(f, "f %d %d %d\n", mesh.indices[i] + 1, mesh.indices[i + 1] + 1, mesh.indices[i + 2] + 1);
  fclose(f);
  return true;
}

Mesh make_quad(jvec3 a, jvec3 b, jvec3 c, jvec3 d) {
  Mesh m;
  m.vertices = {a, b, c, d};
  m.indices = {0, 1, 2, 0, 2, 3};
  return m;
}

Mesh make_box() {
  Mesh m;
  for (int i = 0; i < 8; ++i) m.vertices.push_back(jv((i & 1) ? 0.5f : -0.5f, (i & 2) ? 0.5f : -0.5f, (i & 4) ? 0.5f : -0.5f));
  static const int q[6][4] = {{0, 2, 3, 1}, {4, 5, 7, 6}, {0, 1, 5, 4}, {2, 6, 7, 3}, {0, 4, 6, 2}, {1, 3, 7, 5}};
  for (auto& f : q) {
    int t[6] = {f[0], f[1], f[2], f[0], f[2], f[3]};
    m.indices.insert(m.indices.end(), t, t + 6);
  }
  return m;
}

void append(Mesh& dst, const Mesh& src) {
  int base = (int)dst.vertices.size();
  dst.vertices.insert(dst.vertices.end(), src.vertices.begin(), src.vertices.end());
  for (int i : src.indices) dst.indices.push_back(base + i);
}

namespace {
struct KeyHash {
  size_t operator()(const std::array<uint32_t, 3>& k) const {
    uint64_t h = 1469598103934665603ull;
    for (uint32_t v : k) { h ^= v; h *= 1099511628211ull; }
    return (size_t)h;
  }
};
}  // namespace

Mesh make_geodesic(int freq) {
  if (freq < 1) freq = 1;
  const float t = 1.6180339887498949f;
  const jvec3 iv[12] = {jv(-1, t, 0), jv(1, t, 0), jv(-1, -t, 0), jv(1, -t, 0), jv(0, -1, t), jv(0, 1, t),
                        jv(0, -1, -t), jv(0, 1, -t), jv(t, 0, -1), jv(t, 0, 1), jv(-t, 0, -1), jv(-t, 0, 1)};
  static const int fc[20][3] = {{0, 11, 5}, {0, 5, 1}, {0, 1, 7}, {0, 7, 10}, {0, 10, 11}, {1, 5, 9}, {5, 11, 4},
                                {11, 10, 2}, {10, 7, 6}, {7, 1, 8}, {3, 9, 4}, {3, 4, 2}, {3, 2, 6}, {3, 6, 8},
                                {3, 8, 9}, {4, 9, 5}, {2, 4, 11}, {6, 2, 10}, {8, 6, 7}, {9, 8, 1}};
  Mesh m;
  std::unordered_map<std::array<uint32_t, 3>, int, KeyHash> seen;
  seen.reserve((size_t)10 * freq * freq + 16);
  std::vector<int> row0, row1;
  const float inv = 1.0f / (float)freq;
  for (auto& f : fc) {
    const jvec3 A = iv[f[0]], B = iv[f[1]], C = iv[f[2]];
    // lattice point (i, j, k), i + j + k = freq, weight order A, B, C.  A term
    // with zero weight adds an exact +-0, and x + y commutes, so a point on a
    // shared edge gets the same bits from both faces.
    auto vertex = [&](int i, int j) -> int {
      int k = freq - i - j;
      jvec3 p = jv_add(jv_add(jv_scale(A, (float)i), jv_scale(B, (float)j)), jv_scale(C, (float)k));
      p = jv_scale(p, inv);
      p = jv_normalize(p);
      p = jv(p.x + 0.0f, p.y + 0.0f, p.z + 0.0f);  // -0 -> +0
      std::array<uint32_t, 3> key = {jade_f2u(p.x), jade_f2u(p.y), jade_f2u(p.z)};
      auto it = seen.find(key);
      if (it != seen.end()) return it->second;
      int id = (int)m.vertices.size();
      m.vertices.push_back(p);
      seen.emplace(key, id);
      return id;
    };
    // rows of constant i (weight of A), i = freq .. 0
    row0.assign(1, vertex(freq, 0));
    for (int i = freq - 1; i >= 0; --i) {
      int n = freq - i;  // row has n + 1 points, j = 0..n
      row1.resize(n + 1);
      for (int j = 0; j <= n; ++j) row1[j] = vertex(i, j);
      for (int j = 0; j < n; ++j) {
        int tri[3] = {row0[j], row1[j], row1[j + 1]};
        m.indices.insert(m.indices.end(), tri, tri + 3);
        if (j + 1 < n) {
          int tri2[3] = {row0[j], row1[j + 1], row0[j + 1]};
          m.indices.insert(m.indices.end(), tri2, tri2 + 3);
        }
      }
      row0.swap(row1);
    }
  }
  return m;
}

Mesh make_statue(int freq, uint32_t seed, int style) {
  Mesh m = make_geodesic(freq);
  uint32_t s = seed * 2654435761u + 12345u;
  auto rnd = [&]() { return jade_rand(&s); };
  struct Wave { jvec3 w; float freq, phase, amp; };
  std::vector<Wave> waves;
  const int octaves = style == 1 ? 7 : 6;
  float fr = style == 1 ? 2.5f : 1.7f, amp = style == 1 ? 0.16f : 0.20f;
  for (int o = 0; o < octaves; ++o) {
    for (int k = 0; k < 4; ++k) {
      Wave w;
      float cz = 2.0f * rnd() - 1.0f, ph = 6.2831853f * rnd();
      float sz = jade_sqrt(jade_fmaxf(0.0f, 1.0f - cz * cz));
      w.w = jv(sz * jade_cosf(ph), sz * jade_sinf(ph), cz);
      w.freq = fr * (0.8f + 0.4f * rnd());
      w.phase = 6.2831853f * rnd();
      w.amp = amp * (0.6f + 0.4f * rnd());
      waves.push_back(w);
    }
    fr *= 1.9f;
    amp *= 0.55f;
  }
  // style 0: long axis z (the reference stands its statue up with Rx(-90),
  // PathTrace.cpp:1002); style 1: long axis x, up = y ("loong", :997).
  const jvec3 shape = style == 1 ? jv(1.0f, 0.38f, 0.30f) : jv(0.42f, 0.36f, 1.0f);
  float base = 1e30f;
  for (jvec3& v : m.vertices) {
    float r = 1.0f;
    for (const Wave& w : waves) r += w.amp * jade_sinf(jv_dot(w.w, v) * w.freq * 3.0f + w.phase);
    // a waist / shoulders profile along the long axis keeps it statue-like
    float h = style == 1 ? v.x : v.z;
    r *= 1.0f + 0.25f * jade_sinf(h * 4.0f + 0.7f) - 0.12f * h * h;
    if (r < 0.25f) r = 0.25f;
    v = jv_mul(jv_scale(v, r), shape);
    base = std::min(base, style == 1 ? v.y : v.z);
  }
  // stand it on its base: the reference's normalisation (add_mesh) centres
  // y and z with the X centre, so an offset along the up axis survives it
  for (jvec3& v : m.vertices) {
    if (style == 1) v.y -= base; else v.z -= base;
  }
  return m;
}

// -------------------------------------------------------------- builder ----

static float host_tri_area(const HostTriangle& t) {  // size(), PathTrace.cu:459-465
  jvec3 c = jv_cross(jv_sub(t.p2, t.p1), jv_sub(t.p3, t.p1));
  return 0.5f * jade_sqrt(jv_dot(c, c));
}

void SceneBuilder::add_mesh(const Mesh& mesh, const Material& mat, const Mat4& trans, bool normalize) {
  std::vector<jvec3> vertices = mesh.vertices;
  // PathTrace.cu:370-375, 399-400: note maxy/maxz/miny/minz are derived from
  // maxx/minx and the CURRENT vertex only — reproduced as written.
  float maxx = -11451419.19f, maxy = -11451419.19f, maxz = -11451419.19f;
  float minx = 11451419.19f, miny = 11451419.19f, minz = 11451419.19f;
  for (const jvec3& v : vertices) {
    maxx = std::max(maxx, v.x); maxy = std::max(maxx, v.y); maxz = std::max(maxx, v.z);
    minx = std::min(minx, v.x); miny = std::min(minx, v.y); minz = std::min(minx, v.z);
  }
  if (normalize && !vertices.empty()) {
    float lenx = maxx - minx, leny = maxy - miny, lenz = maxz - minz;
    float maxaxis = std::max(lenx, std::max(leny, lenz));
    jvec3 center = jv((maxx + minx) / 2, (maxy + miny) / 2, (maxz + minz) / 2);
    for (jvec3& v : vertices) {
      v = jv_sub(v, center);
      v.x /= maxaxis; v.y /= maxaxis; v.z /= maxaxis;
    }
  }
  for (jvec3& v : vertices) v = jade_transform(v, 1.0f, trans.m);

  int offset = (int)tris_.size();
  int ntri = (int)mesh.indices.size() / 3;
  if (ntri == 0) return;
  int obj_idx = (int)segs_.size();
  tris_.resize(offset + ntri);
  jade_obj_seg seg;
  seg.begin_idx = offset;
  seg.end_idx = offset + ntri - 1;
  segs_.push_back(seg);
  for (int i = 0; i < ntri; ++i) {
    HostTriangle& t = tris_[offset + i];
    t.index = offset + i;
    t.obj_idx = obj_idx;
    t.p1 = vertices[mesh.indices[3 * i]];
    t.p2 = vertices[mesh.indices[3 * i + 1]];
    t.p3 = vertices[mesh.indices[3 * i + 2]];
    t.norm = jv_normalize(jv_cross(jv_sub(t.p2, t.p1), jv_sub(t.p3, t.p1)));
    t.material = mat;
  }
}

// ------------------------------------------------------------------ BVH ----

namespace {

struct TriBox { float lo[3], hi[3], centroid[3]; };

struct SahBuilder {
  std::vector<HostTriangle>& tris;
  std::vector<jade_bvh_node>& nodes;
  int leaf;
  std::vector<TriBox> box;    // indexed by position in `order`'s values
  std::vector<int> order;     // order[pos] = original slot in tris
  std::vector<float> lmin, lmax, rmin, rmax;
  std::vector<int> scratch;

  SahBuilder(std::vector<HostTriangle>& t, std::vector<jade_bvh_node>& n, int l) : tris(t), nodes(n), leaf(l) {}

  void sort_range(int l, int r, int axis) {
    // centre comparison of cmpx/cmpy/cmpz (PathTrace.cu:468-482); ties (which
    // std::sort leaves unspecified in the reference) fall back to the
    // triangle's original index so the tree is reproducible everywhere.
    std::sort(order.begin() + l, order.begin() + r + 1, [&](int a, int b) {
      float ca = box[a].centroid[axis], cb = box[b].centroid[axis];
      if (ca < cb) return true;
      if (cb < ca) return false;
      return a < b;
    });
  }

  int build(int l, int r) {
    if (l > r) return 0;
    nodes.emplace_back();
    int id = (int)nodes.size() - 1;
    {
      jade_bvh_node& nd = nodes[id];
      nd.left = nd.right = nd.n = nd.index = 0;
      for (int k = 0; k < 3; ++k) { nd.aa[k] = 1145141919.0f; nd.bb[k] = -1145141919.0f; }
      for (int i = l; i <= r; ++i) {
        const TriBox& b = box[order[i]];
        for (int k = 0; k < 3; ++k) {
          nd.aa[k] = std::min(nd.aa[k], b.lo[k]);
          nd.bb[k] = std::max(nd.bb[k], b.hi[k]);
        }
      }
      if ((r - l + 1) <= leaf) {
        nd.n = r - l + 1;
        nd.index = l;
        return id;
      }
    }
    const float INF = 2147483647.0f;
    float Cost = INF;
    int Axis = 0;
    int Split = (l + r) / 2;
    int cnt = r - l + 1;
    for (int axis = 0; axis < 3; ++axis) {
      sort_range(l, r, axis);
      for (int i = l; i <= r; ++i) {
        const TriBox& b = box[order[i]];
        int o = 3 * (i - l), p = (i == l) ? o : o - 3;
        for (int k = 0; k < 3; ++k) {
          lmax[o + k] = (i == l) ? std::max(-INF, b.hi[k]) : std::max(lmax[p + k], b.hi[k]);
          lmin[o + k] = (i == l) ? std::min(INF, b.lo[k]) : std::min(lmin[p + k], b.lo[k]);
        }
      }
      for (int i = r; i >= l; --i) {
        const TriBox& b = box[order[i]];
        int o = 3 * (i - l), p = (i == r) ? o : o + 3;
        for (int k = 0; k < 3; ++k) {
          rmax[o + k] = (i == r) ? std::max(-INF, b.hi[k]) : std::max(rmax[p + k], b.hi[k]);
          rmin[o + k] = (i == r) ? std::min(INF, b.lo[k]) : std::min(rmin[p + k], b.lo[k]);
        }
      }
      float cost = INF;
      int split = l;
      for (int i = l; i <= r - 1; ++i) {
        int o = 3 * (i - l);
        float lenx = lmax[o] - lmin[o], leny = lmax[o + 1] - lmin[o + 1], lenz = lmax[o + 2] - lmin[o + 2];
        float leftS = (float)(2.0 * (double)((lenx * leny) + (lenx * lenz) + (leny * lenz)));
        float leftCost = leftS * (float)(i - l + 1);
        int q = o + 3;
        lenx = rmax[q] - rmin[q]; leny = rmax[q + 1] - rmin[q + 1]; lenz = rmax[q + 2] - rmin[q + 2];
        float rightS = (float)(2.0 * (double)((lenx * leny) + (lenx * lenz) + (leny * lenz)));
        float rightCost = rightS * (float)(r - i);
        float totalCost = leftCost + rightCost;
        if (totalCost < cost) { cost = totalCost; split = i; }
      }
      if (cost < Cost) { Cost = cost; Axis = axis; Split = split; }
    }
    (void)cnt;
    if (Axis != 2) sort_range(l, r, Axis);  // axis 2 is the order we are already in
    int left = build(l, Split);
    int right = build(Split + 1, r);
    nodes[id].left = left;
    nodes[id].right = right;
    return id;
  }
};

}  // namespace

void build_bvh_sah(std::vector<HostTriangle>& tris, std::vector<jade_bvh_node>& nodes, int leaf_size) {
  int n = (int)tris.size();
  SahBuilder b(tris, nodes, leaf_size);
  b.box.resize(n);
  b.order.resize(n);
  for (int i = 0; i < n; ++i) {
    const HostTriangle& t = tris[i];
    const float px[3][3] = {{t.p1.x, t.p2.x, t.p3.x}, {t.p1.y, t.p2.y, t.p3.y}, {t.p1.z, t.p2.z, t.p3.z}};
    for (int k = 0; k < 3; ++k) {
      b.box[i].lo[k] = std::min(px[k][0], std::min(px[k][1], px[k][2]));
      b.box[i].hi[k] = std::max(px[k][0], std::max(px[k][1], px[k][2]));
      b.box[i].centroid[k] = (px[k][0] + px[k][1] + px[k][2]) / 3.0f;  // (p1+p2+p3)/vec3(3,3,3)
    }
    b.order[i] = i;
  }
  b.lmin.resize(3 * (size_t)n); b.lmax.resize(3 * (size_t)n);
  b.rmin.resize(3 * (size_t)n); b.rmax.resize(3 * (size_t)n);
  b.build(0, n - 1);
  std::vector<HostTriangle> sorted(n);
  for (int i = 0; i < n; ++i) sorted[i] = tris[b.order[i]];
  tris.swap(sorted);
}

int bvh_depth(const std::vector<jade_bvh_node>& nodes) {
  if (nodes.size() < 2) return 0;
  std::vector<std::pair<int, int>> st;
  st.push_back({1, 1});
  int best = 0;
  while (!st.empty()) {
    auto [id, d] = st.back();
    st.pop_back();
    best = std::max(best, d);
    const jade_bvh_node& nd = nodes[id];
    if (nd.n > 0) continue;
    if (nd.left > 0) st.push_back({nd.left, d + 1});
    if (nd.right > 0) st.push_back({nd.right, d + 1});
  }
  return best;
}

jade_scene_desc BuiltScene::desc() const {
  jade_scene_desc d;
  std::memset(&d, 0, sizeof d);
  d.abi_version = JADE_ABI_VERSION;
  d.n_triangles = (int32_t)triangles.size();
  d.triangles = triangles.data();
  d.n_nodes = (int32_t)nodes.size();
  d.nodes = nodes.data();
  d.n_emit = (int32_t)emit.size();
  d.emit_indices = emit.data();
  d.index_mapping = mapping.data();
  d.prefix_area = prefix.data();
  d.n_objects = (int32_t)segs.size();
  d.obj_segs = segs.data();
  d.env_width = env.width;
  d.env_height = env.height;
  d.env_rgb = env.rgb.data();
  return d;
}

static void encode_triangle(const HostTriangle& t, jade_triangle& e) {
  const Material& m = t.material;
  e.obj_idx = t.obj_idx;
  e.p1[0] = t.p1.x; e.p1[1] = t.p1.y; e.p1[2] = t.p1.z;
  e.p2[0] = t.p2.x; e.p2[1] = t.p2.y; e.p2[2] = t.p2.z;
  e.p3[0] = t.p3.x; e.p3[1] = t.p3.y; e.p3[2] = t.p3.z;
  e.norm[0] = t.norm.x; e.norm[1] = t.norm.y; e.norm[2] = t.norm.z;
  std::memcpy(e.emissive, m.emissive, sizeof e.emissive);
  std::memcpy(e.brdf, m.brdf, sizeof e.brdf);
  e.reflex_mode = m.reflex_mode;
  e.refract_mode = m.refract_mode;
  std::memcpy(e.refract_rate, m.refract_rate, sizeof e.refract_rate);
  std::memcpy(e.refract_albedo, m.refract_albedo, sizeof e.refract_albedo);
  e.refract_index = m.refract_index;
}

// Everything of PathTrace.cu:1539-1612 except the BVH itself: prefix sums in original order,
// encode in sorted order, original -> sorted mapping, emitter list.
void SceneBuilder::finish(std::vector<HostTriangle>& tris, BuiltScene& out) const {
  size_t n = tris.size();
  out.segs = segs_;
  out.prefix.resize(n);
  for (const jade_obj_seg& seg : segs_) {
    float size_sum = 0;
    for (int idx = seg.begin_idx; idx <= seg.end_idx; ++idx) {
      size_sum += host_tri_area(tris_[idx]);
      out.prefix[idx] = size_sum;
    }
  }
  out.bvh_depth = bvh_depth(out.nodes);
  out.triangles.resize(n);
  out.mapping.resize(n);
  out.emit.clear();
  for (size_t i = 0; i < n; ++i) {
    const HostTriangle& t = tris[i];
    encode_triangle(t, out.triangles[i]);
    out.mapping[t.index] = (int32_t)i;
    const Material& m = t.material;
    if (m.emissive[0] > 1.5e-4f || m.emissive[1] > 1.5e-4f || m.emissive[2] > 1.5e-4f) out.emit.push_back((int32_t)i);
  }
  out.env = env_.width > 0 ? env_ : make_env_constant(0, 0, 0);
}

static jade_bvh_node dummy_node() {  // PathTrace.cu:1557-1563
  jade_bvh_node dummy;
  std::memset(&dummy, 0, sizeof dummy);
  dummy.left = 255; dummy.right = 128; dummy.n = 30;
  dummy.aa[0] = 1; dummy.aa[1] = 1; dummy.bb[1] = 1;
  return dummy;
}

BuiltScene SceneBuilder::build(int leaf_size) const {
  auto t0 = std::chrono::steady_clock::now();
  BuiltScene out;
  std::vector<HostTriangle> tris = tris_;
  out.nodes.push_back(dummy_node());  // root becomes node 1
  if (!tris.empty()) build_bvh_sah(tris, out.nodes, leaf_size);
  finish(tris, out);
  out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return out;
}

std::vector<jade_triangle> SceneBuilder::triangles_original() const {
  std::vector<jade_triangle> v(tris_.size());
  for (size_t i = 0; i < tris_.size(); ++i) encode_triangle(tris_[i], v[i]);
  return v;
}

bool SceneBuilder::build_with_bvh(const int32_t* order, const jade_bvh_node* nodes, int n_nodes, BuiltScene& out,
                                  std::string& err) const {
  auto t0 = std::chrono::steady_clock::now();
  const size_t n = tris_.size();
  if (!order || !nodes || n_nodes < 2) { err = "build_with_bvh: missing BVH"; return false; }
  std::vector<char> seen(n, 0);
  std::vector<HostTriangle> tris(n);
  for (size_t i = 0; i < n; ++i) {
    if (order[i] < 0 || (size_t)order[i] >= n || seen[order[i]]) { err = "build_with_bvh: order is not a permutation"; return false; }
    seen[order[i]] = 1;
    tris[i] = tris_[order[i]];
  }
  out = BuiltScene();
  out.nodes.assign(nodes, nodes + n_nodes);
  out.nodes[0] = dummy_node();
  finish(tris, out);
  out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return true;
}

}  // namespace jadehost

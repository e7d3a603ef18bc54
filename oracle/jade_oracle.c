/*
 * jade_oracle.c — CPU restatement of the reference's jade/BSSRDF integrator.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (libjade_hip.so, the
 * host library, the Python package) links, imports or calls this file; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference ships no tests, golden
 * images or known-answer vectors, its own output is
 * non-deterministic (31 cuRAND states raced by every thread,
 * PathTrace.cu:664-667, 1430-1431) and it cannot be built here (needs nvcc,
 * cuRAND, lib/hdrloader.h which is git-ignored, and an NVIDIA GPU).  This
 * file therefore follows the reference's SOURCE statement by statement and is
 * pinned by hand-derived closed-form unit values (tests/test_oracle_units.py)
 * and by the committed golden fixtures it generated (tests/golden/).
 *
 * What is restated, with the reference lines each function follows:
 *   hit_triangle   PathTrace.cu:705-754      hit_aabb     PathTrace.cu:758-771
 *   hit_array      PathTrace.cu:776-792      hit_bvh      PathTrace.cu:795-859
 *   sample_hdr     PathTrace.cu:686-702      gen_refract  PathTrace.cu:876-894
 *   tri_size       PathTrace.cu:897-903      path_tracing PathTrace.cu:905-1416
 *   render_pixel   PathTrace.cu:1418-1474    aces/pack    PathTrace.cu:680-682, 1461-1473
 *
 * Deliberate, documented departures:
 *   - RNG: the racy cuRAND XORWOW is replaced by the reference's own
 *     deterministic Wang hash (shaders/fshader_render.fsh:82-98), seeded per
 *     pixel and per sample (frame term = frame + sample index, as the
 *     reference's one-sample-per-frame preview does) and drawn in the textual
 *     order of the curand_uniform calls; a pixel's samples are summed in
 *     JADE_SAMPLE_LANES interleaved partial sums (see jade_rt.h);
 *   - libm / FMA contraction / texture filtering come from include/jade_fpmath.h
 *     and the bilinear fetch below (CUDA's 8-bit-weight tex2D is unreproducible);
 *   - width/height are run-time; for width != height the NDC x is scaled by
 *     width/height (factor exactly 1.0 on the reference's square images);
 *   - values the reference leaves undefined are defined: HitResult.index on a
 *     miss is 0 (PathTrace.cu:796-798), float->uchar of NaN/negative is 0.
 *
 * Build: -O2 -ffp-contract=off -mfma (see oracle/Makefile).
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "jade_fpmath.h"
#include "jade_rt.h"

#define INF_F 2147483647.0f /* #define INF 2147483647.0, PathTrace.cu:23 */

static __thread char g_err[256];
static int fail(int code, const char* msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  return code;
}

struct jade_scene {
  jade_scene_desc d; /* pointers below are owned copies */
  jade_triangle* tris;
  jade_bvh_node* nodes;
  int32_t* emit;
  int32_t* mapping;
  float* prefix;
  jade_obj_seg* segs;
  float* env;
  /* progressive render state */
  jade_render_params rp;
  int have_rp;
  float* sum;       /* [JADE_SAMPLE_LANES][pixel][3] partial radiance sums (untouched lanes stay unmapped) */
  int64_t spp_done;
  /* checker-only: jade_oracle_set_tile_filter() restricts a render to listed tiles of the caller's partition */
  uint8_t* tile_keep; /* one byte per tile id (ty * tiles_x + tx), NULL = no filter */
  int32_t tile_keep_n;
  int32_t* sum_slot;  /* with a filter: pixel -> its slot in the (compact) sums, -1 = not rendered; NULL = identity */
  size_t sum_pixels;  /* pixels the sums hold per lane */
};

typedef struct {
  uint64_t rays_primary, rays_secondary, nodes_visited, tris_tested, shaded_hits, samples;
  uint64_t rays_shadow, rays_env, rays_indirect, rays_mirror, rays_refract; /* rays_secondary by hitBVH call site */
} counters;

typedef struct {
  int isHit;
  int index;
  float distance;
  jvec3 hitPoint;
} HitResult;

typedef struct {
  jvec3 startPoint;
  jvec3 direction;
} Ray;

static jvec3 V3(const float* p) { return jv(p[0], p[1], p[2]); }

/* vec3 max/min with the reference's ternaries (PathTrace.cu:484-494): NaN
 * handling is whatever `a > b ? a : b` does. */
static jvec3 vmax3(jvec3 a, jvec3 b) {
  return jv(a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z);
}
static jvec3 vmin3(jvec3 a, jvec3 b) {
  return jv(a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z);
}

/* PathTrace.cu:705-754 */
static HitResult hit_triangle(const jade_triangle* triangle, Ray ray, int index) {
  HitResult res;
  res.isHit = 0;
  res.index = 0;
  res.distance = INF_F;
  res.hitPoint = jv(0, 0, 0);

  jvec3 p1 = V3(triangle->p1), p2 = V3(triangle->p2), p3 = V3(triangle->p3);
  jvec3 normal_direction = jv_normalize(ray.direction);
  jvec3 src_point = ray.startPoint;
  /* make shadow */
  jvec3 shadow_tri_a = jv_sub(p1, jv_scale(normal_direction, jv_dot(normal_direction, jv_sub(p1, src_point))));
  jvec3 shadow_tri_b = jv_sub(p2, jv_scale(normal_direction, jv_dot(normal_direction, jv_sub(p2, src_point))));
  jvec3 shadow_tri_c = jv_sub(p3, jv_scale(normal_direction, jv_dot(normal_direction, jv_sub(p3, src_point))));

  /* check in center */
  jvec3 vec_pa = jv_sub(shadow_tri_a, src_point);
  jvec3 vec_pb = jv_sub(shadow_tri_b, src_point);
  jvec3 vec_pc = jv_sub(shadow_tri_c, src_point);

  float papb = jv_mixed(normal_direction, vec_pa, vec_pb);
  float pbpc = jv_mixed(normal_direction, vec_pb, vec_pc);
  float pcpa = jv_mixed(normal_direction, vec_pc, vec_pa);
  if ((papb > 0 && pbpc > 0 && pcpa > 0) || (papb < 0 && pbpc < 0 && pcpa < 0)) {
    vec_pb = jv_sub(shadow_tri_b, shadow_tri_a);
    vec_pc = jv_sub(shadow_tri_c, shadow_tri_a);
    vec_pa = jv_sub(src_point, shadow_tri_a);
    float divider = jade_diffprod(vec_pb.x, vec_pc.y, vec_pb.y, vec_pc.x);
    float rate_a = jade_diffprod(vec_pc.y, vec_pa.x, vec_pc.x, vec_pa.y) / divider;
    /* -pb.y*pa.x + pb.x*pa.y */
    float rate_b = jade_fma(vec_pb.x, vec_pa.y, (-vec_pb.y) * vec_pa.x) / divider;

    vec_pb = jv_sub(p2, p1);
    vec_pc = jv_sub(p3, p1);
    vec_pa = jv_add(jv_add(p1, jv_scale(vec_pb, rate_a)), jv_scale(vec_pc, rate_b));

    float distance = jv_dot(jv_sub(vec_pa, src_point), normal_direction);
    if (distance > 0) {
      res.isHit = 1;
      res.hitPoint = vec_pa;
      res.distance = distance;
      res.index = index;
    }
  }
  return res;
}

/* PathTrace.cu:758-771 */
static float hit_aabb(Ray r, jvec3 AA, jvec3 BB) {
  jvec3 invdir = jv(1.0f / r.direction.x, 1.0f / r.direction.y, 1.0f / r.direction.z);
  jvec3 f = jv_mul(jv_sub(BB, r.startPoint), invdir);
  jvec3 n = jv_mul(jv_sub(AA, r.startPoint), invdir);
  jvec3 tmax = vmax3(f, n);
  jvec3 tmin = vmin3(f, n);
  float t1 = jade_fminf(tmax.x, jade_fminf(tmax.y, tmax.z));
  float t0 = jade_fmaxf(tmin.x, jade_fmaxf(tmin.y, tmin.z));
  return (t1 >= t0) ? ((t0 > 0.0f) ? t0 : t1) : -1.0f;
}

/* PathTrace.cu:776-792 */
static HitResult hit_array(const jade_scene* s, Ray ray, int l, int r, int src_object_idx, counters* c) {
  HitResult res;
  res.isHit = 0;
  res.index = 0;
  res.distance = INF_F;
  res.hitPoint = jv(0, 0, 0);
  for (int i = l; i <= r; i++) {
    if (i == src_object_idx) continue;
    c->tris_tested++;
    HitResult new_hit = hit_triangle(&s->tris[i], ray, i);
    if (new_hit.isHit && new_hit.distance < res.distance) res = new_hit;
  }
  return res;
}

/* Diagnostics only (not part of jade_rt.h): histogram of node pops + triangle
 * tests per hitBVH call in log2 buckets, and (jade_oracle_stack_histogram) the deepest
 * number of deferred far children per call, linear buckets: what sizes the HIP
 * kernel's LDS stack (tools/ray_histogram.py). */
static uint64_t g_visit_hist[2][32];
static uint64_t g_stack_hist[64];
void jade_oracle_stack_histogram(uint64_t* out, int reset) {
  memcpy(out, g_stack_hist, sizeof g_stack_hist);
  if (reset) memset(g_stack_hist, 0, sizeof g_stack_hist);
}
void jade_oracle_visit_histogram(uint64_t* out, int reset) {
  memcpy(out, g_visit_hist, sizeof g_visit_hist);
  if (reset) memset(g_visit_hist, 0, sizeof g_visit_hist);
}
static void hist_add(int which, uint64_t v) {
  int b = 0;
  while (v > 1 && b < 31) { v >>= 1; ++b; }
  __sync_fetch_and_add(&g_visit_hist[which][b], 1);
}

#ifdef JADE_ORACLE_PROBE
/* NOT in libjade_oracle.so: only `make -C oracle probe` (libjade_oracle_probe.so, used by tools/prune_probe.py alone) compiles
 * this in, so the checker the parity tests load holds nothing but the reference's walk. */
/* Diagnostics only (not part of jade_rt.h, and NOT the reference's algorithm): a walk that leaves out children whose box
 * begins farther along the ray than the nearest hit found so far - what SURVEY section 7 step 5 calls "optional distance
 * pruning".  tools/prune_probe.py uses it to count what such a walk visits and whether the frame keeps its bits; the
 * checker the parity tests use is hit_bvh below, untouched.  mode 1: a child is judged when its parent is visited;
 * mode 2: and again when it is popped. */
static int g_prune_mode = 0;
static float g_prune_rel = 0.0f, g_prune_abs = 0.0f, g_prune_minz = 0.0f;
static uint64_t g_prune_ctr[4]; /* internal nodes popped, leaves popped, triangles tested, calls */
void jade_oracle_set_prune(int mode, float rel, float abs_, float min_dz) {
  g_prune_mode = mode; g_prune_rel = rel; g_prune_abs = abs_; g_prune_minz = min_dz;
}
void jade_oracle_prune_counters(uint64_t* out, int reset) {
  memcpy(out, g_prune_ctr, sizeof g_prune_ctr);
  if (reset) memset(g_prune_ctr, 0, sizeof g_prune_ctr);
}
static HitResult hit_bvh_pruned(const jade_scene* s, Ray ray, int src_object_idx, counters* c) {
  HitResult res;
  res.isHit = 0; res.index = 0; res.distance = INF_F; res.hitPoint = jv(0, 0, 0);
  int stack[JADE_BVH_STACK_CAPACITY];
  float entry[JADE_BVH_STACK_CAPACITY];
  int sp = 0;
  stack[sp] = 1; entry[sp] = 0; sp++;
  const float len = sqrtf(jv_dot(ray.direction, ray.direction));
  jvec3 nd = jv_normalize(ray.direction);
  int may = fabsf(nd.z) >= g_prune_minz;
  uint64_t ni = 0, nl = 0, nt0 = c->tris_tested;
  counters dummy = *c;
  /* mode >= 3: what kind of ray this is, from the counter its call site has just advanced (a probe's shortcut) */
  static __thread uint64_t seen_shadow, seen_env;
  int kind = 0; /* 1 shadow, 2 env */
  if (c->rays_shadow != seen_shadow) { kind = 1; seen_shadow = c->rays_shadow; }
  else if (c->rays_env != seen_env) { kind = 2; seen_env = c->rays_env; }
  float d_e = -1.0f; /* shadow ray: the distance at which it meets the emitter it aims at */
  if (g_prune_mode >= 3 && kind == 1)
    for (int i = 0; i < s->d.n_emit; ++i) {
      HitResult h = hit_triangle(&s->tris[s->emit[i]], ray, s->emit[i]);
      if (h.isHit && (d_e < 0 || h.distance < d_e)) d_e = h.distance;
    }
  if (g_prune_mode == 3) may = 0; /* mode 3: early exits only, nothing judged by distance */
  while (sp > 0) {
    if (g_prune_mode >= 3 && res.isHit && (kind == 2 || (kind == 1 && d_e > 0 && res.distance < d_e))) break;
    --sp;
    int top = stack[sp];
    const jade_bvh_node* node = &s->nodes[top];
    const float limit = res.distance * (1.0f + g_prune_rel) + g_prune_abs;
    if (g_prune_mode >= 2 && may && entry[sp] * len > limit) continue;
    if (node->n > 0) {
      ++nl;
      HitResult r = hit_array(s, ray, node->index, node->index + node->n - 1, src_object_idx, c);
      if (r.isHit && r.distance < res.distance) res = r;
      continue;
    }
    ++ni;
    float d1 = -1, d2 = -1;
    if (node->left > 0) d1 = hit_aabb(ray, V3(s->nodes[node->left].aa), V3(s->nodes[node->left].bb));
    if (node->right > 0) d2 = hit_aabb(ray, V3(s->nodes[node->right].aa), V3(s->nodes[node->right].bb));
    /* the box's entry point along the ray: hit_aabb's value when the origin is outside the box (then it returned t0),
     * recomputed here to tell the two cases apart */
    float e1 = 0, e2 = 0;
    if (d1 > 0) { Ray q = ray; jvec3 inv = jv(1.0f / q.direction.x, 1.0f / q.direction.y, 1.0f / q.direction.z);
      jvec3 f = jv_mul(jv_sub(V3(s->nodes[node->left].bb), q.startPoint), inv), n = jv_mul(jv_sub(V3(s->nodes[node->left].aa), q.startPoint), inv);
      jvec3 tm = vmin3(f, n); float t0 = jade_fmaxf(tm.x, jade_fmaxf(tm.y, tm.z)); e1 = t0 > 0 ? t0 : 0; }
    if (d2 > 0) { Ray q = ray; jvec3 inv = jv(1.0f / q.direction.x, 1.0f / q.direction.y, 1.0f / q.direction.z);
      jvec3 f = jv_mul(jv_sub(V3(s->nodes[node->right].bb), q.startPoint), inv), n = jv_mul(jv_sub(V3(s->nodes[node->right].aa), q.startPoint), inv);
      jvec3 tm = vmin3(f, n); float t0 = jade_fmaxf(tm.x, jade_fmaxf(tm.y, tm.z)); e2 = t0 > 0 ? t0 : 0; }
    if (may && d1 > 0 && e1 * len > limit) d1 = -1;
    if (may && d2 > 0 && e2 * len > limit) d2 = -1;
    if (d1 > 0 && d2 > 0) {
      if (d1 < d2) { stack[sp] = node->right; entry[sp++] = e2; stack[sp] = node->left; entry[sp++] = e1; }
      else { stack[sp] = node->left; entry[sp++] = e1; stack[sp] = node->right; entry[sp++] = e2; }
    } else if (d1 > 0) { stack[sp] = node->left; entry[sp++] = e1; }
    else if (d2 > 0) { stack[sp] = node->right; entry[sp++] = e2; }
  }
  (void)dummy;
  __sync_fetch_and_add(&g_prune_ctr[0], ni);
  __sync_fetch_and_add(&g_prune_ctr[1], nl);
  __sync_fetch_and_add(&g_prune_ctr[2], c->tris_tested - nt0);
  __sync_fetch_and_add(&g_prune_ctr[3], 1);
  return res;
}


/* Round 4 probe (mode 5; diagnostics only, NOT the reference's algorithm): what do the early-exit queries cost by kind and by
 * outcome, and what would they cost with (a) a cache, per (source triangle, query), of where the last such query found its answer
 * - the triangle itself, or the subtree `up` levels above its leaf, walked first - and (b) the larger child first instead of the
 * nearer?  Everything a cached attempt tests is something the reference's walk tests too (boxes are nested, hit_aabb is monotone:
 * a leaf whose box the ray meets is reached by the reference), so an answer found there is exact; an attempt that finds none is
 * followed by the whole walk.  jade_oracle_set_prune(5, up, flags, ways): up < 0 = no cache, 0 = the triangle, k > 0 = k levels above the leaf;
 * flags bit 0 = larger child first, bit 1 = the cached subtree's leaves are not tested again by the whole walk (not modelled: 0). */
static uint64_t g_kind_ctr[4][2][5]; /* [kind: other, shadow, env, indirect][answered early: no, yes][calls, internal pops, leaves, tests, cache answers] */
void jade_oracle_kind_counters(uint64_t* out, int reset) {
  memcpy(out, g_kind_ctr, sizeof g_kind_ctr);
  if (reset) memset(g_kind_ctr, 0, sizeof g_kind_ctr);
}
#define OCC_KEYS 10 /* per source triangle: emitter 0, emitter 1+, environment query by octant */
#define OCC_WAYS 4
static int32_t* g_occ; /* [n_tris + 1][OCC_KEYS][OCC_WAYS] where the last such queries found their answer (triangle or node), -1 = none */
static int32_t* g_tri_leaf; /* triangle -> its leaf node */
static int32_t* g_parent;   /* node -> parent */
static const jade_scene* g_occ_scene;
static pthread_mutex_t g_occ_mu = PTHREAD_MUTEX_INITIALIZER;
void jade_oracle_occ_reset(void) {
  pthread_mutex_lock(&g_occ_mu);
  g_occ_scene = NULL;
  pthread_mutex_unlock(&g_occ_mu);
}
static void occ_prepare(const jade_scene* s) {
  if (__atomic_load_n(&g_occ_scene, __ATOMIC_ACQUIRE) == s) return;
  pthread_mutex_lock(&g_occ_mu);
  if (g_occ_scene != s) {
    free(g_occ); free(g_tri_leaf); free(g_parent);
    const int nt = s->d.n_triangles;
    const size_t n = (size_t)(nt + 1) * OCC_KEYS * OCC_WAYS;
    g_occ = (int32_t*)malloc(sizeof(int32_t) * n);
    for (size_t i = 0; i < n; ++i) g_occ[i] = -1;
    g_tri_leaf = (int32_t*)calloc((size_t)nt, sizeof(int32_t));
    g_parent = (int32_t*)calloc((size_t)s->d.n_nodes, sizeof(int32_t));
    for (int k = 1; k < s->d.n_nodes; ++k) {
      if (s->nodes[k].n > 0) { for (int i = 0; i < s->nodes[k].n; ++i) g_tri_leaf[s->nodes[k].index + i] = k; }
      else { if (s->nodes[k].left > 0) g_parent[s->nodes[k].left] = k; if (s->nodes[k].right > 0) g_parent[s->nodes[k].right] = k; }
    }
    __atomic_store_n(&g_occ_scene, s, __ATOMIC_RELEASE);
  }
  pthread_mutex_unlock(&g_occ_mu);
}
static float box_area(const jade_bvh_node* n) {
  const float dx = n->bb[0] - n->aa[0], dy = n->bb[1] - n->aa[1], dz = n->bb[2] - n->aa[2];
  return dx * dy + dy * dz + dz * dx;
}
/* the walk from `root` (its own box not tested), ended early by a recorded hit below `limit` (limit < 0: never) */
static int walk_from(const jade_scene* s, Ray ray, int src, counters* c, int root, float limit, int area_first, HitResult* res, uint64_t* ni, uint64_t* nl) {
  int stack[JADE_BVH_STACK_CAPACITY];
  int sp = 0;
  stack[sp++] = root;
  while (sp > 0) {
    --sp;
    const jade_bvh_node* node = &s->nodes[stack[sp]];
    if (node->n > 0) {
      ++*nl;
      HitResult r = hit_array(s, ray, node->index, node->index + node->n - 1, src, c);
      if (r.isHit && r.distance < res->distance) *res = r;
      if (limit >= 0 && res->isHit && res->distance < limit) return 1;
      continue;
    }
    ++*ni;
    float d1 = -1, d2 = -1;
    if (node->left > 0) d1 = hit_aabb(ray, V3(s->nodes[node->left].aa), V3(s->nodes[node->left].bb));
    if (node->right > 0) d2 = hit_aabb(ray, V3(s->nodes[node->right].aa), V3(s->nodes[node->right].bb));
    if (d1 > 0 && d2 > 0) {
      int left_first = d1 < d2;
      if (area_first) left_first = box_area(&s->nodes[node->left]) >= box_area(&s->nodes[node->right]);
      if (left_first) { stack[sp++] = node->right; stack[sp++] = node->left; }
      else { stack[sp++] = node->left; stack[sp++] = node->right; }
    } else if (d1 > 0) stack[sp++] = node->left;
    else if (d2 > 0) stack[sp++] = node->right;
  }
  return 0;
}
static HitResult hit_bvh_anyhit(const jade_scene* s, Ray ray, int src_object_idx, counters* c) {
  HitResult res;
  res.isHit = 0; res.index = 0; res.distance = INF_F; res.hitPoint = jv(0, 0, 0);
  static __thread uint64_t seen_shadow, seen_env, seen_ind;
  int kind = 0, stat_kind = 0;
  if (c->rays_shadow != seen_shadow) { kind = 1; seen_shadow = c->rays_shadow; }
  else if (c->rays_env != seen_env) { kind = 2; seen_env = c->rays_env; }
  else if (c->rays_indirect != seen_ind) { stat_kind = 3; seen_ind = c->rays_indirect; }
  if (kind) stat_kind = kind;
  float limit = -1.0f; /* < 0: the nearest hit is wanted */
  int key = 0;
  const int up = (int)g_prune_rel, flags = (int)g_prune_abs;
  if (kind == 1) {
    limit = INF_F;
    float d_e = -1.0f;
    for (int i = 0; i < s->d.n_emit; ++i) {
      HitResult h = hit_triangle(&s->tris[s->emit[i]], ray, s->emit[i]);
      if (h.isHit && (d_e < 0 || h.distance < d_e)) { d_e = h.distance; key = i < 1 ? 0 : 1; }
    }
    if (d_e > 0) limit = d_e;
  } else if (kind == 2) {
    limit = INF_F;
    const int oct = (ray.direction.x < 0) | ((ray.direction.y < 0) << 1) | ((ray.direction.z < 0) << 2);
    key = 2 + ((flags & 4) ? 0 : (flags & 8) ? (ray.direction.y < 0) : oct); /* flags bit 2: one key for all environment queries; bit 3: by the sign of d.y */
  }
  uint64_t ni = 0, nl = 0, nt0 = c->tris_tested, cache_answer = 0;
  int ways = (int)g_prune_minz; if (ways < 1) ways = 1; if (ways > OCC_WAYS) ways = OCC_WAYS;
  const int use_cache = up >= 0 && kind != 0;
  const int area_first = (flags & 1) && kind != 0;
  int32_t* slot = NULL;
  int early = 0;
  if (use_cache) {
    occ_prepare(s);
    slot = &g_occ[((size_t)(src_object_idx < 0 ? s->d.n_triangles : src_object_idx) * OCC_KEYS + key) * OCC_WAYS];
    for (int w = 0; w < ways && !early; ++w) {
      const int32_t t = slot[w];
      if (t < 0) continue;
      if (up == 0) {
        if (t == src_object_idx) continue;
        const jade_bvh_node* leaf = &s->nodes[g_tri_leaf[t]];
        if (hit_aabb(ray, V3(leaf->aa), V3(leaf->bb)) > 0) { /* nested boxes: the reference's walk reaches this leaf */
          c->tris_tested++;
          HitResult h = hit_triangle(&s->tris[t], ray, t);
          if (h.isHit && h.distance < limit) { res = h; early = 1; }
        }
      } else {
        /* (a cached leaf - a tree of one leaf - needs its own box tested; an internal node's children are tested by the walk) */
        if (s->nodes[t].n > 0 && !(hit_aabb(ray, V3(s->nodes[t].aa), V3(s->nodes[t].bb)) > 0)) continue;
        HitResult r2 = res;
        if (walk_from(s, ray, src_object_idx, c, t, limit, 0, &r2, &ni, &nl)) { res = r2; early = 1; }
      }
      if (early) { cache_answer = 1; if (w) { const int32_t x = slot[w]; for (int j = w; j > 0; --j) slot[j] = slot[j - 1]; slot[0] = x; } }
    }
  }
  if (!early) {
    res.isHit = 0; res.index = 0; res.distance = INF_F;
    early = walk_from(s, ray, src_object_idx, c, 1, limit, area_first, &res, &ni, &nl);
    if (use_cache && early) {
      int32_t v = res.index;
      if (up > 0) { v = g_tri_leaf[res.index]; for (int j = 0; j < up && g_parent[v] > 1; ++j) v = g_parent[v]; }
      for (int j = ways - 1; j > 0; --j) slot[j] = slot[j - 1];
      slot[0] = v;
    }
  }
  uint64_t* k = g_kind_ctr[stat_kind][early];
  __sync_fetch_and_add(&k[0], 1);
  __sync_fetch_and_add(&k[1], ni);
  __sync_fetch_and_add(&k[2], nl);
  __sync_fetch_and_add(&k[3], c->tris_tested - nt0);
  __sync_fetch_and_add(&k[4], cache_answer);
  return res;
}
#endif /* JADE_ORACLE_PROBE */

/* PathTrace.cu:795-859 */
static HitResult hit_bvh(const jade_scene* s, Ray ray, int src_object_idx, counters* c) {
#ifdef JADE_ORACLE_PROBE
  if (g_prune_mode >= 5) return hit_bvh_anyhit(s, ray, src_object_idx, c); /* (the probe build only, see above) */
  if (g_prune_mode) return hit_bvh_pruned(s, ray, src_object_idx, c);
#endif
  HitResult res;
  res.isHit = 0;
  res.index = 0;
  res.distance = INF_F;
  res.hitPoint = jv(0, 0, 0);

  int stack[JADE_BVH_STACK_CAPACITY];
  int sp = 0;
  stack[sp] = 1;
  sp++;
  c->nodes_visited++; /* the root record */
  uint64_t pops = 0, t_before = c->tris_tested;
  int deepest = 0;
  while (sp > 0) {
    if (sp - 1 > deepest) deepest = sp - 1; /* entries waiting while the top one is followed */
    --sp;
    ++pops;
    int top = stack[sp];
    const jade_bvh_node* node = &s->nodes[top];

    if (node->n > 0) {
      int L = node->index;
      int R = node->index + node->n - 1;
      HitResult r = hit_array(s, ray, L, R, src_object_idx, c);
      if (r.isHit && r.distance < res.distance) res = r;
      continue;
    }

    float d1 = -1;
    float d2 = -1;
    if (node->left > 0) {
      const jade_bvh_node* leftNode = &s->nodes[node->left];
      c->nodes_visited++;
      d1 = hit_aabb(ray, V3(leftNode->aa), V3(leftNode->bb));
    }
    if (node->right > 0) {
      const jade_bvh_node* rightNode = &s->nodes[node->right];
      c->nodes_visited++;
      d2 = hit_aabb(ray, V3(rightNode->aa), V3(rightNode->bb));
    }

    if (d1 > 0 && d2 > 0) {
      if (d1 < d2) {
        stack[sp++] = node->right;
        stack[sp++] = node->left;
      } else {
        stack[sp++] = node->left;
        stack[sp++] = node->right;
      }
    } else if (d1 > 0) {
      stack[sp++] = node->left;
    } else if (d2 > 0) {
      stack[sp++] = node->right;
    }
  }
  hist_add(0, pops);
  hist_add(1, c->tris_tested - t_before + 1);
  __sync_fetch_and_add(&g_stack_hist[deepest < 63 ? deepest : 63], 1);
  return res;
}

/* Software stand-in for tex2D(linear, mirror, normalized) on one float plane
 * pair (PathTrace.cu:652-665, 699): texel centres at +0.5, mirror addressing,
 * full-precision weights. */
static int mirror_index(int i, int n) {
  int m = i % (2 * n);
  if (m < 0) m += 2 * n;
  if (m >= n) m = 2 * n - 1 - m;
  return m;
}
static jvec3 env_fetch(const jade_scene* s, float u, float v) {
  int W = s->d.env_width, H = s->d.env_height;
  if (jade_isnan(u)) u = 0.0f;
  if (jade_isnan(v)) v = 0.0f;
  float x = u * (float)W - 0.5f;
  float y = v * (float)H - 0.5f;
  float xf = jade_floorf(x), yf = jade_floorf(y);
  float ax = x - xf, ay = y - yf;
  int i0 = mirror_index((int)xf, W), i1 = mirror_index((int)xf + 1, W);
  int j0 = mirror_index((int)yf, H), j1 = mirror_index((int)yf + 1, H);
  float w00 = (1.0f - ax) * (1.0f - ay), w10 = ax * (1.0f - ay);
  float w01 = (1.0f - ax) * ay, w11 = ax * ay;
  const float* t00 = s->env + 3 * ((size_t)j0 * W + i0);
  const float* t10 = s->env + 3 * ((size_t)j0 * W + i1);
  const float* t01 = s->env + 3 * ((size_t)j1 * W + i0);
  const float* t11 = s->env + 3 * ((size_t)j1 * W + i1);
  jvec3 c;
  c.x = ((w00 * t00[0] + w10 * t10[0]) + w01 * t01[0]) + w11 * t11[0];
  c.y = ((w00 * t00[1] + w10 * t10[1]) + w01 * t01[1]) + w11 * t11[1];
  c.z = ((w00 * t00[2] + w10 * t10[2]) + w01 * t01[2]) + w11 * t11[2];
  return c;
}

/* PathTrace.cu:686-702 */
static jvec3 sample_hdr(const jade_scene* s, jvec3 v) {
  jvec3 nv = jv_normalize(v);
  float ux = jade_atan2f(nv.z, nv.x);
  float uy = jade_asinf(nv.y);
  ux = (float)((double)ux / (2.0 * JADE_PI_D));
  uy = (float)((double)uy / JADE_PI_D);
  ux = (float)((double)ux + 0.5);
  uy = (float)((double)uy + 0.5);
  uy = (float)(1.0 - (double)uy);
  jvec3 color = env_fetch(s, ux, uy);
  color = vmin3(color, jv(10, 10, 10));
  return color;
}

/* PathTrace.cu:876-894 */
static jvec3 gen_refract_ray(jvec3 direction_in, jvec3 normal_line, float eta, int* full_reflex) {
  float cosi = jv_dot(direction_in, normal_line);
  if (cosi > 0) {
    normal_line = jv_neg(normal_line);
  } else {
    cosi *= -1;
  }
  float cost2 = 1.0f - eta * eta * (1.0f - cosi * cosi);
  if (cost2 > 0) {
    *full_reflex = 0;
    return jv_add(jv_scale(direction_in, eta), jv_scale(normal_line, eta * cosi - jade_sqrt(cost2)));
  } else {
    *full_reflex = 1;
    return direction_in;
  }
}

/* PathTrace.cu:897-903 */
static float tri_size(const jade_triangle* t) {
  jvec3 v_1 = jv_sub(V3(t->p2), V3(t->p1));
  jvec3 v_2 = jv_sub(V3(t->p3), V3(t->p1));
  jvec3 cp = jv_cross(v_1, v_2);
  return 0.5f * jade_sqrt(jv_dot(cp, cp));
}

/* one uniform-sphere direction, PathTrace.cu:968-971 (same text at 992, 1111,
 * 1136, 1304, 1328) */
static jvec3 sphere_dir(uint32_t* rng) {
  float cosine_theta = (float)(2.0 * ((double)jade_rand(rng) - 0.5));
  float sine_theta = jade_sqrt(1.0f - cosine_theta * cosine_theta);
  float fai_value = (float)(2.0 * JADE_PI_D * (double)jade_rand(rng));
  float sn, cs;
  jade_sincosf(fai_value, &sn, &cs);
  return jv(sine_theta * cs, sine_theta * sn, cosine_theta);
}

static jvec3 tri_point(const jade_triangle* t, float rx, float ry) {
  jvec3 p1 = V3(t->p1);
  return jv_add(jv_add(p1, jv_scale(jv_sub(V3(t->p2), p1), rx)), jv_scale(jv_sub(V3(t->p3), p1), ry));
}

static int nonemissive(const jade_triangle* t) {
  return t->emissive[0] < 1.5e-4f && t->emissive[1] < 1.5e-4f && t->emissive[2] < 1.5e-4f;
}

/* PathTrace.cu:905-1416 */
static jvec3 path_tracing(const jade_scene* s, HitResult hit, jvec3 direction, uint32_t* rng, counters* c) {
  const jade_triangle* T = s->tris;
  const int nEmit = s->d.n_emit;
  const float PI_F = (float)JADE_PI_D;
  const float RR_F = (float)JADE_RR_RATE_D;
  jvec3 l_dir = jv(0, 0, 0);
  int stack_offset = 0;
  jvec3 stack_dir[JADE_STACK_CAPACITY];
  jvec3 stack_indir_rate[JADE_STACK_CAPACITY];
  jvec3 out_direction = direction;
  jvec3 ray_src = hit.hitPoint;
  HitResult obj_hit = hit;
  jvec3 obj_hit_normal = V3(T[obj_hit.index].norm);
  while (stack_offset < JADE_STACK_CAPACITY) {
    c->shaded_hits++;
    const jade_triangle* ot = &T[obj_hit.index];
    jvec3 obj_emissive = V3(ot->emissive);
    if (obj_emissive.x > 1.4e-5f || obj_emissive.y > 1.4e-5f || obj_emissive.z > 1.4e-5f) {
      l_dir = obj_emissive;
      break;
    }
    l_dir = jv(0, 0, 0);
    jvec3 obj_hit_fr = jv_scale(V3(ot->brdf), (float)(1.0 / JADE_PI_D));
    int reflex_refract_select_rate = ot->refract_mode != JADE_NO_REFRACT ? 2 : 1;
    float select_reflex_refract = jade_rand(rng);
    if (select_reflex_refract < 0.5f && ot->refract_mode != JADE_NO_REFRACT) {
      if (ot->refract_mode == JADE_SUB_SURFACE) {
        select_reflex_refract = jade_rand(rng);
        if (select_reflex_refract < (float)JADE_SSS_RATE_D) {
          /* ---- SSS-diffuse, PathTrace.cu:931-1028 ---- */
          jvec3 obj_hit_fr_albedo = jv_scale(V3(ot->refract_albedo), (float)(1.0 / JADE_PI_D));
          for (int i = 0; i < nEmit; ++i) {
            float rand_x = jade_rand(rng);
            float rand_y = jade_rand(rng);
            if (rand_x + rand_y > 1) {
              rand_x = 1 - rand_x;
              rand_y = 1 - rand_y;
            }
            int emit_tri_idx = s->emit[i];
            const jade_triangle* t_i = &T[emit_tri_idx];
            jvec3 random_point = tri_point(t_i, rand_x, rand_y);
            jvec3 obj_light_direction = jv_sub(random_point, ray_src);
            if (jv_dot(obj_light_direction, obj_hit_normal) * jv_dot(out_direction, obj_hit_normal) < 0) continue;
            Ray new_ray;
            new_ray.startPoint = ray_src;
            new_ray.direction = obj_light_direction;
            c->rays_secondary++; c->rays_shadow++;
            HitResult hit_result = hit_bvh(s, new_ray, obj_hit.index, c);
            if (hit_result.isHit && hit_result.index == emit_tri_idx) {
              float dls = jv_dot(obj_light_direction, obj_light_direction);
              jvec3 w = jv_mul(V3(T[hit_result.index].emissive), obj_hit_fr_albedo);
              w = jv_scale(w, jade_fabs(jv_dot(obj_hit_normal, obj_light_direction) *
                                        jv_dot(V3(T[hit_result.index].norm), obj_light_direction)));
              w = jv_divs(jv_divs(w, dls), dls);
              w = jv_scale(w, tri_size(t_i));
              l_dir = jv_add(l_dir, w);
            }
          }
          {
            jvec3 ray_direction = sphere_dir(rng);
            if (jv_dot(ray_direction, obj_hit_normal) * jv_dot(out_direction, obj_hit_normal) < 0)
              ray_direction = jv_neg(ray_direction);
            Ray new_ray;
            new_ray.startPoint = ray_src;
            new_ray.direction = ray_direction;
            c->rays_secondary++; c->rays_env++;
            HitResult hit_result = hit_bvh(s, new_ray, obj_hit.index, c);
            if (!hit_result.isHit) {
              jvec3 skyColor = sample_hdr(s, ray_direction);
              jvec3 w = jv_mul(skyColor, obj_hit_fr_albedo);
              w = jv_scale(w, jade_fabs(jv_dot(obj_hit_normal, ray_direction)));
              w = jv_scale(jv_scale(w, 2.0f), PI_F);
              l_dir = jv_add(l_dir, w);
            }
          }
          l_dir = jv_scale(l_dir, (float)(reflex_refract_select_rate / JADE_SSS_RATE_D));

          float rr_result = jade_rand(rng);
          if (rr_result < RR_F) {
            jvec3 ray_direction = sphere_dir(rng);
            if (jv_dot(ray_direction, obj_hit_normal) * jv_dot(out_direction, obj_hit_normal) < 0)
              ray_direction = jv_neg(ray_direction);
            Ray new_ray;
            new_ray.startPoint = ray_src;
            new_ray.direction = ray_direction;
            c->rays_secondary++; c->rays_indirect++;
            HitResult new_hit = hit_bvh(s, new_ray, obj_hit.index, c);
            if (new_hit.isHit && nonemissive(&T[new_hit.index])) {
              ray_direction = jv_neg(ray_direction);
              jvec3 indir_rate = jv_divs(jv_scale(obj_hit_fr, jade_fabs(jv_dot(ray_direction, obj_hit_normal))), RR_F);
              ray_src = new_hit.hitPoint;
              out_direction = ray_direction;
              stack_dir[stack_offset] = l_dir;
              stack_indir_rate[stack_offset] =
                  jv_divs(jv_scale(indir_rate, (float)reflex_refract_select_rate), (float)JADE_SSS_RATE_D);
              ++stack_offset;
              obj_hit = new_hit;
              obj_hit_normal = V3(T[obj_hit.index].norm);
            } else {
              break;
            }
          } else {
            break;
          }
        } else {
          /* ---- BSSRDF, PathTrace.cu:1029-1178 ---- */
          const jade_obj_seg seg = s->segs[ot->obj_idx];
          float random_idx = jade_rand(rng) * s->prefix[seg.end_idx];
          int left = seg.begin_idx;
          int right = seg.end_idx;
          int middle = 0;
          while (left < right - 1) {
            middle = (left + right) / 2;
            if (random_idx <= s->prefix[middle]) {
              right = middle;
            } else if (random_idx >= s->prefix[middle]) {
              left = middle;
            } else {
              break; /* NaN area: the reference would spin forever */
            }
          }
          middle = s->mapping[middle];

          float rand_x = jade_rand(rng);
          float rand_y = jade_rand(rng);
          if (rand_x + rand_y > 1) {
            rand_x = 1 - rand_x;
            rand_y = 1 - rand_y;
          }
          const jade_triangle* t_i = &T[middle];
          jvec3 t_norm = V3(t_i->norm);
          jvec3 rate = V3(t_i->refract_rate);
          jvec3 random_point = tri_point(t_i, rand_x, rand_y);
          jvec3 inner_direction = jv_sub(random_point, ray_src);
          float inner_distance = jade_sqrt(jv_dot(inner_direction, inner_direction));
          float neg_d = -1.0f * inner_distance;
          float neg_d3 = (float)((double)neg_d / 3.0);
          float E_F = (float)JADE_E_D;
          jvec3 e1 = jv(jade_powf(E_F, neg_d / rate.x), jade_powf(E_F, neg_d / rate.y), jade_powf(E_F, neg_d / rate.z));
          jvec3 e2 = jv(jade_powf(E_F, neg_d3 / rate.x), jade_powf(E_F, neg_d3 / rate.y), jade_powf(E_F, neg_d3 / rate.z));
          jvec3 bssrdf = jv_div(jv_add(e1, e2), jv_scale(rate, (float)(8 * JADE_PI_D * (double)inner_distance)));

          float eta = t_i->refract_index;
          float R0 = (eta - 1) / (eta + 1) * (eta - 1) / (eta + 1);
          float one_cosine_i = 1 - jade_fabs(jv_dot(obj_hit_normal, out_direction));
          float one_cosine_i_sqr = one_cosine_i * one_cosine_i;
          float fresnel_rate_i = R0 + (1 - R0) * one_cosine_i_sqr * one_cosine_i_sqr * one_cosine_i;
          bssrdf = jv_scale(bssrdf, fresnel_rate_i);
          float area_total = s->prefix[s->segs[t_i->obj_idx].end_idx];

          for (int i = 0; i < nEmit; ++i) {
            float rx = jade_rand(rng);
            float ry = jade_rand(rng);
            if (rx + ry > 1) {
              rx = 1 - rx;
              ry = 1 - ry;
            }
            int emit_tri_idx = s->emit[i];
            const jade_triangle* emit_i = &T[emit_tri_idx];
            jvec3 random_emit_point = tri_point(emit_i, rx, ry);
            jvec3 obj_light_direction = jv_sub(random_emit_point, random_point);
            Ray new_ray;
            new_ray.startPoint = random_point;
            new_ray.direction = obj_light_direction;
            c->rays_secondary++; c->rays_shadow++;
            HitResult hit_result = hit_bvh(s, new_ray, middle, c);
            if (hit_result.isHit && hit_result.index == emit_tri_idx) {
              float one_cosine_o = 1 - jade_fabs(jv_dot(jv_normalize(obj_light_direction), t_norm));
              float one_cosine_o_sqr = one_cosine_o * one_cosine_o;
              float fresnel_rate_o = R0 - (1 - R0) * one_cosine_o_sqr * one_cosine_o_sqr * one_cosine_o;
              float dls = jv_dot(obj_light_direction, obj_light_direction);
              jvec3 w = jv_scale(V3(T[hit_result.index].emissive), fresnel_rate_o);
              w = jv_mul(w, bssrdf);
              w = jv_scale(w, jade_fabs(jv_dot(t_norm, obj_light_direction) *
                                        jv_dot(V3(T[hit_result.index].norm), obj_light_direction)));
              w = jv_divs(jv_divs(w, dls), dls);
              w = jv_scale(w, tri_size(emit_i));
              w = jv_divs(w, PI_F);
              w = jv_scale(w, area_total);
              l_dir = jv_add(l_dir, w);
            }
          }

          jvec3 ray_direction = sphere_dir(rng);
          if (jv_dot(ray_direction, t_norm) * jv_dot(inner_direction, t_norm) < 0) ray_direction = jv_neg(ray_direction);
          Ray new_ray;
          new_ray.startPoint = random_point;
          new_ray.direction = ray_direction;
          c->rays_secondary++; c->rays_env++;
          HitResult hit_result = hit_bvh(s, new_ray, middle, c);
          if (!hit_result.isHit) {
            float one_cosine_o = 1 - jade_fabs(jv_dot(ray_direction, t_norm));
            float one_cosine_o_sqr = one_cosine_o * one_cosine_o;
            float fresnel_rate_o = R0 - (1 - R0) * one_cosine_o_sqr * one_cosine_o_sqr * one_cosine_o;
            jvec3 skyColor = sample_hdr(s, ray_direction);
            jvec3 w = jv_mul(jv_scale(skyColor, fresnel_rate_o), bssrdf);
            w = jv_scale(jv_scale(w, jade_fabs(jv_dot(t_norm, ray_direction))), 2.0f);
            l_dir = jv_add(l_dir, w);
          }

          l_dir = jv_scale(l_dir, (float)(reflex_refract_select_rate / (1 - JADE_SSS_RATE_D)));

          ray_direction = sphere_dir(rng);
          if (jv_dot(ray_direction, t_norm) * jv_dot(inner_direction, t_norm) > 0) ray_direction = jv_neg(ray_direction);
          float rr_result = jade_rand(rng);
          if (rr_result < RR_F) {
            new_ray.startPoint = random_point;
            new_ray.direction = ray_direction;
            c->rays_secondary++; c->rays_indirect++;
            HitResult new_hit = hit_bvh(s, new_ray, middle, c);
            if (new_hit.isHit && nonemissive(&T[new_hit.index])) {
              ray_direction = jv_neg(ray_direction);
              float one_cosine_o = 1 - jade_fabs(jv_dot(ray_direction, t_norm));
              float one_cosine_o_sqr = one_cosine_o * one_cosine_o;
              float fresnel_rate_o = R0 - (1 - R0) * one_cosine_o_sqr * one_cosine_o_sqr * one_cosine_o;
              jvec3 indir_rate = jv_scale(bssrdf, fresnel_rate_o);
              indir_rate = jv_scale(indir_rate, jade_fabs(jv_dot(ray_direction, t_norm)));
              indir_rate = jv_scale(indir_rate, area_total);
              indir_rate = jv_divs(jv_scale(indir_rate, 2.0f), RR_F);
              ray_src = new_hit.hitPoint;
              out_direction = ray_direction;
              stack_dir[stack_offset] = l_dir;
              stack_indir_rate[stack_offset] =
                  jv_divs(jv_scale(indir_rate, (float)reflex_refract_select_rate), (float)(1 - JADE_SSS_RATE_D));
              ++stack_offset;
              obj_hit = new_hit;
              obj_hit_normal = V3(T[obj_hit.index].norm);
            } else {
              break;
            }
          } else {
            break;
          }
        }
      } else {
        /* ---- direct refraction, PathTrace.cu:1180-1262 ---- */
        float triangle_miu = ot->refract_index;
        float R0 = (1 - triangle_miu) / (1 + triangle_miu) * (1 - triangle_miu) / (1 + triangle_miu);
        float one_cosine_i = 1 - jade_fabs(jv_dot(obj_hit_normal, out_direction));
        float one_cosine_i_sqr = one_cosine_i * one_cosine_i;
        float fresnel_rate_i = R0 + (1 - R0) * one_cosine_i_sqr * one_cosine_i_sqr * one_cosine_i;

        int full_reflex = 0;
        jvec3 rev_out_direction = jv_scale(out_direction, -1.0f);
        jvec3 refract_ray = gen_refract_ray(rev_out_direction, obj_hit_normal, (float)(1.0 / (double)triangle_miu), &full_reflex);
        jvec3 l_indir_rate = jv(1 - fresnel_rate_i, 1 - fresnel_rate_i, 1 - fresnel_rate_i);

        Ray new_ray;
        new_ray.startPoint = ray_src;
        new_ray.direction = refract_ray;
        HitResult new_hit = obj_hit;
        for (int i = 0; i < JADE_MAX_FULL_REFLEX_TIME; ++i) {
          c->rays_secondary++; c->rays_refract++;
          new_hit = hit_bvh(s, new_ray, new_hit.index, c);
          if (new_hit.isHit) {
            const jade_triangle* ht = &T[new_hit.index];
            jvec3 hn = V3(ht->norm);
            refract_ray = gen_refract_ray(refract_ray, hn, triangle_miu, &full_reflex);
            jvec3 distance = jv_sub(new_ray.startPoint, new_hit.hitPoint);
            float dist = jade_sqrt(jv_dot(distance, distance));
            l_indir_rate = jv_mul(l_indir_rate, jv(jade_powf(ht->refract_rate[0], dist), jade_powf(ht->refract_rate[1], dist),
                                                     jade_powf(ht->refract_rate[2], dist)));
            new_ray.startPoint = new_hit.hitPoint;

            float one_cosine_o = 1 - jade_fabs(jv_dot(refract_ray, hn));
            float one_cosine_o_sqr = one_cosine_o * one_cosine_o;
            float fresnel_rate_o = R0 - (1 - R0) * one_cosine_o_sqr * one_cosine_o_sqr * one_cosine_o;

            float reflex_refract_select = jade_rand(rng);
            if (full_reflex || reflex_refract_select < 0.2f) {
              refract_ray = jv_sub(refract_ray, jv_scale(hn, 2 * jv_dot(refract_ray, hn)));
              new_ray.direction = refract_ray;
              if (!full_reflex) l_indir_rate = jv_scale(l_indir_rate, fresnel_rate_o * 5);
            } else {
              l_indir_rate = jv_scale(l_indir_rate, (float)((1.0 - (double)fresnel_rate_o) * 1.25));
              break;
            }
          } else {
            return jv(0, 0, 0); /* PathTrace.cu:1231 */
          }
        }

        float rr_result = jade_rand(rng);
        if (rr_result < RR_F) {
          new_ray.direction = refract_ray;
          c->rays_secondary++; c->rays_refract++;
          new_hit = hit_bvh(s, new_ray, new_hit.index, c);
          if (new_hit.isHit) {
            out_direction = jv_scale(refract_ray, -1.0f);
            ray_src = new_hit.hitPoint;
            obj_hit = new_hit;
            obj_hit_normal = V3(T[obj_hit.index].norm);
            stack_dir[stack_offset] = jv(0, 0, 0);
            stack_indir_rate[stack_offset] = jv_scale(l_indir_rate, (float)(reflex_refract_select_rate / JADE_RR_RATE_D));
            ++stack_offset;
          } else {
            l_dir = jv_scale(jv_mul(sample_hdr(s, refract_ray), l_indir_rate),
                             (float)(reflex_refract_select_rate / JADE_RR_RATE_D));
            break;
          }
        } else {
          break;
        }
      }
    } else {
      if (ot->reflex_mode == JADE_DIFFUSE) {
        /* ---- diffuse, PathTrace.cu:1266-1364 ---- */
        for (int i = 0; i < nEmit; ++i) {
          float rand_x = jade_rand(rng);
          float rand_y = jade_rand(rng);
          if (rand_x + rand_y > 1) {
            rand_x = 1 - rand_x;
            rand_y = 1 - rand_y;
          }
          int emit_tri_idx = s->emit[i];
          const jade_triangle* t_i = &T[emit_tri_idx];
          jvec3 random_point = tri_point(t_i, rand_x, rand_y);
          jvec3 obj_light_direction = jv_sub(random_point, ray_src);
          if (jv_dot(obj_light_direction, obj_hit_normal) * jv_dot(out_direction, obj_hit_normal) < 0) continue;
          Ray new_ray;
          new_ray.startPoint = ray_src;
          new_ray.direction = obj_light_direction;
          c->rays_secondary++; c->rays_shadow++;
          HitResult hit_result = hit_bvh(s, new_ray, obj_hit.index, c);
          if (hit_result.isHit && hit_result.index == emit_tri_idx) {
            float dls = jv_dot(obj_light_direction, obj_light_direction);
            jvec3 w = jv_mul(V3(T[hit_result.index].emissive), obj_hit_fr);
            w = jv_scale(w, jade_fabs(jv_dot(obj_hit_normal, obj_light_direction) *
                                      jv_dot(V3(T[hit_result.index].norm), obj_light_direction)));
            w = jv_divs(jv_divs(w, dls), dls);
            w = jv_scale(w, tri_size(t_i));
            l_dir = jv_add(l_dir, w);
          }
        }
        {
          jvec3 ray_direction = sphere_dir(rng);
          if (jv_dot(ray_direction, obj_hit_normal) * jv_dot(out_direction, obj_hit_normal) < 0)
            ray_direction = jv_neg(ray_direction);
          Ray new_ray;
          new_ray.startPoint = ray_src;
          new_ray.direction = ray_direction;
          c->rays_secondary++; c->rays_env++;
          HitResult hit_result = hit_bvh(s, new_ray, obj_hit.index, c);
          if (!hit_result.isHit) {
            jvec3 skyColor = sample_hdr(s, ray_direction);
            jvec3 w = jv_mul(skyColor, obj_hit_fr);
            w = jv_scale(w, jade_fabs(jv_dot(obj_hit_normal, ray_direction)));
            w = jv_scale(jv_scale(w, 2.0f), PI_F);
            l_dir = jv_add(l_dir, w);
          }
        }
        l_dir = jv_scale(l_dir, (float)reflex_refract_select_rate);

        float rr_result = jade_rand(rng);
        if (rr_result < RR_F) {
          jvec3 ray_direction = sphere_dir(rng);
          if (jv_dot(ray_direction, obj_hit_normal) * jv_dot(out_direction, obj_hit_normal) < 0)
            ray_direction = jv_neg(ray_direction);
          Ray new_ray;
          new_ray.startPoint = ray_src;
          new_ray.direction = ray_direction;
          c->rays_secondary++; c->rays_indirect++;
          HitResult new_hit = hit_bvh(s, new_ray, obj_hit.index, c);
          if (new_hit.isHit && nonemissive(&T[new_hit.index])) {
            ray_direction = jv_neg(ray_direction);
            jvec3 indir_rate = jv_divs(jv_scale(obj_hit_fr, jade_fabs(jv_dot(ray_direction, obj_hit_normal))), RR_F);
            ray_src = new_hit.hitPoint;
            out_direction = ray_direction;
            stack_dir[stack_offset] = l_dir;
            stack_indir_rate[stack_offset] = jv_scale(indir_rate, (float)reflex_refract_select_rate);
            ++stack_offset;
            obj_hit = new_hit;
            obj_hit_normal = V3(T[obj_hit.index].norm);
          } else {
            break;
          }
        } else {
          break;
        }
      } else {
        /* ---- mirror, PathTrace.cu:1365-1405 ---- */
        if (obj_emissive.x > 1.5e-4f || obj_emissive.y > 1.5e-4f || obj_emissive.x > 1.5e-4f) {
          l_dir = jv_scale(jv_mul(obj_emissive, obj_hit_fr), (float)reflex_refract_select_rate);
          break;
        } else {
          float rr_result = jade_rand(rng);
          if (rr_result < RR_F) {
            out_direction = jv_sub(jv_scale(obj_hit_normal, 2 * jv_dot(out_direction, V3(ot->norm))), out_direction);
            Ray new_ray;
            new_ray.startPoint = ray_src;
            new_ray.direction = out_direction;
            c->rays_secondary++; c->rays_mirror++;
            HitResult new_hit = hit_bvh(s, new_ray, obj_hit.index, c);
            float k = (float)(reflex_refract_select_rate / (JADE_RR_RATE_D / JADE_PI_D));
            if (new_hit.isHit) {
              out_direction = jv_neg(out_direction);
              ray_src = new_hit.hitPoint;
              obj_hit = new_hit;
              obj_hit_normal = V3(T[obj_hit.index].norm);
              stack_dir[stack_offset] = jv(0, 0, 0);
              stack_indir_rate[stack_offset] = jv_scale(obj_hit_fr, k);
              ++stack_offset;
            } else {
              l_dir = jv_scale(jv_mul(sample_hdr(s, out_direction), obj_hit_fr), k);
              break;
            }
          } else {
            break;
          }
        }
      }
    }
  }

  for (int i = stack_offset - 1; i >= 0; --i) {
    l_dir = jv_mul(l_dir, stack_indir_rate[i]);
    l_dir = jv_add(l_dir, stack_dir[i]);
  }
  return l_dir;
}

/* One sample of one pixel: the body of the spp loop, PathTrace.cu:1429-1455. */
static jvec3 render_sample(const jade_scene* s, const jade_render_params* rp, int px, int py, uint32_t* rng, counters* c) {
  Ray ray;
  ray.startPoint = jv(rp->eye[0], rp->eye[1], rp->eye[2]);
  float fx = (float)px + jade_rand(rng);
  double lo = -1.0 + 2.0 / (double)rp->width * ((double)fx - 0.5);
  float left_offset = (float)(lo * ((double)rp->width / (double)rp->height));
  float fy = (float)py + jade_rand(rng);
  float up_offset = (float)(-1.0 + 2.0 / (double)rp->height * ((double)fy - 0.5));

  jvec3 dir = jv(left_offset, up_offset, -1.5f);
  dir = jade_transform(dir, 0.0f, rp->camera);
  ray.direction = jv_normalize(dir);

  c->rays_primary++;
  c->samples++;
  HitResult firstHit = hit_bvh(s, ray, -1, c);
  jvec3 color;
  if (!firstHit.isHit) {
    color = sample_hdr(s, ray.direction);
  } else {
    jvec3 Le = V3(s->tris[firstHit.index].emissive);
    jvec3 Li = path_tracing(s, firstHit, jv_neg(ray.direction), rng, c);
    color = jv_add(Le, Li);
  }
  return color;
}

/* PathTrace.cu:680-682 (ACES) or :669-672 == pass3.fsh:8-18 (toneMapping), then :1461-1473 */
static void tonemap_pack(jvec3 c, int mode, float limit, uint8_t* bgr) {
  float v[3] = {c.x, c.y, c.z};
  float rein = 1.0f;
  if (mode == JADE_TONEMAP_REINHARD) {
    float luminance = (float)(0.3 * (double)c.x + 0.6 * (double)c.y + 0.1 * (double)c.z);
    rein = (float)(1.0 / (1.0 + (double)(luminance / limit)));
  }
  for (int k = 0; k < 3; ++k) {
    float x = v[k];
    if (mode == JADE_TONEMAP_REINHARD) {
      x = x * rein;
    } else {
      float num = x * (x * 2.51f + 0.03f);
      float den = x * (x * 2.43f + 0.59f) + 0.14f;
      x = num / den;
    }
    x = jade_powf(x, (float)(1.0 / 2.2));
    x = x * 255.0f;
    x = x > 255 ? 255 : x;
    v[k] = x;
  }
  for (int k = 0; k < 3; ++k) {
    float x = v[2 - k]; /* B, G, R */
    bgr[k] = (x >= 0.0f) ? (uint8_t)x : 0; /* NaN / negative: defined as 0 */
  }
}

/* ------------------------------------------------------------------ ABI */

int jade_abi_version(void) { return JADE_ABI_VERSION; }
const char* jade_backend_name(void) { return "oracle-cpu"; }
const char* jade_last_error(void) { return g_err; }
int jade_device_count(int* n) {
  if (!n) return fail(JADE_ERR_INVALID, "null argument");
  *n = 0;
  return JADE_OK;
}

static void* dup_mem(const void* p, size_t n) {
  void* q = malloc(n ? n : 1);
  if (q && n) memcpy(q, p, n);
  return q;
}

static int bvh_depth_ok(const jade_scene_desc* d) {
  /* iterative walk; also rejects cycles via a visit budget */
  int32_t n = d->n_nodes;
  int* stk = (int*)malloc(sizeof(int) * 2 * (size_t)(n + 2));
  int* dep = (int*)malloc(sizeof(int) * 2 * (size_t)(n + 2));
  if (!stk || !dep) { free(stk); free(dep); return 0; }
  int sp = 0, ok = 1;
  int64_t budget = 4 * (int64_t)n + 8;
  stk[sp] = 1; dep[sp] = 1; sp++;
  while (sp > 0 && ok) {
    --sp;
    int id = stk[sp], dp = dep[sp];
    if (--budget < 0 || dp > JADE_BVH_STACK_CAPACITY - 1) { ok = 0; break; }
    const jade_bvh_node* nd = &d->nodes[id];
    if (nd->n > 0) {
      if (nd->index < 0 || (int64_t)nd->index + nd->n > d->n_triangles) ok = 0;
      continue;
    }
    if (nd->left < 0 || nd->left >= n || nd->right < 0 || nd->right >= n) { ok = 0; break; }
    if (sp + 2 > 2 * (n + 2)) { ok = 0; break; }
    if (nd->left > 0) { stk[sp] = nd->left; dep[sp] = dp + 1; sp++; }
    if (nd->right > 0) { stk[sp] = nd->right; dep[sp] = dp + 1; sp++; }
  }
  free(stk);
  free(dep);
  return ok;
}

int jade_scene_create(const jade_scene_desc* d, int device_id, jade_scene** out) {
  (void)device_id;
  if (!d || !out) return fail(JADE_ERR_INVALID, "null argument");
  if (d->abi_version != JADE_ABI_VERSION) return fail(JADE_ERR_INVALID, "abi_version mismatch");
  if (d->n_triangles <= 0 || d->n_nodes < 2 || !d->triangles || !d->nodes)
    return fail(JADE_ERR_INVALID, "scene needs triangles and a BVH (dummy node 0 + root 1)");
  if (d->n_emit < 0 || (d->n_emit > 0 && !d->emit_indices)) return fail(JADE_ERR_INVALID, "bad emitter list");
  if (!d->index_mapping || !d->prefix_area || d->n_objects <= 0 || !d->obj_segs)
    return fail(JADE_ERR_INVALID, "missing mapping / prefix areas / object segments");
  if (d->env_width <= 0 || d->env_height <= 0 || !d->env_rgb) return fail(JADE_ERR_INVALID, "missing environment map");
  for (int i = 0; i < d->n_emit; ++i)
    if (d->emit_indices[i] < 0 || d->emit_indices[i] >= d->n_triangles) return fail(JADE_ERR_INVALID, "emitter index out of range");
  for (int i = 0; i < d->n_triangles; ++i) {
    if (d->index_mapping[i] < 0 || d->index_mapping[i] >= d->n_triangles) return fail(JADE_ERR_INVALID, "index_mapping out of range");
    if (d->triangles[i].obj_idx < 0 || d->triangles[i].obj_idx >= d->n_objects) return fail(JADE_ERR_INVALID, "obj_idx out of range");
  }
  for (int i = 0; i < d->n_objects; ++i)
    if (d->obj_segs[i].begin_idx < 0 || d->obj_segs[i].end_idx >= d->n_triangles || d->obj_segs[i].begin_idx > d->obj_segs[i].end_idx)
      return fail(JADE_ERR_INVALID, "object segment out of range");
  if (!bvh_depth_ok(d)) return fail(JADE_ERR_UNSUPPORTED, "BVH malformed or deeper than the traversal stack");

  jade_scene* s = (jade_scene*)calloc(1, sizeof *s);
  if (!s) return fail(JADE_ERR_NOMEM, "out of memory");
  s->d = *d;
  s->tris = (jade_triangle*)dup_mem(d->triangles, sizeof(jade_triangle) * (size_t)d->n_triangles);
  s->nodes = (jade_bvh_node*)dup_mem(d->nodes, sizeof(jade_bvh_node) * (size_t)d->n_nodes);
  s->emit = (int32_t*)dup_mem(d->emit_indices, sizeof(int32_t) * (size_t)d->n_emit);
  s->mapping = (int32_t*)dup_mem(d->index_mapping, sizeof(int32_t) * (size_t)d->n_triangles);
  s->prefix = (float*)dup_mem(d->prefix_area, sizeof(float) * (size_t)d->n_triangles);
  s->segs = (jade_obj_seg*)dup_mem(d->obj_segs, sizeof(jade_obj_seg) * (size_t)d->n_objects);
  s->env = (float*)dup_mem(d->env_rgb, sizeof(float) * 3 * (size_t)d->env_width * d->env_height);
  if (!s->tris || !s->nodes || !s->emit || !s->mapping || !s->prefix || !s->segs || !s->env) {
    jade_scene_destroy(s);
    return fail(JADE_ERR_NOMEM, "out of memory");
  }
  *out = s;
  return JADE_OK;
}

void jade_scene_destroy(jade_scene* s) {
  if (!s) return;
  free(s->tris); free(s->nodes); free(s->emit); free(s->mapping);
  free(s->prefix); free(s->segs); free(s->env); free(s->sum); free(s->tile_keep); free(s->sum_slot);
  free(s);
}

static int owns_pixel_s(const jade_scene* s, int x, int y) {
  const jade_render_params* rp = &s->rp;
  if ((x / JADE_TILE_SIZE + y / JADE_TILE_SIZE) % rp->tile_nranks != rp->tile_rank) return 0;
  if (s->tile_keep) {
    const int tiles_x = (rp->width + JADE_TILE_SIZE - 1) / JADE_TILE_SIZE;
    const int tid = (y / JADE_TILE_SIZE) * tiles_x + x / JADE_TILE_SIZE;
    return tid < s->tile_keep_n && s->tile_keep[tid];
  }
  return 1;
}

/* Checker-only extension (NOT part of jade_rt.h; the HIP module has no such entry point): restrict the renders that
 * follow to the listed 16x16 tiles (id = ty * tiles_x + tx, tiles_x = ceil(width / 16)) of whatever partition the
 * params name.  bench.py's parity_check and the full-size parity tests use it to let the oracle render a few tiles of
 * a 1920x1080 / 3840x2160 frame at the benchmark's full sample count in seconds (samples are independent work items,
 * PathTrace.cu:1418-1474: a pixel's value does not depend on which other pixels are rendered).  n = 0 clears it. */
int jade_oracle_set_tile_filter(jade_scene* s, const int32_t* tile_ids, int32_t n) {
  if (!s || n < 0 || (n > 0 && !tile_ids)) return fail(JADE_ERR_INVALID, "bad tile filter");
  free(s->tile_keep);
  s->tile_keep = NULL;
  s->tile_keep_n = 0;
  /* the sums of a render in progress were laid out for the old filter (sum_slot): that render is over - the next step needs a
   * new jade_render_begin (a filter changed between begin and step indexed the compact sums with -1) */
  s->have_rp = 0;
  free(s->sum);
  s->sum = NULL;
  free(s->sum_slot);
  s->sum_slot = NULL;
  if (n == 0) return JADE_OK;
  int32_t mx = -1;
  for (int i = 0; i < n; ++i) {
    if (tile_ids[i] < 0) return fail(JADE_ERR_INVALID, "negative tile id");
    if (tile_ids[i] > mx) mx = tile_ids[i];
  }
  s->tile_keep = (uint8_t*)calloc((size_t)mx + 1, 1);
  if (!s->tile_keep) return fail(JADE_ERR_NOMEM, "out of memory");
  s->tile_keep_n = mx + 1;
  for (int i = 0; i < n; ++i) s->tile_keep[tile_ids[i]] = 1;
  return JADE_OK;
}

int jade_owned_tile_count(int32_t width, int32_t height, int32_t rank, int32_t nranks) {
  if (width <= 0 || height <= 0 || nranks <= 0 || rank < 0 || rank >= nranks) return -1;
  int tx = (width + JADE_TILE_SIZE - 1) / JADE_TILE_SIZE, ty = (height + JADE_TILE_SIZE - 1) / JADE_TILE_SIZE;
  int count = 0;
  for (int y = 0; y < ty; ++y)
    for (int x = 0; x < tx; ++x) count += (x + y) % nranks == rank;
  return count;
}

int jade_render_begin(jade_scene* s, const jade_render_params* rp) {
  if (!s || !rp) return fail(JADE_ERR_INVALID, "null argument");
  if (rp->width <= 0 || rp->height <= 0 || rp->tile_nranks <= 0 || rp->tile_rank < 0 || rp->tile_rank >= rp->tile_nranks)
    return fail(JADE_ERR_INVALID, "bad image size or tile partition");
  if (rp->env_sampling != JADE_ENV_REFERENCE) /* the oracle states the reference's estimator and nothing else (jade_rt.h, JADE_ENV_IMPORTANCE) */
    return fail(JADE_ERR_UNSUPPORTED, "env_sampling: the oracle renders the reference's uniform hemisphere sampling only");
  size_t np = (size_t)rp->width * rp->height;
  free(s->sum);
  s->sum = NULL;
  free(s->sum_slot);
  s->sum_slot = NULL;
  s->rp = *rp;
  s->sum_pixels = np;
  if (s->tile_keep || rp->tile_nranks > 1) { /* sums for the rendered pixels only: a few tiles of a 4K frame must not reserve 100 GB of address space */
    s->sum_slot = (int32_t*)malloc(np * sizeof(int32_t));
    if (!s->sum_slot) return fail(JADE_ERR_NOMEM, "out of memory");
    size_t k = 0;
    for (int y = 0; y < rp->height; ++y)
      for (int x = 0; x < rp->width; ++x) s->sum_slot[(size_t)y * rp->width + x] = owns_pixel_s(s, x, y) ? (int32_t)k++ : -1;
    s->sum_pixels = k ? k : 1;
  }
  s->sum = (float*)calloc(s->sum_pixels * 3 * JADE_SAMPLE_LANES, sizeof(float));
  if (!s->sum) return fail(JADE_ERR_NOMEM, "out of memory");
  s->have_rp = 1;
  s->spp_done = 0;
  return JADE_OK;
}

typedef struct {
  jade_scene* s;
  int spp;
  int64_t first_sample;
  volatile int* next_row;
  counters c;
} worker_arg;

static void* worker(void* p) {
  worker_arg* shared = (worker_arg*)p;
  worker_arg local = *shared; /* counters on this thread's stack: no false sharing between workers */
  worker_arg* a = &local;
  jade_scene* s = a->s;
  const jade_render_params* rp = &s->rp;
  for (;;) {
    int y = __sync_fetch_and_add(a->next_row, 1);
    if (y >= rp->height) break;
    for (int x = 0; x < rp->width; ++x) {
      if (!owns_pixel_s(s, x, y)) continue;
      const size_t pi = s->sum_slot ? (size_t)s->sum_slot[(size_t)y * rp->width + x] : (size_t)y * rp->width + x;
      const size_t np = s->sum_pixels;
      for (int i = 0; i < a->spp; ++i) {
        int64_t sidx = a->first_sample + i;
        uint32_t rng = jade_rng_seed((uint32_t)x, (uint32_t)y, rp->frame + (uint32_t)sidx);
        jvec3 color = render_sample(s, rp, x, y, &rng, &a->c);
        float* acc = s->sum + 3 * ((size_t)(sidx % JADE_SAMPLE_LANES) * np + pi);
        /* final_result = final_result + color (PathTrace.cu:1454), per lane */
        acc[0] = acc[0] + color.x; acc[1] = acc[1] + color.y; acc[2] = acc[2] + color.z;
      }
    }
  }
  shared->c = local.c;
  return NULL;
}

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static void add_counters(jade_stats* st, const counters* c) {
  st->rays_primary += c->rays_primary;
  st->rays_secondary += c->rays_secondary;
  st->nodes_visited += c->nodes_visited;
  st->tris_tested += c->tris_tested;
  st->shaded_hits += c->shaded_hits;
  st->samples += c->samples;
  st->rays_shadow += c->rays_shadow;
  st->rays_env += c->rays_env;
  st->rays_indirect += c->rays_indirect;
  st->rays_mirror += c->rays_mirror;
  st->rays_refract += c->rays_refract;
}

int jade_render_step(jade_scene* s, int32_t spp, jade_stats* st) {
  if (!s || !s->have_rp) return fail(JADE_ERR_INVALID, "jade_render_begin not called");
  if (spp < 0) return fail(JADE_ERR_INVALID, "negative spp");
  int nt = s->rp.threads > 0 ? s->rp.threads : (int)sysconf(_SC_NPROCESSORS_ONLN);
  if (nt < 1) nt = 1;
  if (nt > 256) nt = 256;
  if (nt > s->rp.height) nt = s->rp.height;
  pthread_t th[256];
  worker_arg* args = (worker_arg*)calloc((size_t)nt, sizeof *args);
  if (!args) return fail(JADE_ERR_NOMEM, "out of memory");
  volatile int next_row = 0;
  double t0 = now_ms();
  for (int i = 0; i < nt; ++i) {
    args[i].s = s; args[i].spp = spp; args[i].first_sample = s->spp_done; args[i].next_row = &next_row;
    if (nt == 1) worker(&args[i]);
    else pthread_create(&th[i], NULL, worker, &args[i]);
  }
  if (nt > 1) for (int i = 0; i < nt; ++i) pthread_join(th[i], NULL);
  double t1 = now_ms();
  s->spp_done += spp;
  if (st) {
    for (int i = 0; i < nt; ++i) add_counters(st, &args[i].c);
    st->kernel_ms += t1 - t0;
  }
  free(args);
  return JADE_OK;
}

/* the oracle finishes every sample inside step(): nothing is ever carried over */
int jade_render_flush(jade_scene* s, jade_stats* st) {
  (void)st;
  if (!s || !s->have_rp) return fail(JADE_ERR_INVALID, "jade_render_begin not called");
  return JADE_OK;
}

int jade_render_resolve(jade_scene* s, float* out_rgb, uint8_t* out_bgr8) {
  return jade_render_resolve_ex(s, JADE_TONEMAP_ACES, 0.0f, out_rgb, out_bgr8);
}

int jade_render_resolve_ex(jade_scene* s, int tonemap, float limit, float* out_rgb, uint8_t* out_bgr8) {
  if (!s || !s->have_rp) return fail(JADE_ERR_INVALID, "jade_render_begin not called");
  if (tonemap != JADE_TONEMAP_ACES && tonemap != JADE_TONEMAP_REINHARD) return fail(JADE_ERR_INVALID, "unknown tone operator");
  const jade_render_params* rp = &s->rp;
  /* final_result * vec3(1.0 / spp), PathTrace.cu:1457 */
  float inv = (float)(1.0 / (double)s->spp_done);
  for (int y = 0; y < rp->height; ++y)
    for (int x = 0; x < rp->width; ++x) {
      if (!owns_pixel_s(s, x, y)) continue;
      const size_t px = (size_t)y * rp->width + x; /* in the caller's frame */
      const size_t pi = s->sum_slot ? (size_t)s->sum_slot[px] : px;
      const size_t np = s->sum_pixels;
      jvec3 tot = jv(s->sum[3 * pi], s->sum[3 * pi + 1], s->sum[3 * pi + 2]);
      /* lanes >= spp_done were never written: they are +0.0, and adding +0.0 any number of times is
       * adding it once (it only turns a -0.0 total into +0.0) - the untouched pages stay unmapped */
      const size_t used = s->spp_done < JADE_SAMPLE_LANES ? (size_t)s->spp_done : (size_t)JADE_SAMPLE_LANES;
      for (size_t l = 1; l < used; ++l) {
        const float* q = s->sum + 3 * (l * np + pi);
        tot = jv_add(tot, jv(q[0], q[1], q[2]));
      }
      if (used < JADE_SAMPLE_LANES) tot = jv_add(tot, jv(0.0f, 0.0f, 0.0f));
      jvec3 m = jv(tot.x * inv, tot.y * inv, tot.z * inv);
      if (out_rgb) { out_rgb[3 * px] = m.x; out_rgb[3 * px + 1] = m.y; out_rgb[3 * px + 2] = m.z; }
      if (out_bgr8) tonemap_pack(m, tonemap, limit, out_bgr8 + 3 * px);
    }
  return JADE_OK;
}

int jade_render(jade_scene* s, const jade_render_params* rp, float* out_rgb, uint8_t* out_bgr8, jade_stats* st) {
  int rc = jade_render_begin(s, rp);
  if (rc) return rc;
  if (rp->spp <= 0) return fail(JADE_ERR_INVALID, "spp must be positive");
  rc = jade_render_step(s, rp->spp, st);
  if (rc) return rc;
  return jade_render_resolve(s, out_rgb, out_bgr8);
}

int jade_render_multi(jade_scene* const* scenes, int ndev, const jade_render_params* rp, float* out_rgb, uint8_t* out_bgr8,
                      jade_stats* st) {
  /* the CPU oracle has no devices: render the shares one after the other into the same frame */
  if (!scenes || ndev <= 0 || !rp) return fail(JADE_ERR_INVALID, "null argument");
  for (int i = 0; i < ndev; ++i) {
    jade_render_params p = *rp;
    p.tile_rank = i;
    p.tile_nranks = ndev;
    int rc = jade_render(scenes[i], &p, out_rgb, out_bgr8, st);
    if (rc) return rc;
  }
  return JADE_OK;
}

int jade_render_query(jade_scene* s, int what, int64_t* value) {
  if (!s || !value || !s->have_rp) return fail(JADE_ERR_INVALID, "jade_render_begin not called");
  switch (what) {
    case JADE_Q_RECORDS_PER_PIXEL: *value = 1; return JADE_OK;
    case JADE_Q_STATE_BYTES: *value = (int64_t)(s->sum_pixels * 3 * JADE_SAMPLE_LANES * sizeof(float)); return JADE_OK;
    case JADE_Q_SUM_LANES: *value = JADE_SAMPLE_LANES; return JADE_OK;
    default: return fail(JADE_ERR_INVALID, "unknown query");
  }
}

int jade_render_resolve_tiles_device(jade_scene* s, float* dev_tiles, void* stream) {
  (void)s; (void)dev_tiles; (void)stream;
  return fail(JADE_ERR_UNSUPPORTED, "the CPU oracle has no device buffers");
}

int jade_trace_rays(jade_scene* s, int32_t n, const float* origins, const float* dirs, const int32_t* skip,
                    int32_t* hit_index, float* hit_dist, float* hit_point, jade_stats* st) {
  if (!s || n < 0 || !origins || !dirs || !skip || !hit_index) return fail(JADE_ERR_INVALID, "null argument");
  counters c;
  memset(&c, 0, sizeof c);
  double t0 = now_ms();
  for (int i = 0; i < n; ++i) {
    Ray r;
    r.startPoint = V3(origins + 3 * i);
    r.direction = V3(dirs + 3 * i);
    c.rays_secondary++;
    HitResult h = hit_bvh(s, r, skip[i], &c);
    hit_index[i] = h.isHit ? h.index : -1;
    if (hit_dist) hit_dist[i] = h.distance;
    if (hit_point) { hit_point[3 * i] = h.hitPoint.x; hit_point[3 * i + 1] = h.hitPoint.y; hit_point[3 * i + 2] = h.hitPoint.z; }
  }
  if (st) { add_counters(st, &c); st->kernel_ms += now_ms() - t0; }
  return JADE_OK;
}

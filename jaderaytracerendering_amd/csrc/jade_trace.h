// jade_trace.h — BVH traversal + ray/triangle intersection for gfx950.
//
// Semantics are those of hitBVH / hitAABB / hitArray / hitTriangle
// (PathTrace.cu:705-859), including the quirks that decide which triangle a
// ray reports: no pruning against the best hit (:806-856), near child first by
// d1 < d2 (:835-848), boxes entered only when the slab value is > 0 (:770,
// :835-855), the source triangle skipped by index (:782), strict "<" so the
// first of two equal distances wins (:787, :816), NaN flowing through the
// ternary min/max of hitAABB (:484-494, :764-765).
//
// What is different is how the work is laid out for a 64-lane wavefront:
//   - one ray per lane, rays taken from a compacted queue by persistent
//     workgroups: a wave claims a chunk with one atomic and refills its idle
//     lanes from that chunk, so short rays do not leave lanes idle behind long ones;
//   - the unit of work per loop iteration is one primitive (a node visit or
//     one triangle test), not a whole 8-triangle leaf;
//   - the near child is followed directly and only the far child is pushed,
//     which visits nodes in exactly the reference's order with half the stack
//     traffic;
//   - the stack lives in LDS as stack[level][lane] (bank-conflict free: lane
//     l always hits bank l % 32 of its half-wave), deeper levels spill to a
//     per-lane global area (JADE_BVH_STACK_CAPACITY entries in total);
//   - both children's boxes come from the parent's 64-B record, so a node
//     visit is four 16-B loads of one line (see jade_device.h);
//   - 1/dir and normalize(dir), which the reference recomputes per node and
//     per triangle (:710, :759), are computed once per ray — same values.
#pragma once
#include "jade_device.h"

struct TraceHit {
  int32_t index;  // -1 = miss
  float dist;
  jvec3 point;
};

struct LdsStack {
  uint32_t* lds;       // base of this lane's column: lds[level * blockDim + tid]
  uint32_t* spill;     // global, spill[(level - JADE_LDS_STACK) * stride + gtid]
  uint32_t stride_lds;
  uint32_t stride_spill;
};

static __device__ __forceinline__ void stack_push(const LdsStack& s, int sp, uint32_t v) {
  if (sp < JADE_LDS_STACK) s.lds[sp * s.stride_lds] = v;
  else s.spill[(size_t)(sp - JADE_LDS_STACK) * s.stride_spill] = v;
}
static __device__ __forceinline__ uint32_t stack_pop(const LdsStack& s, int sp) {
  if (sp < JADE_LDS_STACK) return s.lds[sp * s.stride_lds];
  return s.spill[(size_t)(sp - JADE_LDS_STACK) * s.stride_spill];
}

// hitAABB, PathTrace.cu:758-771, with 1/dir hoisted.  `exact` selects the
// NaN-faithful ternary form; it is only needed when a component of 1/dir is
// not finite (0 * inf is the one way a NaN can appear for a finite scene),
// otherwise v_min/v_max give bit-identical slab values.
static __device__ __forceinline__ float slab(jvec3 o, jvec3 inv, float ax, float ay, float az, float bx, float by,
                                             float bz, bool exact) {
  float fx = (bx - o.x) * inv.x, fy = (by - o.y) * inv.y, fz = (bz - o.z) * inv.z;
  float nx = (ax - o.x) * inv.x, ny = (ay - o.y) * inv.y, nz = (az - o.z) * inv.z;
  float t0, t1;
  if (exact) {
    float tmaxx = fx > nx ? fx : nx, tmaxy = fy > ny ? fy : ny, tmaxz = fz > nz ? fz : nz;
    float tminx = fx < nx ? fx : nx, tminy = fy < ny ? fy : ny, tminz = fz < nz ? fz : nz;
    t1 = jade_fminf(tmaxx, jade_fminf(tmaxy, tmaxz));
    t0 = jade_fmaxf(tminx, jade_fmaxf(tminy, tminz));
  } else {
    t1 = __builtin_fminf(__builtin_fmaxf(fx, nx), __builtin_fminf(__builtin_fmaxf(fy, ny), __builtin_fmaxf(fz, nz)));
    t0 = __builtin_fmaxf(__builtin_fminf(fx, nx), __builtin_fmaxf(__builtin_fminf(fy, ny), __builtin_fminf(fz, nz)));
  }
  return (t1 >= t0) ? ((t0 > 0.0f) ? t0 : t1) : -1.0f;
}

// hitTriangle, PathTrace.cu:705-754, with normalize(dir) hoisted.
static __device__ __forceinline__ bool tri_test(jvec3 p1, jvec3 p2, jvec3 p3, jvec3 o, jvec3 dn, float* dist_out,
                                                jvec3* point_out) {
  jvec3 sa = jv_sub(p1, jv_scale(dn, jv_dot(dn, jv_sub(p1, o))));
  jvec3 sb = jv_sub(p2, jv_scale(dn, jv_dot(dn, jv_sub(p2, o))));
  jvec3 sc = jv_sub(p3, jv_scale(dn, jv_dot(dn, jv_sub(p3, o))));
  jvec3 pa = jv_sub(sa, o), pb = jv_sub(sb, o), pc = jv_sub(sc, o);
  float papb = jv_mixed(dn, pa, pb);
  float pbpc = jv_mixed(dn, pb, pc);
  float pcpa = jv_mixed(dn, pc, pa);
  if ((papb > 0 && pbpc > 0 && pcpa > 0) || (papb < 0 && pbpc < 0 && pcpa < 0)) {
    jvec3 eb = jv_sub(sb, sa), ec = jv_sub(sc, sa), q = jv_sub(o, sa);
    float divider = jade_diffprod(eb.x, ec.y, eb.y, ec.x);
    float rate_a = jade_diffprod(ec.y, q.x, ec.x, q.y) / divider;
    float rate_b = jade_fma(eb.x, q.y, (-eb.y) * q.x) / divider;
    jvec3 P = jv_add(jv_add(p1, jv_scale(jv_sub(p2, p1), rate_a)), jv_scale(jv_sub(p3, p1), rate_b));
    float distance = jv_dot(jv_sub(P, o), dn);
    if (distance > 0) {
      *dist_out = distance;
      *point_out = P;
      return true;
    }
  }
  return false;
}

static __device__ __forceinline__ bool finite_f(float x) { return (jade_f2u(x) & 0x7f800000u) != 0x7f800000u; }

// One lane's traversal state for hitBVH (PathTrace.cu:795-859), advanced one
// small unit per call: either one internal-node visit (both children's slab
// tests, near-first descent, far child pushed) or up to two triangle tests of
// the current leaf.  Keeping the unit of work small is what lets a 64-lane wave
// mix lanes that are deep in a leaf with lanes that are still descending
// without one serialising the other, and lets finished lanes be refilled.
#ifndef JADE_RECOMPUTE_POINT
#define JADE_RECOMPUTE_POINT 0
#endif
struct RayState {
  jvec3 o, inv, dn;
  int32_t skip;
  bool exact;
  uint32_t cur;          // current ref (internal node or leaf)
  uint32_t tri_i, tri_n; // leaf cursor: next triangle, end (tri_i < tri_n while inside a leaf)
  int sp;
  int32_t best_index;
  float best_dist;
#if !JADE_RECOMPUTE_POINT
  jvec3 best_point;
#endif
};

static __device__ __forceinline__ void ray_begin(RayState& r, const DevScene& S, jvec3 o, jvec3 d, int32_t skip, uint32_t& V) {
  r.o = o;
  r.inv = jv(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  r.dn = jv_normalize(d);
  r.skip = skip;
  r.exact = !(finite_f(r.inv.x) && finite_f(r.inv.y) && finite_f(r.inv.z)) || !finite_f(o.x) || !finite_f(o.y) || !finite_f(o.z);
  r.sp = 0;
  r.best_index = -1;
  r.best_dist = JADE_INF_F;
#if !JADE_RECOMPUTE_POINT
  r.best_point = jv(0, 0, 0);
#endif
  r.cur = S.root_ref;
  r.tri_i = r.tri_n = 0;
  if (r.cur & JADE_REF_LEAF) {
    r.tri_i = (r.cur & 0x7fffffffu) >> 4;
    r.tri_n = r.tri_i + (r.cur & 15u);
  }
  V += 1;  // the root record
}

// Make `ref` current; returns false when the stack is empty (ray finished).
static __device__ __forceinline__ bool ray_pop(RayState& r, const LdsStack& stk) {
  if (r.sp == 0) return false;
  r.cur = stack_pop(stk, --r.sp);
  if (r.cur & JADE_REF_LEAF) {
    r.tri_i = (r.cur & 0x7fffffffu) >> 4;
    r.tri_n = r.tri_i + (r.cur & 15u);
  }
  return true;
}
static __device__ __forceinline__ void ray_goto(RayState& r, uint32_t ref) {
  r.cur = ref;
  if (ref & JADE_REF_LEAF) {
    r.tri_i = (ref & 0x7fffffffu) >> 4;
    r.tri_n = r.tri_i + (ref & 15u);
  }
}

// Development ablations (cdna_hip_programming.md rule 17): JADE_ABLATE_* repeat a
// piece of work without changing any result, to price that piece.  Off in product builds.
#ifndef JADE_ABLATE_TRI
#define JADE_ABLATE_TRI 0
#endif
#ifndef JADE_ABLATE_SLAB
#define JADE_ABLATE_SLAB 0
#endif
#ifndef JADE_ABLATE_LOAD
#define JADE_ABLATE_LOAD 0
#endif

// The hit point of the winning triangle.  With JADE_RECOMPUTE_POINT the three
// registers of best_point are not carried through the traversal: the point is
// recomputed once at the end by the same arithmetic on the same operands (same bits).
static __device__ __forceinline__ jvec3 ray_hit_point(const RayState& r, const DevScene& S) {
#if JADE_RECOMPUTE_POINT
  const float4* t0 = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.tverts) + (uint32_t)r.best_index * 48u);
  const float4 a = t0[0], b = t0[1], c = t0[2];
  float dist;
  jvec3 P = jv(0, 0, 0);
  (void)tri_test(jv(a.x, a.y, a.z), jv(b.x, b.y, b.z), jv(c.x, c.y, c.z), r.o, r.dn, &dist, &P);
  return P;
#else
  return r.best_point;
#endif
}

// One traversal unit, split by kind so that a wave can run only one kind per
// iteration (see k_trace).  Both return false when the ray has finished.
static __device__ __forceinline__ bool ray_wants_tri(const RayState& r) { return r.tri_i < r.tri_n; }

static __device__ __forceinline__ bool ray_step_tri(RayState& r, const DevScene& S, const LdsStack& stk, uint32_t& T) {
    // ---- up to two triangles of the current leaf (hitArray, PathTrace.cu:776-792),
    // tested in index order; both vertex records are requested before either is used
    const uint32_t i = r.tri_i;
    const bool two = JADE_TRIS_PER_STEP > 1 && i + 1 < r.tri_n;
    const uint32_t j = two ? i + 1 : i;
    // 32-bit byte offsets from a scalar base (n_tris < 2^27: 48 B * i fits): saddr + voffset loads
    const char* tb = reinterpret_cast<const char*>(S.tverts);
    const float4* t0 = reinterpret_cast<const float4*>(tb + i * 48u);
    const float4* t1 = reinterpret_cast<const float4*>(tb + j * 48u);
    const float4 a0 = t0[0], b0 = t0[1], c0 = t0[2];
    const float4 a1 = t1[0], b1 = t1[1], c1 = t1[2];
    r.tri_i = j + 1;
    float dist;
    jvec3 P;
#if JADE_ABLATE_TRI
    {
      float d2; jvec3 P2;
      jvec3 o2 = jv(r.o.x + 1e-30f, r.o.y, r.o.z);
      bool h2 = tri_test(jv(a0.x, a0.y, a0.z), jv(b0.x, b0.y, b0.z), jv(c0.x, c0.y, c0.z), o2, r.dn, &d2, &P2);
      asm volatile("" ::"v"(h2 ? d2 + P2.x : 0.0f));
    }
#endif
#if JADE_ABLATE_LOAD
    {
      const float4 x = S.tverts[3 * (size_t)i];
      const float4 y = S.tverts[3 * (size_t)(__float_as_uint(x.w) & 1u)];  // dependent on the first
      asm volatile("" ::"v"(y.x));
    }
#endif
    if ((int32_t)i != r.skip) {
      T += 1;
      if (tri_test(jv(a0.x, a0.y, a0.z), jv(b0.x, b0.y, b0.z), jv(c0.x, c0.y, c0.z), r.o, r.dn, &dist, &P) && dist < r.best_dist) {
        r.best_index = (int32_t)i;
        r.best_dist = dist;
#if !JADE_RECOMPUTE_POINT
        r.best_point = P;
#endif
      }
    }
    if (two && (int32_t)j != r.skip) {
      T += 1;
      if (tri_test(jv(a1.x, a1.y, a1.z), jv(b1.x, b1.y, b1.z), jv(c1.x, c1.y, c1.z), r.o, r.dn, &dist, &P) && dist < r.best_dist) {
        r.best_index = (int32_t)j;
        r.best_dist = dist;
#if !JADE_RECOMPUTE_POINT
        r.best_point = P;
#endif
      }
    }
    if (r.tri_i < r.tri_n) return true;
    return ray_pop(r, stk);
}

static __device__ __forceinline__ bool ray_step_node(RayState& r, const DevScene& S, const LdsStack& stk, uint32_t& V) {
  if (r.cur & JADE_REF_LEAF) return ray_pop(r, stk);  // empty leaf (cannot happen for a valid BVH)
  // ---- one internal node
  const float4* nd = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.nodes) + r.cur * 64u);
  const float4 a = nd[0], b = nd[1], c = nd[2];
  const uint4 rf = *reinterpret_cast<const uint4*>(nd + 3);
  float d1 = -1.0f, d2 = -1.0f;
#if JADE_ABLATE_SLAB
  {
    jvec3 o2 = jv(r.o.x + 1e-30f, r.o.y, r.o.z);
    float e1 = slab(o2, r.inv, a.x, a.y, a.z, a.w, b.x, b.y, r.exact), e2 = slab(o2, r.inv, b.z, b.w, c.x, c.y, c.z, c.w, r.exact);
    asm volatile("" ::"v"(e1 + e2));
  }
#endif
#if JADE_ABLATE_LOAD
  {
    const float4 y = S.nodes[4 * (size_t)(rf.x & 1u)];  // dependent on this node's record
    asm volatile("" ::"v"(y.x));
  }
#endif
  if (rf.x != JADE_REF_NONE) {
    V += 1;
    d1 = slab(r.o, r.inv, a.x, a.y, a.z, a.w, b.x, b.y, r.exact);
  }
  if (rf.y != JADE_REF_NONE) {
    V += 1;
    d2 = slab(r.o, r.inv, b.z, b.w, c.x, c.y, c.z, c.w, r.exact);
  }
  if (d1 > 0 && d2 > 0) {
    if (d1 < d2) {
      stack_push(stk, r.sp++, rf.y);
      ray_goto(r, rf.x);
    } else {
      stack_push(stk, r.sp++, rf.x);
      ray_goto(r, rf.y);
    }
    return true;
  }
  if (d1 > 0) {
    ray_goto(r, rf.x);
    return true;
  }
  if (d2 > 0) {
    ray_goto(r, rf.y);
    return true;
  }
  return ray_pop(r, stk);
}

// Returns false when the ray has finished.
static __device__ __forceinline__ bool ray_step(RayState& r, const DevScene& S, const LdsStack& stk, uint32_t& V, uint32_t& T) {
  if (ray_wants_tri(r)) return ray_step_tri(r, S, stk, T);
  return ray_step_node(r, S, stk, V);
}

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
//     workgroups (one atomic per wave per 64 rays);
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

// hitBVH, PathTrace.cu:795-859.  V / T are the exact work counters
// (node records needed, triangles tested).
static __device__ __forceinline__ TraceHit trace_ray(const DevScene& S, jvec3 o, jvec3 d, int32_t skip,
                                                     const LdsStack& stk, uint32_t& V, uint32_t& T) {
  TraceHit best;
  best.index = -1;
  best.dist = JADE_INF_F;
  best.point = jv(0, 0, 0);
  const jvec3 inv = jv(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  const jvec3 dn = jv_normalize(d);
  const bool exact = !(finite_f(inv.x) && finite_f(inv.y) && finite_f(inv.z)) || !finite_f(o.x) || !finite_f(o.y) ||
                     !finite_f(o.z);
  int sp = 0;
  uint32_t cur = S.root_ref;
  V += 1;  // the root record
  for (;;) {
    if (cur & JADE_REF_LEAF) {
      const uint32_t first = (cur & 0x7fffffffu) >> 4, n = cur & 15u;
      for (uint32_t i = first; i < first + n; ++i) {
        if ((int32_t)i == skip) continue;
        const float4 a = S.tverts[3 * (size_t)i], b = S.tverts[3 * (size_t)i + 1], c = S.tverts[3 * (size_t)i + 2];
        float dist;
        jvec3 P;
        T += 1;
        if (tri_test(jv(a.x, a.y, a.z), jv(b.x, b.y, b.z), jv(c.x, c.y, c.z), o, dn, &dist, &P) && dist < best.dist) {
          best.index = (int32_t)i;
          best.dist = dist;
          best.point = P;
        }
      }
      if (sp == 0) break;
      cur = stack_pop(stk, --sp);
    } else {
      const float4* nd = S.nodes + 4 * (size_t)cur;
      const float4 a = nd[0], b = nd[1], c = nd[2];
      const uint4 r = *reinterpret_cast<const uint4*>(nd + 3);
      float d1 = -1.0f, d2 = -1.0f;
      if (r.x != JADE_REF_NONE) {
        V += 1;
        d1 = slab(o, inv, a.x, a.y, a.z, a.w, b.x, b.y, exact);
      }
      if (r.y != JADE_REF_NONE) {
        V += 1;
        d2 = slab(o, inv, b.z, b.w, c.x, c.y, c.z, c.w, exact);
      }
      if (d1 > 0 && d2 > 0) {
        if (d1 < d2) {
          stack_push(stk, sp++, r.y);
          cur = r.x;
        } else {
          stack_push(stk, sp++, r.x);
          cur = r.y;
        }
      } else if (d1 > 0) {
        cur = r.x;
      } else if (d2 > 0) {
        cur = r.y;
      } else {
        if (sp == 0) break;
        cur = stack_pop(stk, --sp);
      }
    }
  }
  return best;
}

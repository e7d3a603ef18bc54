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
//     l always hits bank l % 32 of its half-wave); 12 levels cover every ray of
//     the benchmark scenes, deeper levels spill to a per-lane global area
//     (JADE_BVH_STACK_CAPACITY entries in total);
//   - the parts of the ray state that the triangle test does not read (1/dir,
//     the best hit so far) live in the same LDS column, so the kernel needs 60
//     VGPRs and a SIMD holds 8 waves: the kernel is latency-bound, waves are
//     what hides the latency;
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

// A lane's column of LDS words, lds[word * JADE_TRACE_BLOCK + tid] (bank-conflict free):
// words [0, JADE_LDS_STACK) are the traversal stack, the JADE_LDS_STATE words after it hold
// the parts of the ray state that the triangle test does not read (see RayState).
struct LdsStack {
  uint32_t* lds;       // base of this lane's column
  uint32_t* spill;     // global, spill[(level - JADE_LDS_STACK) * stride + gtid]
  uint32_t stride_spill;
};
enum { LW_INVX = JADE_LDS_STACK, LW_INVY, LW_INVZ, LW_BEST_DIST, LW_BEST_INDEX, LW_PX, LW_PY, LW_PZ, LW_END };
static_assert(LW_END - JADE_LDS_STACK == JADE_LDS_STATE, "JADE_LDS_STATE must count the LW_* state words");

static __device__ __forceinline__ void lds_put(const LdsStack& s, int word, uint32_t v) { s.lds[word * JADE_TRACE_BLOCK] = v; }
static __device__ __forceinline__ uint32_t lds_get(const LdsStack& s, int word) { return s.lds[word * JADE_TRACE_BLOCK]; }
static __device__ __forceinline__ void lds_putf(const LdsStack& s, int word, float v) { lds_put(s, word, jade_f2u(v)); }
static __device__ __forceinline__ float lds_getf(const LdsStack& s, int word) { return jade_u2f(lds_get(s, word)); }

static __device__ __forceinline__ void stack_push(const LdsStack& s, int sp, uint32_t v) {
  if (sp < JADE_LDS_STACK) s.lds[sp * JADE_TRACE_BLOCK] = v;
  else s.spill[(size_t)(sp - JADE_LDS_STACK) * s.stride_spill] = v;
}
static __device__ __forceinline__ uint32_t stack_pop(const LdsStack& s, int sp) {
  if (sp < JADE_LDS_STACK) return s.lds[sp * JADE_TRACE_BLOCK];
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
    // no NaN can occur here, so plain v_max/v_min ARE the ternaries; spelled as instructions
    // because the builtins make the compiler quiet each operand first (six extra v_max x,x,x)
    float hx, hy, hz, lx, ly, lz;
    asm("v_max_f32_e32 %0, %1, %2" : "=v"(hx) : "v"(fx), "v"(nx));
    asm("v_max_f32_e32 %0, %1, %2" : "=v"(hy) : "v"(fy), "v"(ny));
    asm("v_max_f32_e32 %0, %1, %2" : "=v"(hz) : "v"(fz), "v"(nz));
    asm("v_min_f32_e32 %0, %1, %2" : "=v"(lx) : "v"(fx), "v"(nx));
    asm("v_min_f32_e32 %0, %1, %2" : "=v"(ly) : "v"(fy), "v"(ny));
    asm("v_min_f32_e32 %0, %1, %2" : "=v"(lz) : "v"(fz), "v"(nz));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(t1) : "v"(hx), "v"(hy), "v"(hz));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t0) : "v"(lx), "v"(ly), "v"(lz));
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

// One lane's traversal state for hitBVH (PathTrace.cu:795-859), advanced one small unit per
// call: either one internal-node visit (both children's slab tests, near-first descent, far
// child pushed) or one triangle test of the current leaf.  Keeping the unit of work small is
// what lets a 64-lane wave mix lanes that are deep in a leaf with lanes that are still
// descending without one serialising the other, and lets finished lanes be refilled.
//
// The kernel is latency-bound, so what matters is how many waves a SIMD holds: the state is cut
// to what fits 64 VGPRs (8 waves).  In registers: o, normalize(d), the skip index, the
// cursor and the stack pointer.  In the lane's LDS column: 1/d (read by node visits only) and
// the best hit so far (touched only when a triangle is actually hit).  The leaf cursor IS the
// leaf reference: LEAF | 3 * first << 4 | count.  Bits 4-30 are the byte offset of the next
// triangle's 48-B vertex record, so a step needs one AND to address it and +47 to advance
// (offset + 48, count - 1); triangles are identified by that offset until the ray ends.
struct V3ld {
  float x, y, z;
};
struct RayState {
  jvec3 o, dn;
  uint32_t skipx;  // bit 31: a component of o or 1/d is not finite (NaN-faithful slab needed); bits 0-30: 48 * source triangle, 0x7fffffff = none
  uint32_t cur;    // internal-node ref, or leaf cursor
  int sp;
};

static __device__ __forceinline__ void ray_begin(RayState& r, const LdsStack& stk, const DevScene& S, jvec3 o, jvec3 d, int32_t skip) {
  r.o = o;
  const jvec3 inv = jv(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  r.dn = jv_normalize(d);
  const bool exact = !(finite_f(inv.x) && finite_f(inv.y) && finite_f(inv.z)) || !finite_f(o.x) || !finite_f(o.y) || !finite_f(o.z);
  r.skipx = (skip < 0 ? 0x7fffffffu : (uint32_t)skip * 48u) | (exact ? 0x80000000u : 0u);
  r.sp = 0;
  r.cur = S.root_ref;
  lds_putf(stk, LW_INVX, inv.x);
  lds_putf(stk, LW_INVY, inv.y);
  lds_putf(stk, LW_INVZ, inv.z);
  lds_putf(stk, LW_BEST_DIST, JADE_INF_F);
  lds_put(stk, LW_BEST_INDEX, 0xffffffffu);
}

// Make the top of the stack current; returns false when the stack is empty (ray finished).
static __device__ __forceinline__ bool ray_pop(RayState& r, const LdsStack& stk) {
  if (r.sp == 0) return false;
  r.cur = stack_pop(stk, --r.sp);
  return true;
}

// Development ablations (cdna_hip_programming.md rule 17): JADE_ABLATE_* repeat a
// piece of work without changing any result, to price that piece.  Off in product builds.
#ifndef JADE_ABLATE_TRI
#define JADE_ABLATE_TRI 0
#endif
#ifndef JADE_ABLATE_SLAB
#define JADE_ABLATE_SLAB 0
#endif

static __device__ __forceinline__ int32_t ray_best_index(const LdsStack& stk) {
  const uint32_t off = lds_get(stk, LW_BEST_INDEX);  // byte offset of the vertex record, or ~0
  return off == 0xffffffffu ? -1 : (int32_t)(off / 48u);
}
static __device__ __forceinline__ jvec3 ray_hit_point(const LdsStack& stk) {
  return jv(lds_getf(stk, LW_PX), lds_getf(stk, LW_PY), lds_getf(stk, LW_PZ));
}

static __device__ __forceinline__ bool ray_in_leaf(const RayState& r) { return (r.cur & JADE_REF_LEAF) != 0; }

// The next triangle of the current leaf (hitArray, PathTrace.cu:776-792), in index order.
// *tested: this lane ran an intersection test (the skipped source triangle does not count).
static __device__ __forceinline__ bool ray_step_tri(RayState& r, const DevScene& S, const LdsStack& stk, bool* tested) {
  if (r.cur & 15u) {
    const uint32_t off = r.cur & 0x7ffffff0u;
    // saddr + 32-bit voffset loads; three 12-B loads: the pad word of each vertex is never brought into a register
    const char* t0 = reinterpret_cast<const char*>(S.tverts) + off;
    const V3ld a0 = *reinterpret_cast<const V3ld*>(t0), b0 = *reinterpret_cast<const V3ld*>(t0 + 16),
               c0 = *reinterpret_cast<const V3ld*>(t0 + 32);
    r.cur += 47u;  // next record, count - 1
#if JADE_ABLATE_TRI
    {
      float d2; jvec3 P2;
      jvec3 o2 = jv(r.o.x + 1e-30f, r.o.y, r.o.z);
      bool h2 = tri_test(jv(a0.x, a0.y, a0.z), jv(b0.x, b0.y, b0.z), jv(c0.x, c0.y, c0.z), o2, r.dn, &d2, &P2);
      asm volatile("" ::"v"(h2 ? d2 + P2.x : 0.0f));
    }
#endif
    if (off != (r.skipx & 0x7fffffffu)) {
      *tested = true;
      float dist;
      jvec3 P;
      if (tri_test(jv(a0.x, a0.y, a0.z), jv(b0.x, b0.y, b0.z), jv(c0.x, c0.y, c0.z), r.o, r.dn, &dist, &P) &&
          dist < lds_getf(stk, LW_BEST_DIST)) {
        lds_putf(stk, LW_BEST_DIST, dist);
        lds_put(stk, LW_BEST_INDEX, off);
        lds_putf(stk, LW_PX, P.x);
        lds_putf(stk, LW_PY, P.y);
        lds_putf(stk, LW_PZ, P.z);
      }
    }
  }
  if (r.cur & 15u) return true;
  return ray_pop(r, stk);  // leaf finished (or empty: cannot happen for a valid BVH)
}

// One internal node.  *c1, *c2: the child exists (its record counts as visited).
static __device__ __forceinline__ bool ray_step_node(RayState& r, const DevScene& S, const LdsStack& stk, bool* c1, bool* c2) {
  const float4* nd = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.nodes) + r.cur * 64u);
  const float4 a = nd[0], b = nd[1], c = nd[2];
  const uint2 rf = *reinterpret_cast<const uint2*>(nd + 3);
  const jvec3 inv = jv(lds_getf(stk, LW_INVX), lds_getf(stk, LW_INVY), lds_getf(stk, LW_INVZ));
  const bool exact = (int32_t)r.skipx < 0;
  float d1 = -1.0f, d2 = -1.0f;
#if JADE_ABLATE_SLAB
  {
    jvec3 o2 = jv(r.o.x + 1e-30f, r.o.y, r.o.z);
    float e1 = slab(o2, inv, a.x, a.y, a.z, a.w, b.x, b.y, exact), e2 = slab(o2, inv, b.z, b.w, c.x, c.y, c.z, c.w, exact);
    asm volatile("" ::"v"(e1 + e2));
  }
#endif
  if (rf.x != JADE_REF_NONE) {
    *c1 = true;
    d1 = slab(r.o, inv, a.x, a.y, a.z, a.w, b.x, b.y, exact);
  }
  if (rf.y != JADE_REF_NONE) {
    *c2 = true;
    d2 = slab(r.o, inv, b.z, b.w, c.x, c.y, c.z, c.w, exact);
  }
  const bool in1 = d1 > 0, in2 = d2 > 0;
  if (in1 && in2) {  // near child first (d1 < d2, PathTrace.cu:835-848): follow it, push the far one
    const bool first = d1 < d2;
    stack_push(stk, r.sp++, first ? rf.y : rf.x);
    r.cur = first ? rf.x : rf.y;
    return true;
  }
  if (in1 || in2) {
    r.cur = in1 ? rf.x : rf.y;
    return true;
  }
  return ray_pop(r, stk);
}

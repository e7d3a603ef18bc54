// jade_shade.h — the bounce loop of pathTracing (PathTrace.cu:905-1416) and the
// sample loop of render_pixel (:1418-1455) as a per-pixel state machine.
//
// The reference runs one thread per pixel through `spp` samples and calls
// hitBVH inline (nEmit + 2 times per bounce).  Here a pixel is a record in
// HBM; each pass of k_shade (a) folds the hit results of the rays it issued on
// the previous pass into the path, (b) samples the next bounce and (c) emits
// ALL rays of that bounce at once into a compacted queue for k_trace.  That is
// possible because no random draw of a bounce depends on a hit result of the
// same bounce: the draw order below is the textual order of the
// curand_uniform calls, only the hitBVH calls are deferred.
//
// Radiance bookkeeping: the reference pushes (dir, rate) pairs on two
// 128-entry stacks and unwinds them Horner-style (:1410-1413).  The same sum
// is accumulated forward here (acc += thr * dir; thr *= rate), which differs
// from the unwind only by fp32 rounding of the radiance (never of a decision).
#pragma once
#include "jade_device.h"

static __device__ __forceinline__ int mirror_index(int i, int n) {
  int m = i % (2 * n);
  if (m < 0) m += 2 * n;
  if (m >= n) m = 2 * n - 1 - m;
  return m;
}

// sampleHdr / SampleSphericalMap, PathTrace.cu:686-702; software bilinear with
// mirror addressing in place of tex2D (gfx950 has no sampler path).
static __device__ jvec3 sample_hdr(const DevScene& S, jvec3 v) {
  jvec3 nv = jv_normalize(v);
  float ux = jade_atan2f(nv.z, nv.x);
  float uy = jade_asinf(nv.y);
  // (float)((double)x / C) written as (float)((double)x * (1.0 / C)): NOT the same in general, but the same for EVERY float |x| <= 3.2
  // with C = 2 * JADE_PI_D and |x| <= 1.6 with C = JADE_PI_D - all 4.3e9 of them compared (tests/native/dpdiv_exhaustive.c,
  // tests/test_fpmath.py) - and atan2 / asin return nothing outside those ranges.  Two double-precision divisions (~15 instructions
  // each at half rate or less) become two multiplications in a function every sky sample of the first pass runs.
  ux = (float)((double)ux * (1.0 / (2.0 * JADE_PI_D)));
  uy = (float)((double)uy * (1.0 / JADE_PI_D));
  ux = (float)((double)ux + 0.5);
  uy = (float)((double)uy + 0.5);
  uy = (float)(1.0 - (double)uy);
  const int W = S.env_w, H = S.env_h;
  if (jade_isnan(ux)) ux = 0.0f;
  if (jade_isnan(uy)) uy = 0.0f;
  float x = ux * (float)W - 0.5f;
  float y = uy * (float)H - 0.5f;
  float xf = jade_floorf(x), yf = jade_floorf(y);
  float ax = x - xf, ay = y - yf;
  int i0 = mirror_index((int)xf, W), i1 = mirror_index((int)xf + 1, W);
  int j0 = mirror_index((int)yf, H), j1 = mirror_index((int)yf + 1, H);
  float w00 = (1.0f - ax) * (1.0f - ay), w10 = ax * (1.0f - ay);
  float w01 = (1.0f - ax) * ay, w11 = ax * ay;
  const float* t00 = S.env + 3 * ((size_t)j0 * W + i0);
  const float* t10 = S.env + 3 * ((size_t)j0 * W + i1);
  const float* t01 = S.env + 3 * ((size_t)j1 * W + i0);
  const float* t11 = S.env + 3 * ((size_t)j1 * W + i1);
  jvec3 c;
  c.x = ((w00 * t00[0] + w10 * t10[0]) + w01 * t01[0]) + w11 * t11[0];
  c.y = ((w00 * t00[1] + w10 * t10[1]) + w01 * t01[1]) + w11 * t11[1];
  c.z = ((w00 * t00[2] + w10 * t10[2]) + w01 * t01[2]) + w11 * t11[2];
  c.x = c.x < 10.0f ? c.x : 10.0f;
  c.y = c.y < 10.0f ? c.y : 10.0f;
  c.z = c.z < 10.0f ? c.z : 10.0f;
  return c;
}

static __device__ __forceinline__ jvec3 sphere_dir(uint32_t* rng) {  // PathTrace.cu:968-971
  float cosine_theta = (float)(2.0 * ((double)jade_rand(rng) - 0.5));
  float sine_theta = jade_sqrt(1.0f - cosine_theta * cosine_theta);
  float fai_value = (float)(2.0 * JADE_PI_D * (double)jade_rand(rng));
  float sn, cs;
  jade_sincosf(fai_value, &sn, &cs);
  return jv(sine_theta * cs, sine_theta * sn, cosine_theta);
}

// Environment importance sampling (jade_render_params.env_sampling = JADE_ENV_IMPORTANCE; NOT the reference's estimator): a texel of
// the map by the alias method, a point in it uniformly, the direction SampleSphericalMap (PathTrace.cu:686-692) maps to that point.
// ratio = (pdf of the reference's uniform hemisphere sampling, 1 / 2 pi) / (pdf of this direction): what the reference's weight is
// multiplied with.  Four draws.
static __device__ __forceinline__ jvec3 env_sample(const DevScene& S, uint32_t* rng, float* ratio) {
  const uint32_t W = (uint32_t)S.env_w, N = W * (uint32_t)S.env_h;
  uint32_t idx = (uint32_t)(jade_rand(rng) * (float)N);
  idx = idx < N ? idx : N - 1u;
  const float u2 = jade_rand(rng);
  const uint4 e = S.env_alias[idx];
  const bool own = u2 < jade_u2f(e.x);
  const float pdf_n = jade_u2f(own ? e.z : e.w);  // the chosen texel's probability x N
  idx = own ? idx : e.y;
  const uint32_t j = idx / W, i = idx - j * W;
  const float u = ((float)i + jade_rand(rng)) / (float)W, v = ((float)j + jade_rand(rng)) / (float)S.env_h;
  const float theta = (float)JADE_PI_D * v, phi = (float)(2.0 * JADE_PI_D) * (u - 0.5f);
  float st, ct, sp, cp;
  jade_sincosf(theta, &st, &ct);
  jade_sincosf(phi, &sp, &cp);
  // pdf(direction) = pdf_n / (2 pi^2 sin theta); against 1 / (2 pi): ratio = pi sin theta / pdf_n
  *ratio = (float)JADE_PI_D * st / pdf_n;
  return jv(st * cp, ct, st * sp);
}

static __device__ __forceinline__ jvec3 tri_point(const jade_triangle* t, float rx, float ry) {
  jvec3 p1 = V3(t->p1);
  return jv_add(jv_add(p1, jv_scale(jv_sub(V3(t->p2), p1), rx)), jv_scale(jv_sub(V3(t->p3), p1), ry));
}

static __device__ __forceinline__ float tri_size(const jade_triangle* t) {  // PathTrace.cu:897-903
  jvec3 cp = jv_cross(jv_sub(V3(t->p2), V3(t->p1)), jv_sub(V3(t->p3), V3(t->p1)));
  return 0.5f * jade_sqrt(jv_dot(cp, cp));
}

// The LIMIT a shadow ray carries to k_trace (jade_device.h, PathState.early_exit): hitTriangle (PathTrace.cu:705-754) of the ray
// with the emitter it aims at, statement for statement as the walk's own test evaluates it (jade_trace.h pair_core / tri_hit: the
// same helpers, so the same bits - test_gpu_early_exit.py compares them) - the distance if HitResult.isHit and hitArray would
// record it (distance < INF, :787), else JADE_INF_F: hitBVH can then not return this emitter, and any hit settles the query.
// The caller's use of the query (:957, 1097, 1293) is "isHit && index == emitter": true iff no triangle of the walk is recorded
// before the emitter with a distance <= its own, so a recorded hit STRICTLY nearer than this value makes it false for good.
static __device__ float shadow_limit(const jade_triangle* t, jvec3 o, jvec3 d) {
  const jvec3 p1 = V3(t->p1), p2 = V3(t->p2), p3 = V3(t->p3);
  const jvec3 dn = jv_normalize(d);
  const jvec3 sa = jv_sub(p1, jv_scale(dn, jv_dot(dn, jv_sub(p1, o))));
  const jvec3 sb = jv_sub(p2, jv_scale(dn, jv_dot(dn, jv_sub(p2, o))));
  const jvec3 sc = jv_sub(p3, jv_scale(dn, jv_dot(dn, jv_sub(p3, o))));
  const jvec3 pa = jv_sub(sa, o), pb = jv_sub(sb, o), pc = jv_sub(sc, o);
  const float papb = jv_mixed(dn, pa, pb), pbpc = jv_mixed(dn, pb, pc), pcpa = jv_mixed(dn, pc, pa);
  if (!((papb > 0 && pbpc > 0 && pcpa > 0) || (papb < 0 && pbpc < 0 && pcpa < 0))) return JADE_INF_F;
  const jvec3 eb = jv_sub(sb, sa), ec = jv_sub(sc, sa), q = jv_sub(o, sa);
  const float divider = jade_diffprod(eb.x, ec.y, eb.y, ec.x);
  const float rate_a = jade_diffprod(ec.y, q.x, ec.x, q.y) / divider;
  const float rate_b = jade_fma(eb.x, q.y, (-eb.y) * q.x) / divider;
  const jvec3 P = jv_add(jv_add(p1, jv_scale(jv_sub(p2, p1), rate_a)), jv_scale(jv_sub(p3, p1), rate_b));
  const float distance = jv_dot(jv_sub(P, o), dn);
  return (distance > 0 && distance < JADE_INF_F) ? distance : JADE_INF_F;
}

// a triangle as shading sees it: its flat normal and its object's material (jade_device.h, DevMaterial)
struct ShadeTri {
  jvec3 norm;
  const DevMaterial* m;
};
static __device__ __forceinline__ ShadeTri shade_tri(const DevScene& S, int idx) {
  const float4 v = S.tnorm[idx];
  ShadeTri t;
  t.norm = jv(v.x, v.y, v.z);
  t.m = S.mats + __float_as_uint(v.w);
  return t;
}

static __device__ __forceinline__ bool nonemissive(const DevMaterial* t) {
  return t->emissive[0] < 1.5e-4f && t->emissive[1] < 1.5e-4f && t->emissive[2] < 1.5e-4f;
}

static __device__ __forceinline__ float schlick_out(float R0, float cosine_abs) {  // "R0 - (1-R0)(...)^5", :1102
  float one_cosine_o = 1 - cosine_abs;
  float one_cosine_o_sqr = one_cosine_o * one_cosine_o;
  return R0 - (1 - R0) * one_cosine_o_sqr * one_cosine_o_sqr * one_cosine_o;
}

// gen_refract_ray, PathTrace.cu:876-894
static __device__ jvec3 gen_refract_ray(jvec3 direction_in, jvec3 normal_line, float eta, bool* full_reflex) {
  float cosi = jv_dot(direction_in, normal_line);
  if (cosi > 0) normal_line = jv_neg(normal_line);
  else cosi *= -1;
  float cost2 = 1.0f - eta * eta * (1.0f - cosi * cosi);
  if (cost2 > 0) {
    *full_reflex = false;
    return jv_add(jv_scale(direction_in, eta), jv_scale(normal_line, eta * cosi - jade_sqrt(cost2)));
  }
  *full_reflex = true;
  return direction_in;
}

// One lane's view of its pixel record.
struct Px {
  const PathState& P;
  int p;
  __device__ Px(const PathState& ps, int pix) : P(ps), p(pix) {}
  // the record's ray slots: one float4 per slot, side by side per record; one hit point per record (jade_device.h)
  __device__ float4* slot(int k) const { return P.slot + ((size_t)p * P.nslots + k); }
  __device__ jvec3 dir(int k) const {
    const float4 v = slot(k)[0];
    return jv(v.x, v.y, v.z);
  }
  __device__ void set_dir(int k, jvec3 v) const {
    float* f = reinterpret_cast<float*>(slot(k));
    f[0] = v.x; f[1] = v.y; f[2] = v.z;
  }
  __device__ jvec3 hpt(int k) const {  // (k: the slot of the record's one ray whose nearest hit was wanted - the caller knows which)
    (void)k;
    const float4 v = P.hitp[p];
    return jv(v.x, v.y, v.z);
  }
  __device__ int hit(int k) const { return reinterpret_cast<const int*>(slot(k))[3]; }
  __device__ void set_hit(int k, int v) const { reinterpret_cast<int*>(slot(k))[3] = v; }
  __device__ void set_limit(int k, float v) const { reinterpret_cast<float*>(slot(k))[3] = v; }  // a queued ray that may end early (jade_device.h)
  // origin shared by the record's pending rays + the triangle they leave: one float4
  __device__ void set_origin(jvec3 o, int skip) const { P.orgs[p] = make_float4(o.x, o.y, o.z, __int_as_float(skip)); }
  __device__ jvec3 origin() const {
    const float4 v = P.orgs[p];
    return jv(v.x, v.y, v.z);
  }
  __device__ int skip() const { return reinterpret_cast<const int*>(P.orgs + p)[3]; }
  // aux / auxi of the path (jade_device.h, PathState.aux): a float4 of their own, touched by the BSSRDF and refraction stages only
  __device__ jvec3 aux() const {
    const float4 v = P.aux[p];
    return jv(v.x, v.y, v.z);
  }
  __device__ void set_aux(jvec3 v) const {
    float* f = reinterpret_cast<float*>(P.aux + p);
    f[0] = v.x; f[1] = v.y; f[2] = v.z;
  }
  __device__ int auxi() const { return reinterpret_cast<const int*>(P.aux + p)[3]; }
  __device__ void set_auxi(int v) const { reinterpret_cast<int*>(P.aux + p)[3] = v; }
};

// The same view held in registers: k_light traces a record's single ray (camera or mirror) in the kernel that shades
// it, so the ray and its result never travel through memory.  Slot 0 only.
struct RegPx {
  jvec3 d, o, hp;
  int h, sk;
  __device__ jvec3 dir(int) const { return d; }
  __device__ void set_dir(int, jvec3 v) { d = v; }
  __device__ jvec3 hpt(int) const { return hp; }
  __device__ int hit(int) const { return h; }
  __device__ void set_hit(int, int v) { h = v; }
  __device__ void set_origin(jvec3 org, int skip) {
    o = org;
    sk = skip;
  }
};

struct ShadeCtx {
  uint32_t rng;
  uint32_t stage, depth, flags;
  jvec3 thr, acc, le;
  int32_t obj;
  jvec3 src, out;
  // what this pass emits
  int n_emit_rays;  // number of active slots written
  // counters
  uint32_t c_primary, c_shadow, c_shaded, c_samples;
  uint32_t c_cls;  // rays emitted by call site, one byte each: env | indirect << 8 | mirror << 16 | refract << 24 (a record emits once per pass)
};

// push (dir, rate) — forward form of stack_dir / stack_indir_rate.  Returns
// true when the reference's `while (stack_offset < STACK_CAPACITY)` would stop.
static __device__ __forceinline__ bool path_push(ShadeCtx& c, jvec3 dirv, jvec3 rate) {
  c.acc = jv_add(c.acc, jv_mul(c.thr, dirv));
  c.thr = jv_mul(c.thr, rate);
  c.depth += 1;
  return c.depth >= JADE_STACK_CAPACITY;
}

// The mirror branch of the bounce loop (PathTrace.cu:1365-1405): RR, then the reflected ray.
// Shared by the full shading kernel and the lean one (k_shade<true>).
template <class PX>
static __device__ __forceinline__ bool bounce_mirror(PX& px, ShadeCtx& c, const DevMaterial* ot, jvec3 obj_emissive, jvec3 n,
                                                     jvec3* l_final) {
  const float RR_F = (float)JADE_RR_RATE_D;
  jvec3 obj_hit_fr = jv_scale(V3(ot->brdf), (float)(1.0 / JADE_PI_D));
  int k = ot->refract_mode != JADE_NO_REFRACT ? 2 : 1;
  if (obj_emissive.x > 1.5e-4f || obj_emissive.y > 1.5e-4f || obj_emissive.x > 1.5e-4f) {
    *l_final = jv_scale(jv_mul(obj_emissive, obj_hit_fr), (float)k);
    return false;
  }
  float rr_result = jade_rand(&c.rng);
  if (!(rr_result < RR_F)) return false;
  jvec3 refl = jv_sub(jv_scale(n, 2 * jv_dot(c.out, n)), c.out);
  px.set_origin(c.src, c.obj);
  px.set_dir(0, refl);
  px.set_hit(0, -1);
  c.n_emit_rays++;
  c.stage = ST_MIRROR;
  c.flags = 0;
  return true;
}

// What the lean kernel may shade at a vertex: an emitter (the loop's first test ends the path) or a
// pure mirror (no refraction branch, not diffuse).  Everything else needs the full kernel.
static __device__ __forceinline__ bool lean_can_shade(const DevMaterial* ot) {
  if (ot->emissive[0] > 1.4e-5f || ot->emissive[1] > 1.4e-5f || ot->emissive[2] > 1.4e-5f) return true;
  return ot->refract_mode == JADE_NO_REFRACT && ot->reflex_mode != JADE_DIFFUSE;
}

// begin_bounce restricted to those two cases: the same statements in the same order
// (emissive test :916-920, the select draw :924, then the mirror branch).
template <class PX>
static __device__ bool begin_bounce_lean(const DevScene& S, PX& px, ShadeCtx& c, jvec3* l_final) {
  const ShadeTri ots = shade_tri(S, c.obj);
  const DevMaterial* ot = ots.m;
  c.c_shaded += 1;
  jvec3 obj_emissive = V3(ot->emissive);
  if (obj_emissive.x > 1.4e-5f || obj_emissive.y > 1.4e-5f || obj_emissive.z > 1.4e-5f) {
    *l_final = obj_emissive;
    return false;
  }
  *l_final = jv(0, 0, 0);
  (void)jade_rand(&c.rng);  // select_reflex_refract: drawn for every material, decides nothing for a pure mirror
  return bounce_mirror(px, c, ot, obj_emissive, ots.norm, l_final);
}

// Round 4: the bounce in two parts, so that k_shade can run the second one for records GROUPED BY BRANCH (bounce_classify by the
// thread that holds the record, bounce_branch by whichever thread of the block the record is dealt to; k_shade, jade_hip.hip).
// begin_bounce = the two in a row: the statements, and the order of the random draws, are the ones they always were.
enum { BT_END = 0, BT_MIRROR, BT_DIFFUSE, BT_BSSRDF, BT_REFRACT, BT_N };
// The head of the bounce: the emissive test (PathTrace.cu:916-920; BT_END: *l_final = the emission), then the draws that choose
// the branch (:924-930).  BT_DIFFUSE covers the diffuse and the SSS-diffuse branch: stage and flags are set here.
static __device__ __forceinline__ int bounce_classify(const DevScene& S, ShadeCtx& c, jvec3* l_final) {
  const ShadeTri ots = shade_tri(S, c.obj);
  const DevMaterial* ot = ots.m;
  c.c_shaded += 1;
  jvec3 obj_emissive = V3(ot->emissive);
  if (obj_emissive.x > 1.4e-5f || obj_emissive.y > 1.4e-5f || obj_emissive.z > 1.4e-5f) {
    *l_final = obj_emissive;  // PathTrace.cu:916-920
    return BT_END;
  }
  *l_final = jv(0, 0, 0);
  float select_reflex_refract = jade_rand(&c.rng);
  if (select_reflex_refract < 0.5f && ot->refract_mode != JADE_NO_REFRACT) {
    if (ot->refract_mode == JADE_SUB_SURFACE) {
      select_reflex_refract = jade_rand(&c.rng);
      if (select_reflex_refract < (float)JADE_SSS_RATE_D) {
        c.stage = ST_DIFFUSE;
        c.flags = STF_SSS;
        return BT_DIFFUSE;
      }
      return BT_BSSRDF;
    }
    return BT_REFRACT;
  }
  if (ot->reflex_mode == JADE_DIFFUSE) {
    c.stage = ST_DIFFUSE;
    c.flags = 0;
    return BT_DIFFUSE;
  }
  return BT_MIRROR;
}
// The branch itself: its draws, its rays (written through px), stage / flags / n_emit_rays.  Returns false if the path ended at
// this vertex (*l_final holds the last l_dir).  `type` is uniform over the waves k_shade runs it for.
// ENVIS: jade_render_params.env_sampling = JADE_ENV_IMPORTANCE (compiled apart: the parity kernels carry none of its code)
template <bool ENVIS = false>
static __device__ __forceinline__ bool bounce_branch(int type, const DevScene& S, const Px& px, ShadeCtx& c, jvec3* l_final) {
  const jade_triangle* T = S.tris;  // vertices only
  const ShadeTri ots = shade_tri(S, c.obj);
  const DevMaterial* ot = ots.m;
  const int nE = S.n_emit;
  const float RR_F = (float)JADE_RR_RATE_D;
  const jvec3 n = ots.norm;
  *l_final = jv(0, 0, 0);  // (as bounce_classify left it: a worker thread of k_shade brings its own)
  if (type == BT_DIFFUSE) goto diffuse_like;
  if (type == BT_MIRROR) return bounce_mirror(px, c, ot, V3(ot->emissive), n, l_final);  // ---- mirror, PathTrace.cu:1365-1405 ----
  if (type == BT_BSSRDF) {
    {
      // ---- BSSRDF, PathTrace.cu:1029-1178 ----
      const jade_obj_seg seg = S.segs[ot->obj_idx];
      const float u_area = jade_rand(&c.rng);
      float random_idx = u_area * S.prefix[seg.end_idx];
      int left = seg.begin_idx, right = seg.end_idx, middle = 0;
      const uint2 gd = S.guide_obj[ot->obj_idx];
      if (gd.y != 0u) {
        // The reference bisects prefix[] (:1031-1048): ~17 DEPENDENT loads, the longest latency chain of this kernel, and what
        // it returns is not the boundary but the LAST midpoint it looked at.  Both are reproduced without the chain: the
        // boundary b = the first i with random_idx <= prefix[i] comes from a guide table (jade_scene_create: prefix is
        // checked to be finite and non-decreasing, so "random_idx <= prefix[mid]" is "mid >= b"; the cell's bounds hold
        // because rounding is monotone: c / Gn <= u implies fl(c / Gn * A) <= fl(u * A)), then the bisection is replayed
        // on indices alone.
        const uint32_t cell = (uint32_t)(u_area * (float)gd.y);  // u in [0, 1], Gn a power of two: exact
        const uint32_t* gp = S.guide + gd.x + cell;
        const uint32_t b_hi = gp[1];
        uint32_t b = gp[0];
        while (b < b_hi && !(random_idx <= S.prefix[b])) ++b;
        while (left < right - 1) {
          middle = (left + right) / 2;
          if ((uint32_t)middle >= b) right = middle;
          else left = middle;
        }
      } else {
        while (left < right - 1) {
          middle = (left + right) / 2;
          float pm = S.prefix[middle];
          if (random_idx <= pm) right = middle;
          else if (random_idx >= pm) left = middle;
          else break;
        }
      }
      middle = S.mapping[middle];
      float rand_x = jade_rand(&c.rng);
      float rand_y = jade_rand(&c.rng);
      if (rand_x + rand_y > 1) {
        rand_x = 1 - rand_x;
        rand_y = 1 - rand_y;
      }
      const ShadeTri t_is = shade_tri(S, middle);
      const DevMaterial* t_i = t_is.m;
      const jvec3 t_norm = t_is.norm;
      const jvec3 rate = V3(t_i->refract_rate);
      jvec3 random_point = tri_point(&T[middle], rand_x, rand_y);
      jvec3 inner_direction = jv_sub(random_point, c.src);
      float inner_distance = jade_sqrt(jv_dot(inner_direction, inner_direction));
      float neg_d = -1.0f * inner_distance;
      float neg_d3 = (float)((double)neg_d / 3.0);
      const float E_F = (float)JADE_E_D;
      jvec3 e1 = jv(jade_powf(E_F, neg_d / rate.x), jade_powf(E_F, neg_d / rate.y), jade_powf(E_F, neg_d / rate.z));
      jvec3 e2 = jv(jade_powf(E_F, neg_d3 / rate.x), jade_powf(E_F, neg_d3 / rate.y), jade_powf(E_F, neg_d3 / rate.z));
      jvec3 bssrdf = jv_div(jv_add(e1, e2), jv_scale(rate, (float)(8 * JADE_PI_D * (double)inner_distance)));
      float eta = t_i->refract_index;
      float R0 = (eta - 1) / (eta + 1) * (eta - 1) / (eta + 1);
      float one_cosine_i = 1 - jade_fabs(jv_dot(n, c.out));
      float one_cosine_i_sqr = one_cosine_i * one_cosine_i;
      float fresnel_rate_i = R0 + (1 - R0) * one_cosine_i_sqr * one_cosine_i_sqr * one_cosine_i;
      bssrdf = jv_scale(bssrdf, fresnel_rate_i);

      px.set_origin(random_point, middle);
      px.set_aux(bssrdf);
      for (int i = 0; i < nE; ++i) {
        float rx = jade_rand(&c.rng);
        float ry = jade_rand(&c.rng);
        if (rx + ry > 1) {
          rx = 1 - rx;
          ry = 1 - ry;
        }
        const jade_triangle* et = &T[S.emit[i]];
        jvec3 random_emit_point = tri_point(et, rx, ry);
        const jvec3 sd = jv_sub(random_emit_point, random_point);
        px.set_dir(i, sd);
        px.set_limit(i, shadow_limit(et, random_point, sd));
        c.n_emit_rays++;
      }
      uint32_t noenv = 0;
      if (ENVIS) {  // (non-parity mode: by importance; a direction on the wrong side contributes nothing and is not traced)
        float ratio;
        const jvec3 ray_direction = env_sample(S, &c.rng, &ratio);
        if (jv_dot(ray_direction, t_norm) * jv_dot(inner_direction, t_norm) < 0) {
          px.set_hit(nE, -2);
          noenv = STF_NOENV;
        } else {
          px.set_dir(nE, ray_direction);
          px.set_limit(nE, JADE_INF_F);
          px.set_auxi(__float_as_int(ratio));
          c.n_emit_rays++;
        }
      } else {
        jvec3 ray_direction = sphere_dir(&c.rng);
        if (jv_dot(ray_direction, t_norm) * jv_dot(inner_direction, t_norm) < 0) ray_direction = jv_neg(ray_direction);
        px.set_dir(nE, ray_direction);
        px.set_limit(nE, JADE_INF_F);  // environment visibility: any recorded hit
        c.n_emit_rays++;
      }
      jvec3 ray_direction = sphere_dir(&c.rng);
      if (jv_dot(ray_direction, t_norm) * jv_dot(inner_direction, t_norm) > 0) ray_direction = jv_neg(ray_direction);
      float rr_result = jade_rand(&c.rng);
      c.stage = ST_BSSRDF;
      c.flags = noenv;
      if (rr_result < RR_F) {
        c.flags |= STF_RR;
        px.set_dir(nE + 1, ray_direction);
        px.set_hit(nE + 1, -1);
        c.n_emit_rays++;
      } else {
        px.set_hit(nE + 1, -2);
      }
      return true;
    }
  }
  // ---- direct refraction entry, PathTrace.cu:1180-1199 ----
  {
    {
      float triangle_miu = ot->refract_index;
      float R0 = (1 - triangle_miu) / (1 + triangle_miu) * (1 - triangle_miu) / (1 + triangle_miu);
      float one_cosine_i = 1 - jade_fabs(jv_dot(n, c.out));
      float one_cosine_i_sqr = one_cosine_i * one_cosine_i;
      float fresnel_rate_i = R0 + (1 - R0) * one_cosine_i_sqr * one_cosine_i_sqr * one_cosine_i;
      bool full_reflex = false;
      jvec3 rev_out_direction = jv_scale(c.out, -1.0f);
      jvec3 refract_ray = gen_refract_ray(rev_out_direction, n, (float)(1.0 / (double)triangle_miu), &full_reflex);
      px.set_aux(jv(1 - fresnel_rate_i, 1 - fresnel_rate_i, 1 - fresnel_rate_i));
      px.set_auxi(0);
      px.set_origin(c.src, c.obj);
      px.set_dir(0, refract_ray);
      px.set_hit(0, -1);
      c.n_emit_rays++;
      c.stage = ST_REFRACT_LOOP;
      c.flags = 0;
      return true;
    }
  }

diffuse_like:
  // ---- diffuse (:1266-1364) and SSS-diffuse (:931-1028): same ray set ----
  {
    px.set_origin(c.src, c.obj);
    const float side = jv_dot(c.out, n);
    for (int i = 0; i < nE; ++i) {
      float rand_x = jade_rand(&c.rng);
      float rand_y = jade_rand(&c.rng);
      if (rand_x + rand_y > 1) {
        rand_x = 1 - rand_x;
        rand_y = 1 - rand_y;
      }
      const jade_triangle* et = &T[S.emit[i]];
      jvec3 random_point = tri_point(et, rand_x, rand_y);
      jvec3 obj_light_direction = jv_sub(random_point, c.src);
      px.set_dir(i, obj_light_direction);
      if (jv_dot(obj_light_direction, n) * side < 0) {
        px.set_hit(i, -2);  // `continue`: no shadow ray
      } else {
        px.set_limit(i, shadow_limit(et, c.src, obj_light_direction));
        c.n_emit_rays++;
      }
    }
    if (ENVIS) {  // (non-parity mode, as in the BSSRDF branch)
      float ratio;
      const jvec3 ray_direction = env_sample(S, &c.rng, &ratio);
      if (jv_dot(ray_direction, n) * side < 0) {
        px.set_hit(nE, -2);
        c.flags |= STF_NOENV;
      } else {
        px.set_dir(nE, ray_direction);
        px.set_limit(nE, JADE_INF_F);
        px.set_auxi(__float_as_int(ratio));
        c.n_emit_rays++;
      }
    } else {
      jvec3 ray_direction = sphere_dir(&c.rng);
      if (jv_dot(ray_direction, n) * side < 0) ray_direction = jv_neg(ray_direction);
      px.set_dir(nE, ray_direction);
      px.set_limit(nE, JADE_INF_F);  // environment visibility: any recorded hit
      c.n_emit_rays++;
    }
    float rr_result = jade_rand(&c.rng);
    if (rr_result < RR_F) {
      jvec3 ray_direction = sphere_dir(&c.rng);
      if (jv_dot(ray_direction, n) * side < 0) ray_direction = jv_neg(ray_direction);
      c.flags |= STF_RR;
      px.set_dir(nE + 1, ray_direction);
      px.set_hit(nE + 1, -1);
      c.n_emit_rays++;
    } else {
      px.set_hit(nE + 1, -2);
    }
    return true;
  }
}

// Sample the bounce at the current vertex and emit its rays.  Returns false
// if the path ended at this vertex (l_final holds the last l_dir).
template <bool ENVIS = false>
static __device__ bool begin_bounce(const DevScene& S, const Px& px, ShadeCtx& c, jvec3* l_final) {
  const int type = bounce_classify(S, c, l_final);
  if (type == BT_END) return false;
  return bounce_branch<ENVIS>(type, S, px, c, l_final);
}

// Outcome of folding the pending rays' results into the path.
enum { CONSUME_VERTEX = 0, CONSUME_END = 1, CONSUME_ZERO = 2, CONSUME_EMITTED = 3 };

// Result of the mirror ray (PathTrace.cu:1383-1398).  Shared by both shading kernels.
template <class PX>
static __device__ __forceinline__ int consume_mirror(const DevScene& S, const PX& px, ShadeCtx& c, jvec3* l_final) {
  const DevMaterial* ot = shade_tri(S, c.obj).m;
  const int k = ot->refract_mode != JADE_NO_REFRACT ? 2 : 1;
  const jvec3 obj_hit_fr = jv_scale(V3(ot->brdf), (float)(1.0 / JADE_PI_D));
  float kk = (float)(k / (JADE_RR_RATE_D / JADE_PI_D));
  jvec3 refl = px.dir(0);
  int nh = px.hit(0);
  if (nh >= 0) {
    c.out = jv_neg(refl);
    c.src = px.hpt(0);
    c.obj = nh;
    *l_final = jv(0, 0, 0);
    return path_push(c, jv(0, 0, 0), jv_scale(obj_hit_fr, kk)) ? CONSUME_END : CONSUME_VERTEX;
  }
  *l_final = jv_scale(jv_mul(sample_hdr(S, refl), obj_hit_fr), kk);
  return CONSUME_END;
}

// Fold the results of the rays issued last pass.  CONSUME_VERTEX: the path
// moved to a new vertex (c.obj/src/out updated); CONSUME_END: it ended with
// *l_final; CONSUME_ZERO: pathTracing returned 0 (:1231); CONSUME_EMITTED: the
// refraction loop issued its next ray.
template <bool ENVIS = false>
static __device__ int consume(const DevScene& S, const Px& px, ShadeCtx& c, jvec3* l_final) {
  const jade_triangle* T = S.tris;  // vertices only
  const int nE = S.n_emit;
  const float PI_F = (float)JADE_PI_D;
  const float RR_F = (float)JADE_RR_RATE_D;
  const ShadeTri ots = shade_tri(S, c.obj);
  const DevMaterial* ot = ots.m;
  const jvec3 n = ots.norm;
  const int k = ot->refract_mode != JADE_NO_REFRACT ? 2 : 1;
  const jvec3 obj_hit_fr = jv_scale(V3(ot->brdf), (float)(1.0 / JADE_PI_D));
  jvec3 l_dir = jv(0, 0, 0);

  if (c.stage == ST_DIFFUSE) {
    const bool sss = (c.flags & STF_SSS) != 0;
    const jvec3 f = sss ? jv_scale(V3(ot->refract_albedo), (float)(1.0 / JADE_PI_D)) : obj_hit_fr;
    for (int i = 0; i < nE; ++i) {
      int h = px.hit(i);
      int emit_tri_idx = S.emit[i];
      if (h >= 0 && h == emit_tri_idx) {
        jvec3 ld = px.dir(i);
        const ShadeTri t_i = shade_tri(S, emit_tri_idx);
        float dls = jv_dot(ld, ld);
        jvec3 w = jv_mul(V3(t_i.m->emissive), f);
        w = jv_scale(w, jade_fabs(jv_dot(n, ld) * jv_dot(t_i.norm, ld)));
        w = jv_divs(jv_divs(w, dls), dls);
        w = jv_scale(w, tri_size(&T[emit_tri_idx]));
        l_dir = jv_add(l_dir, w);
      }
    }
    if (ENVIS ? px.hit(nE) == -1 : px.hit(nE) < 0) {  // (-1: traced and nothing hit; -2: no environment ray this bounce - env_sampling only)
      jvec3 rd = px.dir(nE);
      jvec3 w = jv_mul(sample_hdr(S, rd), f);
      w = jv_scale(w, jade_fabs(jv_dot(n, rd)));
      w = jv_scale(jv_scale(w, 2.0f), PI_F);
      if (ENVIS) w = jv_scale(w, __int_as_float(px.auxi()));  // x (1 / 2 pi) / pdf: the reference's weight is 1 / its own pdf
      l_dir = jv_add(l_dir, w);
    }
    l_dir = jv_scale(l_dir, sss ? (float)(k / JADE_SSS_RATE_D) : (float)k);
    *l_final = l_dir;
    if (!(c.flags & STF_RR)) return CONSUME_END;
    int nh = px.hit(nE + 1);
    if (nh >= 0 && nonemissive(shade_tri(S, nh).m)) {
      jvec3 rd = jv_neg(px.dir(nE + 1));
      jvec3 indir_rate = jv_divs(jv_scale(obj_hit_fr, jade_fabs(jv_dot(rd, n))), RR_F);
      jvec3 rate = sss ? jv_divs(jv_scale(indir_rate, (float)k), (float)JADE_SSS_RATE_D) : jv_scale(indir_rate, (float)k);
      c.src = px.hpt(nE + 1);
      c.out = rd;
      c.obj = nh;
      return path_push(c, l_dir, rate) ? CONSUME_END : CONSUME_VERTEX;
    }
    return CONSUME_END;
  }

  if (c.stage == ST_BSSRDF) {
    const int middle = px.skip();
    const ShadeTri t_is = shade_tri(S, middle);
    const DevMaterial* t_i = t_is.m;
    const jvec3 t_norm = t_is.norm;
    const jvec3 bssrdf = px.aux();
    const float eta = t_i->refract_index;
    const float R0 = (eta - 1) / (eta + 1) * (eta - 1) / (eta + 1);
    const float area_total = S.prefix[S.segs[t_i->obj_idx].end_idx];
    for (int i = 0; i < nE; ++i) {
      int h = px.hit(i);
      int emit_tri_idx = S.emit[i];
      if (h >= 0 && h == emit_tri_idx) {
        jvec3 ld = px.dir(i);
        const ShadeTri emit_i = shade_tri(S, emit_tri_idx);
        float fresnel_rate_o = schlick_out(R0, jade_fabs(jv_dot(jv_normalize(ld), t_norm)));
        float dls = jv_dot(ld, ld);
        jvec3 w = jv_scale(V3(emit_i.m->emissive), fresnel_rate_o);
        w = jv_mul(w, bssrdf);
        w = jv_scale(w, jade_fabs(jv_dot(t_norm, ld) * jv_dot(emit_i.norm, ld)));
        w = jv_divs(jv_divs(w, dls), dls);
        w = jv_scale(w, tri_size(&T[emit_tri_idx]));
        w = jv_divs(w, PI_F);
        w = jv_scale(w, area_total);
        l_dir = jv_add(l_dir, w);
      }
    }
    if (ENVIS ? px.hit(nE) == -1 : px.hit(nE) < 0) {
      jvec3 rd = px.dir(nE);
      float fresnel_rate_o = schlick_out(R0, jade_fabs(jv_dot(rd, t_norm)));
      jvec3 w = jv_mul(jv_scale(sample_hdr(S, rd), fresnel_rate_o), bssrdf);
      w = jv_scale(jv_scale(w, jade_fabs(jv_dot(t_norm, rd))), 2.0f);
      if (ENVIS) w = jv_scale(w, __int_as_float(px.auxi()));
      l_dir = jv_add(l_dir, w);
    }
    l_dir = jv_scale(l_dir, (float)(k / (1 - JADE_SSS_RATE_D)));
    *l_final = l_dir;
    if (!(c.flags & STF_RR)) return CONSUME_END;
    int nh = px.hit(nE + 1);
    if (nh >= 0 && nonemissive(shade_tri(S, nh).m)) {
      jvec3 rd = jv_neg(px.dir(nE + 1));
      float fresnel_rate_o = schlick_out(R0, jade_fabs(jv_dot(rd, t_norm)));
      jvec3 indir_rate = jv_scale(bssrdf, fresnel_rate_o);
      indir_rate = jv_scale(indir_rate, jade_fabs(jv_dot(rd, t_norm)));
      indir_rate = jv_scale(indir_rate, area_total);
      indir_rate = jv_divs(jv_scale(indir_rate, 2.0f), RR_F);
      jvec3 rate = jv_divs(jv_scale(indir_rate, (float)k), (float)(1 - JADE_SSS_RATE_D));
      c.src = px.hpt(nE + 1);
      c.out = rd;
      c.obj = nh;
      return path_push(c, l_dir, rate) ? CONSUME_END : CONSUME_VERTEX;
    }
    return CONSUME_END;
  }

  if (c.stage == ST_MIRROR) return consume_mirror(S, px, c, l_final);

  if (c.stage == ST_REFRACT_LOOP) {
    // one iteration of the for loop at PathTrace.cu:1201-1234
    const float triangle_miu = ot->refract_index;
    const float R0 = (1 - triangle_miu) / (1 + triangle_miu) * (1 - triangle_miu) / (1 + triangle_miu);
    int nh = px.hit(0);
    if (nh < 0) return CONSUME_ZERO;  // "obj surface is not close": return vec3(0)
    const ShadeTri hts = shade_tri(S, nh);
    const DevMaterial* ht = hts.m;
    const jvec3 hn = hts.norm;
    jvec3 refract_ray = px.dir(0);
    jvec3 start = px.origin();
    jvec3 hp = px.hpt(0);
    jvec3 l_indir_rate = px.aux();
    bool full_reflex = false;
    refract_ray = gen_refract_ray(refract_ray, hn, triangle_miu, &full_reflex);
    jvec3 distance = jv_sub(start, hp);
    float dist = jade_sqrt(jv_dot(distance, distance));
    l_indir_rate = jv_mul(l_indir_rate, jv(jade_powf(ht->refract_rate[0], dist), jade_powf(ht->refract_rate[1], dist),
                                           jade_powf(ht->refract_rate[2], dist)));
    float fresnel_rate_o = schlick_out(R0, jade_fabs(jv_dot(refract_ray, hn)));
    float reflex_refract_select = jade_rand(&c.rng);
    int it = px.auxi() + 1;
    bool leave = true;
    if (full_reflex || reflex_refract_select < 0.2f) {
      refract_ray = jv_sub(refract_ray, jv_scale(hn, 2 * jv_dot(refract_ray, hn)));
      if (!full_reflex) l_indir_rate = jv_scale(l_indir_rate, fresnel_rate_o * 5);
      leave = it >= JADE_MAX_FULL_REFLEX_TIME;  // loop exhausted: fall through to the RR test
    } else {
      l_indir_rate = jv_scale(l_indir_rate, (float)((1.0 - (double)fresnel_rate_o) * 1.25));
    }
    px.set_origin(hp, nh);
    px.set_aux(l_indir_rate);
    px.set_auxi(it);
    px.set_dir(0, refract_ray);
    px.set_hit(0, -1);
    *l_final = jv(0, 0, 0);
    if (!leave) {
      c.n_emit_rays++;
      return CONSUME_EMITTED;
    }
    float rr_result = jade_rand(&c.rng);
    if (!(rr_result < RR_F)) return CONSUME_END;
    c.stage = ST_REFRACT_EXIT;
    c.n_emit_rays++;
    return CONSUME_EMITTED;
  }

  // ST_REFRACT_EXIT, PathTrace.cu:1239-1257
  {
    jvec3 refract_ray = px.dir(0);
    jvec3 l_indir_rate = px.aux();
    float kk = (float)(k / JADE_RR_RATE_D);
    int nh = px.hit(0);
    if (nh >= 0) {
      c.out = jv_scale(refract_ray, -1.0f);
      c.src = px.hpt(0);
      c.obj = nh;
      *l_final = jv(0, 0, 0);
      return path_push(c, jv(0, 0, 0), jv_scale(l_indir_rate, kk)) ? CONSUME_END : CONSUME_VERTEX;
    }
    *l_final = jv_scale(jv_mul(sample_hdr(S, refract_ray), l_indir_rate), kk);
    return CONSUME_END;
  }
}

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
//   - a lane walks one ray, rays taken from a compacted queue by persistent workgroups (k_trace: a wave claims a chunk
//     with one atomic and refills its idle lanes from it, so short rays do not leave lanes idle behind long ones) or
//     from the lane's own path (k_light);
//   - the walk never stops at a leaf: the leaves a ray meets are queued and their triangles tested as a second stream of
//     work - by the same lane from a small FIFO (k_light: RayState), or by ANY lane of the wave from a ring in LDS, with
//     hit candidates resolved 64 at a time (k_trace: WalkState / WaveTrace at the end of this file);
//   - the near child is followed directly and only the far child is pushed, which visits nodes in exactly the
//     reference's order with half the stack traffic;
//   - a lane's column of LDS words, word[k][lane] (bank-conflict free), holds the stack (8 levels cover 99.6 % of the
//     rays of the benchmark scenes, deeper levels spill to a per-lane global area, JADE_BVH_STACK_CAPACITY entries in
//     total) and the parts of the ray state that the triangle test does not read (1/dir, the best hit);
//   - both children's boxes come from the parent's 64-B record, interleaved so that the two slab tests run as packed
//     fp32 (v_pk_*_f32), and triangles are tested two at a time from 80-B pair records, packed across the two;
//   - 1/dir and normalize(dir), which the reference recomputes per node and per triangle (:710, :759), are computed
//     once per ray - same values.
#pragma once
#include "jade_device.h"

struct TraceHit {
  int32_t index;  // -1 = miss
  float dist;
  jvec3 point;
};

// Development ablations (cdna_hip_programming.md rule 17): JADE_ABLATE_* repeat a
// piece of work without changing any result, to price that piece.  Off in product builds.
#ifndef JADE_ABLATE_SLAB
#define JADE_ABLATE_SLAB 0
#endif
#ifndef JADE_ABLATE_LOAD
#define JADE_ABLATE_LOAD 0
#endif

// Development profile (where inside a unit of work do a wave's clocks go?): -DJADE_TRACE_PROFILE=1 brackets the
// pieces of k_trace's loop with s_memtime and FORCES the wait for a piece's loads inside that piece (s_waitcnt), so that
// every lap is "issue ... data arrived".  That serialises what the product build may overlap and costs ~10 % (the guide's
// figure for stamps): the laps are a breakdown, not a timing.  Results: tools/trace_profile.py -> profiles/.  Off = no code.
#ifndef JADE_TRACE_PROFILE
#define JADE_TRACE_PROFILE 0
#endif
enum {
  PL_TOP = 0,      // loop top: has my ray ended (LDS), ballot of idle lanes
  PL_WRITEBACK,    // results of ended rays -> memory (walk_result re-reads the winning record)
  PL_REFILL,       // claim queue entries, load the rays, walk_begin
  PL_PICK,         // iterate(): the ballots that choose the kind of work
  PL_WALK_LOAD,    // walk unit: LDS tree-top read + node record gather, until the data is in registers
  PL_WALK_MATH,    // walk unit: slab tests, decisions, stack push / pop (LDS)
  PL_WALK_RING,    // walk unit: leaves met -> the wave's ring
  PL_TEST_POP,     // test unit: lanes without an item take the next ones from the ring
  PL_TEST_LOAD,    // test unit: pair record gather, until the data is in registers
  PL_TEST_RAY,     // test unit: the owner's ray through ds_bpermute
  PL_TEST_MATH,    // test unit: the packed inside test, finished-leaf counter
  PL_TEST_CAND,    // test unit: candidates -> second ring
  PL_RESOLVE,      // resolve_pass: barycentric solve of <= 64 candidates, best-hit update
  PL_N_LAPS,
  PC_ITER = PL_N_LAPS, PC_WALK_UNITS, PC_TEST_UNITS, PC_RESOLVES, PC_REFILLS, PC_WAVES, PC_WALK_LANES, PC_TEST_LANES,
  PC_WALK_WAIT,    // lanes of walk units that hold a ray whose walk has ended (it waits for its leaves' tests)
  PC_WALK_FREE,    // lanes of walk units that hold no ray
  PC_TEST_FULL,    // test units with 56 or more lanes
  PC_TEST_THIN,    // test units with 16 or fewer lanes
  PC_TEST_THIN_LANES,  // ... and their lanes
  PL_N
};
// ... and of k_light_packet's (the same build; tools/trace_profile.py --packet): laps of a wave's loop over its records, counts of what the packets did
enum {
  PKL_ADVANCE = 0,  // every lane without a ray advances until it has one: sample bookkeeping, camera ray, the mirror bounce
  PKL_PREP,         // packet_trace: 1 / d, normalize(d), the first ballot
  PKL_NODE,         // node visits: the record's scalar load, two slab tests, ballots, push
  PKL_LEAF,         // pair records: scalar load, the packed inside test
  PKL_SOLVE,        // ... and the barycentric solve for the lanes whose origin projects into a triangle
  PKL_POP,          // the next deferred child off the wave's stack
  PKL_FOLD,         // the packet's result folded in: the sky (sample_hdr), consume_mirror, a new vertex
  PKL_STORE,        // header / context stores, the hand-over list
  PKL_N_LAPS,
  PKC_PACKETS = PKL_N_LAPS, PKC_GIVEN_UP, PKC_NODES, PKC_PAIRS, PKC_SOLVES, PKC_LANES, PKC_NODE_LANES, PKC_PAIR_LANES, PKC_SOLVE_LANES,
  PKL_N
};
#if JADE_TRACE_PROFILE
typedef __attribute__((address_space(3))) unsigned long long jade_prof_lds_u64;
// The laps live in LDS (one row of PL_N 64-bit words per wave, added to by lane 0): as per-lane variables they were 40
// more registers and pushed the kernel into scratch, whose traffic then sat inside every lap (first version, round 3).
// `last` stays a scalar: laps are only taken where the wave's control flow is uniform.
struct TraceProf {
  unsigned long long last;
  uint32_t row;  // LDS byte address of this wave's row
  bool lead;     // lane 0
  __device__ __forceinline__ void begin(uint32_t row_addr, int lane) {
    row = row_addr;
    lead = lane == 0;
    if (lane < PL_N) *(jade_prof_lds_u64*)(__SIZE_TYPE__)(row + 8u * (uint32_t)lane) = 0ull;
    last = __builtin_amdgcn_s_memtime();
  }
  __device__ __forceinline__ void add(int i, unsigned long long n) const {
    if (lead) __hip_atomic_fetch_add((jade_prof_lds_u64*)(__SIZE_TYPE__)(row + 8u * (uint32_t)i), n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
  __device__ __forceinline__ void lap(int i) {
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    add(i, now - last);
    last = now;
  }
  __device__ __forceinline__ void count(int i, unsigned long long n = 1) const { add(i, n); }
  __device__ __forceinline__ unsigned long long get(int i) const { return *(jade_prof_lds_u64*)(__SIZE_TYPE__)(row + 8u * (uint32_t)i); }
  static __device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
};
#define PROF_LAP(pr, i) (pr).lap(i)
#define PROF_COUNT(pr, i, n) (pr).count(i, n)
#define PROF_DRAIN() TraceProf::drain()
#else
struct TraceProf {};
#define PROF_LAP(pr, i) ((void)0)
#define PROF_COUNT(pr, i, n) ((void)0)
#define PROF_DRAIN() ((void)0)
#endif

// A lane's column of LDS words, word k of lane t at byte  k * JADE_COL_STRIDE + 4 t  of the block's column array
// (bank-conflict free): words [0, JADE_LDS_STACK) are the traversal stack, then the leaf FIFO, then the JADE_LDS_STATE
// words that hold the parts of the ray state the triangle test does not read (see RayState).
//
// What a unit of work costs is the instructions a wave issues for it (round 3 profile, DESIGN.md 3.4: a unit is ~150 instructions
// and ~2,700 clocks, a third of them waiting for its one record), and a third of the node step was address arithmetic for these words (index -> * 256
// -> select -> * 4 + base, per access).  So positions are kept AS LDS byte addresses: the stack pointer is the address
// of the next free level, the FIFO counters advance in units of one column word, and the column array is 4 KB-aligned
// so that a FIFO slot's address is one v_and_or of the counter and the FIFO's base.
#define JADE_COL_STRIDE (4u * JADE_TRACE_BLOCK) /* bytes between two words of a lane's column */
typedef __attribute__((address_space(3))) uint32_t jade_lds_u32;
static __device__ __forceinline__ uint32_t lds_addr_of(const uint32_t* p) { return (uint32_t)(__SIZE_TYPE__)(jade_lds_u32*)p; }
static __device__ __forceinline__ void lds_st(uint32_t addr, uint32_t v) { *(jade_lds_u32*)(__SIZE_TYPE__)addr = v; }
static __device__ __forceinline__ uint32_t lds_ld(uint32_t addr) { return *(jade_lds_u32*)(__SIZE_TYPE__)addr; }
typedef __attribute__((address_space(3))) unsigned long long jade_lds_u64;
static __device__ __forceinline__ void lds_st64(uint32_t addr, uint32_t lo, uint32_t hi) {
  *(jade_lds_u64*)(__SIZE_TYPE__)addr = (unsigned long long)lo | ((unsigned long long)hi << 32);
}
static __device__ __forceinline__ void lds_ld64(uint32_t addr, uint32_t& lo, uint32_t& hi) {
  const unsigned long long v = *(jade_lds_u64*)(__SIZE_TYPE__)addr;
  lo = (uint32_t)v;
  hi = (uint32_t)(v >> 32);
}

struct LdsStack {
  uint32_t* lds;       // this lane's column (word k at lds[k * JADE_TRACE_BLOCK])
  uint32_t col;        // the same as an LDS byte address
  uint32_t* spill;     // global, spill[(level - JADE_LDS_STACK) * stride + gtid]
  uint32_t stride_spill;
  const float4* top;   // LDS copy of node records [0, top_k): four planes of float4 (plane j = the record's j-th 16 bytes)
  uint32_t top_k;
  const float4* top4;  // wide walk: LDS copy of the wide records [0, top4_k), seven planes of JADE_TRACE_TOP4 float4
  uint32_t top4_k;
};
enum {
  LW_FIFO = JADE_LDS_STACK,  // JADE_LDS_FIFO leaf cursors waiting for their triangle tests (ring)
  LW_INVX = JADE_LDS_STACK + JADE_LDS_FIFO, LW_INVY, LW_INVZ, LW_BEST_DIST, LW_BEST_INDEX, LW_PX, LW_PY, LW_PZ,
  LW_DUMMY,  // where a lane's store goes when the statement it belongs to does not apply to that lane (straight-line steps below)
  LW_END
};
static_assert(LW_END - LW_INVX == JADE_LDS_STATE, "JADE_LDS_STATE must count the LW_* state words");
static_assert((JADE_LDS_FIFO & (JADE_LDS_FIFO - 1)) == 0, "JADE_LDS_FIFO must be a power of two");
static_assert(JADE_LDS_STACK % JADE_LDS_FIFO == 0, "the FIFO's first word must be a multiple of its size (slot address = base | counter bits)");
#define JADE_FIFO_MASK ((JADE_LDS_FIFO - 1u) * JADE_COL_STRIDE)
#define JADE_COLS_ALIGN (JADE_LDS_FIFO * JADE_COL_STRIDE) /* alignment of the block's column array */

static __device__ __forceinline__ void lds_put(const LdsStack& s, int word, uint32_t v) { lds_st(s.col + (uint32_t)word * JADE_COL_STRIDE, v); }
static __device__ __forceinline__ uint32_t lds_get(const LdsStack& s, int word) { return lds_ld(s.col + (uint32_t)word * JADE_COL_STRIDE); }
static __device__ __forceinline__ void lds_putf(const LdsStack& s, int word, float v) { lds_put(s, word, jade_f2u(v)); }
static __device__ __forceinline__ float lds_getf(const LdsStack& s, int word) { return jade_u2f(lds_get(s, word)); }

// Two lanes of packed fp32 (v_pk_add/mul/fma_f32: full rate, IEEE per component, so the
// values are those of the scalar statements).  A unit's time follows its instruction count; the left/right slab
// tests and the projections of p1/p2 are the same statements on two operands.
typedef float f2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ f2 f2s(float s) { return f2{s, s}; }
// The ray's origin and unit direction travel as three aligned register pairs, (o.x, o.y) (o.z, dn.x)
// (dn.y, dn.z): a packed instruction broadcasts either half of a pair through op_sel, so no
// component ever has to be copied next to itself.
struct RayOD {
  f2 a, b, c;
};
#define OD_OX(q) __builtin_shufflevector((q).a, (q).a, 0, 0)
#define OD_OY(q) __builtin_shufflevector((q).a, (q).a, 1, 1)
#define OD_OZ(q) __builtin_shufflevector((q).b, (q).b, 0, 0)
#define OD_DX(q) __builtin_shufflevector((q).b, (q).b, 1, 1)
#define OD_DY(q) __builtin_shufflevector((q).c, (q).c, 0, 0)
#define OD_DZ(q) __builtin_shufflevector((q).c, (q).c, 1, 1)
static __device__ __forceinline__ jvec3 od_o(const RayOD& q) { return jv(q.a.x, q.a.y, q.b.x); }
static __device__ __forceinline__ jvec3 od_dn(const RayOD& q) { return jv(q.b.y, q.c.x, q.c.y); }
static __device__ __forceinline__ f2 f2fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// hitAABB, PathTrace.cu:758-771, with 1/dir hoisted: the reduction of one box from its six
// slab values.  `exact` selects the NaN-faithful ternary form; it is only needed when a
// component of 1/dir or of the origin is not finite (0 * inf is the one way a NaN can appear
// for a finite scene), otherwise v_min/v_max give bit-identical slab values.
// What the walk needs of the returned value r = (t1 >= t0) ? ((t0 > 0) ? t0 : t1) : -1 (:769-770) is "r > 0" - the box is met - and, where
// two boxes are met, which r is smaller.  r > 0  <=>  t1 >= t0 and t1 > 0: with t1 >= t0 (neither is a NaN then), t0 > 0 gives r = t0 > 0 and
// t1 >= t0 > 0; t0 <= 0 gives r = t1.  So a box costs two comparisons (and a scalar `and`), and r itself - a comparison and a select more -
// is formed only by the binary unit, for the order of two boxes both met (slab_dist).  Everything issues at one per 4 clocks
// (profiles/valu_calibration.json): per node visit 11 -> 9 such instructions, per packet visit 11 -> 4, per wide unit 20 -> 8.
struct Slab {
  float t0, t1;
};
static __device__ __forceinline__ bool slab_met(const Slab& s) { return (s.t1 >= s.t0) & (s.t1 > 0.0f); }
static __device__ __forceinline__ float slab_dist(const Slab& s) { return (s.t0 > 0.0f) ? s.t0 : s.t1; }  // r of a box that is met
static __device__ __forceinline__ Slab slab_reduce(float fx, float fy, float fz, float nx, float ny, float nz, bool exact) {
  Slab s;
  if (exact) {
    float tmaxx = fx > nx ? fx : nx, tmaxy = fy > ny ? fy : ny, tmaxz = fz > nz ? fz : nz;
    float tminx = fx < nx ? fx : nx, tminy = fy < ny ? fy : ny, tminz = fz < nz ? fz : nz;
    s.t1 = jade_fminf(tmaxx, jade_fminf(tmaxy, tmaxz));
    s.t0 = jade_fmaxf(tminx, jade_fmaxf(tminy, tminz));
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
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(s.t1) : "v"(hx), "v"(hy), "v"(hz));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(s.t0) : "v"(lx), "v"(ly), "v"(lz));
  }
  return s;
}
// both children of a node record (jade_device.h): lane .x = left box, .y = right box
static __device__ __forceinline__ void slab2(const RayOD& od, jvec3 inv, float4 q0, float4 q1, float4 q2, bool exact, Slab* s1, Slab* s2) {
  const f2 ax = {q0.x, q0.y}, ay = {q0.z, q0.w}, az = {q1.x, q1.y}, bx = {q1.z, q1.w}, by = {q2.x, q2.y}, bz = {q2.z, q2.w};
  const f2 fx = (bx - OD_OX(od)) * f2s(inv.x), fy = (by - OD_OY(od)) * f2s(inv.y), fz = (bz - OD_OZ(od)) * f2s(inv.z);
  const f2 nx = (ax - OD_OX(od)) * f2s(inv.x), ny = (ay - OD_OY(od)) * f2s(inv.y), nz = (az - OD_OZ(od)) * f2s(inv.z);
  *s1 = slab_reduce(fx.x, fy.x, fz.x, nx.x, ny.x, nz.x, exact);
  *s2 = slab_reduce(fx.y, fy.y, fz.y, nx.y, ny.y, nz.y, exact);
}

static __device__ __forceinline__ bool finite_f(float x) { return (jade_f2u(x) & 0x7f800000u) != 0x7f800000u; }

// One lane's traversal state for hitBVH (PathTrace.cu:795-859).
//
// The reference never prunes against the best hit, so WHICH nodes and leaves a ray visits does
// not depend on any triangle test: the node walk and the triangle tests are two independent
// streams of work, joined only by the order in which leaves are met (strict "<" keeps the
// first of two equal distances, so leaves must be tested in the order the walk meets them).
// A lane therefore walks nodes without ever stopping at a leaf — a leaf it meets goes into a
// small FIFO of leaf cursors — and tests triangles from the head of that FIFO.  A wave
// iteration runs ONE of the two kinds of work for all the lanes that have some of it
// (k_trace picks the kind with more lanes), which a lane that still walks nodes and also has
// leaves waiting can always join.  With the kinds interleaved per
// lane, as in a plain loop over hitBVH, each instruction runs for about a third of the lanes.
//
// In registers: o, normalize(d), the skip index, the node cursor, the leaf cursor, the stack pointer and the FIFO
// counters.  In the lane's LDS column: the stack, the FIFO, 1/d (read by node visits only) and the best hit so far
// (touched only when a triangle is actually hit).
// A leaf cursor is the leaf reference itself: LEAF | 5 * first_pair << 4 | pairs.  Bits 4-30 are the byte offset of
// the next 80-B pair record (ray_step_tri_s), so a test needs one AND to address it and +79 to advance (offset + 80,
// one pair fewer).
struct RayState {
  RayOD od;        // origin and normalize(d)
  uint32_t skipx;  // bit 31: a component of o or 1/d is not finite (NaN-faithful slab needed); bits 0-30: the source triangle, 0x7fffffff = none
  uint32_t cur;    // node walk: internal-node ref, a leaf ref not yet queued, or JADE_REF_NONE = walk finished
  uint32_t leaf;   // triangle tests: cursor of the leaf being tested, 0 = none (then the FIFO is empty too)
  uint32_t sp;     // stack pointer: LDS byte address of the next free level (stk.col = empty)
  uint32_t fw, fr; // leaf FIFO: cursors written / read so far, in units of JADE_COL_STRIDE (slot address = FIFO base | (counter & JADE_FIFO_MASK))
};

static __device__ __forceinline__ void ray_begin(RayState& r, const LdsStack& stk, const DevScene& S, jvec3 o, jvec3 d, int32_t skip) {
  const jvec3 inv = jv(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  const jvec3 dn = jv_normalize(d);
  r.od.a = f2{o.x, o.y};
  r.od.b = f2{o.z, dn.x};
  r.od.c = f2{dn.y, dn.z};
  const bool exact = !(finite_f(inv.x) && finite_f(inv.y) && finite_f(inv.z)) || !finite_f(o.x) || !finite_f(o.y) || !finite_f(o.z);
  r.skipx = (skip < 0 ? 0x7fffffffu : (uint32_t)skip) | (exact ? 0x80000000u : 0u);  // pair records carry triangle indices
  r.sp = stk.col;
  r.fw = r.fr = 0;
  r.leaf = 0;
  r.cur = S.root_ref;
  lds_putf(stk, LW_INVX, inv.x);
  lds_putf(stk, LW_INVY, inv.y);
  lds_putf(stk, LW_INVZ, inv.z);
  lds_putf(stk, LW_BEST_DIST, JADE_INF_F);
  lds_put(stk, LW_BEST_INDEX, 0xffffffffu);
}
// a lane that holds no ray
static __device__ __forceinline__ void ray_clear(RayState& r, const LdsStack& stk) {
  r.cur = JADE_REF_NONE;
  r.leaf = 0;
  r.sp = stk.col;
  r.fw = r.fr = 0;
  r.skipx = 0;
  r.od.a = r.od.b = r.od.c = f2{0.0f, 0.0f};
}

static __device__ __forceinline__ bool ray_done(const RayState& r) { return r.cur == JADE_REF_NONE && r.leaf == 0; }
// room for one more leaf cursor
static __device__ __forceinline__ bool leaf_room(const RayState& r) { return r.leaf == 0 || r.fw - r.fr < JADE_LDS_FIFO * JADE_COL_STRIDE; }
// a lane can take part in a node iteration / a triangle iteration
static __device__ __forceinline__ bool ray_can_walk(const RayState& r) {
  return r.cur != JADE_REF_NONE && (!(r.cur & JADE_REF_LEAF) || leaf_room(r));
}
static __device__ __forceinline__ bool ray_can_test(const RayState& r) { return r.leaf != 0; }

static __device__ __forceinline__ int32_t ray_best_index(const LdsStack& stk) { return (int32_t)lds_get(stk, LW_BEST_INDEX); }  // triangle index, ~0 = -1 = miss
static __device__ __forceinline__ jvec3 ray_hit_point(const LdsStack& stk) {
  return jv(lds_getf(stk, LW_PX), lds_getf(stk, LW_PY), lds_getf(stk, LW_PZ));
}

// ---------------------------------------------------------------------------------------------------------------
// The two steps, straight-line.  A wave issues one instruction at a time, of whatever kind, and nested ifs cost it twice: every `if` is 3-4
// scalar instructions of EXEC bookkeeping in the wave's (serial) instruction stream - scalar and branch instructions
// were 40 % of all instructions issued by the first, branchy form (round 1; git history) - and each side of it runs for
// a part of the lanes only.  Here a decision selects values (v_cndmask) and, where it guards a store, the store's
// ADDRESS: a lane the statement does not apply to writes its column's LW_DUMMY word (LDS is 7 % busy).  Only the rare
// cases stay branches: a stack deeper than its LDS levels (0.4 % of the rays) and a triangle that is actually hit.
// ---------------------------------------------------------------------------------------------------------------

// One unit of the node walk for a lane with ray_can_walk: a leaf reference in `cur` (the far child of an earlier visit
// coming off the stack, or a leaf that found the FIFO full) is queued and replaced by the next reference; an internal
// node is visited: both children's slab tests, near-first descent (PathTrace.cu:835-848), the far child pushed.
// vcnt: this lane's count of child records visited.
// GENERAL = false leaves out what almost no wave needs (decided per wave, a scalar branch in the caller): the
// NaN-faithful slab reduction (a ray with a non-finite 1/d or origin) and children that do not exist (the reference's
// "child 0"; jade_scene_create tells whether the tree has any).
// The core shared by the two forms of the walk (a FIFO of leaf cursors per lane: k_light; one queue of leaves per wave:
// k_trace).  W_INV / W_DUMMY: where the column keeps 1/d and the dummy word.  room: a leaf met now can be taken.  Returns
// the leaf met (0 = none); cur and sp advance.
struct NodeRec {  // a node record in registers: the children's boxes (interleaved, jade_device.h) and their references
  float4 a, b, c;
  uint2 rf;
};
// the record of the node `cur` stands at (a leaf reference reads record 0 and ignores it)
template <int TOPN>
static __device__ __forceinline__ NodeRec node_fetch(uint32_t cur, const DevScene& S, const LdsStack& stk) {
  const bool is_leaf = (int32_t)cur < 0;
  const uint32_t node = is_leaf ? 0u : cur;  // such a lane reads record 0 and ignores it
  float4 a, b, c;
  uint2 rf;
#if JADE_LDS_TOP_NODES > 0
  {
    // The top of the tree lives in LDS, one plane per 16 bytes of the record (lane addresses 16 B apart spread over all
    // 64 banks of a ds_read_b128); these visits - the most frequent ones - stay out of the vector-memory path.  Every
    // lane reads LDS (a lane below the top reads entry 0: LDS bandwidth is idle) and only the lanes below the top issue
    // global loads.  The LDS values are made opaque: written as "LDS or global", the loads become per-dword FLAT loads
    // through a selected generic pointer (12 loads per visit instead of 4).
    const bool in_top = node < stk.top_k;
    const float4* t = stk.top + (in_top ? node : 0u);
    const float4 la = t[0], lb = t[TOPN], lc = t[2 * TOPN], lr = t[3 * TOPN];
    a = la;
    b = lb;
    c = lc;
    rf = make_uint2(jade_f2u(lr.x), jade_f2u(lr.y));
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));
    asm volatile("" : "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.w), "+v"(rf.x), "+v"(rf.y));
    if (!in_top) {
      const float4* nd = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.nodes) + node * 64u);
      a = nd[0];
      b = nd[1];
      c = nd[2];
      rf = *reinterpret_cast<const uint2*>(nd + 3);
#if JADE_ABLATE_LOAD == 1
      {  // prices the vector-memory path: one more 16-B gather per lane from the line just fetched (+25 % look-ups, same bytes from L2)
        const float4 x = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(nd) + 40);
        asm volatile("" ::"v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w));
      }
#elif JADE_ABLATE_LOAD == 3
      {  // ... and from another 128-B LINE (two records on: records are 64 B, so `node ^ 1` shares the line that is being fetched anyway)
        const float4 x = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.nodes) + (node ^ 2u) * 64u);
        asm volatile("" ::"v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w));
      }
#elif JADE_ABLATE_LOAD == 2
      {  // ... and from ANOTHER line (the neighbouring record): +25 % look-ups and +1 line from L2 per visit
        const float4 x = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.nodes) + (node ^ 1u) * 64u);
        asm volatile("" ::"v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w));
      }
#endif
    }
  }
#else
  {
    const float4* nd = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.nodes) + node * 64u);
    a = nd[0];
    b = nd[1];
    c = nd[2];
    rf = *reinterpret_cast<const uint2*>(nd + 3);
  }
#endif
  NodeRec nr;
  nr.a = a;
  nr.b = b;
  nr.c = c;
  nr.rf = rf;
  return nr;
}
// the slab tests and what follows from them (see above node_fetch)
template <bool GENERAL, int W_INV, int W_DUMMY>
static __device__ __forceinline__ uint32_t node_decide(const NodeRec& nr, uint32_t& cur_io, uint32_t& sp_io, const RayOD& od, const LdsStack& stk, bool room,
                                                       uint32_t& vcnt, const jvec3* inv_reg = nullptr) {
  const uint32_t cur = cur_io;
  const bool is_leaf = (int32_t)cur < 0;
  const float4 a = nr.a, b = nr.b, c = nr.c;
  const uint2 rf = nr.rf;
  // 1/d: from the caller's registers (k_trace, which has them to spare at 4 waves per SIMD) or from the lane's column
  const jvec3 inv = inv_reg ? *inv_reg : jv(lds_getf(stk, W_INV), lds_getf(stk, W_INV + 1), lds_getf(stk, W_INV + 2));
  Slab s1, s2;
#if JADE_ABLATE_SLAB
  {
    RayOD o2 = od;
    o2.a.x += 1e-30f;
    Slab e1, e2;
    slab2(o2, inv, a, b, c, GENERAL, &e1, &e2);
    asm volatile("" ::"v"(e1.t0 + e2.t0 + e1.t1 + e2.t1));
  }
#endif
  slab2(od, inv, a, b, c, GENERAL, &s1, &s2);
  bool in1, in2;
  if (GENERAL) {
    const bool c1 = !is_leaf && rf.x != JADE_REF_NONE, c2 = !is_leaf && rf.y != JADE_REF_NONE;  // a missing child is neither counted nor entered
    vcnt += (c1 ? 1u : 0u) + (c2 ? 1u : 0u);
    in1 = c1 && slab_met(s1);
    in2 = c2 && slab_met(s2);
  } else {
    vcnt += is_leaf ? 0u : 2u;
    in1 = !is_leaf && slab_met(s1);
    in2 = !is_leaf && slab_met(s2);
  }
  const bool both = in1 && in2, any = in1 || in2;
  const bool first = slab_dist(s1) < slab_dist(s2);  // near child first, PathTrace.cu:835-848 (looked at only where both are met)
  const uint32_t near = both ? (first ? rf.x : rf.y) : (in1 ? rf.x : rf.y);
  const uint32_t far = first ? rf.y : rf.x;
  const bool near_leaf_ok = any && (int32_t)near < 0 && room;  // the near leaf is met now
  // both, near leaf met: the far child is next, nothing to push.  both otherwise: push far, go near.  one: go there (a
  // leaf that is met ends the branch: pop).  none: pop.
  const uint32_t leafv = is_leaf ? cur : (near_leaf_ok ? near : 0u);
  const bool push = both && !near_leaf_ok;
  const uint32_t next = both ? (near_leaf_ok ? far : near) : near;
  const bool need_pop = is_leaf || !any || (!both && near_leaf_ok);
  const uint32_t dummy = stk.col + W_DUMMY * JADE_COL_STRIDE;
  // ---- push the far child
  const uint32_t lds_end = stk.col + JADE_LDS_STACK * JADE_COL_STRIDE;  // address of the first level that is not in LDS
  uint32_t sp = sp_io;
  lds_st((push && sp < lds_end) ? sp : dummy, far);
  if (push && sp >= lds_end) stk.spill[(size_t)((sp - lds_end) / JADE_COL_STRIDE) * stk.stride_spill] = far;  // rare
  sp += push ? JADE_COL_STRIDE : 0u;
  // ---- pop (a lane pushes or pops, never both)
  const bool do_pop = need_pop && sp != stk.col;
  const uint32_t sp1 = sp - JADE_COL_STRIDE;
  uint32_t top = lds_ld((do_pop && sp1 < lds_end) ? sp1 : dummy);
  asm volatile("" : "+v"(top));  // keeps it a ds_read: "LDS, or global for some lanes" would become one FLAT load through a selected generic pointer
  if (do_pop && sp1 >= lds_end) top = stk.spill[(size_t)((sp1 - lds_end) / JADE_COL_STRIDE) * stk.stride_spill];  // rare
  sp_io = do_pop ? sp1 : sp;
  cur_io = need_pop ? (do_pop ? top : JADE_REF_NONE) : next;
  return leafv;
}
template <bool GENERAL, int W_INV, int W_DUMMY, int TOPN>
static __device__ __forceinline__ uint32_t node_core(uint32_t& cur_io, uint32_t& sp_io, const RayOD& od, const DevScene& S, const LdsStack& stk, bool room,
                                                     uint32_t& vcnt, const jvec3* inv_reg = nullptr) {
  const NodeRec nr = node_fetch<TOPN>(cur_io, S, stk);
  return node_decide<GENERAL, W_INV, W_DUMMY>(nr, cur_io, sp_io, od, stk, room, vcnt, inv_reg);
}

// ---------------------------------------------------------------------------------------------------------------
// Wide walk (round 3; k_trace with early exits only).  hitAABB is monotone in float32: a child's box lies inside its parent's
// exactly, every step of the slab test is a monotone function of the box's coordinates, and the return rule keeps "> 0" under
// enlargement (tools/box_monotone_probe.py) - a ray that meets a node's box meets every ancestor's, with the values as computed.
// So the leaves hitBVH tests for a ray are {leaves whose own box it meets}; the inner nodes decide only the ORDER in which they
// are met, which matters to nobody but hitArray's tie rule (:787).  A wide unit visits a node by testing its four GRANDCHILDREN
// (jade_device.h, nodes4) and never looks at the two children: half the dependent record fetches.  Same reference space, same
// stack, same leaf ring as the binary unit - a ray can take either kind of unit at any node.  The order of a wide walk is not
// the reference's, so whenever two candidates from different leaves tie for a ray's best distance the ray is walked again with
// binary units only (JADE_LIMIT_TIE, JADE_FORCE_BINARY: k_trace); rays with a non-finite 1/d take the binary general unit as ever.
// A leaf chosen as the next node is queued by the following unit (as a leaf that comes off the stack is).
// ---------------------------------------------------------------------------------------------------------------
#ifndef JADE_WIDE_ORDERED
#define JADE_WIDE_ORDERED 0 /* 1: round 3's form - the nearest of the four boxes met is next, the others are pushed nearest last (43 selects per unit).  Round 4, same process (profiles/r04_wide_unordered_ab.txt), k_trace per step: C5 114.8 -> 108.2 ms with the boxes taken in slot order; C3 with JADE_WIDE=1 119.8 -> 114.8 (binary units: 117.9), close-up 832 -> 813 (binary: 809) */
#endif
struct NodeRec4 {
  float4 a0, b0, c0, a1, b1, c1;
  uint4 rf;
};
static __device__ __forceinline__ NodeRec4 node_fetch4(uint32_t cur, const DevScene& S, const LdsStack& stk) {
  const bool is_leaf = (int32_t)cur < 0;
  const uint32_t node = is_leaf ? 0u : cur;
  const bool in_top = node < stk.top4_k;
  const float4* t = stk.top4 + (in_top ? node : 0u);
  NodeRec4 n;
  n.a0 = t[0];
  n.b0 = t[JADE_TRACE_TOP4];
  n.c0 = t[2 * JADE_TRACE_TOP4];
  n.a1 = t[3 * JADE_TRACE_TOP4];
  n.b1 = t[4 * JADE_TRACE_TOP4];
  n.c1 = t[5 * JADE_TRACE_TOP4];
  const float4 lr = t[6 * JADE_TRACE_TOP4];
  n.rf = make_uint4(jade_f2u(lr.x), jade_f2u(lr.y), jade_f2u(lr.z), jade_f2u(lr.w));
  // (opaque: "LDS or global" would become FLAT loads through a selected generic pointer, see node_fetch)
  asm volatile("" : "+v"(n.a0.x), "+v"(n.a0.y), "+v"(n.a0.z), "+v"(n.a0.w), "+v"(n.b0.x), "+v"(n.b0.y), "+v"(n.b0.z), "+v"(n.b0.w));
  asm volatile("" : "+v"(n.c0.x), "+v"(n.c0.y), "+v"(n.c0.z), "+v"(n.c0.w), "+v"(n.a1.x), "+v"(n.a1.y), "+v"(n.a1.z), "+v"(n.a1.w));
  asm volatile("" : "+v"(n.b1.x), "+v"(n.b1.y), "+v"(n.b1.z), "+v"(n.b1.w), "+v"(n.c1.x), "+v"(n.c1.y), "+v"(n.c1.z), "+v"(n.c1.w));
  asm volatile("" : "+v"(n.rf.x), "+v"(n.rf.y), "+v"(n.rf.z), "+v"(n.rf.w));
  if (!in_top) {
    const float4* nd = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.nodes4) + (size_t)node * 128u);
    n.a0 = nd[0];
    n.b0 = nd[1];
    n.c0 = nd[2];
    n.a1 = nd[3];
    n.b1 = nd[4];
    n.c1 = nd[5];
    n.rf = *reinterpret_cast<const uint4*>(nd + 6);
  }
  return n;
}
template <int W_DUMMY>
static __device__ __forceinline__ uint32_t node_decide4(const NodeRec4& nr, uint32_t& cur_io, uint32_t& sp_io, const RayOD& od, const LdsStack& stk,
                                                        uint32_t& vcnt, const jvec3& inv) {
  const uint32_t cur = cur_io;
  const bool is_leaf = (int32_t)cur < 0;
  Slab b0, b1, b2, b3;
  slab2(od, inv, nr.a0, nr.b0, nr.c0, false, &b0, &b1);
  slab2(od, inv, nr.a1, nr.b1, nr.c1, false, &b2, &b3);
  const uint4 rf = nr.rf;
  const bool v1 = rf.y != JADE_REF_NONE, v3 = rf.w != JADE_REF_NONE;  // (slots 0 and 2 always hold a child)
  vcnt += is_leaf ? 0u : 2u + (v1 ? 1u : 0u) + (v3 ? 1u : 0u);
  const bool in0 = !is_leaf && slab_met(b0), in1 = !is_leaf && v1 && slab_met(b1), in2 = !is_leaf && slab_met(b2), in3 = !is_leaf && v3 && slab_met(b3);
#if JADE_WIDE_ORDERED
  const float d0 = slab_dist(b0), d1 = slab_dist(b1), d2 = slab_dist(b2), d3 = slab_dist(b3);  // (compared only where both boxes are met)
  // the nearest of the boxes met is next; of the others, the pair it does not belong to goes onto the stack first (it comes off
  // last), its sibling last
  // (decisions as and / or of lane masks: a `?:` between two of them is materialised in a register and compared again)
  const bool s01 = in1 & (!in0 | (d1 < d0)), s23 = in3 & (!in2 | (d3 < d2));  // the nearer of each pair is its second member
  const float m01 = s01 ? d1 : d0, m23 = s23 ? d3 : d2;
  const bool nin01 = in0 | in1, nin23 = in2 | in3;                            // a pair has a box that is met
  const bool p1 = nin23 & (!nin01 | (m23 < m01));                             // the nearest of all lies in pair 1
  const bool any = nin01 | nin23;
  const uint32_t n01 = s01 ? rf.y : rf.x, f01 = s01 ? rf.x : rf.y, n23 = s23 ? rf.w : rf.z, f23 = s23 ? rf.z : rf.w;
  const bool fin01 = in0 & in1, fin23 = in2 & in3;  // the farther one of a pair is met too = both are
  const uint32_t next = p1 ? n23 : n01;
  // pushes, in this order: other pair's farther, other pair's nearer, own pair's farther (in plain slot order instead: the same speed)
  const uint32_t q0 = p1 ? f01 : f23, q1 = p1 ? n01 : n23, q2 = p1 ? f23 : f01;
  const bool w0 = (p1 & fin01) | (!p1 & fin23), w1 = (p1 & nin01) | (!p1 & nin23), w2 = (p1 & fin23) | (!p1 & fin01);
#else
  // Unordered (round 4): the boxes met are walked in SLOT order - the first one met is next, the others go onto the stack so that
  // they come off in slot order.  The order of a wide walk is nobody's business: a ray that wants its nearest hit visits every
  // box it meets whatever the order (nothing is pruned by distance), a tie between leaves is walked again with binary units
  // (JADE_LIMIT_TIE), and a yes/no query only ends sooner or later.  What the nearest-first order cost was 43 selects per unit.
  const bool any = in0 | in1 | in2 | in3;
  const uint32_t next = in0 ? rf.x : in1 ? rf.y : in2 ? rf.z : rf.w;
  const uint32_t q0 = rf.w, q1 = rf.z, q2 = rf.y;
  const bool w0 = in3 & (in0 | in1 | in2), w1 = in2 & (in0 | in1), w2 = in1 & in0;
#endif
  const uint32_t dummy = stk.col + W_DUMMY * JADE_COL_STRIDE;
  const uint32_t lds_end = stk.col + JADE_LDS_STACK * JADE_COL_STRIDE;
  uint32_t sp = sp_io;
  const bool roomy = sp + 3u * JADE_COL_STRIDE <= lds_end;  // whatever this unit pushes stays in the LDS levels
  if (__ballot((w0 | w1 | w2) & !roomy) == 0ull) {
    lds_st(w0 ? sp : dummy, q0);
    sp += w0 ? JADE_COL_STRIDE : 0u;
    lds_st(w1 ? sp : dummy, q1);
    sp += w1 ? JADE_COL_STRIDE : 0u;
    lds_st(w2 ? sp : dummy, q2);
    sp += w2 ? JADE_COL_STRIDE : 0u;
  } else {  // rare: some lane's stack outgrows its LDS levels
    lds_st((w0 && sp < lds_end) ? sp : dummy, q0);
    if (w0 && sp >= lds_end) stk.spill[(size_t)((sp - lds_end) / JADE_COL_STRIDE) * stk.stride_spill] = q0;
    sp += w0 ? JADE_COL_STRIDE : 0u;
    lds_st((w1 && sp < lds_end) ? sp : dummy, q1);
    if (w1 && sp >= lds_end) stk.spill[(size_t)((sp - lds_end) / JADE_COL_STRIDE) * stk.stride_spill] = q1;
    sp += w1 ? JADE_COL_STRIDE : 0u;
    lds_st((w2 && sp < lds_end) ? sp : dummy, q2);
    if (w2 && sp >= lds_end) stk.spill[(size_t)((sp - lds_end) / JADE_COL_STRIDE) * stk.stride_spill] = q2;
    sp += w2 ? JADE_COL_STRIDE : 0u;
  }
  // ---- the nearest box is a leaf's: it is queued by this very unit and the walk goes on with what comes off the stack - which
  // may be what this unit has just pushed: a write and a read of the same LDS word in program order; a stack that has outgrown
  // its LDS levels leaves the leaf to the next unit instead
  const bool leaf_now = !is_leaf & any & ((int32_t)next < 0) & roomy;
  // ---- pop: a leaf that was queued by this unit, or a node none of whose boxes is met (such a lane pushed nothing)
  const bool need_pop = is_leaf | !any | leaf_now;
  const bool do_pop = need_pop & (sp != stk.col);
  const uint32_t sp1 = sp - JADE_COL_STRIDE;
  uint32_t top = lds_ld((do_pop && sp1 < lds_end) ? sp1 : dummy);
  asm volatile("" : "+v"(top));
  if (do_pop && sp1 >= lds_end) top = stk.spill[(size_t)((sp1 - lds_end) / JADE_COL_STRIDE) * stk.stride_spill];  // rare (never after leaf_now)
  sp_io = do_pop ? sp1 : sp;
  cur_io = need_pop ? (do_pop ? top : JADE_REF_NONE) : next;
  return is_leaf ? cur : (leaf_now ? next : 0u);
}

// The walk with a FIFO of leaf cursors per lane (k_light): one unit for a lane with ray_can_walk.
template <bool GENERAL>
static __device__ __forceinline__ void ray_step_node_s(RayState& r, const DevScene& S, const LdsStack& stk, uint32_t& vcnt) {
  const uint32_t leafv = node_core<GENERAL, LW_INVX, LW_DUMMY, JADE_LDS_TOP_NODES>(r.cur, r.sp, r.od, S, stk, leaf_room(r), vcnt);
  // the leaf met goes to the leaf cursor, or behind it into the FIFO
  const bool lq = (leafv & 15u) != 0;  // (an empty leaf cannot happen for a valid BVH)
  const bool to_fifo = lq && r.leaf != 0;
  const uint32_t slot = (stk.col + LW_FIFO * JADE_COL_STRIDE) | (r.fw & JADE_FIFO_MASK);
  lds_st(to_fifo ? slot : stk.col + LW_DUMMY * JADE_COL_STRIDE, leafv);
  r.leaf = (lq && r.leaf == 0) ? leafv : r.leaf;
  r.fw += to_fifo ? JADE_COL_STRIDE : 0u;
}

// ---------------------------------------------------------------------------------------------------------------
// Two triangles per test.  The triangle test is half of k_trace's instructions on the rays that matter (jade paths:
// 40 tests per ray).  Two consecutive triangles of a leaf share every instruction: each statement of hitTriangle
// (PathTrace.cu:705-754, normalize(dir) hoisted) runs once for triangle A in lane .x and for triangle B in lane .y of a
// packed instruction (IEEE per component: the values are those of the scalar statements).  The vertex data is laid out
// for it (jade_scene_create): one 80-B record per pair,
//   {A.p1x B.p1x A.p1y B.p1y} {A.p1z B.p1z A.p2x B.p2x} {A.p2y B.p2y A.p2z B.p2z} {A.p3x B.p3x A.p3y B.p3y}
//   {A.p3z B.p3z indexA flags}          flags bit 0: B is a triangle (an odd leaf's last record repeats A and clears it),
// and a leaf reference is LEAF | 5 * first_pair << 4 | pairs: bits 4-30 are the byte offset of the next record.
// A is resolved before B (hitArray's index order, :776-792, strict "<"), the source triangle is skipped by index in
// either lane (:782).
// ---------------------------------------------------------------------------------------------------------------
// the part of hitTriangle that runs once the projected origin is inside the projected triangle (:732-747)
static __device__ __forceinline__ bool tri_hit(jvec3 p1, jvec3 p2, jvec3 p3, jvec3 sa, jvec3 sb, jvec3 sc, jvec3 o, jvec3 dn, float* dist_out,
                                               jvec3* point_out) {
  jvec3 eb = jv_sub(sb, sa), ec = jv_sub(sc, sa), q = jv_sub(o, sa);
  float divider = jade_diffprod(eb.x, ec.y, eb.y, ec.x);
  float rate_a = jade_diffprod(ec.y, q.x, ec.x, q.y) / divider;
  float rate_b = jade_fma(eb.x, q.y, (-eb.y) * q.x) / divider;
  jvec3 P = jv_add(jv_add(p1, jv_scale(jv_sub(p2, p1), rate_a)), jv_scale(jv_sub(p3, p1), rate_b));
  float distance = jv_dot(jv_sub(P, o), dn);
  *dist_out = distance;
  *point_out = P;
  return distance > 0;
}

// The packed test of one pair record for one ray: is the projected origin inside the projected triangle A / B (:711-731)?
// skip: the ray's source triangle.  tcnt: this lane's count of tests.
struct PairRec {
  float4 q0, q1, q2, q3, q4;
};
static __device__ __forceinline__ PairRec pair_load(const DevScene& S, uint32_t off) {
  const float4* t0 = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.tverts) + off);
  PairRec p;
  p.q0 = t0[0];
  p.q1 = t0[1];
  p.q2 = t0[2];
  p.q3 = t0[3];
  p.q4 = t0[4];
  return p;
}
static __device__ __forceinline__ void pair_core(const RayOD& od, uint32_t skip, const PairRec& rec, uint32_t& tcnt, bool& in_a_out, bool& in_b_out,
                                                 uint32_t& idx_a_out) {
  const float4 q0 = rec.q0, q1 = rec.q1, q2 = rec.q2, q3 = rec.q3, q4 = rec.q4;
  const uint32_t idx_a = jade_f2u(q4.z);
  const bool test_a = idx_a != skip, test_b = (jade_f2u(q4.w) & 1u) != 0 && idx_a + 1u != skip;
  tcnt += (test_a ? 1u : 0u) + (test_b ? 1u : 0u);
  // lane .x: triangle A, lane .y: triangle B
  const f2 p1x = {q0.x, q0.y}, p1y = {q0.z, q0.w}, p1z = {q1.x, q1.y};
  const f2 p2x = {q1.z, q1.w}, p2y = {q2.x, q2.y}, p2z = {q2.z, q2.w};
  const f2 p3x = {q3.x, q3.y}, p3y = {q3.z, q3.w}, p3z = {q4.x, q4.y};
  // s = p - dn * dot(dn, p - o)      (jv_dot: fma(a.z, b.z, fma(a.y, b.y, a.x * b.x)))
  const f2 t1 = f2fma(OD_DZ(od), p1z - OD_OZ(od), f2fma(OD_DY(od), p1y - OD_OY(od), OD_DX(od) * (p1x - OD_OX(od))));
  const f2 t2 = f2fma(OD_DZ(od), p2z - OD_OZ(od), f2fma(OD_DY(od), p2y - OD_OY(od), OD_DX(od) * (p2x - OD_OX(od))));
  const f2 t3 = f2fma(OD_DZ(od), p3z - OD_OZ(od), f2fma(OD_DY(od), p3y - OD_OY(od), OD_DX(od) * (p3x - OD_OX(od))));
  const f2 sax = p1x - OD_DX(od) * t1, say = p1y - OD_DY(od) * t1, saz = p1z - OD_DZ(od) * t1;
  const f2 sbx = p2x - OD_DX(od) * t2, sby = p2y - OD_DY(od) * t2, sbz = p2z - OD_DZ(od) * t2;
  const f2 scx = p3x - OD_DX(od) * t3, scy = p3y - OD_DY(od) * t3, scz = p3z - OD_DZ(od) * t3;
  // pa = sa - o, ...
  const f2 pax = sax - OD_OX(od), pay = say - OD_OY(od), paz = saz - OD_OZ(od);
  const f2 pbx = sbx - OD_OX(od), pby = sby - OD_OY(od), pbz = sbz - OD_OZ(od);
  const f2 pcx = scx - OD_OX(od), pcy = scy - OD_OY(od), pcz = scz - OD_OZ(od);
  // mixed(dn, u, v) = dn.x * (u.y v.z - u.z v.y), then fma with the y and z terms   (jv_mixed / jade_diffprod)
  f2 papb = OD_DX(od) * f2fma(pay, pbz, -(paz * pby));
  papb = f2fma(OD_DY(od), f2fma(paz, pbx, -(pax * pbz)), papb);
  papb = f2fma(OD_DZ(od), f2fma(pax, pby, -(pay * pbx)), papb);
  f2 pbpc = OD_DX(od) * f2fma(pby, pcz, -(pbz * pcy));
  pbpc = f2fma(OD_DY(od), f2fma(pbz, pcx, -(pbx * pcz)), pbpc);
  pbpc = f2fma(OD_DZ(od), f2fma(pbx, pcy, -(pby * pcx)), pbpc);
  f2 pcpa = OD_DX(od) * f2fma(pcy, paz, -(pcz * pay));
  pcpa = f2fma(OD_DY(od), f2fma(pcz, pax, -(pcx * paz)), pcpa);
  pcpa = f2fma(OD_DZ(od), f2fma(pcx, pay, -(pcy * pax)), pcpa);
  const bool in_a = test_a && ((papb.x > 0 && pbpc.x > 0 && pcpa.x > 0) || (papb.x < 0 && pbpc.x < 0 && pcpa.x < 0));
  const bool in_b = test_b && ((papb.y > 0 && pbpc.y > 0 && pcpa.y > 0) || (papb.y < 0 && pbpc.y < 0 && pcpa.y < 0));
  in_a_out = in_a;
  in_b_out = in_b;
  idx_a_out = idx_a;
}
// Triangle k (0 = A, 1 = B) of a pair record the origin projects into: the barycentric solve and the distance (:732-747).
// Rare (one test in ten, 2 % of the kernel's time).  Nothing of the packed test is kept alive for it - the record is read
// again (it is in L1) and the triangle's three projections are recomputed with the same statements - because 36 registers
// held across the test for this block cost the kernel a wave per SIMD.
static __device__ __forceinline__ bool pair_hit(const float4* t0, int k, const RayOD& od, float* dist, jvec3* P) {
  const jvec3 o = od_o(od), dn = od_dn(od);
  // five 16-B loads and a select per component (nine 4-B loads would be nine trips through the L1's tag look-up)
  const float4 q0 = t0[0], q1 = t0[1], q2 = t0[2], q3 = t0[3], q4 = t0[4];
  const bool b = k != 0;
  const jvec3 p1 = jv(b ? q0.y : q0.x, b ? q0.w : q0.z, b ? q1.y : q1.x);
  const jvec3 p2 = jv(b ? q1.w : q1.z, b ? q2.y : q2.x, b ? q2.w : q2.z);
  const jvec3 p3 = jv(b ? q3.y : q3.x, b ? q3.w : q3.z, b ? q4.y : q4.x);
  const jvec3 sa = jv_sub(p1, jv_scale(dn, jv_dot(dn, jv_sub(p1, o))));
  const jvec3 sb = jv_sub(p2, jv_scale(dn, jv_dot(dn, jv_sub(p2, o))));
  const jvec3 sc = jv_sub(p3, jv_scale(dn, jv_dot(dn, jv_sub(p3, o))));
  return tri_hit(p1, p2, p3, sa, sb, sc, o, dn, dist, P);
}

// One pair of the leaf at the head of the FIFO for a lane with ray_can_test (k_light).
static __device__ __forceinline__ void ray_step_tri_s(RayState& r, const DevScene& S, const LdsStack& stk, uint32_t& tcnt) {
  const uint32_t off = r.leaf & 0x7ffffff0u;
  const float4* t0 = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.tverts) + off);
  const uint32_t leaf = r.leaf + 79u;  // next pair record (5 x 16 B), one pair fewer
  bool in_a, in_b;
  uint32_t idx_a;
  pair_core(r.od, r.skipx & 0x7fffffffu, pair_load(S, off), tcnt, in_a, in_b, idx_a);
  if (in_a || in_b) {  // A before B (index order, strict "<")
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (k == 0 ? in_a : in_b) {
        float dist;
        jvec3 P;
        if (pair_hit(t0, k, r.od, &dist, &P) && dist < lds_getf(stk, LW_BEST_DIST)) {
          lds_putf(stk, LW_BEST_DIST, dist);
          lds_put(stk, LW_BEST_INDEX, idx_a + (uint32_t)k);
          lds_putf(stk, LW_PX, P.x);
          lds_putf(stk, LW_PY, P.y);
          lds_putf(stk, LW_PZ, P.z);
        }
      }
    }
  }
  // leaf finished: the next one from the FIFO, if any
  const bool fin = (leaf & 15u) == 0;
  const bool has = r.fw != r.fr;
  const uint32_t nxt = lds_ld((stk.col + LW_FIFO * JADE_COL_STRIDE) | (r.fr & JADE_FIFO_MASK));
  r.leaf = fin ? (has ? nxt : 0u) : leaf;
  r.fr += (fin && has) ? JADE_COL_STRIDE : 0u;
}

// ---------------------------------------------------------------------------------------------------------------
// k_trace's form: the leaves a wave's rays meet go into ONE queue per wave, and ANY lane tests them.
//
// With a FIFO per lane, a lane tests the triangles of its own ray, so an iteration of triangle tests runs for the lanes
// whose own ray has a leaf waiting and an iteration of the walk for the lanes whose own ray still walks (and has room):
// on the incoherent rays of the jade paths 29 of 64 lanes worked in an average VALU instruction, and the kernel is
// bound by the number of instructions it issues.  But a triangle test needs nothing of the lane it runs on: the ray's
// origin, direction and skip index are seven registers of the lane that walks it (read with ds_bpermute), and its
// result is a candidate for that lane's best hit.  So:
//   * the walk pushes every leaf it meets as an item {leaf cursor, owner lane | sequence number} onto the wave's ring
//     in LDS (one prefix count per unit, no atomics: a wave runs in lock step);
//   * a test iteration hands the queued items to the lanes in order - ALL lanes, also the ones that hold no ray or whose
//     ray is through with its walk - one pair record per lane and unit, a leaf stays with its lane until it is finished;
//   * hitArray's order (:776-792: leaves as the walk meets them, triangles by index, strict "<" so that the first of
//     two equal distances wins) does not depend on WHEN a triangle is tested: a candidate replaces the owner's best hit
//     if its distance is smaller, or equal with a smaller sequence number, or both equal and it comes earlier in the leaf;
//   * the test itself only finds CANDIDATES (the origin projects into the triangle: one test in ten).  Solving for the
//     hit point and the distance is 90 dependent instructions, and "one in ten" per lane is "nearly always" per wave: run
//     inline, that block ran in every test unit for 3 lanes of 64 and was two thirds of the unit's instructions.  The
//     candidates go into a second ring of the wave and are resolved 32-64 at a time (resolve_hit); two lanes with
//     candidates for the same ray take turns through a lock word in the owner's column;
//   * a ray has ended when its walk has and as many of its leaves have been finished (an LDS counter in its column,
//     ds_add by whoever finishes one) as it pushed.  Its hit point is computed then, once, from the winning triangle
//     (the same statements as in the test: same bits), instead of travelling with every candidate.
// A lane's column: the stack, the best hit {distance, sequence, record | A/B}, the finished-leaf counter, a dummy (1/d is
// in registers: the kernel runs 4 waves per SIMD and has them to spare).
// ---------------------------------------------------------------------------------------------------------------
enum { TW_BEST_DIST = JADE_LDS_STACK, TW_BEST_SEQ, TW_BEST_REF, TW_FINISHED, TW_DUMMY, TW_LIMIT, TW_END };
#define JADE_LIMIT_TIE 0x7fc00001u /* TW_LIMIT of a ray of the wide walk for whose best distance two leaves tied: a NaN (no early exit any more); k_trace walks the ray again with binary units */
#ifndef JADE_WQ
#define JADE_WQ 128 /* items a wave's ring holds (a power of two, >= 128: a walk unit may push 64) */
#endif
static_assert((JADE_WQ & (JADE_WQ - 1)) == 0 && JADE_WQ >= 128, "JADE_WQ");
#ifndef JADE_HQ
#define JADE_HQ 128 /* candidates a wave's second ring holds (a power of two, >= 128: a test unit may push 64 for A, then 64 for B) */
#endif
static_assert((JADE_HQ & (JADE_HQ - 1)) == 0 && JADE_HQ >= 128, "JADE_HQ");
#ifndef JADE_HQ_BATCH
#define JADE_HQ_BATCH 32 /* candidates waiting that trigger a resolve pass.  With early exits (round 3) a ray learns that it has its answer when its candidates are resolved: 64 / 32 / 16 = 127.6 / 121.0 / 121.2 ms of k_trace per 256-spp step of C3, 912 / 869 / 873 on the close-up (reference walk, round 2: 64 was best by 1-3 %) */
#endif

#ifndef JADE_PREFETCH
#define JADE_PREFETCH 0 /* 1: a walk unit ends by requesting the record of the node the walk goes to next (WalkState.pre), so that its latency passes while the wave pushes leaves, picks its next kind of work, tests triangles */
#endif
#define JADE_CUT 0x40000000u          /* WalkState.skipx: the ray has its answer (early exit) */
#define JADE_FORCE_BINARY 0x20000000u /* ... this ray takes binary units only (it is being walked again after a tie, "Wide walk") */
#define JADE_ATTEMPT 0x10000000u      /* ... the walk in progress covers the cached subtrees only ("Occluder cache" below): without an answer the whole walk follows */
#define JADE_WANTS_POINT 0x08000000u  /* ... the caller wants the nearest hit itself - triangle, hit point, distance (its limit is a NaN); a yes/no query gets the triangle alone */
#define JADE_SKIP_MASK 0x03ffffffu    /* ... its source triangle (all ones = none; triangle indices stay below 2^27 / 3 < 2^26) */

// ---------------------------------------------------------------------------------------------------------------
// Occluder cache (round 4; jade_render_params.walk == JADE_WALK_EARLY_EXIT_CACHED).  A shadow or environment-visibility query only
// asks whether SOME triangle the reference's walk tests is hit below the query's limit.  Which triangles the
// reference tests does not depend on order: a leaf is reached iff the ray meets its box and every ancestor's, and - boxes nested,
// hitAABB monotone (3.3c; jade_scene_create checks the nesting) - a leaf whose own box is met is reached.  So a walk may START
// anywhere: walking the subtree under an internal node X tests X's children's boxes as the reference does, every leaf it reaches
// is one the reference reaches, and a hit below the limit found there settles the query exactly as it would have later.  Such
// queries repeat: from one source triangle towards one emitter (or into one octant of the sky) the occluder is mostly the same
// piece of geometry - the floor under the statue, the statue's other side.  The module keeps, per (source triangle, query kind),
// four internal-node references: the parents of the leaves in which the last whole walks of such queries found their answer
// (a pair record carries its leaf's parent in the upper bits of its flag word, so the walk learns it from the record it re-reads
// for the result anyway).  A query first walks those subtrees (JADE_ATTEMPT: the lane's stack starts with them instead of the
// root - the same units of work, nothing new in the loop); with an answer it is done, without one it is walked again from the
// root and, if that walk ends early, its leaf's parent replaces one of the four.  Measured on the oracle first
// (tools/anyhit_probe.py, profiles/r04_anyhit_probe_oracle.txt): 70 % of C3's occluded shadow queries and 96 % of its occluded
// environment queries are answered by the cached subtrees, -60 % / -85 % node records for those queries.
// What it does NOT change: any answer (the frame, the rays, the samples: asserted bit for bit against the reference walk).  What
// it does change: nodes_visited / tris_tested - they count what was read, and with a cache shared by all waves they are no
// longer the same from run to run.
// ---------------------------------------------------------------------------------------------------------------
#define JADE_ANYHIT_KEYS 10 /* per source triangle: shadow query towards emitter 0 / another emitter, environment query by the octant of its direction */
struct WalkState {
#if JADE_PREFETCH
  NodeRec pre;      // the record of `cur`, requested when cur was set
#endif
  RayOD od;         // origin and normalize(d)
  uint32_t skipx;   // as in RayState
  uint32_t cur;     // internal-node ref, a leaf ref (a far child off the stack), or JADE_REF_NONE = walk finished
  uint32_t sp;      // LDS byte address of the next free stack level
  uint32_t pushed;  // leaves met so far = the next leaf's sequence number
  jvec3 inv;        // 1/d
};
static __device__ __forceinline__ void lds_st_v(uint32_t addr, uint32_t v) { *(volatile jade_lds_u32*)(__SIZE_TYPE__)addr = v; }
static __device__ __forceinline__ uint32_t lds_ld_v(uint32_t addr) { return *(volatile jade_lds_u32*)(__SIZE_TYPE__)addr; }

// limit: the walk may end as soon as the ray's best distance is < limit (jade_device.h, PathState.early_exit; a NaN = never:
// the nearest hit is wanted and the walk is the reference's)
// query_limit: the limit word the caller put beside the ray, whatever the walk mode - a NaN there means "the nearest hit itself is
// wanted" (JADE_WANTS_POINT); `limit` is what may end the walk early (a NaN in the reference's walk mode for every ray)
static __device__ __forceinline__ void walk_begin(WalkState& r, const LdsStack& stk, const DevScene& S, jvec3 o, jvec3 d, int32_t skip, float limit, float query_limit) {
  const jvec3 inv = jv(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  const jvec3 dn = jv_normalize(d);
  r.od.a = f2{o.x, o.y};
  r.od.b = f2{o.z, dn.x};
  r.od.c = f2{dn.y, dn.z};
  const bool exact = !(finite_f(inv.x) && finite_f(inv.y) && finite_f(inv.z)) || !finite_f(o.x) || !finite_f(o.y) || !finite_f(o.z);
  r.skipx = (skip < 0 ? JADE_SKIP_MASK : (uint32_t)skip) | (exact ? 0x80000000u : 0u) | (query_limit == query_limit ? 0u : JADE_WANTS_POINT);
  r.sp = stk.col;
  r.pushed = 0;
  r.cur = S.root_ref;
  r.inv = inv;
  lds_putf(stk, TW_BEST_DIST, JADE_INF_F);
  lds_put(stk, TW_BEST_SEQ, 0xffffffffu);
  lds_put(stk, TW_BEST_REF, 0xffffffffu);
  lds_st_v(stk.col + TW_FINISHED * JADE_COL_STRIDE, 0u);
  lds_putf(stk, TW_LIMIT, limit);
#if JADE_PREFETCH
  r.pre = node_fetch<JADE_TRACE_TOP_NODES>(r.cur, S, stk);
#endif
}
// walk_begin's arithmetic on its own - what a ray record carries (jade_device.h, PathState.rayq): 1 / d, normalize(d) and the skip word
static __device__ __forceinline__ void walk_prepare(jvec3 o, jvec3 d, int32_t skip, float limit, jvec3* inv_out, jvec3* dn_out, uint32_t* skipx_out) {
  const jvec3 inv = jv(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  *dn_out = jv_normalize(d);
  const bool exact = !(finite_f(inv.x) && finite_f(inv.y) && finite_f(inv.z)) || !finite_f(o.x) || !finite_f(o.y) || !finite_f(o.z);
  *skipx_out = (skip < 0 ? JADE_SKIP_MASK : (uint32_t)skip) | (exact ? 0x80000000u : 0u) | (limit == limit ? 0u : JADE_WANTS_POINT);
  *inv_out = inv;
}
// ... and walk_begin from such a record
static __device__ __forceinline__ void walk_begin_prepared(WalkState& r, const LdsStack& stk, const DevScene& S, jvec3 o, jvec3 inv, jvec3 dn, uint32_t skipx, float limit) {
  r.od.a = f2{o.x, o.y};
  r.od.b = f2{o.z, dn.x};
  r.od.c = f2{dn.y, dn.z};
  r.skipx = skipx;
  r.sp = stk.col;
  r.pushed = 0;
  r.cur = S.root_ref;
  r.inv = inv;
  lds_putf(stk, TW_BEST_DIST, JADE_INF_F);
  lds_put(stk, TW_BEST_SEQ, 0xffffffffu);
  lds_put(stk, TW_BEST_REF, 0xffffffffu);
  lds_st_v(stk.col + TW_FINISHED * JADE_COL_STRIDE, 0u);
  lds_putf(stk, TW_LIMIT, limit);
#if JADE_PREFETCH
  r.pre = node_fetch<JADE_TRACE_TOP_NODES>(r.cur, S, stk);
#endif
}
// The same ray once more, from the root, with nothing found yet (k_trace: an attempt over the cached subtrees that found no answer;
// a wide walk that met a tie).  The ray itself - origin, directions, source triangle, its limit in the column - stays as it is.
static __device__ __forceinline__ void walk_restart(WalkState& r, const LdsStack& stk, const DevScene& S) {
  r.sp = stk.col;
  r.pushed = 0;
  r.cur = S.root_ref;
  lds_putf(stk, TW_BEST_DIST, JADE_INF_F);
  lds_put(stk, TW_BEST_SEQ, 0xffffffffu);
  lds_put(stk, TW_BEST_REF, 0xffffffffu);
  lds_st_v(stk.col + TW_FINISHED * JADE_COL_STRIDE, 0u);
#if JADE_PREFETCH
  r.pre = node_fetch<JADE_TRACE_TOP_NODES>(r.cur, S, stk);
#endif
}
// One unit of the walk for a lane whose walk has not ended.  Returns the leaf met (0 = none): the caller queues it.
template <bool GENERAL>
static __device__ __forceinline__ uint32_t walk_step(WalkState& r, const DevScene& S, const LdsStack& stk, uint32_t& vcnt) {
#if JADE_PREFETCH
  const uint32_t leafv = node_decide<GENERAL, 0, TW_DUMMY>(r.pre, r.cur, r.sp, r.od, stk, true, vcnt, &r.inv);
  if (r.cur != JADE_REF_NONE) r.pre = node_fetch<JADE_TRACE_TOP_NODES>(r.cur, S, stk);
#else
  const uint32_t leafv = node_core<GENERAL, 0, TW_DUMMY, JADE_TRACE_TOP_NODES>(r.cur, r.sp, r.od, S, stk, true, vcnt, &r.inv);
#endif
  return (leafv & 15u) != 0 ? leafv : 0u;  // (an empty leaf cannot happen for a valid BVH)
}

// One WIDE unit of the walk ("Wide walk" above) for a lane whose walk has not ended.
static __device__ __forceinline__ uint32_t walk_step4(WalkState& r, const DevScene& S, const LdsStack& stk, uint32_t& vcnt) {
  const NodeRec4 nr = node_fetch4(r.cur, S, stk);
  const uint32_t leafv = node_decide4<TW_DUMMY>(nr, r.cur, r.sp, r.od, stk, vcnt, r.inv);
  return (leafv & 15u) != 0 ? leafv : 0u;
}

// One pair record of the item a lane holds: item_leaf = the leaf cursor (0 afterwards if the leaf is finished), meta =
// owner lane | sequence << 6, od / skip = the owner's ray (the caller's ds_bpermute), rec = the record (the caller's
// pair_load, issued before the ds_bpermutes so that the two latencies overlap), lane = this lane.  in_a / in_b: the origin
// projects into triangle A / B of the record - a candidate the caller queues (resolve_hit).
// cut: the owner's ray already has its answer (early exit) - the rest of the leaf is dropped untested.
static __device__ __forceinline__ void test_step(uint32_t& item_leaf, uint32_t meta, const RayOD& od, uint32_t skip, bool cut, const PairRec& rec,
                                                 const LdsStack& stk, int lane, uint32_t& tcnt, bool& in_a, bool& in_b) {
  const uint32_t leaf = item_leaf + 79u;  // next pair record (5 x 16 B), one pair fewer
  uint32_t idx_a;
  const uint32_t tcnt0 = tcnt;
  pair_core(od, skip, rec, tcnt, in_a, in_b, idx_a);
  if (cut) {
    tcnt = tcnt0;
    in_a = in_b = false;
  }
  const uint32_t ocol = stk.col + ((meta & 63u) - (uint32_t)lane) * 4u;  // the owner's column
  // the ray must not end before its candidates are resolved: each takes one off the finished-leaf count until then
  const uint32_t n_cand = (in_a ? 1u : 0u) + (in_b ? 1u : 0u);
  const bool fin = cut || (leaf & 15u) == 0;
  const uint32_t delta = (fin ? 1u : 0u) - n_cand;
  if (delta != 0u)
    __hip_atomic_fetch_add((jade_lds_u32*)(__SIZE_TYPE__)(ocol + TW_FINISHED * JADE_COL_STRIDE), delta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  item_leaf = fin ? 0u : leaf;
}
// A candidate: triangle (ref & 1) of the pair record at ref & ~15 against the ray of lane meta & 63 (od), a leaf the ray
// met as its (meta >> 6)-th.  The barycentric solve and the distance (:732-747), then hitArray's rule for the best hit
// (:787, strict "<" in the order leaves are met and triangles are indexed), which in terms of the candidates is: smaller
// distance, then earlier leaf, then earlier triangle of the leaf (ref grows with the index inside a leaf).
static __device__ __forceinline__ void resolve_hit(uint32_t ref, uint32_t meta, const RayOD& od, const DevScene& S, const LdsStack& stk, int lane, bool wide_kernel) {
  const float4* t0 = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.tverts) + (ref & 0x7ffffff0u));
  const uint32_t ocol = stk.col + ((meta & 63u) - (uint32_t)lane) * 4u;  // the owner's column
  const uint32_t seq = meta >> 6;
  float dist;
  jvec3 P;
  // The lanes that hold a candidate for the same ray take turns: each writes its number into the owner's lock word, the
  // one that reads its own number back goes first.  The loop runs until NO lane is left waiting (a ballot, the same for all
  // of them): with a per-lane exit, the winner's stores become loop-exit code, which a wave runs once all its lanes have
  // left the loop - the lanes of later turns would compare with a best hit not written yet.
  bool waiting = pair_hit(t0, (int)(ref & 1u), od, &dist, &P);
  while (__ballot(waiting) != 0ull) {
    if (waiting) {
      const float bd = jade_u2f(lds_ld_v(ocol + TW_BEST_DIST * JADE_COL_STRIDE));
      const uint32_t bs = lds_ld_v(ocol + TW_BEST_SEQ * JADE_COL_STRIDE);
      const uint32_t br = lds_ld_v(ocol + TW_BEST_REF * JADE_COL_STRIDE);
      // (wide walk: two leaves tie for the best distance - which of them the reference met first is not known: JADE_LIMIT_TIE)
      if (wide_kernel && dist == bd && seq != bs) lds_st_v(ocol + TW_LIMIT * JADE_COL_STRIDE, JADE_LIMIT_TIE);
      if (!(dist < bd || (dist == bd && (seq < bs || (seq == bs && ref < br))))) {
        waiting = false;
      } else {
        lds_st_v(ocol + TW_DUMMY * JADE_COL_STRIDE, (uint32_t)lane);
        if (lds_ld_v(ocol + TW_DUMMY * JADE_COL_STRIDE) == (uint32_t)lane) {
          lds_st_v(ocol + TW_BEST_DIST * JADE_COL_STRIDE, jade_f2u(dist));
          lds_st_v(ocol + TW_BEST_SEQ * JADE_COL_STRIDE, seq);
          lds_st_v(ocol + TW_BEST_REF * JADE_COL_STRIDE, ref);
          waiting = false;
        }
      }
    }
  }
  __hip_atomic_fetch_add((jade_lds_u32*)(__SIZE_TYPE__)(ocol + TW_FINISHED * JADE_COL_STRIDE), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
// The hit a finished ray reports: triangle index (-1 = miss), distance and hit point of the winning triangle.
// parent1: the winning triangle's leaf's parent + 1 (the pair record's flag word, bits 1-31; 0 = none, or no hit)
// want_point = false: a yes/no query - the triangle alone (one 16-B word of the record instead of the whole solve)
static __device__ __forceinline__ int32_t walk_result(const LdsStack& stk, const DevScene& S, const RayOD& od, float* dist, jvec3* P, uint32_t* parent1 = nullptr,
                                                      bool want_point = true) {
  const uint32_t ref = lds_get(stk, TW_BEST_REF);
  *dist = lds_getf(stk, TW_BEST_DIST);
  if (parent1) *parent1 = 0u;
  if (ref == 0xffffffffu) return -1;
  const float4* t0 = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.tverts) + (ref & 0x7ffffff0u));
  const int k = (int)(ref & 1u);
  if (want_point) {
    float d2;
    pair_hit(t0, k, od, &d2, P);
  }
  const float4 tag = t0[4];
  if (parent1) *parent1 = jade_f2u(tag.w) >> 1;
  return (int32_t)(jade_f2u(tag.z) + (uint32_t)k);
}

// JADE_COST_NODE / JADE_COST_TRI: instructions issued by a walk unit / a test unit (the kind that advances more lanes
// per instruction issued runs); JADE_STEPS_PER_PICK: units of the picked kind per wave iteration.
#ifndef JADE_STEPS_PER_PICK
#define JADE_STEPS_PER_PICK 4
#endif
#ifndef JADE_COST_NODE
#define JADE_COST_NODE 100
#endif
#ifndef JADE_COST_TRI
#define JADE_COST_TRI 120
#endif

// A wave's side of the scheme above: its two rings, the item each lane is testing, and one iteration of work.  Used by
// k_trace (rays from the queue) and k_light (rays from the lane's own path): everything in here is wave-uniform control
// flow around the per-lane steps.
#ifndef JADE_COOP_LEAF
#define JADE_COOP_LEAF 0 /* 1: development variant (north_star: "triangle SoA staged through LDS"): a test unit's 64 pair records are fetched by the wave cooperatively - lanes 5r .. 5r+4 read the five 16-B parts of lane r's record, 80 contiguous bytes - into a 5 KB stage in LDS, from which every lane reads its own record.  Measured, not shipped */
#endif
struct WaveTrace {
#if JADE_COOP_LEAF
  uint32_t stage;                // LDS byte address of this wave's 64 x 80-B stage
#endif
  uint32_t wq, hq;               // LDS byte addresses of the ring of leaves and of the ring of candidates
  uint32_t q_head, q_count;      // leaves waiting (wave-uniform)
  uint32_t h_head, h_count;      // candidates waiting (wave-uniform)
  uint32_t item_leaf, item_meta; // the leaf this lane is testing (cursor, 0 = none) and its owner | sequence << 6
  int lane;
  bool wide_kernel;              // the kernel may walk with wide units: ties are marked (resolve_hit)

  __device__ __forceinline__ void init(uint32_t wq_addr, uint32_t hq_addr, int lane_, bool wide_kernel_ = false) {
    wide_kernel = wide_kernel_;
    wq = wq_addr;
    hq = hq_addr;
    q_head = q_count = h_head = h_count = 0;
    item_leaf = item_meta = 0;
    lane = lane_;
  }
  // lanes of m below this one: v_mbcnt_lo + v_mbcnt_hi - two instructions and no register (until round 4's last day: two masks kept per
  // lane, two ands, two popcounts and an add - at the 96-VGPR limit of k_trace)
  static __device__ __forceinline__ uint32_t rank_in(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  }
  // the ray of lane `owner`: six registers of that lane (every lane must execute this: ds_bpermute returns 0 for a source
  // lane that is masked off)
  static __device__ __forceinline__ RayOD ray_of(const WalkState& r, int owner) {
    RayOD od;
    od.a.x = __shfl(r.od.a.x, owner, 64);
    od.a.y = __shfl(r.od.a.y, owner, 64);
    od.b.x = __shfl(r.od.b.x, owner, 64);
    od.b.y = __shfl(r.od.b.y, owner, 64);
    od.c.x = __shfl(r.od.c.x, owner, 64);
    od.c.y = __shfl(r.od.c.y, owner, 64);
    return od;
  }
  // resolve up to 64 of the waiting candidates, one per lane
  __device__ __forceinline__ void resolve_pass(WalkState& r, const DevScene& S, const LdsStack& stk) {
    const uint32_t nres = h_count < 64u ? h_count : 64u;
    const bool mine = (uint32_t)lane < nres;
    uint32_t ref = 0, meta = 0;
    if (mine) lds_ld64(hq + ((h_head + (uint32_t)lane) & (JADE_HQ - 1u)) * 8u, ref, meta);
    h_head = (h_head + nres) & (JADE_HQ - 1u);
    h_count -= nres;
    const RayOD od = ray_of(r, (int)(meta & 63u));
    if (mine) resolve_hit(ref, meta, od, S, stk, lane, wide_kernel);
    // Early exit (JADE_WALK_EARLY_EXIT): a ray whose best hit is now nearer than its limit has its answer - its walk ends
    // here (stack dropped) and the leaves it has pushed but that no lane has taken yet are dropped unread (JADE_CUT in
    // skipx, seen by whoever takes one).  A lane without a ray has cur == NONE and sp at its base already.
    if (lds_getf(stk, TW_BEST_DIST) < lds_getf(stk, TW_LIMIT)) {
      r.cur = JADE_REF_NONE;
      r.sp = stk.col;
      r.skipx |= JADE_CUT;
    }
  }
  // a lane's ray has ended: its walk has, and every leaf it pushed has been finished (candidates included)
  static __device__ __forceinline__ bool ray_ended(const WalkState& r, const LdsStack& stk) {
    return r.cur == JADE_REF_NONE && lds_ld_v(stk.col + TW_FINISHED * JADE_COL_STRIDE) == r.pushed;
  }
  // One iteration: one kind of work for the wave - the walk, for the lanes whose ray still walks (`active`: this lane
  // holds a ray), or triangle tests, for as many lanes as there are leaves waiting.  The kind that advances more lanes per
  // instruction issued runs; the walk needs room for the 64 leaves one unit of it can push.  When every ray in flight only
  // waits for candidates (fewer than a batch), they are resolved.
  // WIDE: the kernel walks with wide units ("Wide walk") wherever every walking lane of the wave may.
  template <bool WIDE>
  __device__ __forceinline__ void iterate(WalkState& r, bool active, const DevScene& S, const LdsStack& stk, uint32_t& vcnt, uint32_t& tcnt, TraceProf& pr) {
    const int nw = __popcll(__ballot(active && r.cur != JADE_REF_NONE));
    const uint32_t n_items = q_count + (uint32_t)__popcll(__ballot(item_leaf != 0));
    const uint32_t nt = n_items < 64u ? n_items : 64u;
    PROF_COUNT(pr, PC_ITER, 1);
    PROF_LAP(pr, PL_PICK);
    if (nw == 0 && n_items == 0) {
      if (h_count != 0) {
        resolve_pass(r, S, stk);
        PROF_DRAIN();
        PROF_COUNT(pr, PC_RESOLVES, 1);
        PROF_LAP(pr, PL_RESOLVE);
      }
      return;
    }
    if (q_count <= JADE_WQ - 64 && nw > 0 && JADE_COST_TRI * (uint32_t)nw >= JADE_COST_NODE * nt) {
      const bool general = S.general_walk || __ballot(active && (int32_t)r.skipx < 0) != 0ull;  // per WAVE (node_core)
      const bool wide = WIDE && !general && __ballot(active && (r.skipx & JADE_FORCE_BINARY) != 0u) == 0ull;
#pragma nounroll
      for (int rep = 0; rep < JADE_STEPS_PER_PICK; ++rep) {
        if (q_count > JADE_WQ - 64) break;
        const bool go = active && r.cur != JADE_REF_NONE;
        uint32_t leafv = 0;
#if JADE_TRACE_PROFILE
        {  // the walk unit in two laps: the record's fetch (until the data is in registers), then everything else
          PROF_COUNT(pr, PC_WALK_UNITS, 1);
          PROF_COUNT(pr, PC_WALK_LANES, (unsigned long long)__popcll(__ballot(go)));
          PROF_COUNT(pr, PC_WALK_WAIT, (unsigned long long)__popcll(__ballot(active && !go)));
          PROF_COUNT(pr, PC_WALK_FREE, (unsigned long long)__popcll(__ballot(!active)));
          if (WIDE && wide) {
            NodeRec4 n4;
            if (go) n4 = node_fetch4(r.cur, S, stk);
            PROF_DRAIN();
            PROF_LAP(pr, PL_WALK_LOAD);
            if (go) leafv = node_decide4<TW_DUMMY>(n4, r.cur, r.sp, r.od, stk, vcnt, r.inv);
          } else {
            NodeRec nr;
            if (go) nr = node_fetch<JADE_TRACE_TOP_NODES>(r.cur, S, stk);
            PROF_DRAIN();
            PROF_LAP(pr, PL_WALK_LOAD);
            if (general) {
              if (go) leafv = node_decide<true, 0, TW_DUMMY>(nr, r.cur, r.sp, r.od, stk, true, vcnt, &r.inv);
            } else {
              if (go) leafv = node_decide<false, 0, TW_DUMMY>(nr, r.cur, r.sp, r.od, stk, true, vcnt, &r.inv);
            }
          }
          leafv = (leafv & 15u) != 0 ? leafv : 0u;
          PROF_DRAIN();
          PROF_LAP(pr, PL_WALK_MATH);
        }
#else
        if (WIDE && wide) {
          if (go) leafv = walk_step4(r, S, stk, vcnt);
        } else if (general) {
          if (go) leafv = walk_step<true>(r, S, stk, vcnt);
        } else {
          if (go) leafv = walk_step<false>(r, S, stk, vcnt);
        }
#endif
        // the leaves met by this unit, in lane order (any order would do: a leaf's place among its ray's leaves is its
        // sequence number)
        const unsigned long long m = __ballot(leafv != 0);
        if (m != 0ull) {
          if (leafv != 0) {
            lds_st64(wq + ((q_head + q_count + rank_in(m)) & (JADE_WQ - 1u)) * 8u, leafv, (uint32_t)lane | (r.pushed << 6));
            r.pushed += 1u;
          }
          q_count += (uint32_t)__popcll(m);
        }
        PROF_DRAIN();
        PROF_LAP(pr, PL_WALK_RING);
      }
    } else {
#pragma nounroll
      for (int rep = 0; rep < JADE_STEPS_PER_PICK; ++rep) {
        // lanes without an item take the next ones from the ring
        const unsigned long long need = __ballot(item_leaf == 0);
        if (q_count != 0 && need != 0ull) {
          const uint32_t rk = rank_in(need);
          if (item_leaf == 0 && rk < q_count) lds_ld64(wq + ((q_head + rk) & (JADE_WQ - 1u)) * 8u, item_leaf, item_meta);
          const uint32_t want = (uint32_t)__popcll(need);
          const uint32_t npop = want < q_count ? want : q_count;
          q_head = (q_head + npop) & (JADE_WQ - 1u);
          q_count -= npop;
        }
        const bool go = item_leaf != 0;
        PROF_DRAIN();
        PROF_LAP(pr, PL_TEST_POP);
        if (__ballot(go) == 0ull) break;
        PROF_COUNT(pr, PC_TEST_UNITS, 1);
        PROF_COUNT(pr, PC_TEST_LANES, (unsigned long long)__popcll(__ballot(go)));
        PROF_COUNT(pr, PC_TEST_FULL, __popcll(__ballot(go)) >= 56 ? 1ull : 0ull);
        PROF_COUNT(pr, PC_TEST_THIN, __popcll(__ballot(go)) <= 16 ? 1ull : 0ull);
        PROF_COUNT(pr, PC_TEST_THIN_LANES, __popcll(__ballot(go)) <= 16 ? (unsigned long long)__popcll(__ballot(go)) : 0ull);
        // the record first (it depends on the item alone), then the ray the item belongs to
        const uint32_t off = item_leaf & 0x7ffffff0u;
        PairRec rec;
#if JADE_COOP_LEAF
        {
          const uint32_t myoff = go ? off : 0u;  // (a lane without an item stages record 0: never read)
#pragma unroll
          for (uint32_t j = 0; j < 5u; ++j) {
            const uint32_t c = j * 64u + (uint32_t)lane, rr = c / 5u, part = c - 5u * rr;
            const uint32_t o_r = (uint32_t)__shfl((int)myoff, (int)rr, 64);
            const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.tverts) + o_r + 16u * part);
            typedef float coop_f4 __attribute__((ext_vector_type(4)));
            const coop_f4 nv = {v.x, v.y, v.z, v.w};
            *(__attribute__((address_space(3))) coop_f4*)(__SIZE_TYPE__)(stage + 16u * c) = nv;
          }
          typedef float coop_f4 __attribute__((ext_vector_type(4)));
          const __attribute__((address_space(3))) coop_f4* mine = (const __attribute__((address_space(3))) coop_f4*)(__SIZE_TYPE__)(stage + 80u * (uint32_t)lane);
          const coop_f4 m0 = mine[0], m1 = mine[1], m2 = mine[2], m3 = mine[3], m4 = mine[4];
          rec.q0 = make_float4(m0.x, m0.y, m0.z, m0.w);
          rec.q1 = make_float4(m1.x, m1.y, m1.z, m1.w);
          rec.q2 = make_float4(m2.x, m2.y, m2.z, m2.w);
          rec.q3 = make_float4(m3.x, m3.y, m3.z, m3.w);
          rec.q4 = make_float4(m4.x, m4.y, m4.z, m4.w);
        }
#else
        if (go) rec = pair_load(S, off);
#endif
        PROF_DRAIN();
        PROF_LAP(pr, PL_TEST_LOAD);
        const int owner = (int)(item_meta & 63u);
        const RayOD od = ray_of(r, owner);
        const uint32_t sx = (uint32_t)__shfl((int)r.skipx, owner, 64);
        const uint32_t skip = sx & JADE_SKIP_MASK;
        PROF_DRAIN();
        PROF_LAP(pr, PL_TEST_RAY);
        bool in_a = false, in_b = false;
        if (go) test_step(item_leaf, item_meta, od, skip, (sx & JADE_CUT) != 0u, rec, stk, lane, tcnt, in_a, in_b);
        PROF_DRAIN();
        PROF_LAP(pr, PL_TEST_MATH);
        // candidates (the origin projects into the triangle) are resolved later, many at a time
        const unsigned long long ma = __ballot(in_a), mb = __ballot(in_b);
        if (ma != 0ull) {
          if (h_count > JADE_HQ - 64) resolve_pass(r, S, stk);
          if (in_a) lds_st64(hq + ((h_head + h_count + rank_in(ma)) & (JADE_HQ - 1u)) * 8u, off, item_meta);
          h_count += (uint32_t)__popcll(ma);
        }
        if (mb != 0ull) {
          if (h_count > JADE_HQ - 64) resolve_pass(r, S, stk);
          if (in_b) lds_st64(hq + ((h_head + h_count + rank_in(mb)) & (JADE_HQ - 1u)) * 8u, off | 1u, item_meta);
          h_count += (uint32_t)__popcll(mb);
        }
        PROF_DRAIN();
        PROF_LAP(pr, PL_TEST_CAND);
      }
      if (h_count >= JADE_HQ_BATCH) {
        resolve_pass(r, S, stk);
        PROF_DRAIN();
        PROF_COUNT(pr, PC_RESOLVES, 1);
        PROF_LAP(pr, PL_RESOLVE);
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------
// Packet form (k_light, round 3): the 64 rays of a wave walk the tree TOGETHER.
//
// The reference never prunes against the best hit (PathTrace.cu:806-856), so the SET of nodes and leaves a ray visits does
// not depend on the order of the walk: it is every node whose box - and whose ancestors' boxes - the ray enters ("slab
// value > 0", :770, :835-855).  Only hitArray's tie rule sees the order (strict "<": of two equal distances the one met
// first wins, :787, :816).  Camera rays of a 16 x 4 block of pixels, and their reflections off the flat mirror floor, enter
// nearly the same nodes; walked per lane (RayState / WalkState above) every one of them pays for its own stack, its own
// cursor and its own gather of the same record.  Here the wave keeps ONE cursor and ONE stack, both scalar:
//   * a node's record is read once per wave with scalar loads (constant address space: s_load, served by the scalar cache)
//     and every lane whose ray entered that node tests both children with the record in SGPR operands - the same
//     statements (slab2), so the same slab values, V counted per lane exactly as before;
//   * a stack entry is {child reference, the lanes that entered it}: the wave goes left first and pushes the right child
//     whenever some lane entered it.  No per-lane stack, no leaf FIFO;
//   * a leaf is tested by the lanes that entered it, the pair record again through scalar loads (pair_core unchanged);
//   * the order a ray itself would have met its leaves in is NOT kept (until round 3's last day it was, as a 64-bit key per lane
//     with one bit per level - a fifth of a node visit's instructions, for the benefit of twin geometry): inside a leaf the index
//     order and strict "<" decide as always, and when two LEAVES give a ray the same best distance the packet is given up - its
//     rays go to the wavefront passes (k_trace), which walk them in the reference's order.
// Everything wave-uniform lives in SGPRs; the stack is a small LDS array written by lane 0 and read as a broadcast.
// ---------------------------------------------------------------------------------------------------------------
#define JADE_PACKET_MAX_DEPTH 63
typedef float jade_nf4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const jade_nf4 jade_const_f4;  // constant address space + uniform address => scalar loads
static __device__ __forceinline__ float4 ld_const_f4(jade_const_f4* p) {
  const jade_nf4 v = *p;
  return make_float4(v.x, v.y, v.z, v.w);
}
struct PacketBest {
  float dist;      // INF = no hit yet
  uint32_t leaf;   // the leaf the winning triangle sits in (byte offset of its first pair record)
  uint32_t index;  // triangle index
  jvec3 point;
};
static __device__ __forceinline__ jade_const_f4* as_const_f4(const void* base, uint32_t byte_off) {
  return reinterpret_cast<jade_const_f4*>(reinterpret_cast<__SIZE_TYPE__>(reinterpret_cast<const char*>(base) + byte_off));
}
// stack entry k of the wave at LDS byte address stack + 32 k: {ref, -, lanes lo, lanes hi, ...}
static __device__ __forceinline__ void packet_push(uint32_t stack, uint32_t sp, int lane, uint32_t ref, unsigned long long lanes) {
  if (lane == 0) {
    const uint32_t a = stack + 32u * sp;
    lds_st64(a, ref, 0u);
    lds_st64(a + 8u, (uint32_t)lanes, (uint32_t)(lanes >> 32));
  }
}
static __device__ __forceinline__ void packet_pop(uint32_t stack, uint32_t sp, uint32_t& ref, unsigned long long& lanes) {
  const uint32_t a = stack + 32u * sp;  // the same address in every lane: a broadcast read
  uint32_t r, d, l0, l1;
  lds_ld64(a, r, d);
  lds_ld64(a + 8u, l0, l1);
  // (readfirstlane returns an int: through uint32_t, or the low word sign-extends into the lanes 32-63)
  ref = (uint32_t)__builtin_amdgcn_readfirstlane(r);
  lanes = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane(l0) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane(l1) << 32);
}

// hitBVH (PathTrace.cu:795-859) for the rays of the lanes with `active`, all lanes of the wave taking part in the control
// flow.  o, d, skip: this lane's ray; vcnt / tcnt: this lane's counts of child records visited and triangles tested (the
// root record is the caller's, as with ray_begin).  stack: LDS byte address of JADE_PACKET_MAX_DEPTH + 1 entries of 32 bytes
// for this wave.  best: this lane's result (index 0xffffffff = miss).
// budget: node and pair records the packet may read; past it the walk is given up and false returned (vcnt / tcnt / best
// then hold a partial walk: the caller discards them and walks these rays per lane).  A packet pays one record per node of
// the UNION of its rays' walks: rays that fan out into a finely tessellated object (64 lanes, 64 leaves, each reached by one
// lane) cost it 64 walks of one lane each, every record a dependent scalar load.
template <bool GENERAL>
static __device__ __forceinline__ bool packet_trace(const DevScene& S, uint32_t stack, int lane, bool active, jvec3 o, jvec3 d, int32_t skip,
                                                    uint32_t& vcnt, uint32_t& tcnt, PacketBest& best, uint32_t budget, TraceProf& pr) {
  RayOD od;
  const jvec3 inv = jv(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  const jvec3 dn = jv_normalize(d);
  od.a = f2{o.x, o.y};
  od.b = f2{o.z, dn.x};
  od.c = f2{dn.y, dn.z};
  const uint32_t skipu = skip < 0 ? 0x7fffffffu : (uint32_t)skip;
  best.dist = JADE_INF_F;
  best.leaf = 0u;
  best.index = 0xffffffffu;
  best.point = jv(0, 0, 0);
  const unsigned long long me = 1ull << lane;
  uint32_t cur = S.root_ref, sp = 0;
  unsigned long long lanes = __ballot(active);
  if (lanes == 0ull) return true;
  uint32_t records = 0;
  bool tie = false;
  PROF_LAP(pr, PKL_PREP);
  for (;;) {
    if (records > budget) return false;
    const bool in = (lanes & me) != 0ull;
    if ((int32_t)cur < 0) {
      // ---- a leaf: its pair records, one after the other, for the lanes that entered it (hitArray, :776-792)
      const uint32_t leaf_id = cur & 0x7ffffff0u;
      uint32_t off = leaf_id;
      records += cur & 15u;
      for (uint32_t n = cur & 15u; n != 0u; --n, off += 80u) {
        jade_const_f4* t = as_const_f4(S.tverts, off);
        PairRec rec;
        rec.q0 = ld_const_f4(t);
        rec.q1 = ld_const_f4(t + 1);
        rec.q2 = ld_const_f4(t + 2);
        rec.q3 = ld_const_f4(t + 3);
        rec.q4 = ld_const_f4(t + 4);
        bool in_a = false, in_b = false;
        uint32_t idx_a = 0;
        if (in) pair_core(od, skipu, rec, tcnt, in_a, in_b, idx_a);
        PROF_COUNT(pr, PKC_PAIRS, 1);
        PROF_COUNT(pr, PKC_PAIR_LANES, (unsigned long long)__popcll(lanes));
        PROF_COUNT(pr, PKC_SOLVES, __ballot(in_a || in_b) != 0ull ? 1ull : 0ull);
        PROF_COUNT(pr, PKC_SOLVE_LANES, (unsigned long long)(__popcll(__ballot(in_a)) + __popcll(__ballot(in_b))));
        PROF_LAP(pr, PKL_LEAF);
        if (in_a || in_b) {  // the origin projects into a triangle: one test in ten; A before B (index order, strict "<")
          const float4* tv = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.tverts) + off);
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            if (k == 0 ? in_a : in_b) {
              float dist;
              jvec3 P;
              if (pair_hit(tv, k, od, &dist, &P)) {
                // strict "<" (:787): inside a leaf the earlier index keeps an equal distance.  Two LEAVES at the same distance: the
                // one the ray itself would have met first wins, and a packet does not know which that is - it is given up (below)
                tie = tie || (dist == best.dist && best.leaf != leaf_id);
                if (dist < best.dist) {
                  best.dist = dist;
                  best.leaf = leaf_id;
                  best.index = idx_a + (uint32_t)k;
                  best.point = P;
                }
              }
            }
          }
        }
        PROF_LAP(pr, PKL_SOLVE);  // (everything between a pair's inside test and here: the solves of the lanes with a candidate)
      }
      if (__ballot(tie) != 0ull) return false;  // (twin geometry; the wavefront passes walk these rays in the reference's order)
    } else {
      // ---- an internal node: both children's boxes from its record (hitAABB x 2, :825-832)
      records += 1u;
      jade_const_f4* nd = as_const_f4(S.nodes, cur * 64u);
      const float4 a = ld_const_f4(nd), b = ld_const_f4(nd + 1), c = ld_const_f4(nd + 2), r4 = ld_const_f4(nd + 3);
      const uint32_t left = jade_f2u(r4.x), right = jade_f2u(r4.y);
      Slab s1, s2;
      slab2(od, inv, a, b, c, GENERAL, &s1, &s2);
      bool in1, in2;
      if (GENERAL) {
        const bool c1 = left != JADE_REF_NONE, c2 = right != JADE_REF_NONE;  // a missing child is neither counted nor entered
        if (in) vcnt += (c1 ? 1u : 0u) + (c2 ? 1u : 0u);
        in1 = in && c1 && slab_met(s1);
        in2 = in && c2 && slab_met(s2);
      } else {
        if (in) vcnt += 2u;
        in1 = in && slab_met(s1);
        in2 = in && slab_met(s2);
      }
      const unsigned long long m1 = __ballot(in1), m2 = __ballot(in2);
      PROF_COUNT(pr, PKC_NODES, 1);
      PROF_COUNT(pr, PKC_NODE_LANES, (unsigned long long)__popcll(lanes));
      PROF_LAP(pr, PKL_NODE);
      if (m1 != 0ull) {
        if (m2 != 0ull) {
          packet_push(stack, sp, lane, right, m2);
          sp += 1u;
        }
        cur = left;
        lanes = m1;
        continue;
      }
      if (m2 != 0ull) {
        cur = right;
        lanes = m2;
        continue;
      }
    }
    // ---- nothing below: the next deferred child
    if (sp == 0u) break;
    sp -= 1u;
    packet_pop(stack, sp, cur, lanes);
    PROF_LAP(pr, PKL_POP);  // (with the push that preceded it, if any)
  }
  return true;
}

// jade_device.h — device-side data layout of the HIP module (gfx950).
//
// The boundary arrays (include/jade_rt.h) arrive in the reference's AoS
// layouts (Triangle_cu 112 B, BVHNode_cu 40 B).  The traversal kernel does not
// read those: jade_scene_create re-lays the geometry out as
//
//   node records   64 B per INTERNAL node, children's boxes stored in the parent and
//                  interleaved so that the same coordinate of the left and the right box
//                  is one aligned register pair (the two slab tests run as packed fp32):
//                  float4 {L.aa.x, R.aa.x, L.aa.y, R.aa.y} {L.aa.z, R.aa.z, L.bb.x, R.bb.x}
//                  {L.bb.y, R.bb.y, L.bb.z, R.bb.z} + uint4 {left_ref, right_ref, 0, 0}
//                  -> one visit = 4 loads from one 64-B aligned line, instead of the
//                  reference's 3 x 40-B records (parent, and each child read again when
//                  popped: PathTrace.cu:809/826/830)
//   child refs     internal: index into the compacted node array;
//                  leaf: 0x80000000 | 3 * first_triangle << 4 | n  (n <= 15);
//                  JADE_REF_NONE for the reference's "child 0"
//   vertex records 48 B per triangle, p1 and p2 interleaved for the same reason:
//                  {p1.x, p2.x, p1.y, p2.y} {p1.z, p2.z, p3.x, p3.y} {p3.z, pad x3};
//                  the 76 B of material data never enter the traversal cache footprint
//
// Shading reads the untouched 112-B records (they are needed once per
// shaded vertex, not once per intersection test).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jade_fpmath.h"
#include "jade_rt.h"

#define JADE_REF_LEAF 0x80000000u
#define JADE_REF_NONE 0x7fffffffu
#define JADE_MAX_TRIS ((1 << 27) / 3) /* leaf refs carry the byte offset / 16 of an 80-B pair record in bits 4-30 (jade_trace.h; jade_scene_create checks the pair count) */
#define JADE_MAX_LEAF 15
#define JADE_SKIP_CAMERA (-2) /* PathState.skip: camera ray (no source triangle, origin = PathState.eye) */
#define JADE_INF_F 2147483647.0f /* #define INF, PathTrace.cu:23 */

#ifndef JADE_LDS_STACK
#define JADE_LDS_STACK 8 /* traversal stack entries kept in LDS per lane; deeper ones spill to global.  The deepest \
                            stack of any ray is 12 on C3 and C5 (tools/ray_histogram.py), 99.6 % need <= 8 */
#endif
#ifndef JADE_LDS_FIFO
#define JADE_LDS_FIFO 4
#endif
/* k_light: leaf cursors waiting for their triangle tests, same LDS column (power of two) */
#define JADE_LDS_STATE 9  /* k_light: ray-state words, same column (jade_trace.h): 21 words per lane */
// k_trace and k_light run 256-thread workgroups, 4 per CU; each block stages the top of the BVH in LDS once per launch
// (the node records with the largest boxes: jade_scene_create orders the internal nodes by box area so that any prefix
// is a connected top).  A visit of such a node is four ds_read_b128 of a plane layout instead of four 16-B gathers
// through the vector-memory path: 40 % of k_trace's node visits on C3 with 160 nodes.  Measured: a
// few per cent, with the FIFO form of k_trace as with the final one; one
// 1024-thread workgroup per CU with 768-1 272 nodes staged did not pay either time.
#ifndef JADE_TRACE_BLOCK
#define JADE_TRACE_BLOCK 256
#endif
#ifndef JADE_TRACE_TOP_NODES
#define JADE_TRACE_TOP_NODES 144 /* k_trace's share of the top (<= JADE_LDS_TOP_NODES): 14 KB of columns + 8 KB of rings + 9 KB of nodes = 31 744 B per block, 5 blocks per CU.  With 160 nodes the block is 32 768 B and 5 x that is the CU's 160 KB to the byte - but only 4 were resident (LDS is handed out in granules: 118.1 vs 121.7 ms of k_trace per 256-spp step of C3, 835 vs 869 on the close-up, once the early-exit limit had taken a 14th column word).  Round 2: 96 nodes the same speed on C3, 5 % slower on C5; 224: no better */
#endif
#ifndef JADE_LDS_TOP_NODES  /* 0 = no staging (A/B builds); k_light: 21 KB of columns + 10 KB of nodes per block */
#define JADE_LDS_TOP_NODES 160 /* k_light's; k_trace per 1024-spp step of C3 (round 2, FIFO form): 570 ms without, 552 with 80, 545 with 160 */
#endif
#define JADE_RECORD_MEMORY 0.60 /* share of the free device memory that path records + partial sums may take: paths in \
                                  flight are what fills the wide passes (1080p on one GPU: 32 -> 256 records per pixel = +21 %) */

// What shading needs of a triangle besides its vertices (round 3).  The reference copies an object's material into every
// one of its triangles (PathTrace.cu:451: 76 of the 112 bytes of a Triangle_cu), and k_shade - bound by the number of 64-B
// sectors it touches - read two to three sectors of such a record at every vertex, hit and exit point, most of them misses
// (the triangles a path meets are all over the statue).  jade_scene_create keeps the distinct {object, material} tuples in a
// table (a handful of entries for any scene the reference can load: L1-resident) and 16 bytes per triangle: the flat normal
// and the tuple's number.  The values are the bytes of the caller's records; only where they are read from differs.
struct DevMaterial {  // 64 bytes; field names as in jade_triangle
  float emissive[3];
  float brdf[3];
  int32_t reflex_mode, refract_mode;
  float refract_rate[3];
  float refract_albedo[3];
  float refract_index;
  int32_t obj_idx;
};

// The four-wide form of the walk (round 3, jade_trace.h "Wide walk"): with early exits a visit tests a node's four GRANDCHILDREN.
// nodes4: 8 x float4 per internal node, same numbering as `nodes`: {half 0: the three box float4 of the left child's record}
// {half 1: the right child's} {four references} {unused}; a child that is a leaf fills its half with its own box twice and the
// references (leaf, JADE_REF_NONE).  Null when the tree has a missing child (general_walk).
#ifndef JADE_WIDE_WALK
#define JADE_WIDE_WALK 1
#endif
#ifndef JADE_TRACE_TOP4
#define JADE_TRACE_TOP4 72 /* wide records k_trace stages in LDS: 7 planes x 72 x 16 B = 8 064 B */
#endif
struct DevScene {
  const float4* nodes;        // 4 x float4 per internal node
  const float4* nodes4;       // 8 x float4 per internal node (see above), or null
  const float4* tverts;       // 5 x float4 per pair of triangles of a leaf (jade_trace.h)
  const jade_triangle* tris;  // the caller's records (BVH order): read for vertices only (emitters, the BSSRDF exit triangle)
  const float4* tnorm;        // per triangle {flat normal, number of its DevMaterial (uint bits)}
  const DevMaterial* mats;
  const int32_t* emit;
  const int32_t* mapping;
  const float* prefix;
  const jade_obj_seg* segs;
  const float* env;           // interleaved RGB, row 0 = top
  int32_t env_w, env_h;
  int32_t n_tris, n_emit;
  uint32_t root_ref;
  uint32_t top_k;             // internal nodes [0, top_k) are the ones k_trace stages in LDS (largest boxes first)
  // BSSRDF exit-point search (jade_shade.h, begin_bounce): guide_obj[o] = {first entry of object o's guide, cells Gn (a power of
  // two; 0 = no guide, search as the reference does)}; guide[first + c] = the first triangle i (original order) of the object
  // with fl(c / Gn * A) <= prefix[i], c = 0 .. Gn + 1
  const uint32_t* guide;
  const uint2* guide_obj;
  uint32_t general_walk;      // an internal node lacks a child (the reference's "child 0"): every wave takes the general node step (jade_trace.h)
  // Occluder cache (jade_trace.h): JADE_ANYHIT_KEYS entries per triangle, four internal-node references + 1 each (0 = empty); null =
  // the tree does not allow it (boxes not nested, a missing child, too deep).  The one piece of the scene the kernels WRITE: hints
  // only, a stale or torn entry costs a walk and never an answer.
  uint4* anyhit;
  // Environment importance sampling (jade_render_params.env_sampling, non-parity): an alias table over the map's texels, weight =
  // (luminance + a floor) x sin(theta of the row); per texel {acceptance threshold, alias texel, this texel's pdf x W x H, the
  // alias texel's} - one 16-B gather per draw (jade_shade.h, env_sample)
  const uint4* env_alias;
};

// Path records, structure of arrays.  Samples are independent work items
// (jade_rt.h, JADE_SAMPLE_LANES): record r = m * npx + pixel (m < rpp) carries
// the samples s = m, m + rpp, m + 2 rpp, ... of owned pixel `pixel`
// (pixel = owned_tile * 256 + ly * 16 + lx), one after the other, so a rank
// has npix = npx * rpp paths in flight.  rpp (records per pixel, a power of two
// <= JADE_SAMPLE_LANES) is sized per render from the free device memory
// (JADE_RECORD_MEMORY): 256 for a full 1080p frame on one GPU, 1024 for an
// eighth of it - hundreds of millions of paths in flight per GPU whatever the
// tile partition.  Sample s adds
// into partial sum s % JADE_SAMPLE_LANES of its pixel, which only record
// s % rpp ever touches, in increasing s: no atomics, and a result that does
// not depend on rpp.  `nslots` = n_emit + 2 ray slots per record: [0, n_emit)
// shadow rays, n_emit = environment-visibility ray, n_emit + 1 = indirect
// ray; single-ray stages use slot 0.
//
// Which pixel a record works on rotates: its n-th sample inside a block of
// JADE_SAMPLE_LANES samples goes to pixel (home + n * stride) % npx.  Every
// (pixel, sample) is still covered exactly once (for fixed n the map is a
// bijection on pixels), but the few pixels whose samples are 10x more expensive
// (the statue) no longer queue all their samples on the same records.  Inside
// one block each (pixel, lane) partial sum is touched by exactly one record, so
// the host runs a render block by block and the result stays bit-identical.
struct PathState {
  int32_t npix;         // number of path records (= npx * rpp)
  int32_t npx;          // owned pixels (tile-padded)
  int32_t rpp;          // records per pixel
  int32_t stride;       // pixel rotation per sample (see above)
  int32_t nslots;
  uint4* hdr;           // [npix] what every pass of every record reads, one 16-B word: {rng: Wang-hash state of the sample in
                        // flight, done: samples this record has finished, stage | depth << 8 | flags << 16, 0}.  Read in record
                        // order by k_light (coalesced); for k_shade, whose records are scattered, one sector instead of three
  int32_t sum_lanes;    // partial sums kept per pixel: min(JADE_SAMPLE_LANES, announced spp rounded up to a power of two) >= rpp
  float* sum;           // [sum_lanes * npx][3] partial radiance sums per (lane, pixel), RGB side by side; sample s adds into lane s % JADE_SAMPLE_LANES
  // The context of a path in flight, four float4 = one aligned 64-B sector per record (only k_shade and a record k_light parks
  // touch it; what every pass of every record reads - rng, done, stage - is the header):
  //   {thr.xyz, obj} {acc.xyz, out.z} {le.xyz, src.x} {src.y, src.z, out.x, out.y}
  // thr = throughput (product of pushed rates), acc = radiance gathered along the current path, le = emission at the
  // primary hit, obj = current vertex: triangle index, src = its position, out = direction back toward the previous vertex.
  // Until round 3's last day the record was five float4 (80 B: two sectors, read and written at every visit); the fifth -
  // aux = BSSRDF profile / refraction attenuation, auxi = refraction: iteration counter - is an array of its own now, which
  // only the BSSRDF and refraction stages touch.
  float4* ctx;
  float4* aux;          // [npix] {aux.xyz, auxi}
  float eye[3];         // origin of every camera ray (skip == JADE_SKIP_CAMERA): not stored per record
  float4* orgs;         // [npix] {origin shared by this record's pending rays, source triangle of those rays (int bits)}
  // Ray slots, two float4 per slot and a record's slots side by side: slot[(p * nslots + k) * 2] = {direction, hit (int bits:
  // -2 inactive, -1 queued / miss, >= 0 triangle)}, [.. + 1] = {hit point, HitResult.distance of the best hit as k_trace
  // compared it}.  A ray is one 16-B read and one 32-B sector written for k_trace, and a record's whole bounce (nslots rays
  // out, nslots results back) is one or two cache lines for k_shade - as planes (round 1) it was a line per component:
  // 28 lines per record and pass once the active list had thinned out.  A queue entry is the slot number p * nslots + k.
  float4* slot;
  // Round 4: ONE float4 per slot - slot[p * nslots + k] = {direction, hit (int bits) / limit} - so that a record's rays are 16 B x
  // nslots side by side (64 B, one sector, for the two-emitter scenes: k_shade read and wrote two), and ONE {hit point, distance} per
  // record, hitp[p]: of a record's rays only the one whose nearest hit is wanted (the indirect ray, or the single ray of a mirror /
  // refraction / camera stage - never two at once) gets a hit point; a yes/no query is answered with the triangle alone
  // (JADE_WANTS_POINT), so k_trace neither solves for nor writes a point nobody reads.  write_all_hits (jade_trace_rays: one slot
  // per "record"): every ray writes point and distance.
  float4* hitp;
  uint32_t write_all_hits;
  // jade_render_params.walk == JADE_WALK_EARLY_EXIT: k_trace ends a shadow / environment-visibility walk at the first recorded
  // hit that settles what k_shade asks of it.  The word a queued ray carries beside its direction (slot[..].w, overwritten by
  // the result) is the LIMIT as a float: the walk may end once its best distance is < limit.  -1 (a NaN: never) = the nearest
  // hit is wanted; JADE_INF_F = any recorded hit (hitArray records a hit only below INF, PathTrace.cu:787); a shadow ray carries
  // the distance at which hitTriangle meets the emitter it aims at (JADE_INF_F if it does not: then no hit can make it
  // visible).  -2 (also a NaN) = no ray in this slot.  With the reference walk k_trace ignores the word.
  // Ray records (round 4): the first rayq_cap entries of the ray queue also exist as 48-B records - what k_trace's refill needs of a
  // ray, computed by the kernel that queued it: {origin, skip word (walk_begin's skipx)} {1 / d, limit} {normalize(d), slot number} -
  // read back coalesced in ONE round trip instead of queue entry -> origin + slot (two, both scattered) + six divisions and a square
  // root per ray.  Entries beyond rayq_cap (and every launch with rayq_cap = 0: k_tail, ordered queues, jade_trace_rays) take the
  // index alone, as before.  Same functions on the same values, so the same bits.
  float4* rayq;
  uint32_t rayq_cap;
  // An ORDERED queue (k_ray_keys + rocPRIM sort; scenes whose tree does not fit the L2) holds queue POSITIONS in the order k_trace is
  // to take them, not slot numbers: position v's ray is record v (v < rayq_cap) or entry v of the index queue, idxq (otherwise).
  const uint32_t* idxq;  // null = the queue k_trace is given IS the index queue, taken in its own order
  // ... and the kernel that queues a ray writes its sort key beside the entry (keyq[position], position < keyq_cap; null = the queue is
  // not ordered): it holds the kind of ray, the source triangle and the direction in registers at that moment (ray_sort_key, jade_hip.hip).
  uint32_t* keyq;
  uint32_t keyq_cap;
  uint32_t key_tri_bits;  // bits of the largest triangle index
  uint32_t env_sampling;  // JADE_ENV_*: 1 = environment rays drawn by importance (non-parity mode)
  uint32_t early_exit;  // 0: the reference's walk; 1: early exits; 2: early exits + the occluder cache (JADE_WALK_EARLY_EXIT_CACHED)
};

enum : uint32_t {
  ST_IDLE = 0,       // between samples
  ST_PRIMARY = 1,    // camera ray in flight
  ST_VERTEX = 2,     // at a surface vertex, bounce not yet sampled
  ST_DIFFUSE = 3,    // diffuse / SSS-diffuse rays in flight
  ST_BSSRDF = 4,
  ST_MIRROR = 5,
  ST_REFRACT_LOOP = 6,
  ST_REFRACT_EXIT = 7,
  ST_INVALID = 255,  // pixel outside the image (edge tiles)
};
#define STF_SSS 1u      /* ST_DIFFUSE: SSS-diffuse variant (albedo, x4) */
#define STF_RR 2u       /* the indirect ray was issued (RR passed) */
#define STF_FULLREFLEX 4u
#define STF_NOENV 8u    /* ST_DIFFUSE / ST_BSSRDF with env_sampling: the drawn direction lay on the wrong side - no environment ray this bounce */

struct RenderConst {
  int32_t width, height;
  int32_t tiles_x;
  uint32_t frame;
  float eye[3];
  float cam[16];
  double two_over_w, two_over_h, aspect;
};

// Work counters are sharded: a block adds into shard blockIdx % JADE_CTR_SHARDS
// (one 64-B line each) so that a million waves do not serialise on one address
// (same-address device atomics retire at ~12 ns each, MI355X_MICROARCH.md
// "fanin"); the host sums the shards after a step.
#define JADE_CTR_SHARDS 256
struct DevCounters {  // three 64-B lines per shard; shade_tail adds by word index, keep the order
  unsigned long long rays_primary, rays_shadow, nodes_visited, tris_tested, shaded_hits, samples;
  unsigned long long pad[2];  // k_trace development profile (JADE_TRACE_PROFILE)
  unsigned long long rays_env, rays_indirect, rays_mirror, rays_refract;  // with rays_shadow: rays_secondary by call site (jade_rt.h)
  unsigned long long rays_inline;  // rays k_light traced itself (primary + mirror)
  unsigned long long nodes_inline, tris_inline;  // ... and their share of nodes_visited / tris_tested
  unsigned long long rays_cached;  // yes/no queries answered by the cached subtrees alone (occluder cache, jade_trace.h)
  unsigned long long rays_tail, nodes_tail, tris_tail;  // rays k_tail traced, and their share of nodes_visited / tris_tested
  unsigned long long pad3[5];
};
#ifndef JADE_TRACE_CHUNK
#define JADE_TRACE_CHUNK 512 /* most rays a wave claims per queue atomic */
#endif
#ifndef JADE_REFILL_MIN
#define JADE_REFILL_MIN 16 /* idle lanes in a wave that trigger k_trace's write-back + refill.  Round 1, all rays: 8: 343, 16: 331, 32: 313, 48: 316 ms of k_trace per 256-spp step; round 2, heavy rays only: 8: 140, 12: 139, 16: 138, 24: 137-139, 32: 139-142, 40: 147 */
#endif

// The per-(pixel, lane) partial sums: 3 planes of JADE_SAMPLE_LANES * npx floats — a full 4K frame on
// one GPU is 3 x 2.1 G entries, past what an int index reaches, so these take 64-bit plane sizes.
// Round 4: the three components of a sum sit side by side (12 B per (lane, pixel), index lane * npx + pixel), not in three planes.
// k_shade adds a finished sample of a scattered record into ONE sector (now and then two) instead of three - read and written: it is
// bound by the sectors it touches, and a tenth of them were these; the first pass, whose records are neighbours, streams either way.
// (`plane` is what the plane layout needed; kept in the signature so that the call sites read as they did.)
static __device__ __forceinline__ jvec3 ld3w(const float* a, size_t plane, size_t i) {
  (void)plane;
  const float* q = a + 3 * i;
  return jv(q[0], q[1], q[2]);
}
static __device__ __forceinline__ void st3w(float* a, size_t plane, size_t i, jvec3 v) {
  (void)plane;
  float* q = a + 3 * i;
  q[0] = v.x;
  q[1] = v.y;
  q[2] = v.z;
}
static __device__ __forceinline__ jvec3 ld3(const float* a, int npix, int p) {
  return jv(a[p], a[npix + p], a[2 * npix + p]);
}
static __device__ __forceinline__ void st3(float* a, int npix, int p, jvec3 v) {
  a[p] = v.x;
  a[npix + p] = v.y;
  a[2 * npix + p] = v.z;
}
static __device__ __forceinline__ jvec3 V3(const float* p) { return jv(p[0], p[1], p[2]); }

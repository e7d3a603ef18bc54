/*
 * jade_rt.h — the C ABI of the jade path-tracing hot path.
 *
 * This is the seam SURVEY.md §8(b) identifies in the reference: everything
 * between "the flat host arrays are ready" (PathTrace.cu:1570-1612) and "the
 * BGR bytes are back on the host" (PathTrace.cu:1739).  The reference has no
 * plugin/FFI API; its implicit interface is the argument list of the
 * `render_pixel` kernel plus six __constant__ symbols
 * (PathTrace.cu:643-648, 1418, 1704-1710, 1731).  The entry points below carry
 * exactly that information: plain pointers and sizes, `int` status codes, no
 * exceptions, no exit() across the boundary (the reference prints CUDA errors
 * and carries on, PathTrace.cu:1476-1482; here every failure is a status plus
 * a thread-local message).
 *
 * Two shared libraries implement this same header:
 *   libjade_hip.so     the product: hand-written HIP for gfx950 (MI355X)
 *   libjade_oracle.so  test infrastructure: a single-purpose CPU restatement
 *                      of PathTrace.cu:669-1474 (oracle/), never shipped and
 *                      never called by the product path
 *
 * The array element types mirror the reference's device structs byte for
 * byte so that a maintainer can pass `&triangles_encoded[0]` and
 * `&nodes_encoded[0]` straight through.
 */
#ifndef JADE_RT_H
#define JADE_RT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bump whenever a struct layout or a documented semantic changes; callers compare
 * jade_abi_version() with the value they were compiled against */
#define JADE_ABI_VERSION 7

/* status codes */
#define JADE_OK 0
#define JADE_ERR_INVALID 1     /* bad argument / inconsistent scene arrays */
#define JADE_ERR_DEVICE 2      /* HIP runtime failure, no device */
#define JADE_ERR_NOMEM 3       /* host or device allocation failed */
#define JADE_ERR_UNSUPPORTED 4 /* e.g. BVH deeper than the traversal stack */

/* material modes (PathTrace.cu:41-45; DIR_REFRACT: PathTrace.cpp:34) */
#define JADE_DIFFUSE 0
#define JADE_MIRROR 1
#define JADE_NO_REFRACT 0
#define JADE_SUB_SURFACE 1
#define JADE_DIR_REFRACT 2

/* Fixed constants of the integrator (PathTrace.cu:32-39); not parameters. */
#define JADE_TILE_SIZE 16
#define JADE_STACK_CAPACITY 128     /* bounce pushes per sample */
#define JADE_BVH_STACK_CAPACITY 128 /* traversal stack entries */
#define JADE_MAX_FULL_REFLEX_TIME 32

/* Sample scheduling (this ABI's definition; the reference races 31 cuRAND
 * states and is not reproducible).  Sample s (0-based, counted
 * across render_step calls) of pixel (x, y) draws from its own Wang-hash
 * stream seeded (x*1973 + y*9277 + (frame + s)*26699) | 1 — the reference's
 * GLSL seed (fshader_render.fsh:82-85) with the frame counter advanced per
 * sample, which is what its progressive preview does (one sample per frame,
 * fshader_preview.fsh:82-85, 402-403).  Samples are therefore independent work
 * items.  A pixel's radiance is summed in JADE_SAMPLE_LANES interleaved
 * partial sums (sample s goes to lane s % JADE_SAMPLE_LANES, in increasing s)
 * which are added in lane order at resolve time: the result does not depend
 * on how samples are batched into steps, on the tile partition, or on how
 * many of a pixel's samples a backend keeps in flight at once.  The lane
 * count is also the most samples of one pixel that can be in flight at once:
 * 1024 lets a GPU that holds an eighth of a 1080p frame (8-GPU tiling) keep as
 * many paths in flight as one that holds the whole frame.
 *
 * Seed reuse: the seed is 32 bits wide and forced odd, i.e. 2^31 distinct streams.  A 1920x1080 frame at
 * 4096 spp draws 8.5e9 of them, so every seed serves about four (pixel, sample) pairs: their random numbers
 * are identical (correlated noise between those samples; they see different camera rays, so their paths
 * differ from the first hit on).  It is the property of the reference's GLSL generator itself (same seed
 * expression, fshader_render.fsh:82-85), kept so that results stay comparable with it; a host that needs
 * more distinct streams varies `frame` between renders. */
#define JADE_SAMPLE_LANES 1024

/* == Triangle_cu, PathTrace.cu:327-338 (112 bytes). */
typedef struct jade_triangle {
  int32_t obj_idx;
  float p1[3], p2[3], p3[3];
  float norm[3];
  float emissive[3];
  float brdf[3];
  int32_t reflex_mode;
  int32_t refract_mode;
  float refract_rate[3];
  float refract_albedo[3];
  float refract_index;
} jade_triangle;

/* == BVHNode_cu, PathTrace.cu:341-345 (40 bytes).  nodes[0] is a dummy, the
 * root is nodes[1], a child index of 0 means "none", n > 0 marks a leaf
 * holding triangles [index, index + n - 1] (PathTrace.cu:525-529, 804, 825,
 * 1557-1565). */
typedef struct jade_bvh_node {
  int32_t left, right;
  int32_t n, index;
  float aa[3], bb[3];
} jade_bvh_node;

/* == Obj_seg, PathTrace.cu:348-351: inclusive range in ORIGINAL (pre-BVH)
 * triangle order. */
typedef struct jade_obj_seg {
  int32_t begin_idx;
  int32_t end_idx;
} jade_obj_seg;

/* Everything `render_pixel` reads (PathTrace.cu:1418 + 639-648).  All
 * pointers are host pointers; jade_scene_create copies, the caller keeps
 * ownership. */
typedef struct jade_scene_desc {
  uint32_t abi_version;           /* JADE_ABI_VERSION */
  int32_t n_triangles;            /* nTriangles_dv */
  const jade_triangle* triangles; /* triangles_cu, BVH (sorted) order */
  int32_t n_nodes;                /* nNodes_dv, including the dummy node 0 */
  const jade_bvh_node* nodes;     /* node_cu */
  int32_t n_emit;                 /* nEmitTriangles_dv */
  const int32_t* emit_indices;    /* emitTrianglesIndices_cu (sorted order) */
  const int32_t* index_mapping;   /* triangle_index_mapping_cu: original -> sorted, [n_triangles] */
  const float* prefix_area;       /* prefix_size_sum_cu: per-object running area, original order */
  int32_t n_objects;
  const jade_obj_seg* obj_segs;   /* obj_segs_cu */
  int32_t env_width, env_height;  /* HDR environment, equirectangular */
  const float* env_rgb;           /* hdrRes.cols: interleaved RGB, row 0 = top (v = 0) */
} jade_scene_desc;

typedef struct jade_render_params {
  int32_t width, height; /* RENDER_WIDTH / RENDER_HEIGHT, run-time here */
  int32_t spp;           /* samples per pixel rendered by this call */
  uint32_t frame;        /* RNG frame counter of sample 0 (seed term), normally 0 */
  float eye[3];          /* eye_dv */
  float camera[16];      /* camera_transform_dv, [col][row] memory order */
  /* image partition: this call renders the 16x16 tiles (tx, ty) with
   * (tx + ty) % tile_nranks == tile_rank — diagonal interleave, so that an
   * object's pixels spread over all ranks (single GPU: 0 of 1) */
  int32_t tile_rank, tile_nranks;
  int32_t device_id;     /* HIP device ordinal (ignored by the oracle) */
  int32_t threads;       /* oracle: worker threads (0 = all cores); HIP: ignored */
  /* HIP: most device memory (bytes) this render may hold for path records and partial sums; 0 = the
   * default (60 % of what is free at jade_render_begin).  The module keeps as many of a pixel's samples
   * in flight as fit - never more than `spp` rounded up to a power of two, so a small render stays
   * small; a progressive host that will add many samples per step passes that number in `spp` here.
   * The image does not depend on either.  Oracle: ignored. */
  uint64_t max_state_bytes;
  /* How a hitBVH query (PathTrace.cu:795-859) is answered - JADE_WALK_*.  The reference walks every node whose box the
   * ray meets and keeps the nearest hit, whatever the caller then does with it.  Two of its three kinds of secondary
   * query only ask a yes/no question of that hit: a shadow ray (:956, 1096, 1292) is used as "is the nearest hit the
   * emitter it aims at" (:957, 1097, 1293), an environment-visibility ray (:980, 1123, 1316) as "is there any hit"
   * (:981, 1124, 1317).  JADE_WALK_EARLY_EXIT ends such a walk at the first recorded hit that settles the question -
   * any hit for the environment ray; for the shadow ray a hit strictly nearer than the emitter's own hitTriangle
   * distance, computed up front with the same statements - so the answer, and with it every sample, pixel and ray
   * count, is the reference's bit for bit (nothing is judged by a tolerance; no box is left out by distance), while
   * nodes_visited / tris_tested count what was actually read and are smaller.  JADE_WALK_REFERENCE (0, what a
   * zero-filled struct asks for) visits what the reference visits: nodes_visited / tris_tested equal the oracle's.
   * Oracle: ignored (it is the reference walk). */
  int32_t walk;
  /* How the direction of an environment-visibility ray is drawn - JADE_ENV_*.  JADE_ENV_REFERENCE (0, what a zero-filled struct asks
   * for): uniformly over the hemisphere, as the reference does (PathTrace.cu:968-979, 1111-1122, 1304-1315) - the mode every parity
   * statement of this header is about.  JADE_ENV_IMPORTANCE (ABI 7; SURVEY 8f rank 3, "optional importance sampling (non-parity
   * mode)"): proportionally to the environment map's luminance x sin(theta) per texel, weighted by 1 / pdf (a direction on the wrong
   * side of the surface contributes nothing and no ray is traced for it).  A different estimator of the SAME integral: the image
   * converges to the same mean with less noise under a sky with a sun, and is NOT the reference's sample for sample - none of the
   * parity claims apply to it, and the oracle refuses it (JADE_ERR_UNSUPPORTED). */
  int32_t env_sampling;
} jade_render_params;
#define JADE_ENV_REFERENCE 0
#define JADE_ENV_IMPORTANCE 1
#define JADE_WALK_REFERENCE 0
#define JADE_WALK_EARLY_EXIT 1
/* JADE_WALK_EARLY_EXIT plus an occluder cache (ABI 7).  The reference tests a leaf's triangles iff the ray meets the leaf's box
 * and every ancestor's, and boxes are nested (jade_scene_create checks it; otherwise this mode walks as JADE_WALK_EARLY_EXIT), so
 * a yes/no query may look ANYWHERE in the tree first: whatever triangle it finds hit below its limit there, the reference's walk
 * finds too.  The module remembers, per (source triangle, kind of query), the subtrees in which the last such queries found their
 * answer and walks those before the root.  The frame, the rays and the samples are the reference's bit for bit, as with
 * JADE_WALK_EARLY_EXIT; nodes_visited / tris_tested count what was read and - the cache being shared by all waves of the device -
 * are not the same from run to run.  jade_stats.rays_cached counts the queries the cached subtrees answered.  Oracle: ignored. */
#define JADE_WALK_EARLY_EXIT_CACHED 2

/* Exact integer work counters; the oracle's and the HIP module's must be
 * equal for the same inputs.  One "ray" is one hitBVH query
 * (PathTrace.cu:795). */
typedef struct jade_stats {
  uint64_t rays_primary;   /* PathTrace.cu:1440 */
  uint64_t rays_secondary; /* every other hitBVH call site */
  uint64_t nodes_visited;  /* V: root + both children of each internal node popped */
  uint64_t tris_tested;    /* T: hitTriangle calls (source triangle excluded) */
  uint64_t shaded_hits;    /* H: pathTracing loop iterations (vertices shaded) */
  uint64_t samples;        /* pixels * spp rendered by this call */
  double kernel_ms;        /* device (or CPU wall) time inside the integrator */
  /* the dominant kernel alone (HIP: k_trace, timed with HIP events on its own
   * stream, summed over launches; oracle: 0) — feeds the roofline figure */
  double trace_ms;
  uint64_t trace_launches;
  /* rays_secondary by call site (exact; oracle == HIP): NEE shadow rays (PathTrace.cu:956, 1096, 1292),
   * environment-visibility rays (:980, 1123, 1316), indirect rays (:1003, 1150, 1339), mirror rays
   * (:1383), refraction rays (:1202, 1241) */
  uint64_t rays_shadow, rays_env, rays_indirect, rays_mirror, rays_refract;
  uint64_t host_syncs;     /* HIP: times the host waited for the device inside step/flush; oracle: 0 */
  /* HIP: rays (of the counts above) that were traced inside the fused first-pass kernel k_light, not by k_trace:
   * trace_ms / trace_launches cover the other rays only.  Oracle: 0. */
  uint64_t rays_inline;
  double light_ms;         /* HIP: device time of k_light (HIP events), summed over launches; oracle: 0 */
  /* HIP: the node records / triangle tests (of nodes_visited / tris_tested) that belong to the rays_inline rays, so that
   * SURVEY 8(d)'s algorithmic bytes 40 V + 36 T can be stated per kernel: k_trace's are the difference.  Oracle: 0. */
  uint64_t nodes_inline, tris_inline;
  /* HIP, JADE_WALK_EARLY_EXIT_CACHED: shadow / environment-visibility queries (of rays_shadow + rays_env) that the occluder cache
   * answered without a walk from the root.  Oracle and the other walks: 0. */
  uint64_t rays_cached;
  /* HIP (ABI 7): the last paths of a render - an active list of at most a few ten thousand records - are finished by ONE kernel,
   * k_tail, in which every wave shades and traces its own records; its rays (of the counts above), node records, triangle tests,
   * device time and launches.  trace_ms / trace_launches do not include them.  Oracle: 0. */
  uint64_t rays_tail, nodes_tail, tris_tail;
  double tail_ms;
  uint64_t tail_launches;
} jade_stats;

typedef struct jade_scene jade_scene; /* opaque */

int jade_abi_version(void);
const char* jade_backend_name(void);  /* "hip-gfx950" | "oracle-cpu" */
const char* jade_last_error(void);    /* thread-local text of the last failure */
int jade_device_count(int* n);

/* Validates the arrays (index ranges, BVH depth <= JADE_BVH_STACK_CAPACITY - 1)
 * and builds the backend's private copy.  Replaces PathTrace.cu:1618-1698. */
int jade_scene_create(const jade_scene_desc* desc, int device_id, jade_scene** out);
void jade_scene_destroy(jade_scene* scene);

/* Renders the caller's tiles.  Replaces PathTrace.cu:1704-1739 (constant
 * upload, RNG init, kernel launch, sync, D2H copy).
 *   out_rgb  nullable, width*height*3 floats: linear mean radiance before tone
 *            mapping (PathTrace.cu:1457), RGB, pixel (x, y) at 3*(y*width+x),
 *            y grows upward as in the reference (PathTrace.cu:1470).
 *   out_bgr8 nullable, width*height*3 bytes: ACES + gamma + BGR pack exactly
 *            as PathTrace.cu:1461-1473 (what save_image() writes).
 * Pixels of tiles this rank does not own are left untouched.
 * Synchronous; one render in flight per scene. */
int jade_render(jade_scene* scene, const jade_render_params* params, float* out_rgb,
                uint8_t* out_bgr8, jade_stats* stats);

/* Progressive form: the sample counter and the radiance sums persist on the
 * backend between calls, so N calls of spp samples equal one call of N*spp
 * (the reference's own running-mean preview, fshader_preview.fsh:402-403, is
 * the model).  begin() resets the accumulation; step() adds `spp` samples to
 * every owned pixel and leaves the results on the backend; resolve() writes
 * the mean so far.  jade_render == begin + step + resolve. */
int jade_render_begin(jade_scene* scene, const jade_render_params* params);
int jade_render_step(jade_scene* scene, int32_t spp, jade_stats* stats_accum);
int jade_render_resolve(jade_scene* scene, float* out_rgb, uint8_t* out_bgr8);
/* A backend may return from step() while the last, longest paths of that step
 * are still unfinished and carry them into the next step (samples are
 * independent work items, so the result is the same; what it saves is the
 * nearly empty passes at the end of every step).  flush() finishes them: after
 * it every sample requested so far is in the sums and in stats_accum.  Every
 * resolve variant flushes first; a caller only needs flush() to close a timed
 * region or to read final statistics without resolving. */
int jade_render_flush(jade_scene* scene, jade_stats* stats_accum);

/* Resolve with a choice of tone operator for out_bgr8 (out_rgb is always the
 * linear mean):
 *   JADE_TONEMAP_ACES      ACESToneMapping, PathTrace.cu:680-682 (what the CUDA
 *                          program writes; jade_render_resolve uses this)
 *   JADE_TONEMAP_REINHARD  toneMapping(c, limit) = c / (1 + luminance/limit),
 *                          PathTrace.cu:669-672 == shaders/pass3.fsh:8-18, the
 *                          GL preview's post pass (limit 1.5 there)
 * followed in both cases by gamma 1/2.2, x255, clamp, BGR (PathTrace.cu:1464-1473). */
#define JADE_TONEMAP_ACES 0
#define JADE_TONEMAP_REINHARD 1
int jade_render_resolve_ex(jade_scene* scene, int tonemap, float limit, float* out_rgb, uint8_t* out_bgr8);

/* Device-resident resolve for multi-GPU gathers (HIP backend only; the oracle
 * returns JADE_ERR_UNSUPPORTED).  Writes this rank's tiles compactly into
 * device memory: tile t (t-th owned tile in increasing row-major id) occupies
 * floats [t*768, (t+1)*768) as 16x16 RGB rows.  `dev_tiles` must hold
 * jade_owned_tile_count()*768 floats.  `stream` is a hipStream_t (0 = null
 * stream).  Out-of-image pixels of edge tiles are written as 0. */
int jade_render_resolve_tiles_device(jade_scene* scene, float* dev_tiles, void* stream);
int jade_owned_tile_count(int32_t width, int32_t height, int32_t tile_rank, int32_t tile_nranks);

/* One frame on several GPUs from ONE process.  scenes[i] must have been created
 * on the device that renders share i (jade_scene_create(desc, device_i, ..)); tiles are dealt
 * (tx + ty) % ndev exactly as with tile_rank / tile_nranks, every device renders its share
 * concurrently (one host thread per device), then the device scenes[0] lives on collects the
 * compact tile buffers with ONE RCCL gather over xGMI (ncclCommInitAll over the devices + grouped
 * ncclSend / ncclRecv; librccl is loaded on first use) and the frame is assembled and tone-mapped
 * once.  If the same device appears more than once (a one-GPU rehearsal of the partition) the
 * shares are copied device-to-device instead: RCCL cannot put two ranks on one device.
 * The result is bit-identical to jade_render on one device.  params->tile_rank / tile_nranks /
 * device_id are ignored.  (The one-process-per-GPU form with torch.distributed's RCCL gather is
 * what bench.py and jaderaytracerendering_amd/distributed.py use.) */
int jade_render_multi(jade_scene* const* scenes, int ndev, const jade_render_params* params,
                      float* out_rgb, uint8_t* out_bgr8, jade_stats* stats);

/* What a backend holds for the render begun last (after jade_render_begin; values are the backend's own choices and
 * never change a result - jade_render_params.max_state_bytes / .spp are what a caller steers them with):
 *   JADE_Q_RECORDS_PER_PIXEL  HIP: samples of one pixel in flight at once (a power of two <= JADE_SAMPLE_LANES); oracle: 1
 *   JADE_Q_STATE_BYTES        HIP: device bytes held for path records, ray queue, lists and partial sums; oracle: host
 *                             bytes of its partial sums
 *   JADE_Q_SUM_LANES          partial sums kept per pixel: min(JADE_SAMPLE_LANES, the announced spp rounded up to a power
 *                             of two); grows by itself if steps add more samples than were announced
 * (The reference has no counterpart: its one kernel keeps a pixel's whole state in registers, PathTrace.cu:1418-1474.) */
#define JADE_Q_RECORDS_PER_PIXEL 0
#define JADE_Q_STATE_BYTES 1
#define JADE_Q_SUM_LANES 2
int jade_render_query(jade_scene* scene, int what, int64_t* value);

/* Single-query entry point used by the parity tests: traces `n` rays through
 * the scene's BVH with hitBVH semantics (PathTrace.cu:795-859).
 *   origins/dirs: n*3 floats; skip: n source-triangle indices (-1 = none)
 *   hit_index: n (-1 on miss); hit_dist: n; hit_point: n*3
 *   stats: nodes_visited / tris_tested / rays_secondary are accumulated */
int jade_trace_rays(jade_scene* scene, int32_t n, const float* origins, const float* dirs,
                    const int32_t* skip, int32_t* hit_index, float* hit_dist, float* hit_point,
                    jade_stats* stats);

#ifdef __cplusplus
}
#endif
#endif /* JADE_RT_H */

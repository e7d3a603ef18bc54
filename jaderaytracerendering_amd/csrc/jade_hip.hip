// jade_hip.hip — libjade_hip.so: the jade_rt.h C ABI on MI355X (gfx950).
//
// Replaces PathTrace.cu:1618-1741 (upload, constants, RNG init, the single
// render_pixel launch, sync, download).  Kernel structure (one HIP stream):
//
//   k_init          resets the path records (the partial sums are cleared with the allocation)
//   k_light_packet  a step's FIRST pass, fused: camera rays, the sky, emitters and pure mirrors traced and shaded in one
//                   kernel, a wave's 64 rays walked together as a packet (jade_trace.h); records that meet jade / diffuse /
//                   glass, or whose packet fans out, are handed to k_shade.  k_light: the same pass, one lane per ray
//                   (trees deeper than 63, frames the statue fills); k_heavy_scan / k_heavy_pack close the hand-over list
//   k_shade         every branch of pathTracing, over the hand-over list, then over the active list: folds last pass's hit
//                   results into the path, starts the next sample when a path ends, emits the next bounce's rays into a
//                   compacted queue (wave prefix sums + one 64-bit atomic per block)
//   k_ray_keys      (+ rocPRIM pairs sort) orders the queue for scenes whose tree does not fit the L2
//   k_trace         persistent workgroups claim chunks of the queue and run the BVH traversal (jade_trace.h) - the
//                   dominant kernel.  With jade_render_params.walk = JADE_WALK_EARLY_EXIT a shadow / environment-visibility
//                   walk ends at the first recorded hit that settles what the integrator asks of it (the limit k_shade puts
//                   beside the ray); k_trace_wide: the same kernel visiting four grandchildren per node, launched instead for
//                   such renders on trees that do not fit the L2
//   k_resolve       adds the partial sums; mean, ACES, gamma, BGR8 pack (PathTrace.cu:1457-1473)
//   k_arm / k_shade_lean  list the records with work / shade light samples over all records: the schedules without the
//                   fused pass (flushes, JADE_FUSED=0)
//
// shade/trace alternate until a shade pass emits nothing (every path record has
// finished its samples), or until so few are active that the step hands them to the
// next one (jade_render_flush).  There is no CPU fallback: if HIP is unavailable the
// entry points return JADE_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: the library is loaded on first use (jade_render_multi on distinct devices)

#include <algorithm>
#include <cmath>
#include <map>
#include <thread>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include "jade_device.h"
#include "jade_shade.h"
#include "jade_trace.h"

// ------------------------------------------------------------------ kernels --

#ifndef JADE_TRACE_NT
#define JADE_TRACE_NT 1 /* k_trace reads and writes the ray records with non-temporal hints */
#endif
#if JADE_TRACE_NT
#define NT_LD(p) __builtin_nontemporal_load(p)
#define NT_ST(p, v) __builtin_nontemporal_store((v), (p))
#else
#define NT_LD(p) (*(p))
#define NT_ST(p, v) (*(p) = (v))
#endif
#if JADE_TRACE_PROFILE
__device__ unsigned long long g_packet_prof[PKL_N];  // ... of k_light_packet
__device__ unsigned long long g_trace_prof[PL_N];  // development profile of k_trace: shader clocks per piece of the loop, summed over waves
#endif
typedef float jade_v4f __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ float4 nt_ld4(const float4* p) {
  const jade_v4f v = NT_LD(reinterpret_cast<const jade_v4f*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
static __device__ __forceinline__ void nt_st4(float4* p, float x, float y, float z, float w) {
  const jade_v4f v = {x, y, z, w};
  NT_ST(reinterpret_cast<jade_v4f*>(p), v);
}


struct alignas(8) QueueCtl {
  uint32_t count;   // rays emitted by the last shade pass          } one 64-bit word: k_shade reserves its queue
  uint32_t active;  // records with rays in flight after that pass } and list space with ONE atomic per block
  uint32_t next;    // next unclaimed queue entry (trace)
  uint32_t heavy;   // records k_shade_lean handed to k_shade this pass
  uint32_t fp_bad;  // jade_fp_selftest result (checked once)
  uint32_t pad[3];
};

// how many lanes of mask m sit below this one (v_mbcnt_lo + v_mbcnt_hi: two instructions, no per-lane mask kept in registers)
static __device__ __forceinline__ uint32_t lanes_below(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
static __device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t* total) {
  const int lane = threadIdx.x & 63;
  uint32_t x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  *total = __shfl(x, 63, 64);
  return x - v;
}

// The sort key of a queued ray (ordered queues, k_ray_keys below): the kind of ray - shadow ray towards emitter k / environment ray /
// indirect ray / single-ray stage - then the triangle it leaves (its index in BVH order is a place on the tree's own space-filling
// curve), then the octant it heads into; left-aligned in 32 bits.
static __device__ __forceinline__ uint32_t ray_sort_key(uint32_t st, uint32_t k, int n_emit, int32_t skip, float dx, float dy, float dz, uint32_t tri_bits) {
  uint32_t cls = 7u;  // single-ray stages (mirror, refraction, camera)
  if (st == ST_DIFFUSE || st == ST_BSSRDF) cls = (int)k < n_emit ? (k < 5u ? k : 4u) : ((int)k == n_emit ? 5u : 6u);
  const uint32_t tri = skip < 0 ? 0u : ((uint32_t)skip & ((1u << tri_bits) - 1u));
  const uint32_t oct = (dx < 0.0f ? 1u : 0u) | (dy < 0.0f ? 2u : 0u) | (dz < 0.0f ? 4u : 0u);
  return (cls << 29) | (((tri << 3) | oct) << (26u - tri_bits));
}
static __device__ __forceinline__ unsigned long long wave_sum_u32(uint32_t v) {
  unsigned long long x = v;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;  // valid in lane 0
}

// Record p has finished `done` samples: which sample comes next and for which owned pixel.
struct NextSample {
  uint32_t sidx;  // global sample index (jade_rt.h)
  int pixel;      // owned-pixel index it belongs to
};
// (m, home) = (p / npx, p % npx): where record p sits; computed once per thread, an integer division each.
// rpp and JADE_SAMPLE_LANES are powers of two, so everything else is shifts — and the 64-bit modulo of the
// pixel rotation only runs when rotation is on (it is not, by default).
static __device__ __forceinline__ NextSample next_sample_mh(const PathState& P, uint32_t m, uint32_t home, uint32_t done) {
  const uint32_t rpp_log2 = 31u - (uint32_t)__clz(P.rpp);
  const uint32_t per_log2 = (31u - (uint32_t)__clz(JADE_SAMPLE_LANES)) - rpp_log2;  // samples per record per block = LANES / rpp
  const uint32_t blk = done >> per_log2, n = done - (blk << per_log2);
  NextSample r;
  r.sidx = JADE_SAMPLE_LANES * blk + m + (n << rpp_log2);
  r.pixel = P.stride == 0 ? (int)home
                          : (int)(((unsigned long long)home + (unsigned long long)n * (uint32_t)P.stride) % (uint32_t)P.npx);
  return r;
}
static __device__ __forceinline__ NextSample next_sample(const PathState& P, int p, uint32_t done) {
  const uint32_t m = (uint32_t)p / (uint32_t)P.npx;
  return next_sample_mh(P, m, (uint32_t)p - m * (uint32_t)P.npx, done);
}
static __device__ __forceinline__ bool pixel_xy_t(const RenderConst& R, int tid, int pixel, int* x, int* y) {
  int l = pixel & 255;
  *x = (tid % R.tiles_x) * JADE_TILE_SIZE + (l & 15);
  *y = (tid / R.tiles_x) * JADE_TILE_SIZE + (l >> 4);
  return *x < R.width && *y < R.height;
}
static __device__ __forceinline__ bool pixel_xy(const RenderConst& R, const int32_t* tile_ids, int pixel, int* x, int* y) {
  return pixel_xy_t(R, tile_ids[pixel >> 8], pixel, x, y);
}

__global__ void k_selftest(QueueCtl* q, float one) { q->fp_bad = (uint32_t)jade_fp_selftest(one); }

__global__ void k_init(PathState P) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.npix) return;
  P.hdr[p] = make_uint4(0u, 0u, ST_IDLE, 0u);
}

#define JADE_ARM_BLOCK 1024
#define JADE_ARM_PER_THREAD 8
// Lists every record that has work in this step (samples left to start, or a
// path suspended by a previous step): the input of the first shade pass.
__global__ __launch_bounds__(JADE_ARM_BLOCK) void k_arm(PathState P, uint32_t target_spp, uint32_t* active_out, QueueCtl* qc) {
  // a block lists JADE_ARM_PER_THREAD * 1024 consecutive records with ONE list atomic (same-address atomics
  // cost ~12 ns each: at 534 M records one per 1024 was 6 ms), in record order
  constexpr int NW = JADE_ARM_BLOCK / 64, NJ = JADE_ARM_PER_THREAD;
  __shared__ uint32_t sh_cnt[NJ * NW + 1];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t base = (size_t)blockIdx.x * (JADE_ARM_BLOCK * NJ);
  uint32_t wantm = 0, off[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const size_t p = base + (size_t)j * JADE_ARM_BLOCK + threadIdx.x;
    bool want = false;
    if (p < (size_t)P.npix) {
      const uint4 h = P.hdr[p];
      const uint32_t st = h.z & 255u;
      want = st != ST_IDLE || next_sample(P, (int)p, h.y).sidx < target_spp;
    }
    const unsigned long long m = __ballot(want);
    off[j] = lanes_below(m);
    if (want) wantm |= 1u << j;
    if (lane == 0) sh_cnt[j * NW + w] = (uint32_t)__popcll(m);
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // exclusive prefix over (j, wave), then the block's place in the list
    uint32_t tot = 0;
    for (int i = 0; i < NJ * NW; ++i) {
      const uint32_t c = sh_cnt[i];
      sh_cnt[i] = tot;
      tot += c;
    }
    sh_cnt[NJ * NW] = tot ? atomicAdd(&qc->active, tot) : 0u;
  }
  __syncthreads();
  const uint32_t blk = sh_cnt[NJ * NW];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    if (wantm & (1u << j)) active_out[blk + sh_cnt[j * NW + w] + off[j]] = (uint32_t)(base + (size_t)j * JADE_ARM_BLOCK + threadIdx.x);
}

// A step hands its unfinished paths to the next step (or to flush) once fewer than JADE_CARRY_FRACTION of the records it
// started with are still active.  The paths left are the long ones (jade: ~10 bounces against 1-2 for the sky and the
// mirror floor): finishing them inside every step means dozens of thin passes per step, whose sparse record accesses
// waste most of every cache line; carried over, they ride along with the next step's full passes and the thin tail
// is paid once per render (round 1, C3: 364 -> 286 k_trace launches per 4096 spp, +3 % Mray/s at 0.02 against none).  What
// is carried is work moved, not saved: the flush at the end of a render finishes it, so the fraction sets how long that
// flush is - round 3, 4 x 1024 spp of C3 with the flush inside the clock: 0.001 / 0.002 / 0.003 / 0.005 / 0.02 / 0.05 = 708.6 /
// 707.9 / 705.5 / 705.4 / 709.5 / 698 + 167 ms per step, with a final flush of 32 / 40 / 49 / 67 / 210 / 669 ms.  0.003: as fast as
// any, and a render's last call returns in 49 ms.  JADE_CARRY_FRACTION in the environment overrides it (0 = only the
// absolute floor below).
#ifndef JADE_CARRY_FRACTION
#define JADE_CARRY_FRACTION 0.003
#endif
#ifndef JADE_CARRY_RECORDS
#define JADE_CARRY_RECORDS 32768u /* ... and in any case once fewer than this (and < 0.1 % of its records) are active */
#endif
#ifndef JADE_SHADE_BLOCK
#define JADE_SHADE_BLOCK 512 /* threads per k_shade block: one queue + one list atomic per block (512: +1.8 % over 256; 1024: none) */
#endif
#define JADE_SHADE_NW (JADE_SHADE_BLOCK / 64)
#ifndef JADE_LEAN_BLOCK
#define JADE_LEAN_BLOCK 512 /* threads per k_shade_lean block (1024 = half the list atomics, but 26 vs 20 ms per pass at 534 M records) */
#endif
#ifndef JADE_SHADE_WAVES
#define JADE_SHADE_WAVES 6 /* k_shade: 75 VGPRs, no spill (7 would spill 16 B).  +4 % over 5 when it runs alone, +-0 behind k_shade_lean */
#endif
// One record's shade pass (see the file header).  LEAN = the kernel that only knows camera rays,
// the sky and pure mirrors: it hands every other record to the full kernel through `defer`
// (either untouched, or parked at ST_VERTEX with its path state stored).  Same statements either
// way: the lean paths are the shared helpers consume_mirror / begin_bounce_lean / bounce_mirror.
template <bool LEAN, bool ENVIS = false>  // ENVIS: environment rays by importance (non-parity; k_shade_envis only)
static __device__ __forceinline__ void shade_record(const DevScene& S, const PathState& P, const RenderConst& R, const int32_t* tile_ids,
                                                    uint32_t target_spp, const int p, ShadeCtx& c, uint32_t& st_out, bool& defer) {
  const int npix = P.npix;
  uint32_t st = ST_INVALID;
  defer = false;
  // Prologue: everything most record-passes need, requested together so the pass is one
  // round trip deep instead of one per field (the kernel is latency-bound: PMC shows its
  // waves parked on s_waitcnt 74 % of the time).
  const int pp = p < npix ? p : 0;
  const uint4 hdr0 = P.hdr[pp];
  const uint32_t word = hdr0.z;
  const uint32_t rng0 = hdr0.x;
  const uint32_t done0 = hdr0.y;
  const float4 slot0 = P.slot[(size_t)pp * P.nslots];  // slot 0: {direction, hit}
  const int hit0 = __float_as_int(slot0.w);
  const jvec3 dir0 = jv(slot0.x, slot0.y, slot0.z);
  const uint32_t rec_m = (uint32_t)pp / (uint32_t)P.npx;
  const int home_pix = (int)((uint32_t)pp - rec_m * (uint32_t)P.npx);
  const int tid0 = tile_ids[home_pix >> 8];
  if (p < npix) st = word & 255u;
  if (st != ST_INVALID) {
    const Px px(P, p);
    // State is loaded and stored by need: a background record (camera ray ->
    // sky, or -> mirror floor -> sky) touches ~90 B per pass, not the whole
    // ~230-B record; only records on a multi-bounce path carry thr/acc/le/....
    const bool on_path = st >= ST_DIFFUSE && st <= ST_REFRACT_EXIT;
    if (LEAN && on_path && st != ST_MIRROR) {  // needs the full kernel: hand it over untouched
      defer = true;
      st_out = st;
      return;
    }
    const bool has_ctx = on_path || st == ST_VERTEX;  // ST_VERTEX: handed over by the lean kernel this pass
    c.rng = rng0;
    c.depth = (word >> 8) & 255u;
    c.flags = word >> 16;
    c.stage = st;
    c.thr = jv(1, 1, 1);
    c.acc = jv(0, 0, 0);
    c.le = jv(0, 0, 0);
    c.obj = 0;
    c.src = jv(0, 0, 0);
    c.out = jv(0, 0, 0);
    if (has_ctx) {
      // Stored and loaded by need (the lean kernel is bound by these bytes): before the first
      // path_push thr is 1 and acc 0 (only path_push changes them, and it counts depth up); a
      // mirror ray in flight needs neither the vertex it left nor, at depth 0, a stored Le
      // (it is the emissive of the triangle still in obj).
      const float4* cx = P.ctx + (size_t)p * 4;
      const float4 b0 = cx[0], b1 = cx[1], b2 = cx[2], b3 = cx[3];
      c.obj = __float_as_int(b0.w);
      if (c.depth != 0) {
        c.thr = jv(b0.x, b0.y, b0.z);
        c.acc = jv(b1.x, b1.y, b1.z);
      }
      if (st == ST_MIRROR && c.depth == 0) c.le = V3(shade_tri(S, c.obj).m->emissive);
      else c.le = jv(b2.x, b2.y, b2.z);
      if (st != ST_MIRROR) {
        c.src = jv(b2.w, b3.x, b3.y);
        c.out = jv(b3.z, b3.w, b1.w);
      }
    }
    uint32_t done = done0;
    jvec3 l_final;
    bool finished = false;  // a sample ended: colour in `color`
    jvec3 color = jv(0, 0, 0);

    // (a) fold in the results of the rays issued by the previous pass
    if (st == ST_PRIMARY) {
      int h = hit0;
      jvec3 d = dir0;
      if (h < 0) {
        color = sample_hdr(S, d);  // PathTrace.cu:1443-1445
        finished = true;
      } else {
        c.le = V3(shade_tri(S, h).m->emissive);
        c.thr = jv(1, 1, 1);
        c.acc = jv(0, 0, 0);
        c.depth = 0;
        c.obj = h;
        c.src = px.hpt(0);
        c.out = jv_neg(d);
        st = ST_VERTEX;
      }
    } else if (on_path) {
      int r = LEAN ? consume_mirror(S, px, c, &l_final) : consume<ENVIS>(S, px, c, &l_final);
      if (r == CONSUME_VERTEX) {
        st = ST_VERTEX;
      } else if (r == CONSUME_EMITTED) {
        st = c.stage;
      } else {
        if (r == CONSUME_END) color = jv_add(c.le, jv_add(c.acc, jv_mul(c.thr, l_final)));
        else color = c.le;  // pathTracing returned 0
        finished = true;
      }
    }
    // (b) advance until this record has rays in flight or no samples left
    for (;;) {
      if (finished) {
        // final_result = final_result + color (PathTrace.cu:1454), into the partial sum of
        // this sample's (pixel, lane): only this record touches it in this block
        {
          const NextSample cs = next_sample_mh(P, rec_m, (uint32_t)home_pix, done);  // the sample that just ended
          const size_t si = (size_t)(cs.sidx % JADE_SAMPLE_LANES) * (size_t)P.npx + (size_t)cs.pixel;
          const size_t sn = (size_t)P.sum_lanes * (size_t)P.npx;
          st3w(P.sum, sn, si, jv_add(ld3w(P.sum, sn, si), color));
        }
        done += 1;
        c.c_samples += 1;
        finished = false;
        st = ST_IDLE;
      }
      if (st == ST_VERTEX) {
        if (LEAN && !lean_can_shade(shade_tri(S, c.obj).m)) {  // jade / diffuse / glass: the full kernel continues from here
          defer = true;
          break;
        }
        if (LEAN ? begin_bounce_lean(S, px, c, &l_final) : begin_bounce<ENVIS>(S, px, c, &l_final)) {
          st = c.stage;
          if (!ENVIS || c.n_emit_rays != 0) break;
          // env_sampling only: a bounce that emitted NO ray (the drawn sky direction lay on the wrong side, no shadow ray faced its
          // emitter, the roulette ended the path) has all its results already - nothing is visible: folded in right here (with the
          // reference's sampling there is always the environment ray)
          const int r = consume<ENVIS>(S, px, c, &l_final);
          if (r == CONSUME_VERTEX) {
            st = ST_VERTEX;
            continue;
          }
          color = r == CONSUME_END ? jv_add(c.le, jv_add(c.acc, jv_mul(c.thr, l_final))) : c.le;
          finished = true;
          continue;
        }
        color = jv_add(c.le, jv_add(c.acc, jv_mul(c.thr, l_final)));
        finished = true;
        continue;
      }
      if (st == ST_IDLE) {
        // next sample of this record; samples of out-of-image pixels (edge tiles) are skipped
        int x, y;
        NextSample ns = next_sample_mh(P, rec_m, (uint32_t)home_pix, done);
        while (ns.sidx < target_spp && !pixel_xy_t(R, ns.pixel == home_pix ? tid0 : tile_ids[ns.pixel >> 8], ns.pixel, &x, &y)) {
          done += 1;
          ns = next_sample_mh(P, rec_m, (uint32_t)home_pix, done);
        }
        if (ns.sidx >= target_spp) break;
        // camera ray, PathTrace.cu:1428-1437
        c.rng = jade_rng_seed((uint32_t)x, (uint32_t)y, R.frame + ns.sidx);
        float fx = (float)x + jade_rand(&c.rng);
        double lo = -1.0 + R.two_over_w * ((double)fx - 0.5);
        float left_offset = (float)(lo * R.aspect);
        float fy = (float)y + jade_rand(&c.rng);
        float up_offset = (float)(-1.0 + R.two_over_h * ((double)fy - 0.5));
        jvec3 dir = jade_transform(jv(left_offset, up_offset, -1.5f), 0.0f, R.cam);
        dir = jv_normalize(dir);
        reinterpret_cast<int*>(P.orgs + p)[3] = JADE_SKIP_CAMERA;  // no source triangle, and the origin is the eye: k_trace takes it from P.eye
        px.set_dir(0, dir);
        px.set_hit(0, -1);
        c.n_emit_rays = 1;
        c.c_primary += 1;
        st = ST_PRIMARY;
        break;
      }
      break;  // a pending stage that just emitted (refraction loop)
    }
    if (st != ST_PRIMARY && c.n_emit_rays) {  // the secondary rays of this pass by hitBVH call site (jade_rt.h)
      if (st == ST_DIFFUSE || st == ST_BSSRDF) {
        const uint32_t ind = (c.flags & STF_RR) ? 1u : 0u;
        const uint32_t env = (c.flags & STF_NOENV) ? 0u : 1u;  // one environment ray always (env_sampling: unless its direction lay on the wrong side)
        c.c_cls += env | (ind << 8);  // ... and the indirect ray if the roulette passed
        c.c_shadow += (uint32_t)c.n_emit_rays - env - ind;
      } else {
        c.c_cls += st == ST_MIRROR ? (1u << 16) : (1u << 24);
      }
    }
    P.hdr[p] = make_uint4(c.rng, done, st | (c.depth << 8) | (c.flags << 16), 0u);
    if ((st >= ST_DIFFUSE && st <= ST_REFRACT_EXIT) || st == ST_VERTEX) {
      // (aux and auxi - P.aux - are begin_bounce's and consume's: written through Px while this record was shaded)
      float4* cx = P.ctx + (size_t)p * 4;
      cx[0] = make_float4(c.thr.x, c.thr.y, c.thr.z, __int_as_float(c.obj));
      cx[1] = make_float4(c.acc.x, c.acc.y, c.acc.z, c.out.z);
      cx[2] = make_float4(c.le.x, c.le.y, c.le.z, c.src.x);
      cx[3] = make_float4(c.src.y, c.src.z, c.out.x, c.out.y);
    }
  }
  st_out = st;
}

// Queue the emitted rays, list the record for the next pass and/or for the full kernel: wave scans,
// then ONE atomic per block for queue + list (not per wave: see DevCounters).
template <bool LEAN, int NW>  // NW: waves per block
static __device__ __forceinline__ void shade_tail(const PathState& P, int p, uint32_t st, const ShadeCtx& c, bool defer,
                                                  uint32_t* active_out, uint32_t* heavy_out, uint32_t* queue, QueueCtl* qc,
                                                  DevCounters* ctr) {
  __shared__ uint32_t sh_rays[NW], sh_act[NW], sh_def[NW], sh_base[3];
  __shared__ uint32_t sh_ctr[NW][8];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t total;
  const uint32_t off = wave_excl_scan((uint32_t)c.n_emit_rays, &total);
  const bool live = c.n_emit_rays > 0;
  const unsigned long long am = __ballot(live), dm = LEAN ? __ballot(defer) : 0ull;
  const uint32_t aoff = lanes_below(am), doff = lanes_below(dm);
  if (lane == 0) {
    sh_rays[w] = total;
    sh_act[w] = (uint32_t)__popcll(am);
    sh_def[w] = (uint32_t)__popcll(dm);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tr = 0, ta = 0, td = 0;
    for (int i = 0; i < NW; ++i) {
      tr += sh_rays[i];
      ta += sh_act[i];
      td += sh_def[i];
    }
    // count can never carry into active: a pass emits < 2^32 rays (nslots * npix < 2^32 is checked in jade_render_begin)
    const unsigned long long got =
        (tr | ta) ? atomicAdd(reinterpret_cast<unsigned long long*>(&qc->count), (unsigned long long)tr | ((unsigned long long)ta << 32)) : 0ull;
    sh_base[0] = (uint32_t)got;
    sh_base[1] = (uint32_t)(got >> 32);
    sh_base[2] = (LEAN && td) ? atomicAdd(&qc->heavy, td) : 0u;
  }
  __syncthreads();
  if (live || (LEAN && defer)) {
    uint32_t wq = sh_base[0] + off, wa = sh_base[1] + aoff, wd = sh_base[2] + doff;
    for (int i = 0; i < w; ++i) {
      wq += sh_rays[i];
      wa += sh_act[i];
      wd += sh_def[i];
    }
    if (LEAN && defer) heavy_out[wd] = (uint32_t)p;
    if (live && active_out) active_out[wa] = (uint32_t)p;
    (void)wq;
  }
  {
    // The wave's rays go into its stretch of the queue SLOT BY SLOT: all its records' slot-0 rays (shadow rays towards
    // the first emitter), then the slot-1 rays, ...  The 64 entries a wave of k_trace takes are then one kind of ray from
    // 64 neighbouring records - same target or same sky, origins side by side: they walk the same part of the tree
    // (L1 hits) and end together - instead of the four rays of 16 records.
    uint32_t wbase = sh_base[0];
    for (int i = 0; i < w; ++i) wbase += sh_rays[i];
    const int used = !live ? 0 : (st == ST_DIFFUSE || st == ST_BSSRDF) ? P.nslots : 1;
    // ... and, for the first rayq_cap entries, the ray as k_trace's refill needs it (PathState.rayq): origin and source triangle are
    // the record's (camera rays: the eye), direction and limit the slot's, just written by this thread
    jvec3 ro = jv(0, 0, 0);
    int32_t rskip = -1;
    if (used && (P.rayq_cap || P.keyq)) {
      const float4 og = P.orgs[p];
      rskip = __float_as_int(og.w);
      ro = rskip == JADE_SKIP_CAMERA ? jv(P.eye[0], P.eye[1], P.eye[2]) : jv(og.x, og.y, og.z);
    }
    for (int k = 0; k < P.nslots; ++k) {
      const float4* sl = P.slot + ((size_t)p * P.nslots + k);
      const bool q = k < used && reinterpret_cast<const int*>(sl)[3] != -2;  // (-2: no ray in this slot; anything else: the queued ray's limit, jade_device.h)
      const unsigned long long m = __ballot(q);
      const uint32_t pos = wbase + lanes_below(m);
      const uint32_t e = (uint32_t)p * (uint32_t)P.nslots + (uint32_t)k;
      if (q) queue[pos] = e;
      if (q && P.keyq && pos < P.keyq_cap) {  // ordered queue: the key beside the entry (k_ray_keys read three scattered sectors per ray for it)
        const float4 dv = sl[0];
        P.keyq[pos] = ray_sort_key(st, (uint32_t)k, P.nslots - 2, rskip, dv.x, dv.y, dv.z, P.key_tri_bits);
      }
      if (q && pos < P.rayq_cap) {
        const float4 dv = sl[0];
        jvec3 inv, dn;
        uint32_t skipx;
        walk_prepare(ro, jv(dv.x, dv.y, dv.z), rskip, dv.w, &inv, &dn, &skipx);
        float4* rq = P.rayq + (size_t)pos * 3;
        nt_st4(rq, ro.x, ro.y, ro.z, __uint_as_float(skipx));
        nt_st4(rq + 1, inv.x, inv.y, inv.z, P.early_exit ? dv.w : __int_as_float(-1));
        nt_st4(rq + 2, dn.x, dn.y, dn.z, __uint_as_float(e));
      }
      wbase += (uint32_t)__popcll(m);
    }
  }
  // work counters: per wave into LDS, then one set of atomics per block
  const uint32_t s0 = (uint32_t)wave_sum_u32(c.c_primary), s1 = (uint32_t)wave_sum_u32(c.c_shadow),
                 s2 = (uint32_t)wave_sum_u32(c.c_shaded), s3 = (uint32_t)wave_sum_u32(c.c_samples),
                 s4 = (uint32_t)wave_sum_u32(c.c_cls);  // four byte-wide sums of at most 64 each: no carry between them
  if (lane == 0) {
    sh_ctr[w][0] = s0;
    sh_ctr[w][1] = s1;
    sh_ctr[w][2] = s2;
    sh_ctr[w][3] = s3;
    sh_ctr[w][4] = s4 & 255u;
    sh_ctr[w][5] = (s4 >> 8) & 255u;
    sh_ctr[w][6] = (s4 >> 16) & 255u;
    sh_ctr[w][7] = s4 >> 24;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    unsigned long long t = 0;
    for (int i = 0; i < NW; ++i) t += sh_ctr[i][threadIdx.x];
    DevCounters* cs = ctr + (blockIdx.x % JADE_CTR_SHARDS);
    // word of DevCounters: primary 0, shadow 1, shaded 4, samples 5, env / indirect / mirror / refract 8-11
    const int i = (int)threadIdx.x, word = i < 2 ? i : i < 4 ? i + 2 : i + 4;
    if (t) atomicAdd(reinterpret_cast<unsigned long long*>(cs) + word, t);
  }
}

// The full shade kernel: one thread per entry of `list` (the active list, or — after k_shade_lean —
// the records that kernel handed over, whose count lives on the device: n_dev).  75 VGPRs, 6 waves/SIMD (JADE_SHADE_WAVES).
template <bool ENVIS>
static __device__ __forceinline__ void shade_kernel_body(const DevScene& S, const PathState& P, const RenderConst& R, const int32_t* tile_ids,
                                               uint32_t target_spp, const uint32_t* list, uint32_t n_host, const uint32_t* n_dev,
                                               uint32_t* active_out, uint32_t* queue, QueueCtl* qc, DevCounters* ctr, const QueueCtl* prev,
                                               uint32_t stop_below) {
  const uint32_t n = n_dev ? *n_dev : n_host;
  // A pass of a batch (prev = the pass before it, on the device): nothing to do once the paths have ended (that pass
  // emitted no ray) or once fewer than stop_below records are active - the point where the host would hand the rest to
  // the next step (carry-over).  Such a pass leaves the lists alone and passes the count on, so that every later pass
  // of the batch stops too and the host finds, in the ring, where the batch really ended.
  if (prev && (prev->count == 0 || n < stop_below)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) qc->active = prev->count == 0 ? 0u : n;
    return;
  }
  if (blockIdx.x * blockDim.x >= n) return;  // block-uniform: the grid is sized for an upper bound of n_dev
  const uint32_t t_idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int p = t_idx < n ? (int)list[t_idx] : P.npix;
  ShadeCtx c;
  c.n_emit_rays = 0;
  c.c_primary = c.c_shadow = c.c_shaded = c.c_samples = c.c_cls = 0;
  uint32_t st;
  bool defer;
  shade_record<false, ENVIS>(S, P, R, tile_ids, target_spp, p, c, st, defer);
  shade_tail<false, JADE_SHADE_NW>(P, p, st, c, false, active_out, nullptr, queue, qc, ctr);
}
__global__ __launch_bounds__(JADE_SHADE_BLOCK, JADE_SHADE_WAVES) void k_shade(DevScene S, PathState P, RenderConst R, const int32_t* tile_ids,
                                               uint32_t target_spp, const uint32_t* list, uint32_t n_host, const uint32_t* n_dev,
                                               uint32_t* active_out, uint32_t* queue, QueueCtl* qc, DevCounters* ctr, const QueueCtl* prev,
                                               uint32_t stop_below) {
  shade_kernel_body<false>(S, P, R, tile_ids, target_spp, list, n_host, n_dev, active_out, queue, qc, ctr, prev, stop_below);
}
// ... with jade_render_params.env_sampling = JADE_ENV_IMPORTANCE (non-parity): the one kernel that carries that code
__global__ __launch_bounds__(JADE_SHADE_BLOCK, JADE_SHADE_WAVES) void k_shade_envis(DevScene S, PathState P, RenderConst R, const int32_t* tile_ids,
                                               uint32_t target_spp, const uint32_t* list, uint32_t n_host, const uint32_t* n_dev,
                                               uint32_t* active_out, uint32_t* queue, QueueCtl* qc, DevCounters* ctr, const QueueCtl* prev,
                                               uint32_t stop_below) {
  shade_kernel_body<true>(S, P, R, tile_ids, target_spp, list, n_host, n_dev, active_out, queue, qc, ctr, prev, stop_below);
}

// ---------------------------------------------------------------------------------------------------------------
// k_shade_binned (round 4; VERDICT r1-r3: "k_shade is one divergent kernel at 13 of 64 lanes").  k_shade's instructions are mostly
// the BRANCH of the bounce it samples - BSSRDF (six powf, the exit-point search), SSS / diffuse (shadow limits, two sphere
// directions), mirror, refraction - and a wave whose 64 records take four different branches runs all four, each for a quarter of
// its lanes.  Which branch a record takes is known after the emissive test and one or two random draws (bounce_classify), and the
// branch itself needs little of the record: its random state, the vertex (triangle, position, outgoing direction) and the
// record's number, through which it writes its rays.  So the block DEALS the records by branch through LDS: every thread loads
// and folds its own record as before (shade_record's part (a)), classifies it, and puts those few words into a pool in which
// every branch's records form a stretch that starts at a multiple of 64; then every thread runs bounce_branch for the pool entry
// with its own index - a wave's 64 entries are one branch - and writes the outcome back; the record's own thread picks it up,
// finishes the sample / starts the next one if the path ended, and stores and queues as ever (shade_tail).  Same statements, same
// draws in the same order per record: every bit and counter is k_shade's (the schedule matrix runs both: JADE_SHADE_BINNED).
// ---------------------------------------------------------------------------------------------------------------
#ifndef JADE_SHADE_BIN_WAVES
#define JADE_SHADE_BIN_WAVES 4 /* 109 VGPRs, no scratch: "rest" of a 1024-spp step of C3 134.6 ms; 5 waves (96 VGPRs, 20 bytes of scratch) 141.8; 6 (80, 128 bytes) 166.7; k_shade: 132.1 */
#endif
#define JADE_BIN_POOL (JADE_SHADE_BLOCK + (BT_N - 1) * 64) /* pool entries: 512 records + the padding of four stretches */
struct alignas(16) BinSlot {  // 48 bytes
  int32_t p;       // the record
  uint32_t rng;    // its random state: in = after bounce_classify's draws, out = after the branch's
  int32_t obj;     // the vertex: triangle ...
  uint32_t sf;     // in: stage | flags << 8 (BT_DIFFUSE: set by bounce_classify); out: stage | flags << 8 | rays emitted << 16 | path ended << 31
  float src[3];    // ... position            (out: l_final of a path that ended)
  float out[3];    // ... outgoing direction
  float pad[2];
};
__global__ __launch_bounds__(JADE_SHADE_BLOCK, JADE_SHADE_BIN_WAVES) void k_shade_binned(DevScene S, PathState P, RenderConst R, const int32_t* tile_ids,
                                               uint32_t target_spp, const uint32_t* list, uint32_t n_host, const uint32_t* n_dev,
                                               uint32_t* active_out, uint32_t* queue, QueueCtl* qc, DevCounters* ctr, const QueueCtl* prev,
                                               uint32_t stop_below) {
  const uint32_t n = n_dev ? *n_dev : n_host;
  if (prev && (prev->count == 0 || n < stop_below)) {  // (as k_shade: a pass of a batch behind the end of the step)
    if (blockIdx.x == 0 && threadIdx.x == 0) qc->active = prev->count == 0 ? 0u : n;
    return;
  }
  if (blockIdx.x * blockDim.x >= n) return;
  __shared__ BinSlot pool[JADE_BIN_POOL];
  __shared__ uint32_t sh_cnt[JADE_SHADE_NW][BT_N];
  __shared__ uint32_t sh_base[BT_N], sh_n[BT_N];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t t_idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int npix = P.npix;
  const int p = t_idx < n ? (int)list[t_idx] : npix;
  ShadeCtx c;
  c.n_emit_rays = 0;
  c.c_primary = c.c_shadow = c.c_shaded = c.c_samples = c.c_cls = 0;
  // ---- (1) the thread's own record: load, fold in the results of the rays issued last pass (shade_record<false>, prologue + (a))
  uint32_t st = ST_INVALID;
  const int pp = p < npix ? p : 0;
  const uint4 hdr0 = P.hdr[pp];
  const uint32_t word = hdr0.z;
  const float4 slot0 = P.slot[(size_t)pp * P.nslots];
  const int hit0 = __float_as_int(slot0.w);
  const jvec3 dir0 = jv(slot0.x, slot0.y, slot0.z);
  const uint32_t rec_m = (uint32_t)pp / (uint32_t)P.npx;
  const int home_pix = (int)((uint32_t)pp - rec_m * (uint32_t)P.npx);
  const int tid0 = tile_ids[home_pix >> 8];
  if (p < npix) st = word & 255u;
  const bool have = st != ST_INVALID;
  const Px px(P, pp);
  uint32_t done = hdr0.y;
  jvec3 l_final = jv(0, 0, 0), color = jv(0, 0, 0);
  bool finished = false;
  int type = -1;  // the branch this record's bounce takes (BT_*), -1 = none this pass
  if (have) {
    const bool on_path = st >= ST_DIFFUSE && st <= ST_REFRACT_EXIT;
    const bool has_ctx = on_path || st == ST_VERTEX;
    c.rng = hdr0.x;
    c.depth = (word >> 8) & 255u;
    c.flags = word >> 16;
    c.stage = st;
    c.thr = jv(1, 1, 1);
    c.acc = jv(0, 0, 0);
    c.le = jv(0, 0, 0);
    c.obj = 0;
    c.src = jv(0, 0, 0);
    c.out = jv(0, 0, 0);
    if (has_ctx) {
      const float4* cx = P.ctx + (size_t)p * 4;
      const float4 b0 = cx[0], b1 = cx[1], b2 = cx[2], b3 = cx[3];
      c.obj = __float_as_int(b0.w);
      if (c.depth != 0) {
        c.thr = jv(b0.x, b0.y, b0.z);
        c.acc = jv(b1.x, b1.y, b1.z);
      }
      if (st == ST_MIRROR && c.depth == 0) c.le = V3(shade_tri(S, c.obj).m->emissive);
      else c.le = jv(b2.x, b2.y, b2.z);
      if (st != ST_MIRROR) {
        c.src = jv(b2.w, b3.x, b3.y);
        c.out = jv(b3.z, b3.w, b1.w);
      }
    }
    if (st == ST_PRIMARY) {
      if (hit0 < 0) {
        color = sample_hdr(S, dir0);  // PathTrace.cu:1443-1445
        finished = true;
      } else {
        c.le = V3(shade_tri(S, hit0).m->emissive);
        c.thr = jv(1, 1, 1);
        c.acc = jv(0, 0, 0);
        c.depth = 0;
        c.obj = hit0;
        c.src = px.hpt(0);
        c.out = jv_neg(dir0);
        st = ST_VERTEX;
      }
    } else if (on_path) {
      const int r = consume(S, px, c, &l_final);
      if (r == CONSUME_VERTEX) {
        st = ST_VERTEX;
      } else if (r == CONSUME_EMITTED) {
        st = c.stage;
      } else {
        color = r == CONSUME_END ? jv_add(c.le, jv_add(c.acc, jv_mul(c.thr, l_final))) : c.le;
        finished = true;
      }
    }
    // ---- (2) the branch of the bounce at a vertex (the emissive test ends the path right here)
    if (!finished && st == ST_VERTEX) {
      type = bounce_classify(S, c, &l_final);
      if (type == BT_END) {
        color = jv_add(c.le, jv_add(c.acc, jv_mul(c.thr, l_final)));
        finished = true;
        type = -1;
      }
    }
  }
  // ---- (3) deal: every branch's records into a stretch of the pool that starts at a multiple of 64
  uint32_t rank = 0;
#pragma unroll
  for (int T = 1; T < BT_N; ++T) {
    const unsigned long long m = __ballot(type == T);
    if (type == T) rank = lanes_below(m);
    if (lane == 0) sh_cnt[w][T] = (uint32_t)__popcll(m);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t base = 0;
    for (int T = 1; T < BT_N; ++T) {
      uint32_t tot = 0;
      for (int i = 0; i < JADE_SHADE_NW; ++i) {
        const uint32_t v = sh_cnt[i][T];
        sh_cnt[i][T] = tot;  // exclusive over the waves
        tot += v;
      }
      sh_base[T] = base;
      sh_n[T] = tot;
      base += (tot + 63u) & ~63u;
    }
    sh_base[0] = base;  // the end of the pool
  }
  __syncthreads();
  uint32_t my_entry = 0;
  if (type >= 1) {
    my_entry = sh_base[type] + sh_cnt[w][type] + rank;
    BinSlot& e = pool[my_entry];
    e.p = p;
    e.rng = c.rng;
    e.obj = c.obj;
    e.sf = c.stage | (c.flags << 8);
    e.src[0] = c.src.x; e.src[1] = c.src.y; e.src[2] = c.src.z;
    e.out[0] = c.out.x; e.out[1] = c.out.y; e.out[2] = c.out.z;
  }
  __syncthreads();
  // ---- (4) the branch, for the pool entry with this thread's index: one branch per wave
  {
    const uint32_t pool_end = sh_base[0];
    for (uint32_t ei = threadIdx.x; ei < pool_end; ei += JADE_SHADE_BLOCK) {  // (a second turn only when the padding pushes the pool past the block)
      int T = 0;
#pragma unroll
      for (int k = 1; k < BT_N; ++k)
        if (ei >= sh_base[k] && ei < sh_base[k] + sh_n[k]) T = k;
      if (T != 0) {
        BinSlot& e = pool[ei];
        ShadeCtx wc;
        wc.rng = e.rng;
        wc.obj = e.obj;
        wc.stage = e.sf & 255u;
        wc.flags = (e.sf >> 8) & 255u;
        wc.depth = 0;
        wc.src = jv(e.src[0], e.src[1], e.src[2]);
        wc.out = jv(e.out[0], e.out[1], e.out[2]);
        wc.thr = wc.acc = wc.le = jv(0, 0, 0);
        wc.n_emit_rays = 0;
        wc.c_primary = wc.c_shadow = wc.c_shaded = wc.c_samples = wc.c_cls = 0;
        const Px wpx(P, e.p);
        jvec3 wl = jv(0, 0, 0);
        const bool emitted = bounce_branch(T, S, wpx, wc, &wl);
        e.rng = wc.rng;
        e.sf = (wc.stage & 255u) | ((wc.flags & 255u) << 8) | ((uint32_t)wc.n_emit_rays << 16) | (emitted ? 0u : 0x80000000u);
        e.src[0] = wl.x; e.src[1] = wl.y; e.src[2] = wl.z;
      }
    }
  }
  __syncthreads();
  // ---- (5) back with the record's own thread: the branch's outcome, then shade_record's part (b) for a sample that ended
  if (type >= 1) {
    const BinSlot& e = pool[my_entry];
    c.rng = e.rng;
    if (e.sf & 0x80000000u) {
      l_final = jv(e.src[0], e.src[1], e.src[2]);
      color = jv_add(c.le, jv_add(c.acc, jv_mul(c.thr, l_final)));
      finished = true;
    } else {
      c.stage = e.sf & 255u;
      c.flags = (e.sf >> 8) & 255u;
      c.n_emit_rays = (int)((e.sf >> 16) & 0x7fffu);
      st = c.stage;
    }
  }
  if (have) {
    if (finished) {
      // final_result = final_result + color (PathTrace.cu:1454), into the partial sum of this sample's (pixel, lane)
      const NextSample cs = next_sample_mh(P, rec_m, (uint32_t)home_pix, done);  // the sample that just ended
      const size_t si = (size_t)(cs.sidx % JADE_SAMPLE_LANES) * (size_t)P.npx + (size_t)cs.pixel;
      const size_t sn = (size_t)P.sum_lanes * (size_t)P.npx;
      st3w(P.sum, sn, si, jv_add(ld3w(P.sum, sn, si), color));
      done += 1;
      c.c_samples += 1;
      st = ST_IDLE;
    }
    if (st == ST_IDLE) {
      // next sample of this record; samples of out-of-image pixels (edge tiles) are skipped
      int x, y;
      NextSample ns = next_sample_mh(P, rec_m, (uint32_t)home_pix, done);
      while (ns.sidx < target_spp && !pixel_xy_t(R, ns.pixel == home_pix ? tid0 : tile_ids[ns.pixel >> 8], ns.pixel, &x, &y)) {
        done += 1;
        ns = next_sample_mh(P, rec_m, (uint32_t)home_pix, done);
      }
      if (ns.sidx < target_spp) {
        // camera ray, PathTrace.cu:1428-1437
        c.rng = jade_rng_seed((uint32_t)x, (uint32_t)y, R.frame + ns.sidx);
        float fx = (float)x + jade_rand(&c.rng);
        double lo = -1.0 + R.two_over_w * ((double)fx - 0.5);
        float left_offset = (float)(lo * R.aspect);
        float fy = (float)y + jade_rand(&c.rng);
        float up_offset = (float)(-1.0 + R.two_over_h * ((double)fy - 0.5));
        jvec3 dir = jade_transform(jv(left_offset, up_offset, -1.5f), 0.0f, R.cam);
        dir = jv_normalize(dir);
        reinterpret_cast<int*>(P.orgs + p)[3] = JADE_SKIP_CAMERA;
        px.set_dir(0, dir);
        px.set_hit(0, -1);
        c.n_emit_rays = 1;
        c.c_primary += 1;
        st = ST_PRIMARY;
      }
    }
    if (st != ST_PRIMARY && c.n_emit_rays) {  // the secondary rays of this pass by hitBVH call site (jade_rt.h)
      if (st == ST_DIFFUSE || st == ST_BSSRDF) {
        const uint32_t ind = (c.flags & STF_RR) ? 1u : 0u;
        const uint32_t env = (c.flags & STF_NOENV) ? 0u : 1u;
        c.c_cls += env | (ind << 8);
        c.c_shadow += (uint32_t)c.n_emit_rays - env - ind;
      } else {
        c.c_cls += st == ST_MIRROR ? (1u << 16) : (1u << 24);
      }
    }
    P.hdr[p] = make_uint4(c.rng, done, st | (c.depth << 8) | (c.flags << 16), 0u);
    if ((st >= ST_DIFFUSE && st <= ST_REFRACT_EXIT) || st == ST_VERTEX) {
      float4* cx = P.ctx + (size_t)p * 4;
      cx[0] = make_float4(c.thr.x, c.thr.y, c.thr.z, __int_as_float(c.obj));
      cx[1] = make_float4(c.acc.x, c.acc.y, c.acc.z, c.out.z);
      cx[2] = make_float4(c.le.x, c.le.y, c.le.z, c.src.x);
      cx[3] = make_float4(c.src.y, c.src.z, c.out.x, c.out.y);
    }
  }
  shade_tail<false, JADE_SHADE_NW>(P, p, st, c, false, active_out, nullptr, queue, qc, ctr);
}

// The lean shade kernel: camera rays, the sky and pure mirrors only — what most records of most
// scenes do most of the time — in 56 VGPRs (8 waves/SIMD; the kernel is latency-bound).  It walks
// ALL records in record order (no list: perfectly coalesced, and nothing to fragment) and hands
// every record that needs anything else to k_shade through heavy_out / qc->heavy.
__global__ __launch_bounds__(JADE_LEAN_BLOCK) void k_shade_lean(DevScene S, PathState P, RenderConst R, const int32_t* tile_ids,
                                                                 uint32_t target_spp, uint32_t* heavy_out, uint32_t* queue,
                                                                 QueueCtl* qc, DevCounters* ctr) {
  const int p = (int)(blockIdx.x * blockDim.x + threadIdx.x);  // >= npix: no record (shade_record checks)
  ShadeCtx c;
  c.n_emit_rays = 0;
  c.c_primary = c.c_shadow = c.c_shaded = c.c_samples = c.c_cls = 0;
  uint32_t st;
  bool defer;
  shade_record<true>(S, P, R, tile_ids, target_spp, p, c, st, defer);
  shade_tail<true, JADE_LEAN_BLOCK / 64>(P, p, st, c, defer, nullptr, heavy_out, queue, qc, ctr);
}

// ---------------------------------------------------------------------------------------------------------------
// Ray-queue ordering (north_star: "ray compaction/sort via wave-ballot/prefix-sum primitives"), OPT-IN (JADE_SORT=1).  Which ray
// sits beside which in the queue is free to choose: every result is written back to the ray's own slot, so no bit of any
// result depends on the order.  A ray's key: the kind of ray (shadow ray towards emitter i / environment ray / indirect ray /
// single-ray stage), then the triangle it leaves - its index in BVH order is a place on the tree's own space-filling curve -
// then the octant it heads into; rocPRIM sorts (key, queue entry) pairs, so the host has to know the queue's length: passes are
// host-followed while this is on.  Measured (profiles/r03_ray_ordering.json, r03_packet_shadow_probe.json): on C3, whose tree
// sits in the L2, a wave whose rays read the same lines is 2.6 % faster and the sort costs 9 %; it is meant for scenes whose
// geometry does not fit the L2 (C5: k_trace is bound by the rate of 64-B sector misses there).
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_ray_keys(PathState P, const uint32_t* queue, uint32_t n, uint32_t* keys, uint32_t* positions, int n_emit, uint32_t tri_bits) {  // tri_bits: bits of the largest triangle index
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  positions[i] = i;  // what the sort moves with the key: the entry's POSITION (PathState.idxq)
  const uint32_t e = queue[i];
  const uint32_t p = e / (uint32_t)P.nslots, k = e - p * (uint32_t)P.nslots;
  const uint32_t st = P.hdr[p].z & 255u;
  const float4 og = P.orgs[p];
  const int32_t skip = __float_as_int(og.w);
  const float4 dv = P.slot[(size_t)e];
  keys[i] = ray_sort_key(st, k, n_emit, skip, dv.x, dv.y, dv.z, tri_bits);
}
// positions 0 .. n-1: what the sort moves with the keys when the keys came from the queueing kernel (written once per render)
__global__ void k_iota(uint32_t* v, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v[i] = i;
}

#define JADE_CTL_RING 96 /* QueueCtl records: entry 0 for passes the host follows one by one, all of them for a batch of passes (round 4: 96 - a 1024-spp step of C3 is ~65 passes down to its carry-over point, the flush ~65 more down to k_tail's threshold: one batch, one wait each; the launches behind the stop are empty) */
#ifndef JADE_TRACE_WAVES
#define JADE_TRACE_WAVES 5 /* waves per SIMD the register allocation leaves room for: 5 = at most 96 VGPRs (12 bytes of scratch) and 5 x 31 KB of LDS per CU.  Round 3, same process, C3 / statue close-up: 4 waves (102 VGPRs) 133.6 / 1037 ms of k_trace per 256-spp step, 5 waves 126.9 / 983 (round 2's "5 and 6 blocks per CU are no faster" was measured on a 102-VGPR build, which the hardware never ran at more than 4) */
#endif
// occluder cache (jade_trace.h): the key of a yes/no query - k: its slot in the record (shadow ray towards emitter k, then the
// environment-visibility ray), d: its direction
static __device__ __forceinline__ uint32_t anyhit_key(uint32_t k, uint32_t n_emit, jvec3 d) {
  const uint32_t oct = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
  return k < n_emit ? (k != 0u ? 1u : 0u) : 2u + oct;
}
typedef uint32_t jade_v4u __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ uint4 ld_anyhit(const uint4* p) {
  const jade_v4u v = NT_LD(reinterpret_cast<const jade_v4u*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
// occluder cache: a query's walk starts with the cached subtrees - the lane's stack begins with them instead of the root
static __device__ __forceinline__ void anyhit_seed(WalkState& r, const LdsStack& stk, uint4 c) {
  uint32_t cur = JADE_REF_NONE, sp = stk.col;
  const uint32_t e4[4] = {c.w, c.z, c.y, c.x};  // way 0 is walked first: the others go under it
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool have = e4[i] != 0u, push = have && cur != JADE_REF_NONE;
    lds_st(push ? sp : stk.col + TW_DUMMY * JADE_COL_STRIDE, cur);
    sp += push ? JADE_COL_STRIDE : 0u;
    cur = have ? e4[i] - 1u : cur;
  }
  if (cur != JADE_REF_NONE) {
    r.cur = cur;
    r.sp = sp;
    r.skipx |= JADE_ATTEMPT;
  }
}
// WIDE: the walk may take wide units (jade_trace.h, "Wide walk"): k_trace_wide, launched instead of k_trace for renders with
// early exits on trees that have wide records.
// TAIL (k_tail): the kernel also SHADES - every wave owns 64 records of a short list and alternates, on its own, k_shade's pass over
// them (shade_record, the rays emitted into a stretch of the queue that is the wave's own) and this loop over those rays, until its
// records have run out of samples.  No other wave is waited for: what a render's last paths cost is their own chain of bounces,
// not one launch (and, at the end of a batch, one host wait) per bounce.
struct TailArgs {
  RenderConst R;
  const int32_t* tile_ids;
  uint32_t target_spp;
  const uint32_t* list;  // the records (the active list)
  uint32_t n_list;
  uint32_t* queue;       // a wave's stretch: 64 * nslots entries
};
template <bool WIDE, bool TAIL = false>
static __device__ __forceinline__ void trace_body(const DevScene& S, const PathState& P, const uint32_t* queue, QueueCtl* qc, uint32_t* spill, DevCounters* ctr,
                                                  uint32_t chunk, const TailArgs* tail = nullptr) {
  __shared__ uint32_t lds_cols[TW_END * JADE_TRACE_BLOCK];
  __shared__ __attribute__((aligned(8))) uint32_t lds_wq[JADE_TRACE_BLOCK / 64][2 * JADE_WQ];  // a wave's ring of leaves to test (jade_trace.h)
  __shared__ __attribute__((aligned(8))) uint32_t lds_hq[JADE_TRACE_BLOCK / 64][2 * JADE_HQ];  // and its ring of hit candidates  // a wave's queue of leaves to test (jade_trace.h)
  const int lane = threadIdx.x & 63;
  const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
  LdsStack stk;
  stk.lds = lds_cols + threadIdx.x;
  stk.col = lds_addr_of(stk.lds);
  stk.spill = spill + gtid;
  stk.stride_spill = gridDim.x * blockDim.x;
  stk.top = nullptr;
  stk.top_k = 0;
  stk.top4 = nullptr;
  stk.top4_k = 0;
  if (!TAIL && qc->count == 0) return;  // (a pass of a batch behind the one that ended the step: nothing was queued)
  // The grid is sized for a full queue (in a batch of passes the host does not know the length); a short queue keeps one
  // block per JADE_TRACE_BLOCK rays and the others leave before they stage anything: the thin passes at the end of a render
  // (under 10 k rays: 40 blocks instead of 1280) were mostly 1280 copies of the tree top into LDS.  The blocks that stay claim
  // chunks until the queue is empty, as ever.
  if (!TAIL && (unsigned long long)blockIdx.x * JADE_TRACE_BLOCK >= (unsigned long long)qc->count) return;
#if JADE_TRACE_TOP_NODES > 0 && JADE_LDS_TOP_NODES > 0
  static_assert(7 * JADE_TRACE_TOP4 <= 4 * JADE_TRACE_TOP_NODES, "the wide top shares the binary top's LDS");
  __shared__ float4 lds_top[4 * JADE_TRACE_TOP_NODES];
  if (WIDE) {
    // the top of the tree as wide records, seven planes (the eighth 16 bytes of a record are unused); a binary unit in this
    // kernel (a ray walked again after a tie, a ray with a non-finite 1/d) reads its records from memory
    const uint32_t k = S.top_k < JADE_TRACE_TOP4 ? S.top_k : JADE_TRACE_TOP4;
    for (uint32_t i = threadIdx.x; i < 8u * k; i += JADE_TRACE_BLOCK)
      if ((i & 7u) != 7u) lds_top[(i & 7u) * JADE_TRACE_TOP4 + (i >> 3)] = S.nodes4[i];
    __syncthreads();
    stk.top4 = lds_top;
    stk.top4_k = k;
    stk.top = lds_top;  // (node_fetch reads entry 0 of four planes for every lane and ignores it: top_k = 0)
    stk.top_k = 0;
  } else {  // stage the top of the tree: record i's j-th 16 bytes -> plane j, entry i (coalesced reads of S.nodes)
    const uint32_t k = S.top_k < JADE_TRACE_TOP_NODES ? S.top_k : JADE_TRACE_TOP_NODES;
    for (uint32_t i = threadIdx.x; i < 4u * k; i += JADE_TRACE_BLOCK) lds_top[(i & 3u) * JADE_TRACE_TOP_NODES + (i >> 2)] = S.nodes[i];
    __syncthreads();
    stk.top = lds_top;
    stk.top_k = k;
  }
#endif
  uint32_t n = TAIL ? 0u : qc->count;
  if (!TAIL && chunk == 0) {  // batched passes: the host has not seen the queue length (trace_chunk's rule, on the device)
    const uint32_t waves = gridDim.x * (JADE_TRACE_BLOCK / 64);
    uint32_t per = n / (waves * 64u * 8u);
    per = per < 1u ? 1u : (per > JADE_TRACE_CHUNK / 64 ? JADE_TRACE_CHUNK / 64 : per);
    chunk = per * 64u;
  }
  uint32_t V = 0, T = 0;  // wave totals (uniform: they live in SGPRs)
  uint32_t vcnt = 0, tcnt = 0;  // per lane, summed over the wave once at the end (a ballot + popcount per unit was 8 instructions)
  uint32_t ccnt = 0;            // ... queries answered by the occluder cache
  const bool anyhit = P.early_exit == 2u && S.anyhit != nullptr;  // (uniform)
  // wave-local chunk of the queue: [lbase, lend) (wave-uniform)
  uint32_t lbase = 0, lend = 0;
  bool queue_empty = false;
  bool active = false;  // this lane walks a ray, or waits for the last of its leaves to be tested
  bool wb = false;      // this lane's ray has ended and its result is still in the LDS column
  uint32_t my_e = 0;    // this lane's queue entry: the slot number, record * nslots + slot
  WalkState r;
  r.od.a = r.od.b = r.od.c = f2{0.0f, 0.0f};
  r.skipx = 0;
  r.cur = JADE_REF_NONE;
  r.sp = stk.col;
  r.pushed = 0;
  r.inv = jv(0, 0, 0);
#if JADE_PREFETCH
  r.pre.a = r.pre.b = r.pre.c = make_float4(0, 0, 0, 0);
  r.pre.rf = make_uint2(0u, 0u);
#endif
  WaveTrace wt;  // the wave's rings of leaves to test and of hit candidates, and the item this lane is testing (jade_trace.h)
  wt.init(lds_addr_of(&lds_wq[threadIdx.x >> 6][0]), lds_addr_of(&lds_hq[threadIdx.x >> 6][0]), lane, WIDE);
#if JADE_COOP_LEAF
  __shared__ __attribute__((aligned(16))) uint32_t lds_stage[JADE_TRACE_BLOCK / 64][64 * 20];
  wt.stage = lds_addr_of(&lds_stage[threadIdx.x >> 6][0]);
#endif
  TraceProf pr;
#if JADE_TRACE_PROFILE
  __shared__ __attribute__((aligned(8))) unsigned long long lds_prof[JADE_TRACE_BLOCK / 64][PL_N];
  pr.begin(lds_addr_of(reinterpret_cast<const uint32_t*>(&lds_prof[threadIdx.x >> 6][0])), lane);
#endif
  // ---- TAIL: this wave's records, and the totals of what shading them counts (shade_tail's counters, per wave here)
  int tail_p = P.npix;     // this lane's record (npix = none)
  bool tail_alive = false;
  unsigned long long tail_rays = 0;  // rays this wave traced (uniform)
  unsigned long long tc_primary = 0, tc_shadow = 0, tc_shaded = 0, tc_samples = 0, tc_env = 0, tc_ind = 0, tc_mirror = 0, tc_refract = 0;  // (valid in lane 0)
  if (TAIL) {
    const uint32_t wave_id = blockIdx.x * (JADE_TRACE_BLOCK / 64) + (threadIdx.x >> 6);
    const uint32_t t_idx = wave_id * 64u + (uint32_t)lane;
    if (t_idx < tail->n_list) {
      tail_p = (int)tail->list[t_idx];
      tail_alive = true;
    }
    queue = tail->queue + (size_t)wave_id * 64u * (uint32_t)P.nslots;
  }
  for (;;) {  // (TAIL: one turn per bounce of the wave's records; otherwise one turn)
  if (TAIL) {
    // ---- k_shade's pass over the wave's records: last turn's hit results folded in, the next bounce's rays emitted
    __threadfence();  // the results this wave's lanes wrote for each other's records
    ShadeCtx c;
    c.n_emit_rays = 0;
    c.c_primary = c.c_shadow = c.c_shaded = c.c_samples = c.c_cls = 0;
    uint32_t st = ST_INVALID;
    if (tail_alive) {
      bool defer;
      shade_record<false>(S, P, tail->R, tail->tile_ids, tail->target_spp, tail_p, c, st, defer);
      tail_alive = c.n_emit_rays > 0;  // a record without a ray in flight has run out of samples
    }
    tc_primary += wave_sum_u32(c.c_primary); tc_shadow += wave_sum_u32(c.c_shadow); tc_shaded += wave_sum_u32(c.c_shaded); tc_samples += wave_sum_u32(c.c_samples);
    tc_env += wave_sum_u32(c.c_cls & 255u); tc_ind += wave_sum_u32((c.c_cls >> 8) & 255u); tc_mirror += wave_sum_u32((c.c_cls >> 16) & 255u); tc_refract += wave_sum_u32(c.c_cls >> 24);
    if (__ballot(tail_alive) == 0ull) break;
    // the wave's rays, slot by slot as shade_tail queues them
    uint32_t m_rays = 0;
    const int used = !tail_alive ? 0 : (st == ST_DIFFUSE || st == ST_BSSRDF) ? P.nslots : 1;
    for (int k = 0; k < P.nslots; ++k) {
      const bool q = k < used && reinterpret_cast<const int*>(P.slot + ((size_t)tail_p * P.nslots + k))[3] != -2;
      const unsigned long long m = __ballot(q);
      if (q) const_cast<uint32_t*>(queue)[m_rays + wt.rank_in(m)] = (uint32_t)tail_p * (uint32_t)P.nslots + (uint32_t)k;
      m_rays += (uint32_t)__popcll(m);
    }
    __threadfence();  // rays and queue entries are read by other lanes of this wave
    n = m_rays;
    tail_rays += m_rays;
    lbase = 0;
    lend = n;
    queue_empty = false;
  }
  for (;;) {
    // ---- a ray has ended when its walk has and all the leaves it pushed have been finished
    if (active && WaveTrace::ray_ended(r, stk)) {
      // (occluder cache: the cached subtrees held no answer - the whole walk, from the root)
      const bool retry = (r.skipx & (JADE_ATTEMPT | JADE_CUT)) == JADE_ATTEMPT;
      // (wide walk: two leaves tied for this ray's best distance and the walk was not the reference's order - once more, with
      // binary units only; a ray that ended early has its answer whatever the order)
      const bool again = WIDE && lds_get(stk, TW_LIMIT) == JADE_LIMIT_TIE && (r.skipx & (JADE_CUT | JADE_FORCE_BINARY)) == 0u;
      if (retry || again) {
        // the ray is still in the lane's registers (origin, directions, source triangle, the limit in its column): only the walk
        // starts over - from the root, with nothing found (a hit of the first walk is found again; keeping it would mix the two
        // walks' leaf numbers in hitArray's tie rule).  No memory access: this runs whenever ANY lane of the wave restarts.
        walk_restart(r, stk, S);
        r.skipx = (r.skipx & ~JADE_ATTEMPT) | (retry ? 0u : JADE_FORCE_BINARY);
        if (WIDE && lds_get(stk, TW_LIMIT) == JADE_LIMIT_TIE)  // (rare: the tie marker sits where the limit was - the ray's own comes back from its slot)
          lds_putf(stk, TW_LIMIT, P.early_exit ? NT_LD(reinterpret_cast<const float*>(P.slot + (size_t)my_e) + 3) : __int_as_float(-1));
      } else {
        active = false;
        wb = true;
      }
    }
    // ---- once enough lanes are idle (or all are): write their results back and refill them.  Both are done for >=
    // JADE_REFILL_MIN lanes at a time, not whenever a single ray ends: a block of code that
    // runs for one lane costs as much as for 64.
    const unsigned long long idle = __ballot(!active);
    const int n_idle = __popcll(idle);
    PROF_DRAIN();
    PROF_LAP(pr, PL_TOP);
    if (n_idle >= JADE_REFILL_MIN) {
      PROF_COUNT(pr, PC_REFILLS, 1);
      if (wb) {
        // ray records stream through once per pass: non-temporal, so that they do not push the BVH out of the XCD's
        // 4 MB L2 (C3's node + vertex records are 4.1 MB; PMC: 124 of the 172 HBM bytes per ray were BVH lines re-fetched)
        float dist;
        jvec3 hp = jv(0, 0, 0);
        uint32_t parent1 = 0;
        // (a yes/no query gets the triangle alone: nobody reads its hit point - jade_device.h, PathState.hitp)
        const bool want_point = (r.skipx & JADE_WANTS_POINT) != 0u || P.write_all_hits != 0u;
        const int32_t best = walk_result(stk, S, r.od, &dist, &hp, &parent1, want_point);
        NT_ST(reinterpret_cast<int32_t*>(P.slot + (size_t)my_e) + 3, best);
        if (want_point) nt_st4(P.hitp + (size_t)(my_e / (uint32_t)P.nslots), hp.x, hp.y, hp.z, dist);  // (the hit point of a miss is never read; its distance stays INF, PathTrace.cu:799)
        if (anyhit) {
          // occluder cache: an answer the cached subtrees gave is counted; one that took the whole walk leaves its leaf's parent
          // in one of the key's four ways (a plain store: entries are hints)
          ccnt += (r.skipx & (JADE_CUT | JADE_ATTEMPT)) == (JADE_CUT | JADE_ATTEMPT) ? 1u : 0u;
          const uint32_t src = r.skipx & JADE_SKIP_MASK;
          const float lim = lds_getf(stk, TW_LIMIT);
          if ((r.skipx & (JADE_CUT | JADE_ATTEMPT | 0x80000000u)) == JADE_CUT && parent1 != 0u && lim == lim && src != JADE_SKIP_MASK) {
            const uint32_t k = my_e - (my_e / (uint32_t)P.nslots) * (uint32_t)P.nslots;
            const jvec3 dn = od_dn(r.od);
            const uint32_t key = anyhit_key(k, (uint32_t)S.n_emit, dn);
            const uint32_t way = (my_e * 2654435761u) >> 30;
            reinterpret_cast<uint32_t*>(S.anyhit)[((size_t)src * JADE_ANYHIT_KEYS + key) * 4u + way] = parent1;
          }
        }
        wb = false;
      }
      PROF_DRAIN();
      PROF_LAP(pr, PL_WRITEBACK);
      if (!queue_empty) {
        if (TAIL && lbase >= lend) queue_empty = true;  // (the wave's own stretch: nothing to claim.  Found here, with nothing taken -
        // not when the last entries are taken: the lanes that took them are counted idle by this turn of the loop)
        if (!TAIL && lbase >= lend) {
          uint32_t nb = 0;
          if (lane == 0) nb = atomicAdd(&qc->next, chunk);
          nb = __shfl(nb, 0, 64);
          if (nb >= n) {
            queue_empty = true;
            lbase = lend = n;
          } else {
            lbase = nb;
            lend = nb + chunk < n ? nb + chunk : n;
          }
        }
        const uint32_t avail = lend - lbase;
        const uint32_t take = (uint32_t)n_idle < avail ? (uint32_t)n_idle : avail;
        const uint32_t rank = wt.rank_in(idle);
        // (an ordered queue holds positions: PathState.idxq)
        uint32_t qpos = lbase + rank;
        if (!TAIL && P.idxq && !active && rank < take) qpos = NT_LD(&queue[lbase + rank]);
        if (!TAIL && !active && rank < take && qpos < P.rayq_cap) {
          // ---- the ray as a record (PathState.rayq): three 16-B loads (coalesced unless the queue is ordered), nothing to compute
          const float4* rq = P.rayq + (size_t)qpos * 3;
          const float4 r0 = nt_ld4(rq), r1 = nt_ld4(rq + 1), r2 = nt_ld4(rq + 2);
          my_e = __float_as_uint(r2.w);
          const uint32_t skipx = __float_as_uint(r0.w);
          const uint32_t src = skipx & JADE_SKIP_MASK;
          const bool anyq = anyhit && r1.w == r1.w && src != JADE_SKIP_MASK;
          uint4 c = make_uint4(0u, 0u, 0u, 0u);
          if (anyq) {
            const uint32_t p = my_e / (uint32_t)P.nslots;
            c = ld_anyhit(S.anyhit + (size_t)src * JADE_ANYHIT_KEYS + anyhit_key(my_e - p * (uint32_t)P.nslots, (uint32_t)S.n_emit, jv(r2.x, r2.y, r2.z)));
          }
          walk_begin_prepared(r, stk, S, jv(r0.x, r0.y, r0.z), jv(r1.x, r1.y, r1.z), jv(r2.x, r2.y, r2.z), skipx, r1.w);
          if (anyq && (int32_t)skipx >= 0) anyhit_seed(r, stk, c);
          active = true;
        } else if (!active && rank < take) {
          my_e = (!TAIL && P.idxq) ? NT_LD(&P.idxq[qpos]) : NT_LD(&queue[lbase + rank]);
          const uint32_t p = my_e / (uint32_t)P.nslots;  // the entry is the slot number p * nslots + k
          const float4 og = nt_ld4(&P.orgs[p]);
          const int32_t skip = __float_as_int(og.w);
          const jvec3 o = skip == JADE_SKIP_CAMERA ? jv(P.eye[0], P.eye[1], P.eye[2]) : jv(og.x, og.y, og.z);
          const float4 dv = nt_ld4(&P.slot[(size_t)my_e]);
          const jvec3 d = jv(dv.x, dv.y, dv.z);
          // occluder cache (jade_trace.h): a yes/no query starts with the subtrees in which the last such queries from this
          // triangle found their answer - the lane's stack begins with them instead of the root.  (The entry is requested before
          // walk_begin's arithmetic so that the two overlap.)
          const bool anyq = anyhit && dv.w == dv.w && skip >= 0;
          uint4 c = make_uint4(0u, 0u, 0u, 0u);
          if (anyq) c = ld_anyhit(S.anyhit + (size_t)skip * JADE_ANYHIT_KEYS + anyhit_key(my_e - p * (uint32_t)P.nslots, (uint32_t)S.n_emit, d));
          walk_begin(r, stk, S, o, d, skip, P.early_exit ? dv.w : __int_as_float(-1), dv.w);
          if (anyq && (int32_t)r.skipx >= 0) anyhit_seed(r, stk, c);  // (a ray with a non-finite 1/d takes the NaN-faithful walk from the root)
          active = true;
        }
        V += take;  // the root record of every ray started
        lbase += take;
      }
      PROF_DRAIN();
      PROF_LAP(pr, PL_REFILL);
    }
    if (n_idle == 64 && queue_empty) break;  // nothing in flight (so no leaf is waiting either), nothing left to claim
    // ---- one iteration of work for the wave: walk units, test units or a pass over the hit candidates (jade_trace.h)
    wt.template iterate<WIDE>(r, active, S, stk, vcnt, tcnt, pr);
  }
  if (!TAIL) break;
  }
  if (TAIL && lane == 0) {  // what shading counted (shade_tail's words of DevCounters)
    DevCounters* cs = ctr + (blockIdx.x % JADE_CTR_SHARDS);
    if (tc_primary) atomicAdd(&cs->rays_primary, tc_primary);
    if (tc_shadow) atomicAdd(&cs->rays_shadow, tc_shadow);
    if (tc_shaded) atomicAdd(&cs->shaded_hits, tc_shaded);
    if (tc_samples) atomicAdd(&cs->samples, tc_samples);
    if (tc_env) atomicAdd(&cs->rays_env, tc_env);
    if (tc_ind) atomicAdd(&cs->rays_indirect, tc_ind);
    if (tc_mirror) atomicAdd(&cs->rays_mirror, tc_mirror);
    if (tc_refract) atomicAdd(&cs->rays_refract, tc_refract);
  }
#if JADE_TRACE_PROFILE
  pr.count(PC_WAVES, 1);
  PROF_DRAIN();
  if (lane < PL_N) {
    const unsigned long long v = pr.get(lane);
    if (v) atomicAdd(&g_trace_prof[lane], v);
  }
#endif
  V += (uint32_t)wave_sum_u32(vcnt);  // (valid in lane 0, the only lane that uses it)
  T += (uint32_t)wave_sum_u32(tcnt);
  if (lane == 0) {
    DevCounters* cs = ctr + (blockIdx.x % JADE_CTR_SHARDS);
    if (V) atomicAdd(&cs->nodes_visited, (unsigned long long)V);
    if (T) atomicAdd(&cs->tris_tested, (unsigned long long)T);
    if (TAIL) {  // k_tail's own share (jade_stats.rays_tail ...): its rays are not k_trace's
      if (tail_rays) atomicAdd(&cs->rays_tail, tail_rays);
      if (V) atomicAdd(&cs->nodes_tail, (unsigned long long)V);
      if (T) atomicAdd(&cs->tris_tail, (unsigned long long)T);
    }
  }
  if (anyhit) {
    const uint32_t C = (uint32_t)wave_sum_u32(ccnt);
    if (lane == 0 && C) atomicAdd(&(ctr + (blockIdx.x % JADE_CTR_SHARDS))->rays_cached, (unsigned long long)C);
  }
}
__global__ __launch_bounds__(JADE_TRACE_BLOCK, JADE_TRACE_WAVES) void k_trace(DevScene S, PathState P, const uint32_t* queue, QueueCtl* qc,
                                                           uint32_t* spill, DevCounters* ctr, uint32_t chunk) {
  trace_body<false>(S, P, queue, qc, spill, ctr, chunk);
}
#ifndef JADE_TRACE_WIDE_WAVES
#define JADE_TRACE_WIDE_WAVES 4 /* 106 VGPRs, no scratch.  At 5 (96 VGPRs, 20 bytes of scratch) C5's k_trace takes 166.6 instead of 160.8 ms per step: the wide unit's four boxes want the registers more than the kernel wants the fifth wave */
#endif
__global__ __launch_bounds__(JADE_TRACE_BLOCK, JADE_TRACE_WIDE_WAVES) void k_trace_wide(DevScene S, PathState P, const uint32_t* queue, QueueCtl* qc,
                                                                uint32_t* spill, DevCounters* ctr, uint32_t chunk) {
  trace_body<true>(S, P, queue, qc, spill, ctr, chunk);
}
#ifndef JADE_TAIL_WAVES
#define JADE_TAIL_WAVES 2 /* k_tail holds k_shade's registers and k_trace's at once (152 VGPRs; at 4 waves per SIMD it spills 372 bytes); at most a few hundred waves ever run, so occupancy is not what it needs */
#endif
#ifndef JADE_TAIL_MAX
#define JADE_TAIL_MAX 32768u /* records: a shorter active list is finished by k_tail instead of by further passes */
#endif
// k_tail: the end of a render's (or a small render's) paths in ONE launch - see TailArgs.  Binary units only: the walk is the
// reference's in every mode, and early exits work as in k_trace (the limit travels in the slot).
__global__ __launch_bounds__(JADE_TRACE_BLOCK, JADE_TAIL_WAVES) void k_tail(DevScene S, PathState P, TailArgs tail, uint32_t* spill, DevCounters* ctr) {
  trace_body<false, true>(S, P, nullptr, nullptr, spill, ctr, 64u, &tail);
}
// which of the two a launch takes: wide units need early exits (the order of the walk is then nobody's business but a tie's) and
// a tree with wide records
static inline bool trace_wide(const DevScene& S, const PathState& P) { return JADE_WIDE_WALK && P.early_exit && S.nodes4 != nullptr; }

// ---------------------------------------------------------------------------------------------------------------
// k_light: the first pass of a step, fused.  Most samples of most frames are LIGHT: the camera ray misses (sky), or it
// meets an emitter, or a pure mirror whose reflected ray then misses (the floor).  Run as shade / trace passes, such a
// sample crosses memory six times (camera ray out, its result back, mirror ray out, ...) for a handful of arithmetic:
// k_shade_lean was bound by those bytes (3.5 TB/s of HBM, 15 % of a step) and the rays' records by their round trip
// through the queue.  Here a record keeps its light samples in registers: it generates the camera ray, traces it IN
// THIS KERNEL (the same node / triangle steps as k_trace, a wave's rays side by side until all have ended - camera and
// floor-mirror rays of neighbouring pixels are coherent), folds the result in, follows a pure mirror, adds the finished
// sample to its partial sum and starts the next one, until it runs out of samples or meets a surface the light path
// cannot shade (jade, diffuse, glass).  Then it parks the path exactly where k_shade_lean would have (ST_VERTEX, context
// stored) and hands the record to k_shade through the same list.  Statements, draw order and sums are those of
// shade_record<true> + k_trace, so every bit of the result is the same (test_result_independent_of_shade_schedule).
// ---------------------------------------------------------------------------------------------------------------
#ifndef JADE_LIGHT_WAVES
#define JADE_LIGHT_WAVES 4
#endif
#ifndef JADE_LIGHT_REFILL_MIN
#define JADE_LIGHT_REFILL_MIN 32 /* waiting lanes of a wave that trigger k_light's shading block */
#endif
__global__ __launch_bounds__(JADE_TRACE_BLOCK, JADE_LIGHT_WAVES) void k_light(DevScene S, PathState P, RenderConst R, const int32_t* tile_ids,
                                                                            uint32_t target_spp, uint32_t* heavy_regions, uint32_t region_cap,
                                                                            uint32_t* wave_counts, uint32_t* spill, DevCounters* ctr) {
  __shared__ __attribute__((aligned(JADE_COLS_ALIGN))) uint32_t lds_cols[LW_END * JADE_TRACE_BLOCK];
  __shared__ uint32_t sh_ctr[JADE_TRACE_BLOCK / 64][8];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  LdsStack stk;
  stk.lds = lds_cols + threadIdx.x;
  stk.col = lds_addr_of(stk.lds);
  stk.spill = spill + (blockIdx.x * blockDim.x + threadIdx.x);
  stk.stride_spill = gridDim.x * blockDim.x;
  stk.top = nullptr;
  stk.top_k = 0;
#if JADE_LDS_TOP_NODES > 0
  __shared__ float4 lds_top[4 * JADE_LDS_TOP_NODES];
  {
    const uint32_t k = S.top_k;
    for (uint32_t i = threadIdx.x; i < 4u * k; i += JADE_TRACE_BLOCK) lds_top[(i & 3u) * JADE_LDS_TOP_NODES + (i >> 2)] = S.nodes[i];
    __syncthreads();
    stk.top = lds_top;
    stk.top_k = k;
  }
#endif
  const int npix = P.npix;
  const size_t sn = (size_t)P.sum_lanes * (size_t)P.npx;
  uint32_t vcnt = 0, tcnt = 0, n_mirror = 0;
  ShadeCtx c;
  c.n_emit_rays = 0;
  c.c_primary = c.c_shadow = c.c_shaded = c.c_samples = c.c_cls = 0;
  // records are dealt to the resident blocks in chunks of one block (a persistent grid: the LDS copy of the tree top is
  // made once per block, not once per 256 records)
  // (per WAVE: a wave that is through its 64 records takes its next 64 at once - no barrier, no shared counter; the
  // records it hands to k_shade go into a region of the list that is this wave's own, compacted by k_heavy_pack)
  const uint32_t wave_id = blockIdx.x * (JADE_TRACE_BLOCK / 64) + (uint32_t)w;
  uint32_t* const my_region = heavy_regions + (size_t)wave_id * region_cap;
  uint32_t n_deferred = 0;
  for (size_t base = (size_t)wave_id * 64; base < (size_t)npix; base += (size_t)gridDim.x * JADE_TRACE_BLOCK) {
    const size_t p64 = base + (size_t)lane;
    const bool have = p64 < (size_t)npix;
    const int p = have ? (int)p64 : 0;
    const uint4 hdr0 = P.hdr[p];
    const uint32_t word = hdr0.z;
    uint32_t st = have ? (word & 255u) : (uint32_t)ST_INVALID;
    const uint32_t rec_m = (uint32_t)p / (uint32_t)P.npx;
    const int home_pix = (int)((uint32_t)p - rec_m * (uint32_t)P.npx);
    const int tid0 = tile_ids[home_pix >> 8];
    uint32_t done = hdr0.y;
    bool defer = false;  // hand the record to k_shade
    // a record that is in the middle of a path (carried over from the last step, its rays traced and their results
    // waiting in memory) goes to k_shade untouched, as k_shade_lean does; an idle one is this kernel's to run
    bool mine = st == ST_IDLE;
    const bool untouched = have && st != ST_IDLE && st != ST_INVALID;
    if (untouched) defer = true;
    c.rng = hdr0.x;
    c.depth = 0;
    c.flags = 0;
    c.thr = jv(1, 1, 1);
    c.acc = jv(0, 0, 0);
    c.le = jv(0, 0, 0);
    c.obj = 0;
    c.src = jv(0, 0, 0);
    c.out = jv(0, 0, 0);
    RegPx rp;
    rp.d = jv(0, 0, 0);
    rp.o = jv(0, 0, 0);
    rp.hp = jv(0, 0, 0);
    rp.h = -1;
    rp.sk = -1;
    bool finished = false;
    jvec3 color = jv(0, 0, 0), l_final = jv(0, 0, 0);
    // A lane is tracing (`active`), or waits to be shaded: its ray's result folded in and the next ray set up.  Shading
    // runs for >= JADE_LIGHT_REFILL_MIN waiting lanes at a time (or when nothing is being traced): it is k_trace's refill, with
    // the next ray coming from the lane's own path instead of from a queue.
    bool active = false;   // a ray in flight
    bool pending = false;  // its result has not been folded in yet
    RayState r;
    ray_clear(r, stk);
    for (;;) {
      const unsigned long long tracing = __ballot(active);
      const int n_wait = __popcll(__ballot(mine && !active));
      if (tracing == 0ull && n_wait == 0) break;  // every lane is out of samples or parked
      if (n_wait >= JADE_LIGHT_REFILL_MIN || tracing == 0ull) {
        if (mine && !active) {
          // ---- fold the result in (shade_record's part (a))
          if (pending) {
            pending = false;
            rp.h = ray_best_index(stk);
            if (rp.h >= 0) rp.hp = ray_hit_point(stk);
            if (st == ST_PRIMARY) {
              if (rp.h < 0) {
                color = sample_hdr(S, rp.d);  // PathTrace.cu:1443-1445
                finished = true;
              } else {
                c.le = V3(shade_tri(S, rp.h).m->emissive);
                c.thr = jv(1, 1, 1);
                c.acc = jv(0, 0, 0);
                c.depth = 0;
                c.obj = rp.h;
                c.src = rp.hp;
                c.out = jv_neg(rp.d);
                st = ST_VERTEX;
              }
            } else {  // ST_MIRROR
              c.stage = ST_MIRROR;
              const int rr = consume_mirror(S, rp, c, &l_final);
              if (rr == CONSUME_VERTEX) {
                st = ST_VERTEX;
              } else {
                color = jv_add(c.le, jv_add(c.acc, jv_mul(c.thr, l_final)));
                finished = true;
              }
            }
          }
          // ---- advance until this lane has a ray to trace, is out of samples, or parks (part (b))
          bool ray = false;
          for (;;) {
            if (finished) {
              const NextSample cs = next_sample_mh(P, rec_m, (uint32_t)home_pix, done);  // the sample that just ended
              const size_t si = (size_t)(cs.sidx % JADE_SAMPLE_LANES) * (size_t)P.npx + (size_t)cs.pixel;
              st3w(P.sum, sn, si, jv_add(ld3w(P.sum, sn, si), color));
              done += 1;
              c.c_samples += 1;
              finished = false;
              st = ST_IDLE;
            }
            if (st == ST_VERTEX) {
              if (!lean_can_shade(shade_tri(S, c.obj).m)) {  // jade / diffuse / glass: k_shade continues from here
                defer = true;
                mine = false;
                break;
              }
              if (begin_bounce_lean(S, rp, c, &l_final)) {  // the mirror ray is in rp
                n_mirror += 1;
                st = ST_MIRROR;
                ray = true;
                break;
              }
              color = jv_add(c.le, jv_add(c.acc, jv_mul(c.thr, l_final)));
              finished = true;
              continue;
            }
            // ST_IDLE: the next sample of this record; samples of out-of-image pixels (edge tiles) are skipped
            int x, y;
            NextSample ns = next_sample_mh(P, rec_m, (uint32_t)home_pix, done);
            while (ns.sidx < target_spp && !pixel_xy_t(R, ns.pixel == home_pix ? tid0 : tile_ids[ns.pixel >> 8], ns.pixel, &x, &y)) {
              done += 1;
              ns = next_sample_mh(P, rec_m, (uint32_t)home_pix, done);
            }
            if (ns.sidx >= target_spp) {  // nothing left for this record in this step
              mine = false;
              break;
            }
            // camera ray, PathTrace.cu:1428-1437 (the statements of shade_record)
            c.rng = jade_rng_seed((uint32_t)x, (uint32_t)y, R.frame + ns.sidx);
            float fx = (float)x + jade_rand(&c.rng);
            double lo = -1.0 + R.two_over_w * ((double)fx - 0.5);
            float left_offset = (float)(lo * R.aspect);
            float fy = (float)y + jade_rand(&c.rng);
            float up_offset = (float)(-1.0 + R.two_over_h * ((double)fy - 0.5));
            jvec3 dir = jade_transform(jv(left_offset, up_offset, -1.5f), 0.0f, R.cam);
            dir = jv_normalize(dir);
            rp.set_origin(jv(P.eye[0], P.eye[1], P.eye[2]), JADE_SKIP_CAMERA);
            rp.set_dir(0, dir);
            c.c_primary += 1;
            c.depth = 0;
            c.flags = 0;
            st = ST_PRIMARY;
            ray = true;
            break;
          }
          if (ray) {
            ray_begin(r, stk, S, rp.o, rp.d, rp.sk);
            vcnt += 1;  // the root record
            active = true;
            pending = true;
          }
        }
        if (__ballot(active) == 0ull) continue;  // (every lane went out or parked: the loop ends at its top)
      }
      // ---- one kind of work for every tracing lane that has some (hitBVH, PathTrace.cu:795-859).  The lane tests the leaves
      // of its own ray here (a FIFO per lane, jade_trace.h): camera and floor-mirror rays of neighbouring pixels are short and
      // walk side by side, and k_trace's wave-wide queue of leaves costs this kernel more than it saves (282 vs 238 ms per step)
      {
        const bool cw = active && ray_can_walk(r), ct = active && ray_can_test(r);
        const int nw = __popcll(__ballot(cw)), nt = __popcll(__ballot(ct));
        if (JADE_COST_TRI * nw >= JADE_COST_NODE * nt) {
          if (S.general_walk || __ballot(active && (int32_t)r.skipx < 0) != 0ull) {
#pragma nounroll
            for (int rep = 0; rep < JADE_STEPS_PER_PICK; ++rep)
              if (active && ray_can_walk(r)) ray_step_node_s<true>(r, S, stk, vcnt);
          } else {
#pragma nounroll
            for (int rep = 0; rep < JADE_STEPS_PER_PICK; ++rep)
              if (active && ray_can_walk(r)) ray_step_node_s<false>(r, S, stk, vcnt);
          }
        } else {
#pragma nounroll
          for (int rep = 0; rep < JADE_STEPS_PER_PICK; ++rep)
            if (active && ray_can_test(r)) ray_step_tri_s(r, S, stk, tcnt);
        }
        if (active && ray_done(r)) active = false;  // its result stays in the LDS column until the lane is shaded
      }
    }
    // ---- store what the next kernel needs
    if (have && st != ST_INVALID && !untouched) {  // (a carried-over record was not touched)
      P.hdr[p] = make_uint4(c.rng, done, st == ST_VERTEX ? ST_VERTEX | (c.depth << 8) | (c.flags << 16) : (uint32_t)ST_IDLE, 0u);
      if (st == ST_VERTEX) {  // parked: the path context, as shade_record stores it for this stage
        float4* cx = P.ctx + (size_t)p * 4;
        cx[0] = make_float4(c.thr.x, c.thr.y, c.thr.z, __int_as_float(c.obj));
        cx[1] = make_float4(c.acc.x, c.acc.y, c.acc.z, c.out.z);
        cx[2] = make_float4(c.le.x, c.le.y, c.le.z, c.src.x);
        cx[3] = make_float4(c.src.y, c.src.z, c.out.x, c.out.y);
      }
    }
    // ---- hand-over: append to this wave's region
    {
      const unsigned long long dm = __ballot(defer);
      if (defer) my_region[n_deferred + lanes_below(dm)] = (uint32_t)p;
      n_deferred += (uint32_t)__popcll(dm);
    }
  }
  if (lane == 0) wave_counts[wave_id] = n_deferred;
  // ---- work counters: per wave into LDS, then one set of atomics per block (as shade_tail), and V / T as k_trace
  {
    const uint32_t s0 = (uint32_t)wave_sum_u32(c.c_primary), s2 = (uint32_t)wave_sum_u32(c.c_shaded), s3 = (uint32_t)wave_sum_u32(c.c_samples);
    const unsigned long long s6 = wave_sum_u32(n_mirror);
    const unsigned long long sv = wave_sum_u32(vcnt), stt = wave_sum_u32(tcnt);
    if (lane == 0) {
      sh_ctr[w][0] = s0;
      sh_ctr[w][1] = 0;
      sh_ctr[w][2] = s2;
      sh_ctr[w][3] = s3;
      sh_ctr[w][4] = 0;
      sh_ctr[w][5] = 0;
      sh_ctr[w][6] = (uint32_t)s6;
      sh_ctr[w][7] = 0;
      DevCounters* cs = ctr + (blockIdx.x % JADE_CTR_SHARDS);
      if (sv) atomicAdd(&cs->nodes_visited, sv);
      if (stt) atomicAdd(&cs->tris_tested, stt);
      if (sv) atomicAdd(&cs->nodes_inline, sv);
      if (stt) atomicAdd(&cs->tris_inline, stt);
      if (s0 + s6) atomicAdd(&cs->rays_inline, (unsigned long long)s0 + s6);  // every camera and mirror ray of this kernel was traced here
    }
    __syncthreads();
    if (threadIdx.x < 8) {
      unsigned long long t = 0;
      for (int i = 0; i < JADE_TRACE_BLOCK / 64; ++i) t += sh_ctr[i][threadIdx.x];
      DevCounters* cs = ctr + (blockIdx.x % JADE_CTR_SHARDS);
      const int i = (int)threadIdx.x, wordi = i < 2 ? i : i < 4 ? i + 2 : i + 4;
      if (t) atomicAdd(reinterpret_cast<unsigned long long*>(cs) + wordi, t);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k_light_packet: k_light with the wave's rays walked together (jade_trace.h, "Packet form").  What k_light traces - camera
// rays of 64 neighbouring pixels, their reflections off the mirror floor - is as coherent as rays get, and its time is the
// traversal (two thirds of its instructions): one scalar cursor and stack per wave, the node and pair records through
// scalar loads, no per-lane stack / FIFO / column in LDS.  A wave alternates two phases so that a packet holds one kind of
// ray: all its lanes' camera rays, then the mirror rays those produced.  Shading statements, draw order and sums are those
// of k_light (and of shade_record<true> + k_trace): the same bits (test_result_independent_of_shade_schedule).
// ---------------------------------------------------------------------------------------------------------------
#ifndef JADE_PACKET_WAVES
#define JADE_PACKET_WAVES 4 /* 128 VGPRs; at 5 (96 VGPRs, 168 bytes of scratch) the kernel took 228 instead of 191 ms per step on C3 */
#endif
__global__ __launch_bounds__(JADE_TRACE_BLOCK, JADE_PACKET_WAVES) void k_light_packet(DevScene S, PathState P, RenderConst R, const int32_t* tile_ids,
                                                                                   uint32_t target_spp, uint32_t* heavy_regions, uint32_t region_cap,
                                                                                   uint32_t* wave_counts, DevCounters* ctr, uint32_t budget) {
  __shared__ __attribute__((aligned(16))) uint32_t lds_stack[JADE_TRACE_BLOCK / 64][8 * (JADE_PACKET_MAX_DEPTH + 1)];
  __shared__ uint32_t sh_ctr[JADE_TRACE_BLOCK / 64][8];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t stack = lds_addr_of(&lds_stack[w][0]);
  uint32_t n_given_up = 0, n_packets = 0;
  TraceProf pr;
#if JADE_TRACE_PROFILE
  __shared__ __attribute__((aligned(8))) unsigned long long lds_pprof[JADE_TRACE_BLOCK / 64][PKL_N > PL_N ? PKL_N : PL_N];
  pr.begin(lds_addr_of(reinterpret_cast<const uint32_t*>(&lds_pprof[w][0])), lane);
#endif
  const int npix = P.npix;
  const size_t sn = (size_t)P.sum_lanes * (size_t)P.npx;
  uint32_t vcnt = 0, tcnt = 0, n_mirror = 0;
  ShadeCtx c;
  c.n_emit_rays = 0;
  c.c_primary = c.c_shadow = c.c_shaded = c.c_samples = c.c_cls = 0;
  const uint32_t wave_id = blockIdx.x * (JADE_TRACE_BLOCK / 64) + (uint32_t)w;
  uint32_t* const my_region = heavy_regions + (size_t)wave_id * region_cap;
  uint32_t n_deferred = 0;
  for (size_t base = (size_t)wave_id * 64; base < (size_t)npix; base += (size_t)gridDim.x * JADE_TRACE_BLOCK) {
    const size_t p64 = base + (size_t)lane;
    const bool have = p64 < (size_t)npix;
    const int p = have ? (int)p64 : 0;
    const uint4 hdr0 = P.hdr[p];
    const uint32_t word = hdr0.z;
    uint32_t st = have ? (word & 255u) : (uint32_t)ST_INVALID;
    const uint32_t rec_m = (uint32_t)p / (uint32_t)P.npx;
    const int home_pix = (int)((uint32_t)p - rec_m * (uint32_t)P.npx);
    const int tid0 = tile_ids[home_pix >> 8];
    uint32_t done = hdr0.y;
    bool defer = false;
    bool mine = st == ST_IDLE;  // (a record in the middle of a carried-over path goes to k_shade untouched, as in k_light)
    const bool untouched = have && st != ST_IDLE && st != ST_INVALID;
    if (untouched) defer = true;
    c.rng = hdr0.x;
    c.depth = 0;
    c.flags = 0;
    c.thr = jv(1, 1, 1);
    c.acc = jv(0, 0, 0);
    c.le = jv(0, 0, 0);
    c.obj = 0;
    c.src = jv(0, 0, 0);
    c.out = jv(0, 0, 0);
    RegPx rp;
    rp.d = jv(0, 0, 1);
    rp.o = jv(0, 0, 0);
    rp.hp = jv(0, 0, 0);
    rp.h = -1;
    rp.sk = -1;
    bool finished = false;
    jvec3 color = jv(0, 0, 0), l_final = jv(0, 0, 0);
    uint32_t rng_vertex = 0;
    for (;;) {
      // ---- every lane without a ray advances until it has one (camera or mirror), is out of samples, or parks
      if (mine && st != ST_PRIMARY && st != ST_MIRROR) {
        for (;;) {
          if (finished) {
            const NextSample cs = next_sample_mh(P, rec_m, (uint32_t)home_pix, done);  // the sample that just ended
            const size_t si = (size_t)(cs.sidx % JADE_SAMPLE_LANES) * (size_t)P.npx + (size_t)cs.pixel;
            st3w(P.sum, sn, si, jv_add(ld3w(P.sum, sn, si), color));
            done += 1;
            c.c_samples += 1;
            finished = false;
            st = ST_IDLE;
          }
          if (st == ST_VERTEX) {
            if (!lean_can_shade(shade_tri(S, c.obj).m)) {  // jade / diffuse / glass: k_shade continues from here
              defer = true;
              mine = false;
              break;
            }
            rng_vertex = c.rng;  // (a mirror packet that is given up goes back to this vertex)
            if (begin_bounce_lean(S, rp, c, &l_final)) {  // the mirror ray is in rp
              n_mirror += 1;
              st = ST_MIRROR;
              break;
            }
            color = jv_add(c.le, jv_add(c.acc, jv_mul(c.thr, l_final)));
            finished = true;
            continue;
          }
          // ST_IDLE: the next sample of this record; samples of out-of-image pixels (edge tiles) are skipped
          int x, y;
          NextSample ns = next_sample_mh(P, rec_m, (uint32_t)home_pix, done);
          while (ns.sidx < target_spp && !pixel_xy_t(R, ns.pixel == home_pix ? tid0 : tile_ids[ns.pixel >> 8], ns.pixel, &x, &y)) {
            done += 1;
            ns = next_sample_mh(P, rec_m, (uint32_t)home_pix, done);
          }
          if (ns.sidx >= target_spp) {  // nothing left for this record in this step
            mine = false;
            break;
          }
          // camera ray, PathTrace.cu:1428-1437 (the statements of shade_record)
          c.rng = jade_rng_seed((uint32_t)x, (uint32_t)y, R.frame + ns.sidx);
          float fx = (float)x + jade_rand(&c.rng);
          double lo = -1.0 + R.two_over_w * ((double)fx - 0.5);
          float left_offset = (float)(lo * R.aspect);
          float fy = (float)y + jade_rand(&c.rng);
          float up_offset = (float)(-1.0 + R.two_over_h * ((double)fy - 0.5));
          jvec3 dir = jade_transform(jv(left_offset, up_offset, -1.5f), 0.0f, R.cam);
          dir = jv_normalize(dir);
          rp.set_origin(jv(P.eye[0], P.eye[1], P.eye[2]), JADE_SKIP_CAMERA);
          rp.set_dir(0, dir);
          c.c_primary += 1;
          c.depth = 0;
          c.flags = 0;
          st = ST_PRIMARY;
          break;
        }
      }
      const bool cam = mine && st == ST_PRIMARY, mir = mine && st == ST_MIRROR;
      const unsigned long long mc = __ballot(cam), mm = __ballot(mir);
      PROF_DRAIN();
      PROF_LAP(pr, PKL_ADVANCE);  // (and the fold of the packet before)
      if ((mc | mm) == 0ull) break;  // every lane is out of samples or parked
      // ---- one packet of ONE kind of ray, the kind more lanes hold (a lane with the other kind waits: a wave whose pixels
      // all see the floor alternates camera packets and mirror packets of 64 rays each)
      const bool go = __popcll(mc) >= __popcll(mm) ? cam : mir;
      bool whole = true;
      {
        const jvec3 o = rp.o, d = rp.d;
        const bool exact = !(finite_f(1.0f / d.x) && finite_f(1.0f / d.y) && finite_f(1.0f / d.z)) || !finite_f(o.x) || !finite_f(o.y) || !finite_f(o.z);
        PacketBest best;
        uint32_t pv = 0, pt = 0;  // the packet's counts: kept only if it runs to the end
        const bool general = S.general_walk || __ballot(go && exact) != 0ull;
        whole = general ? packet_trace<true>(S, stack, lane, go, o, d, rp.sk, pv, pt, best, budget, pr)
                        : packet_trace<false>(S, stack, lane, go, o, d, rp.sk, pv, pt, best, budget, pr);
        n_packets += 1;
        PROF_COUNT(pr, PKC_PACKETS, 1);
        PROF_COUNT(pr, PKC_GIVEN_UP, whole ? 0ull : 1ull);
        PROF_COUNT(pr, PKC_LANES, (unsigned long long)__popcll(__ballot(go)));
        if (whole) {
          if (go) {
            vcnt += pv + 1u;  // + the root record
            tcnt += pt;
            rp.h = (int32_t)best.index;
            rp.hp = best.point;
          }
        } else if (go) {
          // ---- given up (the rays fan out, e.g. into the statue): these lanes go back to where the ray came from and hand
          // their records to the wavefront passes, whose per-lane walk is the right tool for such rays - a camera ray back to
          // "sample not started" (its stream is seeded per sample), a mirror ray back to the vertex it left, with the random
          // state it had there.  k_shade continues from either (shade_record); nothing of the partial walk is kept.
          if (st == ST_PRIMARY) {
            c.c_primary -= 1;
            st = ST_IDLE;
          } else {
            c.rng = rng_vertex;
            c.c_shaded -= 1;
            n_mirror -= 1;
            st = ST_VERTEX;
          }
          defer = true;
          mine = false;
        }
      }
      if (!whole) {
        n_given_up += 1;
        continue;
      }
      // ---- fold the result in (shade_record's part (a))
      PROF_DRAIN();
      PROF_LAP(pr, PKL_POP);  // (what is left of the walk after its last lap)
      if (go) {
        if (st == ST_PRIMARY) {
          if (rp.h < 0) {
            color = sample_hdr(S, rp.d);  // PathTrace.cu:1443-1445
            finished = true;
            st = ST_IDLE;
          } else {
            c.le = V3(shade_tri(S, rp.h).m->emissive);
            c.thr = jv(1, 1, 1);
            c.acc = jv(0, 0, 0);
            c.depth = 0;
            c.obj = rp.h;
            c.src = rp.hp;
            c.out = jv_neg(rp.d);
            st = ST_VERTEX;
          }
        } else {  // ST_MIRROR
          c.stage = ST_MIRROR;
          const int rr = consume_mirror(S, rp, c, &l_final);
          if (rr == CONSUME_VERTEX) {
            st = ST_VERTEX;
          } else {
            color = jv_add(c.le, jv_add(c.acc, jv_mul(c.thr, l_final)));
            finished = true;
            st = ST_IDLE;
          }
        }
      }
      PROF_DRAIN();
      PROF_LAP(pr, PKL_FOLD);
    }
    // ---- store what the next kernel needs
    if (have && st != ST_INVALID && !untouched) {  // (a carried-over record was not touched)
      P.hdr[p] = make_uint4(c.rng, done, st == ST_VERTEX ? ST_VERTEX | (c.depth << 8) | (c.flags << 16) : (uint32_t)ST_IDLE, 0u);
      if (st == ST_VERTEX) {  // parked: the path context, as shade_record stores it for this stage
        float4* cx = P.ctx + (size_t)p * 4;
        cx[0] = make_float4(c.thr.x, c.thr.y, c.thr.z, __int_as_float(c.obj));
        cx[1] = make_float4(c.acc.x, c.acc.y, c.acc.z, c.out.z);
        cx[2] = make_float4(c.le.x, c.le.y, c.le.z, c.src.x);
        cx[3] = make_float4(c.src.y, c.src.z, c.out.x, c.out.y);
      }
    }
    // ---- hand-over: append to this wave's region
    {
      const unsigned long long dm = __ballot(defer);
      if (defer) my_region[n_deferred + lanes_below(dm)] = (uint32_t)p;
      n_deferred += (uint32_t)__popcll(dm);
    }
    PROF_DRAIN();
    PROF_LAP(pr, PKL_STORE);
  }
#if JADE_TRACE_PROFILE
  PROF_DRAIN();
  if (lane < PKL_N) {
    const unsigned long long v = pr.get(lane);
    if (v) atomicAdd(&g_packet_prof[lane], v);
  }
#endif
  if (lane == 0) wave_counts[wave_id] = n_deferred;
  // ---- work counters, as k_light
  {
    const uint32_t s0 = (uint32_t)wave_sum_u32(c.c_primary), s2 = (uint32_t)wave_sum_u32(c.c_shaded), s3 = (uint32_t)wave_sum_u32(c.c_samples);
    const unsigned long long s6 = wave_sum_u32(n_mirror);
    const unsigned long long sv = wave_sum_u32(vcnt), stt = wave_sum_u32(tcnt);
    if (lane == 0) {
      sh_ctr[w][0] = s0;
      sh_ctr[w][1] = 0;
      sh_ctr[w][2] = s2;
      sh_ctr[w][3] = s3;
      sh_ctr[w][4] = 0;
      sh_ctr[w][5] = 0;
      sh_ctr[w][6] = (uint32_t)s6;
      sh_ctr[w][7] = 0;
      DevCounters* cs = ctr + (blockIdx.x % JADE_CTR_SHARDS);
      if (sv) atomicAdd(&cs->nodes_visited, sv);
      if (stt) atomicAdd(&cs->tris_tested, stt);
      if (sv) atomicAdd(&cs->nodes_inline, sv);
      if (stt) atomicAdd(&cs->tris_inline, stt);
      if (s0 + s6) atomicAdd(&cs->rays_inline, (unsigned long long)s0 + s6);
      if (n_packets) atomicAdd(&cs->pad[0], (unsigned long long)n_packets);    // packets started / given up (JADE_LOG_PASSES)
      if (n_given_up) atomicAdd(&cs->pad[1], (unsigned long long)n_given_up);
    }
    __syncthreads();
    if (threadIdx.x < 8) {
      unsigned long long t = 0;
      for (int i = 0; i < JADE_TRACE_BLOCK / 64; ++i) t += sh_ctr[i][threadIdx.x];
      DevCounters* cs = ctr + (blockIdx.x % JADE_CTR_SHARDS);
      const int i = (int)threadIdx.x, wordi = i < 2 ? i : i < 4 ? i + 2 : i + 4;
      if (t) atomicAdd(reinterpret_cast<unsigned long long*>(cs) + wordi, t);
    }
  }
}

// Development / tests: packet_trace on raw rays, 64 per wave in the order given; per-ray hit, distance, point, V and T.
__global__ __launch_bounds__(JADE_TRACE_BLOCK) void k_packet_rays(DevScene S, int n, const float* origins, const float* dirs, const int32_t* skip, int32_t* hit,
                                                                 float* dist, float* point, uint32_t* v_out, uint32_t* t_out) {
  __shared__ __attribute__((aligned(16))) uint32_t lds_stack[JADE_TRACE_BLOCK / 64][8 * (JADE_PACKET_MAX_DEPTH + 1)];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool have = i < n;
  const int k = have ? i : 0;
  const jvec3 o = jv(origins[3 * k], origins[3 * k + 1], origins[3 * k + 2]), d = jv(dirs[3 * k], dirs[3 * k + 1], dirs[3 * k + 2]);
  const bool exact = !(finite_f(1.0f / d.x) && finite_f(1.0f / d.y) && finite_f(1.0f / d.z)) || !finite_f(o.x) || !finite_f(o.y) || !finite_f(o.z);
  uint32_t vcnt = have ? 1u : 0u, tcnt = 0;
  PacketBest best;
  bool whole;
  TraceProf pr;
#if JADE_TRACE_PROFILE
  __shared__ __attribute__((aligned(8))) unsigned long long lds_pprof[JADE_TRACE_BLOCK / 64][PKL_N > PL_N ? PKL_N : PL_N];
  pr.begin(lds_addr_of(reinterpret_cast<const uint32_t*>(&lds_pprof[w][0])), lane);
#endif
  if (S.general_walk || __ballot(have && exact) != 0ull) whole = packet_trace<true>(S, lds_addr_of(&lds_stack[w][0]), lane, have, o, d, skip[k], vcnt, tcnt, best, 0xffffffffu, pr);
  else whole = packet_trace<false>(S, lds_addr_of(&lds_stack[w][0]), lane, have, o, d, skip[k], vcnt, tcnt, best, 0xffffffffu, pr);
  if (have && !whole) {  // the packet was given up (no budget here: two leaves tied for some ray's best distance) - k_light_packet hands such rays to the wavefront passes
    hit[i] = -3;
    dist[i] = 0.0f;
    point[3 * i] = point[3 * i + 1] = point[3 * i + 2] = 0.0f;
    v_out[i] = t_out[i] = 0u;
  } else if (have) {
    hit[i] = (int32_t)best.index;
    dist[i] = best.dist;
    point[3 * i] = best.point.x;
    point[3 * i + 1] = best.point.y;
    point[3 * i + 2] = best.point.z;
    v_out[i] = vcnt;
    t_out[i] = tcnt;
  }
}

// Hand-over list of k_light: every wave filled a region of its own; one block turns the per-wave counts into offsets
// (and the total into qc->heavy, where k_shade reads it), then each region is copied to its place in the dense list.
__global__ __launch_bounds__(1024) void k_heavy_scan(uint32_t* wave_counts, uint32_t n_waves, QueueCtl* qc) {
  __shared__ uint32_t sh[1024];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n_waves; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < n_waves ? wave_counts[i] : 0u;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
      const uint32_t t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0u;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    const uint32_t incl = sh[threadIdx.x], c0 = carry;
    if (i < n_waves) wave_counts[n_waves + i] = c0 + incl - v;  // exclusive offsets live behind the counts
    __syncthreads();
    if (threadIdx.x == 1023) carry = c0 + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) qc->heavy = carry;
}

__global__ void k_heavy_pack(const uint32_t* heavy_regions, uint32_t region_cap, const uint32_t* wave_counts, uint32_t n_waves, uint32_t* dense) {
  const uint32_t wv = blockIdx.x;
  const uint32_t n = wave_counts[wv], off = wave_counts[n_waves + wv];
  const uint32_t* src = heavy_regions + (size_t)wv * region_cap;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dense[off + i] = src[i];
}

// ACESToneMapping + gamma + BGR pack, PathTrace.cu:680-682, 1457-1473.
__global__ void k_resolve(PathState P, RenderConst R, const int32_t* tile_ids, float inv_spp, int tonemap, float limit,
                          float* out_rgb, uint8_t* out_bgr) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;  // owned pixel
  if (p >= P.npx) return;
  int px_, py_;
  bool valid = pixel_xy(R, tile_ids, p, &px_, &py_);
  jvec3 m = jv(0, 0, 0);
  if (valid) {
    // add the JADE_SAMPLE_LANES partial sums in lane order (jade_rt.h)
    const size_t sn = (size_t)P.sum_lanes * (size_t)P.npx;
    jvec3 s = ld3w(P.sum, sn, (size_t)p);
    for (int l = 1; l < P.sum_lanes; ++l) s = jv_add(s, ld3w(P.sum, sn, (size_t)l * (size_t)P.npx + (size_t)p));
    // lanes the render never needed are not kept (PathState.sum_lanes): they would all be +0.0, and adding +0.0 any number
    // of times is adding it once (it only turns a -0.0 total into +0.0)
    if (P.sum_lanes < JADE_SAMPLE_LANES) s = jv_add(s, jv(0.0f, 0.0f, 0.0f));
    m = jv(s.x * inv_spp, s.y * inv_spp, s.z * inv_spp);
  }
  if (out_rgb) {
    out_rgb[3 * (size_t)p] = m.x;
    out_rgb[3 * (size_t)p + 1] = m.y;
    out_rgb[3 * (size_t)p + 2] = m.z;
  }
  if (out_bgr) {
    float v[3] = {m.x, m.y, m.z};
    float rein = 1.0f;
    if (tonemap == JADE_TONEMAP_REINHARD) {  // toneMapping(c, limit), PathTrace.cu:669-672 / pass3.fsh:8-18
      float luminance = (float)(0.3 * (double)m.x + 0.6 * (double)m.y + 0.1 * (double)m.z);
      rein = (float)(1.0 / (1.0 + (double)(luminance / limit)));
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float x = v[k];
      if (tonemap == JADE_TONEMAP_REINHARD) {
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
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float x = v[2 - k];
      out_bgr[3 * (size_t)p + k] = (valid && x >= 0.0f) ? (uint8_t)x : (uint8_t)0;
    }
  }
}

// ---------------------------------------------------------------- host side --

static thread_local std::string g_err;
int jade_fail(int code, const std::string& msg) {  // shared with jade_bvh.hip
  g_err = msg;
  return code;
}
static int fail(int code, const std::string& msg) { return jade_fail(code, msg); }
#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail(e_ == hipErrorOutOfMemory ? JADE_ERR_NOMEM : JADE_ERR_DEVICE,            \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                      \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t n) {
    if (p) { (void)hipFree(p); p = nullptr; }
    bytes = 0;
    const hipError_t e = hipMalloc(&p, n ? n : 16);
    if (e == hipSuccess) bytes = n;
    else p = nullptr;
    return e;
  }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct DevEvent {  // an event that is destroyed on every return path
  hipEvent_t e = nullptr;
  ~DevEvent() { if (e) (void)hipEventDestroy(e); }
  hipError_t create() { return hipEventCreate(&e); }
};

#ifndef JADE_SORT_GEOMETRY_BYTES
#define JADE_SORT_GEOMETRY_BYTES ((size_t)16 << 20) /* node + pair records above which the ray queue is ordered by default: four XCD L2s' worth */
#endif
#ifndef JADE_PACKET_GIVE_UP_LIMIT
#define JADE_PACKET_GIVE_UP_LIMIT 0.25 /* share of a step's packets given up above which the next steps of the render use the per-lane first pass */
#endif
#ifndef JADE_PACKET_BUDGET
#define JADE_PACKET_BUDGET 32 /* C3: k_light 153 / 159 / 167 / 181 ms per step at 16 / 32 / 64 / 128, and the step as a whole fastest at 32 (a lower budget hands more samples to the wavefront passes); C5: 32 / 33 / 36 ms at 16 / 32 / 64 */
#endif
// Development switches, read from the environment ONCE, at jade_scene_create (the product path reads no environment
// variable per call).  Every one of them changes the schedule only, never a result (tests/test_gpu_parity.py).
struct Tunables {
  bool shade_split = true;    // JADE_SHADE_SPLIT=0: k_shade alone over the active list from the first pass on
  bool fused = true;          // JADE_FUSED=0: the step's first pass as k_shade_lean + k_shade + k_trace instead of k_light
  bool batching = true;       // JADE_BATCH=0: the host follows every pass
  bool carry = true;          // JADE_CARRY=0: every step finishes all its paths
  double carry_frac = JADE_CARRY_FRACTION;
  bool log_passes = false;    // JADE_LOG_PASSES: one line per pass on stderr (forces one host wait per pass)
  bool pixel_rotate = false;  // JADE_PIXEL_ROTATE=1
  int records_per_pixel = 0;  // JADE_RECORDS_PER_PIXEL: test hook, results must not depend on it
  int trace_blocks_per_cu = 0;  // JADE_TRACE_BLOCKS_PER_CU: occupancy sweeps
  bool force_rccl = false;    // JADE_FORCE_RCCL=1 (tests): the RCCL path for a single share too
  bool sort_keys_kernel = false;  // JADE_SORT_KEYS_KERNEL=1: the keys of an ordered queue come from k_ray_keys (a kernel per pass) instead of from the queueing kernel
  int sort_mode = -1;         // JADE_SORT: order the ray queue by (kind, source triangle, octant) before every k_trace launch (host-followed
                              // passes): 1 always, 0 never, unset = when the traversal's records do not fit the L2 (jade_scene.sort_rays)
  uint32_t sort_min = 65536;  // JADE_SORT_MIN: queues shorter than this are traced as they are
  bool light_packet = true;   // JADE_LIGHT_PACKET=0: the fused first pass walks its rays per lane (k_light) instead of as packets
  int packet_budget = JADE_PACKET_BUDGET;  // JADE_PACKET_BUDGET: records a packet may read before it is given up and walked per lane
  int wide_mode = -1;         // JADE_WIDE: with early exits k_trace walks four grandchildren per visit (k_trace_wide): 1 always, 0 never, unset =
                              // when the traversal's records do not fit the L2 (the rule of sort_mode; jade_scene_create then builds wide records)
  bool ray_records = true;    // JADE_RAY_RECORDS=0: k_trace's refill gathers every ray through its queue entry (before round 4)
  bool shade_binned = false;  // JADE_SHADE_BINNED=1: k_shade_binned - the records of a block dealt by branch through LDS (measured level with k_shade: DESIGN.md 3.4)
  bool tail = true;           // JADE_TAIL=0: no k_tail - the last paths are finished by passes, as before round 4
  uint32_t tail_max = JADE_TAIL_MAX;  // JADE_TAIL_MAX: active records at or below which k_tail takes over
  bool anyhit = true;         // JADE_ANYHIT=0: no occluder cache (JADE_WALK_EARLY_EXIT_CACHED then walks as JADE_WALK_EARLY_EXIT)
  void read() {
    auto flag0 = [](const char* n) { const char* e = getenv(n); return e && atoi(e) == 0; };
    anyhit = !flag0("JADE_ANYHIT");
    tail = !flag0("JADE_TAIL");
    if (const char* e = getenv("JADE_TAIL_MAX")) tail_max = (uint32_t)atoi(e);
    auto flag1 = [](const char* n) { const char* e = getenv(n); return e && atoi(e) > 0; };
    shade_binned = flag1("JADE_SHADE_BINNED");
    ray_records = !flag0("JADE_RAY_RECORDS");
    shade_split = !flag0("JADE_SHADE_SPLIT");
    fused = shade_split && !flag0("JADE_FUSED");
    batching = !flag0("JADE_BATCH");
    carry = !flag0("JADE_CARRY");
    if (const char* e = getenv("JADE_CARRY_FRACTION")) carry_frac = atof(e);
    log_passes = getenv("JADE_LOG_PASSES") != nullptr;
    pixel_rotate = flag1("JADE_PIXEL_ROTATE");
    if (const char* e = getenv("JADE_RECORDS_PER_PIXEL")) records_per_pixel = atoi(e);
    if (const char* e = getenv("JADE_TRACE_BLOCKS_PER_CU")) trace_blocks_per_cu = atoi(e);
    force_rccl = getenv("JADE_FORCE_RCCL") != nullptr;
    light_packet = !flag0("JADE_LIGHT_PACKET");
    if (const char* e = getenv("JADE_SORT")) sort_mode = atoi(e) > 0 ? 1 : 0;
    if (const char* e = getenv("JADE_SORT_KEYS_KERNEL")) sort_keys_kernel = atoi(e) > 0;
    if (const char* e = getenv("JADE_SORT_MIN")) sort_min = (uint32_t)atoi(e);
    if (const char* e = getenv("JADE_PACKET_BUDGET")) packet_budget = atoi(e);
    if (const char* e = getenv("JADE_WIDE")) wide_mode = atoi(e) > 0 ? 1 : 0;
  }
};

struct jade_scene {
  int device = 0;
  Tunables tun;
  hipStream_t stream = nullptr;
  DevScene dev{};
  DevBuf b_nodes, b_nodes4, b_tverts, b_tris, b_emit, b_mapping, b_prefix, b_segs, b_env, b_guide, b_guide_obj, b_tnorm, b_mats, b_anyhit, b_env_alias;
  bool boxes_nested = true;   // every child's box lies inside its parent's (jade_scene_create): what the wide walk and the occluder cache need
  int n_emit = 0;
  int bvh_depth = 0;
  bool sort_rays = false;     // the ray queue is ordered before every k_trace launch (Tunables.sort_mode; then passes are host-followed)
  // render state
  bool have_rp = false;
  jade_render_params rp{};
  RenderConst rc{};
  PathState ps{};
  DevBuf b_sortkey, b_sortkey2, b_sortpos, b_sortq, b_sorttmp;  // JADE_SORT: keys in / out, the entries' positions, the ordered queue (of positions), rocPRIM's temporary storage
  size_t sort_cap = 0, sort_tmp_bytes = 0;
  double sort_ms = 0;
  DevBuf b_state, b_sum, b_tiles, b_queue, b_rayq, b_active[2], b_ctl, b_ctr, b_spill, b_out_rgb, b_out_bgr, b_wavecnt;
  std::vector<int32_t> tile_ids;
  int trace_blocks = 0;
  int trace_blocks_wide = 0;  // ... of k_trace_wide (fewer waves per SIMD)
  int light_blocks = 0;       // persistent grid of k_light
  int packet_blocks = 0;      // ... and of k_light_packet (0: the tree is too deep for the packet form)
  double packets_given_up = 0;  // share of the last fused pass's packets that were given up (reset by jade_render_begin)
  int64_t spp_done = 0;
  bool tail_pending = false;  // the last step left its longest paths unfinished (jade_render_flush)
  uint32_t carried_active = 0;  // ... this many records (0: unknown)
  hipEvent_t ev[7] = {};      // run_passes' timing events, made once (ev0, ev1, ta, tb, sa, sb, sm)
  hipEvent_t ev_resolve = nullptr;  // jade_render_resolve_tiles_device: caller's stream -> scene stream
  uint64_t host_syncs = 0;    // host waits inside step/flush since the last advance() reported them
  double light_ms = 0;        // k_light device time since then
  hipEvent_t ev_light[2] = {};
  hipEvent_t ev_tail[2] = {};
  double tail_ms = 0;         // k_tail device time since the last advance() reported it
  uint64_t tail_launches = 0, tail_records = 0;
  hipEvent_t ev_batch[2 * JADE_CTL_RING] = {};  // k_trace timing of a batch of passes
  ~jade_scene() {
    if (ev_resolve) (void)hipEventDestroy(ev_resolve);
    for (hipEvent_t e : ev_light)
      if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_tail)
      if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_batch)
      if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev)
      if (e) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

// Copies on `stream` and waits for it: the scene's stream is non-blocking, so a copy on the null stream would
// not be ordered before the kernels launched on it.
template <class T>
static hipError_t upload(DevBuf& b, const T* src, size_t count, hipStream_t stream) {
  hipError_t e = b.alloc(sizeof(T) * count);
  if (e != hipSuccess) return e;
  if (count) {
    e = hipMemcpyAsync(b.p, src, sizeof(T) * count, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
  }
  return e;
}

// RCCL is loaded on first use (dlopen): the single-GPU product path never needs it, and a process that also hosts
// PyTorch keeps whichever librccl it loaded first.
namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool load(std::string* why) {
    if (h) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) { *why = std::string("cannot load librccl: ") + dlerror(); return false; }
    auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) *why = std::string("librccl lacks ") + n; return p; };
    CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
    GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
    Send = (decltype(Send))sym("ncclSend");
    Recv = (decltype(Recv))sym("ncclRecv");
    GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv && GetErrorString;
  }
};
Rccl g_rccl;
}  // namespace

// Communicators are made once per device list and kept for the life of the process (ncclCommInitAll is a bootstrap of
// all ranks: tens of milliseconds on 8 GPUs - not something to pay inside every frame).
namespace {
struct CommSet {
  std::vector<int> devs;
  std::vector<ncclComm_t> comms;
};
std::mutex g_comm_mu;
std::vector<CommSet> g_comm_sets;
}  // namespace

// Gather of every share's resolved tile buffer (scenes[i]->b_out_rgb, npx * 3 floats) into `dst` on scenes[0]'s device,
// share i at off[i]: one communicator per device from this one process, all sends and receives in one group.
static int rccl_gather(jade_scene* const* scenes, int ndev, float* dst, const std::vector<size_t>& off) {
  std::string why;
  std::lock_guard<std::mutex> lock(g_comm_mu);  // one gather at a time per process: the communicators are shared
  if (!g_rccl.load(&why)) return fail(JADE_ERR_DEVICE, why);
  std::vector<int> devs(ndev);
  for (int i = 0; i < ndev; ++i) devs[i] = scenes[i]->device;
  CommSet* cs = nullptr;
  for (CommSet& c : g_comm_sets)
    if (c.devs == devs) cs = &c;
  if (!cs) {
    CommSet fresh;
    fresh.devs = devs;
    fresh.comms.assign(ndev, nullptr);
    const ncclResult_t r = g_rccl.CommInitAll(fresh.comms.data(), ndev, devs.data());
    if (r != ncclSuccess) return fail(JADE_ERR_DEVICE, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r));
    g_comm_sets.push_back(std::move(fresh));
    cs = &g_comm_sets.back();
  }
  const std::vector<ncclComm_t>& comms = cs->comms;
  // every exit below this line goes through GroupEnd: an open group would stall the next RCCL user of the process
  ncclResult_t r = g_rccl.GroupStart();
  const bool opened = r == ncclSuccess;
  const char* dev_err = nullptr;
  for (int i = 1; i < ndev && r == ncclSuccess && !dev_err; ++i) {
    const size_t count = (size_t)scenes[i]->ps.npx * 3;
    if (!count) continue;
    if (hipSetDevice(scenes[i]->device) != hipSuccess) { dev_err = "hipSetDevice failed"; break; }
    r = g_rccl.Send(scenes[i]->b_out_rgb.p, count, ncclFloat, 0, comms[i], scenes[i]->stream);
    if (r != ncclSuccess) break;
    if (hipSetDevice(scenes[0]->device) != hipSuccess) { dev_err = "hipSetDevice failed"; break; }
    r = g_rccl.Recv(dst + off[i], count, ncclFloat, i, comms[0], scenes[0]->stream);
  }
  if (opened) {
    const ncclResult_t r2 = g_rccl.GroupEnd();
    if (r == ncclSuccess) r = r2;
  }
  if (dev_err || r != ncclSuccess) {
    // a communicator that saw a failed send / receive / group may be in an error state: it is destroyed and forgotten, the next
    // gather over these devices bootstraps fresh ones
    for (size_t k = 0; k < g_comm_sets.size(); ++k)
      if (&g_comm_sets[k] == cs) {
        for (ncclComm_t c : g_comm_sets[k].comms)
          if (c) (void)g_rccl.CommDestroy(c);
        g_comm_sets.erase(g_comm_sets.begin() + (long)k);
        break;
      }
    if (dev_err) return fail(JADE_ERR_DEVICE, dev_err);
    return fail(JADE_ERR_DEVICE, std::string("RCCL gather: ") + g_rccl.GetErrorString(r));
  }
  // rank 0's own share does not travel; then every stream involved drains before the caller reads the buffer
  hipError_t e = hipSetDevice(scenes[0]->device);
  const size_t own = (size_t)scenes[0]->ps.npx * 12;
  if (e == hipSuccess && own) e = hipMemcpyAsync(dst + off[0], scenes[0]->b_out_rgb.p, own, hipMemcpyDeviceToDevice, scenes[0]->stream);
  for (int i = ndev - 1; i >= 0 && e == hipSuccess; --i) {
    e = hipSetDevice(scenes[i]->device);
    if (e == hipSuccess) e = hipStreamSynchronize(scenes[i]->stream);
  }
  if (e != hipSuccess) return fail(JADE_ERR_DEVICE, std::string("RCCL gather: ") + hipGetErrorString(e));
  return JADE_OK;
}

extern "C" {

int jade_abi_version(void) { return JADE_ABI_VERSION; }
const char* jade_backend_name(void) { return "hip-gfx950"; }
const char* jade_last_error(void) { return g_err.c_str(); }

int jade_device_count(int* n) {
  if (!n) return fail(JADE_ERR_INVALID, "null argument");
  *n = 0;
  HIP_TRY(hipGetDeviceCount(n));
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

static int validate_desc(const jade_scene_desc* d, int* depth_out) {
  if (d->abi_version != JADE_ABI_VERSION) return fail(JADE_ERR_INVALID, "abi_version mismatch");
  if (d->n_triangles <= 0 || d->n_nodes < 2 || !d->triangles || !d->nodes)
    return fail(JADE_ERR_INVALID, "scene needs triangles and a BVH (dummy node 0 + root 1)");
  if (d->n_triangles >= JADE_MAX_TRIS) return fail(JADE_ERR_UNSUPPORTED, "too many triangles for the 27-bit leaf cursor (44.7 M)");
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
  // walk the tree: ranges, cycles (visit budget), depth <= stack capacity - 1
  std::vector<std::pair<int, int>> st;
  st.push_back({1, 1});
  int64_t budget = 4 * (int64_t)d->n_nodes + 8;
  int depth = 0;
  while (!st.empty()) {
    auto [id, dp] = st.back();
    st.pop_back();
    if (--budget < 0) return fail(JADE_ERR_UNSUPPORTED, "BVH malformed (cycle)");
    if (dp > JADE_BVH_STACK_CAPACITY - 1) return fail(JADE_ERR_UNSUPPORTED, "BVH deeper than the traversal stack");
    depth = std::max(depth, dp);
    const jade_bvh_node& nd = d->nodes[id];
    if (nd.n > 0) {
      if (nd.index < 0 || (int64_t)nd.index + nd.n > d->n_triangles) return fail(JADE_ERR_INVALID, "leaf range out of bounds");
      if (nd.n > JADE_MAX_LEAF) return fail(JADE_ERR_UNSUPPORTED, "leaf with more than 15 triangles");
      continue;
    }
    if (nd.left < 0 || nd.left >= d->n_nodes || nd.right < 0 || nd.right >= d->n_nodes)
      return fail(JADE_ERR_INVALID, "child index out of range");
    if (nd.left > 0) st.push_back({nd.left, dp + 1});
    if (nd.right > 0) st.push_back({nd.right, dp + 1});
  }
  *depth_out = depth;
  return JADE_OK;
}

int jade_scene_create(const jade_scene_desc* d, int device_id, jade_scene** out) {
  if (!d || !out) return fail(JADE_ERR_INVALID, "null argument");
  int depth = 0;
  int rc = validate_desc(d, &depth);
  if (rc) return rc;
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device_id < 0 || device_id >= ndev) return fail(JADE_ERR_DEVICE, "no such HIP device");
  HIP_TRY(hipSetDevice(device_id));

  // re-lay the BVH: compact the internal nodes, children's boxes in the parent.  Internal nodes are numbered by
  // decreasing surface area of their own box (the SAH's measure of how often a node is visited), so that any prefix
  // [0, k) of the array is a connected top of the tree - a child's box is never larger than its parent's, ties keep
  // the breadth-first order - and k_trace can keep that prefix in LDS (jade_device.h, JADE_LDS_TOP_NODES).
  const int nN = d->n_nodes;
  std::vector<int32_t> compact(nN, -1);
  int n_internal = 0;
  {
    std::vector<int32_t> bfs;  // internal nodes reachable from the root, breadth first
    bfs.reserve(nN);
    if (d->nodes[1].n <= 0) bfs.push_back(1);
    for (size_t h = 0; h < bfs.size(); ++h) {
      const jade_bvh_node& nd = d->nodes[bfs[h]];
      if (nd.left > 0 && d->nodes[nd.left].n <= 0) bfs.push_back(nd.left);
      if (nd.right > 0 && d->nodes[nd.right].n <= 0) bfs.push_back(nd.right);
    }
    auto area = [&](int i) {
      const jade_bvh_node& nd = d->nodes[i];
      const double x = (double)nd.bb[0] - nd.aa[0], y = (double)nd.bb[1] - nd.aa[1], z = (double)nd.bb[2] - nd.aa[2];
      return x * y + y * z + z * x;
    };
    std::vector<double> ar(bfs.size());
    for (size_t h = 0; h < bfs.size(); ++h) ar[h] = area(bfs[h]);
    // a child inherits at most its parent's key, so a prefix of the order is always closed under "parent of"
    std::vector<double> key(nN, 0.0);
    for (size_t h = 0; h < bfs.size(); ++h) {
      const int i = bfs[h];
      if (h == 0) key[i] = ar[0];
      const jade_bvh_node& nd = d->nodes[i];
      for (int ch : {nd.left, nd.right})
        if (ch > 0 && d->nodes[ch].n <= 0) key[ch] = std::min(key[i], area(ch));
    }
    std::vector<int32_t> order(bfs.size());
    for (size_t h = 0; h < bfs.size(); ++h) order[h] = (int32_t)h;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return key[bfs[a]] > key[bfs[b]]; });
    for (int32_t h : order) compact[bfs[h]] = n_internal++;
    // internal nodes the root does not reach (none in a valid tree) keep a slot so that every record exists
    for (int i = 1; i < nN; ++i)
      if (d->nodes[i].n <= 0 && compact[i] < 0) compact[i] = n_internal++;
  }
  // vertex records hold two consecutive triangles of a leaf each (jade_trace.h): number the pairs leaf by leaf
  std::vector<uint32_t> pair_first(nN, 0);
  size_t n_pairs = 0;
  {
    std::vector<int> leaves;
    for (int i = 1; i < nN; ++i)
      if (d->nodes[i].n > 0) leaves.push_back(i);
    std::sort(leaves.begin(), leaves.end(), [&](int a, int b) { return d->nodes[a].index < d->nodes[b].index; });
    for (int i : leaves) {
      pair_first[i] = (uint32_t)n_pairs;
      n_pairs += (size_t)(d->nodes[i].n + 1) / 2;
    }
    if (n_pairs * 5 >= ((size_t)1 << 27)) return fail(JADE_ERR_UNSUPPORTED, "too many triangle pairs for the 27-bit leaf cursor");
  }
  auto ref_of = [&](int child) -> uint32_t {
    if (child <= 0) return JADE_REF_NONE;
    const jade_bvh_node& c = d->nodes[child];
    if (c.n > 0) return JADE_REF_LEAF | ((pair_first[child] * 5u) << 4) | (uint32_t)((c.n + 1) / 2);  // bits 4-30: byte offset / 16 of the first pair record
    return (uint32_t)compact[child];
  };
  std::vector<float4> nodes((size_t)4 * std::max(n_internal, 1));
  bool missing_child = false;  // the reference's "child 0" under an internal node: the walk then needs its general form
  for (int i = 1; i < nN; ++i) {
    const jade_bvh_node& nd = d->nodes[i];
    if (nd.n > 0) continue;
    if (nd.left <= 0 || nd.right <= 0) missing_child = true;
    float la[3] = {0, 0, 0}, lb[3] = {0, 0, 0}, ra[3] = {0, 0, 0}, rb[3] = {0, 0, 0};
    if (nd.left > 0) { memcpy(la, d->nodes[nd.left].aa, 12); memcpy(lb, d->nodes[nd.left].bb, 12); }
    if (nd.right > 0) { memcpy(ra, d->nodes[nd.right].aa, 12); memcpy(rb, d->nodes[nd.right].bb, 12); }
    float4* o = &nodes[(size_t)4 * compact[i]];
    o[0] = make_float4(la[0], ra[0], la[1], ra[1]);
    o[1] = make_float4(la[2], ra[2], lb[0], rb[0]);
    o[2] = make_float4(lb[1], rb[1], lb[2], rb[2]);
    uint32_t refs[4] = {ref_of(nd.left), ref_of(nd.right), 0u, 0u};
    memcpy(&o[3], refs, 16);
  }
  // the four-wide records (jade_device.h): node i's record holds its grandchildren - each child's own record, or the child twice
  // if it is a leaf
  // Built where the wide walk pays: measured with early exits, same process, k_trace per step: C5 (55 MB of node + pair records, bound
  // by dependent 64-B sector misses) 180.0 -> 166.6 ms; C3 (3.8 MB, L2-resident: a unit's own instructions count, and a wide unit
  // has more of them) 106.7 -> 110.2 ms, close-up 723 -> 755.  So: the rule of the ordered ray queue (JADE_SORT_GEOMETRY_BYTES);
  // JADE_WIDE=0 / 1 overrides it.
  std::vector<float4> nodes4;
  Tunables tun0;
  tun0.read();
  const size_t geometry_bytes0 = ((size_t)4 * std::max(n_internal, 1) + (size_t)5 * std::max<size_t>(n_pairs, 1)) * sizeof(float4);
  const bool want_wide = tun0.wide_mode < 0 ? geometry_bytes0 > JADE_SORT_GEOMETRY_BYTES : tun0.wide_mode > 0;
  // The wide walk and the occluder cache both rest on "a ray that meets a node's box meets every ancestor's" (jade_trace.h), which
  // holds when every child's box lies inside its parent's, bound by bound, with no NaN - true of any tree built by min / max over
  // the triangles (this repo's builders, the reference's), but the caller's array is the caller's: a tree with padded,
  // refitted or NaN boxes is walked with binary units from the root only, which needs no such property.
  bool nested = true;
  for (int i = 1; i < nN && nested; ++i) {
    const jade_bvh_node& nd = d->nodes[i];
    if (nd.n > 0) continue;
    for (int ch : {nd.left, nd.right}) {
      if (ch <= 0) continue;
      const jade_bvh_node& c = d->nodes[ch];
      for (int a = 0; a < 3; ++a)
        if (!(c.aa[a] >= nd.aa[a] && c.bb[a] <= nd.bb[a] && c.aa[a] <= c.bb[a])) nested = false;  // (a NaN fails every comparison)
    }
  }
  // stack levels a walk may need (validate_desc bounds the binary walk's: depth <= capacity - 1): a wide unit pushes up to three
  // entries for every two levels it descends; a walk that starts with the cached subtrees has three more under it
  const bool wide_fits = 3 * ((depth + 1) / 2) + 1 + 3 <= JADE_BVH_STACK_CAPACITY;
  const bool cache_fits = depth + 3 <= JADE_BVH_STACK_CAPACITY - 1;
  if (JADE_WIDE_WALK && want_wide && !missing_child && n_internal > 0 && nested && wide_fits) {
    nodes4.assign((size_t)8 * n_internal, make_float4(0, 0, 0, 0));
    for (int i = 1; i < nN; ++i) {
      const jade_bvh_node& nd = d->nodes[i];
      if (nd.n > 0 || compact[i] < 0) continue;
      float4* o = &nodes4[(size_t)8 * compact[i]];
      uint32_t refs[4];
      const int ch[2] = {nd.left, nd.right};
      for (int h = 0; h < 2; ++h) {
        const jade_bvh_node& c = d->nodes[ch[h]];
        if (c.n > 0) {  // a leaf: its own box in both lanes of the half, one reference
          o[3 * h + 0] = make_float4(c.aa[0], c.aa[0], c.aa[1], c.aa[1]);
          o[3 * h + 1] = make_float4(c.aa[2], c.aa[2], c.bb[0], c.bb[0]);
          o[3 * h + 2] = make_float4(c.bb[1], c.bb[1], c.bb[2], c.bb[2]);
          refs[2 * h] = ref_of(ch[h]);
          refs[2 * h + 1] = JADE_REF_NONE;
        } else {
          const float4* src = &nodes[(size_t)4 * compact[ch[h]]];
          o[3 * h + 0] = src[0];
          o[3 * h + 1] = src[1];
          o[3 * h + 2] = src[2];
          refs[2 * h] = ref_of(c.left);
          refs[2 * h + 1] = ref_of(c.right);
        }
      }
      memcpy(&o[6], refs, 16);
    }
  }
  std::vector<float4> tverts((size_t)5 * std::max<size_t>(n_pairs, 1));
  std::vector<int32_t> leaf_parent(nN, 0);
  for (int i = 1; i < nN; ++i) {
    const jade_bvh_node& nd = d->nodes[i];
    if (nd.n > 0) continue;
    if (nd.left > 0 && d->nodes[nd.left].n > 0) leaf_parent[nd.left] = i;
    if (nd.right > 0 && d->nodes[nd.right].n > 0) leaf_parent[nd.right] = i;
  }
  for (int i = 1; i < nN; ++i) {
    const jade_bvh_node& nd = d->nodes[i];
    for (int k = 0; k < nd.n; k += 2) {
      const bool has_b = k + 1 < nd.n;
      const jade_triangle& a = d->triangles[nd.index + k];
      const jade_triangle& b = d->triangles[nd.index + k + (has_b ? 1 : 0)];  // an odd leaf's last record repeats A
      float4* o = &tverts[5 * ((size_t)pair_first[i] + (size_t)k / 2)];
      o[0] = make_float4(a.p1[0], b.p1[0], a.p1[1], b.p1[1]);
      o[1] = make_float4(a.p1[2], b.p1[2], a.p2[0], b.p2[0]);
      o[2] = make_float4(a.p2[1], b.p2[1], a.p2[2], b.p2[2]);
      o[3] = make_float4(a.p3[0], b.p3[0], a.p3[1], b.p3[1]);
      // flag word: bit 0 = B is a triangle; bits 1-31 = the leaf's parent + 1 (occluder cache, jade_trace.h; 0 = the root or none: a
      // walk "from the root" is the whole walk, nothing to cache)
      const int par = leaf_parent[i];
      const uint32_t parent1 = (par > 1 && compact[par] > 0) ? (uint32_t)compact[par] + 1u : 0u;
      const uint32_t tag[2] = {(uint32_t)(nd.index + k), (has_b ? 1u : 0u) | (parent1 << 1)};
      float tagf[2];
      memcpy(tagf, tag, 8);
      o[4] = make_float4(a.p3[2], b.p3[2], tagf[0], tagf[1]);
    }
  }

  // Guide tables for the BSSRDF exit-point search (jade_shade.h, begin_bounce; PathTrace.cu:1031-1048).  Per object with
  // finite, non-decreasing prefix areas: Gn = the power of two >= 4 x its triangles cells, guide[c] = the first triangle i with
  // fl(c / Gn * A) <= prefix[i] - the product rounded once to fp32, as the kernel's `u * A` is (this file is built
  // -ffp-contract=off like the device code) - for c = 0 .. Gn, and one more entry so that cell Gn (u == 1) has an upper bound.
  // An object whose prefix areas are not monotone or not finite gets Gn = 0: the kernel then bisects as the reference does.
  std::vector<uint32_t> guide;
  std::vector<uint2> guide_obj((size_t)d->n_objects, make_uint2(0u, 0u));
  for (int o = 0; o < d->n_objects; ++o) {
    const int b = d->obj_segs[o].begin_idx, e = d->obj_segs[o].end_idx;
    bool ok = true;
    for (int i = b; i <= e && ok; ++i) {
      const float v = d->prefix_area[i];
      ok = v == v && v >= 0.0f && v < 3.0e38f && (i == b || v >= d->prefix_area[i - 1]);
    }
    const size_t nt = (size_t)(e - b + 1);
    if (!ok || nt < 2 || nt > ((size_t)1 << 21)) continue;
    uint32_t gn = 1;
    while (gn < 4 * nt) gn <<= 1;
    const float A = d->prefix_area[e];
    const size_t first = guide.size();
    guide.resize(first + gn + 2);
    int i = b;
    for (uint32_t c = 0; c <= gn; ++c) {
      const float u = (float)c / (float)gn;  // exact: both are powers of two apart
      volatile float x = u * A;              // one rounding (volatile: no excess precision, whatever the host compiler does)
      const float xv = x;
      while (i < e && !(xv <= d->prefix_area[i])) ++i;
      guide[first + c] = (uint32_t)i;
    }
    guide[first + gn + 1] = guide[first + gn];
    guide_obj[(size_t)o] = make_uint2((uint32_t)first, gn);
  }
  if (guide.empty()) guide.push_back(0u);

  // What shading reads of a triangle (jade_device.h, DevMaterial): the distinct {object, material} tuples of the caller's
  // records - the reference copies an object's material into each of its triangles, PathTrace.cu:451 - and per triangle
  // the flat normal + the number of its tuple.  Same bytes, read from 16 B + a cached table instead of a 112-B record.
  std::vector<DevMaterial> mats;
  std::vector<float4> tnorm((size_t)d->n_triangles);
  {
    std::map<std::string, uint32_t> seen;  // key: the 64 bytes of the tuple
    std::string last_key;
    uint32_t last_id = 0;
    for (int i = 0; i < d->n_triangles; ++i) {
      const jade_triangle& t = d->triangles[i];
      DevMaterial m;
      memcpy(m.emissive, t.emissive, 12);
      memcpy(m.brdf, t.brdf, 12);
      m.reflex_mode = t.reflex_mode;
      m.refract_mode = t.refract_mode;
      memcpy(m.refract_rate, t.refract_rate, 12);
      memcpy(m.refract_albedo, t.refract_albedo, 12);
      m.refract_index = t.refract_index;
      m.obj_idx = t.obj_idx;
      std::string key(reinterpret_cast<const char*>(&m), sizeof m);
      uint32_t id;
      if (i > 0 && key == last_key) {
        id = last_id;
      } else {
        auto it = seen.find(key);
        if (it == seen.end()) {
          id = (uint32_t)mats.size();
          mats.push_back(m);
          seen.emplace(key, id);
        } else {
          id = it->second;
        }
        last_key = std::move(key);
        last_id = id;
      }
      float idf;
      memcpy(&idf, &id, 4);
      tnorm[(size_t)i] = make_float4(t.norm[0], t.norm[1], t.norm[2], idf);
    }
  }

  // Environment importance sampling (jade_render_params.env_sampling; non-parity): Vose's alias table over the texels, weight =
  // (luminance + 1 % of the mean luminance) x sin(theta of the row's centre) - the floor keeps every texel drawable, so the
  // estimator stays unbiased wherever the sky is not black
  std::vector<uint4> env_alias;
  {
    const size_t W = (size_t)d->env_width, H = (size_t)d->env_height, N = W * H;
    std::vector<double> wgt(N);
    double lum_sum = 0;
    for (size_t i = 0; i < N; ++i) {
      const float* t = d->env_rgb + 3 * i;
      const double l = 0.2126 * std::max(t[0], 0.0f) + 0.7152 * std::max(t[1], 0.0f) + 0.0722 * std::max(t[2], 0.0f);
      wgt[i] = std::isfinite(l) ? l : 0.0;
      lum_sum += wgt[i];
    }
    const double floor_l = lum_sum > 0 ? 0.01 * lum_sum / (double)N : 1.0;
    double total = 0;
    for (size_t j = 0; j < H; ++j) {
      const double st = std::sin(JADE_PI_D * ((double)j + 0.5) / (double)H);
      for (size_t i = 0; i < W; ++i) {
        wgt[j * W + i] = (wgt[j * W + i] + floor_l) * st;
        total += wgt[j * W + i];
      }
    }
    std::vector<double> q(N);
    std::vector<uint32_t> small, large, alias(N);
    std::vector<float> accept(N, 1.0f);
    for (size_t i = 0; i < N; ++i) {
      q[i] = wgt[i] / total * (double)N;  // the texel's probability x N (mean 1)
      alias[i] = (uint32_t)i;
      (q[i] < 1.0 ? small : large).push_back((uint32_t)i);
    }
    std::vector<double> r = q;
    while (!small.empty() && !large.empty()) {
      const uint32_t a = small.back(), g = large.back();
      small.pop_back();
      accept[a] = (float)r[a];
      alias[a] = g;
      r[g] = (r[g] + r[a]) - 1.0;
      if (r[g] < 1.0) { large.pop_back(); small.push_back(g); }
    }
    env_alias.resize(N);
    for (size_t i = 0; i < N; ++i) {
      const float ps = (float)q[i], pa = (float)q[alias[i]];
      uint4 e;
      memcpy(&e.x, &accept[i], 4);
      e.y = alias[i];
      memcpy(&e.z, &ps, 4);
      memcpy(&e.w, &pa, 4);
      env_alias[i] = e;
    }
  }

  jade_scene* s = new (std::nothrow) jade_scene();
  if (!s) return fail(JADE_ERR_NOMEM, "out of memory");
  s->device = device_id;
  s->tun.read();
  s->n_emit = d->n_emit;
  s->bvh_depth = depth;
  hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = upload(s->b_nodes, nodes.data(), nodes.size(), s->stream);
  if (e == hipSuccess && !nodes4.empty()) e = upload(s->b_nodes4, nodes4.data(), nodes4.size(), s->stream);
  if (e == hipSuccess) e = upload(s->b_tverts, tverts.data(), tverts.size(), s->stream);
  if (e == hipSuccess) e = upload(s->b_tris, d->triangles, (size_t)d->n_triangles, s->stream);
  if (e == hipSuccess) e = upload(s->b_emit, d->emit_indices, (size_t)d->n_emit, s->stream);
  if (e == hipSuccess) e = upload(s->b_mapping, d->index_mapping, (size_t)d->n_triangles, s->stream);
  if (e == hipSuccess) e = upload(s->b_prefix, d->prefix_area, (size_t)d->n_triangles, s->stream);
  if (e == hipSuccess) e = upload(s->b_segs, d->obj_segs, (size_t)d->n_objects, s->stream);
  if (e == hipSuccess) e = upload(s->b_env, d->env_rgb, (size_t)3 * d->env_width * d->env_height, s->stream);
  if (e == hipSuccess) e = upload(s->b_env_alias, env_alias.data(), env_alias.size(), s->stream);
  if (e == hipSuccess) e = upload(s->b_tnorm, tnorm.data(), tnorm.size(), s->stream);
  if (e == hipSuccess) e = upload(s->b_mats, mats.data(), mats.size(), s->stream);
  if (e == hipSuccess) e = upload(s->b_guide, guide.data(), guide.size(), s->stream);
  if (e == hipSuccess) e = upload(s->b_guide_obj, guide_obj.data(), guide_obj.size(), s->stream);
  const bool want_anyhit = s->tun.anyhit && nested && cache_fits && !missing_child && n_internal > 1;
  if (e == hipSuccess && want_anyhit) {
    e = s->b_anyhit.alloc((size_t)d->n_triangles * JADE_ANYHIT_KEYS * sizeof(uint4));
    if (e == hipSuccess) e = hipMemsetAsync(s->b_anyhit.p, 0, s->b_anyhit.bytes, s->stream);
  }
  if (e == hipSuccess) e = s->b_ctl.alloc(sizeof(QueueCtl) * (JADE_CTL_RING + 1));  // (+ 1: k_arm's count when the host does not wait for it)
  if (e == hipSuccess) e = s->b_ctr.alloc(sizeof(DevCounters) * JADE_CTR_SHARDS);
  if (e != hipSuccess) {
    delete s;
    return fail(e == hipErrorOutOfMemory ? JADE_ERR_NOMEM : JADE_ERR_DEVICE, std::string("scene upload: ") + hipGetErrorString(e));
  }
  s->dev.nodes = s->b_nodes.as<float4>();
  s->dev.nodes4 = nodes4.empty() ? nullptr : s->b_nodes4.as<float4>();
  s->dev.tverts = s->b_tverts.as<float4>();
  s->dev.tris = s->b_tris.as<jade_triangle>();
  s->dev.emit = s->b_emit.as<int32_t>();
  s->dev.mapping = s->b_mapping.as<int32_t>();
  s->dev.prefix = s->b_prefix.as<float>();
  s->dev.segs = s->b_segs.as<jade_obj_seg>();
  s->dev.env = s->b_env.as<float>();
  s->dev.tnorm = s->b_tnorm.as<float4>();
  s->dev.mats = s->b_mats.as<DevMaterial>();
  s->dev.guide = s->b_guide.as<uint32_t>();
  s->dev.guide_obj = s->b_guide_obj.as<uint2>();
  s->dev.env_w = d->env_width;
  s->dev.env_h = d->env_height;
  s->dev.n_tris = d->n_triangles;
  s->dev.n_emit = d->n_emit;
  s->dev.root_ref = ref_of(1);
  s->dev.top_k = (uint32_t)std::min(n_internal, (int)JADE_LDS_TOP_NODES);
  s->dev.general_walk = missing_child ? 1u : 0u;
  s->dev.anyhit = want_anyhit ? s->b_anyhit.as<uint4>() : nullptr;
  s->dev.env_alias = s->b_env_alias.as<uint4>();
  s->boxes_nested = nested;
  // Ray ordering pays when the traversal's records do not fit the XCDs' L2s (C5: 55 MB, k_trace bound by the rate of 64-B sector
  // misses: 4 235 -> 5 275 Mray/s); on a tree that does (C3: 3.8 MB) it costs more than it gives
  {
    const size_t geometry_bytes = nodes.size() * sizeof(float4) + tverts.size() * sizeof(float4);
    s->sort_rays = s->tun.sort_mode < 0 ? geometry_bytes > JADE_SORT_GEOMETRY_BYTES : s->tun.sort_mode > 0;
  }

  // the arithmetic contract of jade_fpmath.h, checked on the device once
  hipLaunchKernelGGL(k_selftest, dim3(1), dim3(1), 0, s->stream, s->b_ctl.as<QueueCtl>(), 1.0f);
  QueueCtl qc{};
  e = hipMemcpyAsync(&qc, s->b_ctl.p, sizeof qc, hipMemcpyDeviceToHost, s->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
  if (e != hipSuccess || qc.fp_bad) {
    std::string m = e != hipSuccess ? std::string("selftest: ") + hipGetErrorString(e)
                                    : "device code was built with FP contraction on (see include/jade_fpmath.h)";
    delete s;
    return fail(JADE_ERR_DEVICE, m);
  }
  // persistent trace grid: as many blocks per CU as registers and the LDS columns (20 KB/block) allow
  hipDeviceProp_t prop;
  if (hipError_t pe = hipGetDeviceProperties(&prop, device_id); pe != hipSuccess) {
    delete s;
    return fail(JADE_ERR_DEVICE, std::string("hipGetDeviceProperties: ") + hipGetErrorString(pe));
  }
  int per_cu = 0;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace, JADE_TRACE_BLOCK, 0);
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 8) per_cu = 8;
  if (s->tun.trace_blocks_per_cu >= 1 && s->tun.trace_blocks_per_cu < per_cu) per_cu = s->tun.trace_blocks_per_cu;  // development: occupancy sweeps
  if (s->tun.log_passes) fprintf(stderr, "[jade] k_trace: %d blocks of %d threads per CU, %d CUs\n", per_cu, JADE_TRACE_BLOCK, prop.multiProcessorCount);
  s->trace_blocks = prop.multiProcessorCount * per_cu;
  {  // k_trace_wide keeps fewer waves: its own grid (the stack spill area is sized for the larger one)
    int wide_cu = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&wide_cu, k_trace_wide, JADE_TRACE_BLOCK, 0);
    if (wide_cu < 1) wide_cu = 1;
    if (wide_cu > per_cu) wide_cu = per_cu;
    s->trace_blocks_wide = prop.multiProcessorCount * wide_cu;
  }
  int light_cu = 0;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&light_cu, k_light, JADE_TRACE_BLOCK, 0);
  if (light_cu < 1) light_cu = 1;
  if (light_cu > per_cu) light_cu = per_cu;  // the stack spill area is sized for the k_trace grid
  s->light_blocks = prop.multiProcessorCount * light_cu;
  if (depth <= JADE_PACKET_MAX_DEPTH) {
    int pk_cu = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&pk_cu, k_light_packet, JADE_TRACE_BLOCK, 0);
    if (pk_cu < 1) pk_cu = 1;
    if (pk_cu > 8) pk_cu = 8;
    s->packet_blocks = prop.multiProcessorCount * pk_cu;
  }
  if (s->tun.log_passes) fprintf(stderr, "[jade] first pass: k_light %d blocks, k_light_packet %d blocks (tree depth %d)\n", s->light_blocks, s->packet_blocks, depth);
  *out = s;
  return JADE_OK;
}

void jade_scene_destroy(jade_scene* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  delete s;
}

static int setup_state(jade_scene* s, int npx, int rpp, int nslots, int sum_lanes) {
  const int npix = npx * rpp;
  // carve every per-pixel array out of one allocation
  size_t words = 0;
  auto take = [&](size_t n) { size_t o = words; words += (n + 63) & ~(size_t)63; return o; };
  const size_t N = (size_t)npix, K = (size_t)nslots;
  size_t o_hdr = take(4 * N), o_ctx = take(16 * N), o_aux = take(4 * N), o_orgs = take(4 * N), o_slot = take(4 * K * N), o_hitp = take(4 * N);
  HIP_TRY(s->b_state.alloc(words * 4));
  HIP_TRY(hipMemsetAsync(s->b_state.p, 0, words * 4, s->stream));
  // the partial sums are an allocation of their own: they grow if steps add more samples than were announced (grow_sums)
  HIP_TRY(s->b_sum.alloc((size_t)3 * sum_lanes * npx * 4));
  HIP_TRY(hipMemsetAsync(s->b_sum.p, 0, (size_t)3 * sum_lanes * npx * 4, s->stream));
  uint32_t* b = s->b_state.as<uint32_t>();
  PathState& P = s->ps;
  P.npix = npix;
  P.npx = npx;
  P.rpp = rpp;
  {
    // Pixel rotation is OFF by default: measured on C3 it cuts the passes per 256-sample block
    // from 172 to 142 but de-synchronised records lose coalescing and ray coherence in the
    // heavy passes (2155 vs 2229 Mray/s).  JADE_PIXEL_ROTATE=1 turns it on for experiments.
    const int per = JADE_SAMPLE_LANES / rpp;
    P.stride = (s->tun.pixel_rotate && per > 1) ? (npx / per) | 1 : 0;
  }
  P.nslots = nslots;
  P.hdr = (uint4*)(b + o_hdr);
  P.sum_lanes = sum_lanes;
  P.sum = s->b_sum.as<float>(); P.ctx = (float4*)(b + o_ctx); P.aux = (float4*)(b + o_aux); P.orgs = (float4*)(b + o_orgs);
  P.slot = (float4*)(b + o_slot);
  P.hitp = (float4*)(b + o_hitp);
  P.write_all_hits = 0u;
  HIP_TRY(s->b_queue.alloc(K * N * 4));
  // ray records for the head of the queue (PathState.rayq): as many as an eighth of all slots - a pass after the fused first one
  // queues rays for a few per cent of the records - but every slot of a small render; none when switched off
  {
    size_t cap = s->tun.ray_records ? std::max<size_t>((K * N + 7) / 8, std::min<size_t>(K * N, (size_t)1 << 22)) : 0;
    cap = std::min<size_t>(cap, K * N);
    if (cap) HIP_TRY(s->b_rayq.alloc(cap * 48));
    P.rayq = cap ? s->b_rayq.as<float4>() : nullptr;
    P.rayq_cap = (uint32_t)std::min<size_t>(cap, 0xffffffffu);
  }
  // b_active[0] doubles as k_light's per-wave hand-over regions: up to 64 records of slack per wave of its grid
  const size_t first_pass_blocks = (size_t)std::max(s->light_blocks, s->packet_blocks);
  HIP_TRY(s->b_active[0].alloc((N + first_pass_blocks * JADE_TRACE_BLOCK + 64) * 4));
  HIP_TRY(s->b_active[1].alloc(N * 4));
  HIP_TRY(s->b_wavecnt.alloc((size_t)2 * first_pass_blocks * (JADE_TRACE_BLOCK / 64) * 4));
  s->sort_cap = 0;
  P.keyq = nullptr;
  P.keyq_cap = 0;
  P.key_tri_bits = 1;
  if (s->sort_rays) {
    const size_t cap = std::min<size_t>(K * N, (size_t)1 << 28);
    size_t tmp = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, cap, 0u, 32u, s->stream));
    HIP_TRY(s->b_sortkey.alloc(cap * 4));
    HIP_TRY(s->b_sortkey2.alloc(cap * 4));
    HIP_TRY(s->b_sortpos.alloc(cap * 4));
    HIP_TRY(s->b_sortq.alloc(cap * 4));
    HIP_TRY(s->b_sorttmp.alloc(tmp));
    s->sort_tmp_bytes = tmp;
    s->sort_cap = cap;
    P.keyq = s->tun.sort_keys_kernel ? nullptr : s->b_sortkey.as<uint32_t>();
    P.keyq_cap = s->tun.sort_keys_kernel ? 0u : (uint32_t)std::min<size_t>(cap, 0xffffffffu);
    while (P.key_tri_bits < 23 && ((uint32_t)(s->dev.n_tris - 1) >> P.key_tri_bits)) ++P.key_tri_bits;
    // the queueing kernel writes the keys itself (PathState.keyq); the positions the sort moves with them are 0 .. cap-1, made once
    hipLaunchKernelGGL(k_iota, dim3(1024), dim3(256), 0, s->stream, s->b_sortpos.as<uint32_t>(), (uint32_t)std::min<size_t>(cap, 0xffffffffu));
    HIP_TRY(hipGetLastError());
  }
  if (!s->b_spill.p)
    HIP_TRY(s->b_spill.alloc((size_t)(JADE_BVH_STACK_CAPACITY - JADE_LDS_STACK) * s->trace_blocks * JADE_TRACE_BLOCK * 4));
  return JADE_OK;
}

int jade_render_begin(jade_scene* s, const jade_render_params* rp) {
  if (!s || !rp) return fail(JADE_ERR_INVALID, "null argument");
  if (rp->width <= 0 || rp->height <= 0 || rp->tile_nranks <= 0 || rp->tile_rank < 0 || rp->tile_rank >= rp->tile_nranks)
    return fail(JADE_ERR_INVALID, "bad image size or tile partition");
  if (rp->walk != JADE_WALK_REFERENCE && rp->walk != JADE_WALK_EARLY_EXIT && rp->walk != JADE_WALK_EARLY_EXIT_CACHED)
    return fail(JADE_ERR_INVALID, "unknown walk (JADE_WALK_*)");
  if (rp->env_sampling != JADE_ENV_REFERENCE && rp->env_sampling != JADE_ENV_IMPORTANCE) return fail(JADE_ERR_INVALID, "unknown env_sampling (JADE_ENV_*)");
  HIP_TRY(hipSetDevice(s->device));
  const int tx = (rp->width + JADE_TILE_SIZE - 1) / JADE_TILE_SIZE, ty = (rp->height + JADE_TILE_SIZE - 1) / JADE_TILE_SIZE;
  s->tile_ids.clear();
  for (int y = 0; y < ty; ++y)
    for (int x = 0; x < tx; ++x)
      if ((x + y) % rp->tile_nranks == rp->tile_rank) s->tile_ids.push_back(y * tx + x);
  const int nslots = s->n_emit + 2;
  const int64_t npx64 = (int64_t)s->tile_ids.size() * 256;
  // Records per pixel: as many paths in flight as JADE_RECORD_MEMORY of the free device memory holds (the
  // partial sums come out of the same share), whatever the image share of this GPU: more records = fewer,
  // wider passes.  288 GB is what makes 530 M paths (112 GB) for a full 1080p frame affordable.
  const double bytes_per_record = 4.0 * (34 + 5 * nslots) + (s->tun.ray_records ? 6.0 * nslots : 0.0);  // PathState (a float4 per slot, one hit point per record) + queue entry + two list entries + 48-B ray records for an eighth of the slots
  // partial sums per pixel: one lane per sample up to JADE_SAMPLE_LANES, never more than the render was announced with
  // (rounded up to a power of two): a 1-spp 4K frame is 100 MB of sums, not 102 GB
  int sum_lanes = JADE_SAMPLE_LANES;
  if (rp->spp > 0)
    for (sum_lanes = 1; sum_lanes < rp->spp && sum_lanes < JADE_SAMPLE_LANES;) sum_lanes <<= 1;
  const double sums_bytes = 12.0 * sum_lanes * (double)npx64;
  size_t mem_free = 0, mem_total = 0;
  HIP_TRY(hipMemGetInfo(&mem_free, &mem_total));
  mem_free += s->b_state.bytes + s->b_sum.bytes + s->b_queue.bytes + s->b_rayq.bytes + s->b_active[0].bytes + s->b_active[1].bytes;  // ours to reuse
  // the caller's bound (jade_render_params.max_state_bytes) replaces the default share of the free memory
  double state_budget = JADE_RECORD_MEMORY * (double)mem_free;
  if (rp->max_state_bytes) state_budget = std::min((double)rp->max_state_bytes, 0.95 * (double)mem_free);
  const double budget = state_budget - sums_bytes;
  // never more records per pixel than samples this render was announced with (rounded up to a power of two):
  // a 64-spp render of a small image must not claim, clear and scan gigabytes of records that never get a sample
  int rpp = JADE_SAMPLE_LANES;
  if (rp->spp > 0)
    for (rpp = 1; rpp < rp->spp && rpp < JADE_SAMPLE_LANES;) rpp <<= 1;
  // (3 * npix must fit an int: the three planes of a per-record vec3 are indexed with int arithmetic)
  while (rpp > 1 && ((double)npx64 * rpp * bytes_per_record > budget || npx64 * rpp * nslots >= ((int64_t)1 << 32) ||
                     npx64 * rpp * 3 >= ((int64_t)1 << 31)))
    rpp >>= 1;
  if (const int v = s->tun.records_per_pixel; v >= 1 && v <= JADE_SAMPLE_LANES && (v & (v - 1)) == 0) {  // test hook: results must not depend on it
    rpp = v;
    if (sum_lanes < rpp) sum_lanes = rpp;
  }
  const int64_t npix64 = npx64 * rpp;
  if (npix64 * nslots >= ((int64_t)1 << 32) || npix64 * 3 >= ((int64_t)1 << 31))
    return fail(JADE_ERR_UNSUPPORTED, "pixels x records x (emitters + 2) exceeds the 32-bit ray-slot index");
  if (sums_bytes + (double)npix64 * bytes_per_record > (rp->max_state_bytes ? std::max(state_budget, 0.0) : 0.95 * (double)mem_free))
    return fail(JADE_ERR_NOMEM, rp->max_state_bytes ? "frame does not fit max_state_bytes (partial sums + one record per pixel)"
                                                    : "frame does not fit the device memory (partial sums + one record per pixel)");
  if (s->tun.log_passes)
    fprintf(stderr, "[jade] %lld pixels x %d records, %.1f GB of path state + %.1f GB of partial sums in %d lanes (%.0f GB free)\n", (long long)npx64,
            rpp, npix64 * bytes_per_record / 1e9, 12.0 * sum_lanes * (double)npx64 / 1e9, sum_lanes, mem_free / 1e9);
  s->have_rp = false;
  s->rp = *rp;
  RenderConst& R = s->rc;
  R.width = rp->width; R.height = rp->height; R.tiles_x = tx; R.frame = rp->frame;
  memcpy(R.eye, rp->eye, sizeof R.eye);
  memcpy(R.cam, rp->camera, sizeof R.cam);
  R.two_over_w = 2.0 / (double)rp->width;
  R.two_over_h = 2.0 / (double)rp->height;
  R.aspect = (double)rp->width / (double)rp->height;
  s->spp_done = 0;
  s->tail_pending = false;
  s->carried_active = 0;
  s->packets_given_up = 0;
  if (npix64 == 0) { s->ps.npix = 0; s->ps.npx = 0; s->have_rp = true; return JADE_OK; }
  int rc = setup_state(s, (int)npx64, rpp, nslots, sum_lanes);
  if (rc) return rc;
  s->ps.early_exit = rp->walk == JADE_WALK_EARLY_EXIT_CACHED ? 2u : rp->walk == JADE_WALK_EARLY_EXIT ? 1u : 0u;
  s->ps.env_sampling = rp->env_sampling == JADE_ENV_IMPORTANCE ? 1u : 0u;
  memcpy(s->ps.eye, rp->eye, sizeof s->ps.eye);
  HIP_TRY(upload(s->b_tiles, s->tile_ids.data(), s->tile_ids.size(), s->stream));
  HIP_TRY(hipMemsetAsync(s->b_ctr.p, 0, sizeof(DevCounters) * JADE_CTR_SHARDS, s->stream));
  hipLaunchKernelGGL(k_init, dim3((unsigned)((npix64 + 255) / 256)), dim3(256), 0, s->stream, s->ps);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->have_rp = true;
  return JADE_OK;
}

// Rays claimed per queue atomic: large launches amortise the atomic over up to
// JADE_TRACE_CHUNK rays, small ones keep 64 so every wave gets work.
static uint32_t trace_chunk(const jade_scene* s, uint32_t n_rays) {
  uint64_t waves = (uint64_t)s->trace_blocks * (JADE_TRACE_BLOCK / 64);
  uint64_t per = n_rays / (waves * 64 * 8);  // aim at >= 8 grabs per wave
  if (per < 1) per = 1;
  if (per > JADE_TRACE_CHUNK / 64) per = JADE_TRACE_CHUNK / 64;
  return (uint32_t)per * 64u;
}

static hipError_t sum_counters(jade_scene* s, DevCounters* out) {
  std::vector<DevCounters> sh(JADE_CTR_SHARDS);
  hipError_t e = hipMemcpyAsync(sh.data(), s->b_ctr.p, sizeof(DevCounters) * JADE_CTR_SHARDS, hipMemcpyDeviceToHost, s->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
  memset(out, 0, sizeof *out);
  for (const DevCounters& c : sh) {
    out->rays_primary += c.rays_primary; out->rays_shadow += c.rays_shadow;
    out->nodes_visited += c.nodes_visited; out->tris_tested += c.tris_tested;
    out->shaded_hits += c.shaded_hits; out->samples += c.samples;
    out->rays_env += c.rays_env; out->rays_indirect += c.rays_indirect;
    out->rays_mirror += c.rays_mirror; out->rays_refract += c.rays_refract;
    out->rays_inline += c.rays_inline;
    out->nodes_inline += c.nodes_inline; out->tris_inline += c.tris_inline;
    out->rays_cached += c.rays_cached;
    out->rays_tail += c.rays_tail; out->nodes_tail += c.nodes_tail; out->tris_tail += c.tris_tail;
    out->pad[0] += c.pad[0]; out->pad[1] += c.pad[1];
  }
  return e;
}

// shade/trace passes until a shade pass emits no ray (or the step may carry the rest over).  The host follows the first
// passes of a step one by one (it picks the schedule from the counts) and the list-mode passes in batches of 8-16.
static int run_passes(jade_scene* s, int64_t from_spp, uint32_t target_spp, bool may_carry, double* ms_out, double* trace_ms_out, uint64_t* launches_out) {
  const int npix = s->ps.npix;
  QueueCtl* qc = s->b_ctl.as<QueueCtl>();
  for (hipEvent_t& e : s->ev)
    if (!e) HIP_TRY(hipEventCreate(&e));
  const hipEvent_t ev0 = s->ev[0], ev1 = s->ev[1], ta = s->ev[2], tb = s->ev[3], sa = s->ev[4], sb = s->ev[5], sm = s->ev[6];
  HIP_TRY(hipEventRecord(ev0, s->stream));
  bool trace_pending = false, carried = false;
  double trace_ms = 0;
  uint64_t launches = 0;
  s->tail_pending = false;  // whatever an earlier step left is part of this call's work
  // While at least a quarter of the records are active, a pass is k_shade_lean over all records
  // (record order, no list) followed by k_shade over what it handed over; below that, k_shade alone
  // over the active list, which k_arm rebuilds once at the switch.  JADE_SHADE_SPLIT=0: always the list.
  const bool split_ok = s->tun.shade_split;
  // JADE_FUSED=0: the first pass as shade / trace passes too (k_shade_lean), the schedule before k_light existed
  const bool fused = s->tun.fused;
  // the records with work in this step: all of them when the step gives every record a sample (k_light then walks the
  // records itself); otherwise - a flush, a step of fewer samples than records per pixel - k_arm lists and counts them
  uint32_t host_ctl[3] = {0, 0, 0};
  uint32_t n_active;
  const uint32_t* arm_dev = nullptr;  // k_arm's count, on the device, when the host has not waited for it
  if (fused && s->ps.stride == 0 && (int64_t)target_spp - from_spp >= (int64_t)s->ps.rpp) {
    n_active = (uint32_t)npix;
  } else if ((int64_t)target_spp == from_spp && s->carried_active > 0 && s->tun.batching && !s->sort_rays && !s->tun.log_passes && s->ps.stride == 0) {
    // A flush: no sample is started, so the records with work are exactly the ones the last step carried over, and the host knows
    // how many those were - k_arm lists them, the first pass of the batch below reads the count on the device, nobody waits.
    QueueCtl* qa = qc + JADE_CTL_RING;
    HIP_TRY(hipMemsetAsync(qa, 0, 12, s->stream));
    hipLaunchKernelGGL(k_arm, dim3((unsigned)(((size_t)npix + JADE_ARM_BLOCK * JADE_ARM_PER_THREAD - 1) / (JADE_ARM_BLOCK * JADE_ARM_PER_THREAD))), dim3(JADE_ARM_BLOCK), 0, s->stream, s->ps, target_spp,
                       s->b_active[0].as<uint32_t>(), qa);
    n_active = s->carried_active;  // (an upper bound is all the grids need)
    arm_dev = &qa->active;
  } else {
    HIP_TRY(hipMemsetAsync(qc, 0, 12, s->stream));
    hipLaunchKernelGGL(k_arm, dim3((unsigned)(((size_t)npix + JADE_ARM_BLOCK * JADE_ARM_PER_THREAD - 1) / (JADE_ARM_BLOCK * JADE_ARM_PER_THREAD))), dim3(JADE_ARM_BLOCK), 0, s->stream, s->ps, target_spp,
                       s->b_active[0].as<uint32_t>(), qc);
    HIP_TRY(hipMemcpyAsync(host_ctl, qc, 12, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->host_syncs += 1;
    n_active = host_ctl[1];
  }
  s->carried_active = 0;
  const uint32_t n_armed = n_active;  // records with work at the start of this call
  int cur = 0, pass_no = 0;
  const bool log_passes = s->tun.log_passes;
  bool have_list = true;  // b_active[cur] lists the active records
  bool light_timed = false;
  float shade_ms = 0, lean_ms = 0;
  const double carry_frac = s->tun.carry_frac;
  auto carry_now = [&](uint32_t act) {
    return may_carry && ((act < JADE_CARRY_RECORDS && (uint64_t)act * 1024 < (uint64_t)n_armed) ||
                         (carry_frac > 0 && (double)act < carry_frac * (double)n_armed));
  };
  // JADE_BATCH=0: the host follows every pass (the schedule before batching existed)
  const bool batching = s->tun.batching && !s->sort_rays;  // (rocPRIM wants the queue's length on the host)
  bool closed_by_batch = false;  // the wait at the end of a batch was also the wait for the end of the step
  // k_tail: once the active list is short - and is not about to be carried over - ONE launch finishes its records (every wave
  // shades and traces its own 64 until they are out of samples).  The records' last rays have been traced: k_tail starts by shading.
  // k_shade with the records dealt by branch through LDS (k_shade_binned), unless switched off - or the render draws its
  // environment rays by importance: a bounce may then emit no ray at all and is folded in on the spot, which the binned form does not do
  auto shade_kernel = s->ps.env_sampling ? k_shade_envis : s->tun.shade_binned ? k_shade_binned : k_shade;
  const bool tail_ok = s->tun.tail && s->tun.tail_max > 0 && !s->ps.env_sampling;  // (k_tail shades with the parity code only)
  const uint32_t tail_max = std::min<uint32_t>(s->tun.tail_max, (uint32_t)(s->b_queue.bytes / 4 / (size_t)std::max(s->ps.nslots, 1)));
  while (n_active) {
    const bool lean_mode = split_ok && (uint64_t)n_active * 4 >= (uint64_t)npix;
    if (tail_ok && !lean_mode && have_list && pass_no > 0 && !arm_dev && n_active <= tail_max && !carry_now(n_active)) {
      TailArgs ta;
      ta.R = s->rc;
      ta.tile_ids = s->b_tiles.as<int32_t>();
      ta.target_spp = target_spp;
      ta.list = s->b_active[cur].as<uint32_t>();
      ta.n_list = n_active;
      ta.queue = s->b_queue.as<uint32_t>();
      for (hipEvent_t& e : s->ev_tail)
        if (!e) HIP_TRY(hipEventCreate(&e));
      HIP_TRY(hipEventRecord(s->ev_tail[0], s->stream));
      hipLaunchKernelGGL(k_tail, dim3((n_active + JADE_TRACE_BLOCK - 1) / JADE_TRACE_BLOCK), dim3(JADE_TRACE_BLOCK), 0, s->stream, s->dev, s->ps, ta,
                         s->b_spill.as<uint32_t>(), s->b_ctr.as<DevCounters>());
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipEventRecord(s->ev_tail[1], s->stream));
      HIP_TRY(hipEventRecord(ev1, s->stream));
      HIP_TRY(hipStreamSynchronize(s->stream));
      s->host_syncs += 1;
      float tt = 0;
      HIP_TRY(hipEventElapsedTime(&tt, s->ev_tail[0], s->ev_tail[1]));
      s->tail_ms += tt;
      s->tail_launches += 1;
      s->tail_records += n_active;
      if (log_passes) fprintf(stderr, "[jade] tail: %u records finished by k_tail in %.3f ms\n", n_active, tt);
      n_active = 0;
      closed_by_batch = true;  // (ev1 is recorded and waited for)
      break;
    }
    if (!lean_mode && have_list && batching && !log_passes && (pass_no > 0 || arm_dev)) {
      // ---- a BATCH of list-mode passes without the host in between: pass j's k_shade takes its length from the record
      // count pass j-1 left on the device (QueueCtl ring), k_trace sizes its chunks itself; a pass that finds nothing
      // to do is three empty launches.  The host looks once per batch: where the paths ended, whether to carry.
      const int B = JADE_CTL_RING;
      const unsigned nbb = (n_active + JADE_SHADE_BLOCK - 1) / JADE_SHADE_BLOCK;  // the active count only shrinks: an upper bound for all
      // the carry-over point, for the device: passes of the batch behind it do nothing (k_shade)
      uint32_t stop_below = 0;
      if (may_carry) {
        const uint64_t a = std::min<uint64_t>(JADE_CARRY_RECORDS, ((uint64_t)n_armed + 1023) / 1024);  // act < CARRY_RECORDS && act * 1024 < n_armed
        const uint64_t b = carry_frac > 0 ? (uint64_t)std::ceil(carry_frac * (double)n_armed) : 0;     // act < carry_frac * n_armed
        stop_below = (uint32_t)std::min<uint64_t>(std::max(a, b), 0xffffffffu);
      }
      // ... and the point below which k_tail finishes the list: the batch stops there too, the host then launches it (above)
      if (tail_ok) stop_below = std::max(stop_below, tail_max + 1u);
      for (hipEvent_t& e : s->ev_batch)
        if (!e) HIP_TRY(hipEventCreate(&e));
      HIP_TRY(hipMemsetAsync(qc, 0, sizeof(QueueCtl) * B, s->stream));
      const int cur0 = cur;
      for (int j = 0; j < B; ++j) {
        hipLaunchKernelGGL(shade_kernel, dim3(nbb), dim3(JADE_SHADE_BLOCK), 0, s->stream, s->dev, s->ps, s->rc, s->b_tiles.as<int32_t>(),
                           target_spp, s->b_active[cur].as<uint32_t>(), n_active, j ? &qc[j - 1].active : arm_dev,
                           s->b_active[cur ^ 1].as<uint32_t>(), s->b_queue.as<uint32_t>(), qc + j, s->b_ctr.as<DevCounters>(),
                           j ? qc + (j - 1) : (const QueueCtl*)nullptr, stop_below);
        cur ^= 1;
        HIP_TRY(hipEventRecord(s->ev_batch[2 * j], s->stream));
        hipLaunchKernelGGL(trace_wide(s->dev, s->ps) ? k_trace_wide : k_trace, dim3((unsigned)(trace_wide(s->dev, s->ps) ? s->trace_blocks_wide : s->trace_blocks)), dim3(JADE_TRACE_BLOCK), 0, s->stream, s->dev, s->ps,
                           s->b_queue.as<uint32_t>(), qc + j, s->b_spill.as<uint32_t>(), s->b_ctr.as<DevCounters>(), 0u);
        HIP_TRY(hipEventRecord(s->ev_batch[2 * j + 1], s->stream));
      }
      HIP_TRY(hipGetLastError());
      QueueCtl host_ring[JADE_CTL_RING];
      HIP_TRY(hipMemcpyAsync(host_ring, qc, sizeof(QueueCtl) * B, hipMemcpyDeviceToHost, s->stream));
      HIP_TRY(hipEventRecord(ev1, s->stream));  // the end of the step, if this batch is its last (recorded again otherwise)
      HIP_TRY(hipStreamSynchronize(s->stream));
      s->host_syncs += 1;
      if (trace_pending) {  // the k_trace launch of the pass before the batch
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, ta, tb));
        trace_ms += t;
        launches += 1;
        trace_pending = false;
      }
      bool ended = false, stopped = false;
      int real = 0;        // passes of the batch that ran (the rest found the paths ended, or the carry-over point reached)
      int pass_timed = 0;  // k_trace launches of the batch accounted for so far
      for (int j = 0; j < B; ++j) {
        if (host_ring[j].count == 0) {
          if (host_ring[j].active == 0) {  // this pass emitted nothing and left nothing active: every path has ended
            // (a pass that ran and finished the last paths, or - behind it - one that found nothing to do)
            if (j == 0 || host_ring[j - 1].count != 0) real = j + 1;
            n_active = 0;
            ended = true;
          } else {  // the device stopped here: fewer than stop_below records are active
            n_active = host_ring[j].active;
            stopped = true;
          }
          break;
        }
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, s->ev_batch[2 * j], s->ev_batch[2 * j + 1]));
        trace_ms += t;
        launches += 1;
        pass_timed = j + 1;
        n_active = host_ring[j].active;
        ++pass_no;
        real = j + 1;
      }
      // the launches behind the end of the step were made all the same (k_trace leaves at its first instruction): they count
      // as launches, with their few microseconds, so that launches and time are what a profiler sees
      for (int j = pass_timed; j < B; ++j) {
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, s->ev_batch[2 * j], s->ev_batch[2 * j + 1]));
        trace_ms += t;
        launches += 1;
      }
      cur = cur0 ^ (real & 1);  // the list the last pass that ran wrote
      arm_dev = nullptr;        // (the host has seen a count since)
      if (ended) {
        closed_by_batch = true;
        break;
      }
      if (carry_now(n_active) || (stopped && !(tail_ok && n_active <= tail_max))) {
        s->tail_pending = true;
        s->carried_active = n_active;
        closed_by_batch = true;
        break;
      }
      continue;  // (another batch - or, the list being short now, k_tail)
    }
    if (!lean_mode && !have_list) {
      cur = 0;
      HIP_TRY(hipMemsetAsync(qc, 0, 16, s->stream));
      hipLaunchKernelGGL(k_arm, dim3((unsigned)(((size_t)npix + JADE_ARM_BLOCK * JADE_ARM_PER_THREAD - 1) / (JADE_ARM_BLOCK * JADE_ARM_PER_THREAD))), dim3(JADE_ARM_BLOCK), 0, s->stream, s->ps, target_spp,
                         s->b_active[0].as<uint32_t>(), qc);
      have_list = true;
    }
    HIP_TRY(hipMemsetAsync(qc, 0, 16, s->stream));  // count, active, next, heavy
    if (log_passes) HIP_TRY(hipEventRecord(sa, s->stream));
    const unsigned nb = (n_active + JADE_SHADE_BLOCK - 1) / JADE_SHADE_BLOCK;
    if (lean_mode && fused && pass_no == 0) {
      // the step's first pass, fused: light samples run to completion inside k_light, everything else is handed over
      for (hipEvent_t& e : s->ev_light)
        if (!e) HIP_TRY(hipEventCreate(&e));
      HIP_TRY(hipEventRecord(s->ev_light[0], s->stream));
      light_timed = true;
      {
        // packets, unless the last step gave most of them up (a frame the statue fills: every packet fans out, and the per-lane
        // kernel is then the better first pass - same bits, so the choice is free to make per step)
        const bool packet = s->tun.light_packet && s->packet_blocks > 0 && s->packets_given_up < JADE_PACKET_GIVE_UP_LIMIT;
        const unsigned lb = (unsigned)std::min<size_t>((size_t)(packet ? s->packet_blocks : s->light_blocks), ((size_t)npix + JADE_TRACE_BLOCK - 1) / JADE_TRACE_BLOCK);
        const uint32_t n_waves = lb * (JADE_TRACE_BLOCK / 64);
        // a wave takes every n_waves-th chunk of 64 records: its region must hold all of them
        const uint32_t region_cap = (uint32_t)((((size_t)npix + 63) / 64 + n_waves - 1) / n_waves) * 64u;
        if ((size_t)region_cap * n_waves > s->b_active[0].bytes / 4 || (size_t)2 * n_waves * 4 > s->b_wavecnt.bytes)
          return fail(JADE_ERR_DEVICE, "hand-over regions do not fit (internal sizing error)");
        if (packet)
          hipLaunchKernelGGL(k_light_packet, dim3(lb), dim3(JADE_TRACE_BLOCK), 0, s->stream, s->dev, s->ps, s->rc, s->b_tiles.as<int32_t>(), target_spp,
                             s->b_active[0].as<uint32_t>(), region_cap, s->b_wavecnt.as<uint32_t>(), s->b_ctr.as<DevCounters>(),
                             (uint32_t)s->tun.packet_budget);
        else
          hipLaunchKernelGGL(k_light, dim3(lb), dim3(JADE_TRACE_BLOCK), 0, s->stream, s->dev, s->ps, s->rc, s->b_tiles.as<int32_t>(), target_spp,
                             s->b_active[0].as<uint32_t>(), region_cap, s->b_wavecnt.as<uint32_t>(), s->b_spill.as<uint32_t>(),
                             s->b_ctr.as<DevCounters>());
        hipLaunchKernelGGL(k_heavy_scan, dim3(1), dim3(1024), 0, s->stream, s->b_wavecnt.as<uint32_t>(), n_waves, qc);
        hipLaunchKernelGGL(k_heavy_pack, dim3(n_waves), dim3(256), 0, s->stream, s->b_active[0].as<uint32_t>(), region_cap,
                           s->b_wavecnt.as<uint32_t>(), n_waves, s->b_active[1].as<uint32_t>());
      }
      HIP_TRY(hipEventRecord(s->ev_light[1], s->stream));
      if (log_passes) HIP_TRY(hipEventRecord(sm, s->stream));
      // k_light has run every other record to the end of its samples, so the records this pass leaves active ARE the active
      // list (written over k_light's regions, which k_heavy_pack has emptied): no k_arm scan of all records after it
      hipLaunchKernelGGL(shade_kernel, dim3(nb), dim3(JADE_SHADE_BLOCK), 0, s->stream, s->dev, s->ps, s->rc, s->b_tiles.as<int32_t>(),
                         target_spp, s->b_active[1].as<uint32_t>(), 0u, &qc->heavy, s->b_active[0].as<uint32_t>(), s->b_queue.as<uint32_t>(), qc,
                         s->b_ctr.as<DevCounters>(), (const QueueCtl*)nullptr, 0u);
      have_list = true;
      cur = 0;
    } else if (lean_mode) {
      // b_active[1] carries the hand-over list; no active list is kept in this mode
      hipLaunchKernelGGL(k_shade_lean, dim3((unsigned)((npix + JADE_LEAN_BLOCK - 1) / JADE_LEAN_BLOCK)), dim3(JADE_LEAN_BLOCK), 0,
                         s->stream, s->dev, s->ps, s->rc, s->b_tiles.as<int32_t>(), target_spp, s->b_active[1].as<uint32_t>(),
                         s->b_queue.as<uint32_t>(), qc, s->b_ctr.as<DevCounters>());
      if (log_passes) HIP_TRY(hipEventRecord(sm, s->stream));
      // only a record that was active can be handed over: n_active bounds the grid, the count stays on the device
      hipLaunchKernelGGL(shade_kernel, dim3(nb), dim3(JADE_SHADE_BLOCK), 0, s->stream, s->dev, s->ps, s->rc, s->b_tiles.as<int32_t>(),
                         target_spp, s->b_active[1].as<uint32_t>(), 0u, &qc->heavy, (uint32_t*)nullptr, s->b_queue.as<uint32_t>(), qc,
                         s->b_ctr.as<DevCounters>(), (const QueueCtl*)nullptr, 0u);
      have_list = false;
    } else {
      if (log_passes) HIP_TRY(hipEventRecord(sm, s->stream));
      hipLaunchKernelGGL(shade_kernel, dim3(nb), dim3(JADE_SHADE_BLOCK), 0, s->stream, s->dev, s->ps, s->rc, s->b_tiles.as<int32_t>(),
                         target_spp, s->b_active[cur].as<uint32_t>(), n_active, (const uint32_t*)nullptr,
                         s->b_active[cur ^ 1].as<uint32_t>(), s->b_queue.as<uint32_t>(), qc, s->b_ctr.as<DevCounters>(), (const QueueCtl*)nullptr, 0u);
      cur ^= 1;
    }
    HIP_TRY(hipGetLastError());
    if (log_passes) HIP_TRY(hipEventRecord(sb, s->stream));
    HIP_TRY(hipMemcpyAsync(host_ctl, qc, 12, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
  s->host_syncs += 1;
    if (light_timed) {  // (the stream was just synchronised)
      float lt = 0;
      HIP_TRY(hipEventElapsedTime(&lt, s->ev_light[0], s->ev_light[1]));
      s->light_ms += lt;
      light_timed = false;
    }
    if (log_passes) {
      HIP_TRY(hipEventElapsedTime(&shade_ms, sa, sb));
      HIP_TRY(hipEventElapsedTime(&lean_ms, sa, sm));
    }
    if (trace_pending) {
      float t = 0;
      HIP_TRY(hipEventElapsedTime(&t, ta, tb));
      trace_ms += t;
      launches += 1;
      trace_pending = false;
    }
    n_active = host_ctl[1];
    if (host_ctl[0] == 0) break;
    const uint32_t* trace_queue = s->b_queue.as<uint32_t>();
    HIP_TRY(hipEventRecord(ta, s->stream));  // (the ordering counts as trace time)
    if (s->sort_rays && host_ctl[0] >= s->tun.sort_min && host_ctl[0] <= s->sort_cap) {
      const uint32_t n = host_ctl[0];
      // (the keys are in b_sortkey already: the shading kernel wrote each beside its queue entry - PathState.keyq, round 4; until then
      // k_ray_keys made them here, from three scattered sectors per ray, 2 % of a C5 step.  JADE_SORT_KEYS_KERNEL=1 brings that back.)
      if (s->tun.sort_keys_kernel)
        hipLaunchKernelGGL(k_ray_keys, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->ps, s->b_queue.as<uint32_t>(), n, s->b_sortkey.as<uint32_t>(),
                           s->b_sortpos.as<uint32_t>(), s->n_emit, s->ps.key_tri_bits);
      // The temporary storage was sized once, for (sort_cap entries, bits 0..32).  rocPRIM's need shrinks with the length and with
      // the bit range (fewer digit places, fewer look-back states; a short queue takes its merge-sort path: two buffers of n), and a
      // buffer that is too small is an error return, not a fault (rocprim/detail/temp_storage.hpp: partition) - asked again here,
      // on the host, for THIS length, so that the claim is checked and not assumed ("the round-3 fault").
      size_t need = 0;
      HIP_TRY(rocprim::radix_sort_pairs(nullptr, need, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, 0u, 32u, s->stream));
      if (need > s->sort_tmp_bytes) return fail(JADE_ERR_DEVICE, "ray-queue sort: temporary storage smaller than rocPRIM asks for this length (internal sizing error)");
      size_t tmp = s->sort_tmp_bytes;
      HIP_TRY(rocprim::radix_sort_pairs(s->b_sorttmp.p, tmp, s->b_sortkey.as<uint32_t>(), s->b_sortkey2.as<uint32_t>(), s->b_sortpos.as<uint32_t>(),
                                        s->b_sortq.as<uint32_t>(), (size_t)n, 0u, 32u, s->stream));
      trace_queue = s->b_sortq.as<uint32_t>();  // positions, in the order k_trace is to take them (PathState.idxq)
    }
    PathState tps = s->ps;
    tps.idxq = trace_queue != s->b_queue.as<uint32_t>() ? s->b_queue.as<uint32_t>() : nullptr;
    hipLaunchKernelGGL(trace_wide(s->dev, s->ps) ? k_trace_wide : k_trace, dim3((unsigned)(trace_wide(s->dev, s->ps) ? s->trace_blocks_wide : s->trace_blocks)), dim3(JADE_TRACE_BLOCK), 0, s->stream, s->dev, tps,
                       trace_queue, qc, s->b_spill.as<uint32_t>(), s->b_ctr.as<DevCounters>(),
                       trace_chunk(s, host_ctl[0]));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(tb, s->stream));
    trace_pending = true;
    if (carry_now(n_active)) {
      // The few long paths left would take dozens of nearly empty passes: leave them suspended (their
      // rays are traced, their hits wait to be folded in) for the next step's first pass, or for flush.
      s->tail_pending = true;
      s->carried_active = n_active;
      carried = true;
    }
    if (log_passes) {
      HIP_TRY(hipEventSynchronize(tb));
      float t = 0;
      HIP_TRY(hipEventElapsedTime(&t, ta, tb));
      DevCounters cc{};  // development log only: node records and triangle tests of this launch (counters accumulate over the step)
      HIP_TRY(sum_counters(s, &cc));
      static thread_local unsigned long long v_prev = 0, t_prev = 0;
      if (pass_no == 0 && (cc.nodes_visited < v_prev || cc.tris_tested < t_prev)) v_prev = t_prev = 0;
      if (cc.nodes_visited < v_prev) v_prev = t_prev = 0;
      fprintf(stderr, "[jade] pass %4d active %9u rays %9u shade %7.3f ms (lean %6.3f) trace %8.3f ms (%7.1f Mray/s) V/ray %6.1f T/ray %5.1f\n", pass_no,
              n_active, host_ctl[0], shade_ms, lean_ms, t, host_ctl[0] / (t * 1e3), (double)(cc.nodes_visited - v_prev) / host_ctl[0],
              (double)(cc.tris_tested - t_prev) / host_ctl[0]);
      v_prev = cc.nodes_visited;
      t_prev = cc.tris_tested;
    }
    ++pass_no;
    if (carried) break;
  }
  if (trace_pending) {  // the launch timed by (ta, tb) when the loop stopped right after it
    HIP_TRY(hipEventSynchronize(tb));
    float t = 0;
    HIP_TRY(hipEventElapsedTime(&t, ta, tb));
    trace_ms += t;
    launches += 1;
    trace_pending = false;
  }
  if (!closed_by_batch) {
    HIP_TRY(hipEventRecord(ev1, s->stream));
    HIP_TRY(hipEventSynchronize(ev1));
    s->host_syncs += 1;
  }
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
  *ms_out = ms;
  *trace_ms_out = trace_ms;
  *launches_out = launches;
  return JADE_OK;
}

// runs the passes for every sample up to spp_done; may_carry: the last paths may be left for later
static int advance(jade_scene* s, int64_t from0, bool may_carry, jade_stats* st) {
  HIP_TRY(hipMemsetAsync(s->b_ctr.p, 0, sizeof(DevCounters) * JADE_CTR_SHARDS, s->stream));
  double ms = 0, trace_ms = 0;
  uint64_t launches = 0;
  // pixel rotation only: one block of JADE_SAMPLE_LANES samples at a time (see PathState): records may not
  // run ahead into the next block while another record still owns a (pixel, lane) sum of this one
  for (int64_t from = from0;;) {
    int64_t to = (from / JADE_SAMPLE_LANES + 1) * JADE_SAMPLE_LANES;
    if (to > s->spp_done || s->ps.stride == 0) to = s->spp_done;  // no rotation: no block barrier needed
    double m1 = 0, t1 = 0;
    uint64_t l1 = 0;
    int rc = run_passes(s, from, (uint32_t)to, may_carry && s->ps.stride == 0, &m1, &t1, &l1);
    if (rc) return rc;
    ms += m1; trace_ms += t1; launches += l1;
    from = to;
    if (from >= s->spp_done) break;
  }
  DevCounters c{};
  HIP_TRY(sum_counters(s, &c));  // (32 KB: also read when the caller wants no statistics - the next step's choice of first pass depends on it)
  if (c.pad[0]) s->packets_given_up = (double)c.pad[1] / (double)c.pad[0];
  if (s->tun.log_passes && c.pad[0])
    fprintf(stderr, "[jade] first pass: %llu packets, %llu given up and handed to the wavefront passes (%.2f %%)\n", (unsigned long long)c.pad[0],
            (unsigned long long)c.pad[1], 100.0 * (double)c.pad[1] / (double)c.pad[0]);
  if (st) {
    st->rays_primary += c.rays_primary;
    st->rays_secondary += c.rays_shadow + c.rays_env + c.rays_indirect + c.rays_mirror + c.rays_refract;
    st->rays_shadow += c.rays_shadow;
    st->rays_env += c.rays_env;
    st->rays_indirect += c.rays_indirect;
    st->rays_mirror += c.rays_mirror;
    st->rays_refract += c.rays_refract;
    st->rays_inline += c.rays_inline;
    st->nodes_inline += c.nodes_inline;
    st->tris_inline += c.tris_inline;
    st->rays_cached += c.rays_cached;
    st->rays_tail += c.rays_tail;
    st->nodes_tail += c.nodes_tail;
    st->tris_tail += c.tris_tail;
    st->tail_ms += s->tail_ms;
    st->tail_launches += s->tail_launches;
    st->light_ms += s->light_ms;
    st->nodes_visited += c.nodes_visited;
    st->tris_tested += c.tris_tested;
    st->shaded_hits += c.shaded_hits;
    st->samples += c.samples;
    st->kernel_ms += ms;
    st->trace_ms += trace_ms;
    st->trace_launches += launches;
    st->host_syncs += s->host_syncs;
  }
  s->host_syncs = 0;
  s->light_ms = 0;
  s->tail_ms = 0;
  s->tail_launches = 0;
  s->tail_records = 0;
  return JADE_OK;
}

// The steps add more samples than jade_render_begin was told of: more lanes of partial sums (jade_rt.h,
// JADE_Q_SUM_LANES).
static int grow_sums(jade_scene* s, int64_t spp_total) {
  PathState& P = s->ps;
  int lanes = P.sum_lanes;
  while (lanes < JADE_SAMPLE_LANES && lanes < spp_total) lanes <<= 1;
  if (lanes == P.sum_lanes) return JADE_OK;
  DevBuf nb;
  // (lane l of pixel p sits at (l * npx + p) * 3: more lanes are more of the same behind the old ones; the new lanes are +0)
  const size_t bytes_old = (size_t)P.sum_lanes * P.npx * 12, bytes_new = (size_t)lanes * P.npx * 12;
  HIP_TRY(nb.alloc(bytes_new));
  HIP_TRY(hipMemsetAsync(nb.p, 0, bytes_new, s->stream));
  HIP_TRY(hipMemcpyAsync(nb.p, s->b_sum.p, bytes_old, hipMemcpyDeviceToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  std::swap(s->b_sum.p, nb.p);
  std::swap(s->b_sum.bytes, nb.bytes);
  P.sum = s->b_sum.as<float>();
  P.sum_lanes = lanes;
  return JADE_OK;
}

int jade_render_query(jade_scene* s, int what, int64_t* value) {
  if (!s || !value || !s->have_rp) return fail(JADE_ERR_INVALID, "jade_render_begin not called");
  switch (what) {
    case JADE_Q_RECORDS_PER_PIXEL: *value = s->ps.npix ? s->ps.rpp : 0; return JADE_OK;
    case JADE_Q_STATE_BYTES:
      *value = s->ps.npix ? (int64_t)(s->b_state.bytes + s->b_sum.bytes + s->b_queue.bytes + s->b_rayq.bytes + s->b_active[0].bytes + s->b_active[1].bytes +
                                      s->b_wavecnt.bytes + s->b_spill.bytes + s->b_sortkey.bytes + s->b_sortkey2.bytes + s->b_sortpos.bytes + s->b_sortq.bytes + s->b_sorttmp.bytes)
                          : 0;
      return JADE_OK;
    case JADE_Q_SUM_LANES: *value = s->ps.npix ? s->ps.sum_lanes : 0; return JADE_OK;
    default: return fail(JADE_ERR_INVALID, "unknown query");
  }
}

int jade_render_step(jade_scene* s, int32_t spp, jade_stats* st) {
  if (!s || !s->have_rp) return fail(JADE_ERR_INVALID, "jade_render_begin not called");
  if (spp < 0) return fail(JADE_ERR_INVALID, "negative spp");
  HIP_TRY(hipSetDevice(s->device));
  if (s->ps.npix == 0 || spp == 0) {
    s->spp_done += spp;
    return JADE_OK;
  }
  // (the sample count moves only once the partial sums have room for the new samples: after a failed grow_sums - out of memory -
  // a resolve must not divide by samples that were never rendered)
  if (s->spp_done + spp > s->ps.sum_lanes && s->ps.sum_lanes < JADE_SAMPLE_LANES)
    if (int rc = grow_sums(s, s->spp_done + spp)) return rc;
  s->spp_done += spp;
  return advance(s, s->spp_done - spp, s->tun.carry, st);
}

int jade_render_flush(jade_scene* s, jade_stats* st) {
  if (!s || !s->have_rp) return fail(JADE_ERR_INVALID, "jade_render_begin not called");
  if (!s->tail_pending || s->ps.npix == 0) return JADE_OK;
  HIP_TRY(hipSetDevice(s->device));
  return advance(s, s->spp_done, false, st);
}

static int resolve_to(jade_scene* s, int tonemap, float limit, float* dev_rgb, uint8_t* dev_bgr, hipStream_t stream) {
  const int npix = s->ps.npx;
  if (npix == 0) return JADE_OK;
  float inv = (float)(1.0 / (double)s->spp_done);  // vec3(1.0 / spp), PathTrace.cu:1457
  hipLaunchKernelGGL(k_resolve, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, stream, s->ps, s->rc, s->b_tiles.as<int32_t>(), inv, tonemap, limit, dev_rgb, dev_bgr);
  HIP_TRY(hipGetLastError());
  return JADE_OK;
}

int jade_render_resolve(jade_scene* s, float* out_rgb, uint8_t* out_bgr8) {
  return jade_render_resolve_ex(s, JADE_TONEMAP_ACES, 0.0f, out_rgb, out_bgr8);
}

int jade_render_resolve_ex(jade_scene* s, int tonemap, float limit, float* out_rgb, uint8_t* out_bgr8) {
  if (!s || !s->have_rp) return fail(JADE_ERR_INVALID, "jade_render_begin not called");
  if (tonemap != JADE_TONEMAP_ACES && tonemap != JADE_TONEMAP_REINHARD) return fail(JADE_ERR_INVALID, "unknown tone operator");
  if (s->spp_done <= 0) return fail(JADE_ERR_INVALID, "no samples rendered yet");
  HIP_TRY(hipSetDevice(s->device));
  if (int rc = jade_render_flush(s, nullptr)) return rc;
  const int npix = s->ps.npx;
  if (npix == 0) return JADE_OK;
  if (out_rgb) HIP_TRY(s->b_out_rgb.alloc((size_t)npix * 12));
  if (out_bgr8) HIP_TRY(s->b_out_bgr.alloc((size_t)npix * 3));
  int rc = resolve_to(s, tonemap, limit, out_rgb ? s->b_out_rgb.as<float>() : nullptr, out_bgr8 ? s->b_out_bgr.as<uint8_t>() : nullptr, s->stream);
  if (rc) return rc;
  std::vector<float> hrgb;
  std::vector<uint8_t> hbgr;
  if (out_rgb) {
    hrgb.resize((size_t)npix * 3);
    HIP_TRY(hipMemcpyAsync(hrgb.data(), s->b_out_rgb.p, hrgb.size() * 4, hipMemcpyDeviceToHost, s->stream));
  }
  if (out_bgr8) {
    hbgr.resize((size_t)npix * 3);
    HIP_TRY(hipMemcpyAsync(hbgr.data(), s->b_out_bgr.p, hbgr.size(), hipMemcpyDeviceToHost, s->stream));
  }
  HIP_TRY(hipStreamSynchronize(s->stream));
  // scatter the compact tiles into the caller's frame; other ranks' pixels untouched
  const int W = s->rp.width, H = s->rp.height, tx = s->rc.tiles_x;
  for (size_t t = 0; t < s->tile_ids.size(); ++t) {
    int x0 = (s->tile_ids[t] % tx) * JADE_TILE_SIZE, y0 = (s->tile_ids[t] / tx) * JADE_TILE_SIZE;
    int ww = std::min(JADE_TILE_SIZE, W - x0), hh = std::min(JADE_TILE_SIZE, H - y0);
    for (int ly = 0; ly < hh; ++ly) {
      size_t src = (t * 256 + (size_t)ly * 16) * 3, dst = ((size_t)(y0 + ly) * W + x0) * 3;
      if (out_rgb) memcpy(out_rgb + dst, hrgb.data() + src, (size_t)ww * 12);
      if (out_bgr8) memcpy(out_bgr8 + dst, hbgr.data() + src, (size_t)ww * 3);
    }
  }
  return JADE_OK;
}

int jade_render_resolve_tiles_device(jade_scene* s, float* dev_tiles, void* stream) {
  if (!s || !s->have_rp || !dev_tiles) return fail(JADE_ERR_INVALID, "bad arguments");
  if (s->spp_done <= 0) return fail(JADE_ERR_INVALID, "no samples rendered yet");
  HIP_TRY(hipSetDevice(s->device));
  if (int rc = jade_render_flush(s, nullptr)) return rc;
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (int rc = resolve_to(s, JADE_TONEMAP_ACES, 0.0f, dev_tiles, nullptr, (hipStream_t)stream)) return rc;
  // k_resolve reads the partial sums on the CALLER's stream: the scene's own stream must not start the next
  // step (which adds to them) before it has finished
  if (!s->ev_resolve) HIP_TRY(hipEventCreateWithFlags(&s->ev_resolve, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(s->ev_resolve, (hipStream_t)stream));
  HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_resolve, 0));
  return JADE_OK;
}

int jade_render(jade_scene* s, const jade_render_params* rp, float* out_rgb, uint8_t* out_bgr8, jade_stats* st) {
  if (!rp) return fail(JADE_ERR_INVALID, "null argument");
  if (rp->spp <= 0) return fail(JADE_ERR_INVALID, "spp must be positive");
  int rc = jade_render_begin(s, rp);
  if (rc) return rc;
  rc = jade_render_step(s, rp->spp, st);
  if (rc == JADE_OK) rc = jade_render_flush(s, st);  // (resolve would flush too, but without the statistics)
  if (rc) return rc;
  return jade_render_resolve(s, out_rgb, out_bgr8);
}

// host twin of k_resolve's tone map + pack (same jade_fpmath.h routines, same flags: same bits)
static void tonemap_pack_host(const float* m, int tonemap, float limit, uint8_t* bgr) {
  float v[3] = {m[0], m[1], m[2]};
  float rein = 1.0f;
  if (tonemap == JADE_TONEMAP_REINHARD) {
    float luminance = (float)(0.3 * (double)m[0] + 0.6 * (double)m[1] + 0.1 * (double)m[2]);
    rein = (float)(1.0 / (1.0 + (double)(luminance / limit)));
  }
  for (int k = 0; k < 3; ++k) {
    float x = v[k];
    if (tonemap == JADE_TONEMAP_REINHARD) {
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
    float x = v[2 - k];
    bgr[k] = (x >= 0.0f) ? (uint8_t)x : (uint8_t)0;
  }
}

int jade_render_multi(jade_scene* const* scenes, int ndev, const jade_render_params* rp, float* out_rgb, uint8_t* out_bgr8,
                      jade_stats* st) {
  if (!scenes || ndev <= 0 || !rp) return fail(JADE_ERR_INVALID, "null argument");
  if (rp->spp <= 0 || rp->width <= 0 || rp->height <= 0) return fail(JADE_ERR_INVALID, "bad image size or spp");
  for (int i = 0; i < ndev; ++i)
    if (!scenes[i]) return fail(JADE_ERR_INVALID, "null scene");
  // 1. every device renders its share (one host thread each) and resolves it into a device buffer
  std::vector<int> rcs(ndev, JADE_OK);
  std::vector<std::string> msgs(ndev);
  std::vector<jade_stats> sts(ndev);
  for (auto& x : sts) memset(&x, 0, sizeof x);
  auto work = [&](int i) {
    jade_scene* s = scenes[i];
    jade_render_params p = *rp;
    p.tile_rank = i;
    p.tile_nranks = ndev;
    p.device_id = s->device;
    int rc = jade_render_begin(s, &p);
    if (rc == JADE_OK) rc = jade_render_step(s, p.spp, &sts[i]);
    if (rc == JADE_OK) rc = jade_render_flush(s, &sts[i]);
    if (rc == JADE_OK && s->ps.npx > 0) {
      hipError_t e = s->b_out_rgb.alloc((size_t)s->ps.npx * 12);
      if (e != hipSuccess) rc = jade_fail(JADE_ERR_NOMEM, "tile buffer allocation failed");
      if (rc == JADE_OK) rc = resolve_to(s, JADE_TONEMAP_ACES, 0.0f, s->b_out_rgb.as<float>(), nullptr, s->stream);
      if (rc == JADE_OK && hipStreamSynchronize(s->stream) != hipSuccess) rc = jade_fail(JADE_ERR_DEVICE, "stream sync failed");
    }
    rcs[i] = rc;
    if (rc) msgs[i] = g_err;  // g_err is thread-local: carry the text back to the caller's thread
  };
  std::vector<std::thread> th;
  for (int i = 1; i < ndev; ++i) th.emplace_back(work, i);
  work(0);
  for (auto& t : th) t.join();
  for (int i = 0; i < ndev; ++i)
    if (rcs[i]) return fail(rcs[i], "device share " + std::to_string(i) + ": " + msgs[i]);
  // 2. the ONE exchange step: gather the compact tile buffers on the device of scenes[0].  Distinct devices: an RCCL
  // gather over xGMI (ncclCommInitAll + grouped send/recv, the form ncclGather itself expands to, so
  // that ranks may contribute different tile counts).  The same device several times (a one-GPU rehearsal of the
  // partition): RCCL cannot put two ranks on one device, the shares are copied device-to-device instead.
  jade_scene* s0 = scenes[0];
  HIP_TRY(hipSetDevice(s0->device));
  size_t total = 0;
  std::vector<size_t> off(ndev);
  for (int i = 0; i < ndev; ++i) {
    off[i] = total;
    total += (size_t)scenes[i]->ps.npx * 3;
  }
  DevBuf gather;
  HIP_TRY(gather.alloc(total * 4));
  bool distinct = true;
  for (int i = 0; i < ndev; ++i)
    for (int j = 0; j < i; ++j) distinct = distinct && scenes[i]->device != scenes[j]->device;
  // JADE_FORCE_RCCL=1 (tests): take the RCCL path for a single share too - library load, communicator, empty group
  if (distinct && (ndev > 1 || s0->tun.force_rccl)) {
    if (int rc = rccl_gather(scenes, ndev, gather.as<float>(), off)) return rc;
  } else {
    for (int i = 0; i < ndev; ++i) {
      size_t bytes = (size_t)scenes[i]->ps.npx * 12;
      if (!bytes) continue;
      HIP_TRY(hipMemcpyAsync(gather.as<float>() + off[i], scenes[i]->b_out_rgb.p, bytes, hipMemcpyDeviceToDevice, s0->stream));
    }
  }
  std::vector<float> host(total);
  HIP_TRY(hipMemcpyAsync(host.data(), gather.p, total * 4, hipMemcpyDeviceToHost, s0->stream));
  HIP_TRY(hipStreamSynchronize(s0->stream));
  // 3. un-tile into the caller's frame, tone-map once
  const int W = rp->width, H = rp->height;
  const int tx = (W + JADE_TILE_SIZE - 1) / JADE_TILE_SIZE;
  std::vector<float> frame;
  float* dst_rgb = out_rgb;
  if (!dst_rgb) {
    frame.resize((size_t)W * H * 3);
    dst_rgb = frame.data();
  }
  for (int i = 0; i < ndev; ++i) {
    const jade_scene* s = scenes[i];
    for (size_t t = 0; t < s->tile_ids.size(); ++t) {
      int x0 = (s->tile_ids[t] % tx) * JADE_TILE_SIZE, y0 = (s->tile_ids[t] / tx) * JADE_TILE_SIZE;
      int ww = std::min(JADE_TILE_SIZE, W - x0), hh = std::min(JADE_TILE_SIZE, H - y0);
      for (int ly = 0; ly < hh; ++ly)
        memcpy(dst_rgb + ((size_t)(y0 + ly) * W + x0) * 3, host.data() + off[i] + (t * 256 + (size_t)ly * 16) * 3, (size_t)ww * 12);
    }
  }
  if (out_bgr8)
    for (size_t p = 0; p < (size_t)W * H; ++p) tonemap_pack_host(dst_rgb + 3 * p, JADE_TONEMAP_ACES, 0.0f, out_bgr8 + 3 * p);
  if (st)
    for (int i = 0; i < ndev; ++i) {
      st->rays_primary += sts[i].rays_primary; st->rays_secondary += sts[i].rays_secondary;
      st->rays_shadow += sts[i].rays_shadow; st->rays_env += sts[i].rays_env; st->rays_indirect += sts[i].rays_indirect;
      st->rays_mirror += sts[i].rays_mirror; st->rays_refract += sts[i].rays_refract; st->host_syncs += sts[i].host_syncs;
      st->rays_inline += sts[i].rays_inline;
      st->nodes_inline += sts[i].nodes_inline; st->tris_inline += sts[i].tris_inline;
      st->rays_cached += sts[i].rays_cached; st->rays_tail += sts[i].rays_tail; st->nodes_tail += sts[i].nodes_tail; st->tris_tail += sts[i].tris_tail;
      st->tail_ms = std::max(st->tail_ms, sts[i].tail_ms); st->tail_launches += sts[i].tail_launches;
      st->light_ms = std::max(st->light_ms, sts[i].light_ms);
      st->nodes_visited += sts[i].nodes_visited; st->tris_tested += sts[i].tris_tested;
      st->shaded_hits += sts[i].shaded_hits; st->samples += sts[i].samples;
      st->kernel_ms = std::max(st->kernel_ms, sts[i].kernel_ms);  // the shares run concurrently
      st->trace_ms = std::max(st->trace_ms, sts[i].trace_ms);
      st->trace_launches += sts[i].trace_launches;
    }
  return JADE_OK;
}

// limits (nullable): per ray, the distance below which a recorded hit ends the walk (JADE_WALK_EARLY_EXIT as k_shade asks
// for it, jade_device.h); null = the reference's walk
static int trace_rays_impl(jade_scene* s, int32_t n, const float* origins, const float* dirs, const int32_t* skip, const float* limits,
                           int32_t* hit_index, float* hit_dist, float* hit_point, jade_stats* st, bool cached = false) {
  if (!s || n < 0 || !origins || !dirs || !skip || !hit_index) return fail(JADE_ERR_INVALID, "null argument");
  if (n == 0) return JADE_OK;
  HIP_TRY(hipSetDevice(s->device));
  // a throw-away PathState with one slot per "pixel"
  const size_t N = (size_t)n;
  std::vector<float4> so(N), sl(N), hp(N, make_float4(0.0f, 0.0f, 0.0f, 0.0f));  // (the hit point of a miss is reported as zeros)
  for (size_t i = 0; i < N; ++i) {
    const int32_t sk = skip[i] < 0 ? -1 : skip[i];  // any negative value means "no source triangle" (the device keeps -2 for camera rays)
    float skf, qf;
    const int32_t queued = -1;
    memcpy(&skf, &sk, 4);
    memcpy(&qf, &queued, 4);
    so[i] = make_float4(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2], skf);
    sl[i] = make_float4(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], limits ? limits[i] : qf);
  }
  std::vector<uint32_t> q(N);
  for (size_t i = 0; i < N; ++i) q[i] = (uint32_t)i;
  DevBuf b_orgs, b_slot, b_hitp, b_q, b_spill;
  HIP_TRY(upload(b_orgs, so.data(), so.size(), s->stream));
  HIP_TRY(upload(b_slot, sl.data(), sl.size(), s->stream));
  HIP_TRY(upload(b_hitp, hp.data(), hp.size(), s->stream));
  HIP_TRY(upload(b_q, q.data(), N, s->stream));
  HIP_TRY(b_spill.alloc((size_t)(JADE_BVH_STACK_CAPACITY - JADE_LDS_STACK) * s->trace_blocks * JADE_TRACE_BLOCK * 4));
  PathState P{};
  P.npix = n;
  P.nslots = 1;
  P.orgs = b_orgs.as<float4>();
  P.slot = b_slot.as<float4>();
  P.hitp = b_hitp.as<float4>();
  P.write_all_hits = 1u;  // (this entry point reports point and distance of EVERY ray, also of one that carries a yes/no limit)
  P.early_exit = limits ? (cached ? 2u : 1u) : 0u;
  // everything on the scene's own (non-blocking) stream: the null stream does not order against it
  QueueCtl qc{};
  qc.count = (uint32_t)n;
  HIP_TRY(hipMemcpyAsync(s->b_ctl.p, &qc, 12, hipMemcpyHostToDevice, s->stream));  // count, active, next
  HIP_TRY(hipMemsetAsync(s->b_ctr.p, 0, sizeof(DevCounters) * JADE_CTR_SHARDS, s->stream));
  DevEvent ev0, ev1;
  HIP_TRY(ev0.create());
  HIP_TRY(ev1.create());
  HIP_TRY(hipEventRecord(ev0.e, s->stream));
  hipLaunchKernelGGL(trace_wide(s->dev, P) ? k_trace_wide : k_trace, dim3((unsigned)(trace_wide(s->dev, P) ? s->trace_blocks_wide : s->trace_blocks)), dim3(JADE_TRACE_BLOCK), 0, s->stream, s->dev, P, b_q.as<uint32_t>(),
                     s->b_ctl.as<QueueCtl>(), b_spill.as<uint32_t>(), s->b_ctr.as<DevCounters>(), trace_chunk(s, (uint32_t)n));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(ev1.e, s->stream));
  HIP_TRY(hipEventSynchronize(ev1.e));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, ev0.e, ev1.e));
  // HitResult.distance (PathTrace.cu:740) exactly as the kernel compared it (hitArray's `<`, :787); a miss keeps INF (:799)
  HIP_TRY(hipMemcpyAsync(sl.data(), b_slot.p, N * sizeof(float4), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipMemcpyAsync(hp.data(), b_hitp.p, N * sizeof(float4), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  for (size_t i = 0; i < N; ++i) {
    memcpy(&hit_index[i], &sl[i].w, 4);
    if (hit_dist) hit_dist[i] = hp[i].w;
    if (hit_point) {
      const bool hit = hit_index[i] >= 0;
      hit_point[3 * i] = hit ? hp[i].x : 0.0f;
      hit_point[3 * i + 1] = hit ? hp[i].y : 0.0f;
      hit_point[3 * i + 2] = hit ? hp[i].z : 0.0f;
    }
  }
  if (st) {
    DevCounters c{};
    HIP_TRY(sum_counters(s, &c));
    st->rays_secondary += (uint64_t)n;
    st->nodes_visited += c.nodes_visited;
    st->tris_tested += c.tris_tested;
    st->rays_cached += c.rays_cached;
    st->kernel_ms += ms;
    st->trace_ms += ms;
    st->trace_launches += 1;
  }
  return JADE_OK;
}

int jade_trace_rays(jade_scene* s, int32_t n, const float* origins, const float* dirs, const int32_t* skip, int32_t* hit_index,
                    float* hit_dist, float* hit_point, jade_stats* st) {
  return trace_rays_impl(s, n, origins, dirs, skip, nullptr, hit_index, hit_dist, hit_point, st);
}

// ---- development / test entry points: NOT part of jade_rt.h and NOT in libjade_hip.so.  Only builds with -DJADE_DEBUG_EXPORTS=1
// have them: libjade_hip_debug.so (make hipvariants; tests/conftest.py `hip_debug`) and the JADE_TRACE_PROFILE build.
#ifndef JADE_DEBUG_EXPORTS
#define JADE_DEBUG_EXPORTS JADE_TRACE_PROFILE
#endif
#if JADE_DEBUG_EXPORTS
// Development / tests (not part of jade_rt.h): jade_trace_rays with a limit per ray - k_trace's early exit on its own, outside
// the integrator.  For a ray whose nearest hit is not nearer than its limit the answer is the reference's; otherwise it is SOME
// recorded hit nearer than the limit (which one depends on the schedule of the wave).
int jade_debug_trace_rays_limit(jade_scene* s, int32_t n, const float* origins, const float* dirs, const int32_t* skip, const float* limits,
                                int32_t* hit_index, float* hit_dist, float* hit_point, jade_stats* st) {
  if (!limits) return fail(JADE_ERR_INVALID, "null argument");
  return trace_rays_impl(s, n, origins, dirs, skip, limits, hit_index, hit_dist, hit_point, st);
}
// ... and with the occluder cache (JADE_WALK_EARLY_EXIT_CACHED): a ray with a source triangle and a limit that is not a NaN is a
// yes/no query keyed by that triangle (query kind: "towards emitter 0", or by octant in a scene without emitters)
int jade_debug_trace_rays_cached(jade_scene* s, int32_t n, const float* origins, const float* dirs, const int32_t* skip, const float* limits,
                                 int32_t* hit_index, float* hit_dist, float* hit_point, jade_stats* st) {
  if (!limits) return fail(JADE_ERR_INVALID, "null argument");
  return trace_rays_impl(s, n, origins, dirs, skip, limits, hit_index, hit_dist, hit_point, st, true);
}
// whether jade_scene_create found every child's box inside its parent's (1), what the wide walk and the occluder cache rest on; and
// whether the scene got wide records (bit 1) / an occluder cache (bit 2)
int jade_debug_scene_flags(jade_scene* s) {
  if (!s) return -1;
  return (s->boxes_nested ? 1 : 0) | (s->dev.nodes4 ? 2 : 0) | (s->dev.anyhit ? 4 : 0);
}

// Development / tests (not part of jade_rt.h): shadow_limit (jade_shade.h) for n rays against triangle tri (BVH order): what
// k_shade writes beside a shadow ray's direction.
__global__ void k_debug_shadow_limit(DevScene S, int n, const float* o, const float* d, const int32_t* tri, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = shadow_limit(&S.tris[tri[i]], jv(o[3 * i], o[3 * i + 1], o[3 * i + 2]), jv(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
}
int jade_debug_shadow_limit(jade_scene* s, int32_t n, const float* origins, const float* dirs, const int32_t* tri, float* limit) {
  if (!s || n <= 0 || !origins || !dirs || !tri || !limit) return fail(JADE_ERR_INVALID, "null argument");
  for (int i = 0; i < n; ++i)
    if (tri[i] < 0 || tri[i] >= s->dev.n_tris) return fail(JADE_ERR_INVALID, "triangle index out of range");
  HIP_TRY(hipSetDevice(s->device));
  const size_t N = (size_t)n;
  DevBuf bo, bd, bt, bl;
  HIP_TRY(upload(bo, origins, 3 * N, s->stream));
  HIP_TRY(upload(bd, dirs, 3 * N, s->stream));
  HIP_TRY(upload(bt, tri, N, s->stream));
  HIP_TRY(bl.alloc(N * 4));
  hipLaunchKernelGGL(k_debug_shadow_limit, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s->stream, s->dev, n, bo.as<float>(), bd.as<float>(),
                     bt.as<int32_t>(), bl.as<float>());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(limit, bl.p, N * 4, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return JADE_OK;
}

// Development / tests (not part of jade_rt.h): hitBVH through the PACKET form of the walk (jade_trace.h; what k_light_packet
// runs), rays taken 64 at a time in the order given.  Like jade_trace_rays, plus per-ray counts of node records and triangle tests.
int jade_debug_packet_rays(jade_scene* s, int32_t n, const float* origins, const float* dirs, const int32_t* skip, int32_t* hit_index,
                           float* hit_dist, float* hit_point, uint32_t* v_per_ray, uint32_t* t_per_ray, double* kernel_ms /* nullable */) {
  if (!s || n <= 0 || !origins || !dirs || !skip || !hit_index || !hit_dist || !hit_point || !v_per_ray || !t_per_ray) return fail(JADE_ERR_INVALID, "null argument");
  if (s->bvh_depth > JADE_PACKET_MAX_DEPTH) return fail(JADE_ERR_UNSUPPORTED, "tree too deep for the packet form");
  HIP_TRY(hipSetDevice(s->device));
  const size_t N = (size_t)n;
  DevBuf bo, bd, bs, bh, bt, bp, bv, bc;
  HIP_TRY(upload(bo, origins, 3 * N, s->stream));
  HIP_TRY(upload(bd, dirs, 3 * N, s->stream));
  HIP_TRY(upload(bs, skip, N, s->stream));
  HIP_TRY(bh.alloc(N * 4)); HIP_TRY(bt.alloc(N * 4)); HIP_TRY(bp.alloc(N * 12)); HIP_TRY(bv.alloc(N * 4)); HIP_TRY(bc.alloc(N * 4));
  DevEvent e0, e1;
  HIP_TRY(e0.create());
  HIP_TRY(e1.create());
  HIP_TRY(hipEventRecord(e0.e, s->stream));
  hipLaunchKernelGGL(k_packet_rays, dim3((unsigned)((N + JADE_TRACE_BLOCK - 1) / JADE_TRACE_BLOCK)), dim3(JADE_TRACE_BLOCK), 0, s->stream, s->dev, n,
                     bo.as<float>(), bd.as<float>(), bs.as<int32_t>(), bh.as<int32_t>(), bt.as<float>(), bp.as<float>(), bv.as<uint32_t>(), bc.as<uint32_t>());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(e1.e, s->stream));
  HIP_TRY(hipMemcpyAsync(hit_index, bh.p, N * 4, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipMemcpyAsync(hit_dist, bt.p, N * 4, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipMemcpyAsync(hit_point, bp.p, N * 12, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipMemcpyAsync(v_per_ray, bv.p, N * 4, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipMemcpyAsync(t_per_ray, bc.p, N * 4, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (kernel_ms) {
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0.e, e1.e));
    *kernel_ms = ms;
  }
  return JADE_OK;
}

// Development only (not part of jade_rt.h): the laps of a -DJADE_TRACE_PROFILE=1 build, PL_N values; reset != 0 clears them.
// A product build reports JADE_ERR_UNSUPPORTED.
// the same for k_light_packet's laps and counts (jade_trace.h, PKL_*)
int jade_debug_packet_profile(unsigned long long* out, int n, int reset) {
#if JADE_TRACE_PROFILE
  if (!out || n < PKL_N) return fail(JADE_ERR_INVALID, "need room for PKL_N values");
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_packet_prof), sizeof(unsigned long long) * PKL_N));
  if (reset) {
    unsigned long long z[PKL_N] = {};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_packet_prof), z, sizeof z));
  }
  return PKL_N;
#else
  (void)out; (void)n; (void)reset;
  return -fail(JADE_ERR_UNSUPPORTED, "not a JADE_TRACE_PROFILE build");
#endif
}
int jade_debug_trace_profile(unsigned long long* out, int n, int reset) {
#if JADE_TRACE_PROFILE
  if (!out || n < PL_N) return fail(JADE_ERR_INVALID, "need room for PL_N values");
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace_prof), sizeof(unsigned long long) * PL_N));
  if (reset) {
    unsigned long long z[PL_N] = {};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_prof), z, sizeof z));
  }
  return PL_N;
#else
  (void)out; (void)n; (void)reset;
  return -fail(JADE_ERR_UNSUPPORTED, "not a JADE_TRACE_PROFILE build");
#endif
}
#endif  // JADE_DEBUG_EXPORTS

}  // extern "C"

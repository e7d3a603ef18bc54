// valu_calib - how many wave64 VALU instructions does a SIMD of this GPU issue per clock?  (what "valu_busy" is a fraction of.)
//
// The three hot kernels of this repo run with a VALU instruction in 0.72-0.81 of a SIMD's 4-clock turns (SQ_ACTIVE_INST_VALU) and at
// 0.25-0.35 of the guide's VALU peak (1024 SIMDs x 32 lanes x 2.4 GHz).  Which of the two is the roof they are near depends on what one
// SIMD can issue: this program times long runs of INDEPENDENT instructions of the kinds those kernels are made of - v_fma_f32,
// v_pk_fma_f32, v_min_f32 / v_cndmask_b32 / v_add_u32, v_max3_f32 - at 1, 2, 4, 5 and 8 waves per SIMD on every CU and prints
// wave-instructions per SIMD per shader clock (s_memtime) for each.  No memory traffic inside the timed loop.
//
// build: make calib      run: jaderaytracerendering_amd/lib/valu_calib   (one JSON line per kernel and occupancy)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                       \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                          \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

#define REP16(S) S S S S S S S S S S S S S S S S
enum { K_FMA, K_PK_FMA, K_MIX, K_MAX3, K_N };
static const char* const kNames[K_N] = {"v_fma_f32", "v_pk_fma_f32", "v_min_f32+v_cndmask_b32+v_add_u32+v_max_f32", "v_max3_f32"};

// 16 instructions per block of asm, 8 independent accumulators: nothing waits for a result it needs
template <int KIND>
__global__ __launch_bounds__(256) void k_valu(int iters, float seed, float* sink, unsigned long long* clocks) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * 0.5f, b3 = a3 * 0.5f, b4 = a4 * 0.5f, b5 = a5 * 0.5f, b6 = a6 * 0.5f, b7 = a7 * 0.5f;
  const float m = 0.999f, c = 0.001f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (KIND == K_FMA) {
      asm volatile(
          "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
          "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
          "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
          "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
          : "v"(m), "v"(c));
    } else if (KIND == K_PK_FMA) {
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {a0, b0}, p1 = {a1, b1}, p2 = {a2, b2}, p3 = {a3, b3}, p4 = {a4, b4}, p5 = {a5, b5}, p6 = {a6, b6}, p7 = {a7, b7};
      const f2 mm = {m, m}, cc = {c, c};
      asm volatile(
          "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
          "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
          "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
          "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
          : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
          : "v"(mm), "v"(cc));
      a0 = p0.x; b0 = p0.y; a1 = p1.x; b1 = p1.y; a2 = p2.x; b2 = p2.y; a3 = p3.x; b3 = p3.y;
      a4 = p4.x; b4 = p4.y; a5 = p5.x; b5 = p5.y; a6 = p6.x; b6 = p6.y; a7 = p7.x; b7 = p7.y;
    } else if (KIND == K_MIX) {  // the walk unit's kinds: min / max, select on vcc, integer add
      asm volatile(
          "v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %9\n v_cndmask_b32 %2, %2, %8, vcc\n v_add_u32 %3, %3, %9\n"
          "v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %9\n v_cndmask_b32 %6, %6, %8, vcc\n v_add_u32 %7, %7, %9\n"
          "v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %9\n v_cndmask_b32 %2, %2, %8, vcc\n v_add_u32 %3, %3, %9\n"
          "v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %9\n v_cndmask_b32 %6, %6, %8, vcc\n v_add_u32 %7, %7, %9\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
          : "v"(m), "v"(c)
          : "vcc");
    } else {
      asm volatile(
          "v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
          "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
          "v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
          "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
          : "v"(m), "v"(c));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
  if (s == 12345.678f) sink[0] = s;  // (keeps the chains alive)
  if (threadIdx.x == 0 && blockIdx.x == 0) clocks[0] = t1 - t0;
}

template <int KIND>
static int run(int waves_per_simd, int iters, float* sink, unsigned long long* clocks, int n_cu) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int blocks = n_cu * waves_per_simd;  // a 256-thread block = one wave on each of a CU's four SIMDs
  hipLaunchKernelGGL(k_valu<KIND>, dim3(blocks), dim3(256), 0, 0, iters / 8, 1.0f, sink, clocks);  // warm-up
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k_valu<KIND>, dim3(blocks), dim3(256), 0, 0, iters, 1.0f, sink, clocks);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long clk = 0;
  CHECK(hipMemcpy(&clk, clocks, 8, hipMemcpyDeviceToHost));
  const double insts_per_wave = 16.0 * iters;
  // clocks of ONE wave from its first to its last instruction: with w waves on its SIMD, the SIMD issued w x insts_per_wave in that time
  printf("{\"kernel\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"wave_clocks\": %llu, \"wave_insts_per_simd_per_clock\": %.4f, "
         "\"clock_GHz_seen\": %.3f, \"wave_insts_per_simd_per_ns\": %.4f}\n",
         kNames[KIND], waves_per_simd, ms, clk, waves_per_simd * insts_per_wave / (double)clk, (double)clk / (ms * 1e6),
         waves_per_simd * insts_per_wave / (ms * 1e6));
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  float* sink = nullptr;
  unsigned long long* clocks = nullptr;
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMalloc(&clocks, 64));
  printf("{\"device\": \"%s\", \"compute_units\": %d, \"clock_rate_kHz\": %d}\n", prop.name, n_cu, prop.clockRate);
  const int iters = 1 << 16;
  for (int w : {1, 2, 4, 5, 8}) {
    if (run<K_FMA>(w, iters, sink, clocks, n_cu)) return 1;
    if (run<K_PK_FMA>(w, iters, sink, clocks, n_cu)) return 1;
    if (run<K_MIX>(w, iters, sink, clocks, n_cu)) return 1;
    if (run<K_MAX3>(w, iters, sink, clocks, n_cu)) return 1;
  }
  return 0;
}

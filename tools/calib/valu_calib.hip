// valu_calib - how many wave64 VALU instructions does a SIMD of this GPU issue per clock?  (what "valu_busy" is a fraction of.)
//
// The three hot kernels of this repo run with a VALU instruction in 0.72-0.81 of a SIMD's 4-clock turns (SQ_ACTIVE_INST_VALU) and at
// 0.25-0.35 of the guide's VALU peak (1024 SIMDs x 32 lanes x 2.4 GHz).  Which of the two is the roof they are near depends on what one
// SIMD can issue: this program times long runs of INDEPENDENT instructions of the kinds those kernels are made of - v_fma_f32,
// v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, v_min_f32 / v_cndmask_b32 / v_add_u32, v_max3_f32, compare + select, integer address
// arithmetic, v_mov_b32, v_fma_f64 - at 1, 2, 5 and 8 waves per SIMD on every CU and prints
// wave-instructions per SIMD per nanosecond for each (wall clock: the shader clock under such a load is the GPU's business; run under
// rocprofv3 --pmc, tools/calib/run_valu_calib.sh relates the same launches to SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES, i.e. it gives
// the value of DESIGN's "valu_busy" for a kernel that does nothing but issue).  No memory traffic inside the timed loop.
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

enum { K_FMA, K_PK_FMA, K_MIX, K_MAX3, K_ADD, K_MUL, K_PK_MUL, K_PK_ADD, K_CMP_SEL, K_INT, K_MOV, K_FMA64, K_N };
static const char* const kNames[K_N] = {"v_fma_f32", "v_pk_fma_f32", "v_min_f32+v_cndmask_b32+v_add_u32+v_max_f32", "v_max3_f32", "v_add_f32", "v_mul_f32",
                                        "v_pk_mul_f32", "v_pk_add_f32", "v_cmp_lt_f32+v_cndmask_b32", "v_and_b32+v_lshlrev_b32+v_add_u32+v_xor_b32",
                                        "v_mov_b32", "v_fma_f64"};
typedef float f2 __attribute__((ext_vector_type(2)));

// one block of asm = 16 instructions over 8 independent accumulators (%0-%7; %8, %9 are constants): nothing waits for a result it needs
#define OP3(op) op " %0, %0, %8, %9\n " op " %1, %1, %8, %9\n " op " %2, %2, %8, %9\n " op " %3, %3, %8, %9\n " \
                op " %4, %4, %8, %9\n " op " %5, %5, %8, %9\n " op " %6, %6, %8, %9\n " op " %7, %7, %8, %9\n "
#define OP2(op) op " %0, %0, %8\n " op " %1, %1, %9\n " op " %2, %2, %8\n " op " %3, %3, %9\n " \
                op " %4, %4, %8\n " op " %5, %5, %9\n " op " %6, %6, %8\n " op " %7, %7, %9\n "
#define ACC8(v) "+v"(v##0), "+v"(v##1), "+v"(v##2), "+v"(v##3), "+v"(v##4), "+v"(v##5), "+v"(v##6), "+v"(v##7)

template <int KIND>
__global__ __launch_bounds__(256) void k_valu(int iters, float seed, float* sink, unsigned long long* clocks) {
  extern __shared__ float lds_pad[];  // sized by the host so that exactly `waves_per_simd` blocks fit a CU
  if (seed == 12345.0f) lds_pad[threadIdx.x] = seed;
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 p0 = {a0, a0 * 0.5f}, p1 = {a1, a1 * 0.5f}, p2 = {a2, a2 * 0.5f}, p3 = {a3, a3 * 0.5f}, p4 = {a4, a4 * 0.5f}, p5 = {a5, a5 * 0.5f}, p6 = {a6, a6 * 0.5f},
     p7 = {a7, a7 * 0.5f};
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
  const float m = 0.999f, c = 0.001f;
  const f2 mm = {m, m}, cc = {c, c};
  const double dm = 0.999, dc = 0.001;
  for (int i = 0; i < iters; ++i) {
    if (KIND == K_FMA) asm volatile(OP3("v_fma_f32") OP3("v_fma_f32") : ACC8(a) : "v"(m), "v"(c));
    else if (KIND == K_PK_FMA) asm volatile(OP3("v_pk_fma_f32") OP3("v_pk_fma_f32") : ACC8(p) : "v"(mm), "v"(cc));
    else if (KIND == K_MAX3) asm volatile(OP3("v_max3_f32") OP3("v_max3_f32") : ACC8(a) : "v"(m), "v"(c));
    else if (KIND == K_ADD) asm volatile(OP2("v_add_f32") OP2("v_add_f32") : ACC8(a) : "v"(m), "v"(c));
    else if (KIND == K_MUL) asm volatile(OP2("v_mul_f32") OP2("v_mul_f32") : ACC8(a) : "v"(m), "v"(c));
    else if (KIND == K_PK_MUL) asm volatile(OP2("v_pk_mul_f32") OP2("v_pk_mul_f32") : ACC8(p) : "v"(mm), "v"(cc));
    else if (KIND == K_PK_ADD) asm volatile(OP2("v_pk_add_f32") OP2("v_pk_add_f32") : ACC8(p) : "v"(mm), "v"(cc));
    else if (KIND == K_MOV) asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %9\n v_mov_b32 %2, %8\n v_mov_b32 %3, %9\n v_mov_b32 %4, %8\n v_mov_b32 %5, %9\n v_mov_b32 %6, %8\n v_mov_b32 %7, %9\n"
                                         "v_mov_b32 %0, %9\n v_mov_b32 %1, %8\n v_mov_b32 %2, %9\n v_mov_b32 %3, %8\n v_mov_b32 %4, %9\n v_mov_b32 %5, %8\n v_mov_b32 %6, %9\n v_mov_b32 %7, %8\n"
                                         : ACC8(a) : "v"(m), "v"(c));
    else if (KIND == K_FMA64) asm volatile(OP3("v_fma_f64") OP3("v_fma_f64") : ACC8(d) : "v"(dm), "v"(dc));
    else if (KIND == K_MIX)  // the walk unit's kinds: min / max, select on vcc, integer add
      asm volatile(
          "v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %9\n v_cndmask_b32 %2, %2, %8, vcc\n v_add_u32 %3, %3, %9\n"
          "v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %9\n v_cndmask_b32 %6, %6, %8, vcc\n v_add_u32 %7, %7, %9\n"
          "v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %9\n v_cndmask_b32 %2, %2, %8, vcc\n v_add_u32 %3, %3, %9\n"
          "v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %9\n v_cndmask_b32 %6, %6, %8, vcc\n v_add_u32 %7, %7, %9\n"
          : ACC8(a) : "v"(m), "v"(c) : "vcc");
    else if (KIND == K_CMP_SEL)  // a decision: compare into vcc, select on it (8 of each)
      asm volatile(
          "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
          "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc\n"
          "v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %2, %2, %9, vcc\n"
          "v_cmp_lt_f32 vcc, %5, %8\n v_cndmask_b32 %4, %4, %9, vcc\n v_cmp_lt_f32 vcc, %7, %8\n v_cndmask_b32 %6, %6, %9, vcc\n"
          : ACC8(a) : "v"(m), "v"(c) : "vcc");
    else  // K_INT: address arithmetic
      asm volatile(
          "v_and_b32 %0, %0, %8\n v_lshlrev_b32 %1, 1, %1\n v_add_u32 %2, %2, %9\n v_xor_b32 %3, %3, %8\n"
          "v_and_b32 %4, %4, %8\n v_lshlrev_b32 %5, 1, %5\n v_add_u32 %6, %6, %9\n v_xor_b32 %7, %7, %8\n"
          "v_and_b32 %0, %0, %8\n v_lshlrev_b32 %1, 1, %1\n v_add_u32 %2, %2, %9\n v_xor_b32 %3, %3, %8\n"
          "v_and_b32 %4, %4, %8\n v_lshlrev_b32 %5, 1, %5\n v_add_u32 %6, %6, %9\n v_xor_b32 %7, %7, %8\n"
          : ACC8(a) : "v"(m), "v"(c));
  }
  const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y +
                  p7.x + p7.y + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
  if (s == 12345.678f) sink[0] = s;  // (keeps the chains alive)
}

// Occupancy is set through LDS: a block asks for floor(160 KB / w) bytes, so w blocks (= w waves on each of the CU's four SIMDs) are
// resident per CU and no more; the grid holds 6 x that many blocks, so the figure is a steady state, not a launch transient.
template <int KIND>
static int run(int waves_per_simd, int iters, float* sink, unsigned long long* clocks, int n_cu) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int rounds = 6;
  const int blocks = n_cu * waves_per_simd * rounds;  // a 256-thread block = one wave on each of a CU's four SIMDs
  const size_t lds = (size_t)(160 * 1024 / waves_per_simd) & ~(size_t)1023;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_valu<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = 0;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_valu<KIND>, 256, lds));
  hipLaunchKernelGGL(k_valu<KIND>, dim3(blocks), dim3(256), lds, 0, iters / 8, 1.0f, sink, clocks);  // warm-up
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k_valu<KIND>, dim3(blocks), dim3(256), lds, 0, iters, 1.0f, sink, clocks);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double wave_insts = 16.0 * iters * 4.0 * blocks;  // over all waves
  const double per_simd_per_ns = wave_insts / (4.0 * n_cu) / (ms * 1e6);
  printf("{\"kernel\": \"%s\", \"waves_per_simd\": %d, \"blocks_per_cu_by_occupancy_query\": %d, \"ms\": %.3f, \"wave_insts_per_simd_per_ns\": %.4f, "
         "\"wave_insts_per_simd_per_clock_at_2.4GHz\": %.4f}\n",
         kNames[KIND], waves_per_simd, per_cu, ms, per_simd_per_ns, per_simd_per_ns / 2.4);
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
  const int iters = 1 << 14;
  for (int w : {1, 2, 5, 8}) {
    if (run<K_FMA>(w, iters, sink, clocks, n_cu) || run<K_PK_FMA>(w, iters, sink, clocks, n_cu) || run<K_MIX>(w, iters, sink, clocks, n_cu) ||
        run<K_MAX3>(w, iters, sink, clocks, n_cu) || run<K_ADD>(w, iters, sink, clocks, n_cu) || run<K_MUL>(w, iters, sink, clocks, n_cu) ||
        run<K_PK_MUL>(w, iters, sink, clocks, n_cu) || run<K_PK_ADD>(w, iters, sink, clocks, n_cu) || run<K_CMP_SEL>(w, iters, sink, clocks, n_cu) ||
        run<K_INT>(w, iters, sink, clocks, n_cu) || run<K_MOV>(w, iters, sink, clocks, n_cu) || run<K_FMA64>(w, iters, sink, clocks, n_cu))
      return 1;
  }
  return 0;
}

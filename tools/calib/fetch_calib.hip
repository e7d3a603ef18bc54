// fetch_calib — what do FETCH_SIZE / TCC_* read for k_trace's access pattern?  
//
// MI355X_MICROARCH.md calibrates FETCH_SIZE for wide coalesced streaming reads only ("reports exactly 1/2 of the bytes";
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").  k_trace's pattern
// is one 64-B node record (four 16-B loads of one aligned line) or one 80-B pair record per lane, at unrelated addresses.
// This program issues exactly that from tables of 55 MB (C5's geometry: inside the 256 MB Infinity Cache, outside the
// 32 MB of L2) and 1 GiB (outside everything), plus a coalesced streaming read as the reference point, with KNOWN byte
// counts, and prints one JSON line per kernel with its own timing; run it under rocprofv3 --pmc (one counter group per
// pass) and tools/summarize_calib.py relates the counters to the bytes.
//
// build: make calib      run: jaderaytracerendering_amd/lib/fetch_calib
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

static __device__ __forceinline__ uint32_t wang(uint32_t s) {
  s = (s ^ 61u) ^ (s >> 16);
  s *= 9u;
  s ^= s >> 4;
  s *= 0x27d4eb2du;
  s ^= s >> 15;
  return s;
}

// every lane reads `reps` records of REC bytes (REC = 64: four float4 of one aligned line; 80: five float4, unaligned to
// lines like the pair records; 16: one float4) at hashed record indices
template <int REC>
__global__ __launch_bounds__(256) void k_gather(const float4* table, uint32_t n_rec, int reps, float* sink) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t s = wang(tid * 2654435761u + 12345u);
  float acc = 0.0f;
  for (int r = 0; r < reps; ++r) {
    s = wang(s + (uint32_t)r);
    const uint32_t rec = s % n_rec;
    const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(table) + (size_t)rec * REC);
#pragma unroll
    for (int k = 0; k < REC / 16; ++k) {
      const float4 v = p[k];
      acc += v.x + v.w;
    }
  }
  if (acc == 123.456f) sink[0] = acc;  // never true: keeps the loads
}

// the dependent form: the next record's index comes out of the record just read (a ray's chain of node visits)
__global__ __launch_bounds__(256) void k_chase64(const float4* table, uint32_t n_rec, int reps, float* sink) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t rec = wang(tid * 2654435761u + 777u) % n_rec;
  float acc = 0.0f;
  for (int r = 0; r < reps; ++r) {
    const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(table) + (size_t)rec * 64);
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    acc += a.x + b.y + c.z;
    rec = (__float_as_uint(d.w) + (uint32_t)r) % n_rec;  // d.w holds a hashed index (fill)
  }
  if (acc == 123.456f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void k_stream(const float4* table, size_t n_vec, float* sink) {
  float acc = 0.0f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = table[i];
    acc += v.x + v.w;
  }
  if (acc == 123.456f) sink[0] = acc;
}

__global__ void k_fill(float4* table, size_t n_vec) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t h = wang((uint32_t)i * 2246822519u + 99u);
    table[i] = make_float4(1.0f, 2.0f, 3.0f, __uint_as_float(h));
  }
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 64;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount * 8;  // 8 blocks of 256 threads per CU: 2048 lanes per CU in flight
  float* sink = nullptr;
  CHECK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const size_t sizes[2] = {(size_t)55 << 20, (size_t)1 << 30};
  const char* names[2] = {"55MB", "1GB"};
  for (int t = 0; t < 2; ++t) {
    float4* table = nullptr;
    CHECK(hipMalloc(&table, sizes[t]));
    const size_t n_vec = sizes[t] / 16;
    hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, 0, table, n_vec);
    CHECK(hipDeviceSynchronize());
    const size_t lanes = (size_t)blocks * 256;
    auto report = [&](const char* kernel, int rec, double useful_bytes, float ms, int launches) {
      printf("{\"kernel\": \"%s\", \"table\": \"%s\", \"table_bytes\": %zu, \"record_bytes\": %d, \"lanes\": %zu, \"reps\": %d, \"launches\": %d, "
             "\"useful_bytes_per_launch\": %.0f, \"ms_per_launch\": %.4f, \"useful_GBps\": %.1f}\n",
             kernel, names[t], sizes[t], rec, lanes, reps, launches, useful_bytes, ms / launches, useful_bytes * launches / (ms * 1e-3) / 1e9);
      fflush(stdout);
    };
    const int L = 4;  // launches per kernel: the first warms the caches (a 55 MB table then sits in the Infinity Cache)
    float ms = 0;
#define RUN(KERNEL, NAME, REC, ...)                                                          \
  hipLaunchKernelGGL(KERNEL, dim3(blocks), dim3(256), 0, 0, __VA_ARGS__);                    \
  CHECK(hipDeviceSynchronize());                                                             \
  CHECK(hipEventRecord(e0, 0));                                                              \
  for (int l = 0; l < L; ++l) hipLaunchKernelGGL(KERNEL, dim3(blocks), dim3(256), 0, 0, __VA_ARGS__); \
  CHECK(hipEventRecord(e1, 0));                                                              \
  CHECK(hipEventSynchronize(e1));                                                            \
  CHECK(hipEventElapsedTime(&ms, e0, e1));                                                   \
  report(NAME, REC, (double)lanes * reps * REC, ms, L);
    RUN(k_gather<64>, "k_gather64", 64, table, (uint32_t)(sizes[t] / 64), reps, sink)
    RUN(k_gather<80>, "k_gather80", 80, table, (uint32_t)(sizes[t] / 80), reps, sink)
    RUN(k_gather<16>, "k_gather16", 16, table, (uint32_t)(sizes[t] / 16), reps, sink)
    RUN(k_chase64, "k_chase64", 64, table, (uint32_t)(sizes[t] / 64), reps, sink)
#undef RUN
    {
      hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, 0, table, n_vec, sink);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0, 0));
      for (int l = 0; l < L; ++l) hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, 0, table, n_vec, sink);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf("{\"kernel\": \"k_stream\", \"table\": \"%s\", \"table_bytes\": %zu, \"record_bytes\": 16, \"lanes\": %zu, \"reps\": 0, \"launches\": %d, "
             "\"useful_bytes_per_launch\": %.0f, \"ms_per_launch\": %.4f, \"useful_GBps\": %.1f}\n",
             names[t], sizes[t], lanes, L, (double)sizes[t], ms / L, (double)sizes[t] * L / (ms * 1e-3) / 1e9);
      fflush(stdout);
    }
    CHECK(hipFree(table));
  }
  return 0;
}

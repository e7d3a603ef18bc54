// jade_bvh.hip — linear BVH construction on MI355X (include/jade_bvh.h).
//
// Pipeline (one HIP stream, everything device-resident between the two copies):
//   k_keys     centroid -> 30-bit Morton code, key = code << 32 | triangle index (unique keys)
//   rocPRIM    radix sort of the 64-bit keys (the one library call: a plain sort)
//   k_radix    Karras 2012: one thread per internal node finds its range and split from the
//              longest common prefixes of neighbouring keys (clz of xor); n - 1 internal nodes
//   k_fit      bottom-up boxes: one thread per triangle walks to the root, the second arrival at
//              a node (atomic counter, agent-scope fences either side) unions the children
//   k_flags    a subtree of <= leaf_size triangles becomes ONE leaf; an item is emitted iff its
//              parent is not collapsed; rocPRIM exclusive scan numbers the emitted items
//   k_emit     BVHNode_cu records: node 0 dummy, root 1, child 0 = none (PathTrace.cu:341-345)
#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include <cstring>
#include <string>
#include <vector>

#include "jade_bvh.h"

int jade_fail(int code, const std::string& msg);  // jade_hip.hip: sets jade_last_error()

namespace {

#define BVH_TRY(expr)                                                                                 \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess)                                                                             \
      return jade_fail(e_ == hipErrorOutOfMemory ? JADE_ERR_NOMEM : JADE_ERR_DEVICE,                  \
                       std::string(#expr) + ": " + hipGetErrorString(e_));                            \
  } while (0)

struct Buf {
  void* p = nullptr;
  ~Buf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 16); }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

__device__ __forceinline__ uint32_t expand10(uint32_t v) {
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

// verts: 9 floats per triangle (p1, p2, p3)
__global__ void k_keys(const float* verts, int n, float3 cmin, float3 cinv, unsigned long long* keys, float* lo, float* hi) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* v = verts + 9 * (size_t)i;
  float c[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float a = v[k], b = v[3 + k], d = v[6 + k];
    lo[3 * (size_t)i + k] = fminf(a, fminf(b, d));
    hi[3 * (size_t)i + k] = fmaxf(a, fmaxf(b, d));
    c[k] = (a + b + d) / 3.0f;  // the reference's sort key, PathTrace.cu:468-482
  }
  float fx = fminf(fmaxf((c[0] - cmin.x) * cinv.x, 0.0f), 1023.0f);
  float fy = fminf(fmaxf((c[1] - cmin.y) * cinv.y, 0.0f), 1023.0f);
  float fz = fminf(fmaxf((c[2] - cmin.z) * cinv.z, 0.0f), 1023.0f);
  uint32_t code = (expand10((uint32_t)fx) << 2) | (expand10((uint32_t)fy) << 1) | expand10((uint32_t)fz);
  keys[i] = ((unsigned long long)code << 32) | (unsigned)i;
}

__device__ __forceinline__ int delta(const unsigned long long* keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  return __clzll((long long)(keys[i] ^ keys[j]));  // keys are unique: never 64
}

// Items: internal node k -> k (0 .. n-2); sorted triangle k -> n - 1 + k.
__global__ void k_radix(const unsigned long long* keys, int n, int* left, int* right, int* first, int* last, int* parent) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  int j = i + l * d;
  int dnode = delta(keys, n, i, j);
  int s = 0, t = l;
  do {
    t = (t + 1) >> 1;
    if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
  } while (t > 1);
  int gamma = i + s * d + (d < 0 ? -1 : 0);
  int lo_ = i < j ? i : j, hi_ = i < j ? j : i;
  int L = (lo_ == gamma) ? (n - 1 + gamma) : gamma;
  int R = (hi_ == gamma + 1) ? (n - 1 + gamma + 1) : (gamma + 1);
  left[i] = L;
  right[i] = R;
  first[i] = lo_;
  last[i] = hi_;
  parent[L] = i;
  parent[R] = i;
  if (i == 0) parent[0] = -1;
}

// box arrays are indexed by item (2n - 1 entries of 3 floats each)
__global__ void k_fit(const unsigned long long* keys, int n, const int* left, const int* right, const int* parent,
                      const float* prim_lo, const float* prim_hi, float* blo, float* bhi, int* arrivals) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  int prim = (int)(keys[k] & 0xffffffffull);
  int item = n - 1 + k;
  for (int c = 0; c < 3; ++c) {
    blo[3 * (size_t)item + c] = prim_lo[3 * (size_t)prim + c];
    bhi[3 * (size_t)item + c] = prim_hi[3 * (size_t)prim + c];
  }
  int p = n > 1 ? parent[item] : -1;
  while (p >= 0) {
    __threadfence();  // release: this thread's child box is visible before the arrival is counted
    if (atomicAdd(&arrivals[p], 1) == 0) return;  // the sibling subtree is not done yet
    __threadfence();  // acquire: see the sibling's box
    int a = left[p], b = right[p];
    for (int c = 0; c < 3; ++c) {
      float la = __builtin_nontemporal_load(&blo[3 * (size_t)a + c]), lb = __builtin_nontemporal_load(&blo[3 * (size_t)b + c]);
      float ha = __builtin_nontemporal_load(&bhi[3 * (size_t)a + c]), hb = __builtin_nontemporal_load(&bhi[3 * (size_t)b + c]);
      blo[3 * (size_t)p + c] = fminf(la, lb);
      bhi[3 * (size_t)p + c] = fmaxf(ha, hb);
    }
    p = parent[p];
  }
}

__device__ __forceinline__ bool collapsed(const int* first, const int* last, int i, int leaf_size) {
  return last[i] - first[i] + 1 <= leaf_size;
}

__global__ void k_flags(int n, int leaf_size, const int* first, const int* last, const int* parent, int* flags) {
  int item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= 2 * n - 1) return;
  int f;
  if (n == 1) f = 1;  // the single triangle is the root leaf
  else if (item == 0) f = 1;
  else f = collapsed(first, last, parent[item], leaf_size) ? 0 : 1;
  flags[item] = f;
}

__global__ void k_emit(int n, int leaf_size, const int* left, const int* right, const int* first, const int* last,
                       const int* flags, const int* slot, const float* blo, const float* bhi, jade_bvh_node* nodes) {
  int item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= 2 * n - 1 || !flags[item]) return;
  jade_bvh_node nd;
  nd.left = nd.right = nd.n = nd.index = 0;
  if (item >= n - 1) {  // a single triangle
    nd.n = 1;
    nd.index = item - (n - 1);
  } else if (collapsed(first, last, item, leaf_size)) {
    nd.n = last[item] - first[item] + 1;
    nd.index = first[item];
  } else {
    nd.left = 1 + slot[left[item]];
    nd.right = 1 + slot[right[item]];
  }
  for (int c = 0; c < 3; ++c) {
    nd.aa[c] = blo[3 * (size_t)item + c];
    nd.bb[c] = bhi[3 * (size_t)item + c];
  }
  nodes[1 + slot[item]] = nd;
}

}  // namespace

extern "C" int jade_bvh_build_lbvh(const jade_triangle* tris, int32_t n, int32_t leaf_size, int device_id, int32_t* order_out,
                                   jade_bvh_node* nodes_out, int32_t max_nodes, int32_t* n_nodes_out, double* build_ms) {
  if (!tris || n <= 0 || !order_out || !nodes_out || !n_nodes_out) return jade_fail(JADE_ERR_INVALID, "null argument");
  if (leaf_size < 1 || leaf_size > 15) return jade_fail(JADE_ERR_INVALID, "leaf_size must be 1..15");
  if (n >= (1 << 27)) return jade_fail(JADE_ERR_UNSUPPORTED, "more than 2^27 triangles");
  int ndev = 0;
  BVH_TRY(hipGetDeviceCount(&ndev));
  if (device_id < 0 || device_id >= ndev) return jade_fail(JADE_ERR_DEVICE, "no such HIP device");
  BVH_TRY(hipSetDevice(device_id));

  // vertices only, and the centroid bounds (O(n) on the host, beside the packing loop)
  std::vector<float> verts((size_t)9 * n);
  float cmin[3] = {3e38f, 3e38f, 3e38f}, cmax[3] = {-3e38f, -3e38f, -3e38f};
  for (int i = 0; i < n; ++i) {
    const jade_triangle& t = tris[i];
    float* v = &verts[9 * (size_t)i];
    memcpy(v, t.p1, 12);
    memcpy(v + 3, t.p2, 12);
    memcpy(v + 6, t.p3, 12);
    for (int k = 0; k < 3; ++k) {
      float c = (v[k] + v[3 + k] + v[6 + k]) / 3.0f;
      cmin[k] = c < cmin[k] ? c : cmin[k];
      cmax[k] = c > cmax[k] ? c : cmax[k];
    }
  }
  float3 dmin = make_float3(cmin[0], cmin[1], cmin[2]), dinv;
  {
    float e[3];
    for (int k = 0; k < 3; ++k) e[k] = cmax[k] > cmin[k] ? 1023.999f / (cmax[k] - cmin[k]) : 0.0f;
    dinv = make_float3(e[0], e[1], e[2]);
  }

  const int items = 2 * n - 1;
  Buf b_verts, b_keys, b_keys2, b_lo, b_hi, b_left, b_right, b_first, b_last, b_parent, b_blo, b_bhi, b_arr, b_flags, b_slot,
      b_nodes, b_tmp;
  BVH_TRY(b_verts.alloc(verts.size() * 4));
  BVH_TRY(b_keys.alloc((size_t)n * 8));
  BVH_TRY(b_keys2.alloc((size_t)n * 8));
  BVH_TRY(b_lo.alloc((size_t)n * 12));
  BVH_TRY(b_hi.alloc((size_t)n * 12));
  BVH_TRY(b_left.alloc((size_t)n * 4));
  BVH_TRY(b_right.alloc((size_t)n * 4));
  BVH_TRY(b_first.alloc((size_t)n * 4));
  BVH_TRY(b_last.alloc((size_t)n * 4));
  BVH_TRY(b_parent.alloc((size_t)items * 4));
  BVH_TRY(b_blo.alloc((size_t)items * 12));
  BVH_TRY(b_bhi.alloc((size_t)items * 12));
  BVH_TRY(b_arr.alloc((size_t)n * 4));
  BVH_TRY(b_flags.alloc((size_t)items * 4));
  BVH_TRY(b_slot.alloc((size_t)items * 4));
  BVH_TRY(b_nodes.alloc((size_t)(items + 1) * sizeof(jade_bvh_node)));
  size_t tmp_sort = 0, tmp_scan = 0;
  BVH_TRY(rocprim::radix_sort_keys(nullptr, tmp_sort, b_keys.as<unsigned long long>(), b_keys2.as<unsigned long long>(), (size_t)n));
  BVH_TRY(rocprim::exclusive_scan(nullptr, tmp_scan, b_flags.as<int>(), b_slot.as<int>(), 0, (size_t)items, rocprim::plus<int>()));
  BVH_TRY(b_tmp.alloc(tmp_sort > tmp_scan ? tmp_sort : tmp_scan));

  hipStream_t st = nullptr;  // null stream: rocPRIM and the kernels below are ordered
  BVH_TRY(hipMemcpy(b_verts.p, verts.data(), verts.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  BVH_TRY(hipEventCreate(&e0));
  BVH_TRY(hipEventCreate(&e1));
  BVH_TRY(hipEventRecord(e0, st));
  const unsigned bn = (unsigned)((n + 255) / 256), bi = (unsigned)((items + 255) / 256);
  hipLaunchKernelGGL(k_keys, dim3(bn), dim3(256), 0, st, b_verts.as<float>(), n, dmin, dinv, b_keys.as<unsigned long long>(),
                     b_lo.as<float>(), b_hi.as<float>());
  BVH_TRY(rocprim::radix_sort_keys(b_tmp.p, tmp_sort, b_keys.as<unsigned long long>(), b_keys2.as<unsigned long long>(), (size_t)n, 0,
                                   64, st));
  const unsigned long long* keys = b_keys2.as<unsigned long long>();
  BVH_TRY(hipMemsetAsync(b_arr.p, 0, (size_t)n * 4, st));
  if (n > 1)
    hipLaunchKernelGGL(k_radix, dim3(bn), dim3(256), 0, st, keys, n, b_left.as<int>(), b_right.as<int>(), b_first.as<int>(),
                       b_last.as<int>(), b_parent.as<int>());
  hipLaunchKernelGGL(k_fit, dim3(bn), dim3(256), 0, st, keys, n, b_left.as<int>(), b_right.as<int>(), b_parent.as<int>(),
                     b_lo.as<float>(), b_hi.as<float>(), b_blo.as<float>(), b_bhi.as<float>(), b_arr.as<int>());
  hipLaunchKernelGGL(k_flags, dim3(bi), dim3(256), 0, st, n, leaf_size, b_first.as<int>(), b_last.as<int>(), b_parent.as<int>(),
                     b_flags.as<int>());
  BVH_TRY(rocprim::exclusive_scan(b_tmp.p, tmp_scan, b_flags.as<int>(), b_slot.as<int>(), 0, (size_t)items, rocprim::plus<int>(), st));
  BVH_TRY(hipMemsetAsync(b_nodes.p, 0, sizeof(jade_bvh_node), st));
  hipLaunchKernelGGL(k_emit, dim3(bi), dim3(256), 0, st, n, leaf_size, b_left.as<int>(), b_right.as<int>(), b_first.as<int>(),
                     b_last.as<int>(), b_flags.as<int>(), b_slot.as<int>(), b_blo.as<float>(), b_bhi.as<float>(),
                     b_nodes.as<jade_bvh_node>());
  BVH_TRY(hipGetLastError());
  BVH_TRY(hipEventRecord(e1, st));
  BVH_TRY(hipEventSynchronize(e1));
  float ms = 0;
  BVH_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (build_ms) *build_ms = ms;

  int last_flag = 0, last_slot = 0;
  BVH_TRY(hipMemcpy(&last_flag, b_flags.as<int>() + (items - 1), 4, hipMemcpyDeviceToHost));
  BVH_TRY(hipMemcpy(&last_slot, b_slot.as<int>() + (items - 1), 4, hipMemcpyDeviceToHost));
  const int emitted = last_slot + last_flag;
  if (1 + emitted > max_nodes) return jade_fail(JADE_ERR_INVALID, "nodes_out too small (2*n + 1 always suffices)");
  BVH_TRY(hipMemcpy(nodes_out, b_nodes.p, (size_t)(1 + emitted) * sizeof(jade_bvh_node), hipMemcpyDeviceToHost));
  // node 0: the reference's dummy record (PathTrace.cu:1557-1563)
  memset(&nodes_out[0], 0, sizeof(jade_bvh_node));
  nodes_out[0].left = 255; nodes_out[0].right = 128; nodes_out[0].n = 30;
  nodes_out[0].aa[0] = 1; nodes_out[0].aa[1] = 1; nodes_out[0].bb[1] = 1;
  *n_nodes_out = 1 + emitted;
  std::vector<unsigned long long> hk((size_t)n);
  BVH_TRY(hipMemcpy(hk.data(), keys, (size_t)n * 8, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; ++i) order_out[i] = (int32_t)(hk[i] & 0xffffffffull);
  return JADE_OK;
}

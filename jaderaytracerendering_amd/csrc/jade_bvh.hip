// jade_bvh.hip — BVH construction on MI355X (include/jade_bvh.h): a linear BVH and PLOC.
//
// LBVH pipeline (one HIP stream, everything device-resident between the two copies):
//   k_keys     centroid -> 30-bit Morton code, key = code << 32 | triangle index (unique keys)
//   rocPRIM    radix sort of the 64-bit keys (the one library call: a plain sort)
//   k_radix    Karras 2012: one thread per internal node finds its range and split from the
//              longest common prefixes of neighbouring keys (clz of xor); n - 1 internal nodes
//   k_fit      bottom-up boxes: one thread per triangle walks to the root, the second arrival at
//              a node (atomic counter, agent-scope fences either side) unions the children
//   k_flags    a subtree of <= leaf_size triangles becomes ONE leaf; an item is emitted iff its
//              parent is not collapsed; rocPRIM exclusive scan numbers the emitted items
//   k_emit     BVHNode_cu records: node 0 dummy, root 1, child 0 = none (PathTrace.cu:341-345)
//
// PLOC (parallel locally-ordered clustering, Meister & Bittner 2018) shares k_keys, the sort and the back end; between
// them, instead of the radix tree, clusters (initially the triangles in Morton order) are merged bottom-up: every
// cluster looks for the neighbour within PLOC_RADIUS positions whose union with it has the smallest surface area, and
// pairs that chose each other become one cluster (k_ploc_nn / k_ploc_mark / scan / k_ploc_apply, ~log n rounds).  The
// tree follows the surface-area heuristic the reference's sweep builder minimises top-down (PathTrace.cu:497-628)
// instead of the Morton code's bit boundaries, which is what makes the LBVH 1.75x more expensive to traverse.
#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include <algorithm>
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

// Back end, shared by both builders.  Every item (internal node k -> k, sorted triangle k -> n - 1 + k) carries the number
// of triangles below it and the position of the first of them in the final triangle order.
__global__ void k_flags(int n, int leaf_size, const int* count, const int* parent, int* flags) {
  int item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= 2 * n - 1) return;
  int f;
  if (n == 1) f = 1;  // the single triangle is the root leaf
  else if (item == 0) f = 1;
  else f = count[parent[item]] <= leaf_size ? 0 : 1;  // a subtree of <= leaf_size triangles is ONE leaf: nothing below it is emitted
  flags[item] = f;
}

__global__ void k_emit(int n, int leaf_size, const int* left, const int* right, const int* count, const int* offset,
                       const int* flags, const int* slot, const float* blo, const float* bhi, jade_bvh_node* nodes) {
  int item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= 2 * n - 1 || !flags[item]) return;
  jade_bvh_node nd;
  nd.left = nd.right = nd.n = nd.index = 0;
  if (item >= n - 1 || count[item] <= leaf_size) {  // a single triangle, or a collapsed subtree
    nd.n = count[item];
    nd.index = offset[item];
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

// LBVH: a node's triangles are the key range [first, last]
__global__ void k_lbvh_ranges(int n, const int* first, const int* last, int* count, int* offset) {
  int item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= 2 * n - 1) return;
  if (item >= n - 1) {
    count[item] = 1;
    offset[item] = item - (n - 1);
  } else {
    count[item] = last[item] - first[item] + 1;
    offset[item] = first[item];
  }
}

// final triangle order: sorted position -> original index
__global__ void k_order(const unsigned long long* keys, int n, const int* offset, int* order) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  order[offset[n - 1 + k]] = (int)(keys[k] & 0xffffffffull);
}

// ------------------------------------------------------------------------------------------------------- PLOC --
#define PLOC_RADIUS 16
#define PLOC_BLOCK 256

struct Box6 {
  float lo[3], hi[3];
};
__device__ __forceinline__ float union_area(const Box6& a, const Box6& b) {
  float x = fmaxf(a.hi[0], b.hi[0]) - fminf(a.lo[0], b.lo[0]);
  float y = fmaxf(a.hi[1], b.hi[1]) - fminf(a.lo[1], b.lo[1]);
  float z = fmaxf(a.hi[2], b.hi[2]) - fminf(a.lo[2], b.lo[2]);
  return x * y + y * z + z * x;
}

__global__ void k_ploc_init(const unsigned long long* keys, int n, const float* prim_lo, const float* prim_hi, float* blo, float* bhi,
                            int* count, int* cid, Box6* cbox) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int prim = (int)(keys[k] & 0xffffffffull), item = n - 1 + k;
  Box6 b;
  for (int c = 0; c < 3; ++c) {
    b.lo[c] = prim_lo[3 * (size_t)prim + c];
    b.hi[c] = prim_hi[3 * (size_t)prim + c];
    blo[3 * (size_t)item + c] = b.lo[c];
    bhi[3 * (size_t)item + c] = b.hi[c];
  }
  count[item] = 1;
  cid[k] = item;
  cbox[k] = b;
}

// nearest neighbour by surface area of the union, within PLOC_RADIUS positions.  The cost of a pair is symmetric and
// ties are broken by the pair's (low, high) positions, so "i chose j and j chose i" is well defined.
__global__ __launch_bounds__(PLOC_BLOCK) void k_ploc_nn(int m, const Box6* cbox, int* nn) {
  __shared__ Box6 tile[PLOC_BLOCK + 2 * PLOC_RADIUS];
  const int base = blockIdx.x * PLOC_BLOCK - PLOC_RADIUS;
  for (int t = threadIdx.x; t < PLOC_BLOCK + 2 * PLOC_RADIUS; t += PLOC_BLOCK) {
    const int g = base + t;
    if (g >= 0 && g < m) tile[t] = cbox[g];
  }
  __syncthreads();
  const int i = blockIdx.x * PLOC_BLOCK + threadIdx.x;
  if (i >= m) return;
  const Box6 me = tile[threadIdx.x + PLOC_RADIUS];
  float best = 3.0e38f;
  int bj = -1;
  for (int d = -PLOC_RADIUS; d <= PLOC_RADIUS; ++d) {
    const int j = i + d;
    if (d == 0 || j < 0 || j >= m) continue;
    const float a = union_area(me, tile[threadIdx.x + PLOC_RADIUS + d]);
    // Same area: a total order on PAIRS that both ends evaluate alike, so that the best pair overall is always mutual -
    // and one that cannot chain.  With "the pair with the lowest low end" every cluster of a row of equal areas (a ribbon,
    // instanced duplicates) chose its lower neighbour and ONE pair merged per round (600 identical triangles: 599 rounds).
    // Nearer in Morton order first; of the two neighbours at the same distance dd, the one whose pair starts at an even
    // multiple of dd - (2k, 2k+1), (4k, 4k+2), (4k+1, 4k+3), ...: disjoint pairs, so half of such a row merges per round.
    bool better = a < best;
    if (a == best && bj >= 0) {
      const int d1 = d < 0 ? -d : d, d0 = bj > i ? bj - i : i - bj;
      const int lo1 = i < j ? i : j, lo0 = i < bj ? i : bj;
      const int odd1 = (lo1 / d1) & 1, odd0 = (lo0 / d0) & 1;
      better = d1 < d0 || (d1 == d0 && (odd1 < odd0 || (odd1 == odd0 && lo1 < lo0)));
    }
    if (better) {
      best = a;
      bj = j;
    }
  }
  nn[i] = bj;  // -1 only for m == 1
}

// keep | leader << 32 per cluster: the lower partner of a mutual pair leads (stays, becomes the merged cluster), the
// upper one goes
__global__ void k_ploc_mark(int m, const int* nn, unsigned long long* flag) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int j = nn[i];
  const bool mutual = j >= 0 && nn[j] == i;
  const bool leader = mutual && i < j, removed = mutual && i > j;
  flag[i] = (removed ? 0ull : 1ull) | (leader ? 1ull << 32 : 0ull);
}

__global__ void k_ploc_apply(int m, int id_base, const int* nn, const unsigned long long* flag, const unsigned long long* pos,
                             const int* cid, const Box6* cbox, int* cid2, Box6* cbox2, int* left, int* right, int* parent,
                             float* blo, float* bhi, int* count) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const unsigned long long f = flag[i], p = pos[i];
  if (!(f & 1ull)) return;  // merged into its lower partner
  const int at = (int)(p & 0xffffffffull);
  if (f >> 32) {
    const int id = id_base - 1 - (int)(p >> 32);  // internal ids are handed out downwards: the last merge is item 0, the root
    const int j = nn[i], l = cid[i], r = cid[j];
    Box6 u;
    for (int c = 0; c < 3; ++c) {
      u.lo[c] = fminf(cbox[i].lo[c], cbox[j].lo[c]);
      u.hi[c] = fmaxf(cbox[i].hi[c], cbox[j].hi[c]);
      blo[3 * (size_t)id + c] = u.lo[c];
      bhi[3 * (size_t)id + c] = u.hi[c];
    }
    left[id] = l;
    right[id] = r;
    parent[l] = id;
    parent[r] = id;
    count[id] = count[l] + count[r];
    cid2[at] = id;
    cbox2[at] = u;
  } else {
    cid2[at] = cid[i];
    cbox2[at] = cbox[i];
  }
}

// position of an item's first triangle in depth-first order: the triangles of every left sibling on the way up
__global__ void k_ploc_offsets(int n, const int* left, const int* right, const int* parent, const int* count, int* offset) {
  int item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= 2 * n - 1) return;
  int off = 0, c = item;
  for (int p = parent[c]; p >= 0; c = p, p = parent[c])
    if (right[p] == c) off += count[left[p]];
  offset[item] = off;
}

}  // namespace

namespace {
enum BuildKind { KIND_LBVH = 0, KIND_PLOC = 1 };

struct Ev {  // destroyed on every return path
  hipEvent_t e = nullptr;
  ~Ev() { if (e) (void)hipEventDestroy(e); }
};

int build_bvh(BuildKind kind, const jade_triangle* tris, int32_t n, int32_t leaf_size, int device_id, int32_t* order_out,
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
      b_nodes, b_tmp, b_count, b_offset, b_order, b_cid[2], b_cbox[2], b_nn, b_f64, b_p64;
  BVH_TRY(b_verts.alloc(verts.size() * 4));
  BVH_TRY(b_keys.alloc((size_t)n * 8));
  BVH_TRY(b_keys2.alloc((size_t)n * 8));
  BVH_TRY(b_lo.alloc((size_t)n * 12));
  BVH_TRY(b_hi.alloc((size_t)n * 12));
  BVH_TRY(b_left.alloc((size_t)n * 4));
  BVH_TRY(b_right.alloc((size_t)n * 4));
  BVH_TRY(b_parent.alloc((size_t)items * 4));
  BVH_TRY(b_blo.alloc((size_t)items * 12));
  BVH_TRY(b_bhi.alloc((size_t)items * 12));
  BVH_TRY(b_count.alloc((size_t)items * 4));
  BVH_TRY(b_offset.alloc((size_t)items * 4));
  BVH_TRY(b_order.alloc((size_t)n * 4));
  BVH_TRY(b_flags.alloc((size_t)items * 4));
  BVH_TRY(b_slot.alloc((size_t)items * 4));
  BVH_TRY(b_nodes.alloc((size_t)(items + 1) * sizeof(jade_bvh_node)));
  size_t tmp_sort = 0, tmp_scan = 0, tmp_scan64 = 0;
  BVH_TRY(rocprim::radix_sort_keys(nullptr, tmp_sort, b_keys.as<unsigned long long>(), b_keys2.as<unsigned long long>(), (size_t)n));
  BVH_TRY(rocprim::exclusive_scan(nullptr, tmp_scan, b_flags.as<int>(), b_slot.as<int>(), 0, (size_t)items, rocprim::plus<int>()));
  if (kind == KIND_LBVH) {
    BVH_TRY(b_first.alloc((size_t)n * 4));
    BVH_TRY(b_last.alloc((size_t)n * 4));
    BVH_TRY(b_arr.alloc((size_t)n * 4));
  } else {
    for (int k = 0; k < 2; ++k) {
      BVH_TRY(b_cid[k].alloc((size_t)n * 4));
      BVH_TRY(b_cbox[k].alloc((size_t)n * sizeof(Box6)));
    }
    BVH_TRY(b_nn.alloc((size_t)n * 4));
    BVH_TRY(b_f64.alloc((size_t)n * 8));
    BVH_TRY(b_p64.alloc((size_t)n * 8));
    BVH_TRY(rocprim::exclusive_scan(nullptr, tmp_scan64, b_f64.as<unsigned long long>(), b_p64.as<unsigned long long>(), 0ull, (size_t)n,
                                    rocprim::plus<unsigned long long>()));
  }
  BVH_TRY(b_tmp.alloc(std::max(tmp_sort, std::max(tmp_scan, tmp_scan64))));

  hipStream_t st = nullptr;  // null stream: rocPRIM and the kernels below are ordered
  BVH_TRY(hipMemcpy(b_verts.p, verts.data(), verts.size() * 4, hipMemcpyHostToDevice));
  Ev e0, e1;
  BVH_TRY(hipEventCreate(&e0.e));
  BVH_TRY(hipEventCreate(&e1.e));
  BVH_TRY(hipEventRecord(e0.e, st));
  const unsigned bn = (unsigned)((n + 255) / 256), bi = (unsigned)((items + 255) / 256);
  hipLaunchKernelGGL(k_keys, dim3(bn), dim3(256), 0, st, b_verts.as<float>(), n, dmin, dinv, b_keys.as<unsigned long long>(),
                     b_lo.as<float>(), b_hi.as<float>());
  BVH_TRY(rocprim::radix_sort_keys(b_tmp.p, tmp_sort, b_keys.as<unsigned long long>(), b_keys2.as<unsigned long long>(), (size_t)n, 0,
                                   64, st));
  const unsigned long long* keys = b_keys2.as<unsigned long long>();
  if (kind == KIND_LBVH) {
    BVH_TRY(hipMemsetAsync(b_arr.p, 0, (size_t)n * 4, st));
    if (n > 1)
      hipLaunchKernelGGL(k_radix, dim3(bn), dim3(256), 0, st, keys, n, b_left.as<int>(), b_right.as<int>(), b_first.as<int>(),
                         b_last.as<int>(), b_parent.as<int>());
    hipLaunchKernelGGL(k_fit, dim3(bn), dim3(256), 0, st, keys, n, b_left.as<int>(), b_right.as<int>(), b_parent.as<int>(),
                       b_lo.as<float>(), b_hi.as<float>(), b_blo.as<float>(), b_bhi.as<float>(), b_arr.as<int>());
    hipLaunchKernelGGL(k_lbvh_ranges, dim3(bi), dim3(256), 0, st, n, b_first.as<int>(), b_last.as<int>(), b_count.as<int>(),
                       b_offset.as<int>());
  } else {
    BVH_TRY(hipMemsetAsync(b_parent.p, 0xff, (size_t)items * 4, st));  // -1: the root keeps it
    hipLaunchKernelGGL(k_ploc_init, dim3(bn), dim3(256), 0, st, keys, n, b_lo.as<float>(), b_hi.as<float>(), b_blo.as<float>(),
                       b_bhi.as<float>(), b_count.as<int>(), b_cid[0].as<int>(), b_cbox[0].as<Box6>());
    int m = n, id_base = n - 1, cur = 0;
    for (int round = 0; m > 1; ++round) {
      // every round merges at least one pair (the best pair overall is mutual); regular geometry merges half of a row of
      // equal areas per round (k_ploc_nn's tie rule), so a build is some tens of rounds - the bound is a guard, not a budget
      if (round > 2048) return jade_fail(JADE_ERR_DEVICE, "PLOC did not converge in 2048 rounds");
      const unsigned bm = (unsigned)((m + PLOC_BLOCK - 1) / PLOC_BLOCK);
      hipLaunchKernelGGL(k_ploc_nn, dim3(bm), dim3(PLOC_BLOCK), 0, st, m, b_cbox[cur].as<Box6>(), b_nn.as<int>());
      hipLaunchKernelGGL(k_ploc_mark, dim3(bm), dim3(PLOC_BLOCK), 0, st, m, b_nn.as<int>(), b_f64.as<unsigned long long>());
      BVH_TRY(rocprim::exclusive_scan(b_tmp.p, tmp_scan64, b_f64.as<unsigned long long>(), b_p64.as<unsigned long long>(), 0ull, (size_t)m,
                                      rocprim::plus<unsigned long long>(), st));
      hipLaunchKernelGGL(k_ploc_apply, dim3(bm), dim3(PLOC_BLOCK), 0, st, m, id_base, b_nn.as<int>(), b_f64.as<unsigned long long>(),
                         b_p64.as<unsigned long long>(), b_cid[cur].as<int>(), b_cbox[cur].as<Box6>(), b_cid[cur ^ 1].as<int>(),
                         b_cbox[cur ^ 1].as<Box6>(), b_left.as<int>(), b_right.as<int>(), b_parent.as<int>(), b_blo.as<float>(),
                         b_bhi.as<float>(), b_count.as<int>());
      unsigned long long lastf = 0, lastp = 0;  // totals of the round: clusters kept, pairs merged
      BVH_TRY(hipMemcpyAsync(&lastf, b_f64.as<unsigned long long>() + (m - 1), 8, hipMemcpyDeviceToHost, st));
      BVH_TRY(hipMemcpyAsync(&lastp, b_p64.as<unsigned long long>() + (m - 1), 8, hipMemcpyDeviceToHost, st));
      BVH_TRY(hipStreamSynchronize(st));
      const unsigned long long tot = lastf + lastp;
      const int kept = (int)(tot & 0xffffffffull), merged = (int)(tot >> 32);
      if (merged <= 0 || kept != m - merged) return jade_fail(JADE_ERR_DEVICE, "PLOC round merged nothing");
      m = kept;
      id_base -= merged;
      cur ^= 1;
    }
    if (n > 1 && id_base != 0) return jade_fail(JADE_ERR_DEVICE, "PLOC: merge count is not n - 1");
    hipLaunchKernelGGL(k_ploc_offsets, dim3(bi), dim3(256), 0, st, n, b_left.as<int>(), b_right.as<int>(), b_parent.as<int>(),
                       b_count.as<int>(), b_offset.as<int>());
  }
  hipLaunchKernelGGL(k_flags, dim3(bi), dim3(256), 0, st, n, leaf_size, b_count.as<int>(), b_parent.as<int>(), b_flags.as<int>());
  BVH_TRY(rocprim::exclusive_scan(b_tmp.p, tmp_scan, b_flags.as<int>(), b_slot.as<int>(), 0, (size_t)items, rocprim::plus<int>(), st));
  BVH_TRY(hipMemsetAsync(b_nodes.p, 0, sizeof(jade_bvh_node), st));
  hipLaunchKernelGGL(k_emit, dim3(bi), dim3(256), 0, st, n, leaf_size, b_left.as<int>(), b_right.as<int>(), b_count.as<int>(),
                     b_offset.as<int>(), b_flags.as<int>(), b_slot.as<int>(), b_blo.as<float>(), b_bhi.as<float>(),
                     b_nodes.as<jade_bvh_node>());
  hipLaunchKernelGGL(k_order, dim3(bn), dim3(256), 0, st, keys, n, b_offset.as<int>(), b_order.as<int>());
  BVH_TRY(hipGetLastError());
  BVH_TRY(hipEventRecord(e1.e, st));
  BVH_TRY(hipEventSynchronize(e1.e));
  float ms = 0;
  BVH_TRY(hipEventElapsedTime(&ms, e0.e, e1.e));
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
  BVH_TRY(hipMemcpy(order_out, b_order.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  return JADE_OK;
}
}  // namespace

extern "C" int jade_bvh_build_lbvh(const jade_triangle* tris, int32_t n, int32_t leaf_size, int device_id, int32_t* order_out,
                                   jade_bvh_node* nodes_out, int32_t max_nodes, int32_t* n_nodes_out, double* build_ms) {
  return build_bvh(KIND_LBVH, tris, n, leaf_size, device_id, order_out, nodes_out, max_nodes, n_nodes_out, build_ms);
}

extern "C" int jade_bvh_build_ploc(const jade_triangle* tris, int32_t n, int32_t leaf_size, int device_id, int32_t* order_out,
                                   jade_bvh_node* nodes_out, int32_t max_nodes, int32_t* n_nodes_out, double* build_ms) {
  return build_bvh(KIND_PLOC, tris, n, leaf_size, device_id, order_out, nodes_out, max_nodes, n_nodes_out, build_ms);
}

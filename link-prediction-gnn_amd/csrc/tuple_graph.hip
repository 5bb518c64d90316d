// 2-WL tuple-graph construction (a3) and the sample_block filters (a2), bit-exact incl. order.
//
// Reference: TwoWL/utils.py:36-45 get_ei2 (+ set_mul :13-19): tuple ids are column indices of
// edge = cat(pos_edge, pred_edge); for node i ascending, every (a, b) with pos_edge[1,a] == i and
// edge[0,b] == i, a-major / b-minor.  The reference does this with an O(N*P) Python loop of
// boolean masks; here: two key sorts (tuples by head node, tuples by tail node), an exclusive
// scan of in_i * out_i, and a fill kernel in which every output slot finds its node by binary
// search -- a segmented cartesian product.
// sample_block (TwoWL/utils.py:48-68): mark blocked ids, build keep flags, stable compaction.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "lpg_common.hpp"

namespace {
using namespace lpg;

inline unsigned bits_for(int64_t n) {
  unsigned b = 1;
  while ((int64_t(1) << b) <= n) ++b;
  return b;
}

struct Ei2Ws {
  uint64_t *in_keys, *in_sorted, *out_keys, *out_sorted;
  int64_t *in_ptr, *out_ptr, *cnt, *node_off;
  void* tmp;
  size_t tmp_bytes;
  int64_t total;
};

Ei2Ws carve(void* ws, int64_t n, int64_t E1, int64_t P) {
  Ei2Ws w;
  char* p = static_cast<char*>(ws);
  auto take = [&](int64_t bytes) { char* q = p; p += align_up(bytes > 0 ? bytes : 8, 256); return q; };
  w.in_keys = reinterpret_cast<uint64_t*>(take(E1 * 8));
  w.in_sorted = reinterpret_cast<uint64_t*>(take(E1 * 8));
  w.out_keys = reinterpret_cast<uint64_t*>(take(P * 8));
  w.out_sorted = reinterpret_cast<uint64_t*>(take(P * 8));
  w.in_ptr = reinterpret_cast<int64_t*>(take((n + 2) * 8));
  w.out_ptr = reinterpret_cast<int64_t*>(take((n + 2) * 8));
  w.cnt = reinterpret_cast<int64_t*>(take((n + 2) * 8));
  w.node_off = reinterpret_cast<int64_t*>(take((n + 2) * 8));
  size_t s1 = 0, s2 = 0;
  (void)rocprim::radix_sort_keys(nullptr, s1, static_cast<uint64_t*>(nullptr), static_cast<uint64_t*>(nullptr),
                                 static_cast<size_t>(P > 0 ? P : 1), 0u, 32 + bits_for(n), hipStream_t(0), false);
  (void)rocprim::exclusive_scan(nullptr, s2, static_cast<int64_t*>(nullptr), static_cast<int64_t*>(nullptr),
                                int64_t(0), static_cast<size_t>(n + 1), rocprim::plus<int64_t>(), hipStream_t(0), false);
  w.tmp_bytes = s1 > s2 ? s1 : s2;
  w.tmp = p;
  w.total = (p - static_cast<char*>(ws)) + static_cast<int64_t>(align_up(w.tmp_bytes, 256));
  return w;
}

// key = (node << 32 | tuple id); node = tail (row 1) of a positive tuple for the "in" list,
// head (row 0) of any tuple for the "out" list.  Out-of-range nodes go to bucket n (dropped).
__global__ __launch_bounds__(kBlock) void ei2_keys_kernel(const int64_t* __restrict__ pos_edge, int64_t E1,
                                                          const int64_t* __restrict__ pred_edge, int64_t Ep,
                                                          int64_t n, uint64_t* __restrict__ in_keys,
                                                          uint64_t* __restrict__ out_keys, int32_t* err_flag) {
  const int64_t P = E1 + Ep;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < P; t += stride) {
    int64_t head = t < E1 ? pos_edge[t] : pred_edge[t - E1];
    if (head < 0 || head >= n) {
      flag_error(err_flag, LPG_F_INDEX_RANGE);
      head = n;
    }
    out_keys[t] = (static_cast<uint64_t>(head) << 32) | static_cast<uint64_t>(t);
    if (t < E1) {
      int64_t tail = pos_edge[E1 + t];
      if (tail < 0 || tail >= n) {
        flag_error(err_flag, LPG_F_INDEX_RANGE);
        tail = n;
      }
      in_keys[t] = (static_cast<uint64_t>(tail) << 32) | static_cast<uint64_t>(t);
    }
  }
}

__global__ __launch_bounds__(kBlock) void ptr_from_sorted_kernel(const uint64_t* __restrict__ keys, int64_t M,
                                                                 int64_t n, int64_t* __restrict__ ptr) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t k = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; k <= M; k += stride) {
    const int64_t a = (k == 0) ? -1 : static_cast<int64_t>(keys[k - 1] >> 32);
    int64_t b = (k == M) ? n : static_cast<int64_t>(keys[k] >> 32);
    if (b > n) b = n;
    for (int64_t r = a + 1; r <= b; ++r) ptr[r] = k;
  }
}

__global__ __launch_bounds__(kBlock) void ei2_count_kernel(const int64_t* __restrict__ in_ptr,
                                                           const int64_t* __restrict__ out_ptr, int64_t n,
                                                           int64_t* __restrict__ cnt) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i <= n; i += stride)
    cnt[i] = i < n ? (in_ptr[i + 1] - in_ptr[i]) * (out_ptr[i + 1] - out_ptr[i]) : 0;
}

__global__ __launch_bounds__(kBlock) void ei2_fill_kernel(const uint64_t* __restrict__ in_sorted,
                                                          const uint64_t* __restrict__ out_sorted,
                                                          const int64_t* __restrict__ in_ptr,
                                                          const int64_t* __restrict__ out_ptr,
                                                          const int64_t* __restrict__ node_off, int64_t n, int64_t E2,
                                                          int64_t* __restrict__ ei2) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t p = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; p < E2; p += stride) {
    int64_t lo = 0, hi = n;  // largest i with node_off[i] <= p  (node_off is non-decreasing)
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (node_off[mid] <= p) lo = mid; else hi = mid;
    }
    const int64_t i = lo;
    const int64_t local = p - node_off[i];
    const int64_t od = out_ptr[i + 1] - out_ptr[i];
    const int64_t ia = local / od, ib = local - ia * od;
    ei2[p] = static_cast<int64_t>(in_sorted[in_ptr[i] + ia] & 0xffffffffull);
    ei2[E2 + p] = static_cast<int64_t>(out_sorted[out_ptr[i] + ib] & 0xffffffffull);
  }
}

// ---- sample_block pieces ------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void mark_kernel(const int64_t* __restrict__ idx, int64_t n_idx,
                                                      uint8_t* __restrict__ mask, int64_t n_mask, int32_t* err_flag) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n_idx; i += stride) {
    const int64_t k = idx[i];
    if (k >= 0 && k < n_mask) mask[k] = 1; else flag_error(err_flag, LPG_F_INDEX_RANGE);
  }
}

__global__ __launch_bounds__(kBlock) void keep_kernel(const int64_t* __restrict__ key, int64_t E,
                                                      const uint8_t* __restrict__ blocked, int64_t n_mask, int invert,
                                                      uint8_t* __restrict__ keep, int32_t* err_flag) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < E; e += stride) {
    const int64_t k = key ? key[e] : e;
    uint8_t b = 0;
    if (k >= 0 && k < n_mask) b = blocked[k]; else flag_error(err_flag, LPG_F_INDEX_RANGE);
    keep[e] = invert ? b : static_cast<uint8_t>(!b);
  }
}

struct KeepToI64 {
  __host__ __device__ int64_t operator()(uint8_t k) const { return k ? 1 : 0; }
};

__global__ __launch_bounds__(kBlock) void compact_kernel(const int64_t* __restrict__ in, int64_t rows, int64_t E,
                                                         const uint8_t* __restrict__ keep,
                                                         const int64_t* __restrict__ pos, int64_t count,
                                                         int64_t* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < E; e += stride)
    if (keep[e]) {
      const int64_t q = pos[e];
      for (int64_t r = 0; r < rows; ++r) out[r * count + q] = in[r * E + e];
    }
}
}  // namespace

LPG_EXPORT int64_t lpg_get_ei2_bytes(int64_t n_node, int64_t E1, int64_t Ep) {
  if (n_node < 0 || E1 < 0 || Ep < 0) return LPG_E_INVALID;
  return carve(nullptr, n_node, E1, E1 + Ep).total;
}

LPG_EXPORT int lpg_get_ei2_count(const int64_t* pos_edge, int64_t E1, const int64_t* pred_edge, int64_t Ep,
                                 int64_t n_node, void* workspace, int64_t workspace_bytes, int64_t* e2_host,
                                 int32_t* err_flag, void* stream) {
  LPG_REQUIRE(n_node >= 0 && E1 >= 0 && Ep >= 0 && e2_host, "get_ei2_count: bad argument");
  const int64_t P = E1 + Ep;
  LPG_REQUIRE(P < (int64_t(1) << 31) && n_node < (int64_t(1) << 31), "get_ei2_count: sizes exceed int32");
  LPG_REQUIRE((E1 == 0 || pos_edge) && (Ep == 0 || pred_edge), "get_ei2_count: null pointer");
  Ei2Ws w = carve(workspace, n_node, E1, P);
  if (!workspace || workspace_bytes < w.total) {
    set_error("get_ei2_count: workspace too small");
    return LPG_E_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  const unsigned end_bit = 32 + bits_for(n_node);
  if (P > 0) {
    hipLaunchKernelGGL(ei2_keys_kernel, dim3(grid_for(P)), dim3(kBlock), 0, st, pos_edge, E1, pred_edge, Ep, n_node,
                       w.in_keys, w.out_keys, err_flag);
    size_t tb = w.tmp_bytes;
    LPG_HIP(rocprim::radix_sort_keys(w.tmp, tb, w.out_keys, w.out_sorted, static_cast<size_t>(P), 0u, end_bit, st, false));
    if (E1 > 0) {
      tb = w.tmp_bytes;
      LPG_HIP(rocprim::radix_sort_keys(w.tmp, tb, w.in_keys, w.in_sorted, static_cast<size_t>(E1), 0u, end_bit, st, false));
    }
  }
  hipLaunchKernelGGL(ptr_from_sorted_kernel, dim3(grid_for(E1 + 1)), dim3(kBlock), 0, st, w.in_sorted, E1, n_node, w.in_ptr);
  hipLaunchKernelGGL(ptr_from_sorted_kernel, dim3(grid_for(P + 1)), dim3(kBlock), 0, st, w.out_sorted, P, n_node, w.out_ptr);
  hipLaunchKernelGGL(ei2_count_kernel, dim3(grid_for(n_node + 1)), dim3(kBlock), 0, st, w.in_ptr, w.out_ptr, n_node, w.cnt);
  size_t tb = w.tmp_bytes;
  LPG_HIP(rocprim::exclusive_scan(w.tmp, tb, w.cnt, w.node_off, int64_t(0), static_cast<size_t>(n_node + 1),
                                  rocprim::plus<int64_t>(), st, false));
  LPG_HIP(hipMemcpyAsync(e2_host, w.node_off + n_node, 8, hipMemcpyDeviceToHost, st));
  LPG_HIP(hipStreamSynchronize(st));
  return check_launch("get_ei2_count");
}

LPG_EXPORT int lpg_get_ei2_fill(const void* workspace, int64_t n_node, int64_t E1, int64_t Ep, int64_t E2,
                                int64_t* ei2, void* stream) {
  LPG_REQUIRE(workspace && n_node >= 0 && E2 >= 0, "get_ei2_fill: bad argument");
  if (E2 == 0) return LPG_OK;
  LPG_REQUIRE(ei2, "get_ei2_fill: null output");
  Ei2Ws w = carve(const_cast<void*>(workspace), n_node, E1, E1 + Ep);
  hipLaunchKernelGGL(ei2_fill_kernel, dim3(grid_for(E2)), dim3(kBlock), 0, as_stream(stream), w.in_sorted,
                     w.out_sorted, w.in_ptr, w.out_ptr, w.node_off, n_node, E2, ei2);
  return check_launch("get_ei2_fill");
}

LPG_EXPORT int lpg_mark_i64(const int64_t* idx, int64_t n_idx, uint8_t* mask, int64_t n_mask, int32_t* err_flag,
                            void* stream) {
  LPG_REQUIRE(n_idx >= 0 && n_mask >= 0, "mark: bad sizes");
  if (n_idx == 0) return LPG_OK;
  LPG_REQUIRE(idx && mask, "mark: null pointer");
  hipLaunchKernelGGL(mark_kernel, dim3(grid_for(n_idx)), dim3(kBlock), 0, as_stream(stream), idx, n_idx, mask, n_mask, err_flag);
  return check_launch("mark");
}

LPG_EXPORT int lpg_keep_from_blocked(const int64_t* key, int64_t E, const uint8_t* blocked, int64_t n_mask, int invert,
                                     uint8_t* keep, int32_t* err_flag, void* stream) {
  LPG_REQUIRE(E >= 0 && n_mask >= 0, "keep_from_blocked: bad sizes");
  if (E == 0) return LPG_OK;
  LPG_REQUIRE(blocked && keep, "keep_from_blocked: null pointer");
  hipLaunchKernelGGL(keep_kernel, dim3(grid_for(E)), dim3(kBlock), 0, as_stream(stream), key, E, blocked, n_mask, invert, keep, err_flag);
  return check_launch("keep_from_blocked");
}

LPG_EXPORT int64_t lpg_compact_cols_bytes(int64_t E) {
  if (E < 0) return LPG_E_INVALID;
  size_t s = 0;
  (void)rocprim::exclusive_scan(nullptr, s, rocprim::make_transform_iterator(static_cast<const uint8_t*>(nullptr), KeepToI64()),
                                static_cast<int64_t*>(nullptr), int64_t(0), static_cast<size_t>(E + 1),
                                rocprim::plus<int64_t>(), hipStream_t(0), false);
  return lpg::align_up((E + 2) * 8, 256) + lpg::align_up(static_cast<int64_t>(s), 256) + 256;
}

LPG_EXPORT int lpg_compact_cols_i64(const int64_t* in, int64_t rows, int64_t E, const uint8_t* keep, int64_t* out,
                                    void* workspace, int64_t workspace_bytes, int64_t* count_host, void* stream) {
  LPG_REQUIRE(rows >= 1 && E >= 0 && count_host, "compact_cols: bad argument");
  if (E == 0) {
    *count_host = 0;
    return LPG_OK;
  }
  LPG_REQUIRE(in && keep && out && workspace, "compact_cols: null pointer");
  if (workspace_bytes < lpg_compact_cols_bytes(E)) {
    set_error("compact_cols: workspace too small");
    return LPG_E_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  int64_t* pos = static_cast<int64_t*>(workspace);  // E+1 entries; keep[E] is never read (size E scan + total)
  void* tmp = static_cast<char*>(workspace) + align_up((E + 2) * 8, 256);
  size_t tb = static_cast<size_t>(workspace_bytes - align_up((E + 2) * 8, 256));
  // scan over E elements, then total = pos[E-1] + keep[E-1]
  LPG_HIP(rocprim::exclusive_scan(tmp, tb, rocprim::make_transform_iterator(keep, KeepToI64()), pos, int64_t(0),
                                  static_cast<size_t>(E), rocprim::plus<int64_t>(), st, false));
  int64_t last_pos = 0;
  uint8_t last_keep = 0;
  LPG_HIP(hipMemcpyAsync(&last_pos, pos + (E - 1), 8, hipMemcpyDeviceToHost, st));
  LPG_HIP(hipMemcpyAsync(&last_keep, keep + (E - 1), 1, hipMemcpyDeviceToHost, st));
  LPG_HIP(hipStreamSynchronize(st));
  const int64_t count = last_pos + (last_keep ? 1 : 0);
  *count_host = count;
  if (count > 0)
    hipLaunchKernelGGL(compact_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, in, rows, E, keep, pos, count, out);
  return check_launch("compact_cols");
}

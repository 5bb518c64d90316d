// In-degree histogram for BIG edge lists (K1 at scale): out[index[e]] += 1, int64, bit-exact.
// Reference: degree(ei, num_node) = scatter_add(ones_like(ei[1]), ei[1], dim_size=N)  (TwoWL/utils.py:8-10).
//
// scatter.hip counts with atomics: an LDS histogram for N <= 16 K, an LDS hash that absorbs hub destinations
// plus global 64-bit atomics otherwise.  The latter is bound by the L2's scattered-atomic rate (~10 G/s):
// 10 ms for |E| = 100 M, |V| = 5 M, against 0.13 ms for just reading the indices.  No amount of privatisation
// fixes that while the keys arrive in random order -- so for big lists the keys are SORTED first:
//   narrow   int64 index -> uint32 key (out-of-range -> flagged, sent to a sentinel key that is never counted)
//   sort     rocPRIM radix sort on the ceil(log2 N) significant bits only (3 passes of 8 bits at N = 5 M)
//   runs     every key's run in the sorted stream is [first, last]: run starts store `first`, run ends add
//            last + 1 - first to out -- one plain 64-bit add per distinct key, no atomics, any order.
// ~6 GB of streaming traffic instead of 100 M scattered atomics.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "lpg_common.hpp"

namespace {
using namespace lpg;

inline unsigned bits_for(int64_t n) {  // bits needed to represent values in [0, n]
  unsigned b = 1;
  while ((int64_t(1) << b) <= n) ++b;
  return b;
}

struct DegWs {
  uint32_t* keys_in;
  uint32_t* keys_out;
  int32_t* first;
  void* sort_tmp;
  size_t sort_tmp_bytes;
  int64_t total;
};

DegWs carve(void* ws, int64_t E, int64_t N) {
  DegWs w;
  char* p = static_cast<char*>(ws);
  w.keys_in = reinterpret_cast<uint32_t*>(p);
  p += align_up(E * 4, 256);
  w.keys_out = reinterpret_cast<uint32_t*>(p);
  p += align_up(E * 4, 256);
  w.first = reinterpret_cast<int32_t*>(p);
  p += align_up((N + 1) * 4, 256);
  w.sort_tmp = p;
  size_t bytes = 0;
  (void)rocprim::radix_sort_keys(nullptr, bytes, static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr),
                                 static_cast<size_t>(E > 0 ? E : 1), 0u, bits_for(N), hipStream_t(0), false);
  w.sort_tmp_bytes = bytes;
  w.total = (p - static_cast<char*>(ws)) + static_cast<int64_t>(align_up(bytes, 256));
  return w;
}

__global__ __launch_bounds__(kBlock) void narrow_kernel(const int64_t* __restrict__ index, int64_t E, int64_t N,
                                                        uint32_t* __restrict__ keys, int32_t* err_flag) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < E; e += stride) {
    const int64_t v = index[e];
    const bool ok = v >= 0 && v < N;
    if (!ok) flag_error(err_flag, LPG_F_INDEX_RANGE);
    keys[e] = ok ? static_cast<uint32_t>(v) : static_cast<uint32_t>(N);
  }
}

// run starts: first[key] = position.  One writer per key.
__global__ __launch_bounds__(kBlock) void run_first_kernel(const uint32_t* __restrict__ keys, int64_t E,
                                                           int32_t* __restrict__ first) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < E; e += stride) {
    const uint32_t k = keys[e];
    if (e == 0 || keys[e - 1] != k) first[k] = static_cast<int32_t>(e);
  }
}
// run ends: out[key] += length.  One writer per key; the sentinel key N (out-of-range indices) is skipped.
__global__ __launch_bounds__(kBlock) void run_count_kernel(const uint32_t* __restrict__ keys, int64_t E, int64_t N,
                                                           const int32_t* __restrict__ first,
                                                           int64_t* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < E; e += stride) {
    const uint32_t k = keys[e];
    if ((e + 1 == E || keys[e + 1] != k) && k < N) out[k] += e + 1 - first[k];
  }
}
}  // namespace

extern "C" int lpg_scatter_add_i64(const int64_t* src, const int64_t* index, int64_t E, int64_t d, int64_t* out,
                                   int64_t N, int32_t* err_flag, void* stream);

LPG_EXPORT int64_t lpg_degree_i64_bytes(int64_t E, int64_t N) {
  if (E < 0 || N < 0) return LPG_E_INVALID;
  if (E >= (int64_t(1) << 31) || N >= (int64_t(1) << 31)) return 0;      // falls back to the atomic kernels
  return carve(nullptr, E, N).total;
}

LPG_EXPORT int lpg_degree_i64(const int64_t* index, int64_t E, int64_t* out, int64_t N, void* workspace,
                              int64_t workspace_bytes, int32_t* err_flag, void* stream) {
  LPG_REQUIRE(E >= 0 && N >= 0, "degree_i64: bad sizes E=%lld N=%lld", (long long)E, (long long)N);
  if (E == 0) return LPG_OK;
  LPG_REQUIRE(index && out, "degree_i64: null pointer");
  // small lists, tiny N (LDS histogram), or no workspace: the atomic kernels of scatter.hip
  const bool sorted_path = E >= (int64_t(1) << 22) && N > 16384 && E < (int64_t(1) << 31) && N < (int64_t(1) << 31) &&
                           workspace;
  if (!sorted_path) return lpg_scatter_add_i64(nullptr, index, E, 1, out, N, err_flag, stream);
  DegWs w = carve(workspace, E, N);
  if (workspace_bytes < w.total) {
    set_error("degree_i64: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)w.total);
    return LPG_E_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(narrow_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, index, E, N, w.keys_in, err_flag);
  size_t tmp = w.sort_tmp_bytes;
  LPG_HIP(rocprim::radix_sort_keys(w.sort_tmp, tmp, w.keys_in, w.keys_out, static_cast<size_t>(E), 0u, bits_for(N), st,
                                   false));
  hipLaunchKernelGGL(run_first_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, w.keys_out, E, w.first);
  hipLaunchKernelGGL(run_count_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, w.keys_out, E, N, w.first, out);
  return check_launch("degree_i64");
}

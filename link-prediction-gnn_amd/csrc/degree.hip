// In-degree histogram for BIG edge lists (K1 at scale): out[index[e]] += 1, int64, bit-exact.
// Reference: degree(ei, num_node) = scatter_add(ones_like(ei[1]), ei[1], dim_size=N)  (TwoWL/utils.py:8-10).
//
// scatter.hip counts with atomics: an LDS histogram for N <= 16 K, an LDS hash that absorbs hub destinations
// plus global 64-bit atomics otherwise.  The latter is bound by the L2's scattered-atomic rate (~10 G/s):
// 10 ms for |E| = 100 M, |V| = 5 M, against 0.13 ms for just reading the indices.  No amount of privatisation
// fixes that while the keys arrive in random order -- so for big lists the keys are first brought into an order
// in which a workgroup's LDS can hold every counter it touches:
//   narrow     int64 index -> uint32 key (out-of-range -> flagged, sent to a sentinel key that is never counted)
//   partition  ONE radix pass (rocPRIM) over the bits above the low 15: keys of one 32 K-bin bucket become contiguous
//   bounds     where each bucket starts, and how many 256 K-key parts it is cut into (hub buckets: many)
//   count      a workgroup per part: 32 K counters in 128 KB of LDS (ds_add), written out as a partial histogram
//   fold       out[bin] += sum of the bucket's partial histograms -- plain adds, every bin has one owner
// ~2.5 GB of streaming traffic at |E| = 100 M instead of 100 M scattered atomics (or the ~6 GB of a full three-pass
// sort, which is what this file did first and still does when the buckets would be tiny: N > 32 M or E < 2 N).
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

// ---------------------------------------------------------------- partition + LDS histogram
constexpr int kLowBits = 15, kBins = 1 << kLowBits;        // counters per bucket: 128 KB of LDS
constexpr int64_t kPart = int64_t(1) << 18;                // keys per workgroup
constexpr int kHistBlock = 1024;
constexpr int64_t kMaxBuckets = 1024;                      // N <= 32 M; beyond, buckets get tiny: full sort instead

struct PartWs {
  uint32_t* keys_in;
  uint32_t* keys_out;
  int64_t* bstart;      // [nb + 1] first key of each bucket in the partitioned stream
  int32_t* blk_off;     // [nb + 1] first workgroup of each bucket
  uint32_t* partial;    // [max_parts][kBins]
  void* sort_tmp;
  size_t sort_tmp_bytes;
  int64_t nb, max_parts, total;
};

PartWs carve_part(void* ws, int64_t E, int64_t N) {
  PartWs w;
  w.nb = (N >> kLowBits) + 1;                              // the sentinel key N has a bucket too
  w.max_parts = ceil_div(E, kPart) + w.nb;                 // sum_b ceil(len_b / kPart) <= this
  char* p = static_cast<char*>(ws);
  w.keys_in = reinterpret_cast<uint32_t*>(p);
  p += align_up(E * 4, 256);
  w.keys_out = reinterpret_cast<uint32_t*>(p);
  p += align_up(E * 4, 256);
  w.bstart = reinterpret_cast<int64_t*>(p);
  p += align_up((w.nb + 1) * 8, 256);
  w.blk_off = reinterpret_cast<int32_t*>(p);
  p += align_up((w.nb + 1) * 4, 256);
  w.partial = reinterpret_cast<uint32_t*>(p);
  p += align_up(w.max_parts * kBins * 4, 256);
  w.sort_tmp = p;
  size_t bytes = 0;
  if (bits_for(N) > kLowBits)
    (void)rocprim::radix_sort_keys(nullptr, bytes, static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr),
                                   static_cast<size_t>(E > 0 ? E : 1), unsigned(kLowBits), bits_for(N), hipStream_t(0),
                                   false);
  w.sort_tmp_bytes = bytes;
  w.total = (p - static_cast<char*>(ws)) + static_cast<int64_t>(align_up(bytes, 256));
  return w;
}

inline bool use_partition(int64_t E, int64_t N) {
  return (N >> kLowBits) + 1 <= kMaxBuckets && E >= 2 * N;
}

// one workgroup: bstart[b] = first position whose bucket is >= b (binary search in the partitioned keys), then the
// exclusive scan of the parts per bucket
__global__ __launch_bounds__(kHistBlock) void part_bounds_kernel(const uint32_t* __restrict__ keys, int64_t E, int nb,
                                                                 int64_t* __restrict__ bstart,
                                                                 int32_t* __restrict__ blk_off) {
  __shared__ int32_t cnt[kMaxBuckets + 1];
  const int t = threadIdx.x;
  for (int b = t; b <= nb; b += kHistBlock) {
    int64_t lo = 0, hi = E;                                // first k with (keys[k] >> kLowBits) >= b
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (static_cast<int64_t>(keys[mid] >> kLowBits) < b) lo = mid + 1; else hi = mid;
    }
    bstart[b] = lo;
  }
  __syncthreads();
  for (int b = t; b < nb; b += kHistBlock) cnt[b] = static_cast<int32_t>((bstart[b + 1] - bstart[b] + kPart - 1) / kPart);
  __syncthreads();
  if (t == 0) {                                            // nb <= 1024: a serial scan is a few microseconds
    int32_t run = 0;
    for (int b = 0; b < nb; ++b) { const int32_t c = cnt[b]; blk_off[b] = run; run += c; }
    blk_off[nb] = run;
  }
}

__global__ __launch_bounds__(kHistBlock) void part_count_kernel(const uint32_t* __restrict__ keys,
                                                                const int64_t* __restrict__ bstart,
                                                                const int32_t* __restrict__ blk_off, int nb,
                                                                uint32_t* __restrict__ partial) {
  __shared__ uint32_t hist[kBins];
  const int t = threadIdx.x;
  const int blk = blockIdx.x;
  if (blk >= blk_off[nb]) return;                          // the grid is the upper bound (block-uniform exit)
  int lo = 0, hi = nb;                                     // last bucket whose first workgroup is <= blk
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (blk_off[mid] <= blk) lo = mid; else hi = mid;
  }
  const int64_t k0 = bstart[lo] + static_cast<int64_t>(blk - blk_off[lo]) * kPart;
  const int64_t k1 = k0 + kPart < bstart[lo + 1] ? k0 + kPart : bstart[lo + 1];
  uint4* h4 = reinterpret_cast<uint4*>(hist);
  for (int i = t; i < kBins / 4; i += kHistBlock) h4[i] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  // 16-byte loads from the first aligned key on; the ragged head and tail go one key at a time
  const int64_t a0 = (k0 + 3) & ~int64_t(3), a1 = k1 & ~int64_t(3);
  if (a0 < a1) {
    for (int64_t k = k0 + t; k < a0; k += kHistBlock) atomicAdd(&hist[keys[k] & (kBins - 1)], 1u);
    const uint4* k4 = reinterpret_cast<const uint4*>(keys);
    for (int64_t q = (a0 >> 2) + t; q < (a1 >> 2); q += kHistBlock) {
      const uint4 v = k4[q];
      atomicAdd(&hist[v.x & (kBins - 1)], 1u);
      atomicAdd(&hist[v.y & (kBins - 1)], 1u);
      atomicAdd(&hist[v.z & (kBins - 1)], 1u);
      atomicAdd(&hist[v.w & (kBins - 1)], 1u);
    }
    for (int64_t k = a1 + t; k < k1; k += kHistBlock) atomicAdd(&hist[keys[k] & (kBins - 1)], 1u);
  } else {
    for (int64_t k = k0 + t; k < k1; k += kHistBlock) atomicAdd(&hist[keys[k] & (kBins - 1)], 1u);
  }
  __syncthreads();
  uint4* out4 = reinterpret_cast<uint4*>(partial + static_cast<int64_t>(blk) * kBins);
  for (int i = t; i < kBins / 4; i += kHistBlock) out4[i] = h4[i];
}

// out[b * kBins + i] += sum over the bucket's parts; bins >= N (the sentinel's) are never written
__global__ __launch_bounds__(kBlock) void part_fold_kernel(const uint32_t* __restrict__ partial,
                                                           const int32_t* __restrict__ blk_off, int64_t N,
                                                           int64_t* __restrict__ out) {
  const int b = blockIdx.y;
  const int p0 = blk_off[b], p1 = blk_off[b + 1];
  if (p0 == p1) return;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < kBins; i += gridDim.x * kBlock) {
    const int64_t bin = (static_cast<int64_t>(b) << kLowBits) + i;
    if (bin >= N) break;
    int64_t sum = 0;
    for (int p = p0; p < p1; ++p) sum += partial[static_cast<int64_t>(p) * kBins + i];
    if (sum) out[bin] += sum;
  }
}
}  // namespace

extern "C" int lpg_scatter_add_i64(const int64_t* src, const int64_t* index, int64_t E, int64_t d, int64_t* out,
                                   int64_t N, int32_t* err_flag, void* stream);

LPG_EXPORT int64_t lpg_degree_i64_bytes(int64_t E, int64_t N) {
  if (E < 0 || N < 0) return LPG_E_INVALID;
  if (E >= (int64_t(1) << 31) || N >= (int64_t(1) << 31)) return 0;      // falls back to the atomic kernels
  return use_partition(E, N) ? carve_part(nullptr, E, N).total : carve(nullptr, E, N).total;
}

LPG_EXPORT int lpg_degree_i64(const int64_t* index, int64_t E, int64_t* out, int64_t N, void* workspace,
                              int64_t workspace_bytes, int32_t* err_flag, void* stream) {
  LPG_REQUIRE(E >= 0 && N >= 0, "degree_i64: bad sizes E=%lld N=%lld", (long long)E, (long long)N);
  if (E == 0) return LPG_OK;
  LPG_REQUIRE(index && out, "degree_i64: null pointer");
  // small lists, tiny N (LDS histogram), or no workspace: the atomic kernels of scatter.hip
  const bool sorted_path = E >= (int64_t(1) << 21) && N > 16384 && E < (int64_t(1) << 31) && N < (int64_t(1) << 31) &&
                           workspace;
  if (!sorted_path) return lpg_scatter_add_i64(nullptr, index, E, 1, out, N, err_flag, stream);
  hipStream_t st = as_stream(stream);
  if (use_partition(E, N)) {
    PartWs w = carve_part(workspace, E, N);
    if (workspace_bytes < w.total) {
      set_error("degree_i64: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)w.total);
      return LPG_E_WORKSPACE;
    }
    hipLaunchKernelGGL(narrow_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, index, E, N, w.keys_in, err_flag);
    const uint32_t* keys = w.keys_in;
    if (bits_for(N) > kLowBits) {
      size_t tmp = w.sort_tmp_bytes;
      LPG_HIP(rocprim::radix_sort_keys(w.sort_tmp, tmp, w.keys_in, w.keys_out, static_cast<size_t>(E),
                                       unsigned(kLowBits), bits_for(N), st, false));
      keys = w.keys_out;
    }
    const int nb = static_cast<int>(w.nb);
    hipLaunchKernelGGL(part_bounds_kernel, dim3(1), dim3(kHistBlock), 0, st, keys, E, nb, w.bstart, w.blk_off);
    hipLaunchKernelGGL(part_count_kernel, dim3(static_cast<unsigned>(w.max_parts)), dim3(kHistBlock), 0, st, keys,
                       w.bstart, w.blk_off, nb, w.partial);
    hipLaunchKernelGGL(part_fold_kernel, dim3(kBins / kBlock / 4, static_cast<unsigned>(nb)), dim3(kBlock), 0, st,
                       w.partial, w.blk_off, N, out);
    return check_launch("degree_i64");
  }
  DegWs w = carve(workspace, E, N);
  if (workspace_bytes < w.total) {
    set_error("degree_i64: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)w.total);
    return LPG_E_WORKSPACE;
  }
  hipLaunchKernelGGL(narrow_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, index, E, N, w.keys_in, err_flag);
  size_t tmp = w.sort_tmp_bytes;
  LPG_HIP(rocprim::radix_sort_keys(w.sort_tmp, tmp, w.keys_in, w.keys_out, static_cast<size_t>(E), 0u, bits_for(N), st,
                                   false));
  hipLaunchKernelGGL(run_first_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, w.keys_out, E, w.first);
  hipLaunchKernelGGL(run_count_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, w.keys_out, E, N, w.first, out);
  return check_launch("degree_i64");
}

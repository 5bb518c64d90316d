// In-degree histogram for BIG edge lists (K1 at scale): out[index[e]] += 1, int64, bit-exact.
// Reference: degree(ei, num_node) = scatter_add(ones_like(ei[1]), ei[1], dim_size=N)  (TwoWL/utils.py:8-10).
//
// scatter.hip counts with atomics: an LDS histogram for N <= 16 K, an LDS hash that absorbs hub destinations
// plus global 64-bit atomics otherwise.  The latter is bound by the L2's scattered-atomic rate (~10 G/s):
// 10 ms for |E| = 100 M, |V| = 5 M, against 0.13 ms for just reading the indices.  No amount of privatisation
// fixes that while the keys arrive in random order -- so for big lists the keys are first brought into an order
// in which a workgroup's LDS can hold every counter it touches.  The int64 indices are read ONCE:
//   partition  a workgroup takes 16 K consecutive indices, range-checks and narrows them in registers, and groups them
//              by BUCKET (the bits above the low 15: 32 K bins per bucket) inside its own chunk -- counting sort in
//              LDS -- writing only the low 16 bits of every key (2 B instead of 8) plus the chunk's bucket offsets
//   totals     keys per bucket over all chunks, and from them how many ~256 K-key parts a bucket is cut into
//   count      a workgroup per (bucket, part): walks that bucket's segment of every chunk of its part, 32 K counters in
//              128 KB of LDS (ds_add), written out as a partial histogram
//   fold       out[bin] += sum of the bucket's partial histograms -- plain adds, every bin has one owner
// 0.8 GB read + 0.2 GB written + 0.2 GB read at |E| = 100 M (round 1: narrow to uint32, one rocPRIM radix pass over
// the high bits, count: 2.8 GB).  N > 32 M or E < 2 N (tiny buckets): full sort of the narrowed keys + run lengths.
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

// `ones` (may be null): the source vector of scatter_add(src, index) that the caller BELIEVES is all ones -- the
// reference's degree() spells the histogram that way (TwoWL/utils.py:8-10); it is read beside the index and any other
// value sets *not_ones (the caller then falls back to the general kernel), instead of a separate pass and host sync.
__global__ __launch_bounds__(kBlock) void narrow_kernel(const int64_t* __restrict__ index, int64_t E, int64_t N,
                                                        uint32_t* __restrict__ keys, int32_t* err_flag,
                                                        const int64_t* __restrict__ ones, int32_t* not_ones) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  bool odd = false;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < E; e += stride) {
    const int64_t v = index[e];
    if (ones) odd |= ones[e] != 1;
    const bool ok = v >= 0 && v < N;
    if (!ok) flag_error(err_flag, LPG_F_INDEX_RANGE);
    keys[e] = ok ? static_cast<uint32_t>(v) : static_cast<uint32_t>(N);
  }
  if (odd) atomicOr(not_ones, 1);
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

// ---------------------------------------------------------------- chunk-local partition + LDS histogram
constexpr int kLowBits = 15, kBins = 1 << kLowBits;        // counters per bucket: 128 KB of LDS
constexpr int64_t kPart = int64_t(1) << 18;                // keys per counting workgroup (target)
constexpr int kHistBlock = 1024;
constexpr int64_t kMaxBuckets = 1024;                      // N <= 32 M; beyond, buckets get tiny: full sort instead
constexpr int kChunk = 1 << 14;                            // indices per partition workgroup: 16 per thread, in registers
constexpr int kKeysPerThread = kChunk / kHistBlock;

struct PartWs {
  uint16_t* keys16;     // [nchunks][kChunk] low 16 bits of the keys, grouped by bucket inside every chunk
  uint16_t* tab;        // [nchunks][nb + 1] where each bucket starts inside its chunk; [nb] = valid keys of the chunk
  int64_t* totals;      // [nb] keys per bucket
  int32_t* blk_off;     // [nb + 1] first counting workgroup of each bucket
  int32_t* gchunks;     // [nb] chunks per part of each bucket
  uint32_t* partial;    // [max_parts][kBins]
  int64_t nb, nchunks, max_parts, total;
};

PartWs carve_part(void* ws, int64_t E, int64_t N) {
  PartWs w;
  w.nb = (N >> kLowBits) + 1;                              // the sentinel key N has a bucket too
  w.nchunks = ceil_div(E, kChunk);
  w.max_parts = ceil_div(E, kPart) + w.nb;                 // sum_b ceil(len_b / kPart) <= this
  char* p = static_cast<char*>(ws);
  w.keys16 = reinterpret_cast<uint16_t*>(p);
  p += align_up(w.nchunks * kChunk * 2, 256);
  w.tab = reinterpret_cast<uint16_t*>(p);
  p += align_up(w.nchunks * (w.nb + 1) * 2, 256);
  w.totals = reinterpret_cast<int64_t*>(p);
  p += align_up(w.nb * 8, 256);
  w.blk_off = reinterpret_cast<int32_t*>(p);
  p += align_up((w.nb + 1) * 4, 256);
  w.gchunks = reinterpret_cast<int32_t*>(p);
  p += align_up(w.nb * 4, 256);
  w.partial = reinterpret_cast<uint32_t*>(p);
  p += align_up(w.max_parts * kBins * 4, 256);
  w.total = p - static_cast<char*>(ws);
  return w;
}

inline bool use_partition(int64_t E, int64_t N) {
  // strictly fewer buckets than threads: thread t = nb writes the chunk's end offset tab[.., nb], so nb must be <= 1023
  // (with nb == 1024 that entry stayed uninitialised and N in [33 521 664, 33 554 431] counted bucket 1023 wrongly)
  return (N >> kLowBits) + 1 < kMaxBuckets && E >= 2 * N;
}

// counting sort of one 16 K-index chunk by bucket, in LDS; the int64 indices are read once, 2 bytes per key leave
__global__ __launch_bounds__(kHistBlock) void local_partition_kernel(const int64_t* __restrict__ index, int64_t E, int64_t N,
                                                                     int nb, uint16_t* __restrict__ keys16,
                                                                     uint16_t* __restrict__ tab, int32_t* err_flag,
                                                                     const int64_t* __restrict__ ones, int32_t* not_ones) {
  __shared__ uint32_t cnt[kMaxBuckets];
  __shared__ uint32_t cur[kMaxBuckets];
  __shared__ uint32_t wsum[kHistBlock / 64];
  __shared__ __attribute__((aligned(16))) uint16_t stage[kChunk];
  const int t = threadIdx.x;
  const int64_t c = blockIdx.x, k0 = c * kChunk;
  cnt[t] = 0u;
  __syncthreads();
  uint32_t key[kKeysPerThread];
  const bool al16 = (reinterpret_cast<uintptr_t>(index) & 15) == 0;
  bool bad = false, odd = false;
#pragma unroll
  for (int i = 0; i < kKeysPerThread / 2; ++i) {
    const int64_t e = k0 + 2 * (static_cast<int64_t>(i) * kHistBlock + t);
    int64_t v0 = 0, v1 = 0;
    if (e + 1 < E && al16) {
      const longlong2 q = *reinterpret_cast<const longlong2*>(index + e);
      v0 = q.x; v1 = q.y;
    } else {
      if (e < E) v0 = index[e];
      if (e + 1 < E) v1 = index[e + 1];
    }
    if (ones) {
      if (e < E) odd |= ones[e] != 1;
      if (e + 1 < E) odd |= ones[e + 1] != 1;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t v = j ? v1 : v0;
      uint32_t k = 0xFFFFFFFFu;                          // no such entry (past the end of the list)
      if (j ? e + 1 < E : e < E) {
        const bool ok = v >= 0 && v < N;
        bad |= !ok;
        k = ok ? static_cast<uint32_t>(v) : static_cast<uint32_t>(N);     // out of range: the sentinel key, never counted
        atomicAdd(&cnt[k >> kLowBits], 1u);
      }
      key[2 * i + j] = k;
    }
  }
  if (bad) flag_error(err_flag, LPG_F_INDEX_RANGE);
  if (odd) atomicOr(not_ones, 1);
  __syncthreads();
  // exclusive scan of the bucket counts (thread t <-> bucket t)
  const uint32_t mine = t < nb ? cnt[t] : 0u;
  uint32_t inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t up = __shfl_up(inc, o, 64);
    if ((t & 63) >= o) inc += up;
  }
  if ((t & 63) == 63) wsum[t >> 6] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < (t >> 6); ++w) base += wsum[w];
  const uint32_t excl = base + inc - mine;
  cur[t] = excl;
  if (t <= nb) tab[c * (nb + 1) + t] = static_cast<uint16_t>(excl);       // t == nb: the chunk's valid keys (<= 16384)
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kKeysPerThread; ++i) {
    const uint32_t k = key[i];
    if (k != 0xFFFFFFFFu) stage[atomicAdd(&cur[k >> kLowBits], 1u)] = static_cast<uint16_t>(k & (kBins - 1));
  }
  __syncthreads();
  uint4* dst = reinterpret_cast<uint4*>(keys16 + k0);                      // (slots past the valid keys are never read)
  const uint4* src = reinterpret_cast<const uint4*>(stage);
  for (int i = t; i < kChunk / 8; i += kHistBlock) dst[i] = src[i];
}

// totals[b] = keys of bucket b over all chunks (one workgroup per bucket)
__global__ __launch_bounds__(kBlock) void bucket_totals_kernel(const uint16_t* __restrict__ tab, int64_t nchunks, int nb,
                                                               int64_t* __restrict__ totals) {
  __shared__ int64_t red[kBlock / 64];
  const int b = blockIdx.x, t = threadIdx.x;
  int64_t s = 0;
  for (int64_t c = t; c < nchunks; c += kBlock) s += static_cast<int64_t>(tab[c * (nb + 1) + b + 1]) - tab[c * (nb + 1) + b];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((t & 63) == 0) red[t >> 6] = s;
  __syncthreads();
  if (t == 0) {
    int64_t tot = 0;
    for (int w = 0; w < kBlock / 64; ++w) tot += red[w];
    totals[b] = tot;
  }
}

// parts per bucket (~kPart keys each, cut at chunk borders: gchunks[b] chunks per part) and their exclusive scan
__global__ __launch_bounds__(kHistBlock) void part_bounds_kernel(const int64_t* __restrict__ totals, int64_t nchunks, int nb,
                                                                 int32_t* __restrict__ blk_off, int32_t* __restrict__ gchunks) {
  __shared__ int32_t cnt[kMaxBuckets + 1];
  const int t = threadIdx.x;
  for (int b = t; b < nb; b += kHistBlock) {
    const int64_t tb = totals[b];
    const int64_t parts = tb > 0 ? (tb + kPart - 1) / kPart : 0;
    const int64_t g = parts > 0 ? (nchunks + parts - 1) / parts : nchunks;
    gchunks[b] = static_cast<int32_t>(g);
    cnt[b] = parts > 0 ? static_cast<int32_t>((nchunks + g - 1) / g) : 0;       // parts that own at least one chunk
  }
  __syncthreads();
  if (t == 0) {                                            // nb <= 1024: a serial scan is a few microseconds
    int32_t run = 0;
    for (int b = 0; b < nb; ++b) { const int32_t c = cnt[b]; blk_off[b] = run; run += c; }
    blk_off[nb] = run;
  }
}

__global__ __launch_bounds__(kHistBlock) void part_count_kernel(const uint16_t* __restrict__ keys16,
                                                                const uint16_t* __restrict__ tab, int64_t nchunks,
                                                                const int32_t* __restrict__ blk_off,
                                                                const int32_t* __restrict__ gchunks, int nb,
                                                                uint32_t* __restrict__ partial) {
  __shared__ uint32_t hist[kBins];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int blk = blockIdx.x;
  if (blk >= blk_off[nb]) return;                          // the grid is the upper bound (block-uniform exit)
  int lo = 0, hi = nb;                                     // last bucket whose first workgroup is <= blk
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (blk_off[mid] <= blk) lo = mid; else hi = mid;
  }
  const int b = lo;
  const int64_t g = gchunks[b];
  const int64_t c_lo = static_cast<int64_t>(blk - blk_off[b]) * g, c_hi = c_lo + g < nchunks ? c_lo + g : nchunks;
  uint4* h4 = reinterpret_cast<uint4*>(hist);
  for (int i = t; i < kBins / 4; i += kHistBlock) h4[i] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  // Two shapes of work, chosen per bucket (block-uniform): a part of g chunks holds ~kPart keys, so a bucket's segment
  // in one chunk is ~kPart / g keys long.
  if (g <= 512) {
    // LONG segments (hub buckets: hundreds to thousands of keys per chunk): a wave per chunk, the lanes walk the
    // segment together with 16-byte loads (8 keys), two loads in flight
    for (int64_t c = c_lo + wave; c < c_hi; c += kHistBlock / 64) {
      const int64_t s0 = c * kChunk + tab[c * (nb + 1) + b], s1 = c * kChunk + tab[c * (nb + 1) + b + 1];
      int64_t a0 = (s0 + 7) & ~int64_t(7), a1 = s1 & ~int64_t(7);
      if (a0 > a1) { a0 = s1; a1 = s1; }
      for (int64_t k = s0 + lane; k < a0; k += 64) atomicAdd(&hist[keys16[k]], 1u);
      const uint4* k8 = reinterpret_cast<const uint4*>(keys16);
      int64_t q = (a0 >> 3) + lane;
      const int64_t q1 = a1 >> 3;
      for (; q + 192 < q1; q += 256) {
        const uint4 va = k8[q], vb = k8[q + 64], vc = k8[q + 128], vd = k8[q + 192];
        const uint32_t w[16] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w, vc.x, vc.y, vc.z, vc.w, vd.x, vd.y, vd.z, vd.w};
#pragma unroll
        for (int i = 0; i < 16; ++i) { atomicAdd(&hist[w[i] & 0xFFFFu], 1u); atomicAdd(&hist[w[i] >> 16], 1u); }
      }
      for (; q < q1; q += 64) {
        const uint4 va = k8[q];
        const uint32_t w[4] = {va.x, va.y, va.z, va.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { atomicAdd(&hist[w[i] & 0xFFFFu], 1u); atomicAdd(&hist[w[i] >> 16], 1u); }
      }
      for (int64_t k = a1 + lane; k < s1; k += 64) atomicAdd(&hist[keys16[k]], 1u);
    }
  } else {
    // SHORT segments (~100 keys = 200 B for an average bucket): a wave takes 64 consecutive chunks, every LANE walks the
    // segment of its own chunk -- as the aligned 16-byte blocks that cover it, four independent loads per step, keys
    // outside [g0, g1) masked.  (Walking the segments one after the other as a wave paid one global round trip per
    // segment: 0.80 ms of 1.13 at |E| = 100 M; 2-byte loads, four per step, still 27 dependent round trips: 0.50.)
    const uint4* k8 = reinterpret_cast<const uint4*>(keys16);
    for (int64_t cg = c_lo + 64 * wave; cg < c_hi; cg += 64 * (kHistBlock / 64)) {
      const int64_t c = cg + lane;
      int64_t g0 = 0, g1 = 0;
      if (c < c_hi) {
        g0 = c * kChunk + tab[c * (nb + 1) + b];
        g1 = c * kChunk + tab[c * (nb + 1) + b + 1];
      }
      const int64_t qe = (g1 + 7) >> 3;
      for (int64_t qb = g0 >> 3; qb < qe; qb += 4) {
        uint4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = qb + i < qe ? k8[qb + i] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int64_t k0 = (qb + i) << 3;
          const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int64_t k = k0 + j;
            if (k >= g0 && k < g1) atomicAdd(&hist[(w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu], 1u);
          }
        }
      }
    }
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
  // four consecutive bins per thread (16-byte loads of the partial histograms, two parts in flight)
  const uint4* p4 = reinterpret_cast<const uint4*>(partial);
  for (int i4 = blockIdx.x * kBlock + threadIdx.x; i4 < kBins / 4; i4 += gridDim.x * kBlock) {
    const int64_t bin = (static_cast<int64_t>(b) << kLowBits) + 4 * i4;
    if (bin >= N) break;
    int64_t sum[4] = {0, 0, 0, 0};
    int p = p0;
    for (; p + 1 < p1; p += 2) {
      const uint4 a = p4[static_cast<int64_t>(p) * (kBins / 4) + i4], c = p4[static_cast<int64_t>(p + 1) * (kBins / 4) + i4];
      sum[0] += static_cast<int64_t>(a.x) + c.x; sum[1] += static_cast<int64_t>(a.y) + c.y;
      sum[2] += static_cast<int64_t>(a.z) + c.z; sum[3] += static_cast<int64_t>(a.w) + c.w;
    }
    if (p < p1) {
      const uint4 a = p4[static_cast<int64_t>(p) * (kBins / 4) + i4];
      sum[0] += a.x; sum[1] += a.y; sum[2] += a.z; sum[3] += a.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (sum[j] && bin + j < N) out[bin + j] += sum[j];
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

namespace {
int degree_impl(const int64_t* index, const int64_t* ones, int32_t* not_ones, int64_t E, int64_t* out, int64_t N, void* workspace,
                int64_t workspace_bytes, int32_t* err_flag, void* stream);
}

LPG_EXPORT int lpg_degree_i64(const int64_t* index, int64_t E, int64_t* out, int64_t N, void* workspace,
                              int64_t workspace_bytes, int32_t* err_flag, void* stream) {
  return degree_impl(index, nullptr, nullptr, E, out, N, workspace, workspace_bytes, err_flag, stream);
}

LPG_EXPORT int lpg_degree_i64_ones(const int64_t* index, const int64_t* ones, int64_t E, int64_t* out, int64_t N,
                                   void* workspace, int64_t workspace_bytes, int32_t* not_ones, int32_t* err_flag,
                                   void* stream) {
  LPG_REQUIRE(E == 0 || (ones && not_ones), "degree_i64_ones: null pointer");
  LPG_REQUIRE(E == 0 || (E >= (int64_t(1) << 21) && N > 16384 && E < (int64_t(1) << 31) && N < (int64_t(1) << 31) && workspace),
              "degree_i64_ones: only the atomic-free paths check the source (E >= 2^21, N > 16384, workspace given)");
  return degree_impl(index, ones, not_ones, E, out, N, workspace, workspace_bytes, err_flag, stream);
}

namespace {
int degree_impl(const int64_t* index, const int64_t* ones, int32_t* not_ones, int64_t E, int64_t* out, int64_t N, void* workspace,
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
    const int nb = static_cast<int>(w.nb);
    hipLaunchKernelGGL(local_partition_kernel, dim3(static_cast<unsigned>(w.nchunks)), dim3(kHistBlock), 0, st, index, E, N, nb,
                       w.keys16, w.tab, err_flag, ones, not_ones);
    hipLaunchKernelGGL(bucket_totals_kernel, dim3(static_cast<unsigned>(nb)), dim3(kBlock), 0, st, w.tab, w.nchunks, nb, w.totals);
    hipLaunchKernelGGL(part_bounds_kernel, dim3(1), dim3(kHistBlock), 0, st, w.totals, w.nchunks, nb, w.blk_off, w.gchunks);
    hipLaunchKernelGGL(part_count_kernel, dim3(static_cast<unsigned>(w.max_parts)), dim3(kHistBlock), 0, st, w.keys16, w.tab,
                       w.nchunks, w.blk_off, w.gchunks, nb, w.partial);
    hipLaunchKernelGGL(part_fold_kernel, dim3(kBins / 4 / kBlock / 2, static_cast<unsigned>(nb)), dim3(kBlock), 0, st,
                       w.partial, w.blk_off, N, out);
    return check_launch("degree_i64");
  }
  DegWs w = carve(workspace, E, N);
  if (workspace_bytes < w.total) {
    set_error("degree_i64: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)w.total);
    return LPG_E_WORKSPACE;
  }
  hipLaunchKernelGGL(narrow_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, index, E, N, w.keys_in, err_flag, ones, not_ones);
  size_t tmp = w.sort_tmp_bytes;
  LPG_HIP(rocprim::radix_sort_keys(w.sort_tmp, tmp, w.keys_in, w.keys_out, static_cast<size_t>(E), 0u, bits_for(N), st,
                                   false));
  hipLaunchKernelGGL(run_first_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, w.keys_out, E, w.first);
  hipLaunchKernelGGL(run_count_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, w.keys_out, E, N, w.first, out);
  return check_launch("degree_i64");
}
}  // namespace

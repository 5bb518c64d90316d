// SEAL structure path on the device (a10, a11, a12): batched 0/1-hop enclosing subgraphs, DRNL
// node labels, block-diagonal A~ = A + I with D^-1, and X = [onehot(tag) || emb].
//
// Reference: SEAL/config/subgraph.py:105-141 (extract_subgraph), SEAL/utils/utils.py:35-54
// (node_labeling), SEAL/config/data.py:94-140 (create_input_for_gnn_fly).  The reference walks
// candidates one at a time with networkx (2(n-2) BFS calls per subgraph) and feeds dense [n,n]
// float matrices; here one workgroup owns one candidate pair, the subgraph lives in LDS as a
// 128x128 bit matrix, and both BFS runs are level-synchronous bitset sweeps.
#include <cstring>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "lpg_common.hpp"

namespace {
using namespace lpg;

constexpr int MAXN = LPG_SEAL_MAXN;  // 128 >= the reference's max_neighbors = 100
constexpr int WORDS = MAXN / 32;
constexpr int kSealBlock = 128;      // thread i <-> subgraph vertex i
constexpr int kDeferGroup = 8;       // consecutive pairs a 128-thread workgroup owns in the fused launch on big batches (it handles those no wave can)
__device__ __forceinline__ int64_t ceil_div_dev(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ bool row_has(const int32_t* __restrict__ row, int len, int32_t key) {
  int lo = 0, hi = len;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const int32_t v = row[mid];
    if (v < key) lo = mid + 1; else hi = mid;
  }
  return lo < len && row[lo] == key;
}

struct ListPlan {
  int32_t node;     // whose neighbour list
  int count;        // entries taken from it
  int64_t deg;      // its degree
  uint64_t a, b;    // entry i is neighbour (a*i + b) mod deg  (identity for a full list)
};

__device__ __forceinline__ uint64_t mix64(uint64_t z) {   // splitmix64 finaliser
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__device__ __forceinline__ int64_t gcd64(int64_t x, int64_t y) {
  while (y) { const int64_t r = x % y; x = y; y = r; }
  return x;
}

// `m` distinct neighbours of `node` in a pseudo-random order: an affine permutation of [0, deg)
__device__ ListPlan truncated_list(int32_t node, int64_t deg, int m, uint64_t h) {
  ListPlan L{node, 0, deg, 1, 0};
  if (deg <= 0 || m <= 0) return L;
  L.count = static_cast<int>(m < deg ? m : deg);
  if (L.count > MAXN) L.count = MAXN;
  if (deg > 1) {
    int64_t a = 1 + static_cast<int64_t>(h % static_cast<uint64_t>(deg - 1));
    while (gcd64(a, deg) != 1) a = a % (deg - 1) + 1;
    L.a = static_cast<uint64_t>(a);
    L.b = mix64(h) % static_cast<uint64_t>(deg);
  }
  return L;
}

// ---- small pairs: ONE WAVE per candidate pair ---------------------------------------------------------------------
// 60 % of the candidate pairs of a collab-shaped batch have fewer than 8 subgraph vertices, 94 % fewer than 64; in
// the 128-thread workgroup above such a pair is ~6 us of barriers and dependent round trips with most threads idle.
// Here a pair whose two neighbour lists hold at most 62 entries together (so n <= 64 and the reference's cap cannot
// fire) is handled by one 64-lane wave without a single workgroup barrier: lane i <-> subgraph vertex i, an adjacency
// row is ONE 64-bit word, the DRNL sweeps are wave-wide OR-reductions.  Same node order, adjacency bits and tags as
// the workgroup path, bit for bit (tests run both on the same pairs).  Every other pair (long lists, the cap) is taken
// by a 128-thread workgroup of the SAME launch (seal_subgraph_kernel, fused mode): the two kinds of work overlap, the
// long hub-rich pairs start first.
constexpr int kWaveN = 64;                  // vertices a wave handles: lane i <-> vertex i

struct WavePair {
  int32_t nodes[kWaveN];
  unsigned long long adj[kWaveN];
  int pre[kWaveN + 1];
  int64_t rbase[kWaveN];
  int rdeg[kWaveN];
  int big[kWaveN];
};

__device__ __forceinline__ void wave_lds_sync() {       // LDS written by some lanes of the wave, read by others
  __builtin_amdgcn_s_waitcnt(0xC07F);                    // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ uint64_t wave_or64(uint64_t v) {   // OR over the 64 lanes, result in every lane
  uint32_t lo = static_cast<uint32_t>(v), hi = static_cast<uint32_t>(v >> 32);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    lo |= __shfl_xor(lo, o, 64);
    hi |= __shfl_xor(hi, o, 64);
  }
  return (static_cast<uint64_t>(hi) << 32) | lo;
}

// DRNL sweeps of a 128-thread team without a workgroup barrier: wave w runs the whole BFS from target w on its own --
// lane l holds the 128-bit adjacency rows of vertices l and l + 64, a level is two wave-wide OR-reductions of the
// frontier's rows.  (The level-synchronous version of rounds 1-2 spent three __syncthreads and up to n same-address LDS atomics
// per level: a third of a heavy pair's time, profiles/seal_phases.py.)  Call with all 128 threads after a barrier that
// publishes adj; dist0 / dist1 are complete after the caller's next barrier.
__device__ __forceinline__ void bfs_team_waves(const uint32_t (*adj)[WORDS], int n, int* dist0, int* dist1) {
  const int lane = threadIdx.x & 63, src = threadIdx.x >> 6;       // kSealBlock = 128: two waves, one source each
  const int va = lane, vb = lane + 64;
  uint64_t a_lo = 0, a_hi = 0, b_lo = 0, b_hi = 0;
  if (va < n) {
    a_lo = adj[va][0] | (static_cast<uint64_t>(adj[va][1]) << 32);
    a_hi = adj[va][2] | (static_cast<uint64_t>(adj[va][3]) << 32);
  }
  if (vb < n) {
    b_lo = adj[vb][0] | (static_cast<uint64_t>(adj[vb][1]) << 32);
    b_hi = adj[vb][2] | (static_cast<uint64_t>(adj[vb][3]) << 32);
  }
  uint64_t vis_lo = 1ull << src, vis_hi = 0ull, fr_lo = vis_lo, fr_hi = 0ull;
  int da = lane == src ? 0 : -1, db = -1;
  for (int level = 1; level <= n; ++level) {
    const bool ina = (fr_lo >> lane) & 1ull, inb = (fr_hi >> lane) & 1ull;
    const uint64_t n_lo = wave_or64((ina ? a_lo : 0ull) | (inb ? b_lo : 0ull)) & ~vis_lo;
    const uint64_t n_hi = wave_or64((ina ? a_hi : 0ull) | (inb ? b_hi : 0ull)) & ~vis_hi;
    if ((n_lo >> lane) & 1ull) da = level;
    if ((n_hi >> lane) & 1ull) db = level;
    vis_lo |= n_lo; vis_hi |= n_hi; fr_lo = n_lo; fr_hi = n_hi;
    if (!(n_lo | n_hi)) break;                                     // wave-uniform
  }
  int* dist = src ? dist1 : dist0;
  dist[va] = da;
  dist[vb] = db;
}

// the rule both kinds of workgroup apply to decide who owns a pair (targets already range-checked)
__device__ __forceinline__ bool pair_fits_wave(int hop, int max_nb, int32_t u, int32_t v, int64_t du, int64_t dv) {
  const int size0 = (u == v) ? 1 : 2;
  return hop == 0 || (size0 + du + dv < max_nb && du + dv <= kWaveN - 2);
}

__device__ __forceinline__ void wave_pair(
    WavePair& L, int64_t b, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t n_graph,
    const int64_t* __restrict__ pairs, int hop, int max_nb, int32_t* __restrict__ nodes_out, int32_t* __restrict__ n_out,
    int32_t* __restrict__ tags_out, uint32_t* __restrict__ adj_out, uint8_t* __restrict__ capped_out, int32_t* err_flag,
    uint8_t* __restrict__ tag_present, const int32_t* __restrict__ hub_index, const uint32_t* __restrict__ hub_bits,
    int hub_words, int big_deg) {
  const int lane = threadIdx.x & 63;
  const int64_t u64 = pairs[2 * b], v64 = pairs[2 * b + 1];
  if (u64 < 0 || u64 >= n_graph || v64 < 0 || v64 >= n_graph) {
    if (lane == 0) { flag_error(err_flag, LPG_F_INDEX_RANGE); n_out[b] = 0; capped_out[b] = 0; }
    return;
  }
  const int32_t u = static_cast<int32_t>(u64), v = static_cast<int32_t>(v64);
  const int64_t bu = rowptr[u], du = rowptr[u + 1] - bu;
  const int64_t bv = rowptr[v], dv = rowptr[v + 1] - bv;
  const int64_t dv_used = (u == v) ? 0 : dv;             // the second list is not walked when the targets coincide
  if (!pair_fits_wave(hop, max_nb, u, v, du, dv)) return;        // a 128-thread workgroup of this launch owns it
  // ---- node set {u, v} U N(u) U N(v): duplicates dropped (first occurrence kept), the rest ascending
  int n = 2;
  L.adj[lane] = 0ull;
  if (hop >= 1) {
    const int nc = static_cast<int>(du + dv_used);
    int32_t x = -1;
    if (lane < nc) {
      x = lane < du ? col[bu + lane] : col[bv + (lane - du)];
      if (x == u || x == v) x = -1;
    }
    bool keep = x >= 0;
    for (int j = 0; j < nc; ++j) {
      const int32_t y = __shfl(x, j, 64);
      if (j < lane && y == x) keep = false;
    }
    const int32_t xs = keep ? x : 0x7FFFFFFF;
    int rank = 0;
    for (int j = 0; j < nc; ++j) rank += __shfl(xs, j, 64) < x;
    if (keep) L.nodes[2 + rank] = x;
    n = 2 + __popcll(__ballot(keep));
  }
  if (lane == 0) { L.nodes[0] = u; L.nodes[1] = v; }
  wave_lds_sync();
  // ---- induced adjacency (one word per row), exactly the rules of seal_subgraph_kernel
  bool isbig = false;
  int mylen = 0;
  if (lane < n) {
    const int32_t a = L.nodes[lane];
    const int64_t ba = rowptr[a];
    const int da = static_cast<int>(rowptr[a + 1] - ba);
    L.rbase[lane] = ba;
    L.rdeg[lane] = da;
    isbig = da > big_deg;
    mylen = isbig ? 0 : da;
  }
  const uint64_t bigmask = __ballot(isbig);
  const int nbig = __popcll(bigmask);
  if (isbig) L.big[__popcll(bigmask & ((uint64_t(1) << lane) - 1))] = lane;
  int inc = mylen;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(inc, o, 64);
    if (lane >= o) inc += up;
  }
  L.pre[lane] = inc - mylen;
  const int total = __shfl(inc, 63, 64);
  if (lane == 0) L.pre[kWaveN] = total;
  wave_lds_sync();
  constexpr int IT = 4;
  for (int p0 = IT * lane; p0 < total; p0 += IT * 64) {
    int32_t x[IT];
    uint32_t rows = 0;
    {
      int lo = 0, hi = n;                                // last row i with pre[i] <= p0
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (L.pre[mid] <= p0) lo = mid; else hi = mid;
      }
      int rr = lo;
#pragma unroll
      for (int q = 0; q < IT; ++q) {
        const int p = p0 + q;
        while (rr + 1 < n && L.pre[rr + 1] <= p) ++rr;
        x[q] = p < total ? col[L.rbase[rr] + (p - L.pre[rr])] : -1;
        rows |= static_cast<uint32_t>(rr) << (8 * q);
      }
    }
#pragma unroll
    for (int q = 0; q < IT; ++q) {
      const int i = static_cast<int>((rows >> (8 * q)) & 0xFFu);
      const int32_t xv = x[q];
      if (xv < 0) continue;
      const unsigned long long ibit = 1ull << i;
      if (xv == u && i != 1) { atomicOr(&L.adj[i], 1ull); atomicOr(&L.adj[0], ibit); }
      if (xv == v && i != 0) { atomicOr(&L.adj[i], 2ull); atomicOr(&L.adj[1], ibit); }
      if (xv != u && xv != v) {
        int lo = 2, hi = n;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (L.nodes[mid] < xv) lo = mid + 1; else hi = mid;
        }
        if (lo < n && L.nodes[lo] == xv) {
          atomicOr(&L.adj[i], 1ull << lo);
          atomicOr(&L.adj[lo], ibit);
        }
      }
    }
  }
  // rows with more than 256 neighbours are not streamed: symmetry filled them from the other side; what is left are
  // the entries between two such rows -- the hub bit matrix when the graph has one, else a binary search
  for (int p = lane; p < nbig * nbig; p += 64) {
    const int a = p / nbig, b2 = p - a * nbig;
    if (a > b2) continue;
    const int i = L.big[a], j = L.big[b2];
    if ((i == 0 && j == 1) || (i == 1 && j == 0)) continue;
    bool hit;
    const int ia = hub_index ? hub_index[L.nodes[i]] : -1, ib = hub_index ? hub_index[L.nodes[j]] : -1;
    if (ia >= 0 && ib >= 0) {
      hit = (hub_bits[static_cast<int64_t>(ia) * hub_words + (ib >> 5)] >> (ib & 31)) & 1u;
    } else {
      hit = row_has(col + L.rbase[i], L.rdeg[i], L.nodes[j]);
    }
    if (hit) { atomicOr(&L.adj[i], 1ull << j); atomicOr(&L.adj[j], 1ull << i); }
  }
  wave_lds_sync();
  // ---- DRNL (utils.py:35-54): both BFS sweeps as wave-wide OR-reductions of the frontier's rows
  const uint64_t row = lane < n ? L.adj[lane] : 0ull;
  uint64_t vis0 = 1ull, fr0 = 1ull, vis1 = 2ull, fr1 = 2ull;
  int d0 = lane == 0 ? 0 : -1, d1 = lane == 1 ? 0 : -1;
  for (int level = 1; level <= n; ++level) {
    const bool in0 = (fr0 >> lane) & 1ull, in1 = (fr1 >> lane) & 1ull;
    const uint64_t f0 = wave_or64(in0 ? row : 0ull) & ~vis0, f1 = wave_or64(in1 ? row : 0ull) & ~vis1;
    if ((f0 >> lane) & 1ull) d0 = level;
    if ((f1 >> lane) & 1ull) d1 = level;
    vis0 |= f0; vis1 |= f1; fr0 = f0; fr1 = f1;
    if (!(f0 | f1)) break;
  }
  int tag = -1, nd = -1;
  if (lane < n) {
    nd = L.nodes[lane];
    if (lane < 2) {
      tag = 1;
    } else if (d0 < 0 || d1 < 0) {
      tag = 0;
    } else {
      const int d = d0 + d1, dv2 = d >> 1, md = d & 1;
      tag = 1 + (d0 < d1 ? d0 : d1) + dv2 * (dv2 + md - 1);
    }
  }
  __builtin_nontemporal_store(tag, tags_out + b * MAXN + lane);
  __builtin_nontemporal_store(-1, tags_out + b * MAXN + 64 + lane);
  __builtin_nontemporal_store(nd, nodes_out + b * MAXN + lane);
  __builtin_nontemporal_store(-1, nodes_out + b * MAXN + 64 + lane);
  if (tag_present && tag >= 0) tag_present[tag] = 1;
  using u4 = __attribute__((ext_vector_type(4))) uint32_t;
  u4* ao = reinterpret_cast<u4*>(adj_out + b * MAXN * WORDS);
  const u4 r4 = {static_cast<uint32_t>(row), static_cast<uint32_t>(row >> 32), 0u, 0u}, z4 = {0u, 0u, 0u, 0u};
  __builtin_nontemporal_store(r4, ao + lane);
  __builtin_nontemporal_store(z4, ao + 64 + lane);
  if (lane == 0) { n_out[b] = n; capped_out[b] = 0; }
}

// -DLPG_TUNING builds only (profiles/seal_phases.py): where a 128-thread team's time goes, summed over the launch in ticks
// of the 100 MHz wall clock -- [0] node set, [1] row descriptors + filter, [2] streamed lists, [3] hub bit matrix,
// [4] searches, [5] DRNL, [6] outputs, [7] pairs handled by teams.  Adds a barrier per phase; release builds have none of it.
#ifdef LPG_TUNING
__device__ unsigned long long g_seal_phase[8];
#define SEAL_PHASE(k)                                              \
  do {                                                             \
    __syncthreads();                                               \
    if (threadIdx.x == 0) {                                        \
      const unsigned long long now_ = wall_clock64();              \
      ph_acc[k] += now_ - ph_t0;                                   \
      ph_t0 = now_;                                                \
    }                                                              \
  } while (0)
#else
#define SEAL_PHASE(k) do { } while (0)
#endif

// LDS of a 128-thread team working on one pair
struct BigPairLds {
  int32_t cand[2 * MAXN];
  int32_t nodes[MAXN];
  uint32_t adj[MAXN][WORDS];
  int dist0[MAXN], dist1[MAXN];
  int s_n, s_capped, s_cnt[2], s_nbig, s_search;
  ListPlan s_list[2];
  int64_t rbase[MAXN];                  // per subgraph vertex: its neighbour list in the big graph
  int rdeg[MAXN];
  int big[MAXN];                        // rows whose list is searched instead of streamed
  int pre[MAXN + 2];                    // exclusive prefix of the streamed rows' list lengths
  uint32_t member[128];                 // 4096-bit hash filter of the subgraph's vertices (no false negatives)
  int hidx[MAXN];                       // hub-matrix row of each big row's node (-1: none)
};

// latency-bound: 8 waves per SIMD (64 VGPRs, 6 of them spilled) beat 6 waves with none
__global__ __launch_bounds__(kSealBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void seal_subgraph_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t n_graph, const int64_t* __restrict__ pairs,
    int64_t B, int hop, int max_nb, int replay, int cap_policy, uint64_t seed, int32_t* __restrict__ nodes_out, int32_t* __restrict__ n_out,
    int32_t* __restrict__ tags_out, uint32_t* __restrict__ adj_out, uint8_t* __restrict__ capped_out, int32_t* err_flag,
    uint8_t* __restrict__ tag_present, const int32_t* __restrict__ hub_index, const uint32_t* __restrict__ hub_bits,
    int hub_words, int fused, int directed, int big_deg) {
  const int group = fused > 0 ? fused : 1;           // fused = consecutive pairs per team workgroup (0: not fused)
  // directed != 0 (SEAL/config/subgraph.py:30-31,69 with network_type = 1): the CSR lists SUCCESSORS only and the induced
  // adjacency is not symmetric -- no entry is mirrored, and rows that are not streamed are searched against every column.
  // fused == 0: workgroup w takes pairs w, w + gridDim.x, ... with all 128 threads (every pair; replayed node lists).
  // fused != 0: ONE launch, two kinds of workgroup.  The first ceil(B / kDeferGroup) workgroups each own kDeferGroup
  // consecutive pairs and handle, with all 128 threads, those that do not fit a wave (long neighbour lists, the cap);
  // the workgroups after them are two independent waves, one small pair each (wave_pair above).  The heavy pairs are
  // dispatched first and overlap with the stream of small ones.  (First tried: a wave kernel that queued the rest for
  // a second launch popping an atomic counter -- 100 K same-address atomics take 4-6 ms on this chip, ~50 ns each,
  // whichever side pushes or pops; and two back-to-back launches serialise the heavy tail behind the small pairs.)
  // a workgroup is either two waves (wave_pair) or one 128-thread team: the two sets of LDS arrays share storage
  // (10 KB per workgroup keeps 16 workgroups = 8 waves per SIMD on a CU's 160 KB)
  __shared__ union {
    WavePair wave[kSealBlock / 64];
    BigPairLds team;
  } lds;
  WavePair* const wp = lds.wave;
  __shared__ uint8_t s_defer[kDeferGroup];
  const int64_t n_groups = fused ? ceil_div_dev(B, group) : 0;
  if (fused && static_cast<int64_t>(blockIdx.x) >= n_groups) {
    const int64_t bw = (static_cast<int64_t>(blockIdx.x) - n_groups) * (kSealBlock / 64) + (threadIdx.x >> 6);
    if (bw < B)                                          // wave-uniform; wave_pair has no workgroup barrier
      wave_pair(wp[threadIdx.x >> 6], bw, rowptr, col, n_graph, pairs, hop, max_nb, nodes_out, n_out, tags_out, adj_out,
                capped_out, err_flag, tag_present, hub_index, hub_bits, hub_words, big_deg);
    return;
  }
  int32_t (&cand)[2 * MAXN] = lds.team.cand;
  int32_t (&nodes)[MAXN] = lds.team.nodes;
  uint32_t (&adj)[MAXN][WORDS] = lds.team.adj;
  int (&dist0)[MAXN] = lds.team.dist0;
  int (&dist1)[MAXN] = lds.team.dist1;
  int& s_n = lds.team.s_n;
  int& s_capped = lds.team.s_capped;
  int (&s_cnt)[2] = lds.team.s_cnt;
  int& s_nbig = lds.team.s_nbig;
  int& s_search = lds.team.s_search;
  ListPlan (&s_list)[2] = lds.team.s_list;
  int64_t (&rbase)[MAXN] = lds.team.rbase;
  int (&rdeg)[MAXN] = lds.team.rdeg;
  int (&big)[MAXN] = lds.team.big;
  int (&pre)[MAXN + 2] = lds.team.pre;
  uint32_t (&member)[128] = lds.team.member;
  int (&hidx)[MAXN] = lds.team.hidx;
  const int t = threadIdx.x;
#ifdef LPG_TUNING
  unsigned long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // thread 0's sums; added to the global table once per workgroup
#endif

  for (int64_t iter = 0;; ++iter) {
    __syncthreads();
    int64_t b;
    if (fused) {
      if (iter == 0) {                                   // who owns my kDeferGroup pairs: one thread per pair, in parallel
        if (t < group) {
          const int64_t bq = static_cast<int64_t>(blockIdx.x) * group + t;
          bool mine = false;
          if (bq < B) {
            const int64_t uq = pairs[2 * bq], vq = pairs[2 * bq + 1];
            if (uq >= 0 && uq < n_graph && vq >= 0 && vq < n_graph)       // (an out-of-range pair is flagged by its wave)
              mine = !pair_fits_wave(hop, max_nb, static_cast<int32_t>(uq), static_cast<int32_t>(vq),
                                     rowptr[uq + 1] - rowptr[uq], rowptr[vq + 1] - rowptr[vq]);
          }
          s_defer[t] = mine;
        }
        __syncthreads();
      }
      if (iter >= group) break;
      b = static_cast<int64_t>(blockIdx.x) * group + iter;
      if (b >= B) break;
      if (!s_defer[iter]) continue;                      // block-uniform
    } else {
      b = static_cast<int64_t>(blockIdx.x) + iter * gridDim.x;
      if (b >= B) break;
    }
    const int64_t u64 = pairs[2 * b], v64 = pairs[2 * b + 1];
    const bool bad = u64 < 0 || u64 >= n_graph || v64 < 0 || v64 >= n_graph;
    if (bad) {                                       // block-uniform
      if (t == 0) { flag_error(err_flag, LPG_F_INDEX_RANGE); n_out[b] = 0; capped_out[b] = 0; }
      continue;
    }
    const int32_t u = static_cast<int32_t>(u64), v = static_cast<int32_t>(v64);
#ifdef LPG_TUNING
    unsigned long long ph_t0 = wall_clock64();
    ph_acc[7] += 1ull;
#endif
    if (t == 0) { s_n = 0; s_capped = 0; }
    for (int i = t; i < MAXN; i += kSealBlock) {
#pragma unroll
      for (int w = 0; w < WORDS; ++w) adj[i][w] = 0;
    }
    __syncthreads();

    if (replay) {
      int n = n_out[b];
      if (n < 2 || n > MAXN) n = 0;
      for (int i = t; i < n; i += kSealBlock) {
        int32_t x = nodes_out[b * MAXN + i];
        if (x < 0 || x >= n_graph) { flag_error(err_flag, LPG_F_INDEX_RANGE); x = u; }
        nodes[i] = x;
      }
      if (t == 0) s_n = n;
      __syncthreads();
    } else {
      // ---- node set: {u, v} U N(u) U N(v) with the reference's cap rule (subgraph.py:118-135)
      // The reference visits the two targets in a SHUFFLED order (:122); the truncation branch
      // (:124-130) fires when len(sub) + len(neighbours) >= max_neighbors at a visit.  `capped` is
      // reported when it would fire in EITHER order, so an uncapped result is order- and RNG-free.
      // Everything below is block-uniform; the two neighbour counts are taken by all threads at once
      // (one thread walking up to 2 x 100 list entries was ~10 us of dependent loads per pair).
      const int size0 = (u == v) ? 1 : 2;
      const int64_t bu = rowptr[u], du = rowptr[u + 1] - bu;
      const int64_t bv = rowptr[v], dv = rowptr[v + 1] - bv;
      if (t < 2) s_cnt[t] = 0;
      __syncthreads();
      if (hop >= 1) {                                  // neighbours other than the targets, lists shorter than the cap
        if (size0 + du < max_nb && t < du) { const int32_t x = col[bu + t]; if (x != u && x != v) atomicAdd(&s_cnt[0], 1); }
        if (size0 + dv < max_nb && t < dv) { const int32_t x = col[bv + t]; if (x != u && x != v) atomicAdd(&s_cnt[1], 1); }
      }
      __syncthreads();
      if (t == 0) {
        const int eu = s_cnt[0], ev = s_cnt[1];
        int cap = 0;
        if (hop >= 1) {
          if (size0 + du >= max_nb || size0 + dv >= max_nb) cap = 1;
          else if (size0 + eu + dv >= max_nb || size0 + ev + du >= max_nb) cap = 1;
        }
        s_capped = cap;
        // plan the two neighbour lists: (node, count, affine permutation a*i+b mod deg)
        s_list[0] = s_list[1] = ListPlan{-1, 0, 0, 1, 0};
        if (hop >= 1 && !(cap && cap_policy == 0)) {
          if (!cap) {
            s_list[0] = ListPlan{u, static_cast<int>(du), du, 1, 0};
            if (u != v) s_list[1] = ListPlan{v, static_cast<int>(dv), dv, 1, 0};
          } else {
            // deterministic stand-in for np.random.shuffle: visiting order and the kept neighbours
            // come from a counter-based hash of (seed, u, v) -- of the pair itself, not of its position in
            // the batch, so a candidate list sharded across GPUs extracts the same subgraphs as one batch
            const uint64_t h = mix64(seed ^ (((static_cast<uint64_t>(static_cast<uint32_t>(u)) << 32) |
                                              static_cast<uint32_t>(v)) * 0x9E3779B97F4A7C15ull));
            const bool swap = (h >> 63) != 0;
            const int32_t first = swap ? v : u, second = swap ? u : v;
            const int64_t d1 = swap ? dv : du, d2 = swap ? du : dv;
            int size = size0;
            if (size + d1 < max_nb) {
              s_list[0] = ListPlan{first, static_cast<int>(d1), d1, 1, 0};
              size += swap ? ev : eu;                  // counted above: this list is shorter than the cap
              if (u != v) {
                if (size + d2 < max_nb)
                  s_list[1] = ListPlan{second, static_cast<int>(d2), d2, 1, 0};
                else
                  s_list[1] = truncated_list(second, d2, max_nb - size, mix64(h + 1));
              }
            } else {
              s_list[0] = truncated_list(first, d1, max_nb - size, mix64(h + 2));   // then `break` (:130)
            }
          }
        }
      }
      __syncthreads();
      if (s_capped && cap_policy == 0) {             // RNG-dependent in the reference: caller replays
        if (t == 0) { n_out[b] = 0; capped_out[b] = 1; }
        continue;
      }
      const int cA = s_list[0].count, cB = s_list[1].count;
      const int n_c = cA + cB;                        // <= 2 * MAXN by construction
      for (int i = t; i < n_c; i += kSealBlock) {
        const ListPlan& L = i < cA ? s_list[0] : s_list[1];
        const int64_t k = i < cA ? i : i - cA;
        const int64_t idx = (L.a * static_cast<uint64_t>(k) + L.b) % static_cast<uint64_t>(L.deg);
        const int32_t x = col[rowptr[L.node] + static_cast<int64_t>(idx)];
        cand[i] = (x == u || x == v) ? -1 : x;
      }
      __syncthreads();
      // drop duplicates (keep the first occurrence), then rank sort the survivors
      int32_t mine[2];
      bool keep[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int i = t + q * kSealBlock;
        mine[q] = i < n_c ? cand[i] : -1;
        keep[q] = mine[q] >= 0;
        if (keep[q])
          for (int j = 0; j < i; ++j)
            if (cand[j] == mine[q]) { keep[q] = false; break; }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int i = t + q * kSealBlock;
        if (i < n_c && !keep[q]) cand[i] = -1;
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if (keep[q]) {
          int rank = 0;
          for (int j = 0; j < n_c; ++j) { const int32_t y = cand[j]; rank += (y >= 0 && y < mine[q]); }
          if (2 + rank < MAXN) nodes[2 + rank] = mine[q];
          atomicAdd(&s_n, 1);
        }
      if (t == 0) { nodes[0] = u; nodes[1] = v; }
      __syncthreads();
      if (t == 0) s_n = (s_n + 2 > MAXN) ? MAXN : s_n + 2;
      __syncthreads();
    }
    const int n = s_n;
    if (n == 0) {
      if (t == 0) { n_out[b] = 0; capped_out[b] = replay ? 0 : 1; }
      continue;
    }
    const int was_capped = replay ? 0 : s_capped;
    SEAL_PHASE(0);

    // ---- induced adjacency: bit (i, j) = A[nodes[i], nodes[j]], target entries cleared (:136-137)
    // Row i's neighbour list is STREAMED (16 lanes per row, coalesced) and each neighbour looked up among
    // the subgraph's vertices in LDS (nodes[2..n) is sorted ascending by construction).  Only rows with a
    // long list (hubs: thousands of neighbours) are not streamed: symmetry fills them from the other side, and
    // only hub x hub entries are binary-searched.  The first version searched every (i, j): 10 K dependent-load
    // chains for a capped subgraph.
    if (t == 0) s_nbig = 0;
    __syncthreads();
    if (t < n) {
      const int32_t a = nodes[t];
      const int64_t ba = rowptr[a];
      const int da = static_cast<int>(rowptr[a + 1] - ba);
      rbase[t] = ba;
      rdeg[t] = da;
      if (replay || da > big_deg) big[atomicAdd(&s_nbig, 1)] = t;     // replayed node lists need not be sorted
    }
    __syncthreads();
    if (!replay) {
      // exclusive prefix of the streamed rows' list lengths (wave scan + the first wave's total), then the
      // (row, k) items are dealt round-robin to the threads: no thread walks a row alone, and the 4 loads of a
      // round are independent (per-row loops paid one global round trip per row: 20 us for a 100-vertex subgraph)
      const int mylen = (t < n && rdeg[t] <= big_deg) ? rdeg[t] : 0;
      int inc = mylen;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(inc, o, 64);
        if ((t & 63) >= o) inc += up;
      }
      if (t == 63) pre[MAXN] = inc;                    // total of wave 0
      __syncthreads();
      const int excl = inc - mylen + (t >= 64 ? pre[MAXN] : 0);
      pre[t] = excl;
      if (t == kSealBlock - 1) pre[MAXN + 1] = excl + mylen;
      __syncthreads();
      const int total = pre[MAXN + 1];
      // Most streamed neighbours are NOT subgraph vertices (lists of up to 256 against <= 128 members): a hash bit
      // filter answers that with ONE LDS read instead of a 7-probe binary search; only filter hits are searched.
      // A thread takes 4 CONSECUTIVE items: one search for the row of the first, the others follow by stepping.
      for (int i = t; i < 128; i += kSealBlock) member[i] = 0u;
      __syncthreads();
      if (t >= 2 && t < n) {
        const uint32_t hsh = (static_cast<uint32_t>(nodes[t]) * 0x9E3779B1u) >> 20;
        atomicOr(&member[hsh >> 5], 1u << (hsh & 31));
      }
      __syncthreads();
      SEAL_PHASE(1);
      constexpr int IT = 4;                              // consecutive items per thread and round (8: no gain on hub-rich subgraphs -- the
                                                         // phase is instruction-bound there, ~40 per entry -- and 7 % lost on the small ones)
      for (int p0 = IT * t; p0 < total; p0 += IT * kSealBlock) {
        int32_t x[IT];
        uint32_t rows_lo = 0, rows_hi = 0;               // the items' rows, 8 bits each (n <= 128)
        {
          int lo = 0, hi = n;                            // last row i with pre[i] <= p0 (empty rows share a prefix value)
          while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pre[mid] <= p0) lo = mid; else hi = mid;
          }
          int rr = lo;
#pragma unroll
          for (int q = 0; q < IT; ++q) {
            const int p = p0 + q;
            while (rr + 1 < n && pre[rr + 1] <= p) ++rr;  // the next item may start a later row (empty rows skipped)
            x[q] = p < total ? col[rbase[rr] + (p - pre[rr])] : -1;
            if (q < 4) rows_lo |= static_cast<uint32_t>(rr) << (8 * q); else rows_hi |= static_cast<uint32_t>(rr) << (8 * (q - 4));
          }
        }
#pragma unroll
        for (int q = 0; q < IT; ++q) {
          const int i = static_cast<int>(((q < 4 ? rows_lo : rows_hi) >> (8 * (q & 3))) & 0xFFu);
          const int32_t xv = x[q];
          if (xv < 0) continue;
          // the adjacency is symmetric (ABI contract): every hit also sets its mirror entry, which is how the
          // rows of hubs get filled without ever touching their lists
          const uint32_t ibit = 1u << (i & 31);
          if (xv == u && i != 1) { atomicOr(&adj[i][0], 1u); if (!directed) atomicOr(&adj[0][i >> 5], ibit); }   // (1,0) is a target entry
          if (xv == v && i != 0) { atomicOr(&adj[i][0], 2u); if (!directed) atomicOr(&adj[1][i >> 5], ibit); }   // (0,1) likewise
          if (xv != u && xv != v) {
            const uint32_t hsh = (static_cast<uint32_t>(xv) * 0x9E3779B1u) >> 20;
            if (!((member[hsh >> 5] >> (hsh & 31)) & 1u)) continue;
            int lo = 2, hi = n;
            while (lo < hi) {
              const int mid = (lo + hi) >> 1;
              if (nodes[mid] < xv) lo = mid + 1; else hi = mid;
            }
            if (lo < n && nodes[lo] == xv) {
              atomicOr(&adj[i][lo >> 5], 1u << (lo & 31));
              if (!directed) atomicOr(&adj[lo][i >> 5], ibit);
            }
          }
        }
      }
    }
    SEAL_PHASE(2);
    // what is left: hub x hub entries (neither list was streamed), or everything when a node list is replayed
    // A thread runs TWO binary searches in lockstep (the probes of a step are independent loads; four-way cost
    // a wave of occupancy in registers and lost more on the common small subgraphs than it won on the hub-rich ones): a hub-rich
    // subgraph has ~30 hubs = 1000 searches of ~12 dependent probes each, 8 per thread back to back before.
    // Without a replayed list only a <= b of the (symmetric) hub x hub block is searched and mirrored.
    const bool allcols = replay || directed;          // no symmetry to lean on: a big row is searched against every column
    const int nbig = s_nbig, ncol = allcols ? n : nbig;
    // With the graph's hub bit matrix (hub_index[node] = its row, -1 for a node that is not a hub; one bit per hub
    // pair) a hub x hub entry is two index loads and one word instead of a ~12-probe binary search in a list of
    // thousands.  A big row whose node has no hub index (tables built with another threshold) falls back to the search.
    bool search = true;
    if (hub_index && !allcols) {
      if (t == 0) s_search = 0;
      if (t < nbig) hidx[t] = hub_index[nodes[big[t]]];           // one load per big row, all in flight together
      __syncthreads();
      // Only the upper triangle a <= b of the (symmetric) hub x hub block is looked up and mirrored: entry p of the
      // nbig (nbig + 1) / 2 is unranked with one v_sqrt_f32 (round 2 walked all nbig^2 index pairs with an integer
      // division each and dropped half of them: this phase was 50-60 % of a team's time, profiles/seal_phases.py), and
      // HW independent bit-matrix words are in flight per thread.
      constexpr int HW = 8;
      const int np = nbig * (nbig + 1) / 2;
      const float fn = 2.0f * static_cast<float>(nbig) + 1.0f;
      for (int p0 = t; p0 < np; p0 += HW * kSealBlock) {
        uint32_t wv[HW];
        int ij[HW];                                    // i | j << 8 | ok << 16
#pragma unroll
        for (int q = 0; q < HW; ++q) {
          const int p = p0 + q * kSealBlock;
          const bool valid = p < np;
          const int pc = valid ? p : 0;
          int r = static_cast<int>((fn - sqrtf(fn * fn - 8.0f * static_cast<float>(pc))) * 0.5f);
          r = r < 0 ? 0 : (r >= nbig ? nbig - 1 : r);
          int off = r * nbig - ((r * (r - 1)) >> 1);
          if (off > pc) { --r; off = r * nbig - ((r * (r - 1)) >> 1); }
          else if (off + (nbig - r) <= pc) { off += nbig - r; ++r; }
          const int b2 = r + (pc - off);
          const int ia = hidx[r], ib = hidx[b2];
          if (valid && (ia < 0 || ib < 0)) s_search = 1;
          const bool ok = valid && ia >= 0 && ib >= 0;
          wv[q] = ok ? hub_bits[static_cast<int64_t>(ia) * hub_words + (ib >> 5)] : 0u;
          ij[q] = big[r] | (big[b2] << 8) | ((ok ? (1 + (ib & 31)) : 0) << 16);
        }
#pragma unroll
        for (int q = 0; q < HW; ++q) {
          const int i = ij[q] & 0xFF, j = (ij[q] >> 8) & 0xFF, sh = ij[q] >> 16;
          if (sh && ((wv[q] >> (sh - 1)) & 1u) && !((i == 0 && j == 1) || (i == 1 && j == 0))) {
            atomicOr(&adj[i][j >> 5], 1u << (j & 31));
            atomicOr(&adj[j][i >> 5], 1u << (i & 31));
          }
        }
      }
      __syncthreads();
      search = s_search != 0;
    }
    SEAL_PHASE(3);
    for (int p0 = t; search && p0 < nbig * ncol; p0 += 2 * kSealBlock) {
      int lo[2], hi[2], len[2], ii[2], jj[2];
      int32_t key[2];
      const int32_t* row[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int p = p0 + q * kSealBlock;
        const bool valid = p < nbig * ncol;
        const int a = valid ? p / ncol : 0, b2 = valid ? p % ncol : 0;
        ii[q] = big[a];
        jj[q] = allcols ? b2 : big[b2];
        const bool take = valid && (allcols || a <= b2);
        row[q] = col + rbase[ii[q]];
        len[q] = take ? rdeg[ii[q]] : 0;
        key[q] = nodes[jj[q]];
        lo[q] = 0;
        hi[q] = len[q];
      }
      bool any = true;
      while (any) {
        int32_t v[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) v[q] = row[q][lo[q] < hi[q] ? (lo[q] + hi[q]) >> 1 : 0];
        any = false;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          if (lo[q] < hi[q]) {
            const int mid = (lo[q] + hi[q]) >> 1;
            if (v[q] < key[q]) lo[q] = mid + 1; else hi[q] = mid;
            any |= lo[q] < hi[q];
          }
        }
      }
      int32_t fv[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) fv[q] = row[q][lo[q] < len[q] ? lo[q] : 0];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int i = ii[q], j = jj[q];
        if (lo[q] < len[q] && fv[q] == key[q] && !((i == 0 && j == 1) || (i == 1 && j == 0))) {
          atomicOr(&adj[i][j >> 5], 1u << (j & 31));
          if (!allcols) atomicOr(&adj[j][i >> 5], 1u << (i & 31));
        }
      }
    }
    __syncthreads();

    SEAL_PHASE(4);
    // ---- DRNL (utils.py:35-54): two BFS sweeps on the bit matrix
    static_assert(kSealBlock == 128 && MAXN == 128 && WORDS == 4, "bfs_team_waves: two waves, 128-bit rows");
    bfs_team_waves(adj, n, dist0, dist1);
    __syncthreads();
    SEAL_PHASE(5);
    if (t < MAXN) {
      int tag = -1, nd = -1;
      if (t < n) {
        nd = nodes[t];
        if (t < 2) {
          tag = 1;
        } else {
          const int dx = dist0[t], dy = dist1[t];
          if (dx < 0 || dy < 0) {
            tag = 0;
          } else {
            const int d = dx + dy, dv2 = d >> 1, md = d & 1;
            tag = 1 + (dx < dy ? dx : dy) + dv2 * (dv2 + md - 1);
          }
        }
      }
      tags_out[b * MAXN + t] = tag;
      if (tag_present && tag >= 0) tag_present[tag] = 1;      // presence table for the global tag compaction (same value from every writer)
      if (!replay) nodes_out[b * MAXN + t] = nd;
#pragma unroll
      for (int w = 0; w < WORDS; ++w) adj_out[(b * MAXN + t) * WORDS + w] = t < n ? adj[t][w] : 0u;
    }
    if (t == 0) { n_out[b] = n; capped_out[b] = static_cast<uint8_t>(was_capped); }
    SEAL_PHASE(6);
  }
#ifdef LPG_TUNING
  if (t == 0 && ph_acc[7])
    for (int k = 0; k < 8; ++k) atomicAdd(&g_seal_phase[k], ph_acc[k]);
#endif
}

// ---- block-diagonal CSR of A~ = A + I --------------------------------------------------------
// global vertex g = node_off[b] + i.  Row length = popcount(adj row) + (diagonal clear ? 1 : 0);
// the diagonal value is A[i][i] + 1 (uint8 add at SEAL/config/data.py:94-95), rowsum = popcount + 1.
__device__ __forceinline__ int64_t graph_of(const int64_t* __restrict__ node_off, int64_t B, int64_t g) {
  int64_t lo = 0, hi = B;   // largest b with node_off[b] <= g
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (node_off[mid] <= g) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(kBlock) void seal_rowlen_kernel(const int64_t* __restrict__ node_off,
                                                             const uint32_t* __restrict__ adj_bits, int64_t B,
                                                             int64_t total, int64_t* __restrict__ len,
                                                             float* __restrict__ d_inv) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; g <= total; g += stride) {
    if (g == total) { len[g] = 0; continue; }
    const int64_t b = graph_of(node_off, B, g);
    const int i = static_cast<int>(g - node_off[b]);
    const uint32_t* row = adj_bits + (b * MAXN + i) * WORDS;
    int pc = 0;
#pragma unroll
    for (int w = 0; w < WORDS; ++w) pc += __popc(row[w]);
    const bool diag = (row[i >> 5] >> (i & 31)) & 1u;
    len[g] = pc + (diag ? 0 : 1);
    d_inv[g] = static_cast<float>(1.0 / static_cast<double>(pc + 1));
  }
}

__global__ __launch_bounds__(kBlock) void seal_fill_kernel(const int64_t* __restrict__ node_off,
                                                           const uint32_t* __restrict__ adj_bits, int64_t B,
                                                           int64_t total, const int64_t* __restrict__ rowptr,
                                                           int32_t* __restrict__ col, float* __restrict__ val,
                                                           int64_t e_cap, int fold_dinv) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; g < total; g += stride) {
    const int64_t b = graph_of(node_off, B, g);
    const int64_t base = node_off[b];
    const int i = static_cast<int>(g - base);
    const uint32_t* row = adj_bits + (b * MAXN + i) * WORDS;
    int64_t k = rowptr[g];
    bool diag_done = false;
    float one = 1.f;                 // fold_dinv: the stored values are D^-1 A~ (row scale folded into the edges)
    if (fold_dinv) {
      int pc = 0;
#pragma unroll
      for (int w = 0; w < WORDS; ++w) pc += __popc(row[w]);
      one = static_cast<float>(1.0 / static_cast<double>(pc + 1));
    }
    const float two = 2.f * one;
    for (int w = 0; w < WORDS; ++w) {
      uint32_t m = row[w];
      while (m) {
        const int j = (w << 5) + __ffs(m) - 1;
        m &= m - 1;
        if (!diag_done && j > i) {                  // diagonal not set in A: insert the +I entry in order
          if (k < e_cap) { col[k] = static_cast<int32_t>(base + i); val[k] = one; }
          ++k;
          diag_done = true;
        }
        if (k < e_cap) { col[k] = static_cast<int32_t>(base + j); val[k] = (j == i) ? two : one; }
        if (j == i) diag_done = true;
        ++k;
      }
    }
    if (!diag_done) {
      if (k < e_cap) { col[k] = static_cast<int32_t>(base + i); val[k] = one; }
    }
  }
}

// One workgroup walks whole subgraphs (grid-stride over b), a wave takes one vertex row at a time and its
// lanes the columns: the one-hot part is computed, the embedding part is a coalesced copy of emb[node]
// (512 B for dim 128).  No per-element division or offset search (the first version spent 2.9 ms on 0.8 GB).
template <typename TO> __device__ __forceinline__ TO feat_out(float v);
template <> __device__ __forceinline__ float feat_out<float>(float v) { return v; }
template <> __device__ __forceinline__ uint16_t feat_out<uint16_t>(float v) { return f32_to_bf16(v); }   // RNE, like torch

template <typename TO>
__global__ __launch_bounds__(kBlock) void seal_features_kernel(const int32_t* __restrict__ nodes,
                                                               const int64_t* __restrict__ node_off,
                                                               const int32_t* __restrict__ tags,
                                                               const int32_t* __restrict__ tag_map, int64_t tag_map_len,
                                                               int64_t T, const float* __restrict__ emb, int64_t emb_dim,
                                                               int64_t B, TO* __restrict__ x, int64_t ldx,
                                                               int wstore, int32_t* err_flag) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int width = static_cast<int>(T + emb_dim);     // wstore >= width: columns [width, wstore) are zero-filled
  constexpr int R = 4;                               // rows per wave in flight: the chain node id -> emb row -> store
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {   // is three dependent round trips, so rows are batched
    const int64_t base = node_off[b];
    int n = static_cast<int>(node_off[b + 1] - base);
    if (n > MAXN) n = MAXN;
    for (int i0 = wave * R; i0 < n; i0 += (kBlock / 64) * R) {
      // every load below comes from a clamped, always-valid address and is issued before anything looks at
      // a value (compiler barrier): written as `cond ? load : x` hipcc branches around each load and waits for
      // it -- 12 serialized round trips per 4 rows in the first version of this loop
      int32_t tg[R], nd[R];
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const int i = i0 + q < n ? i0 + q : n - 1;   // clamped: the duplicate is not stored
        tg[q] = tags[b * MAXN + i];
        nd[q] = nodes[b * MAXN + i];
      }
      asm volatile("" ::: "memory");
      if (tag_map && tag_map_len <= 0) {             // uniform: an empty map maps nothing
#pragma unroll
        for (int q = 0; q < R; ++q) tg[q] = -1;
        if (lane == 0) flag_error(err_flag, LPG_F_INDEX_RANGE);
      } else if (tag_map) {
        int32_t tm[R];
#pragma unroll
        for (int q = 0; q < R; ++q) tm[q] = tag_map[(tg[q] >= 0 && tg[q] < tag_map_len) ? tg[q] : 0];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int q = 0; q < R; ++q) {
          const bool ok = tg[q] >= 0 && tg[q] < tag_map_len;
          if (!ok && lane == 0 && i0 + q < n) flag_error(err_flag, LPG_F_INDEX_RANGE);
          tg[q] = ok ? tm[q] : -1;
        }
      }
      const float* ebase = emb_dim > 0 ? emb : reinterpret_cast<const float*>(tags);   // emb_dim == 0: any valid address, never used
      for (int c0 = 0; c0 < wstore; c0 += 192) {     // 3 columns x R rows in flight per lane
        float v[R][3];
#pragma unroll
        for (int q = 0; q < R; ++q)
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            const int c = c0 + u * 64 + lane;
            const bool is_emb = c >= T && c < width;
            v[q][u] = ebase[is_emb ? static_cast<int64_t>(nd[q]) * emb_dim + (c - static_cast<int>(T)) : 0];
          }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int q = 0; q < R; ++q)
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            const int c = c0 + u * 64 + lane;
            const bool is_emb = c >= T && c < width;
            v[q][u] = is_emb ? v[q][u] : ((tg[q] == c) ? 1.f : 0.f);
          }
#pragma unroll
        for (int q = 0; q < R; ++q) {
          if (i0 + q >= n) continue;
          TO* xrow = x + (base + i0 + q) * ldx;
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            const int c = c0 + u * 64 + lane;
            if (c < wstore) xrow[c] = feat_out<TO>(v[q][u]);
          }
        }
      }
    }
  }
}
// ---- layer 1 without materialising X -------------------------------------------------------------------
// H_1 = X W_1 with X = [onehot(tag) || emb[node]]  (SEAL/config/data.py:129-140 builds X, SEAL/model/gnn.py:46
// multiplies it).  A one-hot row times W_1 is a ROW of W_1, and emb[node] W_1[T:] is row `node` of the table
// P = emb W_1[T:] (one small matmul per (emb, W_1), |V| x 32).  So  H_1[g] = W_1[tag_g] + P[node_g]:  two 128-byte
// gathers and an add per vertex instead of writing 544 B of X and reading it back through a GEMM
// (features 0.46 ms + linear 0.22 ms -> 0.1 ms per 100 K candidates).  Same sum, different association.
constexpr int kH1Block = 256;    // ONE wave per subgraph (8 rows per step cover the typical subgraph in two), four subgraphs per workgroup:
                                 // 100 K one-wave workgroups were bound by the dispatcher
template <typename TOUT>
__global__ __launch_bounds__(kH1Block) void seal_h1_tables_kernel(const int32_t* __restrict__ nodes,
                                                               const int64_t* __restrict__ node_off,
                                                               const int32_t* __restrict__ tags,
                                                               const int32_t* __restrict__ tag_map, int64_t tag_map_len,
                                                               int64_t T, const float* __restrict__ w1_tag,
                                                               const float* __restrict__ p_table, int64_t n_graph,
                                                               int64_t B, TOUT* __restrict__ h1, int64_t ldh,
                                                               int32_t* err_flag) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c4 = (lane & 7) * 4, sub = lane >> 3;   // 8 lanes x float4 cover a 32-wide row; 8 rows per wave step
  for (int64_t b = static_cast<int64_t>(blockIdx.x) * (kH1Block / 64) + wave; b < B; b += static_cast<int64_t>(gridDim.x) * (kH1Block / 64)) {
    const int64_t base = node_off[b];
    int n = static_cast<int>(node_off[b + 1] - base);
    if (n > MAXN) n = MAXN;
    for (int i0 = 0; i0 < n; i0 += 8) {
      const int i = i0 + sub;
      const int ic = i < n ? i : n - 1;              // clamped: the duplicate is not stored
      int32_t tg = tags[b * MAXN + ic];
      const int32_t nd = nodes[b * MAXN + ic];
      asm volatile("" ::: "memory");
      bool ok = true;
      if (tag_map) {
        ok = tg >= 0 && tg < tag_map_len;
        tg = tag_map[ok ? tg : 0];
        asm volatile("" ::: "memory");
      }
      ok = ok && tg >= 0 && tg < T && nd >= 0 && nd < n_graph;
      if (!ok && i < n && c4 == 0) flag_error(err_flag, LPG_F_INDEX_RANGE);
      const float4 a = *reinterpret_cast<const float4*>(w1_tag + static_cast<int64_t>(ok ? tg : 0) * 32 + c4);
      const float4 q = *reinterpret_cast<const float4*>(p_table + static_cast<int64_t>(ok ? nd : 0) * 32 + c4);
      const float4 sum = ok ? make_float4(a.x + q.x, a.y + q.y, a.z + q.z, a.w + q.w) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < n) {
        if constexpr (sizeof(TOUT) == 4) {
          *reinterpret_cast<float4*>(h1 + (base + i) * ldh + c4) = sum;
        } else {                                       // bf16 storage: ONE rounding of the fp32 sum (nearest even)
          *reinterpret_cast<uint2*>(h1 + (base + i) * ldh + c4) =
              make_uint2(f32_to_bf16(sum.x) | (static_cast<uint32_t>(f32_to_bf16(sum.y)) << 16),
                         f32_to_bf16(sum.z) | (static_cast<uint32_t>(f32_to_bf16(sum.w)) << 16));
        }
      }
    }
  }
}

// ---- SortPooling: one workgroup per subgraph; rank sort of <= 128 keys in LDS -------------------
__global__ __launch_bounds__(kSealBlock) void seal_sortpool_kernel(const float* __restrict__ z, int64_t ldz,
                                                                   const float* __restrict__ key,
                                                                   const int64_t* __restrict__ node_off, int64_t B,
                                                                   int64_t k, int64_t C, float* __restrict__ out, int vec) {
  __shared__ float keys[MAXN];
  __shared__ int order[MAXN];     // order[rank] = vertex
  const int t = threadIdx.x;
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    const int64_t base = node_off[b];
    int n = static_cast<int>(node_off[b + 1] - base);
    if (n > MAXN) n = MAXN;
    if (t < n) keys[t] = key[base + t];
    __syncthreads();
    if (t < n) {
      const float mine = keys[t];
      const bool mine_nan = mine != mine;
      int rank = 0;                // number of vertices that come before t: larger key, or equal key and lower index
      for (int j = 0; j < n; ++j) {
        const float o = keys[j];
        // a TOTAL order even with NaN keys (NaN sorts last, ties by index): ranks must stay a permutation, `order`
        // indexes global memory below (a NaN key once made two threads claim rank 0 and left slots unwritten)
        const bool o_nan = o != o;
        const bool tie = (o == mine) || (o_nan && mine_nan);
        rank += (o > mine) || (mine_nan && !o_nan) || (tie && j < t);
      }
      order[rank] = t;
    }
    __syncthreads();
    if (vec) {                                       // C % 4 == 0, 16-byte aligned rows: whole float4 pieces
      const int C4 = static_cast<int>(C) / 4, total = static_cast<int>(k) * C4;
      float4* ob = reinterpret_cast<float4*>(out + b * k * C);
      for (int q0 = t; q0 < total; q0 += 4 * kSealBlock) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int q = q0 + u * kSealBlock, r = q / C4, c = q - r * C4;
          const bool ok = q < total && r < n;
          const float4 t4 = *reinterpret_cast<const float4*>(z + (base + (ok ? order[r] : 0)) * ldz + (ok ? c * 4 : 0));
          v[u] = ok ? t4 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int q = q0 + u * kSealBlock;
          if (q < total) ob[q] = v[u];
        }
      }
    } else {
      const int total = static_cast<int>(k * C), Ci = static_cast<int>(C);
      for (int q = t; q < total; q += kSealBlock) {
        const int r = q / Ci, c = q - r * Ci;
        out[b * k * C + q] = r < n ? z[(base + order[r]) * ldz + c] : 0.f;
      }
    }
  }
}

// One WAVE per subgraph (four per workgroup), no workgroup barrier: 100 K workgroups of 128 threads were bound by the
// dispatcher, not by the 18 rows a subgraph contributes (0.100 -> 0.05 ms per 100 K subgraphs).  Same total order as
// above (key descending, NaN last, ties by index); C % 4 == 0, 16-byte aligned rows.
__global__ __launch_bounds__(kBlock) void seal_sortpool_wave_kernel(const float* __restrict__ z, int64_t ldz,
                                                                    const float* __restrict__ key,
                                                                    const int64_t* __restrict__ node_off, int64_t B,
                                                                    int k, int C4, float* __restrict__ out) {
  __shared__ float keys[kBlock / 64][MAXN];
  __shared__ int order[kBlock / 64][MAXN];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t b = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + wave; b < B; b += static_cast<int64_t>(gridDim.x) * (kBlock / 64)) {
    const int64_t base = node_off[b];
    int n = static_cast<int>(node_off[b + 1] - base);
    if (n > MAXN) n = MAXN;
    float mine[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int t = lane + 64 * q;
      mine[q] = t < n ? key[base + t] : 0.f;
      if (t < n) keys[wave][t] = mine[q];
    }
    wave_lds_sync();
    int rank[2] = {0, 0};
    for (int j = 0; j < n; ++j) {
      const float o = keys[wave][j];                     // the same address in every lane: an LDS broadcast
      const bool o_nan = o != o;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const bool mine_nan = mine[q] != mine[q];
        const bool tie = (o == mine[q]) || (o_nan && mine_nan);
        rank[q] += (o > mine[q]) || (mine_nan && !o_nan) || (tie && j < lane + 64 * q);
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (lane + 64 * q < n) order[wave][rank[q]] = lane + 64 * q;
    wave_lds_sync();
    const int total = k * C4;
    float4* ob = reinterpret_cast<float4*>(out) + b * total;
    for (int q0 = lane; q0 < total; q0 += 4 * 64) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = q0 + u * 64, r = q / C4, c = q - r * C4;
        const bool ok = q < total && r < n;
        const float4 t4 = *reinterpret_cast<const float4*>(z + (base + (ok ? order[wave][r] : 0)) * ldz + (ok ? c * 4 : 0));
        v[u] = ok ? t4 : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = q0 + u * 64;
        if (q < total) ob[q] = v[u];
      }
    }
    wave_lds_sync();                                     // keys / order are rewritten by the wave's next subgraph
  }
}

__global__ __launch_bounds__(kBlock) void softmax2_kernel(const float* __restrict__ pre_y, int64_t B,
                                                          float* __restrict__ score) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t b = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; b < B; b += stride) {
    const float y0 = pre_y[2 * b], y1 = pre_y[2 * b + 1];
    const float m = fmaxf(y0, y1);
    const float e0 = expf(y0 - m), e1 = expf(y1 - m);
    score[b] = e1 / (e0 + e1);
  }
}

// ---- the head's last layer and the softmax in one launch (SEAL/model/gnn.py:127-137): pre_y = h W + b with W [H, 2]
// (TensorFlow's x @ W), score = softmax(pre_y)[1].  16 lanes per row, float4 pieces, xor-shuffle tree (fixed order).
__global__ __launch_bounds__(kBlock) void seal_logits_kernel(const float* __restrict__ h, int64_t ldh, int64_t B, int H,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ pre_y, float* __restrict__ score) {
  const int lane16 = threadIdx.x & 15;
  const int64_t grp = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) >> 4;
  const int64_t ngrp = (static_cast<int64_t>(gridDim.x) * kBlock) >> 4;
  for (int64_t b0 = grp; b0 < B + (ngrp - (B % ngrp)) % ngrp; b0 += ngrp) {   // whole waves stay in step for the shuffles
    const bool on = b0 < B;
    const int64_t b = on ? b0 : B - 1;
    float y0 = 0.f, y1 = 0.f;
    for (int c = 4 * lane16; c < H; c += 64) {
      const float4 v = *reinterpret_cast<const float4*>(h + b * ldh + c);
      const float4 wa = *reinterpret_cast<const float4*>(w + 2 * c);          // (c, 0) (c, 1) (c+1, 0) (c+1, 1)
      const float4 wb = *reinterpret_cast<const float4*>(w + 2 * c + 4);
      y0 += v.x * wa.x + v.y * wa.z + v.z * wb.x + v.w * wb.z;
      y1 += v.x * wa.y + v.y * wa.w + v.z * wb.y + v.w * wb.w;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      y0 += __shfl_xor(y0, o, 64);
      y1 += __shfl_xor(y1, o, 64);
    }
    if (on && lane16 == 0) {
      if (bias) { y0 += bias[0]; y1 += bias[1]; }
      pre_y[2 * b] = y0;
      pre_y[2 * b + 1] = y1;
      const float m = fmaxf(y0, y1);
      const float e0 = expf(y0 - m), e1 = expf(y1 - m);
      score[b] = e1 / (e0 + e1);
    }
  }
}

// ---- batch index: node_off = exclusive scan of the subgraph sizes, the tag map = ranks of the present tags, and
// one 4-word head {total vertices, error flag, distinct tags, 0} for the batch's single host read.  One launch, two
// workgroups (a dozen tiny scan / elementwise launches before: 40 of the 210 us of fb-pages-food's structure stage).
constexpr int kIdxBlock = 1024;
constexpr int64_t kIdxOneBlockMax = int64_t(1) << 14;     // beyond: the sizes are scanned by rocPRIM first (one workgroup takes ~2 us
                                                          // per 4096 sizes, serially: 55 us at 100 K subgraphs, and the whole forward waits for the total)

__device__ __forceinline__ int64_t wave_incl_scan(int64_t v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int64_t up = __shfl_up(v, o, 64);
    if (lane >= o) v += up;
  }
  return v;
}

// inclusive block scan of one value per thread on top of `carry`; returns the thread's EXCLUSIVE prefix, updates carry
__device__ __forceinline__ int64_t block_excl_scan(int64_t mine, int64_t& carry, int64_t* wsum, int t) {
  const int lane = t & 63, wv = t >> 6;
  const int64_t incl = wave_incl_scan(mine, lane);
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  int64_t before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kIdxBlock / 64; ++w) {
    const int64_t x = wsum[w];
    before += w < wv ? x : 0;
    all += x;
  }
  const int64_t excl = carry + before + incl - mine;
  carry += all;
  __syncthreads();                                 // wsum is reused by the next tile
  return excl;
}

__global__ __launch_bounds__(kIdxBlock) void seal_batch_index_kernel(const int32_t* __restrict__ n, int64_t B,
                                                                     int64_t* __restrict__ off, int scan_sizes,
                                                                     const uint8_t* __restrict__ present, int K,
                                                                     int32_t* __restrict__ tag_map,
                                                                     const int32_t* __restrict__ err_flag,
                                                                     int64_t* __restrict__ head,
                                                                     int32_t* __restrict__ first, int64_t win) {
  __shared__ int64_t wsum[kIdxBlock / 64];
  const int t = threadIdx.x;
  int64_t carry = 0;
  if (blockIdx.x == 0) {
    if (scan_sizes) {
      for (int64_t base = 0; base < B; base += 4 * kIdxBlock) {
        const int64_t i0 = base + 4 * t;
        int v[4] = {0, 0, 0, 0};
        if (i0 + 3 < B) {
          const int4 q = *reinterpret_cast<const int4*>(n + i0);
          v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = i0 + j < B ? n[i0 + j] : 0;
        }
        int64_t run = block_excl_scan(static_cast<int64_t>(v[0]) + v[1] + v[2] + v[3], carry, wsum, t);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          run += v[j];
          if (i0 + j < B) off[i0 + j + 1] = run;
        }
      }
      if (t == 0) off[0] = 0;
    } else {
      carry = off[B];                                // scanned by the caller (rocPRIM) on the same stream
    }
    if (t == 0) {
      head[0] = carry;
      head[1] = err_flag ? err_flag[0] : 0;
      if (!present) head[2] = head[3] = 0;
    }
    if (first) {
      // the conv stack's row windows (lpg_seal_conv_windows): first[w] = the first subgraph whose first row is >= w * win,
      // for w = 0 .. ceil(total / win); node_off was written by this workgroup (or before the launch)
      __threadfence_block();
      __syncthreads();
      const int64_t nw = (carry + win - 1) / win;
      for (int64_t w = t; w <= nw; w += kIdxBlock) {
        const int64_t target = w * win;
        int64_t lo = 0, hi = B + 1;
        while (lo < hi) {
          const int64_t mid = (lo + hi) >> 1;
          if (off[mid] < target) lo = mid + 1; else hi = mid;
        }
        first[w] = static_cast<int32_t>(lo < B ? lo : B);
      }
    }
  } else if (present) {
    // 16 presence bytes per thread: the whole DRNL tag range (16 258 entries) is ONE tile, one block scan
    __shared__ int smax;
    int mx = -1;                                       // largest present tag (head[3]): the host's test for whether
    if (t == 0) smax = -1;                             // CPython's set order can differ from ascending order
    for (int base = 0; base < K; base += 16 * kIdxBlock) {
      const int i0 = base + 16 * t;
      uint32_t w[4] = {0u, 0u, 0u, 0u};
      if (i0 + 15 < K && (reinterpret_cast<uintptr_t>(present) & 15) == 0) {
        const uint4 q = *reinterpret_cast<const uint4*>(present + i0);
        w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
      } else {
        for (int j = 0; j < 16; ++j)
          if (i0 + j < K) w[j >> 2] |= static_cast<uint32_t>(present[i0 + j]) << (8 * (j & 3));
      }
      int cnt = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) cnt += ((w[j >> 2] >> (8 * (j & 3))) & 0xffu) != 0;
      int64_t rank = block_excl_scan(cnt, carry, wsum, t);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const bool p = ((w[j >> 2] >> (8 * (j & 3))) & 0xffu) != 0;
        if (i0 + j < K) tag_map[i0 + j] = p ? static_cast<int32_t>(rank) : -1;
        if (p && i0 + j < K) mx = i0 + j;
        rank += p;
      }
    }
    __syncthreads();
    if (mx >= 0) atomicMax(&smax, mx);
    __syncthreads();
    if (t == 0) {
      head[2] = carry;
      head[3] = smax;
    }
  }
}

struct I32ToI64 {
  __device__ int64_t operator()(int32_t v) const { return v; }
};
using SizeIt = rocprim::transform_iterator<const int32_t*, I32ToI64, int64_t>;
}  // namespace

LPG_EXPORT int lpg_seal_subgraphs(const int64_t* adj_rowptr, const int32_t* adj_col, int64_t n_graph_nodes,
                                  const int64_t* pairs, int64_t B, int hop, int max_neighbors, int replay,
                                  int cap_policy, uint64_t seed, int32_t* nodes, int32_t* n_nodes, int32_t* tags, uint32_t* adj_bits, uint8_t* capped,
                                  uint8_t* tag_present, const int32_t* hub_index, const uint32_t* hub_bits,
                                  int64_t hub_words, int64_t hub_degree, int directed, int32_t* err_flag, void* stream) {
  LPG_REQUIRE(B >= 0 && n_graph_nodes >= 0, "seal_subgraphs: bad sizes");
  LPG_REQUIRE(hop == 0 || hop == 1, "seal_subgraphs: hop must be 0 or 1 (the reference itself fails for hop >= 2)");
  LPG_REQUIRE(max_neighbors >= 2 && max_neighbors <= MAXN, "seal_subgraphs: max_neighbors must be in [2, %d]", MAXN);
  LPG_REQUIRE(hub_degree >= 1 && hub_degree < (int64_t(1) << 30), "seal_subgraphs: hub_degree must be >= 1");
  if (B == 0) return LPG_OK;
  LPG_REQUIRE(adj_rowptr && pairs && nodes && n_nodes && tags && adj_bits && capped, "seal_subgraphs: null pointer");
  hipStream_t st = as_stream(stream);
  const int32_t* hidx = hub_bits ? hub_index : nullptr;
  const bool fused = !replay && !directed && options().seal_pair_small && B < (int64_t(1) << 30) &&
                     reinterpret_cast<uintptr_t>(adj_bits) % 16 == 0;
  // fused: ceil(B / 8) workgroups for the pairs that need 128 threads, then one WAVE per pair (two per workgroup) for
  // the rest -- most of a batch.  Otherwise a workgroup per pair (the hardware dispatcher is the work queue: a static
  // share of ~24 pairs per persistent workgroup left the chip waiting for the unluckiest one, 0.755 vs 0.581 ms).
  // team workgroups own 8 consecutive pairs on big batches (1/8 of the dispatches; most own no heavy pair at all), one pair
  // on small ones (fb-pages-food's 4202 candidates are mostly heavy: 8 in series per workgroup left the chip idle)
  const int group = B >= (int64_t(1) << 15) ? kDeferGroup : 1;
  const int64_t grid = fused ? ceil_div(B, group) + ceil_div(B, kSealBlock / 64) : (B < (int64_t(1) << 24) ? B : (int64_t(1) << 24));
  hipLaunchKernelGGL(seal_subgraph_kernel, dim3(static_cast<unsigned>(grid)), dim3(kSealBlock), 0, st, adj_rowptr, adj_col,
                     n_graph_nodes, pairs, B, hop, max_neighbors, replay, cap_policy, seed, nodes, n_nodes, tags, adj_bits, capped, err_flag,
                     tag_present, directed ? nullptr : hidx, hub_bits, static_cast<int>(hub_words), fused ? group : 0, directed, static_cast<int>(hub_degree));
  return check_launch("seal_subgraphs");
}

LPG_EXPORT int lpg_seal_phase_ticks(uint64_t* out8_host, int reset) {
#ifdef LPG_TUNING
  LPG_REQUIRE(out8_host, "seal_phase_ticks: null pointer");
  LPG_HIP(hipDeviceSynchronize());
  LPG_HIP(hipMemcpyFromSymbol(out8_host, HIP_SYMBOL(g_seal_phase), 8 * sizeof(unsigned long long)));
  if (reset) {
    const unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    LPG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_seal_phase), zero, sizeof(zero)));
  }
  return LPG_OK;
#else
  (void)out8_host; (void)reset;
  set_error("seal_phase_ticks: only in a -DLPG_TUNING build of the library");
  return LPG_E_UNSUPPORTED;
#endif
}

LPG_EXPORT int64_t lpg_seal_block_csr_bytes(int64_t total_nodes) {
  if (total_nodes < 0) return LPG_E_INVALID;
  size_t s = 0;
  (void)rocprim::exclusive_scan(nullptr, s, static_cast<int64_t*>(nullptr), static_cast<int64_t*>(nullptr), int64_t(0),
                                static_cast<size_t>(total_nodes + 1), rocprim::plus<int64_t>(), hipStream_t(0), false);
  return lpg::align_up((total_nodes + 2) * 8, 256) + lpg::align_up(static_cast<int64_t>(s), 256) + 256;
}

// Phase 1 (col == NULL): rowptr[total+1] and d_inv[total].  Phase 2 (col != NULL): fill col/val.
LPG_EXPORT int lpg_seal_block_csr(const int32_t* n_nodes, const int64_t* node_off, const uint32_t* adj_bits, int64_t B,
                                  int64_t* rowptr, int32_t* col, float* val, float* d_inv, int64_t e_cap,
                                  int fold_dinv, void* workspace, int64_t workspace_bytes, void* stream) {
  LPG_REQUIRE(B >= 0 && node_off && rowptr, "seal_block_csr: bad argument");
  (void)n_nodes;
  hipStream_t st = as_stream(stream);
  int64_t total = 0;
  LPG_HIP(hipMemcpyAsync(&total, node_off + B, 8, hipMemcpyDeviceToHost, st));
  LPG_HIP(hipStreamSynchronize(st));
  LPG_REQUIRE(total >= 0 && total < (int64_t(1) << 31), "seal_block_csr: bad node_off");
  if (!col) {
    LPG_REQUIRE(d_inv && adj_bits && workspace, "seal_block_csr: null pointer");
    if (workspace_bytes < lpg_seal_block_csr_bytes(total)) {
      set_error("seal_block_csr: workspace too small");
      return LPG_E_WORKSPACE;
    }
    int64_t* len = static_cast<int64_t*>(workspace);
    void* tmp = static_cast<char*>(workspace) + align_up((total + 2) * 8, 256);
    size_t tb = static_cast<size_t>(workspace_bytes - align_up((total + 2) * 8, 256));
    hipLaunchKernelGGL(seal_rowlen_kernel, dim3(grid_for(total + 1)), dim3(kBlock), 0, st, node_off, adj_bits, B, total, len, d_inv);
    LPG_HIP(rocprim::exclusive_scan(tmp, tb, len, rowptr, int64_t(0), static_cast<size_t>(total + 1),
                                    rocprim::plus<int64_t>(), st, false));
  } else {
    LPG_REQUIRE(val && adj_bits, "seal_block_csr: null pointer");
    if (total > 0)
      hipLaunchKernelGGL(seal_fill_kernel, dim3(grid_for(total)), dim3(kBlock), 0, st, node_off, adj_bits, B, total, rowptr, col, val, e_cap, fold_dinv);
  }
  return check_launch("seal_block_csr");
}

namespace {
template <typename TO>
int seal_features_entry(const int32_t* nodes, const int64_t* node_off, const int32_t* tags, const int32_t* tag_map,
                        int64_t tag_map_len, int64_t T, const float* emb, int64_t emb_dim, int64_t B, TO* x, int64_t ldx,
                        int zero_pad, int32_t* err_flag, void* stream) {
  LPG_REQUIRE(B >= 0 && T >= 0 && emb_dim >= 0 && ldx >= T + emb_dim, "seal_features: bad sizes");
  if (B == 0 || T + emb_dim == 0) return LPG_OK;
  LPG_REQUIRE(nodes && node_off && tags && x && (emb_dim == 0 || emb), "seal_features: null pointer");
  LPG_REQUIRE(ldx < (int64_t(1) << 30), "seal_features: row too wide");
  const int wstore = static_cast<int>(zero_pad ? ldx : T + emb_dim);
  const int grid = static_cast<int>(B < (int64_t(1) << 24) ? B : (int64_t(1) << 24));   // a workgroup per subgraph (see lpg_seal_subgraphs)
  hipLaunchKernelGGL(seal_features_kernel<TO>, dim3(grid), dim3(kBlock), 0, as_stream(stream), nodes, node_off, tags,
                     tag_map, tag_map_len, T, emb, emb_dim, B, x, ldx, wstore, err_flag);
  return check_launch("seal_features");
}
}  // namespace

LPG_EXPORT int lpg_seal_features_f32(const int32_t* nodes, const int32_t* n_nodes, const int64_t* node_off,
                                     const int32_t* tags, const int32_t* tag_map, int64_t tag_map_len, int64_t T,
                                     const float* emb, int64_t emb_dim, int64_t B, float* x, int64_t ldx, int zero_pad,
                                     int32_t* err_flag, void* stream) {
  (void)n_nodes;
  return seal_features_entry<float>(nodes, node_off, tags, tag_map, tag_map_len, T, emb, emb_dim, B, x, ldx, zero_pad,
                                    err_flag, stream);
}

LPG_EXPORT int lpg_seal_features_bf16(const int32_t* nodes, const int32_t* n_nodes, const int64_t* node_off,
                                      const int32_t* tags, const int32_t* tag_map, int64_t tag_map_len, int64_t T,
                                      const float* emb, int64_t emb_dim, int64_t B, uint16_t* x, int64_t ldx,
                                      int zero_pad, int32_t* err_flag, void* stream) {
  (void)n_nodes;
  return seal_features_entry<uint16_t>(nodes, node_off, tags, tag_map, tag_map_len, T, emb, emb_dim, B, x, ldx, zero_pad,
                                       err_flag, stream);
}

LPG_EXPORT int lpg_seal_h1_tables_f32(const int32_t* nodes, const int64_t* node_off, const int32_t* tags,
                                      const int32_t* tag_map, int64_t tag_map_len, int64_t T, const float* w1_tag,
                                      const float* p_table, int64_t n_graph, int64_t B, float* h1, int64_t ldh,
                                      int32_t* err_flag, void* stream) {
  LPG_REQUIRE(B >= 0 && T >= 1 && n_graph >= 1 && ldh >= 32 && ldh % 4 == 0, "seal_h1_tables: bad sizes");
  if (B == 0) return LPG_OK;
  LPG_REQUIRE(nodes && node_off && tags && w1_tag && p_table && h1, "seal_h1_tables: null pointer");
  auto al16 = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  LPG_REQUIRE(al16(w1_tag) && al16(p_table) && al16(h1), "seal_h1_tables: tables and output must be 16-byte aligned");
  const int64_t g4 = ceil_div(B, kH1Block / 64);
  const int grid = static_cast<int>(g4 < (int64_t(1) << 24) ? g4 : (int64_t(1) << 24));   // a wave per subgraph
  hipLaunchKernelGGL(seal_h1_tables_kernel<float>, dim3(grid), dim3(kH1Block), 0, as_stream(stream), nodes, node_off, tags, tag_map,
                     tag_map_len, T, w1_tag, p_table, n_graph, B, h1, ldh, err_flag);
  return check_launch("seal_h1_tables");
}

LPG_EXPORT int lpg_seal_h1_tables_bf16(const int32_t* nodes, const int64_t* node_off, const int32_t* tags,
                                       const int32_t* tag_map, int64_t tag_map_len, int64_t T, const float* w1_tag,
                                       const float* p_table, int64_t n_graph, int64_t B, uint16_t* h1, int64_t ldh,
                                       int32_t* err_flag, void* stream) {
  LPG_REQUIRE(B >= 0 && T >= 1 && n_graph >= 1 && ldh >= 32 && ldh % 4 == 0, "seal_h1_tables_bf16: bad sizes (ldh >= 32, a multiple of 4)");
  if (B == 0) return LPG_OK;
  LPG_REQUIRE(nodes && node_off && tags && w1_tag && p_table && h1, "seal_h1_tables_bf16: null pointer");
  auto al = [](const void* p, int a) { return reinterpret_cast<uintptr_t>(p) % a == 0; };
  LPG_REQUIRE(al(w1_tag, 16) && al(p_table, 16) && al(h1, 8), "seal_h1_tables_bf16: tables 16-byte, output 8-byte aligned");
  const int64_t g4 = ceil_div(B, kH1Block / 64);
  const int grid = static_cast<int>(g4 < (int64_t(1) << 24) ? g4 : (int64_t(1) << 24));
  hipLaunchKernelGGL(seal_h1_tables_kernel<uint16_t>, dim3(grid), dim3(kH1Block), 0, as_stream(stream), nodes, node_off, tags,
                     tag_map, tag_map_len, T, w1_tag, p_table, n_graph, B, h1, ldh, err_flag);
  return check_launch("seal_h1_tables_bf16");
}

LPG_EXPORT int lpg_seal_sortpool_f32(const float* z, int64_t ldz, const float* key, const int64_t* node_off, int64_t B,
                                     int64_t k, int64_t C, float* out, void* stream) {
  LPG_REQUIRE(B >= 0 && k >= 1 && C >= 1 && ldz >= C, "seal_sortpool: bad sizes");
  if (B == 0) return LPG_OK;
  LPG_REQUIRE(z && key && node_off && out, "seal_sortpool: null pointer");
  const int grid = static_cast<int>(B < (int64_t(1) << 24) ? B : (int64_t(1) << 24));   // a workgroup per subgraph (see lpg_seal_subgraphs)
  LPG_REQUIRE(k * C < (int64_t(1) << 30), "seal_sortpool: k * C too large");
  const int vec = C % 4 == 0 && ldz % 4 == 0 && reinterpret_cast<uintptr_t>(z) % 16 == 0 &&
                  reinterpret_cast<uintptr_t>(out) % 16 == 0;
  if (vec && B > 64) {
    const int64_t g4 = ceil_div(B, kBlock / 64);
    hipLaunchKernelGGL(seal_sortpool_wave_kernel, dim3(static_cast<unsigned>(g4 < (int64_t(1) << 24) ? g4 : (int64_t(1) << 24))),
                       dim3(kBlock), 0, as_stream(stream), z, ldz, key, node_off, B, static_cast<int>(k), static_cast<int>(C / 4), out);
    return check_launch("seal_sortpool");
  }
  hipLaunchKernelGGL(seal_sortpool_kernel, dim3(grid), dim3(kSealBlock), 0, as_stream(stream), z, ldz, key, node_off, B,
                     k, C, out, vec);
  return check_launch("seal_sortpool");
}

LPG_EXPORT int lpg_softmax2_pos_score_f32(const float* pre_y, int64_t B, float* score, void* stream) {
  LPG_REQUIRE(B >= 0, "softmax2: bad size");
  if (B == 0) return LPG_OK;
  LPG_REQUIRE(pre_y && score, "softmax2: null pointer");
  hipLaunchKernelGGL(softmax2_kernel, dim3(grid_for(B)), dim3(kBlock), 0, as_stream(stream), pre_y, B, score);
  return check_launch("softmax2");
}

LPG_EXPORT int64_t lpg_seal_batch_index_bytes(int64_t B) {
  if (B < 0) return LPG_E_INVALID;
  if (B <= kIdxOneBlockMax) return 0;
  size_t bytes = 0;
  (void)rocprim::inclusive_scan(nullptr, bytes, SizeIt(nullptr, I32ToI64{}),
                                static_cast<int64_t*>(nullptr), static_cast<size_t>(B), rocprim::plus<int64_t>(),
                                hipStream_t(0), false);
  return lpg::align_up(static_cast<int64_t>(bytes), 256);
}

LPG_EXPORT int lpg_seal_batch_index(const int32_t* n_nodes, int64_t B, int64_t* node_off, const uint8_t* tag_present,
                                    int64_t n_tags, int32_t* tag_map, const int32_t* err_flag, int64_t* head,
                                    int32_t* first, int64_t window_rows, void* workspace, int64_t workspace_bytes,
                                    void* stream) {
  LPG_REQUIRE(B >= 0 && n_tags >= 0 && n_tags < (int64_t(1) << 30), "seal_batch_index: bad sizes");
  LPG_REQUIRE(!first || (window_rows >= 1 && B < (int64_t(1) << 31)), "seal_batch_index: bad window_rows");
  LPG_REQUIRE(node_off && head && (B == 0 || n_nodes) && (!tag_present || tag_map), "seal_batch_index: null pointer");
  hipStream_t st = as_stream(stream);
  const int one_block = B <= kIdxOneBlockMax;
  if (!one_block) {
    size_t bytes = static_cast<size_t>(lpg_seal_batch_index_bytes(B));
    if (!workspace || workspace_bytes < static_cast<int64_t>(bytes)) {
      lpg::set_error("seal_batch_index: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)bytes);
      return LPG_E_WORKSPACE;
    }
    LPG_HIP(hipMemsetAsync(node_off, 0, 8, st));
    LPG_HIP(rocprim::inclusive_scan(workspace, bytes, SizeIt(n_nodes, I32ToI64{}), node_off + 1,
                                    static_cast<size_t>(B), rocprim::plus<int64_t>(), st, false));
  }
  hipLaunchKernelGGL(seal_batch_index_kernel, dim3(tag_present ? 2 : 1), dim3(kIdxBlock), 0, st, n_nodes, B, node_off,
                     one_block, tag_present, static_cast<int>(n_tags), tag_map, err_flag, head, first, window_rows);
  return lpg::check_launch("seal_batch_index");
}

LPG_EXPORT int lpg_seal_logits_f32(const float* h, int64_t ldh, int64_t B, int64_t H, const float* w, const float* bias,
                                   float* pre_y, float* score, void* stream) {
  LPG_REQUIRE(B >= 0 && H >= 4 && H % 4 == 0 && H < (int64_t(1) << 30) && ldh >= H && ldh % 4 == 0,
              "seal_logits: needs H % 4 == 0 and ldh % 4 == 0");
  if (B == 0) return LPG_OK;
  LPG_REQUIRE(h && w && pre_y && score, "seal_logits: null pointer");
  LPG_REQUIRE(reinterpret_cast<uintptr_t>(h) % 16 == 0 && reinterpret_cast<uintptr_t>(w) % 16 == 0,
              "seal_logits: h and w must be 16-byte aligned");
  hipLaunchKernelGGL(seal_logits_kernel, dim3(grid_for(B * 16)), dim3(kBlock), 0, as_stream(stream), h, ldh, B,
                     static_cast<int>(H), w, bias, pre_y, score);
  return lpg::check_launch("seal_logits");
}

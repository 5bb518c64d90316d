// Small-graph GCN blocks: the launch-bound regime of the reference's own dataset (fb-pages-food: N = 620 nodes,
// P = 4252 tuples, 4 K / 74 K edges) where every kernel of the generic path lasts 2-5 us and the forward is
// the number of launches.  A GCNConv + GraphNorm + Dropout(eval) + ReLU block (TwoWL/model/model.py:36-41,
// :73, :77) becomes TWO launches instead of five or six:
//
//   sg_linear_t_kernel   hT[m][n] = (X W^T)^T, X given as rows, as the pair product x[pos[:,0]] * x[pos[:,1]]
//                        (model.py:75, no [P,C] intermediate) or as the sum of the two orientation outputs of the
//                        previous 2-WL layer (model.py:77); two weight matrices in one pass (conv2s / conv2s_r).
//                        Output TRANSPOSED: a feature column is contiguous.
//   sg_gcn_kernel        a 1024-thread workgroup owns (one feature column) x (one block of rows): the whole column
//                        goes to LDS pre-scaled by D^-1/2, the block's rows are aggregated out of LDS, + bias,
//                        stored row-major, and the block's column sums (y, y^2, fp64) are written as partials.
//
// GraphNorm + ReLU are DEFERRED to the consumer: its statistics are per column, the consumers (the next linear,
// the readout) hold one column per lane, so a lane folds the few per-block partials (fixed order, fp64), derives
// shift / scale once, and normalises values as it loads them.  No kernel ever waits for column statistics.
//
// Aggregation inside a workgroup: the block's edges (CSR order) are cut into 1024 strips, one per thread; a thread
// walks its strip left to right and keeps the running sum of the current row.  Rows that begin and end inside a
// strip are written straight to LDS; the first / last piece of a strip that may continue into a neighbour go
// to per-thread slots, and after one barrier the thread that holds the FIRST piece of such a row adds the
// following pieces in strip order.  No atomics, every sum has a fixed order => bitwise reproducible.  The edge
// stream is pre-packed once per graph as (col | row << 16) words laid out [row block][strip step][thread] so
// that the 16-byte loads of a wave are contiguous (lpg_small_pack_build); GCN weights are never read: w_e =
// dinv[row] * dinv[col] (PyG gcn_norm with unit edge weights) is applied as a column pre-scale and a row scale.
#include <stdlib.h>

#include <string.h>

#include "lpg_common.hpp"

namespace {
using namespace lpg;

constexpr int kSgBlock = 1024;        // threads of the column workgroup = strips of the edge stream
constexpr int kSgMaxRows = 8192;      // rows a column workgroup can hold (2 x 32 KB of LDS)
constexpr int kSgMaxRowBlocks = 8;
constexpr uint32_t kSgPad = 0xFFFFFFFFu;

__host__ __device__ inline int64_t sg_strip(int64_t E) {
  const int64_t s = ((E > 0 ? E : 1) + kSgBlock - 1) / kSgBlock;      // >= 1 step even for an empty block
  return (s + 3) & ~int64_t(3);
}
__host__ __device__ inline int sg_rows_per_block(int n, int rb) { return (n + rb - 1) / rb; }

// block b: words [b * cap_words, ...); word (q * 1024 + t) * 4 + j  <->  edge e0 + t * S + 4 q + j
__global__ __launch_bounds__(kBlock) void sg_pack_kernel(const int64_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ col,
                                                         const int32_t* __restrict__ rowid, int n, int rb,
                                                         int64_t cap_words, uint32_t* __restrict__ pack) {
  const int b = blockIdx.y;
  const int rpb = sg_rows_per_block(n, rb);
  const int r0 = b * rpb < n ? b * rpb : n, r1 = r0 + rpb < n ? r0 + rpb : n;
  const int64_t e0 = rowptr[r0], e1 = rowptr[r1];
  const int64_t S = sg_strip(e1 - e0);
  const int64_t words = S * kSgBlock;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  uint32_t* __restrict__ out = pack + static_cast<int64_t>(b) * cap_words;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < words; i += stride) {
    const int64_t q = i / (kSgBlock * 4), rem = i % (kSgBlock * 4);
    const int64_t t = rem >> 2, j = rem & 3;
    const int64_t e = e0 + t * S + 4 * q + j;
    uint32_t w = kSgPad;
    if (e < e1) w = (static_cast<uint32_t>(col[e]) & 0xFFFFu) | (static_cast<uint32_t>(rowid[e]) << 16);
    out[i] = w;
  }
}

// ------------------------------------------------------------------------------------------------ deferred GraphNorm
struct SgNorm {
  const double* part;    // [rb][2][C] per-row-block sums of y and y^2, or null: the tensor is used as it is
  const float* gw;       // GraphNorm weight / bias / mean_scale, [C] each or null
  const float* gb;
  const float* gms;
  int rb, relu, n, C;    // n = rows the statistics cover
  float eps;
};

struct SgCoef {
  float sh, sc, bi;
  int relu;
};

// torch_geometric.nn.GraphNorm (batch=None) [3P, TwoWL/model/model.py:38]: o = x - mean * mean_scale;
// y = weight * o / sqrt(mean(o^2) + eps) + bias; one-pass variance in fp64 (var = E[x^2] - mean^2 ms (2 - ms)).
__device__ __forceinline__ SgCoef sg_coef(const SgNorm& nm, int col) {
  SgCoef c{0.f, 1.f, 0.f, 0};
  if (!nm.part) return c;
  double p0[kSgMaxRowBlocks], p1[kSgMaxRowBlocks];
#pragma unroll
  for (int b = 0; b < kSgMaxRowBlocks; ++b) {              // every load first, clamped
    const int bb = b < nm.rb ? b : nm.rb - 1;
    p0[b] = nm.part[(static_cast<int64_t>(bb) * 2 + 0) * nm.C + col];
    p1[b] = nm.part[(static_cast<int64_t>(bb) * 2 + 1) * nm.C + col];
  }
  const float gw = nm.gw ? nm.gw[col] : 1.f, gb = nm.gb ? nm.gb[col] : 0.f, gms = nm.gms ? nm.gms[col] : 1.f;
  double s = 0.0, s2 = 0.0;
#pragma unroll
  for (int b = 0; b < kSgMaxRowBlocks; ++b)                // fixed order
    if (b < nm.rb) { s += p0[b]; s2 += p1[b]; }
  const double mean = s / static_cast<double>(nm.n), ex2 = s2 / static_cast<double>(nm.n);
  const double ms = static_cast<double>(gms);
  double var = ex2 - mean * mean * ms * (2.0 - ms);
  if (var < 0.0) var = 0.0;
  const double inv = 1.0 / sqrt(var + static_cast<double>(nm.eps));
  c.sh = static_cast<float>(mean * ms);
  c.sc = static_cast<float>(static_cast<double>(gw) * inv);
  c.bi = gb;
  c.relu = nm.relu;
  return c;
}
__device__ __forceinline__ float sg_apply(const SgCoef& c, float y) {
  const float o = (y - c.sh) * c.sc + c.bi;      // identity coefficients (0, 1, 0): exact
  return (c.relu && o < 0.f) ? 0.f : o;
}

// ------------------------------------------------------------------------------------------------ linear
// Outputs across lanes: a workgroup (4 waves x 4 rows) stages a tile of up to 64 weight rows TRANSPOSED in LDS
// ([k][64 + 1]: conflict-free for "lanes = outputs") and its 16 input rows (normalised / multiplied / added
// while loading) as plain rows; then lane s owns output s and walks k: the input value is one LDS broadcast per
// 4 k's and 4 rows, the weight one conflict-free read, and NO cross-lane reduction exists (the first version
// spread k over the lanes and spent its time in 272 dependent shuffles per 4 rows).  With fewer than 64 output
// slots per workgroup (8..32: long k, few outputs) the 64 / slots lane groups split k and meet in 1-3 shuffles;
// the grid is (row tiles) x (output tiles).  A lane ends with its output for 4 consecutive rows = one 16-byte
// piece of the transposed result.
struct SgLinArgs {
  const float* xa;       // mode 0: rows [n, k]; mode 1: table [n_src, k]; mode 2: first addend [n, k]
  const float* xb;       // mode 2: second addend
  const int64_t* pos;    // mode 1: [n, 2]
  const float* w;        // [m, k]
  const float* w2;       // [m, k] or null
  float* hT;             // [m, ldt]
  float* h2T;            // [m, ldt] or null
  int32_t* err_flag;
  int n, n_src, k, m, mode, slots;
  int64_t ldt;
  SgNorm na, nb;         // deferred GraphNorm of xa / xb
};

constexpr int kSgLinK = 256;            // largest contraction length
constexpr int kSgLinRows = 16;          // rows per workgroup (4 waves x 4)
constexpr int kSgWtLd = 65;             // padded leading dimension of the transposed weight tile

template <int KPL>                       // ceil(k / 64): input values per lane while staging a row
__global__ __launch_bounds__(kBlock) void sg_linear_t_kernel(const SgLinArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sg_lds[];     // sized by k: small k => several workgroups per CU
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r0 = (blockIdx.x * (kBlock / kWave) + wv) * 4;
  const int kpad = (a.k + 3) & ~3;
  float* __restrict__ xs = sg_lds;                          // [16][kpad]
  float* __restrict__ wt = sg_lds + kSgLinRows * kpad;      // [kpad][65]
  const int m_total = a.w2 ? 2 * a.m : a.m;
  const int SL = a.slots;                                  // output slots of a workgroup (8..64); blockIdx.y = output tile
  const int kparts = 64 / SL;
  // ---- the wave's 4 input rows -> LDS (lane = k index, KPL = ceil(k / 64) steps)
  {
    int64_t ra[4], rb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + i < a.n ? r0 + i : a.n - 1;       // clamped rows: finite duplicates, stored into padding
      ra[i] = rb[i] = r;
    }
    if (a.mode == 1) {
      int64_t p0[4], p1[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {                        // all eight index loads first
        p0[i] = a.pos[2 * ra[i]];
        p1[i] = a.pos[2 * ra[i] + 1];
      }
      bool bad = false;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool b = p0[i] < 0 || p0[i] >= a.n_src || p1[i] < 0 || p1[i] >= a.n_src;
        bad |= b;
        ra[i] = b ? 0 : p0[i];
        rb[i] = b ? 0 : p1[i];
      }
      if (bad && lane == 0) flag_error(a.err_flag, LPG_F_INDEX_RANGE);
    }
    // every load of the 4 rows first (clamped), then the arithmetic
    float u[KPL][4], v[KPL][4];
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      const int kk = lane + 64 * j;
      const int kc = kk < a.k ? kk : a.k - 1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u[j][i] = a.xa[ra[i] * a.k + kc];
        v[j][i] = a.mode == 0 ? 0.f : (a.mode == 1 ? a.xa[rb[i] * a.k + kc] : a.xb[rb[i] * a.k + kc]);
      }
    }
    SgCoef ca[KPL], cb[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      const int kk = lane + 64 * j;
      const int kc = kk < a.k ? kk : a.k - 1;
      ca[j] = sg_coef(a.na, kc);
      cb[j] = a.mode == 2 ? sg_coef(a.nb, kc) : ca[j];
    }
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      const int kk = lane + 64 * j;
      if (kk < kpad) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float fu = sg_apply(ca[j], u[j][i]), fv = sg_apply(cb[j], v[j][i]);
          const float t = a.mode == 0 ? fu : (a.mode == 1 ? fu * fv : fu + fv);
          xs[(wv * 4 + i) * kpad + kk] = kk < a.k ? t : 0.f;         // zero tail up to a multiple of 4
        }
      }
    }
  }
  const int s = lane & (SL - 1), kp = lane / SL;
  const int kh = kparts == 1 ? kpad : (((kpad + kparts - 1) / kparts + 3) & ~3);     // k's per lane group, a multiple of 4
  const int k_lo = kp * kh, k_hi = (k_lo + kh < kpad) ? k_lo + kh : kpad;
  {
    const int g0 = blockIdx.y * SL;
    // ---- stage the weight tile transposed: element e = slot * kpad + kk (kk fastest: coalesced reads)
    const int total = SL * kpad;
    const int d_sl = kBlock / kpad, d_kk = kBlock - d_sl * kpad;       // one step of 256 elements in (slot, k)
    int sl = tid / kpad, kk = tid - sl * kpad;
    for (int e0 = 0; e0 < total; e0 += kBlock * 8) {
      float wv8[8];
      int sl8[8], kk8[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        sl8[q] = sl; kk8[q] = kk;
        int g = g0 + sl;
        g = g < m_total ? g : m_total - 1;
        const float* __restrict__ wm = g >= a.m ? a.w2 : a.w;
        const int o = g >= a.m ? g - a.m : g;
        wv8[q] = wm[o * a.k + (kk < a.k ? kk : a.k - 1)];
        sl += d_sl; kk += d_kk;
        if (kk >= kpad) { kk -= kpad; ++sl; }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (sl8[q] < SL) wt[kk8[q] * kSgWtLd + sl8[q]] = kk8[q] < a.k ? wv8[q] : 0.f;
    }
    __syncthreads();
    // ---- lane = output slot: walk k
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int kk = k_lo; kk < k_hi; kk += 4) {
      float4 xv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) xv[i] = *reinterpret_cast<const float4*>(&xs[(wv * 4 + i) * kpad + kk]);
      const float w0 = wt[(kk + 0) * kSgWtLd + s], w1 = wt[(kk + 1) * kSgWtLd + s], w2 = wt[(kk + 2) * kSgWtLd + s],
                  w3 = wt[(kk + 3) * kSgWtLd + s];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = fmaf(xv[i].x, w0, acc[i]);
        acc[i] = fmaf(xv[i].y, w1, acc[i]);
        acc[i] = fmaf(xv[i].z, w2, acc[i]);
        acc[i] = fmaf(xv[i].w, w3, acc[i]);
      }
    }
    for (int o = 32; o >= SL; o >>= 1) {                   // the lane groups that split k meet
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] += __shfl_xor(acc[i], o, 64);
    }
    const int g = g0 + s;
    if (kp == 0 && g < m_total && r0 < a.n) {
      float* __restrict__ out = g >= a.m ? a.h2T : a.hT;
      const int o = g >= a.m ? g - a.m : g;
      *reinterpret_cast<float4*>(out + static_cast<int64_t>(o) * a.ldt + r0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ aggregation
struct SgSide {
  const float* hT;          // [c, ldt] transposed features (sg_linear_t_kernel)
  const uint32_t* pack;     // packed edge stream of this side's graph (rb row blocks)
  const int64_t* rowptr;    // [n + 1]
  const float* dinv;        // [n] D^-1/2
  const float* bias;        // [c] or null (GCNConv.bias)
  float* y;                 // [n, ldo] row-major: A_hat h + bias, NOT yet normalised
  double* part;             // [rb][2][c] column sums of y and y^2 per row block
};

// conv1 of LocalWLNet without node features (TwoWL/model/model.py:71-73): the layer's input is GraphNorm(Embedding(deg)),
// i.e. a table of n_table rows gathered by idx.  With `table` set, the column workgroup makes its feature column itself
// -- idx histogram -> GraphNorm statistics (the fp64 one-pass formula of gn_small_kernel, weighted by the histogram) ->
// one value per DISTINCT index: T'[v] = sum_k GN(table[v][k]) W[c][k] -> hs[r] = T'[idx[r]] -- instead of reading it from
// sg_linear_t_kernel's output: lpg_embedding_graphnorm_f32 + lpg_small_linear_t_f32 + lpg_small_gcn_f32 in one launch.
struct SgEmbed {
  const float* table;       // [n_table, k] (null: not used)
  const int64_t* idx;       // [n]
  int n_table, k;
  const float* gn_w;        // GraphNorm weight / bias / mean_scale [k] (null: 1 / 0 / 1)
  const float* gn_b;
  const float* gn_ms;
  float eps;
  const float* lin_w;       // [c, k] GCNConv.lin.weight
  int32_t* err_flag;
};

// RT = rows per thread of the LONGER of (all rows: the column fill) and nothing else -- ceil(n / 1024); absent rows are
// clamped duplicate loads, and at 620 rows seven of eight were
template <int RT, bool EMB>
__global__ __launch_bounds__(kSgBlock) void sg_gcn_kernel(const SgSide s0, const SgSide s1, int n, int rb, int64_t ldt,
                                                          int64_t ldo, int64_t cap_words, const SgEmbed em) {
  __shared__ float hs[kSgMaxRows];
  __shared__ float acc[kSgMaxRows];
  __shared__ int first_row[kSgBlock], last_row[kSgBlock];
  __shared__ float l_sum[kSgBlock];
  __shared__ double red[2][kSgBlock / 64];
  const SgSide s = blockIdx.z ? s1 : s0;
  const int c = blockIdx.x, blk = blockIdx.y, C = gridDim.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int rpb = sg_rows_per_block(n, rb);
  const int r0 = blk * rpb < n ? blk * rpb : n, r1 = r0 + rpb < n ? r0 + rpb : n;
  // every parameter / index load up front
  const int64_t e0 = s.rowptr[r0], e1 = s.rowptr[r1];
  const float bias = s.bias ? s.bias[c] : 0.f;
  const uint4* __restrict__ pk = reinterpret_cast<const uint4*>(s.pack + static_cast<int64_t>(blk) * cap_words);
  // the first strip steps are requested before the strip length is known (a dependent round trip less); steps
  // beyond it hold stale words of the capacity-sized buffer and are masked by `live` below
  constexpr int kBatch = 6;
  const int q_cap = static_cast<int>(cap_words / (kSgBlock * 4));
  uint4 u[kBatch], un[kBatch];
#pragma unroll
  for (int b = 0; b < kBatch; ++b) u[b] = pk[static_cast<int64_t>(b < q_cap ? b : q_cap - 1) * kSgBlock + t];
  float dv[RT];
  int iv[RT];                                            // EMB: the rows' table indices
  if constexpr (EMB) {
    // LDS that the aggregation uses later serves the prologue: hist / the list of present indices / T' / the column's
    // GraphNorm shift, scale, bias and weight row
    int* const hist = first_row;
    int* const plist = last_row;
    float* const tcol = l_sum;
    float* const sh = acc;
    float* const sc = acc + kSgBlock;
    float* const bi = acc + 2 * kSgBlock;
    float* const wr = acc + 3 * kSgBlock;
    double* const psum = reinterpret_cast<double*>(hs);  // [2][kSgBlock] partial column sums (16 KB of the 32)
    __shared__ int wcnt[kSgBlock / 64];
    const int k = em.k;
    hist[t] = 0;
    // parameter loads up front
    float p_w = 1.f, p_b = 0.f, p_ms = 1.f, wv = 0.f;
    if (t < k) {
      if (em.gn_w) p_w = em.gn_w[t];
      if (em.gn_b) p_b = em.gn_b[t];
      if (em.gn_ms) p_ms = em.gn_ms[t];
      wv = em.lin_w[static_cast<int64_t>(c) * k + t];
    }
    int64_t v64[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int r = t + i * kSgBlock;
      v64[i] = em.idx[r < n ? r : n - 1];
    }
    __syncthreads();
    bool bad = false;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const bool b = v64[i] < 0 || v64[i] >= em.n_table;
      bad |= b;
      iv[i] = b ? 0 : static_cast<int>(v64[i]);
      if (t + i * kSgBlock < n) atomicAdd(&hist[iv[i]], 1);
    }
    if (bad) flag_error(em.err_flag, LPG_F_INDEX_RANGE);
    __syncthreads();
    // the distinct indices in ascending order (the summation order below must not depend on the run)
    const bool pres = t < em.n_table && hist[t] > 0;
    const uint64_t bal = __ballot(pres);
    if (lane == 0) wcnt[wave] = __popcll(bal);
    __syncthreads();
    int base = 0, npres = 0;
    for (int w2 = 0; w2 < kSgBlock / 64; ++w2) {
      if (w2 < wave) base += wcnt[w2];
      npres += wcnt[w2];
    }
    if (pres) plist[base + __popcll(bal & ((uint64_t(1) << lane) - 1))] = t;
    __syncthreads();
    // column sums over the n gathered rows = histogram-weighted sums over the distinct table rows; the threads beyond
    // k take every G-th present row of column t mod kp so that the loads of a column are spread over G threads
    int kp = 64;
    while (kp < k) kp <<= 1;
    const int G = kSgBlock / kp, col = t & (kp - 1), grp = t / kp;
    {
      double ps = 0.0, ps2 = 0.0;
      if (col < k) {
        constexpr int U = 8;
        for (int j0 = grp; j0 < npres; j0 += G * U) {
          float xq[U];
          int hq[U];
#pragma unroll
          for (int q = 0; q < U; ++q) {
            const int j = j0 + q * G;
            const int vv = plist[j < npres ? j : npres - 1];
            hq[q] = j < npres ? hist[vv] : 0;
            xq[q] = em.table[static_cast<int64_t>(vv) * k + col];
          }
#pragma unroll
          for (int q = 0; q < U; ++q) {
            const double xd = static_cast<double>(xq[q]), hd = static_cast<double>(hq[q]);
            ps += hd * xd;
            ps2 += hd * xd * xd;
          }
        }
      }
      psum[t] = ps;
      psum[kSgBlock + t] = ps2;
    }
    __syncthreads();
    if (t < k) {
      double ts = 0.0, ts2 = 0.0;
      for (int gq = 0; gq < G; ++gq) { ts += psum[gq * kp + t]; ts2 += psum[kSgBlock + gq * kp + t]; }   // fixed order
      const double mean = ts / static_cast<double>(n), ex2 = ts2 / static_cast<double>(n);
      const double ms = static_cast<double>(p_ms);
      double var = ex2 - mean * mean * ms * (2.0 - ms);
      if (var < 0.0) var = 0.0;
      const double inv = 1.0 / sqrt(var + static_cast<double>(em.eps));
      sh[t] = static_cast<float>(mean * ms);
      sc[t] = static_cast<float>(static_cast<double>(p_w) * inv);
      bi[t] = p_b;
      wr[t] = wv;
    }
    __syncthreads();
    // T'[v] = sum_k ((table[v][k] - shift_k) * scale_k + bias_k) * W[c][k] for the distinct v: a wave per v
    for (int j = wave; j < npres; j += kSgBlock / 64) {
      const int vv = plist[j];
      float pacc = 0.f;
      for (int cc = lane; cc < k; cc += 64) {
        const float xg = (em.table[static_cast<int64_t>(vv) * k + cc] - sh[cc]) * sc[cc] + bi[cc];
        pacc = fmaf(xg, wr[cc], pacc);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) pacc += __shfl_xor(pacc, o);
      if (lane == 0) tcol[vv] = pacc;
    }
    __syncthreads();
  }
  {
    float hv[RT], dd[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int r = t + i * kSgBlock;
      const int rc = r < n ? r : n - 1;
      if constexpr (EMB) hv[i] = l_sum[iv[i]];
      else hv[i] = s.hT[static_cast<int64_t>(c) * ldt + rc];
      dd[i] = s.dinv[rc];
      const int rl = r0 + r;
      dv[i] = s.dinv[rl < n ? rl : n - 1];
    }
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int r = t + i * kSgBlock;
      if (r < n) hs[r] = hv[i] * dd[i];
      if (r < r1 - r0) acc[r] = 0.f;
    }
  }
  const int Q = static_cast<int>(sg_strip(e1 - e0) >> 2);   // 16-byte steps of a strip
  __syncthreads();

  // ---- strip walk: the loads of step group b + 1 are in flight while group b is walked
  int first = -1, cur = -1;
  float sum = 0.f, lpiece = 0.f;
  for (int q0 = 0; q0 < Q; q0 += kBatch) {
#pragma unroll
    for (int b = 0; b < kBatch; ++b) {
      const int q = q0 + kBatch + b < Q ? q0 + kBatch + b : Q - 1;
      un[b] = pk[static_cast<int64_t>(q) * kSgBlock + t];
    }
#pragma unroll
    for (int b = 0; b < kBatch; ++b) {
      const bool live = q0 + b < Q;
      const uint32_t w4[4] = {u[b].x, u[b].y, u[b].z, u[b].w};
      float v4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v4[j] = hs[(live && w4[j] != kSgPad) ? (w4[j] & 0xFFFFu) : 0];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (live && w4[j] != kSgPad) {
          const int r = static_cast<int>(w4[j] >> 16) - r0;
          if (r != cur) {
            if (cur >= 0) {
              if (cur == first) lpiece = sum;          // the strip's first row ends here
              else acc[cur] = sum;                     // a row that begins and ends inside this strip
            } else {
              first = r;
            }
            cur = r;
            sum = 0.f;
          }
          sum += v4[j];
        }
      }
    }
#pragma unroll
    for (int b = 0; b < kBatch; ++b) u[b] = un[b];
  }
  const bool single = cur == first;                    // one row (or nothing) in the whole strip
  if (cur >= 0 && single) lpiece = sum;
  first_row[t] = first;
  last_row[t] = cur;
  l_sum[t] = lpiece;
  __syncthreads();

  // ---- rows cut by strip borders: the holder of a row's FIRST piece adds the following pieces in strip order
  if (first >= 0) {
    const int prev_last = t > 0 ? last_row[t - 1] : -1;
    if (prev_last != first) {
      float total = lpiece;
      if (single)
        for (int tt = t + 1; tt < kSgBlock && first_row[tt] == first; ++tt) {
          total += l_sum[tt];
          if (last_row[tt] != first) break;
        }
      acc[first] = total;
    }
    if (!single) {
      float total = sum;
      for (int tt = t + 1; tt < kSgBlock && first_row[tt] == cur; ++tt) {
        total += l_sum[tt];
        if (last_row[tt] != cur) break;
      }
      acc[cur] = total;
    }
  }
  __syncthreads();

  // ---- dinv[row] * sum + bias -> y; the block's column sums for the deferred GraphNorm
  double ds = 0.0, ds2 = 0.0;
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    const int r = t + i * kSgBlock;
    if (r < r1 - r0) {
      const float y = fmaf(acc[r], dv[i], bias);
      s.y[static_cast<int64_t>(r0 + r) * ldo + c] = y;
      const double q = static_cast<double>(y);
      ds += q;
      ds2 += q * q;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ds += __shfl_xor(ds, o);
    ds2 += __shfl_xor(ds2, o);
  }
  if (lane == 0) { red[0][wave] = ds; red[1][wave] = ds2; }
  __syncthreads();
  if (t < 2) {
    double ts = 0.0;
    for (int w = 0; w < kSgBlock / 64; ++w) ts += red[t][w];                           // fixed order
    s.part[(static_cast<int64_t>(blk) * 2 + t) * C + c] = ts;
  }
}

// ------------------------------------------------------------------------------------------------ readout
// TwoWL/model/model.py:78-83 over the two orientation outputs of the last 2-WL layer, both still un-normalised:
// x = relu(GN_a(ya)) + relu(GN_r(yr)); out[b] = sum_c x[idx[2b], c] * x[idx[2b+1], c] * w[c] + b0.
// One wave per scored edge, lanes = channels (each lane derives its channel's shift / scale once).
__global__ __launch_bounds__(kBlock) void sg_readout_kernel(const float* __restrict__ ya, const float* __restrict__ yr,
                                                            int P, int C, const SgNorm na, const SgNorm nr,
                                                            const int64_t* __restrict__ idx, int B,
                                                            const float* __restrict__ w, const float* __restrict__ b0,
                                                            float* __restrict__ out, int32_t* err_flag) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * kBlock) >> 6;
  const float bb = b0 ? b0[0] : 0.f;
  for (int cb = 0; cb < C; cb += 64) {                      // C <= 64 in every reference configuration: one pass
    const int c = cb + lane < C ? cb + lane : C - 1;
    const SgCoef ca = sg_coef(na, c), cr = yr ? sg_coef(nr, c) : ca;
    const float wc = cb + lane < C ? w[c] : 0.f;
    for (int b = wave; b < B; b += nwaves) {
      const int64_t i0 = idx[2 * b], i1 = idx[2 * b + 1];
      if (i0 < 0 || i0 >= P || i1 < 0 || i1 >= P) {
        if (lane == 0 && cb == 0) flag_error(err_flag, LPG_F_INDEX_RANGE);
        continue;
      }
      const float a0 = ya[i0 * C + c], a1 = ya[i1 * C + c];
      float q0 = 0.f, q1 = 0.f;
      if (yr) { q0 = yr[i0 * C + c]; q1 = yr[i1 * C + c]; }
      float x0 = sg_apply(ca, a0), x1 = sg_apply(ca, a1);
      if (yr) { x0 += sg_apply(cr, q0); x1 += sg_apply(cr, q1); }
      float sacc = x0 * x1 * wc;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);
      if (lane == 0) out[b] = cb == 0 ? sacc + bb : out[b] + sacc;
    }
  }
}

inline bool al16(const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }
inline bool rb_ok(int64_t rb) { return rb >= 1 && rb <= kSgMaxRowBlocks; }

bool make_norm(SgNorm* nm, const double* part, const float* gw, const float* gb, const float* gms, int64_t rb, int relu,
               int64_t n_stat, int64_t C, float eps) {
  nm->part = part; nm->gw = gw; nm->gb = gb; nm->gms = gms;
  nm->rb = static_cast<int>(rb); nm->relu = relu; nm->n = static_cast<int>(n_stat); nm->C = static_cast<int>(C);
  nm->eps = eps;
  return !part || (rb_ok(rb) && n_stat >= 1);
}
}  // namespace

LPG_EXPORT int64_t lpg_small_max_rows(void) { return kSgMaxRows; }
LPG_EXPORT int64_t lpg_small_max_row_blocks(void) { return kSgMaxRowBlocks; }

LPG_EXPORT int64_t lpg_small_pack_bytes(int64_t e_cap, int64_t row_blocks) {
  if (e_cap < 0 || !rb_ok(row_blocks)) return LPG_E_INVALID;
  return row_blocks * sg_strip(e_cap) * kSgBlock * 4;
}

LPG_EXPORT int lpg_small_pack_build(const void* plan, const int64_t* rowptr, const int32_t* col, int64_t n,
                                    int64_t e_cap, int64_t row_blocks, void* pack, int64_t pack_bytes, void* stream) {
  LPG_REQUIRE(n >= 1 && n <= kSgMaxRows && e_cap >= 0 && rb_ok(row_blocks),
              "small_pack_build: needs 1 <= n <= %d rows, 1 <= row_blocks <= %d", kSgMaxRows, kSgMaxRowBlocks);
  LPG_REQUIRE(plan && rowptr && pack && (e_cap == 0 || col), "small_pack_build: null pointer");
  LPG_REQUIRE(al16(pack), "small_pack_build: pack must be 16-byte aligned");
  if (pack_bytes < lpg_small_pack_bytes(e_cap, row_blocks)) {
    set_error("small_pack_build: buffer too small");
    return LPG_E_WORKSPACE;
  }
  const int32_t* rowid = reinterpret_cast<const int32_t*>(static_cast<const char*>(plan) + 256);
  const int64_t cap_words = sg_strip(e_cap) * kSgBlock;
  hipLaunchKernelGGL(sg_pack_kernel, dim3(grid_for(cap_words, kBlock, 256), static_cast<unsigned>(row_blocks)), dim3(kBlock),
                     0, as_stream(stream), rowptr, col, rowid, static_cast<int>(n), static_cast<int>(row_blocks), cap_words,
                     static_cast<uint32_t*>(pack));
  return check_launch("small_pack_build");
}

LPG_EXPORT int lpg_small_linear_t_f32(const float* xa, const float* xb, const int64_t* pos, int mode, int64_t n_src,
                                      int64_t n, int64_t k, const float* w, const float* w2, int64_t m, float* hT,
                                      float* h2T, int64_t ldt, const double* part_a, const float* gn_weight_a,
                                      const float* gn_bias_a, const float* gn_mean_scale_a, int64_t row_blocks_a,
                                      int relu_a, const double* part_b, const float* gn_weight_b, const float* gn_bias_b,
                                      const float* gn_mean_scale_b, int64_t row_blocks_b, int relu_b, float eps,
                                      int32_t* err_flag, void* stream) {
  LPG_REQUIRE(n >= 0 && k >= 1 && k <= kSgLinK && m >= 1 && mode >= 0 && mode <= 2, "small_linear_t: bad sizes (k <= %d)", kSgLinK);
  if (n == 0) return LPG_OK;
  LPG_REQUIRE(n <= (int64_t(1) << 30) && xa && w && hT && (!w2 || h2T), "small_linear_t: null pointer");
  LPG_REQUIRE((mode != 1 || (pos && n_src >= 1)) && (mode != 2 || xb), "small_linear_t: mode operand missing");
  LPG_REQUIRE(ldt >= n && ldt % 4 == 0 && ldt >= ((n + 3) & ~int64_t(3)) && al16(hT) && (!h2T || al16(h2T)),
              "small_linear_t: ldt must be a multiple of 4 covering n rounded up to 4, outputs 16-byte aligned");
  SgLinArgs a;
  a.xa = xa; a.xb = xb; a.pos = pos; a.w = w; a.w2 = w2; a.hT = hT; a.h2T = h2T; a.err_flag = err_flag;
  a.n = static_cast<int>(n); a.n_src = static_cast<int>(mode == 1 ? n_src : n); a.k = static_cast<int>(k);
  a.m = static_cast<int>(m); a.mode = mode; a.ldt = ldt;
  const int64_t n_stat = mode == 1 ? n_src : n;
  LPG_REQUIRE(make_norm(&a.na, part_a, gn_weight_a, gn_bias_a, gn_mean_scale_a, row_blocks_a, relu_a, n_stat, k, eps) &&
              make_norm(&a.nb, part_b, gn_weight_b, gn_bias_b, gn_mean_scale_b, row_blocks_b, relu_b, n_stat, k, eps),
              "small_linear_t: bad deferred-norm descriptor");
  const int64_t kpad = (k + 3) & ~int64_t(3);
  // output slots per workgroup: the transposed weight tile (slots x kpad) is staged in ONE batch of 8 loads per
  // thread when it has <= 2048 elements; fewer slots = more workgroups, each lane group walking a shorter piece of k
  const int64_t m_total = w2 ? 2 * m : m;
  int slots = 64;
  while (slots > 8 && (slots * kpad > 2048 || slots / 2 >= m_total)) slots >>= 1;
  a.slots = slots;
  const size_t lds = static_cast<size_t>(kSgLinRows * kpad + kpad * kSgWtLd) * sizeof(float);   // <= 83 KB
  const dim3 grid(static_cast<unsigned>(ceil_div(n, kSgLinRows)), static_cast<unsigned>(ceil_div(m_total, slots)));
  hipStream_t st = as_stream(stream);
#define LPG_SG_LIN(KPLV)                                                                                             \
  do {                                                                                                               \
    static const hipError_t optin = hipFuncSetAttribute(reinterpret_cast<const void*>(sg_linear_t_kernel<KPLV>),     \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);      \
    LPG_REQUIRE(optin == hipSuccess, "small_linear_t: cannot reserve dynamic LDS");                                  \
    hipLaunchKernelGGL(sg_linear_t_kernel<KPLV>, grid, dim3(kBlock), lds, st, a);                                    \
  } while (0)
  switch ((k + 63) / 64) {
    case 1: LPG_SG_LIN(1); break;
    case 2: LPG_SG_LIN(2); break;
    case 3: LPG_SG_LIN(3); break;
    default: LPG_SG_LIN(4); break;
  }
#undef LPG_SG_LIN
  return check_launch("small_linear_t");
}

LPG_EXPORT int lpg_small_gcn_f32(int64_t n, int64_t c, int64_t ldt, int64_t ldo, int64_t row_blocks, int64_t e_cap,
                                 const float* hT, const void* pack, const int64_t* rowptr, const float* dinv,
                                 const float* bias, float* y, double* part, const float* hT2, const void* pack2,
                                 const int64_t* rowptr2, const float* dinv2, const float* bias2, float* y2,
                                 double* part2, void* stream) {
  LPG_REQUIRE(n >= 1 && n <= kSgMaxRows && c >= 1 && c < 65536 && rb_ok(row_blocks) && e_cap >= 0,
              "small_gcn: needs 1 <= n <= %d, 1 <= row_blocks <= %d", kSgMaxRows, kSgMaxRowBlocks);
  LPG_REQUIRE(hT && pack && rowptr && dinv && y && part && ldt >= n && ldo >= c, "small_gcn: bad argument");
  const bool two = hT2 != nullptr;
  LPG_REQUIRE(!two || (pack2 && rowptr2 && dinv2 && y2 && part2), "small_gcn: second side incomplete");
  LPG_REQUIRE(al16(pack) && (!two || al16(pack2)), "small_gcn: pack must be 16-byte aligned");
  SgSide s0{hT, static_cast<const uint32_t*>(pack), rowptr, dinv, bias, y, part};
  SgSide s1 = s0;
  if (two) s1 = SgSide{hT2, static_cast<const uint32_t*>(pack2), rowptr2, dinv2, bias2, y2, part2};
  const dim3 grid(static_cast<unsigned>(c), static_cast<unsigned>(row_blocks), two ? 2 : 1);
  const int rows = static_cast<int>((n + kSgBlock - 1) / kSgBlock);
  const SgEmbed em{};
#define LPG_SG_GCN(RTV)                                                                                               \
  hipLaunchKernelGGL((sg_gcn_kernel<RTV, false>), grid, dim3(kSgBlock), 0, as_stream(stream), s0, s1, static_cast<int>(n), \
                     static_cast<int>(row_blocks), ldt, ldo, sg_strip(e_cap) * kSgBlock, em)
  if (rows <= 1) LPG_SG_GCN(1);
  else if (rows <= 2) LPG_SG_GCN(2);
  else if (rows <= 4) LPG_SG_GCN(4);
  else LPG_SG_GCN(8);
#undef LPG_SG_GCN
  return check_launch("small_gcn");
}

LPG_EXPORT int lpg_small_gcn_embed_f32(int64_t n, int64_t c, int64_t ldo, int64_t row_blocks, int64_t e_cap,
                                       const float* table, int64_t n_table, int64_t k, const int64_t* idx,
                                       const float* gn_weight, const float* gn_bias, const float* gn_mean_scale, float eps,
                                       const float* lin_w, const void* pack, const int64_t* rowptr, const float* dinv,
                                       const float* bias, float* y, double* part, int32_t* err_flag, void* stream) {
  LPG_REQUIRE(n >= 1 && n <= kSgMaxRows && c >= 1 && c < 65536 && rb_ok(row_blocks) && e_cap >= 0,
              "small_gcn_embed: needs 1 <= n <= %d, 1 <= row_blocks <= %d", kSgMaxRows, kSgMaxRowBlocks);
  LPG_REQUIRE(n_table >= 1 && n_table <= kSgBlock && k >= 1 && k <= kSgBlock,
              "small_gcn_embed: needs a table of at most %d rows and %d columns", kSgBlock, kSgBlock);
  LPG_REQUIRE(table && idx && lin_w && pack && rowptr && dinv && y && part && ldo >= c, "small_gcn_embed: bad argument");
  LPG_REQUIRE(al16(pack), "small_gcn_embed: pack must be 16-byte aligned");
  const SgSide s0{nullptr, static_cast<const uint32_t*>(pack), rowptr, dinv, bias, y, part};
  const SgEmbed em{table, idx, static_cast<int>(n_table), static_cast<int>(k), gn_weight, gn_bias, gn_mean_scale, eps, lin_w,
                   err_flag};
  const dim3 grid(static_cast<unsigned>(c), static_cast<unsigned>(row_blocks), 1);
  const int rows = static_cast<int>((n + kSgBlock - 1) / kSgBlock);
#define LPG_SG_GCN_E(RTV)                                                                                             \
  hipLaunchKernelGGL((sg_gcn_kernel<RTV, true>), grid, dim3(kSgBlock), 0, as_stream(stream), s0, s0, static_cast<int>(n), \
                     static_cast<int>(row_blocks), n, ldo, sg_strip(e_cap) * kSgBlock, em)
  if (rows <= 1) LPG_SG_GCN_E(1);
  else if (rows <= 2) LPG_SG_GCN_E(2);
  else if (rows <= 4) LPG_SG_GCN_E(4);
  else LPG_SG_GCN_E(8);
#undef LPG_SG_GCN_E
  return check_launch("small_gcn_embed");
}

LPG_EXPORT int lpg_small_readout_f32(const float* ya, const float* yr, int64_t P, int64_t C, const double* part_a,
                                     const float* gn_weight_a, const float* gn_bias_a, const float* gn_mean_scale_a,
                                     int64_t row_blocks_a, int relu_a, const double* part_r, const float* gn_weight_r,
                                     const float* gn_bias_r, const float* gn_mean_scale_r, int64_t row_blocks_r,
                                     int relu_r, float eps, const int64_t* idx, int64_t B, const float* w,
                                     const float* b0, float* out, int32_t* err_flag, void* stream) {
  LPG_REQUIRE(P >= 0 && C >= 1 && B >= 0 && P < (int64_t(1) << 31) && B < (int64_t(1) << 31), "small_readout: bad sizes");
  if (B == 0) return LPG_OK;
  LPG_REQUIRE(ya && idx && w && out && P >= 1, "small_readout: null pointer");
  SgNorm na, nr;
  LPG_REQUIRE(make_norm(&na, part_a, gn_weight_a, gn_bias_a, gn_mean_scale_a, row_blocks_a, relu_a, P, C, eps) &&
              make_norm(&nr, part_r, gn_weight_r, gn_bias_r, gn_mean_scale_r, row_blocks_r, relu_r, P, C, eps),
              "small_readout: bad deferred-norm descriptor");
  hipLaunchKernelGGL(sg_readout_kernel, dim3(grid_for(B * 64)), dim3(kBlock), 0, as_stream(stream), ya, yr,
                     static_cast<int>(P), static_cast<int>(C), na, nr, idx, static_cast<int>(B), w, b0, out, err_flag);
  return check_launch("small_readout");
}

// ---- the whole small-graph scoring forward as ONE call ------------------------------------------------------------
// A forward on fb-pages-food is six launches of ~5-12 us; enqueued one ctypes call at a time the host spends 150 us
// on them (argument marshalling, tensor bookkeeping, one call per kernel).  The host mirror therefore prepares an
// ARGUMENT BLOCK once per (model, graph, candidate set): the launches of the forward, in order, as int64 words
//     { op, a_0, ..., a_{k-1} }  per launch,  op in {1: lpg_embedding_graphnorm_f32 (12 words),
//     2: lpg_small_linear_t_f32 (27), 3: lpg_small_gcn_f32 (20), 4: lpg_small_readout_f32 (23),
//     5: lpg_small_gcn_embed_f32 (21)},
// a_i = the entry point's i-th argument (pointers and integers as they are, `float eps` as the bit pattern of a
// double), the trailing `stream` argument left out.  lpg_twowl_forward_small walks the block and enqueues every
// launch on `stream` through the very same entry points (same checks, same kernels, same results as calling them
// one by one).
namespace {
template <typename T> inline T* wp(int64_t v) { return reinterpret_cast<T*>(static_cast<intptr_t>(v)); }
inline float wf(int64_t v) {
  double d;
  memcpy(&d, &v, sizeof(d));
  return static_cast<float>(d);
}
}  // namespace

LPG_EXPORT int lpg_twowl_forward_small(const int64_t* block, int64_t n_words, void* stream) {
  LPG_REQUIRE(block && n_words >= 0, "twowl_forward_small: null block");
  int64_t i = 0;
  while (i < n_words) {
    const int64_t op = block[i];
    const int64_t* a = block + i + 1;
    const int64_t need = op == 1 ? 12 : op == 2 ? 27 : op == 3 ? 20 : op == 4 ? 23 : op == 5 ? 21 : op == 6 ? 19 : op == 7 ? 31 : op == 8 ? 15 : -1;
    LPG_REQUIRE(need > 0 && i + 1 + need <= n_words, "twowl_forward_small: malformed block at word %lld (op %lld)",
                (long long)i, (long long)op);
    int rc = LPG_OK;
    if (op == 1) {
      rc = lpg_embedding_graphnorm_f32(wp<const float>(a[0]), a[1], a[2], wp<const int64_t>(a[3]), a[4], wp<const float>(a[5]),
                                       wp<const float>(a[6]), wp<const float>(a[7]), wf(a[8]), static_cast<int>(a[9]),
                                       wp<float>(a[10]), wp<int32_t>(a[11]), stream);
    } else if (op == 2) {
      rc = lpg_small_linear_t_f32(wp<const float>(a[0]), wp<const float>(a[1]), wp<const int64_t>(a[2]), static_cast<int>(a[3]),
                                  a[4], a[5], a[6], wp<const float>(a[7]), wp<const float>(a[8]), a[9], wp<float>(a[10]),
                                  wp<float>(a[11]), a[12], wp<const double>(a[13]), wp<const float>(a[14]),
                                  wp<const float>(a[15]), wp<const float>(a[16]), a[17], static_cast<int>(a[18]),
                                  wp<const double>(a[19]), wp<const float>(a[20]), wp<const float>(a[21]),
                                  wp<const float>(a[22]), a[23], static_cast<int>(a[24]), wf(a[25]), wp<int32_t>(a[26]), stream);
    } else if (op == 3) {
      rc = lpg_small_gcn_f32(a[0], a[1], a[2], a[3], a[4], a[5], wp<const float>(a[6]), wp<const void>(a[7]),
                             wp<const int64_t>(a[8]), wp<const float>(a[9]), wp<const float>(a[10]), wp<float>(a[11]),
                             wp<double>(a[12]), wp<const float>(a[13]), wp<const void>(a[14]), wp<const int64_t>(a[15]),
                             wp<const float>(a[16]), wp<const float>(a[17]), wp<float>(a[18]), wp<double>(a[19]), stream);
    } else if (op == 5) {
      rc = lpg_small_gcn_embed_f32(a[0], a[1], a[2], a[3], a[4], wp<const float>(a[5]), a[6], a[7], wp<const int64_t>(a[8]),
                                   wp<const float>(a[9]), wp<const float>(a[10]), wp<const float>(a[11]), wf(a[12]),
                                   wp<const float>(a[13]), wp<const void>(a[14]), wp<const int64_t>(a[15]),
                                   wp<const float>(a[16]), wp<const float>(a[17]), wp<float>(a[18]), wp<double>(a[19]),
                                   wp<int32_t>(a[20]), stream);
    } else if (op == 6) {
      rc = lpg_tuple_layer_f32(static_cast<int>(a[0]), a[1], a[2], a[3], wp<const float>(a[4]), a[5], wp<const float>(a[6]), a[7],
                               wp<const float>(a[8]), wp<const int32_t>(a[9]), wp<const int32_t>(a[10]), wp<const float>(a[11]),
                               wp<const float>(a[12]), wp<const float>(a[13]), wp<const float>(a[14]), wp<float>(a[15]),
                               wp<float>(a[16]), wp<double>(a[17]), a[18], stream);
    } else if (op == 8) {
      rc = lpg_tuple_ax_small_f32(a[0], a[1], wp<const float>(a[2]), a[3], wp<const int64_t>(a[4]), wp<const int32_t>(a[5]),
                                  wp<const double>(a[6]), wp<const float>(a[7]), wp<const float>(a[8]), wp<const float>(a[9]),
                                  a[10], static_cast<int>(a[11]), wf(a[12]), wp<float>(a[13]), wp<float>(a[14]), stream);
    } else if (op == 7) {
      rc = lpg_tuple_readout_f32(a[0], a[1], a[2], a[3], a[4], wp<const int64_t>(a[5]), wp<const float>(a[6]),
                                 wp<const float>(a[7]), wp<const int32_t>(a[8]), wp<const int32_t>(a[9]), wp<const double>(a[10]),
                                 a[11], a[12], wp<const float>(a[13]), wp<const float>(a[14]), wp<const float>(a[15]), wf(a[16]),
                                 static_cast<int>(a[17]), wp<const float>(a[18]), wp<const float>(a[19]), wp<const float>(a[20]),
                                 wf(a[21]), static_cast<int>(a[22]), wp<const float>(a[23]), wp<const float>(a[24]),
                                 wp<const int64_t>(a[25]), a[26], wp<const float>(a[27]), wp<const float>(a[28]), wp<float>(a[29]),
                                 wp<int32_t>(a[30]), stream);
    } else {
      rc = lpg_small_readout_f32(wp<const float>(a[0]), wp<const float>(a[1]), a[2], a[3], wp<const double>(a[4]),
                                 wp<const float>(a[5]), wp<const float>(a[6]), wp<const float>(a[7]), a[8],
                                 static_cast<int>(a[9]), wp<const double>(a[10]), wp<const float>(a[11]),
                                 wp<const float>(a[12]), wp<const float>(a[13]), a[14], static_cast<int>(a[15]), wf(a[16]),
                                 wp<const int64_t>(a[17]), a[18], wp<const float>(a[19]), wp<const float>(a[20]),
                                 wp<float>(a[21]), wp<int32_t>(a[22]), stream);
    }
    if (rc != LPG_OK) return rc;
    i += 1 + need;
  }
  return LPG_OK;
}

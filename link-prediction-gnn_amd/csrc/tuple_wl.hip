// The 2-WL aggregation of LocalWLNet WITHOUT the tuple graph (TwoWL/utils.py:36-50,71-78 -> TwoWL/model/model.py:75-83).
//
// get_ei2 links, for every node i, every graph tuple ENDING at i with every tuple STARTING at i: a complete bipartite
// block per node, E2 = sum_i indeg(i) * outdeg(i) edges.  With pos_edge in double()'s layout (columns 2k / 2k + 1 are
// the two directions of one edge; blocked ids come in such pairs, TwoWL/model/train.py:26) the two orientations that
// reverse() makes of it have, for a destination tuple b,
//     orientation a: in-neighbours = the MEMBER tuples (graph tuples that are not blocked) starting at node_a(b) = start(b)
//     orientation r: in-neighbours = the members ending at node_r(b) = start(b ^ 1)   (= end(b) for doubled candidates)
// plus, through add_remaining_self_loops, b itself exactly when b is not a member; and all members of one group have the
// same degree cnt(node).  So GCNConv over either orientation is a per-node sum over the members and a broadcast:
//     y[b] = T[node(b)] / cnt + bias                                        b member
//     y[b] = T[node(b)] / sqrt(cnt (cnt + 1)) + (h[b] W^T) / (cnt + 1) + bias   otherwise  (first term 0 when cnt = 0)
// with T[i] = (sum of h over the group of i) W^T.  Members' features are node-level themselves -- x[u] * x[v] in front of
// the first layer, F_a[start] + F_r[end] after one -- so the group sums are products with the member adjacency A
// (sum_{t: i -> w} x[i] x[w] = x[i] * (A x)[i]), and a whole layer costs O((N + Q) C^2 + E1 C) instead of O(E2 C)
// (Q = non-member tuples: the candidates, and blocked graph tuples).  GraphNorm's column statistics over all P rows are
// sum_i cnt(i) M[i] + sum_q y[q]: never a pass over P rows either.  fp64 restatement: oracle/aggregation.py
// (localwl_forward_collapsed), equal to the explicit path to 1e-10 on random graphs with loops, duplicates and blocking.
//
// Kernels (lanes 0-31 = columns of orientation a, 32-63 = orientation r; C <= 32, every reference configuration):
//   tw_layer_kernel   a wave per node row / non-member row: builds the row's input vector per orientation, multiplies it
//                     by W (rows of W in registers, the vector broadcast from a wave-private LDS slot), writes T / y and
//                     fp64 column-sum partials per workgroup (fixed order: reproducible);
//   tw_fold_kernel    many partial blocks -> 16;   tw_apply_kernel  GraphNorm + ReLU between two layers;
//   tw_readout_kernel model.py:78-83 with the last layer's GraphNorm applied on load;
//   tw_bcast_kernel   the broadcast half of the operator form (GCNConv(x, orientation) on arbitrary [P, C] input).
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "lpg_common.hpp"

namespace lpg {
namespace {

constexpr int kTwBlock = 1024;             // 16 waves that never synchronise before the final fold of their sums
constexpr int kTwWaves = kTwBlock / kWave;
constexpr int kTwMaxBlocks = 1024;
constexpr int kTwMaxPart = 64;             // partial blocks a consumer folds by itself

struct FlagToI32 {
  __host__ __device__ int32_t operator()(uint8_t v) const { return v ? 1 : 0; }
};

// ------------------------------------------------------------------------------------------------ plan
__global__ __launch_bounds__(kBlock) void tw_plan_flags_kernel(const int64_t* __restrict__ pos_edge, int64_t E1,
                                                               const int64_t* __restrict__ pred_edge, int64_t Ep,
                                                               int64_t n_node, const uint8_t* __restrict__ blocked,
                                                               const int64_t* __restrict__ rowptr, int32_t* __restrict__ cnt,
                                                               uint8_t* __restrict__ flags, int32_t* head, int32_t* err_flag) {
  const int64_t P = E1 + Ep, work = P > n_node ? P : n_node;
  for (int64_t b = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; b < work;
       b += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    if (b < n_node) cnt[b] = static_cast<int32_t>(rowptr[b + 1] - rowptr[b]);
    if (b >= P) continue;
    const int64_t s = b < E1 ? pos_edge[b] : pred_edge[b - E1];
    const int64_t e = b < E1 ? pos_edge[E1 + b] : pred_edge[Ep + b - E1];
    if (s < 0 || s >= n_node || e < 0 || e >= n_node) {
      flag_error(err_flag, LPG_F_INDEX_RANGE);
      atomicOr(head + 1, 2);
    }
    bool member = b < E1;
    if (b < E1) {
      const int64_t p = b ^ 1;
      if (pos_edge[p] != e || pos_edge[E1 + p] != s) atomicOr(head + 1, 1);          // not double()'s layout
      if (blocked) {
        if ((blocked[b] != 0) != (blocked[p] != 0)) atomicOr(head + 1, 1);           // one direction blocked only
        member = blocked[b] == 0;
      }
    }
    flags[b] = member ? 0 : 1;
  }
}

__global__ __launch_bounds__(kBlock) void tw_plan_fill_kernel(const int64_t* __restrict__ pos_edge, int64_t E1,
                                                              const int64_t* __restrict__ pred_edge, int64_t Ep,
                                                              int64_t n_node, const uint8_t* __restrict__ flags, int scanned,
                                                              int32_t* __restrict__ qpos, int32_t* __restrict__ qinfo,
                                                              int64_t q_cap, int32_t* __restrict__ tnode,
                                                              int64_t* __restrict__ seg_a, int64_t* __restrict__ seg_r,
                                                              int32_t* head) {
  const int64_t P = E1 + Ep;
  auto start_of = [&](int64_t t) { return t < E1 ? pos_edge[t] : pred_edge[t - E1]; };
  auto clampn = [&](int64_t v) { return v < 0 ? int64_t(0) : (v >= n_node ? n_node - 1 : v); };
  for (int64_t b = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; b < P;
       b += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t s = clampn(start_of(b));
    const int64_t e = clampn(b < E1 ? pos_edge[E1 + b] : pred_edge[Ep + b - E1]);
    const int64_t nr = clampn(start_of(b ^ 1));                     // P is even (checked by the entry point)
    const bool non = flags[b] != 0;
    const int32_t q = scanned ? qpos[b] : static_cast<int32_t>(b - E1);
    if (tnode) {
      tnode[2 * b] = static_cast<int32_t>(s);
      tnode[2 * b + 1] = static_cast<int32_t>(nr);
    }
    if (b < E1) {
      if (seg_a) seg_a[b] = non ? n_node : s;
      if (seg_r) seg_r[b] = non ? n_node : e;
    }
    if (non) {
      if (q < q_cap) {
        int32_t* o = qinfo + 4 * static_cast<int64_t>(q);
        o[0] = static_cast<int32_t>(s);
        o[1] = static_cast<int32_t>(e);
        o[2] = static_cast<int32_t>(nr);
        o[3] = static_cast<int32_t>(b);
      }
      qpos[b] = q;
    } else {
      qpos[b] = -1;
    }
    if (b == P - 1) head[0] = scanned ? q + (non ? 1 : 0) : static_cast<int32_t>(Ep);
  }
}

// ------------------------------------------------------------------------------------------------ GraphNorm on load
struct TwCoef {
  float sh, sc, bi;
  int relu;
};
__device__ __forceinline__ float tw_apply(const TwCoef& c, float y) {
  const float o = (y - c.sh) * c.sc + c.bi;
  return (c.relu && o < 0.f) ? 0.f : o;
}
// torch_geometric.nn.GraphNorm (batch=None) [3P, TwoWL/model/model.py:38]: o = x - mean * mean_scale;
// y = weight * o / sqrt(mean(o^2) + eps) + bias; one-pass variance in fp64 (the formula of small_graph.hip's sg_coef)
__device__ __forceinline__ TwCoef tw_coef_from_sums(double s, double s2, double n, float gw, float gb, float gms, float eps,
                                                    int relu) {
  const double mean = s / n, ex2 = s2 / n, ms = static_cast<double>(gms);
  double var = ex2 - mean * mean * ms * (2.0 - ms);
  if (var < 0.0) var = 0.0;
  const double inv = 1.0 / sqrt(var + static_cast<double>(eps));
  return TwCoef{static_cast<float>(mean * ms), static_cast<float>(static_cast<double>(gw) * inv), gb, relu};
}

struct TwNorm {          // statistics of one 2-WL layer: part [n_part][2][64] (column = side * 32 + c), P rows
  const double* part;
  const float *gw_a, *gb_a, *gms_a, *gw_r, *gb_r, *gms_r;
  double n_rows;
  float eps_a, eps_r;
  int n_part, relu_a, relu_r;
};
// the 128 column sums (sum y | sum y^2 for 64 columns) of a layer, folded by a whole 256-thread workgroup: thread
// (p, h) adds the partial blocks h, h + 2, ... of sum p (all loads in flight at once), the two halves meet in LDS in a
// fixed order.  Call from every thread of the workgroup; `sums` = 256 doubles of LDS.
__device__ __forceinline__ void tw_fold_block(const double* __restrict__ part, int n_part, int first, int step, double* sums) {
  const int p = threadIdx.x & 127, h = threadIdx.x >> 7;
  double acc = 0.0;
  int g = first + h * step;
  for (; g + 6 * step < n_part; g += 8 * step) {
    const double v0 = part[static_cast<int64_t>(g) * 128 + p], v1 = part[static_cast<int64_t>(g + 2 * step) * 128 + p];
    const double v2 = part[static_cast<int64_t>(g + 4 * step) * 128 + p], v3 = part[static_cast<int64_t>(g + 6 * step) * 128 + p];
    acc += v0; acc += v1; acc += v2; acc += v3;
  }
  for (; g < n_part; g += 2 * step) acc += part[static_cast<int64_t>(g) * 128 + p];
  sums[h * 128 + p] = acc;
  __syncthreads();
}
__device__ __forceinline__ TwCoef tw_coef(const TwNorm& nm, const double* sums, int side, int c) {
  const int col = side * 32 + c;
  const double s = sums[col] + sums[128 + col], s2 = sums[64 + col] + sums[128 + 64 + col];
  const float* gw = side ? nm.gw_r : nm.gw_a;
  const float* gb = side ? nm.gb_r : nm.gb_a;
  const float* gms = side ? nm.gms_r : nm.gms_a;
  return tw_coef_from_sums(s, s2, nm.n_rows, gw ? gw[c] : 1.f, gb ? gb[c] : 0.f, gms ? gms[c] : 1.f,
                           side ? nm.eps_r : nm.eps_a, side ? nm.relu_r : nm.relu_a);
}

struct TwNormIn {        // deferred GraphNorm of the first layer's input (small-graph conv1: y + [rb][2][C] column sums)
  const double* part;
  const float *gw, *gb, *gms;
  int rb, relu, n;
  float eps;
};
__device__ __forceinline__ TwCoef tw_coef_in(const TwNormIn& nm, int C, int c) {
  double s = 0.0, s2 = 0.0;
  for (int b = 0; b < nm.rb; ++b) {
    s += nm.part[(static_cast<int64_t>(b) * 2 + 0) * C + c];
    s2 += nm.part[(static_cast<int64_t>(b) * 2 + 1) * C + c];
  }
  return tw_coef_from_sums(s, s2, static_cast<double>(nm.n), nm.gw ? nm.gw[c] : 1.f, nm.gb ? nm.gb[c] : 0.f,
                           nm.gms ? nm.gms[c] : 1.f, nm.eps, nm.relu);
}

// value of a member row of node i: every member tuple of the group shares it.  One v_rcp_f32 (1 ulp) instead of the
// IEEE division sequence; every kernel that needs the value goes through here, so statistics and rows agree bit for bit.
__device__ __forceinline__ float tw_member(float t, int cn, float bias) {
  return fmaf(t, __builtin_amdgcn_rcpf(static_cast<float>(cn)), bias);
}

// ------------------------------------------------------------------------------------------------ A x on small graphs
// x1n = GraphNorm(y1) (+ ReLU; the small-graph conv1 leaves its output un-normalised with per-row-block column sums)
// and ax = A x1n, a 256-thread workgroup per node row: its eight 32-lane slots take the row's entries round robin,
// eight gathers in flight each, and meet in LDS in a fixed order.  (Summing the member row inside the layer kernel
// instead -- one launch less -- cost 33-58 us on fb-pages-food: a candidate row on a hub walks 109 entries alone.)
__global__ __launch_bounds__(kBlock) void tw_ax_small_kernel(int C, const float* __restrict__ y1, int64_t ldy,
                                                             const int64_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ col, const TwNormIn nin,
                                                             float* __restrict__ x1n, float* __restrict__ ax) {
  __shared__ float red[8][32];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int slot = wv * 2 + (lane >> 5), c = lane & 31;
  const int cc = c < C ? c : C - 1;
  TwCoef cin{0.f, 1.f, 0.f, 0};
  if (nin.part) cin = tw_coef_in(nin, C, cc);
  const int64_t i = blockIdx.x;
  const int64_t e0 = rowptr[i], e1 = rowptr[i + 1];
  float acc = 0.f;
  for (int64_t e = e0 + slot; e < e1; e += 64) {             // entries e, e + 8, ..., e + 56 of this slot
    int cj[8];
    float t8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) cj[u] = col[e + 8 * u < e1 ? e + 8 * u : e];
#pragma unroll
    for (int u = 0; u < 8; ++u) t8[u] = tw_apply(cin, y1[static_cast<int64_t>(cj[u]) * ldy + cc]);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += e + 8 * u < e1 ? t8[u] : 0.f;
  }
  red[slot][c] = acc;
  __syncthreads();
  if (threadIdx.x < 32 && c < C) {
    float sacc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) sacc += red[k][c];
    ax[i * C + c] = sacc;
    x1n[i * C + c] = tw_apply(cin, y1[i * ldy + c]);
  }
}

// ------------------------------------------------------------------------------------------------ one 2-WL layer
struct TwLayerArgs {
  const float* x;        // MODE 1: x1 [n, C] (the 1-WL output); MODE 2: FF [n, 2C] = (F_a | F_r)
  const float* ax;       // MODE 1: A x1 [n, C]; MODE 2: A FF [n, 2C]
  const float* hq_prev;  // MODE 2: the non-member rows of the previous layer [Q, C]
  const int32_t* cnt;    // [n] members starting (= ending) at a node
  const int32_t* qinfo;  // [Q][4] = (start, end, node_r, tuple id)
  const float *w_a, *w_r, *b_a, *b_r;
  float* t_node;         // [n, 2C] = (T_a | T_r), may be null
  float* y_q;            // [Q, 2C]
  double* part;          // [gridDim.x][2 kinds][2][64]: column sums of the node rows (kind 0) and of the non-member rows (kind 1)
  int64_t ldx, ldax;
  int n_node, Q, C, tile;
};

constexpr int kTwRows = 4;                 // rows a wave keeps in flight
__device__ __forceinline__ float rl_f(float v, int lane) {      // v_readlane_b32 of a float (the builtin is typed int)
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// A wave owns tiles of `tile` (<= 64) consecutive node rows or non-member rows: lane j reads row j's descriptors
// (member count / (start, end, node_r)) with one coalesced load and derives the row's two scale factors, then the rows
// go through in groups of kTwRows with every feature-row load of the group issued before anything is computed -- a row
// is a chain of dependent round trips (descriptor -> feature rows -> product), and a wave that walks them one at a
// time spends its life waiting (first version: 230 us for 436 K rows).
template <int MODE>
__global__ __launch_bounds__(kTwBlock) void tw_layer_kernel(const TwLayerArgs a) {
  __shared__ float sv[kTwWaves][kTwRows][64];
  __shared__ double sred[kTwWaves][2][64];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int side = lane >> 5, c = lane & 31, C = a.C;
  const bool live = c < C;
  const int cc = live ? c : C - 1;
  // W_a, W_r -> LDS with coalesced loads (rows padded to 33 words: lane c reads row c conflict-free), then each lane's
  // row into registers.  Read straight from global memory, lane c walking row c, every load instruction touches 64
  // different lines and the 16 waves of the workgroup queue behind each other: 13 of the 18 us the kernel took on
  // fb-pages-food.
  __shared__ float sw[2][32][33];
  for (int i = threadIdx.x; i < 2 * C * C; i += kTwBlock) {
    const int sd = i >= C * C, rem = i - sd * C * C;
    sw[sd][rem / C][rem % C] = (sd ? a.w_r : a.w_a)[rem];
  }
  __syncthreads();
  float w[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) w[k] = (live && k < C) ? sw[side][cc][k < C ? k : 0] : 0.f;
  const float bias = (side ? a.b_r : a.b_a)[cc];
  // 32-bit element offsets (the entry point checks that every operand has < 2^31 elements): one VGPR per address
  const uint32_t ldx = static_cast<uint32_t>(a.ldx), ldax = static_cast<uint32_t>(a.ldax), C2 = 2u * C;
  // column sums of y and y^2, node rows and non-member rows apart: a candidate list sharded over GPUs sums the second kind
  // across ranks (lpgnn_amd.dist.ShardedTwoWLScorer) while every rank holds all node rows
  double s1 = 0.0, s2 = 0.0, q1 = 0.0, q2 = 0.0;
  const int ts = a.tile;
  const int nt_node = (a.n_node + ts - 1) / ts, nt_q = (a.Q + ts - 1) / ts;
  const int gw = blockIdx.x * kTwWaves + wv, nw = gridDim.x * kTwWaves;
  const uint32_t own = side * C + cc, oth = (1 - side) * C + cc;
  const uint32_t xcol = MODE == 1 ? cc : own, acol = MODE == 1 ? cc : oth;
  for (int tile = gw; tile < nt_node + nt_q; tile += nw) {
    const bool isq = tile >= nt_node;
    const int r0 = (isq ? tile - nt_node : tile) * ts;
    const int left = (isq ? a.Q : a.n_node) - r0;
    const int nr = left < ts ? left : ts;
    const int jl = lane < nr ? lane : nr - 1;
    // lane j <-> row r0 + j: node ids, member counts and the two factors of a non-member row per orientation
    int d_s = r0 + jl, d_e = 0, d_nr = 0, d_cs, d_cr = 0;
    float f_aa = 0.f, f_ba = 0.f, f_ar = 0.f, f_br = 0.f;
    if (isq) {
      const int4 inf = *reinterpret_cast<const int4*>(a.qinfo + 4u * static_cast<uint32_t>(r0 + jl));
      d_s = inf.x; d_e = inf.y; d_nr = inf.z;
      d_cr = a.cnt[d_nr];
    }
    d_cs = a.cnt[d_s];
    if (isq) {
      const float fs = static_cast<float>(d_cs), fr = static_cast<float>(d_cr);
      f_aa = d_cs > 0 ? __builtin_amdgcn_rsqf(fs * (fs + 1.0f)) : 0.f;
      f_ba = __builtin_amdgcn_rcpf(fs + 1.0f);
      f_ar = d_cr > 0 ? __builtin_amdgcn_rsqf(fr * (fr + 1.0f)) : 0.f;
      f_br = __builtin_amdgcn_rcpf(fr + 1.0f);
    }
    for (int j0 = 0; j0 < nr; j0 += kTwRows) {
      float v[kTwRows];
      int cn[kTwRows];
      if (!isq) {
        float xo[kTwRows], xa[kTwRows];
#pragma unroll
        for (int u = 0; u < kTwRows; ++u) {
          const int j = j0 + u < nr ? j0 + u : nr - 1;
          const uint32_t i = r0 + j;
          cn[u] = __builtin_amdgcn_readlane(d_cs, j);
          xo[u] = a.x[i * ldx + xcol];
          xa[u] = a.ax[i * ldax + acol];
        }
#pragma unroll
        for (int u = 0; u < kTwRows; ++u) v[u] = MODE == 1 ? xo[u] * xa[u] : fmaf(static_cast<float>(cn[u]), xo[u], xa[u]);
      } else {
        float h[kTwRows], xn[kTwRows], an[kTwRows], fa[kTwRows], fb[kTwRows];
#pragma unroll
        for (int u = 0; u < kTwRows; ++u) {
          const int j = j0 + u < nr ? j0 + u : nr - 1;
          // every readlane unconditional, the orientation picked by a select: written as side ? readlane : readlane
          // hipcc branches per half-wave around each of them and drains vmcnt in between
          const int s = __builtin_amdgcn_readlane(d_s, j), nrn = __builtin_amdgcn_readlane(d_nr, j);
          const int cs_ = __builtin_amdgcn_readlane(d_cs, j), cr_ = __builtin_amdgcn_readlane(d_cr, j);
          const float aa_ = rl_f(f_aa, j), ar_ = rl_f(f_ar, j), ba_ = rl_f(f_ba, j), br_ = rl_f(f_br, j);
          const uint32_t node = side ? nrn : s;
          cn[u] = side ? cr_ : cs_;
          fa[u] = side ? ar_ : aa_;
          fb[u] = side ? br_ : ba_;
          if (MODE == 1) {
            const int e = __builtin_amdgcn_readlane(d_e, j);
            const float xs = a.x[static_cast<uint32_t>(s) * ldx + cc], xe = a.x[static_cast<uint32_t>(e) * ldx + cc];
            h[u] = xs * xe;
            xn[u] = a.x[node * ldx + cc];
          } else {
            h[u] = a.hq_prev[static_cast<uint32_t>(r0 + j) * C + cc];
            xn[u] = a.x[node * ldx + own];
          }
          an[u] = a.ax[node * ldax + acol];
        }
#pragma unroll
        for (int u = 0; u < kTwRows; ++u) {
          const float sx = MODE == 1 ? xn[u] * an[u] : fmaf(static_cast<float>(cn[u]), xn[u], an[u]);
          v[u] = fmaf(fa[u], sx, fb[u] * h[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < kTwRows; ++u) sv[wv][u][lane] = live ? v[u] : 0.f;
      __builtin_amdgcn_wave_barrier();
      float acc[kTwRows];
#pragma unroll
      for (int u = 0; u < kTwRows; ++u) acc[u] = 0.f;
#pragma unroll
      for (int k4 = 0; k4 < 8; ++k4) {
        if (4 * k4 < C) {
#pragma unroll
          for (int u = 0; u < kTwRows; ++u) {
            const float4 t = *reinterpret_cast<const float4*>(&sv[wv][u][side * 32 + 4 * k4]);
            acc[u] = fmaf(t.x, w[4 * k4 + 0], acc[u]);
            acc[u] = fmaf(t.y, w[4 * k4 + 1], acc[u]);
            acc[u] = fmaf(t.z, w[4 * k4 + 2], acc[u]);
            acc[u] = fmaf(t.w, w[4 * k4 + 3], acc[u]);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int u = 0; u < kTwRows; ++u) {
        if (j0 + u < nr && live) {
          const uint32_t r = r0 + j0 + u;
          if (!isq) {
            if (a.t_node) a.t_node[r * C2 + own] = acc[u];
            if (cn[u] > 0) {
              const double m = static_cast<double>(tw_member(acc[u], cn[u], bias)), cd = static_cast<double>(cn[u]);
              s1 += cd * m;
              s2 += cd * m * m;
            }
          } else {
            const float y = acc[u] + bias;
            a.y_q[r * C2 + own] = y;
            q1 += static_cast<double>(y);
            q2 += static_cast<double>(y) * static_cast<double>(y);
          }
        }
      }
    }
  }
#pragma unroll
  for (int kind = 0; kind < 2; ++kind) {
    if (kind) __syncthreads();                               // the first kind's sums have been read
    sred[wv][0][lane] = kind ? q1 : s1;
    sred[wv][1][lane] = kind ? q2 : s2;
    __syncthreads();
    if (threadIdx.x < 128) {
      const int which = threadIdx.x >> 6, ln = threadIdx.x & 63;
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < kTwWaves; ++k) s += sred[k][which][ln];
      a.part[((static_cast<int64_t>(blockIdx.x) * 2 + kind) * 2 + which) * 64 + ln] = s;
    }
  }
}

// partial blocks blockIdx, blockIdx + gridDim, ... -> one
__global__ __launch_bounds__(kBlock) void tw_fold_kernel(const double* __restrict__ part, int n_in, double* __restrict__ out) {
  __shared__ double sums[256];
  tw_fold_block(part, n_in, blockIdx.x, gridDim.x, sums);
  if (threadIdx.x < 128) out[static_cast<int64_t>(blockIdx.x) * 128 + threadIdx.x] = sums[threadIdx.x] + sums[128 + threadIdx.x];
}

// ------------------------------------------------------------------------------------------------ between two layers
__global__ __launch_bounds__(kBlock) void tw_apply_kernel(int n_node, int Q, int C, const float* __restrict__ t_node,
                                                          const float* __restrict__ y_q, const int32_t* __restrict__ cnt,
                                                          const TwNorm nm, const float* __restrict__ b_a,
                                                          const float* __restrict__ b_r, float* __restrict__ ff,
                                                          float* __restrict__ hq) {
  __shared__ double sums[256];
  tw_fold_block(nm.part, nm.n_part, 0, 1, sums);
  const int lane = threadIdx.x & 63;
  const int side = lane >> 5, c = lane & 31;
  const bool live = c < C;
  const int cc = live ? c : C - 1;
  const TwCoef co = tw_coef(nm, sums, side, cc);
  const float bias = (side ? b_r : b_a)[cc];
  const int gw = (blockIdx.x * kBlock + threadIdx.x) >> 6, nw = (gridDim.x * kBlock) >> 6;
  for (int r = gw; r < n_node + Q; r += nw) {
    if (r < n_node) {
      const int cn = cnt[r];
      const float t = t_node[static_cast<int64_t>(r) * (2 * C) + side * C + cc];
      const float f = cn > 0 ? tw_apply(co, tw_member(t, cn, bias)) : 0.f;
      if (live) ff[static_cast<int64_t>(r) * (2 * C) + side * C + c] = f;
    } else {
      const int q = r - n_node;
      float f = tw_apply(co, y_q[static_cast<int64_t>(q) * (2 * C) + side * C + cc]);
      f += __shfl_xor(f, 32);
      if (live && side == 0) hq[static_cast<int64_t>(q) * C + c] = f;
    }
  }
}

// ------------------------------------------------------------------------------------------------ readout
// TwoWL/model/model.py:78-83: rows idx of x = relu(GN_a(y_a)) + relu(GN_r(y_r)); out[j] = sum_c x[idx[2j]] x[idx[2j+1]] w + b0.
// A wave per scored pair: lanes 0-31 hold the row of idx[2j], 32-63 the row of idx[2j + 1].
__global__ __launch_bounds__(kBlock) void tw_readout_kernel(int n_node, int Q, int C, int64_t E1, int64_t P,
                                                            const int64_t* __restrict__ pos_edge,
                                                            const float* __restrict__ t_node, const float* __restrict__ y_q,
                                                            const int32_t* __restrict__ cnt, const int32_t* __restrict__ qpos,
                                                            const TwNorm nm, const float* __restrict__ b_a,
                                                            const float* __restrict__ b_r, const int64_t* __restrict__ idx,
                                                            int B, const float* __restrict__ w, const float* __restrict__ b0,
                                                            float* __restrict__ out, int32_t* err_flag) {
  __shared__ double sums[256];
  tw_fold_block(nm.part, nm.n_part, 0, 1, sums);
  const int lane = threadIdx.x & 63;
  const int t = lane >> 5, c = lane & 31;
  const bool live = c < C;
  const int cc = live ? c : C - 1;
  const TwCoef ca = tw_coef(nm, sums, 0, cc), cr = tw_coef(nm, sums, 1, cc);
  const float ba = b_a[cc], br = b_r[cc], wc = live ? w[cc] : 0.f, bb = b0 ? b0[0] : 0.f;
  const int gw = (blockIdx.x * kBlock + threadIdx.x) >> 6, nw = (gridDim.x * kBlock) >> 6;
  // a wave takes 64 consecutive pairs: lane j reads pair j's two tuple ids (16 bytes, coalesced) and their non-member
  // positions once, then the pairs go through four at a time with their row loads issued together (walking the pairs
  // one by one was three dependent round trips each: 29 us for 100 K pairs)
  int ts = 1;                                                   // pairs per tile: ~8192 tiles (every wave slot of the chip), between one group and a wave's 64 lanes
  while (ts < 64 && B > 8192 * ts) ts *= 2;
  const int n_tiles = (B + ts - 1) / ts;
  for (int tile = gw; tile < n_tiles; tile += nw) {
    const int j0 = tile * ts, np_ = B - j0 < ts ? B - j0 : ts;
    const int jl = lane < np_ ? lane : np_ - 1;
    const longlong2 pr = *reinterpret_cast<const longlong2*>(idx + 2 * static_cast<int64_t>(j0 + jl));
    const bool bad = pr.x < 0 || pr.x >= P || pr.y < 0 || pr.y >= P;
    if (bad && lane < np_) flag_error(err_flag, LPG_F_INDEX_RANGE);
    const int d_b0 = bad ? 0 : static_cast<int>(pr.x), d_b1 = bad ? 0 : static_cast<int>(pr.y);
    const int d_q0 = P > 0 ? qpos[d_b0] : 0, d_q1 = P > 0 ? qpos[d_b1] : 0;
    const uint64_t badmask = __ballot(bad);
    for (int g0 = 0; g0 < np_; g0 += 4) {
      float ya[4], yr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (g0 + u >= np_) {                                   // wave-uniform: a short last group (one pair per wave on small batches)
          ya[u] = yr[u] = 0.f;
          continue;
        }
        const int j = g0 + u;
        const int b0 = __builtin_amdgcn_readlane(d_b0, j), b1 = __builtin_amdgcn_readlane(d_b1, j);
        const int q0 = __builtin_amdgcn_readlane(d_q0, j), q1 = __builtin_amdgcn_readlane(d_q1, j);
        const int b = t ? b1 : b0, q = t ? q1 : q0;
        if (q0 >= 0 && q1 >= 0) {                              // wave-uniform: both tuples are non-members (the scoring case)
          ya[u] = y_q[static_cast<int64_t>(q) * (2 * C) + cc];
          yr[u] = y_q[static_cast<int64_t>(q) * (2 * C) + C + cc];
        } else if (q < 0) {                                    // a member: b < E1
          const int64_t s = pos_edge[b], e = pos_edge[E1 + b];
          ya[u] = tw_member(t_node[s * (2 * C) + cc], cnt[s], ba);
          yr[u] = tw_member(t_node[e * (2 * C) + C + cc], cnt[e], br);
        } else {
          ya[u] = y_q[static_cast<int64_t>(q) * (2 * C) + cc];
          yr[u] = y_q[static_cast<int64_t>(q) * (2 * C) + C + cc];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (g0 + u >= np_) continue;
        const float row = tw_apply(ca, ya[u]) + tw_apply(cr, yr[u]);
        const float other = __shfl_xor(row, 32);
        float sacc = t == 0 ? row * other * wc : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);
        const int j = g0 + u;
        if (lane == 0 && j < np_ && !((badmask >> j) & 1ull)) out[j0 + j] = sacc + bb;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ operator form
// y[b] = T[node(b)] / cnt + bias (member) | T[node(b)] / sqrt(cnt (cnt + 1)) + hq[q] / (cnt + 1) + bias (otherwise):
// GCNConv(x, orientation) for an ARBITRARY [P, C] input x once T = (group sums of x) W^T and hq = (x[non-members]) W^T exist.
__global__ __launch_bounds__(kBlock) void tw_bcast_kernel(int64_t P, int C, int orient, const int32_t* __restrict__ tnode,
                                                          const int32_t* __restrict__ cnt, const int32_t* __restrict__ qpos,
                                                          const float* __restrict__ t_node, int64_t ldt,
                                                          const float* __restrict__ hq, int64_t ldh,
                                                          const float* __restrict__ bias, float* __restrict__ y, int64_t ldy) {
  const int64_t total = P * C;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t b = i / C;
    const int c = static_cast<int>(i - b * C);
    const int node = tnode[2 * b + orient], cn = cnt[node], q = qpos[b];
    const float bi = bias ? bias[c] : 0.f, t = t_node[static_cast<int64_t>(node) * ldt + c];
    float o;
    if (q < 0) {
      o = tw_member(t, cn, bi);
    } else {
      const float cf = static_cast<float>(cn);
      o = (cn > 0 ? t * __builtin_amdgcn_rsqf(cf * (cf + 1.0f)) : 0.f) +
          (hq ? hq[static_cast<int64_t>(q) * ldh + c] : 0.f) * __builtin_amdgcn_rcpf(cf + 1.0f) + bi;   // hq == NULL with non-member rows is a caller error (header): no fault, the self term is dropped
    }
    y[b * ldy + c] = o;
  }
}

// the same, four columns per thread (C % 4 == 0, 16-byte aligned rows): one descriptor read and one 16-byte streaming
// store per piece instead of a 64-bit division, three index loads and a 4-byte store per element (0.24 -> ~0.1 ms for
// 2.77 M rows x 32)
__global__ __launch_bounds__(kBlock) void tw_bcast4_kernel(int64_t P, int C4, int orient, const int32_t* __restrict__ tnode,
                                                           const int32_t* __restrict__ cnt, const int32_t* __restrict__ qpos,
                                                           const float* __restrict__ t_node, int64_t ldt,
                                                           const float* __restrict__ hq, int64_t ldh,
                                                           const float* __restrict__ bias, float* __restrict__ y, int64_t ldy) {
  const int c4 = threadIdx.x % C4, rin = threadIdx.x / C4, rpb = kBlock / C4;
  const float4 bi = bias ? *reinterpret_cast<const float4*>(bias + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t b = static_cast<int64_t>(blockIdx.x) * rpb + rin; b < P; b += static_cast<int64_t>(gridDim.x) * rpb) {
    const int node = tnode[2 * b + orient], cn = cnt[node], q = qpos[b];
    const float4 t = *reinterpret_cast<const float4*>(t_node + static_cast<int64_t>(node) * ldt + 4 * c4);
    float4 o;
    if (q < 0) {
      const float r = __builtin_amdgcn_rcpf(static_cast<float>(cn));
      o = make_float4(fmaf(t.x, r, bi.x), fmaf(t.y, r, bi.y), fmaf(t.z, r, bi.z), fmaf(t.w, r, bi.w));
    } else {
      const float cf = static_cast<float>(cn);
      const float ca = cn > 0 ? __builtin_amdgcn_rsqf(cf * (cf + 1.0f)) : 0.f, cb = __builtin_amdgcn_rcpf(cf + 1.0f);
      const float4 h = hq ? *reinterpret_cast<const float4*>(hq + static_cast<int64_t>(q) * ldh + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
      o = make_float4(t.x * ca + h.x * cb + bi.x, t.y * ca + h.y * cb + bi.y, t.z * ca + h.z * cb + bi.z,
                      t.w * ca + h.w * cb + bi.w);
    }
    store_stream(y + b * ldy + 4 * c4, o);
  }
}

bool make_norm(TwNorm* nm, const double* part, int64_t n_part, int64_t n_rows, const float* gw_a, const float* gb_a,
               const float* gms_a, float eps_a, int relu_a, const float* gw_r, const float* gb_r, const float* gms_r,
               float eps_r, int relu_r) {
  *nm = TwNorm{part, gw_a, gb_a, gms_a, gw_r, gb_r, gms_r, static_cast<double>(n_rows), eps_a, eps_r,
               static_cast<int>(n_part), relu_a, relu_r};
  return part && n_part >= 1 && n_part <= kTwMaxPart && n_rows >= 1;
}

}  // namespace
}  // namespace lpg

using namespace lpg;

LPG_EXPORT int64_t lpg_tuple_plan_bytes(int64_t P) {
  if (P < 0) return LPG_E_INVALID;
  size_t s = 0;
  (void)rocprim::exclusive_scan(nullptr, s, rocprim::make_transform_iterator(static_cast<const uint8_t*>(nullptr), FlagToI32()),
                                static_cast<int32_t*>(nullptr), int32_t(0), static_cast<size_t>(P + 1),
                                rocprim::plus<int32_t>(), hipStream_t(0), false);
  return align_up(P + 16, 256) + align_up(static_cast<int64_t>(s), 256) + 256;
}

LPG_EXPORT int lpg_tuple_plan_build(const int64_t* pos_edge, int64_t E1, const int64_t* pred_edge, int64_t Ep, int64_t n_node,
                                    const uint8_t* blocked, const int64_t* rowptr, int32_t* cnt, int32_t* qpos, int32_t* qinfo,
                                    int64_t q_cap, int32_t* tnode, int64_t* seg_a, int64_t* seg_r, int32_t* head,
                                    void* workspace, int64_t workspace_bytes, int32_t* err_flag, void* stream) {
  LPG_REQUIRE(E1 >= 0 && Ep >= 0 && n_node >= 1 && q_cap >= 0, "tuple_plan_build: bad sizes");
  LPG_REQUIRE(E1 % 2 == 0 && Ep % 2 == 0, "tuple_plan_build: E1 and Ep must be even (tuples come in partner pairs)");
  const int64_t P = E1 + Ep;
  LPG_REQUIRE(P < (int64_t(1) << 31) && n_node < (int64_t(1) << 31), "tuple_plan_build: more than 2^31 tuples / nodes");
  LPG_REQUIRE(rowptr && cnt && head && workspace && (P == 0 || (qpos && (E1 == 0 || pos_edge) && (Ep == 0 || pred_edge))) &&
                  (q_cap == 0 || qinfo),
              "tuple_plan_build: null pointer");
  if (workspace_bytes < lpg_tuple_plan_bytes(P)) {
    set_error("tuple_plan_build: workspace too small");
    return LPG_E_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  LPG_HIP(hipMemsetAsync(head, 0, 16, st));
  uint8_t* flags = static_cast<uint8_t*>(workspace);
  void* tmp = static_cast<char*>(workspace) + align_up(P + 16, 256);
  size_t tb = static_cast<size_t>(workspace_bytes - align_up(P + 16, 256));
  const int64_t work = P > n_node ? P : n_node;
  hipLaunchKernelGGL(tw_plan_flags_kernel, dim3(grid_for(work)), dim3(kBlock), 0, st, pos_edge, E1, pred_edge, Ep, n_node,
                     blocked, rowptr, cnt, flags, head, err_flag);
  if (P == 0) return check_launch("tuple_plan_build");
  if (blocked)
    LPG_HIP(rocprim::exclusive_scan(tmp, tb, rocprim::make_transform_iterator(static_cast<const uint8_t*>(flags), FlagToI32()),
                                    qpos, int32_t(0), static_cast<size_t>(P), rocprim::plus<int32_t>(), st, false));
  hipLaunchKernelGGL(tw_plan_fill_kernel, dim3(grid_for(P)), dim3(kBlock), 0, st, pos_edge, E1, pred_edge, Ep, n_node, flags,
                     blocked ? 1 : 0, qpos, qinfo, q_cap, tnode, seg_a, seg_r, head);
  return check_launch("tuple_plan_build");
}

namespace {
// rows per tile: about 4096 tiles (16 waves on each of 256 CUs), between one group of rows and a wave's 64 lanes
int tw_tile(int64_t n_node, int64_t Q) {
  int t = kTwRows;
  while (t < 64 && (n_node + Q) > int64_t(4096) * t) t *= 2;
  return t;
}
int64_t tw_blocks(int64_t n_node, int64_t Q) {
  const int t = tw_tile(n_node, Q);
  const int64_t g = ceil_div(ceil_div(n_node, t) + ceil_div(Q, t), kTwWaves);
  return g < 1 ? 1 : (g > kTwMaxBlocks ? kTwMaxBlocks : g);
}
}  // namespace

LPG_EXPORT int64_t lpg_tuple_layer_blocks(int64_t n_node, int64_t Q) {
  if (n_node < 0 || Q < 0) return LPG_E_INVALID;
  return tw_blocks(n_node, Q);
}

LPG_EXPORT int lpg_tuple_ax_small_f32(int64_t n_node, int64_t C, const float* y1, int64_t ldy, const int64_t* rowptr,
                                      const int32_t* col, const double* in_part, const float* in_gw, const float* in_gb,
                                      const float* in_gms, int64_t in_rb, int in_relu, float in_eps, float* x1n, float* ax,
                                      void* stream) {
  LPG_REQUIRE(n_node >= 1 && n_node < (int64_t(1) << 24) && C >= 1 && C <= 32 && ldy >= C, "tuple_ax_small: bad sizes");
  LPG_REQUIRE(y1 && rowptr && col && x1n && ax, "tuple_ax_small: null pointer");
  LPG_REQUIRE(!in_part || (in_rb >= 1 && in_rb <= 64), "tuple_ax_small: bad deferred-norm arguments");
  const TwNormIn nin{in_part, in_gw, in_gb, in_gms, static_cast<int>(in_rb), in_relu, static_cast<int>(n_node), in_eps};
  hipLaunchKernelGGL(tw_ax_small_kernel, dim3(static_cast<unsigned>(n_node)), dim3(kBlock), 0, as_stream(stream),
                     static_cast<int>(C), y1, ldy, rowptr, col, nin, x1n, ax);
  return check_launch("tuple_ax_small");
}

LPG_EXPORT int lpg_tuple_layer_f32(int mode, int64_t n_node, int64_t Q, int64_t C, const float* x, int64_t ldx, const float* ax,
                                   int64_t ldax, const float* hq_prev, const int32_t* cnt, const int32_t* qinfo,
                                   const float* w_a, const float* w_r, const float* b_a, const float* b_r, float* t_node,
                                   float* y_q, double* part, int64_t n_blocks, void* stream) {
  LPG_REQUIRE((mode == 1 || mode == 2) && n_node >= 1 && Q >= 0 && C >= 1 && C <= 32 && n_node + Q < (int64_t(1) << 31),
              "tuple_layer: needs mode 1 | 2, n_node >= 1, 1 <= C <= 32");
  LPG_REQUIRE(x && ax && cnt && w_a && w_r && b_a && b_r && part && (Q == 0 || (qinfo && y_q)), "tuple_layer: null pointer");
  LPG_REQUIRE(n_blocks == tw_blocks(n_node, Q), "tuple_layer: n_blocks must be lpg_tuple_layer_blocks(n_node, Q)");
  LPG_REQUIRE(reinterpret_cast<uintptr_t>(qinfo) % 16 == 0, "tuple_layer: qinfo must be 16-byte aligned");
  LPG_REQUIRE(mode == 1 || (hq_prev || Q == 0), "tuple_layer: mode 2 needs the previous layer's non-member rows");
  LPG_REQUIRE(ldx >= (mode == 1 ? C : 2 * C) && ldax >= (mode == 1 ? C : 2 * C), "tuple_layer: row strides too small");
  LPG_REQUIRE(n_node * (ldx > ldax ? ldx : ldax) < (int64_t(1) << 31) && (n_node + Q) * 2 * C < (int64_t(1) << 31),
              "tuple_layer: operands of 2^31 elements or more (32-bit element offsets inside)");
  TwLayerArgs a{};
  a.x = x; a.ax = ax; a.hq_prev = hq_prev; a.cnt = cnt; a.qinfo = qinfo;
  a.w_a = w_a; a.w_r = w_r; a.b_a = b_a; a.b_r = b_r; a.t_node = t_node; a.y_q = y_q; a.part = part;
  a.ldx = ldx; a.ldax = ldax; a.n_node = static_cast<int>(n_node); a.Q = static_cast<int>(Q); a.C = static_cast<int>(C);
  a.tile = tw_tile(n_node, Q);
  const dim3 grid(static_cast<unsigned>(n_blocks)), block(kTwBlock);
  hipStream_t st = as_stream(stream);
  if (mode == 2) hipLaunchKernelGGL((tw_layer_kernel<2>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((tw_layer_kernel<1>), grid, block, 0, st, a);
  return check_launch("tuple_layer");
}

LPG_EXPORT int lpg_tuple_fold_f64(const double* part, int64_t n_in, double* out, int64_t n_out, void* stream) {
  LPG_REQUIRE(part && out && n_in >= 1 && n_out >= 1 && n_out <= kTwMaxPart, "tuple_fold: bad argument");
  hipLaunchKernelGGL(tw_fold_kernel, dim3(static_cast<unsigned>(n_out)), dim3(kBlock), 0, as_stream(stream), part,
                     static_cast<int>(n_in), out);
  return check_launch("tuple_fold");
}

LPG_EXPORT int lpg_tuple_apply_f32(int64_t n_node, int64_t Q, int64_t C, const float* t_node, const float* y_q,
                                   const int32_t* cnt, const double* part, int64_t n_part, int64_t n_rows, const float* gw_a,
                                   const float* gb_a, const float* gms_a, float eps_a, int relu_a, const float* gw_r,
                                   const float* gb_r, const float* gms_r, float eps_r, int relu_r, const float* b_a,
                                   const float* b_r, float* ff, float* hq, void* stream) {
  LPG_REQUIRE(n_node >= 1 && Q >= 0 && C >= 1 && C <= 32 && n_node + Q < (int64_t(1) << 31), "tuple_apply: bad sizes");
  LPG_REQUIRE(t_node && cnt && b_a && b_r && ff && (Q == 0 || (y_q && hq)), "tuple_apply: null pointer");
  TwNorm nm;
  LPG_REQUIRE(make_norm(&nm, part, n_part, n_rows, gw_a, gb_a, gms_a, eps_a, relu_a, gw_r, gb_r, gms_r, eps_r, relu_r),
              "tuple_apply: 1 <= n_part <= %d partial blocks and n_rows >= 1 needed", kTwMaxPart);
  hipLaunchKernelGGL(tw_apply_kernel, dim3(grid_for((n_node + Q) * 64)), dim3(kBlock), 0, as_stream(stream),
                     static_cast<int>(n_node), static_cast<int>(Q), static_cast<int>(C), t_node, y_q, cnt, nm, b_a, b_r, ff, hq);
  return check_launch("tuple_apply");
}

LPG_EXPORT int lpg_tuple_readout_f32(int64_t n_node, int64_t Q, int64_t C, int64_t E1, int64_t P, const int64_t* pos_edge,
                                     const float* t_node, const float* y_q, const int32_t* cnt, const int32_t* qpos,
                                     const double* part, int64_t n_part, int64_t n_rows, const float* gw_a, const float* gb_a,
                                     const float* gms_a, float eps_a, int relu_a, const float* gw_r, const float* gb_r,
                                     const float* gms_r, float eps_r, int relu_r, const float* b_a, const float* b_r,
                                     const int64_t* idx, int64_t B, const float* w, const float* b0, float* out,
                                     int32_t* err_flag, void* stream) {
  LPG_REQUIRE(n_node >= 1 && Q >= 0 && C >= 1 && C <= 32 && E1 >= 0 && P >= E1 && B >= 0 && B < (int64_t(1) << 31),
              "tuple_readout: bad sizes");
  if (B == 0) return LPG_OK;
  LPG_REQUIRE(t_node && cnt && qpos && b_a && b_r && idx && w && out && (E1 == 0 || pos_edge) && (Q == 0 || y_q),
              "tuple_readout: null pointer");
  TwNorm nm;
  LPG_REQUIRE(make_norm(&nm, part, n_part, n_rows, gw_a, gb_a, gms_a, eps_a, relu_a, gw_r, gb_r, gms_r, eps_r, relu_r),
              "tuple_readout: 1 <= n_part <= %d partial blocks and n_rows >= 1 needed", kTwMaxPart);
  LPG_REQUIRE(reinterpret_cast<uintptr_t>(idx) % 16 == 0, "tuple_readout: idx must be 16-byte aligned");
  int ts = 1;
  while (ts < 64 && B > 8192 * ts) ts *= 2;
  hipLaunchKernelGGL(tw_readout_kernel, dim3(grid_for(ceil_div(B, ts) * 64)), dim3(kBlock), 0, as_stream(stream), static_cast<int>(n_node),
                     static_cast<int>(Q), static_cast<int>(C), E1, P, pos_edge, t_node, y_q, cnt, qpos, nm, b_a, b_r, idx,
                     static_cast<int>(B), w, b0, out, err_flag);
  return check_launch("tuple_readout");
}

LPG_EXPORT int lpg_tuple_bcast_f32(int64_t P, int64_t C, int orient, const int32_t* tnode, const int32_t* cnt,
                                   const int32_t* qpos, const float* t_node, int64_t ldt, const float* hq, int64_t ldh,
                                   const float* bias, float* y, int64_t ldy, void* stream) {
  LPG_REQUIRE(P >= 0 && C >= 1 && (orient == 0 || orient == 1) && ldt >= C && ldy >= C, "tuple_bcast: bad argument");
  if (P == 0) return LPG_OK;
  LPG_REQUIRE(tnode && cnt && qpos && t_node && y, "tuple_bcast: null pointer");
  auto al16 = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  const int64_t c4 = C / 4;
  if (C % 4 == 0 && c4 <= kBlock && kBlock % c4 == 0 && ldt % 4 == 0 && ldy % 4 == 0 && (!hq || ldh % 4 == 0) && al16(t_node) &&
      al16(y) && (!hq || al16(hq)) && (!bias || al16(bias)))
    hipLaunchKernelGGL(tw_bcast4_kernel, dim3(grid_for(P * c4)), dim3(kBlock), 0, as_stream(stream), P, static_cast<int>(c4),
                       orient, tnode, cnt, qpos, t_node, ldt, hq, ldh, bias, y, ldy);
  else
    hipLaunchKernelGGL(tw_bcast_kernel, dim3(grid_for(P * C)), dim3(kBlock), 0, as_stream(stream), P, static_cast<int>(C), orient,
                       tnode, cnt, qpos, t_node, ldt, hq, ldh, bias, y, ldy);
  return check_launch("tuple_bcast");
}

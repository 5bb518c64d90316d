// Dense feature linear (K5):  Y[n,m] = act(X[n,k] @ W^T + bias)   on the fp32-input matrix cores.
// Reference: GCNConv.lin (no bias, weight [out,in]) inside TwoWL/model/model.py:37 [3P PyG];
// SEAL's tf.matmul(X, W) with W [in,out] (SEAL/model/gnn.py:46,50,54,58) -> w_is_km; SEAL's dense
// layers relu(xW+b) (gnn.py:127,135).
//
// v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate (bitwise a k-ordered fmaf chain),
// which keeps the 1e-5 parity bar without any reduced-precision path.
//
// Two shapes of workgroup:
//  * ROWS (big n): a wave owns 32 rows and ALL of up to NT*32 output columns (NT accumulator tiles),
//    so X is read from HBM exactly once; per 64-deep K step the workgroup stages its four 32-row X
//    slices (16-byte loads) and the [NT*32, 64] W slice in LDS (+1 padded rows: conflict-free
//    ds_read_b32 fragments).  At GNN sizes (k, m <= 256) the op is HBM-bound on X and Y.
//  * SPLITK (small n, k > 64): the 4 waves share one 32-row tile and take every 4th K slice, partial
//    tiles folded through LDS in wave order: 4x the workgroups and a quarter of the dependent
//    staging steps -- the reference's own sizes ([620,256]x[256,32]) are latency bound.
#include "lpg_common.hpp"

namespace {
using namespace lpg;

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int KT = 64;       // k values staged per step
constexpr int LDP = KT + 1;  // padded LDS row

__device__ __forceinline__ void store_tile(const f32x16& acc, const float* __restrict__ bias, int act,
                                           float* __restrict__ y, int64_t ldy, int64_t n, int64_t m, int64_t row0,
                                           int64_t col, int lane) {
  // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  if (col >= m) return;
  const float b = bias ? bias[col] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < n) y[row * ldy + col] = apply_act(acc[r] + b, act);
  }
}

// FAST: every operand is 16-byte aligned (k % 4 == 0, W's contiguous dimension % 4 == 0, m % 4 == 0):
//   * register prefetch: the global loads of K step s+1 are issued right after the barrier that
//     publishes step s in LDS, so they fly during the MFMAs of step s;
//   * the output tile goes back through LDS and is written as whole 16-byte pieces of a row.
// !FAST: scalar staging with bounds selects (any alignment, any k / m).
template <int NT, bool FAST>
__global__ __launch_bounds__(kBlock, 2) void linear_rows_f32_kernel(const float* __restrict__ x, int64_t ldx,
                                                                    const float* __restrict__ w, int w_is_km,
                                                                    const float* __restrict__ bias, int act,
                                                                    float* __restrict__ y, int64_t ldy, int64_t n,
                                                                    int64_t k, int64_t m) {
  extern __shared__ float smem[];
  float(*Xs)[32][LDP] = reinterpret_cast<float(*)[32][LDP]>(smem);                       // [4][32][LDP]
  float(*Ws)[LDP] = reinterpret_cast<float(*)[LDP]>(smem + 4 * 32 * LDP);                // [NT*32][LDP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = static_cast<int64_t>(blockIdx.x) * 128 + wave * 32;
  const int64_t j0 = static_cast<int64_t>(blockIdx.y) * (NT * 32);
  const int fr = lane & 31, fk = lane >> 5;
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  auto mfma_step = [&]() {
#pragma unroll 4
    for (int kk = 0; kk < KT; kk += 2) {
      const float a = Xs[wave][fr][kk + fk];
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Ws[t * 32 + fr][kk + fk], acc[t], 0, 0, 0);
    }
  };

  if constexpr (FAST) {
    float4 xr[8], wr[NT * 2];   // X: 32 rows x 64 k = 512 pieces, 8 per lane; W: NT*512 pieces, NT*2 per thread
    auto load_stage = [&](int64_t k0) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int piece = it * 64 + lane, r = piece >> 4, seg = piece & 15;
        const int64_t row = row0 + r, kk = k0 + seg * 4;
        const bool ok = row < n && kk < k;
        xr[it] = *reinterpret_cast<const float4*>(x + (ok ? row * ldx + kk : 0));
      }
#pragma unroll
      for (int it = 0; it < NT * 2; ++it) {
        const int piece = it * kBlock + tid;
        int j, kk;
        if (w_is_km) { kk = piece / (NT * 8); j = (piece % (NT * 8)) * 4; }    // 4 consecutive output columns
        else { j = piece >> 4; kk = (piece & 15) * 4; }                        // 4 consecutive k
        const bool ok = k0 + kk < k && j0 + j < m;
        const int64_t off = w_is_km ? (k0 + kk) * m + j0 + j : (j0 + j) * k + k0 + kk;
        wr[it] = *reinterpret_cast<const float4*>(w + (ok ? off : 0));
      }
    };
    auto store_stage = [&](int64_t k0) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int piece = it * 64 + lane, r = piece >> 4, seg = piece & 15;
        const bool ok = row0 + r < n && k0 + seg * 4 < k;
        float* dst = &Xs[wave][r][seg * 4];
        dst[0] = ok ? xr[it].x : 0.f; dst[1] = ok ? xr[it].y : 0.f;
        dst[2] = ok ? xr[it].z : 0.f; dst[3] = ok ? xr[it].w : 0.f;
      }
#pragma unroll
      for (int it = 0; it < NT * 2; ++it) {
        const int piece = it * kBlock + tid;
        int j, kk;
        if (w_is_km) { kk = piece / (NT * 8); j = (piece % (NT * 8)) * 4; }
        else { j = piece >> 4; kk = (piece & 15) * 4; }
        const bool ok = k0 + kk < k && j0 + j < m;
        const float4 v = ok ? wr[it] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (w_is_km) { Ws[j][kk] = v.x; Ws[j + 1][kk] = v.y; Ws[j + 2][kk] = v.z; Ws[j + 3][kk] = v.w; }
        else { Ws[j][kk] = v.x; Ws[j][kk + 1] = v.y; Ws[j][kk + 2] = v.z; Ws[j][kk + 3] = v.w; }
      }
    };
    load_stage(0);
    for (int64_t k0 = 0; k0 < k; k0 += KT) {
      asm volatile("" ::: "memory");
      store_stage(k0);
      __syncthreads();
      if (k0 + KT < k) load_stage(k0 + KT);          // in flight while this step's MFMAs run
      mfma_step();
      __syncthreads();
    }
    // ---- epilogue through LDS: the wave's [32][NT*32] tile, then 16-byte row pieces
    constexpr int TW = NT * 32 + (NT == 4 ? 0 : 4);  // tile row (floats); 4 * 32 * TW * 4 B fits the staging area
    static_assert(4 * 32 * TW <= (4 * 32 + NT * 32) * LDP, "output tiles must fit in the staging LDS");
    float* tile = smem + wave * 32 * TW;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int c = t * 32 + (lane & 31);
      const float b = (bias && j0 + c < m) ? bias[j0 + c] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        tile[row * TW + c] = apply_act(acc[t][r] + b, act);
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0); the tile is wave-private
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < NT * 4; ++it) {            // 32 rows x NT*8 pieces = NT*256 pieces, NT*4 per lane
      const int piece = it * 64 + lane, r = piece / (NT * 8), c = (piece % (NT * 8)) * 4;
      const int64_t row = row0 + r, col = j0 + c;
      if (row < n && col < m)                        // m % 4 == 0: a piece is all-in or all-out
        *reinterpret_cast<float4*>(y + row * ldy + col) = *reinterpret_cast<const float4*>(&tile[r * TW + c]);
    }
  } else {
    for (int64_t k0 = 0; k0 < k; k0 += KT) {
      float xr[32];
#pragma unroll
      for (int it = 0; it < 32; ++it) {
        const int64_t r = row0 + it, kk = k0 + lane;
        xr[it] = x[(r < n ? r : n - 1) * ldx + (kk < k ? kk : k - 1)];
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int it = 0; it < 32; ++it) Xs[wave][it][lane] = (row0 + it < n && k0 + lane < k) ? xr[it] : 0.f;
#pragma unroll 4
      for (int it = 0; it < NT * 8; ++it) {
        const int idx = it * kBlock + tid;
        const int kk = w_is_km ? idx / (NT * 32) : idx & 63, j = w_is_km ? idx % (NT * 32) : idx >> 6;
        const bool ok = k0 + kk < k && j0 + j < m;
        const int64_t off = w_is_km ? (k0 + kk) * m + j0 + j : (j0 + j) * k + k0 + kk;
        Ws[j][kk] = ok ? w[off] : 0.f;
      }
      __syncthreads();
      mfma_step();
      __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) store_tile(acc[t], bias, act, y, ldy, n, m, row0, j0 + t * 32 + (lane & 31), lane);
  }
}

__global__ __launch_bounds__(kBlock) void linear_splitk_f32_kernel(const float* __restrict__ x, int64_t ldx,
                                                                   const float* __restrict__ w, int w_is_km,
                                                                   const float* __restrict__ bias, int act,
                                                                   float* __restrict__ y, int64_t ldy, int64_t n,
                                                                   int64_t k, int64_t m) {
  __shared__ float Xs[4][32][LDP];
  __shared__ float Ws[4][32][LDP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = static_cast<int64_t>(blockIdx.x) * 32;
  const int64_t j0 = static_cast<int64_t>(blockIdx.y) * 32;
  const int fr = lane & 31, fk = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int64_t k0 = static_cast<int64_t>(wave) * KT; k0 < k; k0 += 4 * KT) {
    // every wave stages its own X slice and its own W slice: no workgroup barrier in the loop.
    // Two phases with a compiler barrier between them: all 64 loads are issued from clamped
    // addresses first, the selects and LDS writes follow (written as `cond ? load : 0`, hipcc
    // branches around every load and waits vmcnt(0) 64 times -- 64 dependent round trips).
    float xr[32], wr[32];
#pragma unroll
    for (int it = 0; it < 32; ++it) {
      const int64_t r = row0 + it, kk = k0 + lane;
      xr[it] = x[(r < n ? r : n - 1) * ldx + (kk < k ? kk : k - 1)];
    }
#pragma unroll
    for (int it = 0; it < 32; ++it) {
      if (w_is_km) {            // W[k][m]: lanes 0-31 walk m for k-row 2*it + fk
        const bool ok = k0 + 2 * it + fk < k && j0 + fr < m;
        wr[it] = w[(ok ? k0 + 2 * it + fk : 0) * m + (ok ? j0 + fr : 0)];
      } else {                  // W[m][k]: lanes walk k for output column `it`
        const bool ok = k0 + lane < k && j0 + it < m;
        wr[it] = w[(ok ? j0 + it : 0) * k + (ok ? k0 + lane : 0)];
      }
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int it = 0; it < 32; ++it) {
      const int64_t r = row0 + it, kk = k0 + lane;
      Xs[wave][it][lane] = (r < n && kk < k) ? xr[it] : 0.f;
      if (w_is_km)
        Ws[wave][fr][2 * it + fk] = (k0 + 2 * it + fk < k && j0 + fr < m) ? wr[it] : 0.f;
      else
        Ws[wave][it][lane] = (k0 + lane < k && j0 + it < m) ? wr[it] : 0.f;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's LDS writes are done (wave-private slices)
    __builtin_amdgcn_wave_barrier();
#pragma unroll 8
    for (int kk = 0; kk < KT; kk += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Xs[wave][fr][kk + fk], Ws[wave][fr][kk + fk], acc, 0, 0, 0);
    __builtin_amdgcn_wave_barrier();
  }
  // fold the 4 partial tiles in wave order through LDS (reuse Xs as [4][16][64] floats)
  float* red = &Xs[0][0][0];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int r = 0; r < 16; ++r)
    acc[r] = ((red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane]) + red[(2 * 16 + r) * 64 + lane]) +
             red[(3 * 16 + r) * 64 + lane];
  store_tile(acc, bias, act, y, ldy, n, m, row0, j0 + (lane & 31), lane);
}

template <int NT>
int launch_rows(const float* x, int64_t ldx, const float* w, int w_is_km, const float* bias, int act, float* y,
                int64_t ldy, int64_t n, int64_t k, int64_t m, hipStream_t st) {
  const int64_t gx = ceil_div(n, 128), gy = ceil_div(m, NT * 32);
  LPG_REQUIRE(gx < (int64_t(1) << 31) && gy < 65536, "linear: grid too large");
  auto al16 = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  const bool fast = k % 4 == 0 && ldx % 4 == 0 && m % 4 == 0 && ldy % 4 == 0 && al16(x) && al16(w) && al16(y) &&
                    (w_is_km ? true : k % 4 == 0);
  const size_t lds = static_cast<size_t>(4 * 32 + NT * 32) * LDP * sizeof(float);
  const dim3 grid(static_cast<unsigned>(gx), static_cast<unsigned>(gy));
  if (fast)
    hipLaunchKernelGGL((linear_rows_f32_kernel<NT, true>), grid, dim3(kBlock), lds, st, x, ldx, w, w_is_km, bias, act, y,
                       ldy, n, k, m);
  else
    hipLaunchKernelGGL((linear_rows_f32_kernel<NT, false>), grid, dim3(kBlock), lds, st, x, ldx, w, w_is_km, bias, act,
                       y, ldy, n, k, m);
  return LPG_OK;
}
}  // namespace

LPG_EXPORT int lpg_linear_f32(const float* x, int64_t ldx, const float* w, int w_is_km, const float* bias, int act,
                              float* y, int64_t ldy, int64_t n, int64_t k, int64_t m, void* stream) {
  LPG_REQUIRE(act >= LPG_ACT_NONE && act <= LPG_ACT_TANH, "linear: unknown activation");
  LPG_REQUIRE(n >= 0 && k >= 1 && m >= 1 && ldx >= 1 && ldy >= m, "linear: bad sizes n=%lld k=%lld m=%lld",
              (long long)n, (long long)k, (long long)m);
  if (n == 0) return LPG_OK;
  LPG_REQUIRE(x && w && y, "linear: null pointer");
  hipStream_t st = as_stream(stream);
  {
    const int r = linear_x3_f32(x, ldx, w, w_is_km, bias, act, y, ldy, n, k, m, st);
    if (r < 0) return r;
    if (r > 0) return check_launch("linear (bf16 planes)");
  }
  {
    const int r = linear_wres_f32(x, ldx, w, w_is_km, bias, act, y, ldy, n, k, m, st);
    if (r < 0) return r;
    if (r > 0) return check_launch("linear (W-resident)");
  }
  // split K across the 4 waves when there is more than one K-slice and the problem is small enough
  // to be latency bound (few row tiles); big problems read X once with all columns per wave.
  // (measured, k = 288 / 448, m = 128: the split form wins up to n ~ 6 K -- 18 vs 32 us at n <= 2 K -- and loses from
  // 8 K on, 4.5x at n = 50 K: it re-reads X once per 32 output columns; profiles/r01/linear_small_n.log)
  if (k > KT && n <= 6144) {
    const int64_t gx = ceil_div(n, 32), gy = ceil_div(m, 32);
    LPG_REQUIRE(gy < 65536, "linear: grid too large");
    hipLaunchKernelGGL(linear_splitk_f32_kernel, dim3(static_cast<unsigned>(gx), static_cast<unsigned>(gy)),
                       dim3(kBlock), 0, st, x, ldx, w, w_is_km, bias, act, y, ldy, n, k, m);
    return check_launch("linear");
  }
  int rc;
  if (m <= 32) rc = launch_rows<1>(x, ldx, w, w_is_km, bias, act, y, ldy, n, k, m, st);
  else if (m <= 64) rc = launch_rows<2>(x, ldx, w, w_is_km, bias, act, y, ldy, n, k, m, st);
  else rc = launch_rows<4>(x, ldx, w, w_is_km, bias, act, y, ldy, n, k, m, st);      // wider m: column tiles of 128
  if (rc != LPG_OK) return rc;
  return check_launch("linear");
}

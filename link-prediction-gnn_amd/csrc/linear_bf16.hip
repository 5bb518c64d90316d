// Dense feature linear, bf16 storage (K5 for the bf16 configurations):
//   Y[n,m] = act(X[n,k] @ W^T + bias),  X/W/Y bfloat16, fp32 accumulate on v_mfma_f32_32x32x16_bf16.
// Reference anchor: same op as linear.hip (GCNConv.lin, SEAL's tf.matmul(X, W)); the bf16 storage is
// BASELINE.json configs[2] ("d = 128, bf16"), not something the fp32-only reference has.
//
// A wave owns a 32x32 output tile.  Fragment layout of the instruction: lane l (r = l & 31,
// h = l >> 5) holds A[row r][k = 8h .. 8h+7] and B[k = 8h .. 8h+7][col r] per 16-deep step, i.e.
// 8 CONTIGUOUS bf16 of a row-major [rows][k] tile -- so X and W ([m,k] layout) are staged row-major
// in LDS (rows padded by 16 B) and a fragment is one ds_read_b128.
#include "lpg_common.hpp"

namespace {
using namespace lpg;

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;

constexpr int KT = 64;        // k values staged per step (4 MFMA steps of 16)
constexpr int LDP = KT + 8;   // padded LDS row: 144 B, keeps 16-B alignment

// NT accumulator tiles per wave: a wave owns 32 rows and NT*32 output columns, so X is read once.
template <int NT>
__global__ __launch_bounds__(kBlock, 2) void linear_mfma_bf16_kernel(const uint16_t* __restrict__ x, int64_t ldx,
                                                                  const uint16_t* __restrict__ w, int w_is_km,
                                                                  const float* __restrict__ bias, int act,
                                                                  uint16_t* __restrict__ y, int64_t ldy, int64_t n,
                                                                  int64_t k, int64_t m, int vec_ok) {
  __shared__ __attribute__((aligned(16))) uint16_t Xs[4][32][LDP];
  __shared__ __attribute__((aligned(16))) uint16_t Ws[NT * 32][LDP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = static_cast<int64_t>(blockIdx.x) * 128 + wave * 32;
  const int64_t j0 = static_cast<int64_t>(blockIdx.y) * (NT * 32);
  const int fr = lane & 31, fk = lane >> 5;
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  for (int64_t k0 = 0; k0 < k; k0 += KT) {
    if (vec_ok) {
      // X: 32 rows x 128 B = 256 x 16 B pieces, 4 per lane; W [NT*32 cols][64 k]: NT pieces per thread.
      // Loads first, selects + LDS writes after a compiler barrier (see linear.hip).
      uint4 xr[4], wr[NT];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int piece = it * 64 + lane, r = piece >> 3, seg = piece & 7;
        const int64_t row = row0 + r, kk = k0 + seg * 8;
        const bool ok = row < n && kk < k;                 // k % 8 == 0: a piece is all-in or all-out
        xr[it] = *reinterpret_cast<const uint4*>(x + (ok ? row * ldx + kk : 0));
      }
      if (!w_is_km) {
#pragma unroll
        for (int it = 0; it < NT; ++it) {
          const int piece = it * kBlock + tid, j = piece >> 3, seg = piece & 7;
          const bool ok = j0 + j < m && k0 + seg * 8 < k;
          wr[it] = *reinterpret_cast<const uint4*>(w + (ok ? (j0 + j) * k + k0 + seg * 8 : 0));
        }
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int piece = it * 64 + lane, r = piece >> 3, seg = piece & 7;
        const bool ok = row0 + r < n && k0 + seg * 8 < k;
        *reinterpret_cast<uint4*>(&Xs[wave][r][seg * 8]) = ok ? xr[it] : make_uint4(0, 0, 0, 0);
      }
      if (!w_is_km) {
#pragma unroll
        for (int it = 0; it < NT; ++it) {
          const int piece = it * kBlock + tid, j = piece >> 3, seg = piece & 7;
          const bool ok = j0 + j < m && k0 + seg * 8 < k;
          *reinterpret_cast<uint4*>(&Ws[j][seg * 8]) = ok ? wr[it] : make_uint4(0, 0, 0, 0);
        }
      }
    } else {
#pragma unroll 4
      for (int it = 0; it < 32; ++it) {
        const int64_t r = row0 + it, kk = k0 + lane;
        Xs[wave][it][lane] = (r < n && kk < k) ? x[r * ldx + kk] : uint16_t(0);
      }
      if (!w_is_km)
        for (int it = 0; it < (NT * 32 * KT) / kBlock; ++it) {
          const int idx = it * kBlock + tid, j = idx >> 6, kk = idx & 63;
          Ws[j][kk] = (k0 + kk < k && j0 + j < m) ? w[(j0 + j) * k + k0 + kk] : uint16_t(0);
        }
    }
    if (w_is_km)                                           // W[k][m]: transpose while staging
      for (int it = 0; it < (NT * 32 * KT) / kBlock; ++it) {
        const int idx = it * kBlock + tid, kk = idx / (NT * 32), j = idx % (NT * 32);
        Ws[j][kk] = (k0 + kk < k && j0 + j < m) ? w[(k0 + kk) * m + j0 + j] : uint16_t(0);
      }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KT / 16; ++s) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(&Xs[wave][fr][s * 16 + 8 * fk]);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(&Ws[t * 32 + fr][s * 16 + 8 * fk]);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int64_t col = j0 + t * 32 + (lane & 31);
    if (col < m) {
      const float b = bias ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < n) y[row * ldy + col] = f32_to_bf16(apply_act(acc[t][r] + b, act));
      }
    }
  }
}

template <int NT>
void launch_bf16(const uint16_t* x, int64_t ldx, const uint16_t* w, int w_is_km, const float* bias, int act,
                 uint16_t* y, int64_t ldy, int64_t n, int64_t k, int64_t m, int vec_ok, hipStream_t st) {
  const int64_t gx = ceil_div(n, 128), gy = ceil_div(m, NT * 32);
  hipLaunchKernelGGL(linear_mfma_bf16_kernel<NT>, dim3(static_cast<unsigned>(gx), static_cast<unsigned>(gy)),
                     dim3(kBlock), 0, st, x, ldx, w, w_is_km, bias, act, y, ldy, n, k, m, vec_ok);
}
}  // namespace

LPG_EXPORT int lpg_linear_bf16(const uint16_t* x, int64_t ldx, const uint16_t* w, int w_is_km, const float* bias,
                               int act, uint16_t* y, int64_t ldy, int64_t n, int64_t k, int64_t m, void* stream) {
  LPG_REQUIRE(act >= LPG_ACT_NONE && act <= LPG_ACT_TANH, "linear_bf16: unknown activation");
  LPG_REQUIRE(n >= 0 && k >= 1 && m >= 1 && ldx >= 1 && ldy >= m, "linear_bf16: bad sizes");
  if (n == 0) return LPG_OK;
  LPG_REQUIRE(x && w && y, "linear_bf16: null pointer");
  LPG_REQUIRE(ceil_div(n, 128) < (int64_t(1) << 31) && ceil_div(m, 32) < 65536, "linear_bf16: grid too large");
  const int vec_ok = k % 8 == 0 && ldx % 8 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(w) % 16 == 0;
  hipStream_t st = as_stream(stream);
  if (m <= 32) launch_bf16<1>(x, ldx, w, w_is_km, bias, act, y, ldy, n, k, m, vec_ok, st);
  else if (m <= 64) launch_bf16<2>(x, ldx, w, w_is_km, bias, act, y, ldy, n, k, m, vec_ok, st);
  else launch_bf16<4>(x, ldx, w, w_is_km, bias, act, y, ldy, n, k, m, vec_ok, st);
  return check_launch("linear_bf16");
}

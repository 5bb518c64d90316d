// Dense feature linear (K5):  Y[n,m] = X[n,k] @ W^T (+ bias)   on the fp32-input matrix cores.
// Reference: GCNConv.lin (no bias, weight [out,in]) inside TwoWL/model/model.py:37 [3P PyG];
// SEAL's tf.matmul(X, W) with W [in,out] (SEAL/model/gnn.py:46,50,54,58) -> w_is_km.
//
// v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate (bitwise a k-ordered fmaf chain),
// which keeps the 1e-5 parity bar without any reduced-precision path.  A wave owns a 32x32
// output tile; X and W tiles are staged through LDS with coalesced global reads and a +1 padded
// row so the per-lane fragment reads (row = lane&31, k = 2*step + lane>>5) are conflict-free.
#include "lpg_common.hpp"

namespace {
using namespace lpg;

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int KT = 64;       // k values staged per step
constexpr int LDP = KT + 1;  // padded LDS row

__global__ __launch_bounds__(kBlock) void linear_mfma_f32_kernel(const float* __restrict__ x, int64_t ldx,
                                                                 const float* __restrict__ w, int w_is_km,
                                                                 const float* __restrict__ bias, int act,
                                                                 float* __restrict__ y, int64_t ldy, int64_t n,
                                                                 int64_t k, int64_t m) {
  __shared__ float Xs[4][32][LDP];
  __shared__ float Ws[32][LDP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = static_cast<int64_t>(blockIdx.x) * 128 + wave * 32;
  const int64_t j0 = static_cast<int64_t>(blockIdx.y) * 32;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  for (int64_t k0 = 0; k0 < k; k0 += KT) {
    // X tile: each wave stages its own 32 rows, one 256-byte row segment per instruction
#pragma unroll
    for (int it = 0; it < 32; ++it) {          // 32 independent loads in flight: one latency, not eight
      const int64_t r = row0 + it, kk = k0 + lane;
      Xs[wave][it][lane] = (r < n && kk < k) ? x[r * ldx + kk] : 0.f;
    }
    // W tile (32 output columns x KT), shared by the 4 waves
#pragma unroll
    for (int it = 0; it < (32 * KT) / kBlock; ++it) {
      const int idx = it * kBlock + tid;
      int j, kk;
      float v = 0.f;
      if (w_is_km) {            // W[k][m]: consecutive threads walk m
        kk = idx >> 5; j = idx & 31;
        if (k0 + kk < k && j0 + j < m) v = w[(k0 + kk) * m + j0 + j];
      } else {                  // W[m][k]: consecutive threads walk k
        j = idx >> 6; kk = idx & 63;
        if (k0 + kk < k && j0 + j < m) v = w[(j0 + j) * k + k0 + kk];
      }
      Ws[j][kk] = v;
    }
    __syncthreads();
    const int fr = lane & 31, fk = lane >> 5;
#pragma unroll 8
    for (int kk = 0; kk < KT; kk += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Xs[wave][fr][kk + fk], Ws[fr][kk + fk], acc, 0, 0, 0);
    __syncthreads();
  }
  // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int64_t col = j0 + (lane & 31);
  if (col < m) {
    const float b = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < n) y[row * ldy + col] = apply_act(acc[r] + b, act);
    }
  }
}
}  // namespace

LPG_EXPORT int lpg_linear_f32(const float* x, int64_t ldx, const float* w, int w_is_km, const float* bias, int act,
                              float* y, int64_t ldy, int64_t n, int64_t k, int64_t m, void* stream) {
  LPG_REQUIRE(act >= LPG_ACT_NONE && act <= LPG_ACT_TANH, "linear: unknown activation");
  LPG_REQUIRE(n >= 0 && k >= 1 && m >= 1 && ldx >= 1 && ldy >= m, "linear: bad sizes n=%lld k=%lld m=%lld",
              (long long)n, (long long)k, (long long)m);
  if (n == 0) return LPG_OK;
  LPG_REQUIRE(x && w && y, "linear: null pointer");
  const int64_t gx = ceil_div(n, 128), gy = ceil_div(m, 32);
  LPG_REQUIRE(gx < (int64_t(1) << 31) && gy < 65536, "linear: grid too large");
  hipLaunchKernelGGL(linear_mfma_f32_kernel, dim3(static_cast<unsigned>(gx), static_cast<unsigned>(gy)), dim3(kBlock),
                     0, as_stream(stream), x, ldx, w, w_is_km, bias, act, y, ldy, n, k, m);
  return check_launch("linear");
}

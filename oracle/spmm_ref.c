/* Oracle (TEST INFRASTRUCTURE): plain-C restatement of the CSR SpMM  Y = A @ X  in fp32 with the
 * SAME association order as the HIP kernel, so the GPU result can be checked BIT FOR BIT
 * ("bit-exact vs same-order fp32 CPU loop for the atomic-free kernels").
 *
 * Operation restated: the propagate step of GCNConv  out[dst] += w_e * h[src]  [3P PyG; call sites
 * TwoWL/model/model.py:73,77] / torch_sparse.spmm, on a destination-major CSR.
 *
 * Order (link-prediction-gnn_amd/csrc/spmm.hip): the nnz array is cut into chunks of `chunk` edges;
 * inside a chunk the edges of a row are added one by one in CSR order starting from +0 (weighted:
 * acc = fmaf(w, x, acc)); a row that spans several chunks is the left-to-right sum of its per-chunk
 * pieces -- unless it has more than `long_row` pieces after the first and groups > 1 (round 3: the fix-up of a hub row
 * is shared by the `groups` lane groups of a wave): then piece j (j = 0 for the first piece after the first) goes to
 * sub-sum j % groups, each sub-sum runs left to right from +0, the sub-sums are folded by a balanced binary tree
 * (s[i] + s[i ^ 1], then ^ 2, ...) and the result is added to the first piece.  groups = 1 or long_row < 0: always
 * left to right.  Only tests/ and bench.py's cpu_baseline may call this.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: no implicit FMA contraction)
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static float piece_sum(const int32_t* col, const float* val, const float* x, int64_t d, int64_t c, int64_t k, int64_t stop) {
  float acc = 0.0f;
  for (int64_t q = k; q < stop; ++q) {
    const float v = x[(int64_t)col[q] * d + c];
    acc = val ? fmaf(val[q], v, acc) : acc + v;
  }
  return acc;
}

void lpg_oracle_spmm_f32_chunked2(const int64_t* rowptr, const int32_t* col, const float* val, const float* x,
                                  int64_t n_rows, int64_t d, int64_t chunk, int64_t groups, int64_t long_row, float* y) {
  for (int64_t r = 0; r < n_rows; ++r) {
    float* out = y + r * d;
    const int64_t b = rowptr[r], e = rowptr[r + 1];
    memset(out, 0, (size_t)d * sizeof(float));
    if (b == e) continue;
    const int64_t first = b / chunk, last = (e - 1) / chunk;      /* chunks holding the first / last edge */
    if (groups > 1 && groups <= 64 && long_row >= 0 && last - first > long_row) {
      for (int64_t c = 0; c < d; ++c) {
        float sub[64];
        for (int64_t g = 0; g < groups; ++g) sub[g] = 0.0f;
        const int64_t stop0 = (first + 1) * chunk;
        const float head = piece_sum(col, val, x, d, c, b, stop0);
        for (int64_t ch = first + 1; ch <= last; ++ch) {
          const int64_t k = ch * chunk, stop = (ch + 1) * chunk < e ? (ch + 1) * chunk : e;
          sub[(ch - first - 1) % groups] += piece_sum(col, val, x, d, c, k, stop);
        }
        for (int64_t off = 1; off < groups; off <<= 1) {
          float nxt[64];
          for (int64_t g = 0; g < groups; ++g) nxt[g] = sub[g] + sub[g ^ off];
          for (int64_t g = 0; g < groups; ++g) sub[g] = nxt[g];
        }
        out[c] = head + sub[0];
      }
      continue;
    }
    int first_piece = 1;
    int64_t k = b;
    while (k < e) {
      int64_t stop = (k / chunk + 1) * chunk; /* end of the chunk holding edge k */
      if (stop > e) stop = e;
      for (int64_t c = 0; c < d; ++c) {
        float acc = 0.0f;
        for (int64_t q = k; q < stop; ++q) {
          const float v = x[(int64_t)col[q] * d + c];
          acc = val ? fmaf(val[q], v, acc) : acc + v;
        }
        /* pieces are folded left to right; the first one is taken as is (0 + p == p) */
        out[c] = first_piece ? acc : out[c] + acc;
      }
      first_piece = 0;
      k = stop;
    }
  }
}

void lpg_oracle_spmm_f32_chunked(const int64_t* rowptr, const int32_t* col, const float* val, const float* x,
                                 int64_t n_rows, int64_t d, int64_t chunk, float* y) {
  lpg_oracle_spmm_f32_chunked2(rowptr, col, val, x, n_rows, d, chunk, 1, -1, y);
}

/* the reference's CPU aggregation order (ATen scatter_add_ walks the edge list sequentially):
 * y[dst[e]] += w_e * x[src[e]] for e = 0..E-1.  Used as the single-thread "port" baseline. */
void lpg_oracle_scatter_add_f32(const int64_t* src, const int64_t* dst, const float* w, const float* x, int64_t n_edges,
                                int64_t n_rows, int64_t d, float* y) {
  memset(y, 0, (size_t)(n_rows * d) * sizeof(float));
  for (int64_t e = 0; e < n_edges; ++e) {
    const float* xs = x + src[e] * d;
    float* out = y + dst[e] * d;
    if (w)
      for (int64_t c = 0; c < d; ++c) out[c] += w[e] * xs[c];
    else
      for (int64_t c = 0; c < d; ++c) out[c] += xs[c];
  }
}

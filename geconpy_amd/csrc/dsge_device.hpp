// Wavefront-level dense linear algebra for one parameter draw (gfx950 / CDNA4).
//
// Execution model used by every kernel in this directory: ONE 64-lane wavefront owns ONE
// draw.  The workgroup is exactly one wave, so `__syncthreads()` is a wave-local LDS fence
// (no cross-wave barrier traffic).  Matrices of the draw live in LDS, row-major, zero-padded
// to NP = 8*BS rows/cols with an ODD leading dimension (LD = NP|1) so that both access
// patterns of the register-blocked product below are bank-conflict free for ds_read_b64:
//   - lanes form an 8x8 grid (lr = lane>>3, lc = lane&7); lane (lr,lc) owns the BSxBS block
//     of rows lr*BS.. and columns lc*BS.. of every n x n result and keeps it in VGPRs;
//   - a product C += A*B walks k and needs A[lr*BS+i][k] (8 distinct addresses per
//     instruction, broadcast over lc) and B[k][lc*BS+j] (8 distinct, broadcast over lr):
//     2*BS LDS reads feed BS*BS fp64 FMAs.
// FP64 FMA on the VALU runs at the same rate as v_mfma_f64_16x16x4 on gfx950, and an
// 8x8 lane grid tiles n = 40 with no padding waste (MFMA 16x16 tiles would pad 40 -> 48).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dsge {

template <int BS>
struct Tile {
  static constexpr int NP = 8 * BS;
  static constexpr int LD = NP | 1;
};

__device__ __forceinline__ void wave_sync() { __syncthreads(); }

// NaN-propagating max (np.max semantics, used by the induced 1-norm)
__device__ __forceinline__ double nanmax(double a, double b) { return (a > b || a != a) ? a : b; }

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) { return __shfl_xor(v, mask, 64); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_f64(v, m);
  return v;
}
__device__ __forceinline__ double wave_nanmax(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = nanmax(v, shfl_xor_f64(v, m));
  return v;
}

template <int BS>
__device__ __forceinline__ void blk_zero(double (&x)[BS][BS]) {
#pragma unroll
  for (int i = 0; i < BS; ++i)
#pragma unroll
    for (int j = 0; j < BS; ++j) x[i][j] = 0.0;
}

// register block <- global row-major [n][ldg] matrix (zero outside n x ncols)
template <int BS>
__device__ __forceinline__ void blk_load_global(double (&x)[BS][BS], const double* __restrict__ g, int n,
                                                int ncols, int ldg, int lr, int lc) {
#pragma unroll
  for (int i = 0; i < BS; ++i)
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      const int r = lr * BS + i, c = lc * BS + j;
      x[i][j] = (r < n && c < ncols) ? g[(size_t)r * ldg + c] : 0.0;
    }
}

template <int BS>
__device__ __forceinline__ void blk_store_global(const double (&x)[BS][BS], double* __restrict__ g, int n,
                                                 int ncols, int ldg, int lr, int lc, double scale = 1.0) {
#pragma unroll
  for (int i = 0; i < BS; ++i)
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      const int r = lr * BS + i, c = lc * BS + j;
      if (r < n && c < ncols) g[(size_t)r * ldg + c] = scale * x[i][j];
    }
}

template <int BS>
__device__ __forceinline__ void blk_store_lds(const double (&x)[BS][BS], double* s, int ld, int lr, int lc,
                                              double scale = 1.0) {
#pragma unroll
  for (int i = 0; i < BS; ++i)
#pragma unroll
    for (int j = 0; j < BS; ++j) s[(lr * BS + i) * ld + lc * BS + j] = scale * x[i][j];
}

template <int BS>
__device__ __forceinline__ void blk_load_lds(double (&x)[BS][BS], const double* s, int ld, int lr, int lc) {
#pragma unroll
  for (int i = 0; i < BS; ++i)
#pragma unroll
    for (int j = 0; j < BS; ++j) x[i][j] = s[(lr * BS + i) * ld + lc * BS + j];
}

// transposed block read: x[i][j] = s[col-block row][row-block col]  (for sym(X) = (X+X^T)/2)
template <int BS>
__device__ __forceinline__ void blk_load_lds_t(double (&x)[BS][BS], const double* s, int ld, int lr, int lc) {
#pragma unroll
  for (int i = 0; i < BS; ++i)
#pragma unroll
    for (int j = 0; j < BS; ++j) x[i][j] = s[(lc * BS + j) * ld + lr * BS + i];
}

// LDS matrix (NP x LD) <- global [n][ncols] (coalesced), zero padding
__device__ __forceinline__ void lds_load_matrix(double* s, int ld, int np_rows, int np_cols,
                                                const double* __restrict__ g, int n, int ncols, int lane) {
  for (int idx = lane; idx < np_rows * np_cols; idx += 64) {
    const int r = idx / np_cols, c = idx - r * np_cols;
    s[r * ld + c] = (r < n && c < ncols) ? g[(size_t)r * ncols + c] : 0.0;
  }
}

// acc += A(rows lr*BS.., :) * B(:, cols lc*BS..)           (TB = false)
// acc += A(rows lr*BS.., :) * B(rows lc*BS.., :)^T          (TB = true)
template <int BS, bool TB>
__device__ __forceinline__ void mm_acc(double (&acc)[BS][BS], const double* A, int lda, const double* B,
                                       int ldb, int K, int lr, int lc) {
  const double* a0 = A + lr * BS * lda;
  const double* b0 = TB ? (B + lc * BS * ldb) : (B + lc * BS);
#pragma unroll 2
  for (int k = 0; k < K; ++k) {
    double a[BS], b[BS];
#pragma unroll
    for (int i = 0; i < BS; ++i) a[i] = a0[i * lda + k];
#pragma unroll
    for (int j = 0; j < BS; ++j) b[j] = TB ? b0[j * ldb + k] : b0[k * ldb + j];
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
  }
}

// induced 1-norm (max absolute column sum, NaN-propagating) of a register-block matrix
template <int BS>
__device__ __forceinline__ double blk_norm1(const double (&x)[BS][BS]) {
  double m = 0.0;
#pragma unroll
  for (int j = 0; j < BS; ++j) {
    double cs = 0.0;
#pragma unroll
    for (int i = 0; i < BS; ++i) cs += fabs(x[i][j]);
    cs += shfl_xor_f64(cs, 8);
    cs += shfl_xor_f64(cs, 16);
    cs += shfl_xor_f64(cs, 32);
    m = (j == 0) ? cs : nanmax(m, cs);
  }
  m = nanmax(m, shfl_xor_f64(m, 1));
  m = nanmax(m, shfl_xor_f64(m, 2));
  m = nanmax(m, shfl_xor_f64(m, 4));
  return m;
}

template <int BS>
__device__ __forceinline__ double blk_maxabs(const double (&x)[BS][BS]) {
  double m = 0.0;
#pragma unroll
  for (int i = 0; i < BS; ++i)
#pragma unroll
    for (int j = 0; j < BS; ++j) m = nanmax(m, fabs(x[i][j]));
  return wave_nanmax(m);
}

// In-place Gauss-Jordan elimination with partial (row) pivoting on an LDS-resident augmented
// system [M | RHS]: rows 0..n-1, row stride ldw, M in columns [0,n), right-hand sides up to
// column `ncols` (zero padding columns in between are harmless).  On exit columns >= n of
// row j hold row j of M^-1 RHS.  Lanes own columns, so within one elimination step no lane
// reads what another lane writes; one wave_sync per step orders consecutive steps.
// A zero pivot yields inf/NaN in the result (no trap, no hang) -- callers test for it.
__device__ __forceinline__ void gauss_jordan_lds(double* W, int ldw, int n, int ncols, int lane) {
  for (int j = 0; j < n; ++j) {
    wave_sync();
    // pivot search, redundantly in every lane (broadcast reads): first max |W[i][j]|, i >= j
    int r = j;
    double best = fabs(W[j * ldw + j]);
    for (int i = j + 1; i < n; ++i) {
      const double v = fabs(W[i * ldw + j]);
      if (v > best) {
        best = v;
        r = i;
      }
    }
    const double wjj = W[j * ldw + j];
    const double inv = 1.0 / W[r * ldw + j];
    for (int c = j + 1 + lane; c < ncols; c += 64) {
      const double xr = W[r * ldw + c];
      const double xj = W[j * ldw + c];
      const double pr = xr * inv;
      W[j * ldw + c] = pr;
      if (r != j) W[r * ldw + c] = xj;
      for (int i = 0; i < n; ++i) {
        if (i == j) continue;
        const double f = (i == r) ? wjj : W[i * ldw + j];
        W[i * ldw + c] = fma(-f, pr, W[i * ldw + c]);
      }
    }
  }
  wave_sync();
}

}  // namespace dsge

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

#include "dsge_filter_conv.hpp"

namespace dsge {

template <int BS>
struct Tile {
  static constexpr int NP = 8 * BS;
  static constexpr int LD = NP | 1;
};

// internal per-draw status value: "a structure-exploiting kernel could not take this draw; the general
// kernel must".  Never returned to the caller.
constexpr int32_t DSGE_ST_INTERNAL_RERUN = 1 << 30;

__device__ __forceinline__ void wave_sync() { __syncthreads(); }

// Second passes of the kernel cascades visit only the draws an earlier kernel flagged DSGE_ST_INTERNAL_RERUN -- normally
// none.  True if none of the draws this workgroup would visit (indices blockIdx.x + i gridDim.x, through `order` if
// given) carries the flag: ONE parallel load per 64 draws instead of a dependent load per draw, so that an empty pass
// costs its launch and little else.
__device__ __forceinline__ bool rerun_pass_is_empty(const int32_t* __restrict__ status, int batch,
                                                    const int32_t* __restrict__ order = nullptr) {
  const int lane = threadIdx.x & 63;
  bool any = false;
  for (int b0 = blockIdx.x; b0 < batch; b0 += 64 * gridDim.x) {
    const int bi = b0 + lane * gridDim.x;
    if (bi < batch) any = any || ((status[order ? order[bi] : bi] & DSGE_ST_INTERNAL_RERUN) != 0);
  }
  return __ballot(any) == 0ull;
}

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
  // Unconditional loads from clamped (always valid) addresses, a scheduling barrier, then the selects: a conditional
  // load compiles to its own exec-masked block, and the scheduler, when short of registers, pairs each load with its
  // use -- BS * BS serial round trips to HBM instead of BS * BS loads in flight (32 of them opened the compact
  // cycle-reduction kernel).
#pragma unroll
  for (int i = 0; i < BS; ++i)
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      const int r = lr * BS + i, c = lc * BS + j;
      const int rc = r < n ? r : n - 1, cc = c < ncols ? c : ncols - 1;
      x[i][j] = g[(size_t)rc * ldg + cc];
    }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < BS; ++i)
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      const int r = lr * BS + i, c = lc * BS + j;
      x[i][j] = (r < n && c < ncols) ? x[i][j] : 0.0;
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
  // eight loads in flight per trip (clamped, unconditional addresses; the selects wait behind a scheduling barrier):
  // the plain loop -- conditional load, store to LDS -- is one round trip to memory per 64 elements
  const int total = np_rows * np_cols;
  for (int base = 0; base < total; base += 8 * 64) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * 64 + lane;
      const int ic = idx < total ? idx : total - 1;
      const int r = ic / np_cols, c = ic - r * np_cols;
      const int rc = r < n ? r : n - 1, cc = c < ncols ? c : ncols - 1;
      v[u] = g[(size_t)rc * ncols + cc];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * 64 + lane;
      const int r = idx / np_cols, c = idx - r * np_cols;
      if (idx < total) s[r * ld + c] = (r < n && c < ncols) ? v[u] : 0.0;
    }
  }
}

// Lane-strided loop `for (idx = lane; idx < total; idx += 64) store(idx, load(idx))` with U loads in flight per trip: the
// loads of a trip are issued (from clamped, always valid indices) before the first store waits for one.  The plain loop is
// one round trip to memory per 64 elements.  load: int -> V (any trivially copyable value), store: (int, V) -> void.
template <int U = 8, int NT = 64, class LoadF, class StoreF>
__device__ __forceinline__ void lane_loop_batched(int total, int lane, LoadF load, StoreF store) {
  for (int base = 0; base < total; base += U * NT) {
    decltype(load(0)) v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * NT + lane;
      v[u] = load(idx < total ? idx : total - 1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * NT + lane;
      if (idx < total) store(idx, v[u]);
    }
  }
}

// acc += A(rows lr*BS.., :) * B(:, cols lc*BS..)           (TB = false)
// acc += A(rows lr*BS.., :) * B(rows lc*BS.., :)^T          (TB = true)
template <int BS, bool TB>
__device__ __forceinline__ void mm_acc(double (&acc)[BS][BS], const double* A, int lda, const double* B,
                                       int ldb, int K, int lr, int lc) {
  // Software-pipelined by hand: two operand register sets ping-pong; the LDS loads of step k+1 are
  // issued before the FMAs of step k, and scheduling barriers keep hipcc from sinking them.
  const double* a0p = A + lr * BS * lda;
  const double* b0p = TB ? (B + lc * BS * ldb) : (B + lc * BS);
  double a0[BS], b0[BS], a1[BS], b1[BS];
#define MM_LOAD(a, b, k)                                                              \
  do {                                                                                \
    _Pragma("unroll") for (int i = 0; i < BS; ++i) a[i] = a0p[i * lda + (k)];         \
    _Pragma("unroll") for (int j = 0; j < BS; ++j) b[j] = TB ? b0p[j * ldb + (k)] : b0p[(k)*ldb + j]; \
  } while (0)
#define MM_FMA(a, b)                                                                  \
  do {                                                                                \
    _Pragma("unroll") for (int i = 0; i < BS; ++i)                                    \
      _Pragma("unroll") for (int j = 0; j < BS; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]); \
  } while (0)
  if (K <= 0) return;
  MM_LOAD(a0, b0, 0);
  int k = 0;
  for (; k + 2 <= K; k += 2) {
    MM_LOAD(a1, b1, k + 1);
    __builtin_amdgcn_sched_barrier(0);
    MM_FMA(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    const int kn = (k + 2 < K) ? k + 2 : K - 1;
    MM_LOAD(a0, b0, kn);
    __builtin_amdgcn_sched_barrier(0);
    MM_FMA(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (k < K) MM_FMA(a0, b0);
#undef MM_LOAD
#undef MM_FMA
}

// out = R - A(rows lr*BS.., :K) * B(:K, cols lc*BS..) with the inner products in twice the working precision (Dot2 of Ogita,
// Rump & Oishi 2005: every product split exactly by an FMA, every addition by TwoSum, the error terms summed on the side): the
// residual of the refinement step of the cycle-reduction solves.  A residual formed in plain float64 limits one step of
// iterative refinement to the accuracy of a backward-stable solver -- |dX| ~ cond(A1) eps |X|, what LAPACK's LU gives the
// reference --; with this one the corrected X is good to working precision as long as cond(A1) eps << 1, i.e. BETTER than
// the reference's on exactly the draws where two float64 algorithms cannot agree to 1e-9 anyway (tools/cr_accuracy_study.py:
// a 54-variable system with cond(A1) = 2..5e6 and |C| = 3e5, reference 0.8..4.5e-8 from the 40-digit T).  Ten operations per
// term instead of one FMA; only the refined iterations of the flagged draws run it (+0.04 ms on the 0.71 ms solver launch of
// the bench: a refined iteration of the 32-wide tile is two eliminations + 19 k cycles of this).  Tried: a plain first step and
// this one only when the first correction exceeds 1e-11 |X| (the level at which the error reaches 1e-9 in T) -- the refined
// draws of the bench are above that level too and then pay a third elimination: slower.  Contraction is switched off: the
// error-free transformations need p = fl(a b) and t = fl(s + p) as written.
template <int BS>
__device__ __forceinline__ void mm_residual_dot2(double (&out)[BS][BS], const double (&R)[BS][BS], const double* A, int lda,
                                                 const double* B, int ldb, int K, int lr, int lc) {
#pragma clang fp contract(off)
  const double* ap = A + lr * BS * lda;
  const double* bp = B + lc * BS;
  double s[BS][BS], c[BS][BS];
  blk_zero<BS>(s);
  blk_zero<BS>(c);
  for (int k = 0; k < K; ++k) {
    double a[BS], b[BS];
#pragma unroll
    for (int i = 0; i < BS; ++i) a[i] = ap[i * lda + k];
#pragma unroll
    for (int j = 0; j < BS; ++j) b[j] = bp[k * ldb + j];
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        const double p = a[i] * b[j];
        const double e = __builtin_fma(a[i], b[j], -p);  // a b = p + e exactly
        const double t = s[i][j] + p;
        const double z = t - s[i][j];
        const double err = (s[i][j] - (t - z)) + (p - z);  // s + p = t + err exactly
        s[i][j] = t;
        c[i][j] += e + err;
      }
  }
#pragma unroll
  for (int i = 0; i < BS; ++i)
#pragma unroll
    for (int j = 0; j < BS; ++j) out[i][j] = (R[i][j] - s[i][j]) - c[i][j];
}

// acc += A(:, rows lr*BS..)^T * B(:, cols lc*BS..)   i.e. (A' B) with A, B row-major in LDS: both operands are read
// along rows (conflict-free with the odd leading dimension), no transposed copy of A needed.
template <int BS>
__device__ __forceinline__ void mm_acc_ta(double (&acc)[BS][BS], const double* A, int lda, const double* B, int ldb, int K,
                                          int lr, int lc) {
  const double* a0p = A + lr * BS;
  const double* b0p = B + lc * BS;
  for (int k = 0; k < K; ++k) {
    double a[BS], b[BS];
#pragma unroll
    for (int i = 0; i < BS; ++i) a[i] = a0p[k * lda + i];
#pragma unroll
    for (int j = 0; j < BS; ++j) b[j] = b0p[k * ldb + j];
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

// ===========================================================================================
// Blocked Gauss-Jordan with partial pivoting inside each BS-column panel.
//
// Same elimination as gauss_jordan_lds (identical pivot sequence: the panel is eliminated column
// by column on a copy, exactly as the unblocked sweep would update those columns), but the
// O(n^2 ncols) work is done as rank-BS updates on BSxBS register blocks, and the O(n BS^2) panel
// work runs in registers with one matrix row per lane, pivot search by a DPP wave reduction and
// pivot-row broadcast by v_readlane.  Rows are never swapped: row r that served as pivot for
// column j ends up holding solution row j; `prow[j] = r` records it and gj_unpermute() restores
// the natural order of the right-hand-side columns.
//
// Block step for panel columns J (width bw), pivot rows Rk = (r_0..r_{bw-1}), M = W[Rk, J]:
//   Lhat[i,:] = W[i,J] M^-1 (i not in Rk),  Lhat[r_a,:] = e_a - M^-1[a,:]
//   Y = M^-1 W[Rk,:]                         W[i,:] -= Lhat[i,:] W[Rk,:]  (all rows)
// Lhat and M^-1 come out of the augmented panel elimination [W[:,J] | 0 or e_a].
// ===========================================================================================
namespace dsge {

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_move_u64(unsigned long long v) {
  int lo = (int)(v & 0xffffffffull), hi = (int)(v >> 32);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo;
}

// max over the 64 lanes of an unsigned 64-bit key (identity 0); result is wave-uniform
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  unsigned long long t;
  t = dpp_move_u64<0x111, 0xf>(v);  // row_shr:1
  v = t > v ? t : v;
  t = dpp_move_u64<0x112, 0xf>(v);  // row_shr:2
  v = t > v ? t : v;
  t = dpp_move_u64<0x114, 0xf>(v);  // row_shr:4
  v = t > v ? t : v;
  t = dpp_move_u64<0x118, 0xf>(v);  // row_shr:8
  v = t > v ? t : v;
  t = dpp_move_u64<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
  v = t > v ? t : v;
  t = dpp_move_u64<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
  v = t > v ? t : v;
  int lo = (int)(v & 0xffffffffull), hi = (int)(v >> 32);
  lo = __builtin_amdgcn_readlane(lo, 63);
  hi = __builtin_amdgcn_readlane(hi, 63);
  return ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo;
}

// max over the 64 lanes of an unsigned 32-bit key (identity 0); wave-uniform result.  One v_max_u32 with a
// DPP operand per step.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#define DSGE_DPP_MAX(CTRL, ROWS)                                                                   \
  do {                                                                                             \
    const unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xf, false);   \
    v = t > v ? t : v;                                                                             \
  } while (0)
  DSGE_DPP_MAX(0x111, 0xf);  // row_shr:1
  DSGE_DPP_MAX(0x112, 0xf);  // row_shr:2
  DSGE_DPP_MAX(0x114, 0xf);  // row_shr:4
  DSGE_DPP_MAX(0x118, 0xf);  // row_shr:8
  DSGE_DPP_MAX(0x142, 0xa);  // row_bcast:15 -> rows 1,3
  DSGE_DPP_MAX(0x143, 0xc);  // row_bcast:31 -> rows 2,3
#undef DSGE_DPP_MAX
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}

// sum over the 64 lanes (wave-uniform result): DPP row shifts + row broadcasts, no LDS traffic.
// Lanes that receive nothing get +0.0 (identity).  Summation order differs from wave_sum().
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_move_f64<0x111, 0xf>(v);  // row_shr:1
  v += dpp_move_f64<0x112, 0xf>(v);  // row_shr:2
  v += dpp_move_f64<0x114, 0xf>(v);  // row_shr:4
  v += dpp_move_f64<0x118, 0xf>(v);  // row_shr:8
  v += dpp_move_f64<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
  v += dpp_move_f64<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
  int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

// 1/sqrt(x) for x in a safe range: hardware estimate + two Newton steps (~1 ulp)
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = y * fma(-hx * y, y, 1.5);
  y = y * fma(-hx * y, y, 1.5);
  return y;
}

// 1/x for x in a safe range: hardware estimate + two Newton steps
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y * fma(-x, y, 2.0);
  y = y * fma(-x, y, 2.0);
  return y;
}

// reference implementation with plain shuffles (self-test of the DPP encodings)
__device__ __forceinline__ unsigned long long wave_max_u64_shfl(unsigned long long v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const unsigned long long t = __shfl_xor(v, m, 64);
    v = t > v ? t : v;
  }
  return v;
}

__device__ __forceinline__ double readlane_dyn_f64(double v, int src_lane_uniform) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, src_lane_uniform);
  hi = __builtin_amdgcn_readlane(hi, src_lane_uniform);
  return __hiloint2double(hi, lo);
}

// W: n rows x (ngroups*NP) columns, row stride ldw, matrix in columns [0,n).  Scratch in LDS:
// Lbuf (NP*BS doubles), Ybuf (BS * ngroups*NP doubles), prow (NP ints).
// Largest / smallest |pivot| of a solve beyond which it is refined once.  The ratio is a free but loose proxy of cond(A1)
// (1e2 .. 3e4 below it), and the blocked elimination loses more digits the wider its panels: on the tiles of up to 40
// variables (panels of <= 5 columns) 1e4 in ANY iteration keeps the fuzz campaigns at the fixed 1e-9 bar (about two
// SW-shaped draws in a thousand refine an iteration); on the wider tiles (panels of 6..8 columns, and the four-wavefront
// kernel) a draw with a ratio of 3.3e3 at cond 9e7 came out 7e-9 off (fuzz seed 2): 1e3 there (one or two draws in a
// hundred).  Every refinement lengthens its draw by an elimination, and the launch by that draw's tail: the 1e3 rule on the
// 32-wide tile of the bench costs 0.1 ms of the 0.65 ms solver launch, the 1e4 rule 0.04 ms.
// Round 5: 3e3 on the tiles of up to 40 variables.  The round-4 campaign (ten times the suite's trial counts) left ONE system in
// 3000 at 2.7e-9 from the 40-digit T (an ill-conditioned intermediate A1 with a pivot ratio between 3e3 and 1e4) against the
// suite's own 1e-9 bar; 3e3 takes it.  Cost on the bench: see profiles/r5/ (about three draws in a thousand more refine an iteration).
constexpr double CR_REFINE_PIVOT_RATIO = 1e3;
template <int BS>
__device__ __forceinline__ constexpr double cr_refine_ratio() {
  // (round 6: 1e3 on the tiles of up to 24 variables as well -- the final campaign's one system above the bar, n = 18, 2.0e-9 from
  //  the 40-digit T at cond(B + C T) = 17: an intermediate A1 again; the bench's 32- and 40-wide instances keep 3e3)
  return (BS <= 3 || BS > 5) ? 1e3 : 3e3;
}

template <int BS>
__device__ __forceinline__ void gauss_jordan_blocked(double* W, int ldw, int n, int ngroups, double* Lbuf, double* Ybuf,
                                                     int* prow, int lane, long long* ph, double& inv_min, double& inv_max,
                                                     double* rec_L = nullptr) {
  // rec_L (optional, nsteps x NP x BS doubles): the elimination's multipliers Lhat of every block step are kept, so that
  // gj_replay can apply the SAME elimination to further right-hand sides (with prow, which holds the pivot rows)
  // inv_min / inv_max: smallest and largest |1 / pivot| met (wave-uniform): their ratio is a free lower estimate of the
  // condition number, which crc_iterate uses to decide on a step of iterative refinement
  constexpr int NP = 8 * BS;
  const int lr = lane >> 3, lc = lane & 7;
  const int wcols = ngroups * NP;
  unsigned long long used = 0ull;
  const int nsteps = (n + BS - 1) / BS;
  for (int kb = 0; kb < nsteps; ++kb) {
    const int j0 = kb * BS;
    const int bw = (n - j0 < BS) ? (n - j0) : BS;
    wave_sync();
    const long long tk_p = ph ? clock64() : 0;
    // ---- panel: one matrix row per lane, augmented with the identity slots ------------
    double pw[BS], id[BS];
#pragma unroll
    for (int c = 0; c < BS; ++c) {
      pw[c] = (lane < n && c < bw) ? W[lane * ldw + j0 + c] : 0.0;
      id[c] = 0.0;
    }
    int rsel[BS];
    double inv_own = 1.0;  // pivot lanes: 1 / pivot of their row (the row is scaled once, after the panel)
#pragma unroll
    for (int c = 0; c < BS; ++c) rsel[c] = 0;
    // one pivot of the panel (column c: a compile-time constant after unrolling)
    auto panel_pivot = [&](int c) {
      // pivot = largest |entry| among the unused rows, compared on the high 32 bits of the double (sign
      // cleared; exponent + 20 mantissa bits; low 6 bits carry 63 - lane so that ties go to the first row):
      // within 2^-14 of the true maximum, which is all partial pivoting needs, at one v_max_u32 per DPP step
      const bool cand = (lane < n) && !((used >> lane) & 1ull);
      unsigned key = 0u;
      if (cand) key = (((unsigned)__double2hiint(pw[c]) & 0x7fffffffu) & ~63u) | (unsigned)(63 - lane);
      key = wave_max_u32(key);
      const int r = 63 - (int)(key & 63u);
      rsel[c] = r;
      used |= 1ull << r;
      const bool is_r = (lane == r);
      if (is_r) id[c] = 1.0;
      // broadcast the pivot lane's row scaled by 1 / pivot and eliminate everywhere else.  The pivot lane itself is
      // left alone (multiplier 0: no divergent branch) and scaled after the panel -- later columns only ever read
      // a row through its own elimination multiplier, which is consistent with the unscaled row.
      const double inv = fast_rcp(readlane_dyn_f64(pw[c], r));
      inv_min = fmin(inv_min, fabs(inv));
      inv_max = fmax(inv_max, fabs(inv));
      const double f = is_r ? 0.0 : pw[c];
      inv_own = is_r ? inv : inv_own;
#pragma unroll
      for (int c2 = 0; c2 < BS; ++c2) {
        if (c2 > c) pw[c2] = fma(-f, readlane_dyn_f64(pw[c2], r) * inv, pw[c2]);
        if (c2 <= c) id[c2] = fma(-f, readlane_dyn_f64(id[c2], r) * inv, id[c2]);
      }
    };
    // (round 6) a FULL panel -- every panel but possibly the last -- runs its BS pivots without the per-column test `c < bw`: the
    // (uniform) branches cut the chain of a panel into one basic block per pivot, which the scheduler cannot overlap; same
    // operations in the same order, bit-identical results
    if (bw == BS) {
#pragma unroll
      for (int c = 0; c < BS; ++c) panel_pivot(c);
    } else {
#pragma unroll
      for (int c = 0; c < BS; ++c)
        if (c < bw) panel_pivot(c);
    }
#pragma unroll
    for (int c = 0; c < BS; ++c) id[c] *= inv_own;  // (pw is dead from here on)
    // Lhat row of this lane: -id for ordinary rows; pivot lane r_a holds id = Minv[a,:] and
    // needs e_a - Minv[a,:]
    if (lane < NP) {
      double lh[BS];
#pragma unroll
      for (int b = 0; b < BS; ++b) lh[b] = -id[b];
#pragma unroll
      for (int a = 0; a < BS; ++a)
        if (a < bw && lane == rsel[a]) lh[a] += 1.0;
#pragma unroll
      for (int b = 0; b < BS; ++b) Lbuf[lane * BS + b] = (lane < n) ? lh[b] : 0.0;
      if (rec_L) {
#pragma unroll
        for (int b = 0; b < BS; ++b) rec_L[((size_t)kb * NP + lane) * BS + b] = (lane < n) ? lh[b] : 0.0;
      }
    }
    // pivot rows of W (original values) -> Ybuf, one column per lane
    for (int c = lane; c < wcols; c += 64) {
#pragma unroll
      for (int b = 0; b < BS; ++b) Ybuf[b * wcols + c] = (b < bw) ? W[rsel[b] * ldw + c] : 0.0;
    }
#pragma unroll
    for (int a = 0; a < BS; ++a)
      if (a < bw && lane == 0) prow[j0 + a] = rsel[a];
    wave_sync();
    const long long tk_t = ph ? clock64() : 0;
    if (ph) ph[0] += tk_t - tk_p;
    // ---- trailing update on register blocks: W[i,:] -= Lhat[i,:] Wpiv ---------------------
    double lh[BS][BS];
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int b = 0; b < BS; ++b) lh[i][b] = Lbuf[(lr * BS + i) * BS + b];
    for (int g = 0; g < ngroups; ++g) {
      // block columns at or left of the panel inside the matrix part are dead (never read again)
      if (g == 0 && lc <= kb) continue;
      const int c0 = g * NP + lc * BS;
      double wb[BS][BS], yb[BS][BS];
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) wb[i][j] = W[(lr * BS + i) * ldw + c0 + j];
#pragma unroll
      for (int b = 0; b < BS; ++b)
#pragma unroll
        for (int j = 0; j < BS; ++j) yb[b][j] = Ybuf[b * wcols + c0 + j];
#pragma unroll
      for (int b = 0; b < BS; ++b)
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) wb[i][j] = fma(-lh[i][b], yb[b][j], wb[i][j]);
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) W[(lr * BS + i) * ldw + c0 + j] = wb[i][j];
    }
    if (ph) ph[1] += clock64() - tk_t;
  }
  wave_sync();
}

template <int BS>
__device__ __forceinline__ void gauss_jordan_blocked(double* W, int ldw, int n, int ngroups, double* Lbuf, double* Ybuf,
                                                     int* prow, int lane, long long* ph = nullptr, double* rec_L = nullptr) {
  double lo = 1e300, hi = 0.0;
  gauss_jordan_blocked<BS>(W, ldw, n, ngroups, Lbuf, Ybuf, prow, lane, ph, lo, hi, rec_L);
}

// The elimination recorded by gauss_jordan_blocked (rec_L, prow) applied to ONE more right-hand-side column group Wg (NP columns,
// row stride ldw): Wg <- A^-1 Wg, rows in pivot order as there.  The arithmetic is that of the trailing update above, operation
// by operation (same multipliers, same pivot rows, same order of the FMAs): the result is bit-identical to eliminating
// [A | Wg] again, at the cost of the rank-BS updates alone -- no pivot search, no panel, nothing on the matrix part.
// Ybuf2: BS x NP doubles of scratch.  Syncs on entry and exit.
template <int BS>
__device__ __forceinline__ void gj_replay(double* Wg, int ldw, int n, const double* rec_L, const int* prow, double* Ybuf2,
                                          int lane) {
  constexpr int NP = 8 * BS;
  const int lr = lane >> 3, lc = lane & 7;
  const int nsteps = (n + BS - 1) / BS;
  for (int kb = 0; kb < nsteps; ++kb) {
    const int j0 = kb * BS;
    const int bw = (n - j0 < BS) ? (n - j0) : BS;
    wave_sync();
    for (int c = lane; c < NP; c += 64) {
#pragma unroll
      for (int b = 0; b < BS; ++b) Ybuf2[b * NP + c] = (b < bw) ? Wg[prow[j0 + (b < bw ? b : 0)] * ldw + c] : 0.0;
    }
    wave_sync();
    double lh[BS][BS], wb[BS][BS], yb[BS][BS];
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int b = 0; b < BS; ++b) lh[i][b] = rec_L[((size_t)kb * NP + lr * BS + i) * BS + b];
    const int c0 = lc * BS;
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) wb[i][j] = Wg[(lr * BS + i) * ldw + c0 + j];
#pragma unroll
    for (int b = 0; b < BS; ++b)
#pragma unroll
      for (int j = 0; j < BS; ++j) yb[b][j] = Ybuf2[b * NP + c0 + j];
#pragma unroll
    for (int b = 0; b < BS; ++b)
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) wb[i][j] = fma(-lh[i][b], yb[b][j], wb[i][j]);
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) Wg[(lr * BS + i) * ldw + c0 + j] = wb[i][j];
  }
  wave_sync();
}

// Restore natural row order of the right-hand-side column groups [g_first, ngroups): row j of the
// solution sits in row prow[j] of W.  Rows >= n are left untouched.
template <int BS>
__device__ __forceinline__ void gj_unpermute(double* W, int ldw, int n, int g_first, int ngroups, const int* prow,
                                             int lane) {
  constexpr int NP = 8 * BS;
  const int lr = lane >> 3, lc = lane & 7;
  for (int g = g_first; g < ngroups; ++g) {
    const int c0 = g * NP + lc * BS;
    double t[BS][BS];
#pragma unroll
    for (int i = 0; i < BS; ++i) {
      const int r = lr * BS + i;
      const int src = (r < n) ? prow[r] : r;
#pragma unroll
      for (int j = 0; j < BS; ++j) t[i][j] = W[src * ldw + c0 + j];
    }
    wave_sync();
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) W[(lr * BS + i) * ldw + c0 + j] = t[i][j];
    wave_sync();
  }
}

// Dispatch order for the Kalman launch: draws sorted by DESCENDING key (counting sort, keys clamped to 0..63; the order inside
// a bin is arbitrary).  The key is the number of cycle-reduction iterations of the draw: both grow with the persistence of
// the model (roots close to the unit circle), and a persistent model is the one whose covariance recursion reaches its
// fixed point late.  One workgroup.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void kalman_order_kernel(const int32_t* __restrict__ key, int batch,
                                                            int32_t* __restrict__ order, int never_value = -1) {
  // never_value >= 0: the key is a filter's first steady step (-1: never steady = never_value steps), binned by four steps
  // one histogram per wavefront (the 64 bins are hot: a single shared histogram serialises the LDS atomics of the whole
  // block), bases by a wave scan over the bins: 10 -> 4 us per 4096 draws with 1024 threads
  constexpr int NW = BLOCK / 64;
  __shared__ int hist[NW][64], base[NW][64];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  hist[w][lane] = 0;
  __syncthreads();
  for (int i = tid; i < batch; i += BLOCK) {
    int kq = key[i];
    if (never_value >= 0) kq = (kq < 0 ? never_value : kq) >> 2;
    kq = kq < 0 ? 0 : (kq > 63 ? 63 : kq);
    atomicAdd(&hist[w][kq], 1);
  }
  __syncthreads();
  if (w == 0) {  // lane b owns bin 63 - b (descending keys first)
    const int bin = 63 - lane;
    int tot = 0;
#pragma unroll
    for (int q = 0; q < NW; ++q) tot += hist[q][bin];
    int incl = tot;  // inclusive scan over the lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    int acc = incl - tot;
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      base[q][bin] = acc;
      acc += hist[q][bin];
    }
  }
  __syncthreads();
  for (int i = tid; i < batch; i += BLOCK) {
    int kq = key[i];
    if (never_value >= 0) kq = (kq < 0 ? never_value : kq) >> 2;
    kq = kq < 0 ? 0 : (kq > 63 ? 63 : kq);
    order[atomicAdd(&base[w][kq], 1)] = i;
  }
}



// Persistence key of a draw for the dispatch order of the Kalman launches: a rough spectral-radius estimate of the
// transition matrix, rho ~ (||T^k v|| / ||v||)^(1/k) after k = 24 power-iteration steps from v = 1 (one row of T per lane
// in registers, the vector exchanged by v_readlane), mapped to key = 8 * -log2(1 - rho) clamped to 0..63 -- finer towards
// the unit circle, where the covariance recursion of the filter converges slowly.  Failed draws (status != 0) get key 0.
// One wavefront per draw, n <= 64; a few thousand cycles per draw.
template <int NMAX>
__global__ __launch_bounds__(64) void persistence_key_kernel(const double* __restrict__ T, const int32_t* __restrict__ status,
                                                             int batch, int n, int32_t* __restrict__ key) {
  const int lane = threadIdx.x;
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    if (status && status[draw] != 0) {
      if (lane == 0) key[draw] = 0;
      continue;
    }
    const double* Tg = T + (size_t)draw * n * n;
    double row[NMAX];
#pragma unroll
    for (int c = 0; c < NMAX; ++c) row[c] = (lane < n && c < n) ? Tg[(size_t)lane * n + c] : 0.0;
    double v = (lane < n) ? 1.0 : 0.0, lg = 0.0;
    for (int it = 0; it < 24; ++it) {
      double a0 = 0.0, a1 = 0.0;
#pragma unroll
      for (int c = 0; c < NMAX; c += 2) {
        if (c < n) {
          a0 = fma(row[c], readlane_dyn_f64(v, c), a0);
          a1 = fma(row[c + 1], readlane_dyn_f64(v, c + 1), a1);
        }
      }
      v = a0 + a1;
      if ((it & 3) == 3) {  // renormalise every fourth step
        const double nrm = sqrt(wave_sum_dpp(v * v));
        if (!(nrm > 0.0) || !(nrm < 1e300)) break;
        lg += log2(nrm);
        v *= 1.0 / nrm;
      }
    }
    // ||v_0|| = sqrt(n): rho^24 ~ 2^lg / sqrt(n)
    const double rho = exp2((lg - 0.5 * log2((double)n)) / 24.0);
    double kq = (rho < 1.0) ? -8.0 * log2(1.0 - rho) : 63.0;
    if (!(kq == kq)) kq = 0.0;
    kq = kq < 0.0 ? 0.0 : (kq > 63.0 ? 63.0 : kq);
    if (lane == 0) key[draw] = (int32_t)kq;
  }
}

}  // namespace dsge

// Models with 65 .. 96 variables (round 4): one WORKGROUP per draw instead of one wavefront.
//
// The kernels of dsge_kernels.hpp / dsge_cr_*.hpp give every lane one row or one 8 x 8 register block of an n <= 64 matrix and
// keep all operands of cycle reduction in 64 lanes' registers and LDS.  The reference has no size limit
// (gEconpy/model/statespace.py:822-839; packaged examples of the sims_2024 class sit at the edge of 64), so this file restates
// the same recursion -- _cycle_reduction_core, gEconpy/solvers/cycle_reduction.py:127-183; the scan variant :246-294 -- for
// padded sizes NP = 80 and 96 with 640 / 512 threads per draw:
//
//   * elimination  X = A1^-1 [A0 | A2]  (:151-153, np.linalg.solve): Gauss-Jordan with partial pivoting on the augmented
//     n x 3n matrix, every thread holding a TR x TC register block of each of the three matrices (3 x 96 x 96 doubles = 221 KB:
//     registers are the only on-chip memory that holds them).  Pivoting is IMPLICIT -- rows never move; pivot column j's row is
//     broadcast through LDS, every other row eliminates against it, and the solution's row order is restored by the scatter
//     that writes X -- so a step costs two barriers, ~14 LDS reads and 3 TR TC FMAs per thread;
//   * the four products A0 X0, A0 X2, A2 X0, A2 X2 (:155-169) on the FP64 matrix core (v_mfma_f64_16x16x4_f64: 16 x 16 tiles,
//     K streamed from two LDS panels of NP x (NP + 2) doubles -- the odd-ish stride makes the 16 x 4 A fragment and the 4 x 16
//     B fragment of a half-wave land on 32 distinct bank pairs), tiles dealt round-robin to the wavefronts, the update of
//     A1, A1_hat (read-modify-write in the workgroup's L2-resident workspace) and the induced 1-norms of the stopping rule
//     (:171-179) in the epilogues;
//   * T = -A1_hat^-1 A and, from the same elimination, R = -A1_hat^-1 D (= -(C T + B)^-1 D at convergence, as in cr_kernel).
//
// The filter does not grow with n: T has non-zero columns only for state variables, so the state-space
// model restricted to F = {states} u {observed variables} is EXACT, and with |F| <= 64 the existing filter kernels run on the
// gathered model (big_compress_kernel).  |F| > 64 is refused with DSGE_ERR_TOO_LARGE.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dsge_hip.h"
#include "dsge_device.hpp"

namespace dsge {

typedef double big_v4f64 __attribute__((ext_vector_type(4)));

template <int NP_, int TR_, int TC_>
struct BigCfg {
  static constexpr int NP = NP_, TR = TR_, TC = TC_;
  static constexpr int GY = NP / TR, GX = NP / TC, NT = GY * GX, NW = NT / 64;
  static constexpr int LD = NP + 2, MT = NP / 16;
  static_assert(NP % 16 == 0 && NP % TR == 0 && NP % TC == 0 && NT % 64 == 0 && NT <= 1024 && NP <= 128, "tile grid");
  // LDS (doubles): left panel, right panel (NP x LD each) | column-sum partials (MT x NP) | reduction scratch (64).  The
  // elimination runs while no product is in flight: its broadcast buffers alias the left panel.
  static constexpr int OFF_R = NP * LD, OFF_PART = 2 * NP * LD, OFF_RED = OFF_PART + MT * NP;
  static constexpr size_t lds_bytes = (size_t)(OFF_RED + 64) * 8;
  static constexpr size_t ws_doubles = 6 * (size_t)NP * NP;  // A0, A1, A2, A1_hat, X0, X2 per workgroup
  // elimination buffers inside the left panel
  static constexpr int E_COL = 0, E_ROW = 2 * NP, E_PIV = 5 * NP;  // col[2][NP] | row[3][NP] | int pivcol[NP]
  static_assert(E_PIV + NP / 2 + 1 <= NP * LD, "elimination buffers fit the left panel");
};

// ---- register blocks <-> the workgroup's padded NP x NP workspace matrices ---------------------------------------------------------
template <class Cfg>
__device__ __forceinline__ void big_tile_load(double (&t)[Cfg::TR][Cfg::TC], const double* __restrict__ G, int r0, int c0) {
#pragma unroll
  for (int i = 0; i < Cfg::TR; ++i)
#pragma unroll
    for (int jc = 0; jc < Cfg::TC; ++jc) t[i][jc] = G[(size_t)(r0 + i) * Cfg::NP + c0 + jc];
}

// NP x NP workspace matrix -> LDS panel (row stride LD), 16 bytes per thread and trip
template <class Cfg>
__device__ __forceinline__ void big_panel_load(double* __restrict__ buf, const double* __restrict__ G, int tid) {
  constexpr int NP = Cfg::NP, H = NP / 2;
  for (int idx = tid; idx < NP * H; idx += Cfg::NT) {
    const int r = idx / H, c2 = idx - r * H;
    const double2 v = *reinterpret_cast<const double2*>(G + (size_t)r * NP + 2 * c2);
    *reinterpret_cast<double2*>(buf + r * Cfg::LD + 2 * c2) = v;
  }
}

// ---- Gauss-Jordan elimination of [t1 | t0 | t2] with implicit partial pivoting -----------------------------------------------------
// On exit t1 is (a row permutation of) the identity and physical row r of t0, t2 holds row pivcol[r] of t1_in^-1 [t0_in | t2_in];
// pivcol (int[NP], LDS) is valid after the closing barrier.  A zero pivot divides by zero: the NaN / Inf flows into the
// stopping rule's norm like LAPACK's singular-matrix error does in the reference (cycle_reduction.py:176-177).
template <class Cfg>
__device__ __forceinline__ void big_eliminate(double (&t1)[Cfg::TR][Cfg::TC], double (&t0)[Cfg::TR][Cfg::TC],
                                              double (&t2)[Cfg::TR][Cfg::TC], int n, double* __restrict__ lds, int r0, int c0,
                                              int tid) {
  constexpr int NP = Cfg::NP, TR = Cfg::TR, TC = Cfg::TC;
  double* rowb = lds + Cfg::E_ROW;
  int* pivcol = reinterpret_cast<int*>(lds + Cfg::E_PIV);
  const int lane = tid & 63;
  unsigned long long used_lo = 0ull, used_hi = 0ull;
  __syncthreads();  // (the panel this aliases is no longer read)
  for (int j = 0; j < n; ++j) {
    double* colb = lds + Cfg::E_COL + (j & 1) * NP;
    // column j of t1: the multipliers of this step and the pivot candidates
#pragma unroll
    for (int jc = 0; jc < TC; ++jc)
      if (c0 + jc == j) {
#pragma unroll
        for (int i = 0; i < TR; ++i) colb[r0 + i] = t1[i][jc];
      }
    __syncthreads();
    // pivot: largest |.| among the rows not used yet (every wavefront runs the same search; NaN counts as the largest)
    int p;
    {
      const int i0 = lane, i1 = lane + 64;
      double v0 = -1.0, v1 = -1.0;
      if (i0 < n && !((used_lo >> i0) & 1ull)) {
        const double x = colb[i0];
        v0 = (x != x) ? INFINITY : fabs(x);
      }
      if (i1 < n && !((used_hi >> (i1 - 64)) & 1ull)) {
        const double x = colb[i1];
        v1 = (x != x) ? INFINITY : fabs(x);
      }
      double v = v0;
      int bi = i0;
      if (v1 > v0) {
        v = v1;
        bi = i1;
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const double ov = __shfl_xor(v, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ov > v || (ov == v && oi < bi)) {
          v = ov;
          bi = oi;
        }
      }
      p = __builtin_amdgcn_readfirstlane(bi);
    }
    if (p < 64)
      used_lo |= 1ull << p;
    else
      used_hi |= 1ull << (p - 64);
    // the pivot row of all three matrices
#pragma unroll
    for (int i = 0; i < TR; ++i)
      if (r0 + i == p) {
#pragma unroll
        for (int jc = 0; jc < TC; ++jc) {
          rowb[c0 + jc] = t1[i][jc];
          rowb[NP + c0 + jc] = t0[i][jc];
          rowb[2 * NP + c0 + jc] = t2[i][jc];
        }
      }
    if (tid == 0) pivcol[p] = j;
    __syncthreads();
    const double dinv = 1.0 / colb[p];
    double pr1[TC], pr0[TC], pr2[TC];
#pragma unroll
    for (int jc = 0; jc < TC; ++jc) {
      pr1[jc] = rowb[c0 + jc] * dinv;
      pr0[jc] = rowb[NP + c0 + jc] * dinv;
      pr2[jc] = rowb[2 * NP + c0 + jc] * dinv;
    }
#pragma unroll
    for (int i = 0; i < TR; ++i) {
      const double f = colb[r0 + i];
#pragma unroll
      for (int jc = 0; jc < TC; ++jc) {
        t1[i][jc] = fma(-f, pr1[jc], t1[i][jc]);
        t0[i][jc] = fma(-f, pr0[jc], t0[i][jc]);
        t2[i][jc] = fma(-f, pr2[jc], t2[i][jc]);
      }
      if (r0 + i == p) {  // the pivot row itself: scaled, not eliminated
#pragma unroll
        for (int jc = 0; jc < TC; ++jc) {
          t1[i][jc] = pr1[jc];
          t0[i][jc] = pr0[jc];
          t2[i][jc] = pr2[jc];
        }
      }
    }
  }
  __syncthreads();  // pivcol complete; the next step-1 writes of a following elimination cannot overtake this one's reads
}

// solution rows back in order: physical row r -> row pivcol[r] of the NP x NP workspace matrix (rows >= n stay zero)
template <class Cfg>
__device__ __forceinline__ void big_scatter_rows(const double (&t)[Cfg::TR][Cfg::TC], double* __restrict__ G, int n,
                                                 const double* __restrict__ lds, int r0, int c0) {
  const int* pivcol = reinterpret_cast<const int*>(lds + Cfg::E_PIV);
#pragma unroll
  for (int i = 0; i < Cfg::TR; ++i) {
    const int r = r0 + i;
    if (r < n) {
      int q = pivcol[r];
      q = q < 0 ? 0 : (q >= n ? n - 1 : q);  // (NaN input: the search may have left garbage; stay inside the matrix)
#pragma unroll
      for (int jc = 0; jc < Cfg::TC; ++jc) G[(size_t)q * Cfg::NP + c0 + jc] = t[i][jc];
    }
  }
}

// ---- C = L R on the matrix core: L, R = the two LDS panels; epi(tile row, tile col, row0, col, acc): acc[e] is element
// (row0 + 4 e, col) (layout of v_mfma_f64_16x16x4_f64 probed in tools/mfma_probe) ----------------------------------------------------
template <class Cfg, class Epi>
__device__ __forceinline__ void big_gemm(const double* __restrict__ bufL, const double* __restrict__ bufR, int tid, Epi epi) {
  constexpr int MT = Cfg::MT, LD = Cfg::LD, NP = Cfg::NP;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int t = wave; t < MT * MT; t += Cfg::NW) {
    const int ti = t / MT, tj = t - ti * MT;
    big_v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    const double* pa = bufL + (16 * ti + (lane & 15)) * LD + (lane >> 4);
    const double* pb = bufR + (lane >> 4) * LD + 16 * tj + (lane & 15);
#pragma unroll 4
    for (int k4 = 0; k4 < NP / 4; ++k4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * k4], pb[4 * k4 * LD], acc, 0, 0, 0);
    epi(ti, tj, 16 * ti + (lane >> 4), 16 * tj + (lane & 15), acc);
  }
}

// column sums of |tile| into the partial array (deterministic: one writer per (tile row, column))
template <class Cfg>
__device__ __forceinline__ void big_colsum_part(double* __restrict__ part, int ti, int col, big_v4f64 v, int lane) {
  double s = fabs(v[0]) + fabs(v[1]) + fabs(v[2]) + fabs(v[3]);
  s += shfl_xor_f64(s, 16);
  s += shfl_xor_f64(s, 32);
  if (lane < 16) part[ti * Cfg::NP + col] = s;
}

// NaN-propagating maximum over the workgroup; `red`: NW doubles of LDS.  Barriers inside (entry and exit).
template <class Cfg>
__device__ __forceinline__ double big_block_nanmax(double v, double* __restrict__ red, int tid) {
  v = wave_nanmax(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double m = red[0];
#pragma unroll
  for (int w = 1; w < Cfg::NW; ++w) m = nanmax(m, red[w]);
  return m;
}
template <class Cfg>
__device__ __forceinline__ double big_block_sum(double v, double* __restrict__ red, int tid) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double s = red[0];
#pragma unroll
  for (int w = 1; w < Cfg::NW; ++w) s += red[w];
  return s;
}

// induced 1-norm from the partial column sums of the last product (barrier before reading them)
template <class Cfg>
__device__ __forceinline__ double big_norm1(const double* __restrict__ part, double* __restrict__ red, int tid) {
  __syncthreads();
  double m = 0.0;
  for (int c = tid; c < Cfg::NP; c += Cfg::NT) {
    double cs = 0.0;
#pragma unroll
    for (int ti = 0; ti < Cfg::MT; ++ti) cs += part[ti * Cfg::NP + c];
    m = nanmax(m, cs);
  }
  return big_block_nanmax<Cfg>(m, red, tid);
}

// ---- cycle reduction (njit rule; scan_mode: the scan variant's rule and 1e-16 stabilisation) ----------------------------------------
template <class Cfg>
__global__ __launch_bounds__(Cfg::NT) void cr_big_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                         const double* __restrict__ C, int batch, int n, int max_iter, double tol,
                                                         double* __restrict__ ws, double* __restrict__ T_out,
                                                         int32_t* __restrict__ status, int32_t* __restrict__ n_iter_out,
                                                         int scan_mode, const double* __restrict__ D, int k,
                                                         double* __restrict__ R_out) {
  constexpr int NP = Cfg::NP, TR = Cfg::TR, TC = Cfg::TC, NT = Cfg::NT;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* bufL = lds;
  double* bufR = lds + Cfg::OFF_R;
  double* part = lds + Cfg::OFF_PART;
  double* red = lds + Cfg::OFF_RED;
  const int tid = threadIdx.x, lane = tid & 63;
  const int ty = tid / Cfg::GX, tx = tid - ty * Cfg::GX, r0 = ty * TR, c0 = tx * TC;
  double* W = ws + (size_t)blockIdx.x * Cfg::ws_doubles;
  double *A0g = W, *A1g = W + NP * NP, *A2g = W + 2 * NP * NP, *Ahg = W + 3 * NP * NP, *X0g = W + 4 * NP * NP,
         *X2g = W + 5 * NP * NP;
  for (int idx = tid; idx < NP * NP; idx += NT) {  // (rows >= n of X0, X2 are never written again)
    X0g[idx] = 0.0;
    X2g[idx] = 0.0;
  }
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    const size_t off = (size_t)draw * n * n;
    __syncthreads();
    for (int idx = tid; idx < NP * NP; idx += NT) {
      const int r = idx / NP, c = idx - r * NP;
      const bool in = r < n && c < n;
      const size_t g = off + (size_t)r * n + c;
      const double b = in ? B[g] : 0.0;
      A0g[idx] = in ? A[g] : 0.0;
      A1g[idx] = b;
      Ahg[idx] = b;
      A2g[idx] = in ? C[g] : 0.0;
    }
    __syncthreads();
    bool converged = false, saw_nan = false;
    int it = 0;
    double t1[TR][TC], t0[TR][TC], t2[TR][TC];
    for (; it < max_iter;) {
      big_tile_load<Cfg>(t1, A1g, r0, c0);
      big_tile_load<Cfg>(t0, A0g, r0, c0);
      big_tile_load<Cfg>(t2, A2g, r0, c0);
      if (scan_mode) {  // stabilize(A1): 1e-16 on the diagonal of the solve only (shared.py:6-9)
#pragma unroll
        for (int i = 0; i < TR; ++i)
#pragma unroll
          for (int jc = 0; jc < TC; ++jc)
            if (r0 + i == c0 + jc && r0 + i < n) t1[i][jc] += 1e-16;
      }
      big_eliminate<Cfg>(t1, t0, t2, n, lds, r0, c0, tid);
      big_scatter_rows<Cfg>(t0, X0g, n, lds, r0, c0);
      big_scatter_rows<Cfg>(t2, X2g, n, lds, r0, c0);
      __syncthreads();
      // m00 = A0 X0 -> A0 := -m00
      big_panel_load<Cfg>(bufL, A0g, tid);
      big_panel_load<Cfg>(bufR, X0g, tid);
      __syncthreads();
      big_gemm<Cfg>(bufL, bufR, tid, [&](int ti, int, int row0, int col, big_v4f64 v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) A0g[(size_t)(row0 + 4 * e) * NP + col] = -v[e];
        big_colsum_part<Cfg>(part, ti, col, v, lane);
      });
      const double nrm0 = big_norm1<Cfg>(part, red, tid);
      // m02 = A0 X2 -> A1 -= m02
      big_panel_load<Cfg>(bufR, X2g, tid);
      __syncthreads();
      big_gemm<Cfg>(bufL, bufR, tid, [&](int, int, int row0, int col, big_v4f64 v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) A1g[(size_t)(row0 + 4 * e) * NP + col] -= v[e];
      });
      __syncthreads();
      // m22 = A2 X2 -> A2 := -m22
      big_panel_load<Cfg>(bufL, A2g, tid);
      __syncthreads();
      big_gemm<Cfg>(bufL, bufR, tid, [&](int ti, int, int row0, int col, big_v4f64 v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) A2g[(size_t)(row0 + 4 * e) * NP + col] = -v[e];
        big_colsum_part<Cfg>(part, ti, col, v, lane);
      });
      const double nrm2 = big_norm1<Cfg>(part, red, tid);
      // m20 = A2 X0 -> A1 -= m20, A1_hat -= m20
      big_panel_load<Cfg>(bufR, X0g, tid);
      __syncthreads();
      big_gemm<Cfg>(bufL, bufR, tid, [&](int, int, int row0, int col, big_v4f64 v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const size_t o = (size_t)(row0 + 4 * e) * NP + col;
          A1g[o] -= v[e];
          Ahg[o] -= v[e];
        }
      });
      __syncthreads();
      ++it;
      if (nrm0 < tol) {
        if (nrm2 < tol || scan_mode) {  // the scan variant tests the A0 norm only (cycle_reduction.py:269-277)
          converged = true;
          break;
        }
      } else if (nrm0 != nrm0) {
        saw_nan = true;
        break;
      }
    }
    const bool solve_T = converged || (scan_mode && !saw_nan);
    const bool want_R = (R_out != nullptr) && !scan_mode;
    const size_t offk = (size_t)draw * n * k;
    if (solve_T) {
      // T = -A1_hat^-1 A  (cycle_reduction.py:181); with D the same elimination gives R = -A1_hat^-1 D
      big_tile_load<Cfg>(t1, Ahg, r0, c0);
#pragma unroll
      for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int jc = 0; jc < TC; ++jc) {
          const int r = r0 + i, c = c0 + jc;
          if (scan_mode && r == c && r < n) t1[i][jc] += 1e-16;
          t0[i][jc] = (r < n && c < n) ? A[off + (size_t)r * n + c] : 0.0;
          t2[i][jc] = (want_R && r < n && c < k) ? D[offk + (size_t)r * k + c] : 0.0;
        }
      big_eliminate<Cfg>(t1, t0, t2, n, lds, r0, c0, tid);
      const int* pivcol = reinterpret_cast<const int*>(lds + Cfg::E_PIV);
#pragma unroll
      for (int i = 0; i < TR; ++i) {
        const int r = r0 + i;
        if (r < n) {
          int q = pivcol[r];
          q = q < 0 ? 0 : (q >= n ? n - 1 : q);
#pragma unroll
          for (int jc = 0; jc < TC; ++jc) {
            const int c = c0 + jc;
            if (c < n) T_out[off + (size_t)q * n + c] = -t0[i][jc];
            if (want_R && c < k) R_out[offk + (size_t)q * k + c] = -t2[i][jc];
          }
        }
      }
    } else {
      for (int idx = tid; idx < n * n; idx += NT) T_out[off + idx] = 0.0;
      if (R_out != nullptr && !scan_mode)
        for (int idx = tid; idx < n * k; idx += NT) R_out[offk + idx] = 0.0;
    }
    if (tid == 0) {
      status[draw] = solve_T ? DSGE_ST_OK : (DSGE_ST_NOT_CONVERGED | (saw_nan ? DSGE_ST_NAN : 0));
      if (n_iter_out) n_iter_out[draw] = it;
    }
  }
}

// ---- selection matrix and policy residual for a given T (assemble_kernel's do_selection part, n > 64):
//      R = -(C T + B)^-1 D (shared.py:74-75), resid = sum((A + (B + C T) T)^2); failed draws: R = 0, resid = inf ------------------------
template <class Cfg>
__global__ __launch_bounds__(Cfg::NT) void selection_big_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                                const double* __restrict__ C, const double* __restrict__ D,
                                                                const double* __restrict__ T, int batch, int n, int k,
                                                                double* __restrict__ ws, double* __restrict__ R_out,
                                                                double* __restrict__ resid_out,
                                                                const int32_t* __restrict__ status) {
  constexpr int NP = Cfg::NP, TR = Cfg::TR, TC = Cfg::TC, NT = Cfg::NT;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* bufL = lds;
  double* bufR = lds + Cfg::OFF_R;
  double* red = lds + Cfg::OFF_RED;
  const int tid = threadIdx.x;
  const int ty = tid / Cfg::GX, tx = tid - ty * Cfg::GX, r0 = ty * TR, c0 = tx * TC;
  double* W = ws + (size_t)blockIdx.x * Cfg::ws_doubles;
  double *Cg = W, *Tg = W + NP * NP, *Mg = W + 2 * NP * NP;
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    const size_t off = (size_t)draw * n * n, offk = (size_t)draw * n * k;
    if (status && status[draw] != 0) {
      if (R_out)
        for (int idx = tid; idx < n * k; idx += NT) R_out[offk + idx] = 0.0;
      if (resid_out && tid == 0) resid_out[draw] = INFINITY;
      continue;
    }
    __syncthreads();
    for (int idx = tid; idx < NP * NP; idx += NT) {
      const int r = idx / NP, c = idx - r * NP;
      const bool in = r < n && c < n;
      const size_t g = off + (size_t)r * n + c;
      Cg[idx] = in ? C[g] : 0.0;
      Tg[idx] = in ? T[g] : 0.0;
    }
    __syncthreads();
    big_panel_load<Cfg>(bufL, Cg, tid);
    big_panel_load<Cfg>(bufR, Tg, tid);
    __syncthreads();
    big_gemm<Cfg>(bufL, bufR, tid, [&](int, int, int row0, int col, big_v4f64 v) {  // M = B + C T
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = row0 + 4 * e;
        Mg[(size_t)r * NP + col] = (r < n && col < n) ? B[off + (size_t)r * n + col] + v[e] : 0.0;
      }
    });
    __syncthreads();
    if (resid_out) {  // A + M T
      big_panel_load<Cfg>(bufL, Mg, tid);
      __syncthreads();
      double s = 0.0;
      big_gemm<Cfg>(bufL, bufR, tid, [&](int, int, int row0, int col, big_v4f64 v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = row0 + 4 * e;
          if (r < n && col < n) {
            const double x = A[off + (size_t)r * n + col] + v[e];
            s = fma(x, x, s);
          }
        }
      });
      s = big_block_sum<Cfg>(s, red, tid);
      if (tid == 0) resid_out[draw] = s;
      __syncthreads();
    }
    if (R_out) {
      double t1[TR][TC], t0[TR][TC], t2[TR][TC];
      big_tile_load<Cfg>(t1, Mg, r0, c0);
#pragma unroll
      for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int jc = 0; jc < TC; ++jc) {
          const int r = r0 + i, c = c0 + jc;
          t0[i][jc] = (r < n && c < k) ? D[offk + (size_t)r * k + c] : 0.0;
          t2[i][jc] = 0.0;
        }
      big_eliminate<Cfg>(t1, t0, t2, n, lds, r0, c0, tid);
      const int* pivcol = reinterpret_cast<const int*>(lds + Cfg::E_PIV);
#pragma unroll
      for (int i = 0; i < TR; ++i) {
        const int r = r0 + i;
        if (r < n) {
          int q = pivcol[r];
          q = q < 0 ? 0 : (q >= n ? n - 1 : q);
#pragma unroll
          for (int jc = 0; jc < TC; ++jc)
            if (c0 + jc < k) R_out[offk + (size_t)q * k + c0 + jc] = -t0[i][jc];
        }
      }
    }
  }
}

// ---- the filtered variables of the batch: bit j of mask[0..1] = some draw has a non-zero column j in A (a state variable:
// T = -(B + C T)^-1 A has a non-zero column exactly there) or in Z (an observed variable) -------------------------------------------
__global__ __launch_bounds__(128) void big_mask_kernel(const double* __restrict__ A, const double* __restrict__ Z, int z_batched,
                                                       int batch, int n, int p, unsigned long long* __restrict__ mask) {
  const int j = threadIdx.x;
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    bool st = false, ob = false;
    if (j < n) {
      const double* a = A + (size_t)draw * n * n;
      for (int r = 0; r < n; ++r) st = st || (a[(size_t)r * n + j] != 0.0);
      if (z_batched || draw == 0) {
        const double* z = Z + (z_batched ? (size_t)draw * p * n : 0);
        for (int r = 0; r < p; ++r) ob = ob || (z[(size_t)r * n + j] != 0.0);
      }
    }
    const unsigned long long bs = __ballot(st), bo = __ballot(ob);
    if ((threadIdx.x & 63) == 0) {
      const int w = threadIdx.x >> 6;
      if (bs) atomicOr(&mask[w], bs);      // [0..1]: states
      if (bo) atomicOr(&mask[2 + w], bo);  // [2..3]: observed
    }
  }
}

struct BigIndex {
  uint8_t idx[64];
  int u;
};

// gathered model: T_r = T[F, F], R_r = R[F, :], Z_r = Z[:, F]
__global__ __launch_bounds__(256) void big_compress_kernel(const double* __restrict__ T, const double* __restrict__ R,
                                                           const double* __restrict__ Z, int z_batched, int batch, int n, int k,
                                                           int p, BigIndex F, double* __restrict__ T_r, double* __restrict__ R_r,
                                                           double* __restrict__ Z_r) {
  const int u = F.u, tid = threadIdx.x;
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    const double* t = T + (size_t)draw * n * n;
    const double* r = R + (size_t)draw * n * k;
    for (int idx = tid; idx < u * u; idx += 256) {
      const int i = idx / u, j = idx - i * u;
      T_r[(size_t)draw * u * u + idx] = t[(size_t)F.idx[i] * n + F.idx[j]];
    }
    for (int idx = tid; idx < u * k; idx += 256) {
      const int i = idx / k, c = idx - i * k;
      R_r[(size_t)draw * u * k + idx] = r[(size_t)F.idx[i] * k + c];
    }
    if (z_batched || draw == 0) {
      const double* z = Z + (z_batched ? (size_t)draw * p * n : 0);
      double* zr = Z_r + (z_batched ? (size_t)draw * p * u : 0);
      for (int idx = tid; idx < p * u; idx += 256) {
        const int i = idx / u, j = idx - i * u;
        zr[idx] = z[(size_t)i * n + F.idx[j]];
      }
    }
  }
}

}  // namespace dsge

// Cycle reduction on the column-compact form of the quadratic matrix equation A + B T + C T^2 = 0.
//
// In a DSGE model A = dF/dy_{t-1} has non-zero columns only for the state variables S (|S| = s) and
// C = dF/dy_{t+1} only for the forward-looking variables L (|L| = l); gEconpy's own tests rely on the
// first fact (tests/model/test_perturbation.py:201-203).  Cycle reduction preserves both supports:
//   X0 = A1^-1 A0 and A0' = -A0 X0 vanish outside the columns S,  X2 = A1^-1 A2 and A2' = -A2 X2 outside L.
// So the iteration (gEconpy/solvers/cycle_reduction.py:127-183) runs on R = [A0[:,S] | A2[:,L]], n x (s+l):
//   X  = A1^-1 R                                   (blocked Gauss-Jordan on [A1 | R], 2 column groups)
//   P1 = A0[:,S] X[S,:] = [m00 | m02],  P2 = A2[:,L] X[L,:] = [m20 | m22]       (K = s and K = l)
//   A1 -= scatter(m02 -> columns L) ; A1 -= scatter(m20 -> columns S) ; A1_hat -= scatter(m20)
//   R  = [-m00 | -m22]
// Dropping the exactly-zero columns only removes additions of +0.0, so T is bit-identical to the dense
// kernel's (tests/test_gpu_parity.py::test_cycle_reduction_compact_equals_dense) at
//   Gauss-Jordan 2/3 of the work, products (s^2 + 2 s l + l^2)/(4 n^2) of it  (SW-shaped: 900/6400).
// A draw with s + l > 8*BS does not fit the compact tile; it is flagged DSGE_ST_INTERNAL_RERUN and the
// dense cr_kernel picks it up.
#pragma once
#include "dsge_device.hpp"

#include "../../include/dsge_hip.h"

namespace dsge {

// Bank conflicts of the even (BS = 4) tile, measured 39 % of its LDS cycles.  ds_read_b128 is served in four groups of 16
// lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- i.e., in the 8 x 8 lane grid, block rows 0..3 with
// the column halves (low, high, high, low).  In 16-byte slots a block row's access is a run of four same-parity slots:
// slot = lr (4 sigma + kappa) + i sigma + 2 lc (mod 16), sigma = LDW / 2, kappa = a skew per block row.  Four runs need the
// 16 slots exactly; with delta = 4 sigma + kappa odd the runs of block rows 0 and 2 are both even and would have to be
// complementary (2 delta + 8 = 8 mod 16, i.e. delta = 0 or 8: a contradiction), with delta even all four are even: a
// linear skew is 2-way at best (LDW = 66: 4-way on two slots).  The XOR swizzle chunk ^= [0, 1, 8, 9][lr & 3] is conflict-free
// for block accesses but costs an XOR per k-step in mm_acc (operand columns at arbitrary offsets) and in the row-per-lane
// panel reads; LDS is active 10 % of the kernel's cycles, so it was not built.
template <int BS>
struct CrcSmem {
  static constexpr int NP = Tile<BS>::NP, LDW = 2 * NP + (BS % 2 == 0 ? 2 : 1);  // even tiles: rows of a register block are 16-byte aligned (b128 LDS accesses; half the conflict cycles of the odd stride on the 4 x 4 tile: 0.90 -> 0.84 ms)
  // W = [A1 | R], Gauss-Jordan scratch (Lbuf NP*BS, Ybuf BS*2NP), ints: prow, cmap, posS, posL, rsrc
  static constexpr size_t bytes = sizeof(double) * (size_t)(NP * LDW + NP * BS + BS * 2 * NP) + sizeof(int) * 5 * NP;
};

// non-zero column mask of a register-block matrix (bit v = column v has a non-zero or NaN entry)
template <int BS>
__device__ __forceinline__ unsigned long long blk_colmask(const double (&x)[BS][BS]) {
  unsigned long long mask = 0ull;
#pragma unroll
  for (int j = 0; j < BS; ++j) {
    bool nz = false;
#pragma unroll
    for (int i = 0; i < BS; ++i) nz = nz || (x[i][j] != 0.0);
    unsigned long long b = __ballot(nz);  // bit lr*8+lc
    b |= b >> 32;
    b |= b >> 16;
    b |= b >> 8;
    b &= 0xffull;  // bit lc: some row block has a non-zero in its column lc*BS+j
    while (b) {
      const int lcb = __ffsll((long long)b) - 1;
      b &= b - 1;
      const int col = lcb * BS + j;
      if (col < 64) mask |= 1ull << col;
    }
  }
  return mask;
}

// The cycle-reduction iteration on the compact form (register blocks A1, A1_hat, R = [A0c | A2c]; W = [A1 | R] in LDS).
// ph (nullable): debug stamps [0] GJ panels, [1] GJ trailing updates, [2] row gather + staging, [3] products,
// [4] scatter/update/norms.
template <int BS, typename IT>
__device__ __forceinline__ void crc_iterate(double (&A1)[BS][BS], double (&Ah)[BS][BS], double (&Rb)[BS][BS], double* W,
                                            double* Lbuf, double* Ybuf, int* prow, const IT* cmap, IT* rsrc,
                                            const IT* posS, const IT* posL, int n, int s, int l, int max_iter,
                                            double tol, int scan_mode, int lane_in, long long* ph, int& it, bool& converged,
                                            bool& saw_nan) {
  constexpr int NP = CrcSmem<BS>::NP, LDW = CrcSmem<BS>::LDW;
  double* G1 = W + NP;
  const int wr = s + l;
  converged = false;
  saw_nan = false;
  it = 0;
  for (; it < max_iter;) {
    // The lane index is re-derived from an opaque copy in every iteration: everything computed from it (LDS addresses of the
    // register blocks, of the scatter, ...) is loop-invariant, gets hoisted in front of the loop by the compiler and -- with
    // 256 registers taken by the blocks -- spilled there and re-loaded from scratch inside (60 dwords per iteration in
    // cr_fused_kernel_occ2<5, 4>); a few integer instructions per iteration are cheaper than that.
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int lr = lane >> 3, lc = lane & 7;
    // W = [A1 | R] -> [. | A1^-1 R] (rows in pivot order)
    blk_store_lds<BS>(A1, W, LDW, lr, lc);
    if (scan_mode && lr == lc) {  // stabilize(A1): 1e-16 on the diagonal of the solve only (shared.py:6-9)
#pragma unroll
      for (int i = 0; i < BS; ++i) W[(lr * BS + i) * LDW + lc * BS + i] = A1[i][i] + 1e-16;
    }
    blk_store_lds<BS>(Rb, G1, LDW, lr, lc);
    double inv_lo = 1e300, inv_hi = 0.0;
    gauss_jordan_blocked<BS>(W, LDW, n, 2, Lbuf, Ybuf, prow, lane, ph, inv_lo, inv_hi);  // syncs on entry and exit
    // One step of iterative refinement, X += A1^-1 (R - A1 X), in every iteration whose pivots span more than
    // CR_REFINE_PIVOT_RATIO: the blocked Gauss-Jordan loses ~1e-15 x cond(A1) where the reference's LAPACK LU
    // (cycle_reduction.py:150-160) keeps 1e-10, and the worst A1 of a draw may come as late as the third or fourth iteration
    // (round 2 only looked at the first two: fuzz seed 23 found draws with a pivot ratio of 2.5e6 / 9e5 in iteration 2, 4..7e-9
    // off in T; with the step they are at 6e-12 / 1e-10, tools/cr_refine_model.py).  One or two draws in a hundred take the
    // branch in some iteration; the others are untouched (bit-identical).  The residual is formed in twice the working
    // precision (mm_residual_dot2): the corrected X is then better than the reference's own LU on the systems where float64
    // algorithms cannot agree to 1e-9 (cond(A1) ~ 1e6: tools/cr_accuracy_study.py, tests/golden/cr_ill_conditioned_54.npz).
    if (__builtin_amdgcn_readfirstlane((int)(inv_hi > cr_refine_ratio<BS>() * inv_lo))) {
      gj_unpermute<BS>(W, LDW, n, 1, 2, prow, lane);  // X in natural row order (syncs inside)
      double xh[BS][BS], rr[BS][BS];
      blk_load_lds<BS>(xh, G1, LDW, lr, lc);
      blk_store_lds<BS>(A1, W, LDW, lr, lc);
      if (scan_mode && lr == lc) {
#pragma unroll
        for (int i = 0; i < BS; ++i) W[(lr * BS + i) * LDW + lc * BS + i] = A1[i][i] + 1e-16;
      }
      wave_sync();
      mm_residual_dot2<BS>(rr, Rb, W, LDW, G1, LDW, n, lr, lc);  // R - A1 X, inner products in twice the working precision
      wave_sync();
      blk_store_lds<BS>(rr, G1, LDW, lr, lc);
      gauss_jordan_blocked<BS>(W, LDW, n, 2, Lbuf, Ybuf, prow, lane);  // [A1 | R - A1 X] -> the correction (syncs on entry and exit)
      gj_unpermute<BS>(W, LDW, n, 1, 2, prow, lane);
      blk_load_lds<BS>(rr, G1, LDW, lr, lc);
      wave_sync();
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) xh[i][j] += rr[i][j];
      blk_store_lds<BS>(xh, G1, LDW, lr, lc);
      if (lane < NP) prow[lane] = lane;  // the solution sits in natural row order now
      wave_sync();
    }
    long long tk0 = ph ? clock64() : 0;
    // gather the rows S then L of the solution into compact order: XC[r] = X[cmap[r]]
    if (lane < NP) rsrc[lane] = (IT)((lane < wr) ? prow[cmap[lane]] : 0);
    wave_sync();
    {
      double t[BS][BS];
#pragma unroll
      for (int i = 0; i < BS; ++i) {
        const int src = rsrc[lr * BS + i];
#pragma unroll
        for (int j = 0; j < BS; ++j) t[i][j] = G1[src * LDW + lc * BS + j];
      }
      wave_sync();
      blk_store_lds<BS>(t, G1, LDW, lr, lc);
    }
    blk_store_lds<BS>(Rb, W, LDW, lr, lc);  // left operands [A0c | A2c] -> dead column group 0
    wave_sync();
    if (ph) {
      const long long tk1 = clock64();
      ph[2] += tk1 - tk0;
      tk0 = tk1;
    }
    double acc1[BS][BS], acc2[BS][BS];
    blk_zero<BS>(acc1);
    blk_zero<BS>(acc2);
    mm_acc<BS, false>(acc1, W, LDW, G1, LDW, s, lr, lc);                    // [m00 | m02] = A0c X[S,:]
    mm_acc<BS, false>(acc2, W + s, LDW, G1 + s * LDW, LDW, l, lr, lc);      // [m20 | m22] = A2c X[L,:]
    wave_sync();
    if (ph) {
      const long long tk1 = clock64();
      ph[3] += tk1 - tk0;
      tk0 = tk1;
    }
    blk_store_lds<BS>(acc1, W, LDW, lr, lc);
    blk_store_lds<BS>(acc2, G1, LDW, lr, lc);
    wave_sync();
    double t0[BS][BS], t2[BS][BS];
    // compact positions of my variable columns: read from the LDS tables where they are used (kept in registers across the
    // iteration they were spilled: 20 scratch re-loads per iteration in cr_fused_kernel_occ2<5, 4>)
    int vS[BS], vL[BS];
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      vS[j] = posS[lc * BS + j];
      vL[j] = posL[lc * BS + j];
    }
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        const int row = lr * BS + i, c = lc * BS + j;
        const double d02 = (vL[j] >= 0) ? W[row * LDW + vL[j]] : 0.0;   // m02[:, posL(v)]
        const double d20 = (vS[j] >= 0) ? G1[row * LDW + vS[j]] : 0.0;  // m20[:, posS(v)]
        A1[i][j] -= d02;
        A1[i][j] -= d20;
        Ah[i][j] -= d20;
        t0[i][j] = (c < s) ? acc1[i][j] : 0.0;
        t2[i][j] = (c >= s && c < wr) ? acc2[i][j] : 0.0;
        Rb[i][j] = -(t0[i][j] + t2[i][j]);
      }
    ++it;
    const double nrm0 = blk_norm1<BS>(t0);
    if (ph) ph[4] += clock64() - tk0;
    if (nrm0 < tol) {
      // (the A2 norm is only looked at once the A0 norm has passed: 15 cross-lane exchanges saved per iteration before)
      if (scan_mode || blk_norm1<BS>(t2) < tol) {  // the scan variant tests the A0 norm only (cycle_reduction.py:269-277)
        converged = true;
        break;
      }
    } else if (nrm0 != nrm0) {
      saw_nan = true;
      break;
    }
    wave_sync();
  }
}

template <int BS>
__device__ __forceinline__ void cr_compact_body(const double* __restrict__ A, const double* __restrict__ B,
                                                const double* __restrict__ C, int batch, int n, int max_iter,
                                                double tol, double* __restrict__ T_out,
                                                int32_t* __restrict__ status,
                                                int32_t* __restrict__ n_iter_out,
                                                long long* __restrict__ dbg, int scan_mode,
                                                const double* __restrict__ D, int k,
                                                double* __restrict__ R_out) {
  constexpr int NP = CrcSmem<BS>::NP, LDW = CrcSmem<BS>::LDW;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* W = smem;
  double* G1 = W + NP;
  double* Lbuf = W + NP * LDW;
  double* Ybuf = Lbuf + NP * BS;
  int* prow = (int*)(Ybuf + BS * 2 * NP);
  int* cmap = prow + NP;   // compact column -> variable (S first, then L)
  int* posS = cmap + NP;   // variable -> compact column of R if it is a state, else -1
  int* posL = posS + NP;   // variable -> compact column of R if it is a lead variable, else -1
  int* rsrc = posL + NP;   // compact row r of X sits in row rsrc[r] of W after the elimination
  const int lane = threadIdx.x, lr = lane >> 3, lc = lane & 7;

  for (int draw = blockIdx.x; draw < batch; draw = batch) {  // (one draw per workgroup, grid = batch: see kalman_nt_kernel)
    const size_t off = (size_t)draw * n * n;
    wave_sync();
    for (int idx = lane; idx < NP * LDW; idx += 64) W[idx] = 0.0;
    double A1[BS][BS], Ah[BS][BS], Rb[BS][BS];
    unsigned long long maskS, maskL;
    {
      double t[BS][BS];
      blk_load_global<BS>(t, A + off, n, n, n, lr, lc);
      maskS = blk_colmask<BS>(t);
      blk_load_global<BS>(t, C + off, n, n, n, lr, lc);
      maskL = blk_colmask<BS>(t);
    }
    const int s = __popcll(maskS), l = __popcll(maskL), wr = s + l;
    if (wr > NP) {  // does not fit the compact tile: the dense kernel handles this draw
      if (lane == 0) status[draw] = DSGE_ST_INTERNAL_RERUN;
      continue;
    }
    if (lane < NP) {
      const unsigned long long below = (1ull << lane) - 1ull;
      const bool isS = (maskS >> lane) & 1ull, isL = (maskL >> lane) & 1ull;
      const int ps = __popcll(maskS & below), pl = s + __popcll(maskL & below);
      posS[lane] = isS ? ps : -1;
      posL[lane] = isL ? pl : -1;
      if (isS) cmap[ps] = lane;
      if (isL) cmap[pl] = lane;
    }
    wave_sync();
    // R = [A[:,S] | C[:,L]] (column gather), A1 = B
    int ccol[BS];   // source variable of my compact columns (-1 = padding)
    int vS[BS], vL[BS];  // compact positions of my variable columns
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      const int c = lc * BS + j;
      ccol[j] = (c < wr) ? cmap[c] : -1;
      vS[j] = posS[c];
      vL[j] = posL[c];
    }
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        const int r = lr * BS + i, c = lc * BS + j;
        double v = 0.0;
        if (r < n && ccol[j] >= 0) v = (c < s) ? A[off + (size_t)r * n + ccol[j]] : C[off + (size_t)r * n + ccol[j]];
        Rb[i][j] = v;
      }
    blk_load_global<BS>(A1, B + off, n, n, n, lr, lc);
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) Ah[i][j] = A1[i][j];

    bool converged, saw_nan;
    int it;
    // debug stamps (draw 0): [0] GJ panels, [1] GJ trailing updates, [2] row gather + staging, [3] products,
    // [4] scatter/update/norms, [5] final solve, [6] total
    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long tk_start = dbg ? clock64() : 0;
    crc_iterate<BS>(A1, Ah, Rb, W, Lbuf, Ybuf, prow, cmap, rsrc, posS, posL, n, s, l, max_iter, tol, scan_mode, lane,
                    dbg ? ph : nullptr, it, converged, saw_nan);
    const long long tk_fin = dbg ? clock64() : 0;

    double Tb[BS][BS];
    blk_zero<BS>(Tb);
    // scan variant: T is formed from whatever A1_hat the fixed trip count reached (:292), NaN excepted
    const bool solve_T = converged || (scan_mode && !saw_nan);
    if (solve_T) {
      // T[:,S] = -A1_hat^-1 A[:,S]   (cycle_reduction.py:181); every other column is exactly zero.
      // With D given the same elimination also yields the shock-impact matrix: at convergence A1_hat = B + C T
      // (to the square of the last iterate's norms), so R = -(C T + B)^-1 D (shared.py:74-75) = -A1_hat^-1 D.
      const bool want_R = (R_out != nullptr) && !scan_mode;
      const bool r_fits = want_R && (s + k <= NP);
      wave_sync();
      blk_store_lds<BS>(Ah, W, LDW, lr, lc);
      if (scan_mode && lr == lc) {
#pragma unroll
        for (int i = 0; i < BS; ++i) W[(lr * BS + i) * LDW + lc * BS + i] = Ah[i][i] + 1e-16;
      }
      {
        double t[BS][BS];
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) {
            const int r = lr * BS + i, c = lc * BS + j;
            double v = 0.0;
            if (r < n && c < s) v = A[off + (size_t)r * n + ccol[j]];
            else if (r_fits && r < n && c >= s && c < s + k) v = D[(size_t)draw * n * k + (size_t)r * k + (c - s)];
            t[i][j] = v;
          }
        blk_store_lds<BS>(t, G1, LDW, lr, lc);
      }
      gauss_jordan_blocked<BS>(W, LDW, n, 2, Lbuf, Ybuf, prow, lane);
      gj_unpermute<BS>(W, LDW, n, 1, 2, prow, lane);
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j)
          Tb[i][j] = (vS[j] >= 0) ? -G1[(lr * BS + i) * LDW + vS[j]] : 0.0;
      if (r_fits) {
        for (int idx = lane; idx < n * k; idx += 64) {
          const int r = idx / k, c = idx - r * k;
          R_out[(size_t)draw * n * k + idx] = -G1[r * LDW + s + c];
        }
      } else if (want_R) {  // s + k does not fit next to A[:,S]: one more elimination for D alone
        wave_sync();
        blk_store_lds<BS>(Ah, W, LDW, lr, lc);
        {
          double t[BS][BS];
          blk_load_global<BS>(t, D + (size_t)draw * n * k, n, k, k, lr, lc);
          blk_store_lds<BS>(t, G1, LDW, lr, lc);
        }
        gauss_jordan_blocked<BS>(W, LDW, n, 2, Lbuf, Ybuf, prow, lane);
        gj_unpermute<BS>(W, LDW, n, 1, 2, prow, lane);
        for (int idx = lane; idx < n * k; idx += 64) {
          const int r = idx / k, c = idx - r * k;
          R_out[(size_t)draw * n * k + idx] = -G1[r * LDW + c];
        }
      }
    }
    if (R_out != nullptr && !scan_mode && !solve_T)
      for (int idx = lane; idx < n * k; idx += 64) R_out[(size_t)draw * n * k + idx] = 0.0;
    blk_store_global<BS>(Tb, T_out + off, n, n, n, lr, lc);
    if (lane == 0) {
      status[draw] = solve_T ? DSGE_ST_OK : (DSGE_ST_NOT_CONVERGED | (saw_nan ? DSGE_ST_NAN : 0));
      if (n_iter_out) n_iter_out[draw] = it;
      if (dbg && draw == 0) {
        const long long tk_end = clock64();
        ph[5] = tk_end - tk_fin;
        ph[6] = tk_end - tk_start;
        ph[7] = it;
        for (int q = 0; q < 8; ++q) dbg[q] = ph[q];
      }
    }
  }
}

// (BS <= 3: the body sits at the 256-register boundary of two waves per SIMD -- 258 without the bound, i.e. ONE wave and
// 1.5x the time on the 24-variable full_nk system)
template <int BS>
__global__ __launch_bounds__(64, (BS <= 3 ? 2 : 1)) void cr_compact_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                         const double* __restrict__ C, int batch, int n, int max_iter,
                                                         double tol, double* __restrict__ T_out,
                                                         int32_t* __restrict__ status,
                                                         int32_t* __restrict__ n_iter_out,
                                                         long long* __restrict__ dbg, int scan_mode,
                                                         const double* __restrict__ D, int k,
                                                         double* __restrict__ R_out) {
  cr_compact_body<BS>(A, B, C, batch, n, max_iter, tol, T_out, status, n_iter_out, dbg, scan_mode, D, k, R_out);
}

// The same body with the register budget of two waves per SIMD.  On the 4 x 4 tile (369 registers at one wave per SIMD)
// the 528 B of scratch this costs are cheaper than the idle issue slots of the serial panel factorisation: 0.84 -> 0.75 ms
// per 4096 deflated SW-shaped draws.  (The 5 x 5 tile, 508 registers, loses 40 % the same way and stays at one wave.)
template <int BS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void cr_compact_kernel_occ2(
    const double* __restrict__ A, const double* __restrict__ B, const double* __restrict__ C, int batch, int n, int max_iter,
    double tol, double* __restrict__ T_out, int32_t* __restrict__ status, int32_t* __restrict__ n_iter_out,
    long long* __restrict__ dbg, int scan_mode, const double* __restrict__ D, int k, double* __restrict__ R_out) {
  cr_compact_body<BS>(A, B, C, batch, n, max_iter, tol, T_out, status, n_iter_out, dbg, scan_mode, D, k, R_out);
}

}  // namespace dsge

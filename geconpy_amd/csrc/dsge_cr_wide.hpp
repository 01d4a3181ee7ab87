// Cycle reduction for 49 .. 64 variables on FOUR wavefronts per draw ("cr_wide_kernel").
//
// One wavefront per draw stops scaling at n = 48: the 7 x 7 / 8 x 8 register blocks of the 56- and 64-wide tiles spill
// (1.9 / 3.4 KB of scratch in cr_compact_kernel<7> / <8>) and 61 KB of LDS leave two wavefronts on a CU -- 8.6 ms per 4096
// systems of 56 variables.  Here a draw is a workgroup of 256 threads: the 64 x 128 matrix W = [A1 | R] is tiled by a
// 16 x 16 thread grid with 4 x 4 register blocks (the block helpers of dsge_device.hpp take any (row, column) of the
// grid), so the trailing updates of the Gauss-Jordan elimination, the products and the scatter run on four wavefronts with
// 96 registers of persistent state each, while the panel factorisation -- a chain per column: pivot search, reciprocal,
// row broadcast -- stays on wavefront 0 (one matrix row per lane, n <= 64 rows) between two workgroup barriers.
// Same column-compact algorithm, stopping rule and outputs as cr_compact_kernel (dsge_cr_compact.hpp); the norms are
// summed in a different order (per wavefront, then across), everything else is the same arithmetic.
#pragma once
#include "dsge_cr_compact.hpp"

namespace dsge {

struct CrwSmem {
  static constexpr int NP = 64, BS = 4, LDW = 2 * NP + 2;
  // W (NP x LDW), Lbuf (NP x BS), Ybuf (BS x 2 NP), column-sum partials (4 x NP), reduction slots (8)
  static constexpr size_t dbl = (size_t)NP * LDW + NP * BS + BS * 2 * NP + 4 * NP + 8;
  // ints: prow, cmap, posS, posL, rsrc (NP each), misc (8)
  static constexpr size_t bytes = sizeof(double) * dbl + sizeof(int) * (5 * NP + 8);
};

// Blocked Gauss-Jordan with partial pivoting on W (n rows, two column groups of 64), 256 threads: see
// gauss_jordan_blocked<4> for the panel (run by wavefront 0 here) and the trailing update (all four).
// piv (nullable, LDS): piv[0], piv[1] receive the smallest and largest |1 / pivot| of the elimination (the free condition
// estimate of gauss_jordan_blocked, same arithmetic), valid for every thread after the call.
__device__ __forceinline__ void gauss_jordan_wide(double* W, int n, double* Lbuf, double* Ybuf, int* prow, int tid,
                                                  double* piv = nullptr) {
  constexpr int NP = CrwSmem::NP, BS = CrwSmem::BS, LDW = CrwSmem::LDW, wcols = 2 * NP;
  const int lane = tid & 63, wv = tid >> 6, tr = tid >> 4, tc = tid & 15;
  unsigned long long used = 0ull;  // (wavefront 0 only)
  double inv_lo = 1e300, inv_hi = 0.0;  // (wavefront 0 only, wave-uniform)
  const int nsteps = (n + BS - 1) / BS;
  for (int kb = 0; kb < nsteps; ++kb) {
    const int j0 = kb * BS;
    const int bw = (n - j0 < BS) ? (n - j0) : BS;
    __syncthreads();
    if (wv == 0) {
      // ---- panel: one matrix row per lane, augmented with the identity slots ------------
      __builtin_amdgcn_s_setprio(3);  // (the chain everybody waits for: ahead of the other workgroup's wave on this SIMD)
      double pw[BS], id[BS];
#pragma unroll
      for (int c = 0; c < BS; ++c) {
        pw[c] = (lane < n && c < bw) ? W[lane * LDW + j0 + c] : 0.0;
        id[c] = 0.0;
      }
      int rsel[BS];
      double inv_own = 1.0;
#pragma unroll
      for (int c = 0; c < BS; ++c) {
        rsel[c] = 0;
        if (c < bw) {
          const bool cand = (lane < n) && !((used >> lane) & 1ull);
          unsigned key = 0u;
          if (cand) key = (((unsigned)__double2hiint(pw[c]) & 0x7fffffffu) & ~63u) | (unsigned)(63 - lane);
          key = wave_max_u32(key);
          const int r = 63 - (int)(key & 63u);
          rsel[c] = r;
          used |= 1ull << r;
          const bool is_r = (lane == r);
          if (is_r) id[c] = 1.0;
          const double inv = fast_rcp(readlane_dyn_f64(pw[c], r));
          inv_lo = fmin(inv_lo, fabs(inv));
          inv_hi = fmax(inv_hi, fabs(inv));
          const double f = is_r ? 0.0 : pw[c];
          inv_own = is_r ? inv : inv_own;
#pragma unroll
          for (int c2 = 0; c2 < BS; ++c2) {
            if (c2 > c) pw[c2] = fma(-f, readlane_dyn_f64(pw[c2], r) * inv, pw[c2]);
            if (c2 <= c) id[c2] = fma(-f, readlane_dyn_f64(id[c2], r) * inv, id[c2]);
          }
        }
      }
#pragma unroll
      for (int c = 0; c < BS; ++c) id[c] *= inv_own;
      if (piv && kb == nsteps - 1 && lane == 0) {
        piv[0] = inv_lo;
        piv[1] = inv_hi;
      }
      {
        double lh[BS];
#pragma unroll
        for (int b = 0; b < BS; ++b) lh[b] = -id[b];
#pragma unroll
        for (int a = 0; a < BS; ++a)
          if (a < bw && lane == rsel[a]) lh[a] += 1.0;
#pragma unroll
        for (int b = 0; b < BS; ++b) Lbuf[lane * BS + b] = (lane < n) ? lh[b] : 0.0;
      }
      for (int c = lane; c < wcols; c += 64) {
#pragma unroll
        for (int b = 0; b < BS; ++b) Ybuf[b * wcols + c] = (b < bw) ? W[rsel[b] * LDW + c] : 0.0;
      }
#pragma unroll
      for (int a = 0; a < BS; ++a)
        if (a < bw && lane == 0) prow[j0 + a] = rsel[a];
      __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
    // ---- trailing update on register blocks: W[i,:] -= Lhat[i,:] Wpiv (16 x 16 thread grid) ----------
    double lh[BS][BS];
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int b = 0; b < BS; ++b) lh[i][b] = Lbuf[(tr * BS + i) * BS + b];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (g == 0 && tc <= kb) continue;  // block columns at or left of the panel inside the matrix part are dead
      const int c0 = g * NP + tc * BS;
      double wb[BS][BS], yb[BS][BS];
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) wb[i][j] = W[(tr * BS + i) * LDW + c0 + j];
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
        for (int j = 0; j < BS; ++j) W[(tr * BS + i) * LDW + c0 + j] = wb[i][j];
    }
  }
  __syncthreads();
}

// rows of the right-hand-side group back into natural order (row j of the solution sits in row prow[j] of W)
__device__ __forceinline__ void gj_unpermute_wide(double* W, int n, const int* prow, int tid) {
  constexpr int NP = CrwSmem::NP, BS = CrwSmem::BS, LDW = CrwSmem::LDW;
  const int tr = tid >> 4, tc = tid & 15;
  const int c0 = NP + tc * BS;
  double t[BS][BS];
#pragma unroll
  for (int i = 0; i < BS; ++i) {
    const int r = tr * BS + i;
    const int src = (r < n) ? prow[r] : r;
#pragma unroll
    for (int j = 0; j < BS; ++j) t[i][j] = W[src * LDW + c0 + j];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < BS; ++i)
#pragma unroll
    for (int j = 0; j < BS; ++j) W[(tr * BS + i) * LDW + c0 + j] = t[i][j];
  __syncthreads();
}

// induced 1-norm (max absolute column sum, NaN-propagating) of a matrix held as 4 x 4 blocks on the 16 x 16 thread grid
__device__ __forceinline__ double norm1_wide(const double (&x)[4][4], double* part, double* red, int tid) {
  const int lane = tid & 63, wv = tid >> 6, tc = tid & 15;
  __syncthreads();  // (part / red may still be read by a previous call)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double cs = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) cs += fabs(x[i][j]);
    cs += shfl_xor_f64(cs, 16);  // the four block rows of this wavefront (lane = (tr & 3) * 16 + tc)
    cs += shfl_xor_f64(cs, 32);
    if (lane < 16) part[wv * 64 + tc * 4 + j] = cs;
  }
  __syncthreads();
  if (wv == 0) {
    const double tot = (part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane]);
    const double m = wave_nanmax(tot);
    if (lane == 0) red[0] = m;
  }
  __syncthreads();
  return red[0];
}

// One draw on the workgroup.  REFINE = true (what the kernel runs): when the pivots of the elimination of an iteration span
// more than CR_REFINE_PIVOT_RATIO (an ill-conditioned A1: the blocked Gauss-Jordan then loses ~1e-15 x cond
// where the reference's LAPACK LU, cycle_reduction.py:150-160, keeps 1e-10) the solve gets the one step of iterative
// refinement X += A1^-1 (R - A1 X) of crc_iterate (dsge_cr_compact.hpp; same test, same arithmetic).  REFINE = false leaves
// at that point and returns true (kept for the experiment below).  Measured at n = 56, 4096 systems without static variables
// (tools/crw_variants, profiles/r3/crw_variants.txt): no test 3.85 ms; test + leave 3.97; test + refinement inlined (this)
// 3.95; test + the refined instance behind a device function call 4.09 -- the call's ABI registers cost the common loop more
// scalar spills than the inlined block costs in code size.
template <bool REFINE>
__device__ __forceinline__ bool crw_solve(double* smem, const double* __restrict__ A, const double* __restrict__ B,
                                       const double* __restrict__ C, int draw, int n, int max_iter, double tol,
                                       double* __restrict__ T_out, int32_t* __restrict__ status,
                                       int32_t* __restrict__ n_iter_out, int scan_mode, const double* __restrict__ D, int k,
                                       double* __restrict__ R_out) {
  constexpr int NP = CrwSmem::NP, BS = CrwSmem::BS, LDW = CrwSmem::LDW;
  double* W = smem;
  double* G1 = W + NP;
  double* Lbuf = W + NP * LDW;
  double* Ybuf = Lbuf + NP * BS;
  double* part = Ybuf + BS * 2 * NP;
  double* red = part + 4 * NP;
  int* prow = (int*)(red + 8);
  int* cmap = prow + NP;
  int* posS = cmap + NP;
  int* posL = posS + NP;
  int* rsrc = posL + NP;
  int* misc = rsrc + NP;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, tr = tid >> 4, tc = tid & 15;
  const size_t off = (size_t)draw * n * n;
  for (int idx = tid; idx < NP * LDW; idx += 256) W[idx] = 0.0;
  // non-zero columns of A (states) and C (leads): lane j of wavefront 0 looks down column j, all rows in flight
  if (wv == 0) {
    int nzA = 0, nzC = 0;
    const int cl = lane < n ? lane : n - 1;
    const double* ap = A + off + cl;
    const double* cp = C + off + cl;
    for (int r0 = 0; r0 < n; r0 += 16) {
      double av[16], cv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int ri = r0 + u < n ? r0 + u : n - 1;
        av[u] = ap[(size_t)ri * n];
        cv[u] = cp[(size_t)ri * n];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        nzA |= (av[u] != 0.0) ? 1 : 0;  // (a NaN counts as non-zero)
        nzC |= (cv[u] != 0.0) ? 1 : 0;
      }
    }
    const unsigned long long mS = __ballot(nzA != 0 && lane < n), mL = __ballot(nzC != 0 && lane < n);
    if (lane == 0) {
      misc[0] = (int)(mS & 0xffffffffull);
      misc[1] = (int)(mS >> 32);
      misc[2] = (int)(mL & 0xffffffffull);
      misc[3] = (int)(mL >> 32);
    }
  }
  __syncthreads();
  const unsigned long long maskS = (unsigned long long)(unsigned)misc[0] | ((unsigned long long)(unsigned)misc[1] << 32);
  const unsigned long long maskL = (unsigned long long)(unsigned)misc[2] | ((unsigned long long)(unsigned)misc[3] << 32);
  const int s = __popcll(maskS), l = __popcll(maskL), wr = s + l;
  if (wr > NP) {  // does not fit the compact tile: the dense kernel handles this draw
    if (tid == 0) status[draw] = DSGE_ST_INTERNAL_RERUN;
    return false;
  }
  if (tid < NP) {
    const unsigned long long below = (1ull << tid) - 1ull;
    const bool isS = (maskS >> tid) & 1ull, isL = (maskL >> tid) & 1ull;
    const int ps = __popcll(maskS & below), pl = s + __popcll(maskL & below);
    posS[tid] = isS ? ps : -1;
    posL[tid] = isL ? pl : -1;
    if (isS) cmap[ps] = tid;
    if (isL) cmap[pl] = tid;
  }
  __syncthreads();
  // R = [A[:,S] | C[:,L]] (column gather), A1 = B
  int ccol[BS], vS[BS], vL[BS];
#pragma unroll
  for (int j = 0; j < BS; ++j) {
    const int c = tc * BS + j;
    ccol[j] = (c < wr) ? cmap[c] : -1;
    vS[j] = posS[c];
    vL[j] = posL[c];
  }
  double A1[BS][BS], Ah[BS][BS], Rb[BS][BS];
#pragma unroll
  for (int i = 0; i < BS; ++i)
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      const int r = tr * BS + i, c = tc * BS + j;
      const int rr = r < n ? r : n - 1, cc = ccol[j] >= 0 ? ccol[j] : 0;
      Rb[i][j] = (c < s) ? A[off + (size_t)rr * n + cc] : C[off + (size_t)rr * n + cc];
    }
  blk_load_global<BS>(A1, B + off, n, n, n, tr, tc);
#pragma unroll
  for (int i = 0; i < BS; ++i)
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      const int r = tr * BS + i;
      Rb[i][j] = (r < n && ccol[j] >= 0) ? Rb[i][j] : 0.0;
      Ah[i][j] = A1[i][j];
    }

  bool converged = false, saw_nan = false;
  int it = 0;
  for (; it < max_iter;) {
    // W = [A1 | R] -> [. | A1^-1 R] (rows in pivot order)
    blk_store_lds<BS>(A1, W, LDW, tr, tc);
    if (scan_mode && tr == tc) {  // stabilize(A1): 1e-16 on the diagonal of the solve only (shared.py:6-9)
#pragma unroll
      for (int i = 0; i < BS; ++i) W[(tr * BS + i) * LDW + tc * BS + i] = A1[i][i] + 1e-16;
    }
    blk_store_lds<BS>(Rb, G1, LDW, tr, tc);
    gauss_jordan_wide(W, n, Lbuf, Ybuf, prow, tid, red + 2);  // barriers on entry and exit
    if (red[3] > CR_REFINE_PIVOT_RATIO * red[2]) {  // (workgroup-uniform: LDS values behind a barrier)
      if constexpr (!REFINE) {
        return true;
      } else {
        gj_unpermute_wide(W, n, prow, tid);  // X in natural row order (barriers inside)
        double xh[BS][BS], rr[BS][BS];
        blk_load_lds<BS>(xh, G1, LDW, tr, tc);
        blk_store_lds<BS>(A1, W, LDW, tr, tc);
        if (scan_mode && tr == tc) {
#pragma unroll
          for (int i = 0; i < BS; ++i) W[(tr * BS + i) * LDW + tc * BS + i] = A1[i][i] + 1e-16;
        }
        __syncthreads();
        mm_residual_dot2<BS>(rr, Rb, W, LDW, G1, LDW, n, tr, tc);  // R - A1 X, inner products in twice the working precision
        __syncthreads();
        blk_store_lds<BS>(rr, G1, LDW, tr, tc);
        gauss_jordan_wide(W, n, Lbuf, Ybuf, prow, tid);  // [A1 | R - A1 X] -> the correction
        gj_unpermute_wide(W, n, prow, tid);
        blk_load_lds<BS>(rr, G1, LDW, tr, tc);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) xh[i][j] += rr[i][j];
        blk_store_lds<BS>(xh, G1, LDW, tr, tc);
        if (tid < NP) prow[tid] = tid;  // the solution sits in natural row order now
        __syncthreads();
      }
    }
    // gather the rows S then L of the solution into compact order: XC[r] = X[cmap[r]]
    if (tid < NP) rsrc[tid] = (tid < wr) ? prow[cmap[tid]] : 0;
    __syncthreads();
    {
      double t[BS][BS];
#pragma unroll
      for (int i = 0; i < BS; ++i) {
        const int src = rsrc[tr * BS + i];
#pragma unroll
        for (int j = 0; j < BS; ++j) t[i][j] = G1[src * LDW + tc * BS + j];
      }
      __syncthreads();
      blk_store_lds<BS>(t, G1, LDW, tr, tc);
    }
    blk_store_lds<BS>(Rb, W, LDW, tr, tc);  // left operands [A0c | A2c] -> dead column group 0
    __syncthreads();
    double acc1[BS][BS], acc2[BS][BS];
    blk_zero<BS>(acc1);
    blk_zero<BS>(acc2);
    mm_acc<BS, false>(acc1, W, LDW, G1, LDW, s, tr, tc);                // [m00 | m02] = A0c X[S,:]
    mm_acc<BS, false>(acc2, W + s, LDW, G1 + s * LDW, LDW, l, tr, tc);  // [m20 | m22] = A2c X[L,:]
    __syncthreads();
    blk_store_lds<BS>(acc1, W, LDW, tr, tc);
    blk_store_lds<BS>(acc2, G1, LDW, tr, tc);
    __syncthreads();
    double t0[BS][BS], t2[BS][BS];
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        const int row = tr * BS + i, c = tc * BS + j;
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
    const double nrm0 = norm1_wide(t0, part, red, tid);
    if (nrm0 < tol) {
      if (scan_mode || norm1_wide(t2, part, red, tid) < tol) {  // the scan variant tests the A0 norm only
        converged = true;
        break;
      }
    } else if (nrm0 != nrm0) {
      saw_nan = true;
      break;
    }
    __syncthreads();
  }

  double Tb[BS][BS];
  blk_zero<BS>(Tb);
  const bool solve_T = converged || (scan_mode && !saw_nan);
  const bool want_R = (R_out != nullptr) && !scan_mode;
  if (solve_T) {
    // T[:,S] = -A1_hat^-1 A[:,S] (cycle_reduction.py:181); with D the same elimination yields R = -A1_hat^-1 D
    const bool r_fits = want_R && (s + k <= NP);
    __syncthreads();
    blk_store_lds<BS>(Ah, W, LDW, tr, tc);
    if (scan_mode && tr == tc) {
#pragma unroll
      for (int i = 0; i < BS; ++i) W[(tr * BS + i) * LDW + tc * BS + i] = Ah[i][i] + 1e-16;
    }
    {
      double t[BS][BS];
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) {
          const int r = tr * BS + i, c = tc * BS + j;
          double v = 0.0;
          if (r < n && c < s) v = A[off + (size_t)r * n + ccol[j]];
          else if (r_fits && r < n && c >= s && c < s + k) v = D[(size_t)draw * n * k + (size_t)r * k + (c - s)];
          t[i][j] = v;
        }
      blk_store_lds<BS>(t, G1, LDW, tr, tc);
    }
    gauss_jordan_wide(W, n, Lbuf, Ybuf, prow, tid);
    gj_unpermute_wide(W, n, prow, tid);
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) Tb[i][j] = (vS[j] >= 0) ? -G1[(tr * BS + i) * LDW + vS[j]] : 0.0;
    if (r_fits) {
      for (int idx = tid; idx < n * k; idx += 256) {
        const int r = idx / k, c = idx - r * k;
        R_out[(size_t)draw * n * k + idx] = -G1[r * LDW + s + c];
      }
    } else if (want_R) {  // s + k does not fit next to A[:,S]: one more elimination for D alone
      __syncthreads();
      blk_store_lds<BS>(Ah, W, LDW, tr, tc);
      {
        double t[BS][BS];
        blk_load_global<BS>(t, D + (size_t)draw * n * k, n, k, k, tr, tc);
        blk_store_lds<BS>(t, G1, LDW, tr, tc);
      }
      gauss_jordan_wide(W, n, Lbuf, Ybuf, prow, tid);
      gj_unpermute_wide(W, n, prow, tid);
      for (int idx = tid; idx < n * k; idx += 256) {
        const int r = idx / k, c = idx - r * k;
        R_out[(size_t)draw * n * k + idx] = -G1[r * LDW + c];
      }
    }
  }
  if (want_R && !solve_T)
    for (int idx = tid; idx < n * k; idx += 256) R_out[(size_t)draw * n * k + idx] = 0.0;
  blk_store_global<BS>(Tb, T_out + off, n, n, n, tr, tc);
  if (tid == 0) {
    status[draw] = solve_T ? DSGE_ST_OK : (DSGE_ST_NOT_CONVERGED | (saw_nan ? DSGE_ST_NAN : 0));
    if (n_iter_out) n_iter_out[draw] = it;
  }
  return false;
}

__global__ __launch_bounds__(256, 2) void cr_wide_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                       const double* __restrict__ C, int batch, int n, int max_iter,
                                                       double tol, double* __restrict__ T_out,
                                                       int32_t* __restrict__ status, int32_t* __restrict__ n_iter_out,
                                                       int scan_mode, const double* __restrict__ D, int k,
                                                       double* __restrict__ R_out) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int draw = blockIdx.x;  // one draw per workgroup
  if (draw >= batch) return;
  crw_solve<true>(smem, A, B, C, draw, n, max_iter, tol, T_out, status, n_iter_out, scan_mode, D, k, R_out);
}

}  // namespace dsge

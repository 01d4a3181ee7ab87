// gensys by spectral division (round 5, dsge_options.gensys_doubling = 1: the default; 0 = the ordered QZ for every draw).
//
// What gensys needs of the pencil is its SPLIT at the unit circle: the stable deflating subspace gives T = G1[:n,:n], the counts
// give eu (gEconpy/solvers/gensys.py:237-250, 282-310).  The ordered QZ computes far more -- every eigenvalue, triangular factors --
// at ~13.6 MFLOP per draw on one dependent chain; the split alone is what the doubling iteration (cycle reduction) converges to
// quadratically.  For A + B T + C T^2 = 0 with a solvent T,
//     A + B lambda + C lambda^2 = (C lambda + M)(lambda - T),   M = B + C T,
// so the roots of the pencil are eig(T) and lambda = -1 / mu, mu in eig(G), G = M^-1 C (mu = 0: an infinite root).  The columns of
// G outside the lead set L (non-zero columns of C) vanish, so eig(G) = eig(G[L,L]) u {0}; likewise eig(T) = eig(T[S,S]) u {0} with
// S the state columns.  gensys's verdict eu = [1, 1, 0] -- as many unstable roots as lead columns, solvent exists -- holds iff
//     rho(T[S,S]) < 1   and   rho(G[L,L]) < 1 .
// This kernel CERTIFIES both for a draw whose cycle reduction converged: repeated squares P, P^2, P^4, ... until a Frobenius norm
// drops below 1/2 (then rho < 2^(-1/2^k) < 1; at most 12 squarings: rho < 0.99983).  A certified draw keeps the T of the doubling
// iteration (it agrees with the QZ's to 1e-12: the reference's own cross-solver test, tests/model/test_perturbation.py:205-206, asks
// 1e-8) and gets eu = [1, 1, 0].  EVERYTHING ELSE -- no convergence, no certificate within 12 squarings (a root within 2e-4 of the
// unit circle), more lead / state columns than the hints, a column of C with 0 < sum|C_ij| <= tol (gensys drops it from the pencil,
// gensys.py:587), a singular M, a solvent with entries beyond 1e6 -- is flagged for the ordered QZ (the window launches on the compacted list of flagged draws,
// gensys_compact_kernel), which decides as it always did: the non-regular verdicts ([1,0,k], [0,1,0], [-2,-2,0], ...) are the QZ's alone.
#pragma once
#include "dsge_device.hpp"

#include "../../include/dsge_hip.h"

namespace dsge {

constexpr int GD_MAX_SQUARINGS = 12;

// LDS doubles: W NP x LDW ([M | C_L], the column groups of the blocked Gauss-Jordan), its scratch (Lbuf NP BS, Ybuf BS 2 NP),
// Gm 2 x lcap x (lcap | 1), Ts max(2 scap (scap | 1), lcap n), index lists (prow NP, lead 64, state 64 ints), the scale guards'
// two vectors (128)
template <int BS>
__host__ __device__ inline size_t gd_lds_doubles(int n, int lcap, int scap) {
  constexpr int NP = 8 * BS, LDW = 2 * NP + 1;
  const size_t ts = 2 * (size_t)scap * (scap | 1), tl = (size_t)lcap * n;
  return (size_t)NP * LDW + (size_t)NP * BS + (size_t)BS * 2 * NP + 2 * (size_t)lcap * (lcap | 1) + (ts > tl ? ts : tl) + NP / 2 + 1 + 64 + 128;
}

// P <- P^2 (d x d, row stride ld) from src into dst; returns the squared Frobenius norm of the result (wave-uniform)
__device__ __forceinline__ double gd_square(const double* src, double* dst, int d, int ld, int lane) {
  double fro = 0.0;
  for (int idx = lane; idx < d * d; idx += 64) {
    const int i = idx / d, j = idx - i * d;
    double s0 = 0.0, s1 = 0.0;
    int k = 0;
    for (; k + 1 < d; k += 2) {
      s0 = fma(src[i * ld + k], src[k * ld + j], s0);
      s1 = fma(src[i * ld + k + 1], src[(k + 1) * ld + j], s1);
    }
    if (k < d) s0 = fma(src[i * ld + k], src[k * ld + j], s0);
    const double v = s0 + s1;
    dst[i * ld + j] = v;
    fro = fma(v, v, fro);
  }
  return wave_sum(fro);
}

// true iff rho(P) < 1 is certified: ||P^(2^k)||_F < 1/2 for some k <= GD_MAX_SQUARINGS.  buf: 2 x d x ld doubles, P in the first half.
__device__ __forceinline__ bool gd_certify_contraction(double* buf, int d, int ld, int lane) {
  if (d == 0) return true;
  double* cur = buf;
  double* nxt = buf + (size_t)d * ld;
  double fro = 0.0;
  for (int idx = lane; idx < d * d; idx += 64) {
    const double v = cur[(idx / d) * ld + (idx % d)];
    fro = fma(v, v, fro);
  }
  fro = wave_sum(fro);
  for (int k = 0; k <= GD_MAX_SQUARINGS; ++k) {
    if (!(fro == fro) || !(fro < 1e300)) return false;  // NaN / overflow: an explosive or undefined block
    if (fro < 0.25) return true;                        // ||.||_F < 1/2
    if (k == GD_MAX_SQUARINGS) break;
    wave_sync();
    fro = gd_square(cur, nxt, d, ld, lane);
    wave_sync();
    double* t = cur;
    cur = nxt;
    nxt = t;
  }
  return false;
}

// status in: the cycle reduction's word (0 = converged); out: 0 (certified; eu = [1,1,0]) or DSGE_ST_INTERNAL_RERUN (for the QZ).
// BS = tile of n (NP = 8 BS >= n); lcap <= NP.
template <int BS>
__global__ __launch_bounds__(64) void gensys_certify_kernel(const double* __restrict__ B, const double* __restrict__ C,
                                                             const double* __restrict__ T, int batch, int n, int lcap, int scap,
                                                             double tol, int32_t* __restrict__ eu_out, int32_t* __restrict__ status,
                                                             int32_t* __restrict__ qz_mark) {
  constexpr int NP = 8 * BS, LDW = 2 * NP + 1;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int ldg = lcap | 1, lds_ = scap | 1;
  double* W = smem;                                  // NP x LDW: [M | C_L | 0] -> rows in pivot order: [. | G_L]
  double* G1 = W + NP;                               //   the right-hand-side column group
  double* Lbuf = W + (size_t)NP * LDW;               // NP x BS
  double* Ybuf = Lbuf + NP * BS;                     // BS x 2 NP
  double* Gm = Ybuf + BS * 2 * NP;                   // 2 x lcap x ldg
  double* Ts = Gm + 2 * (size_t)lcap * ldg;          // 2 x scap x lds_  (first: T[L, :], lcap x n)
  int* prow = (int*)(Ts + ((2 * (size_t)scap * lds_ > (size_t)lcap * n) ? 2 * (size_t)scap * lds_ : (size_t)lcap * n));  // NP ints
  int* lidx = prow + NP + (NP & 1);                  // 64 ints: lead columns
  int* sidx = lidx + 64;                             // 64 ints: state columns
  double* Vs = (double*)(sidx + 64);                 // 128 doubles: vectors of the scale guards
  const int lane = threadIdx.x;
  const int draw = blockIdx.x;
  if (draw >= batch) return;
  const int32_t st_in = __builtin_amdgcn_readfirstlane(status[draw]);
  bool ok = (st_in == 0);
  const size_t off = (size_t)draw * n * n;
  // lead columns (gensys.py:587: sum_i |C_ij| > tol) and state columns (non-zero columns of T)
  double csum = 0.0, tmax = 0.0;
  bool tnz = false;
  if (ok) {  // (eight rows of both matrices requested together: a loop of single loads is a round trip to memory per row)
    const int col = lane < n ? lane : n - 1;
    for (int r0 = 0; r0 < n; r0 += 8) {
      double cv[8], tv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const size_t g = off + (size_t)(r0 + u < n ? r0 + u : n - 1) * n + col;
        cv[u] = C[g];
        tv[u] = T[g];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (r0 + u < n) {
          csum += fabs(cv[u]);
          tnz = tnz || (tv[u] != 0.0);
          tmax = fmax(tmax, fabs(tv[u]));
        }
      }
    }
  }
  const unsigned long long lmask = __ballot(ok && lane < n && csum > tol);
  const unsigned long long grey = __ballot(ok && lane < n && csum > 0.0 && !(csum > tol));  // dropped by gensys, kept by the iteration
  const unsigned long long smask = __ballot(ok && lane < n && tnz);
  // (a solvent with entries beyond 1e6 means a nearly rank-deficient Q2 Pi: gensys's existence test works with a tolerance there,
  //  gensys.py:282-283 -- the QZ decides)
  const unsigned long long bad = __ballot(ok && lane < n && (!(csum == csum) || !(tmax < 1e6)));
  const int l = __popcll(lmask), s = __popcll(smask);
  ok = ok && grey == 0ull && bad == 0ull && l <= lcap && s <= scap;
  if (ok) {
    for (int idx = lane; idx < NP * LDW; idx += 64) W[idx] = 0.0;  // (zero padding of both column groups)
    if ((lmask >> lane) & 1ull) lidx[__popcll(lmask & ((1ull << lane) - 1ull))] = lane;
    if ((smask >> lane) & 1ull) sidx[__popcll(smask & ((1ull << lane) - 1ull))] = lane;
    wave_sync();
    // stage C[:, L] into the right-hand-side group of W and T[L, :] into the scratch
    lane_loop_batched<8>(n * l, lane, [&](int idx) { return C[off + (size_t)(idx / l) * n + lidx[idx % l]]; },
                         [&](int idx, double v) { G1[(idx / l) * LDW + idx % l] = v; });
    double tl2 = 0.0;  // ||T[L, :]||_F^2
    lane_loop_batched<8>(l * n, lane, [&](int idx) { return T[off + (size_t)lidx[idx / n] * n + idx % n]; },
                         [&](int idx, double v) {
                           Ts[idx] = v;
                           tl2 = fma(v, v, tl2);
                         });
    tl2 = wave_sum(tl2);
    wave_sync();
    // M = B + C[:, L] T[L, :]
    lane_loop_batched<8>(n * n, lane, [&](int idx) { return B[off + idx]; },
                         [&](int idx, double bv) {
                           const int i = idx / n, j = idx - i * n;
                           double s0 = bv, s1 = 0.0;
                           int r = 0;
                           for (; r + 1 < l; r += 2) {
                             s0 = fma(G1[i * LDW + r], Ts[r * n + j], s0);
                             s1 = fma(G1[i * LDW + r + 1], Ts[(r + 1) * n + j], s1);
                           }
                           if (r < l) s0 = fma(G1[i * LDW + r], Ts[r * n + j], s0);
                           W[i * LDW + j] = s0 + s1;
                         });
    wave_sync();
    // [M | I] -> [. | M^-1]: the blocked elimination of the cycle-reduction kernels (rows stay in pivot order: row j of the
    // inverse sits in row prow[j]; syncs on entry and exit).  The trailing update works on whole column groups, so the identity
    // costs what the l columns of C_L cost -- and the full inverse is what the scale guards below need.
    for (int idx = lane; idx < n * l; idx += 64) G1[(idx / l) * LDW + idx % l] = 0.0;
    wave_sync();
    if (lane < n) G1[lane * LDW + lane] = 1.0;
    double piv_lo = 1e300, piv_hi = 0.0;
    gauss_jordan_blocked<BS>(W, LDW, n, 2, Lbuf, Ybuf, prow, lane, nullptr, piv_lo, piv_hi);
    // C[:, L] again, into the (dead) matrix group
    lane_loop_batched<8>(n * l, lane, [&](int idx) { return C[off + (size_t)(idx / l) * n + lidx[idx % l]]; },
                         [&](int idx, double v) { W[(idx / l) * LDW + idx % l] = v; });
    wave_sync();
    // N_L = (M^-1)[L, :] (row a = row prow[lidx[a]] of the inverse);  G[L, L] = N_L C[:, L];  ||N_L||_F^2, ||M^-1||_F^2
    double nl2 = 0.0, mi2 = 0.0;
    for (int idx = lane; idx < l * l; idx += 64) {
      const int a = idx / l, b = idx - a * l;
      const double* nr = G1 + prow[lidx[a]] * LDW;
      double s0 = 0.0, s1 = 0.0;
      int i = 0;
      for (; i + 1 < n; i += 2) {
        s0 = fma(nr[i], W[i * LDW + b], s0);
        s1 = fma(nr[i + 1], W[(i + 1) * LDW + b], s1);
      }
      if (i < n) s0 = fma(nr[i], W[i * LDW + b], s0);
      Gm[a * ldg + b] = s0 + s1;
    }
    for (int idx = lane; idx < n * n; idx += 64) {
      const double v = G1[(idx / n) * LDW + idx % n];
      mi2 = fma(v, v, mi2);
    }
    for (int idx = lane; idx < l * n; idx += 64) {
      const double v = G1[prow[lidx[idx / n]] * LDW + idx % n];
      nl2 = fma(v, v, nl2);
    }
    mi2 = wave_sum(mi2);
    nl2 = wave_sum(nl2);
    wave_sync();
    // ---- the scale guards (round 6).  gensys's verdict is not scale-free: a diagonal pair of the QZ with |alpha|, |beta| < tol
    // is "coincident zeros" (eu = [-2,-2,0], gensys.py:243-265) and existence is rank(Q2 Pi) by singular values > tol (:276-283).
    // Both are decided here from M^-1, with the QZ as the judge of everything not PROVEN regular:
    //  (E) the left unstable deflating subspace of the pencil is the row space of X = [N_L, I] (X G0 = -G_LL [-T_L, I],
    //      X G1 = [-T_L, I]), so Q2 = W X with W X X' W' = I and Q2 Pi = W: the singular values of Q2 Pi are EXACTLY
    //      1 / sqrt(1 + sigma_i(N_L)^2) whatever basis the QZ chose (tests/test_device_models.py checks the identity against
    //      LAPACK).  Existence <=> sigma_max(N_L) < sqrt(1/tol^2 - 1); ||N_L||_F bounds sigma_max from above.  The same number
    //      bounds the unstable block's diagonal of G1 from below (B22 = W V^-1, |diag| >= sigma_min of a triangular matrix, V^-1
    //      expands), so no pair of that block is a coincident zero.
    //  (Z) the stable block's G0 diagonal: A11 = Q1 G0 Z1 with A11^-1 = -(I + T_L'T_L)^1/2 M^-1 (I + N_L'N_L)^-1/2, hence
    //      |alpha_i| >= 1 / (sqrt(1 + ||T_L||_F^2) ||M^-1 (I + N_L'N_L)^-1/2||_F).  For ANY vector v, with w = N_L'N_L v and
    //      tau = |N_L v|^2:  N_L'N_L >= w w' / tau, so (I + N_L'N_L)^-1 <= I - w w' / (tau + |w|^2) and
    //          ||M^-1 (I + N_L'N_L)^-1/2||_F^2 <= ||M^-1||_F^2 - |M^-1 w|^2 / (tau + |w|^2):
    //      the subtraction removes the direction in which the lead rows of M^-1 blow up (where the factor (I + N'N)^-1/2 damps
    //      M^-1; a nearly singular B + C T that the lead rows see -- the bench's draw 752 -- passes), v from two steps of the power
    //      iteration; 1e-10 ||M^-1||_F^2 is added back as the rounding allowance of the difference.
    // A draw passes with a margin of 1.25 on both; everything else is the QZ's to decide.
    {
      double* vb = Vs;        // 64 doubles: the vector being multiplied
      double* tb = Vs + 64;   // 64 doubles: N_L v
      // start: the row of N_L of largest norm
      double rown = 0.0;
      if (lane < l) {
        const double* nr = G1 + prow[lidx[lane]] * LDW;
        for (int i = 0; i < n; ++i) rown = fma(nr[i], nr[i], rown);
      }
      int abest = 0;
      {
        double best = rown;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
          const double ob = __shfl_xor(best, m, 64);
          best = fmax(best, ob);
        }
        const unsigned long long hit = __ballot(lane < l && rown == best);
        abest = hit ? (int)__builtin_ctzll(hit) : 0;
      }
      double vj = (lane < n && l > 0) ? G1[prow[lidx[abest]] * LDW + lane] : 0.0;  // element `lane` of v
      double tau = 0.0, w2 = 0.0;
      for (int it = 0; it < 3; ++it) {
        const double nv = wave_sum(vj * vj);
        vj = nv > 0.0 ? vj * (1.0 / sqrt(nv)) : 0.0;
        wave_sync();
        if (lane < n) vb[lane] = vj;
        wave_sync();
        double ta = 0.0;
        if (lane < l) {
          const double* nr = G1 + prow[lidx[lane]] * LDW;
          for (int i = 0; i < n; ++i) ta = fma(nr[i], vb[i], ta);
          tb[lane] = ta;
        }
        tau = wave_sum(ta * ta);
        wave_sync();
        double wj = 0.0;
        if (lane < n)
          for (int a = 0; a < l; ++a) wj = fma(G1[prow[lidx[a]] * LDW + lane], tb[a], wj);
        w2 = wave_sum(wj * wj);
        vj = wj;  // (un-normalised w on exit: the bound uses w itself)
      }
      wave_sync();
      if (lane < n) vb[lane] = vj;
      wave_sync();
      double zj = 0.0;
      if (lane < n) {
        const double* mr = G1 + lane * LDW;
        for (int i = 0; i < n; ++i) zj = fma(mr[i], vb[i], zj);
      }
      const double z2 = wave_sum(zj * zj);
      const double rs = tol > 0.0 ? tol : 2.220446049250313e-16;
      const double m2 = 1.5625 * rs * rs;
      const double den = tau + w2;
      const double cut = den > 0.0 ? z2 / den : 0.0;
      const double mw2 = fmax(mi2 - cut, 0.0) + 1e-10 * mi2;
      const bool pass_e = (1.0 + nl2) * m2 < 1.0;
      const bool pass_z = (1.0 + tl2) * mw2 * m2 < 1.0;
      ok = pass_e && pass_z;  // (NaN in any of them: false)
    }
    wave_sync();
    if (ok) {
      lane_loop_batched<8>(s * s, lane, [&](int idx) { return T[off + (size_t)sidx[idx / s] * n + sidx[idx % s]]; },
                           [&](int idx, double v) { Ts[(idx / s) * lds_ + idx % s] = v; });
      wave_sync();
      ok = gd_certify_contraction(Gm, l, ldg, lane);
      if (ok) ok = gd_certify_contraction(Ts, s, lds_, lane);
    }
  }
  if (lane == 0) {
    if (ok) {
      status[draw] = 0;
      eu_out[3 * draw] = 1;
      eu_out[3 * draw + 1] = 1;
      eu_out[3 * draw + 2] = 0;
    } else {
      status[draw] = DSGE_ST_INTERNAL_RERUN;
    }
    if (qz_mark) qz_mark[draw] = ok ? 0 : 1;
  }
}

// {count, the marked draws in ascending order} for the window launches' `act` (one workgroup)
__global__ __launch_bounds__(1024) void gensys_compact_kernel(const int32_t* __restrict__ mark, int batch, int32_t* __restrict__ act) {
  __shared__ int wsum[16];
  __shared__ int base;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < batch; c0 += 1024) {
    const int i = c0 + tid;
    const bool flag = i < batch && mark[i] != 0;
    const unsigned long long bal = __ballot(flag);
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int pre = 0, tot = 0;
    for (int w = 0; w < 16; ++w) {
      pre += (w < wave) ? wsum[w] : 0;
      tot += wsum[w];
    }
    if (flag) act[1 + base + pre + __popcll(bal & ((1ull << lane) - 1ull))] = i;
    __syncthreads();
    if (tid == 0) base += tot;
    __syncthreads();
  }
  if (tid == 0) act[0] = base;
}

}  // namespace dsge

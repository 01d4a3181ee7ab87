// Kernels of the solve + Kalman-logp pipeline; one wavefront per draw (see dsge_device.hpp).
//   cr_kernel        : cycle reduction           A,B,C -> T, status, n_iter
//   bdirect_kernel   : backward-direct           A,B,D -> T, R
//   assemble_kernel  : R, resid, sym(RQR'), P0   (selection matrix + doubling Lyapunov)
//   kalman_kernel    : standard Kalman filter    T,RQR,P0,Z,d,H,y -> logp
#pragma once
#include "dsge_device.hpp"

#include "../../include/dsge_hip.h"

namespace dsge {

// ---------------------------------------------------------------------------------------
// Cycle reduction (Bini-Latouche-Meini), njit-variant semantics of
// gEconpy/solvers/cycle_reduction.py:127-183.
//   LDS: A0s, A2s (left operands of the four products) and the augmented system
//   W = [A1 | A0 | A2] that Gauss-Jordan turns into [. | A1^-1 A0 | A1^-1 A2].
//   A1 and A1_hat never serve as product operands, so they live in the lanes' register blocks.
// ---------------------------------------------------------------------------------------
template <int BS>
struct CrSmem {
  static constexpr int NP = Tile<BS>::NP, LD = Tile<BS>::LD, LDW = 3 * NP + 1;
  // W, then the blocked Gauss-Jordan scratch: Lbuf NP*BS, Ybuf BS*3NP, prow NP ints.  A0, A1, A2 and
  // A1_hat live in register blocks; the left operands of the four products are staged one after the
  // other in the dead first column group of W, which keeps the footprint at 45 KB (3 draws per CU)
  static constexpr size_t bytes = sizeof(double) * (size_t)(NP * LDW + NP * BS + BS * 3 * NP + NP / 2);
};

template <int BS>
__global__ __launch_bounds__(64) void cr_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                 const double* __restrict__ C, int batch, int n, int max_iter,
                                                 double tol, double* __restrict__ T_out,
                                                 int32_t* __restrict__ status, int32_t* __restrict__ n_iter_out,
                                                 int rerun_only, int scan_mode, const double* __restrict__ D, int k,
                                                 double* __restrict__ R_out) {
  constexpr int NP = CrSmem<BS>::NP, LD = CrSmem<BS>::LD, LDW = CrSmem<BS>::LDW;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* W = smem;
  double* Lbuf = W + NP * LDW;
  double* Ybuf = Lbuf + NP * BS;
  int* prow = (int*)(Ybuf + BS * 3 * NP);
  const int lane = threadIdx.x, lr = lane >> 3, lc = lane & 7;
  (void)LD;

  if (rerun_only && rerun_pass_is_empty(status, batch)) return;
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    // second pass of the cascade: only the draws the column-compact kernel could not take
    if (rerun_only && status[draw] != DSGE_ST_INTERNAL_RERUN) continue;
    const size_t off = (size_t)draw * n * n;
    wave_sync();
    for (int idx = lane; idx < NP * LDW; idx += 64) W[idx] = 0.0;
    double A0[BS][BS], A1[BS][BS], A2[BS][BS], Ah[BS][BS];
    blk_load_global<BS>(A0, A + off, n, n, n, lr, lc);
    blk_load_global<BS>(A1, B + off, n, n, n, lr, lc);
    blk_load_global<BS>(A2, C + off, n, n, n, lr, lc);
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) Ah[i][j] = A1[i][j];
    wave_sync();

    bool converged = false, saw_nan = false;
    int it = 0;
    for (; it < max_iter;) {
      // W = [A1 | A0 | A2]
      blk_store_lds<BS>(A1, W, LDW, lr, lc);
      if (scan_mode && lr == lc) {  // stabilize(A1): 1e-16 on the diagonal of the solve only (shared.py:6-9)
#pragma unroll
        for (int i = 0; i < BS; ++i) W[(lr * BS + i) * LDW + lc * BS + i] = A1[i][i] + 1e-16;
      }
      blk_store_lds<BS>(A0, W + NP, LDW, lr, lc);
      blk_store_lds<BS>(A2, W + 2 * NP, LDW, lr, lc);
      double inv_lo = 1e300, inv_hi = 0.0;
      gauss_jordan_blocked<BS>(W, LDW, n, 3, Lbuf, Ybuf, prow, lane, nullptr, inv_lo, inv_hi);  // syncs on entry and exit
      gj_unpermute<BS>(W, LDW, n, 1, 3, prow, lane);
      // the same single step of iterative refinement as crc_iterate (dsge_cr_compact.hpp), with the same test and the same
      // arithmetic per column: the column-compact kernels stay bit-identical to this one
      if (__builtin_amdgcn_readfirstlane((int)(inv_hi > cr_refine_ratio<BS>() * inv_lo))) {
        double x0h[BS][BS], x2h[BS][BS], r0[BS][BS], r2[BS][BS];
        blk_load_lds<BS>(x0h, W + NP, LDW, lr, lc);
        blk_load_lds<BS>(x2h, W + 2 * NP, LDW, lr, lc);
        blk_store_lds<BS>(A1, W, LDW, lr, lc);
        if (scan_mode && lr == lc) {
#pragma unroll
          for (int i = 0; i < BS; ++i) W[(lr * BS + i) * LDW + lc * BS + i] = A1[i][i] + 1e-16;
        }
        wave_sync();
        mm_residual_dot2<BS>(r0, A0, W, LDW, W + NP, LDW, n, lr, lc);      // A0 - A1 X0
        mm_residual_dot2<BS>(r2, A2, W, LDW, W + 2 * NP, LDW, n, lr, lc);  // A2 - A1 X2
        wave_sync();
        blk_store_lds<BS>(r0, W + NP, LDW, lr, lc);
        blk_store_lds<BS>(r2, W + 2 * NP, LDW, lr, lc);
        gauss_jordan_blocked<BS>(W, LDW, n, 3, Lbuf, Ybuf, prow, lane);
        gj_unpermute<BS>(W, LDW, n, 1, 3, prow, lane);
        blk_load_lds<BS>(r0, W + NP, LDW, lr, lc);
        blk_load_lds<BS>(r2, W + 2 * NP, LDW, lr, lc);
        wave_sync();
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) {
            x0h[i][j] += r0[i][j];
            x2h[i][j] += r2[i][j];
          }
        blk_store_lds<BS>(x0h, W + NP, LDW, lr, lc);
        blk_store_lds<BS>(x2h, W + 2 * NP, LDW, lr, lc);
        wave_sync();
      }
      const double* X0 = W + NP;
      const double* X2 = W + 2 * NP;
      // left operand A0 -> dead column group 0
      blk_store_lds<BS>(A0, W, LDW, lr, lc);
      wave_sync();
      double acc[BS][BS], m00[BS][BS];
      blk_zero<BS>(acc);
      blk_zero<BS>(m00);
      mm_acc<BS, false>(acc, W, LDW, X2, LDW, n, lr, lc);  // m02 = A0 X2
      mm_acc<BS, false>(m00, W, LDW, X0, LDW, n, lr, lc);  // m00 = A0 X0
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) A1[i][j] -= acc[i][j];
      wave_sync();
      blk_store_lds<BS>(A2, W, LDW, lr, lc);  // left operand A2
      wave_sync();
      double m22[BS][BS];
      blk_zero<BS>(acc);
      blk_zero<BS>(m22);
      mm_acc<BS, false>(acc, W, LDW, X0, LDW, n, lr, lc);  // m20 = A2 X0
      mm_acc<BS, false>(m22, W, LDW, X2, LDW, n, lr, lc);  // m22 = A2 X2
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) {
          A1[i][j] -= acc[i][j];
          Ah[i][j] -= acc[i][j];
          A0[i][j] = -m00[i][j];
          A2[i][j] = -m22[i][j];
        }
      ++it;
      const double nrm0 = blk_norm1<BS>(m00);
      if (nrm0 < tol) {
        const double nrm2 = blk_norm1<BS>(m22);
        if (nrm2 < tol || scan_mode) {  // the scan variant tests the A0 norm only (cycle_reduction.py:269-277)
          converged = true;
          break;
        }
      } else if (nrm0 != nrm0) {
        saw_nan = true;
        break;
      }
      wave_sync();
    }

    double Tb[BS][BS];
    blk_zero<BS>(Tb);
    const bool solve_T = converged || (scan_mode && !saw_nan);
    if (solve_T) {
      // T = -A1_hat^-1 A0_initial   (cycle_reduction.py:181)
      wave_sync();
      blk_store_lds<BS>(Ah, W, LDW, lr, lc);
      if (scan_mode && lr == lc) {
#pragma unroll
        for (int i = 0; i < BS; ++i) W[(lr * BS + i) * LDW + lc * BS + i] = Ah[i][i] + 1e-16;
      }
      {
        double t[BS][BS];
        blk_load_global<BS>(t, A + off, n, n, n, lr, lc);
        blk_store_lds<BS>(t, W + NP, LDW, lr, lc);
      }
      // with D: the same elimination gives R = -A1_hat^-1 D (= -(C T + B)^-1 D at convergence, see dsge_cr_compact.hpp)
      const bool want_R = (R_out != nullptr) && !scan_mode;
      if (want_R) {
        double t[BS][BS];
        blk_load_global<BS>(t, D + (size_t)draw * n * k, n, k, k, lr, lc);
        blk_store_lds<BS>(t, W + 2 * NP, LDW, lr, lc);
      }
      gauss_jordan_blocked<BS>(W, LDW, n, want_R ? 3 : 2, Lbuf, Ybuf, prow, lane);
      gj_unpermute<BS>(W, LDW, n, 1, want_R ? 3 : 2, prow, lane);
      blk_load_lds<BS>(Tb, W + NP, LDW, lr, lc);
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) Tb[i][j] = -Tb[i][j];
      if (want_R)
        for (int idx = lane; idx < n * k; idx += 64) {
          const int r = idx / k, c = idx - r * k;
          R_out[(size_t)draw * n * k + idx] = -W[r * LDW + 2 * NP + c];
        }
    }
    if (R_out != nullptr && !scan_mode && !solve_T)
      for (int idx = lane; idx < n * k; idx += 64) R_out[(size_t)draw * n * k + idx] = 0.0;
    blk_store_global<BS>(Tb, T_out + off, n, n, n, lr, lc);
    if (lane == 0) {
      status[draw] = solve_T ? DSGE_ST_OK : (DSGE_ST_NOT_CONVERGED | (saw_nan ? DSGE_ST_NAN : 0));
      if (n_iter_out) n_iter_out[draw] = it;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Backward-direct (gEconpy/solvers/backward_looking.py:8-24,46-78): T = (-B)^-1 A, R = -B^-1 D
// ---------------------------------------------------------------------------------------
template <int BS>
struct BdSmem {
  static constexpr int NP = Tile<BS>::NP, LDW = 3 * NP + 1;
  static constexpr size_t bytes = sizeof(double) * (size_t)(NP * LDW + NP * BS + BS * 3 * NP + NP / 2);
};

template <int BS>
__global__ __launch_bounds__(64) void bdirect_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                      const double* __restrict__ D, int batch, int n, int k,
                                                      double* __restrict__ T_out, double* __restrict__ R_out) {
  constexpr int NP = BdSmem<BS>::NP, LDW = BdSmem<BS>::LDW;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* W = smem;
  double* Lbuf = W + NP * LDW;
  double* Ybuf = Lbuf + NP * BS;
  int* prow = (int*)(Ybuf + BS * 3 * NP);
  const int lane = threadIdx.x;
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    wave_sync();
    for (int idx = lane; idx < NP * LDW; idx += 64) W[idx] = 0.0;
    wave_sync();
    const double* Bg = B + (size_t)draw * n * n;
    const double* Ag = A + (size_t)draw * n * n;
    const double* Dg = D + (size_t)draw * n * k;
    for (int idx = lane; idx < n * n; idx += 64) {
      const int r = idx / n, c = idx - r * n;
      W[r * LDW + c] = Bg[idx];
      W[r * LDW + NP + c] = Ag[idx];
    }
    for (int idx = lane; idx < n * k; idx += 64) {
      const int r = idx / k, c = idx - r * k;
      W[r * LDW + 2 * NP + c] = Dg[idx];
    }
    gauss_jordan_blocked<BS>(W, LDW, n, 3, Lbuf, Ybuf, prow, lane);
    gj_unpermute<BS>(W, LDW, n, 1, 3, prow, lane);
    for (int idx = lane; idx < n * n; idx += 64) {
      const int r = idx / n, c = idx - r * n;
      T_out[(size_t)draw * n * n + idx] = -W[r * LDW + NP + c];
    }
    for (int idx = lane; idx < n * k; idx += 64) {
      const int r = idx / k, c = idx - r * k;
      R_out[(size_t)draw * n * k + idx] = -W[r * LDW + 2 * NP + c];
    }
  }
}

// ---------------------------------------------------------------------------------------
// Assemble: R = -(C T + B)^-1 D (shared.py:74-75), resid = sum((A + (B + C T) T)^2)
// (statespace.py:213), RQR = sym(R Q R'), P0 = dlyap(T, RQR) by doubling:
//   P <- P + A_k P A_k',  A_{k+1} = A_k^2,  A_0 = T   (P = sum_j T^j RQR T'^j)
// ---------------------------------------------------------------------------------------
template <int BS>
struct AsmSmem {
  static constexpr int NP = Tile<BS>::NP, LD = Tile<BS>::LD, LDW = 2 * NP + 1;
  // M1 (NP x LD), W (NP x LDW: two column groups G0 | G1), blocked Gauss-Jordan scratch (Lbuf, Ybuf, prow)
  static constexpr size_t bytes = sizeof(double) * (size_t)(NP * LD + NP * LDW + NP * BS + BS * 2 * NP + NP / 2);
};

// sym(R diag(q) R') alone -- what is left for the assemble step of the fused call once R comes out of the solver.  One
// column of the result per lane: lane j keeps row j of R in registers, the rows of R are broadcast from LDS (n x k doubles),
// each row of the result leaves coalesced.  (r_ic r_jc) q_c is symmetric in (i, j) bit for bit, so there is no symmetrise
// pass.  2 KB of LDS and 40 registers instead of the 26 KB / 8 x 8 register-block machinery of assemble_kernel: 0.06 ->
// 0.02 ms per 4096 draws at n = 40.  k <= RQR_KMAX (the launcher falls back to assemble_kernel above that).
constexpr int RQR_KMAX = 16;
template <int KMAX>  // (a template so that the header can be included by several translation units)
__global__ __launch_bounds__(64) void rqr_kernel(const double* __restrict__ R, const double* __restrict__ q, int q_batched,
                                                  int batch, int n, int k, const int32_t* __restrict__ status,
                                                  double* __restrict__ RQR_out, int rerun_only) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  const int kp = (k + 1) & ~1;  // LDS row stride (even: b128 reads)
  if (rerun_only && rerun_pass_is_empty(status, batch)) return;
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    const size_t off = (size_t)draw * n * n, offk = (size_t)draw * n * k;
    if (rerun_only) {  // only the draws a structure-exploiting kernel handed on (it formed its own block of the product)
      if (status[draw] != DSGE_ST_INTERNAL_RERUN) continue;
    } else if (status && status[draw] != 0) {  // failed solve: zeros, so that downstream stays finite
      for (int idx = lane; idx < n * n; idx += 64) RQR_out[off + idx] = 0.0;
      continue;
    }
    wave_sync();
    lane_loop_batched<8>(
        n * k, lane, [&](int idx) { return R[offk + idx]; },
        [&](int idx, double v) {
          const int i = idx / k, c = idx - i * k;
          smem[i * kp + c] = v;
        });
    if (kp > k)
      for (int i = lane; i < n; i += 64) smem[i * kp + k] = 0.0;
    const double* qd = q + (q_batched ? (size_t)draw * k : 0);
    double qv[KMAX], rj[KMAX];
#pragma unroll
    for (int c = 0; c < KMAX; ++c) qv[c] = qd[c < k ? c : k - 1];
    wave_sync();
    const int lj = lane < n ? lane : n - 1;
#pragma unroll
    for (int c = 0; c < KMAX; ++c) {
      rj[c] = (c < kp) ? smem[lj * kp + (c < kp ? c : 0)] : 0.0;
      qv[c] = (c < k) ? qv[c] : 0.0;
    }
    double* out = RQR_out + off + lane;
    for (int i = 0; i < n; ++i) {
      const double2* ri = reinterpret_cast<const double2*>(smem + i * kp);
      double a0 = 0.0, a1 = 0.0;
#pragma unroll
      for (int c2 = 0; c2 < KMAX / 2; ++c2) {
        if (2 * c2 < kp) {
          const double2 t = ri[c2];
          a0 = fma(t.x * rj[2 * c2], qv[2 * c2], a0);
          a1 = fma(t.y * rj[2 * c2 + 1], qv[2 * c2 + 1], a1);
        }
      }
      if (lane < n) out[(size_t)i * n] = a0 + a1;
    }
  }
}

constexpr int LYAP_MAX_DOUBLINGS = 64;

template <int BS>
__global__ __launch_bounds__(64) void assemble_kernel(
    const double* __restrict__ A, const double* __restrict__ B, const double* __restrict__ C,
    const double* __restrict__ D, const double* __restrict__ T, const double* __restrict__ R_in,
    const double* __restrict__ Q, int q_mode, int batch, int n, int k, double* __restrict__ R_out,
    double* __restrict__ resid_out, double* __restrict__ RQR_out, double* __restrict__ P0_out,
    int32_t* __restrict__ status, int do_selection, int do_lyapunov, const int32_t* __restrict__ only_marked = nullptr) {
  constexpr int NP = AsmSmem<BS>::NP, LD = AsmSmem<BS>::LD, LDW = AsmSmem<BS>::LDW;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  // sym(R Q R') alone (the fused call): neither T nor the Gauss-Jordan scratch is touched, so the launcher allocates the
  // two column groups of W only (26 instead of 44 KB at n = 40: six draws per CU instead of three)
  const bool rqr_only = !do_selection && do_lyapunov == 2;
  double* M1 = smem;                              // T, later A_k
  double* W = rqr_only ? smem : M1 + NP * LD;     // two column groups (ld = LDW)
  double* G0 = W;               //   C-stage: B + C T, then [M | .] of the solve; later R, W1 / transposes
  double* G1 = W + NP;          //   C, then D -> X; later R Q, then P_k
  // Gauss-Jordan scratch: behind W -- or, when no doubling iteration follows (T in M1 is dead once the solve starts), in
  // M1 itself, and the launcher allocates M1 + W only (39 instead of 44 KB at n = 40: four draws per CU instead of three)
  const bool scratch_in_m1 = do_selection && (do_lyapunov == 0 || do_lyapunov == 2);
  static_assert(NP * BS + BS * 2 * NP + NP / 2 <= NP * LD, "Gauss-Jordan scratch must fit M1");
  double* Lbuf = scratch_in_m1 ? M1 : W + NP * LDW;
  double* Ybuf = Lbuf + NP * BS;
  int* prow = (int*)(Ybuf + BS * 2 * NP);
  const int lane = threadIdx.x, lr = lane >> 3, lc = lane & 7;

  // do_lyapunov: 0 none | 1 RQR' and P0 from R, Q | 2 RQR' only | 3 P0 from RQR_out (read) for the draws
  // flagged DSGE_ST_INTERNAL_RERUN only | 4 P0 from RQR_out (read) for every healthy draw
  const bool lyap_from_rqr = (do_lyapunov == 3 || do_lyapunov == 4);
  if (do_lyapunov == 3 && rerun_pass_is_empty(status, batch)) return;
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    const size_t off = (size_t)draw * n * n;
    const size_t offk = (size_t)draw * n * k;
    // only_marked: the selection alone, for the marked draws (gensys by spectral division: the draws the ordered QZ solved; the
    // others carry R from the cycle reduction's final elimination)
    if (only_marked && only_marked[draw] == 0) continue;
    if (do_lyapunov == 3) {
      if (status[draw] != DSGE_ST_INTERNAL_RERUN) continue;
    } else if (status && status[draw] != 0) {
      // failed solve: nothing to assemble; outputs zero-filled so downstream stays finite
      double z[BS][BS];
      blk_zero<BS>(z);
      if (do_selection && R_out) blk_store_global<BS>(z, R_out + offk, n, k, k, lr, lc);
      if (do_selection && resid_out && lane == 0) resid_out[draw] = INFINITY;
      if (do_lyapunov == 1 || do_lyapunov == 2) {
        if (RQR_out) blk_store_global<BS>(z, RQR_out + off, n, n, n, lr, lc);
      }
      if ((do_lyapunov == 1 || do_lyapunov == 4) && P0_out) blk_store_global<BS>(z, P0_out + off, n, n, n, lr, lc);
      continue;
    }
    wave_sync();
    // T is an operand of the selection (B + C T) and of the doubling iteration; sym(R Q R') alone (the fused call, where
    // R comes from the solver and P0 is left to the Kalman kernel) does not touch it
    if (!rqr_only) lds_load_matrix(M1, LD, NP, NP, T + off, n, n, lane);
    for (int idx = lane; idx < NP * LDW; idx += 64) W[idx] = 0.0;
    wave_sync();

    double Rb[BS][BS];  // R in register blocks (rows lr*BS.., cols lc*BS.. < k)
    if (do_selection) {
      {
        double Cb[BS][BS];
        blk_load_global<BS>(Cb, C + off, n, n, n, lr, lc);
        blk_store_lds<BS>(Cb, G1, LDW, lr, lc);
      }
      wave_sync();
      double Mb[BS][BS];
      blk_load_global<BS>(Mb, B + off, n, n, n, lr, lc);
      mm_acc<BS, false>(Mb, G1, LDW, M1, LD, n, lr, lc);  // M = B + C T
      wave_sync();
      blk_store_lds<BS>(Mb, G0, LDW, lr, lc);
      {
        double Db[BS][BS];
        blk_load_global<BS>(Db, D + offk, n, k, k, lr, lc);
        blk_store_lds<BS>(Db, G1, LDW, lr, lc);  // k < NP columns, the rest zero
      }
      wave_sync();
      if (resid_out) {  // before the solve destroys M:  A + (B + C T) T
        double Eb[BS][BS];
        blk_load_global<BS>(Eb, A + off, n, n, n, lr, lc);
        mm_acc<BS, false>(Eb, G0, LDW, M1, LD, n, lr, lc);
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) s = fma(Eb[i][j], Eb[i][j], s);
        s = wave_sum(s);
        if (lane == 0) resid_out[draw] = s;
      }
      wave_sync();  // (every lane is done with T in M1, which may hold the scratch from here)
      gauss_jordan_blocked<BS>(W, LDW, n, 2, Lbuf, Ybuf, prow, lane);
      gj_unpermute<BS>(W, LDW, n, 1, 2, prow, lane);
      blk_load_lds<BS>(Rb, G1, LDW, lr, lc);
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) Rb[i][j] = -Rb[i][j];
      if (R_out) blk_store_global<BS>(Rb, R_out + offk, n, k, k, lr, lc);
    } else if (!lyap_from_rqr) {
      blk_load_global<BS>(Rb, R_in + offk, n, k, k, lr, lc);
    }
    if (!do_lyapunov) continue;

    double Pb[BS][BS];
    if (!lyap_from_rqr) {
    // ---- R -> G0 (the transposed operand), R Q -> G1
    wave_sync();
    blk_store_lds<BS>(Rb, G0, LDW, lr, lc);
    if (q_mode == DSGE_Q_DIAG_SHARED || q_mode == DSGE_Q_DIAG_BATCHED) {
      const double* q = Q + (q_mode == DSGE_Q_DIAG_BATCHED ? (size_t)draw * k : 0);
      double RQ[BS][BS];
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) {
          const int c = lc * BS + j;
          RQ[i][j] = (c < k) ? Rb[i][j] * q[c] : 0.0;
        }
      blk_store_lds<BS>(RQ, G1, LDW, lr, lc);
      wave_sync();
    } else {
      const double* q = Q + (q_mode == DSGE_Q_FULL_BATCHED ? (size_t)draw * k * k : 0);
      {
        double Qb[BS][BS];
        blk_load_global<BS>(Qb, q, k, k, k, lr, lc);
        blk_store_lds<BS>(Qb, G1, LDW, lr, lc);
      }
      wave_sync();
      double RQ[BS][BS];
      blk_zero<BS>(RQ);
      mm_acc<BS, false>(RQ, G0, LDW, G1, LDW, k, lr, lc);
      wave_sync();
      blk_store_lds<BS>(RQ, G1, LDW, lr, lc);
      wave_sync();
    }
    blk_zero<BS>(Pb);
    mm_acc<BS, true>(Pb, G1, LDW, G0, LDW, k, lr, lc);  // (R Q) R'
    // symmetrise through G0
    wave_sync();
    blk_store_lds<BS>(Pb, G0, LDW, lr, lc);
    wave_sync();
    {
      double Pt[BS][BS];
      blk_load_lds_t<BS>(Pt, G0, LDW, lr, lc);
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) Pb[i][j] = 0.5 * (Pb[i][j] + Pt[i][j]);
    }
    if (RQR_out) blk_store_global<BS>(Pb, RQR_out + off, n, n, n, lr, lc);
    if (do_lyapunov == 2) continue;
    } else {
      blk_load_global<BS>(Pb, RQR_out + off, n, n, n, lr, lc);  // sym(R Q R') computed by an earlier launch
    }
    wave_sync();
    blk_store_lds<BS>(Pb, G1, LDW, lr, lc);  // P_0 = RQR
    wave_sync();

    // ---- doubling iteration; M1 = A_k, G1 = P_k, G0 = scratch
    bool lyap_ok = false;
    for (int itl = 0; itl < LYAP_MAX_DOUBLINGS; ++itl) {
      double W1[BS][BS];
      blk_zero<BS>(W1);
      mm_acc<BS, true>(W1, G1, LDW, M1, LD, n, lr, lc);  // P A_k'
      wave_sync();
      blk_store_lds<BS>(W1, G0, LDW, lr, lc);
      wave_sync();
      double Db[BS][BS], A2b[BS][BS];
      blk_zero<BS>(Db);
      blk_zero<BS>(A2b);
      mm_acc<BS, false>(Db, M1, LD, G0, LDW, n, lr, lc);   // A_k P A_k'
      mm_acc<BS, false>(A2b, M1, LD, M1, LD, n, lr, lc);   // A_k^2
      wave_sync();
      blk_store_lds<BS>(Db, G0, LDW, lr, lc);
      blk_store_lds<BS>(A2b, M1, LD, lr, lc);
      wave_sync();
      {
        double Dt[BS][BS];
        blk_load_lds_t<BS>(Dt, G0, LDW, lr, lc);
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) {
            Db[i][j] = 0.5 * (Db[i][j] + Dt[i][j]);
            Pb[i][j] += Db[i][j];
          }
      }
      const double dmax = blk_maxabs<BS>(Db);
      const double pmax = blk_maxabs<BS>(Pb);
      wave_sync();
      blk_store_lds<BS>(Pb, G1, LDW, lr, lc);
      wave_sync();
      if (!(dmax == dmax) || !(pmax < 1e300)) break;  // NaN / overflow: rho(T) >= 1
      if (dmax <= 1e-17 * pmax) {
        lyap_ok = true;
        break;
      }
    }
    blk_store_global<BS>(Pb, P0_out + off, n, n, n, lr, lc);
    if (!lyap_ok && status && lane == 0) status[draw] |= DSGE_ST_LYAP_FAIL;
  }
}

// ---------------------------------------------------------------------------------------
// Adjoints of the policy function (reverse mode through A + B T + C T T = 0), replacing the
// n^2 x n^2 Kronecker solve of o1_policy_function_adjoints (gEconpy/solvers/shared.py:12-71):
//   (kron(T, C') + kron(I, T'C') + kron(I, B')) vec(S) = -vec(T_bar)   <=>   M' S + C' S T' = -T_bar,
// M = B + C T.  With H = -M^-T T_bar, G = -M^-T C', F = T' this is the Stein equation S = H + G S F,
// solved by the doubling iteration S <- S + G_k S F_k, G_{k+1} = G_k^2, F_{k+1} = F_k^2, which
// converges because the eigenvalues of M^-1 C are -1/(unstable roots) and rho(T) < 1
// ((C lambda + M)(lambda - T) = C lambda^2 + B lambda + A).  Then A_bar = S, B_bar = S T',
// C_bar = S T' T' (shared.py:67-69).  ~10 doublings x 4 products of n^3 instead of a 2.7 GFLOP LU.
// ---------------------------------------------------------------------------------------
template <int BS>
struct AdjSmem {
  static constexpr int NP = Tile<BS>::NP, LD = Tile<BS>::LD, LDW = 3 * NP + 1;
  // W (three column groups) and one NP x LD matrix; T is kept in the third column group of W while it is free and read
  // again from global memory for the final products, and the Gauss-Jordan scratch lives in the NP x LD matrix while that
  // is free: 51.8 instead of 71.5 KB at n = 40, three draws per CU instead of two
  static constexpr size_t bytes = sizeof(double) * (size_t)(NP * LDW + NP * LD);
  static_assert(NP * BS + BS * 3 * NP + NP / 2 <= NP * LD, "Gauss-Jordan scratch must fit the NP x LD matrix");
  // the second pass (REFINE) only: the recorded elimination of its fixed-point fall-back (8 block steps x NP x BS multipliers) and
  // the replay's pivot-row scratch (BS x NP) behind the first pass's layout -- few draws take that pass, its occupancy is free
  static constexpr size_t refine_extra_doubles = (size_t)8 * NP * BS + (size_t)BS * NP + NP / 2 + 1;  // (+ NP ints: the pivot rows)
  static constexpr size_t bytes_refine = bytes + sizeof(double) * refine_extra_doubles;
};

// One Stein solve on the wavefront: S = H + G S F with H = -M^-T rhs, G = -M^-T C', F = T' (see above), everything from
// global memory; on return S is in `Sb` and in the second column group of W; ks = last non-zero column of T + 1.
template <int BS>
__device__ __forceinline__ bool adj_stein_solve(double* W, double* Tk, const double* __restrict__ B,
                                                const double* __restrict__ C, const double* __restrict__ T, size_t off, int n,
                                                const double (&Hb)[BS][BS], double (&Sb)[BS][BS], int lane, double& gmax) {
  constexpr int NP = AdjSmem<BS>::NP, LD = AdjSmem<BS>::LD, LDW = AdjSmem<BS>::LDW;
  double* Ts = W + 2 * NP;      // T (row stride LDW): before W is filled
  double* Lbuf = Tk;
  double* Ybuf = Lbuf + NP * BS;
  int* prow = (int*)(Ybuf + BS * 3 * NP);
  const int lr = lane >> 3, lc = lane & 7;
  wave_sync();
  for (int idx = lane; idx < NP * LDW; idx += 64) W[idx] = 0.0;
  wave_sync();
  lds_load_matrix(Ts, LDW, NP, NP, T + off, n, n, lane);
  lds_load_matrix(Tk, LD, NP, NP, C + off, n, n, lane);
  wave_sync();
  // T (and every power of it) has non-zero columns only for the state variables; when those end at column ks the two
  // products that contract over the columns of T_k run over ks terms instead of n (18 instead of 40 on the SW-shaped
  // systems, whose states lead; ks = n and nothing changes when a state sits in the last column)
  int ks = n;
  {
    bool nz = false;
    if (lane < n)
      for (int r = 0; r < n; ++r) nz = nz | (Ts[r * LDW + lane] != 0.0);
    const unsigned long long cm = __ballot(nz);
    ks = cm ? 64 - __clzll((long long)cm) : 0;
  }
  {
    double Mb[BS][BS], Cb[BS][BS];
    blk_load_global<BS>(Mb, B + off, n, n, n, lr, lc);
    mm_acc<BS, false>(Mb, Tk, LD, Ts, LDW, n, lr, lc);  // M = B + C T
    wave_sync();  // T (third column group of W) and C (Tk) are dead from here: C' and the Gauss-Jordan scratch take over
    blk_load_global<BS>(Cb, C + off, n, n, n, lr, lc);
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        W[(lc * BS + j) * LDW + lr * BS + i] = Mb[i][j];            // M'
        W[(lc * BS + j) * LDW + 2 * NP + lr * BS + i] = Cb[i][j];   // C'
      }
    blk_store_lds<BS>(Hb, W + NP, LDW, lr, lc);
  }
  gauss_jordan_blocked<BS>(W, LDW, n, 3, Lbuf, Ybuf, prow, lane);
  gj_unpermute<BS>(W, LDW, n, 1, 3, prow, lane);
  {
    double Gb[BS][BS];
    blk_load_lds<BS>(Sb, W + NP, LDW, lr, lc);
    blk_load_lds<BS>(Gb, W + 2 * NP, LDW, lr, lc);
    wave_sync();
    double Tb[BS][BS];
    blk_load_global<BS>(Tb, T + off, n, n, n, lr, lc);
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        Sb[i][j] = -Sb[i][j];
        Gb[i][j] = -Gb[i][j];
      }
    gmax = blk_maxabs<BS>(Gb);
    blk_store_lds<BS>(Sb, W + NP, LDW, lr, lc);      // S_0 = H
    blk_store_lds<BS>(Gb, W + 2 * NP, LDW, lr, lc);  // G_0
    blk_store_lds<BS>(Tb, Tk, LD, lr, lc);           // F_0' = T
    wave_sync();
  }
  bool ok = false;
  for (int it = 0; it < LYAP_MAX_DOUBLINGS; ++it) {
    double W1[BS][BS];
    blk_zero<BS>(W1);
    mm_acc<BS, true>(W1, W + NP, LDW, Tk, LD, ks, lr, lc);  // S F_k = S (T^(2^k))'
    blk_store_lds<BS>(W1, W, LDW, lr, lc);
    wave_sync();
    double Ib[BS][BS], G2[BS][BS], T2[BS][BS];
    blk_zero<BS>(Ib);
    blk_zero<BS>(G2);
    blk_zero<BS>(T2);
    mm_acc<BS, false>(Ib, W + 2 * NP, LDW, W, LDW, n, lr, lc);           // G_k S F_k
    mm_acc<BS, false>(G2, W + 2 * NP, LDW, W + 2 * NP, LDW, n, lr, lc);  // G_k^2
    mm_acc<BS, false>(T2, Tk, LD, Tk, LD, ks, lr, lc);                   // T_k^2
    wave_sync();
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) Sb[i][j] += Ib[i][j];
    blk_store_lds<BS>(Sb, W + NP, LDW, lr, lc);
    blk_store_lds<BS>(G2, W + 2 * NP, LDW, lr, lc);
    blk_store_lds<BS>(T2, Tk, LD, lr, lc);
    const double dmax = blk_maxabs<BS>(Ib), smax = blk_maxabs<BS>(Sb);
    wave_sync();
    if (!(dmax == dmax) || !(smax < 1e300)) break;
    if (dmax <= 1e-17 * smax) {
      ok = true;
      break;
    }
    gmax = fmax(gmax, blk_maxabs<BS>(G2));  // growth of the powers of G: what the refinement rule looks at
  }
  return ok;
}

// Round 6: the same Stein equation on its COMPACT form.  C has non-zero columns only for the nl variables that appear with a
// lead (index set L), T only for the ns state variables (index set St), so  G = -M^-T C' = -Wm V  with  Wm = M^-T E_L (n x nl),
// V = C_L' (nl x n), and  S = H - Wm (V S) T'.  Z = V S (nl x n) solves  Z = V H + Gs Z T',  Gs = -V Wm (nl x nl), and Z T' needs
// only the columns St of Z:  Zs = Z0s + Gs Zs Tss',  Tss = T[St, St] -- a Stein equation of size nl x ns (12 x 18 on the SW-shaped
// systems instead of 40 x 40), solved by the same doubling on tiles of 8 BSC; then  S = H - (Wm Zs) T[:, St]'.  Per doubling
// 4 products of <= 24^3 instead of 4 of 40^3.  And it is the BETTER conditioned iteration: the powers of the embedded G grow before
// they decay (max |G^(2^k)| up to 1e3 on the SW-shaped draws, 1e25 in float64 on draw 752 whose B + C T has condition 3e8 --
// which is why that form needs a refinement pass and an elimination-based fall-back), those of Gs do not (|Gs| < 1 on all of the
// first 800 draws): against the reference's Kronecker solve (shared.py:53-71) the full form is off by up to 2e-8 before refinement,
// the compact one by 1e-12, and 1.9e-8 on draw 752 (numpy model: tests/device_models/adjoint_compact_model.py).
// `took` = false (and nothing else done) when L or St do not fit the compact tile; the caller then runs adj_stein_solve.
template <int BS>
struct AdjCompact {
  static constexpr int BSC = BS <= 3 ? 1 : (BS - 2 > 4 ? 4 : BS - 2);
  static constexpr int NPC = 8 * BSC, LDC = NPC + 1;
  static constexpr bool enabled = BS >= 4;
};

// The pullback of R = -(B + C T)^-1 D riding on the same elimination (round 6, the gradient pipeline's fused assembly + adjoint launch):
// with Rbar the cotangent of R,  X = M^-T Rbar  gives  Dbar = -X,  Mbar = -X R'  (the cotangent of M: Bbar += Mbar, Cbar += Mbar T')
// and a further cotangent  C' Mbar  of T -- which enters the Stein equation through  H = -M^-T (T_bar + C' Mbar) = H_f +
// Wm (C_L' X) R'  (M^-T C' = Wm C_L'): three thin products, no second factorisation of M.  Rbar rides in the third column group
// behind E_L (columns nl .. nl + k); X stays there until the caller has formed  Bbar = -X R' + S T',  Cbar = Bbar T'.
template <int BS>
struct AdjFuse {
  const double (*Rbar)[BS];     // register block image of Rbar (n x k, columns >= k zero)
  const double* R;              // this draw's selection matrix, n x k (global)
  int k;
  double* D_bar;                // this draw's output, n x k (global)
  int x_col;                    // out: column of X inside the third group
};

template <int BS>
__device__ __forceinline__ bool adj_stein_solve_compact(double* W, double* Tk, const double* __restrict__ B,
                                                        const double* __restrict__ C, const double* __restrict__ T, size_t off,
                                                        int n, const double (&Hb)[BS][BS], double (&Sb)[BS][BS], int lane,
                                                        double& gmax, bool& took, int& t_cols, AdjFuse<BS>* fz = nullptr) {
  constexpr int NP = AdjSmem<BS>::NP, LD = AdjSmem<BS>::LD, LDW = AdjSmem<BS>::LDW;
  constexpr int BSC = AdjCompact<BS>::BSC, NPC = AdjCompact<BS>::NPC, LDC = AdjCompact<BS>::LDC;
  constexpr int TK_FREE = NP * LD - NPC;  // the two index lists (2 NPC ints) sit at the end of the NP x LD matrix
  static_assert(NP * BS + BS * 3 * NP + NP / 2 <= TK_FREE && 2 * NPC * LDC <= TK_FREE && NP * LDC <= TK_FREE, "compact layout");
  double* Ts = W + 2 * NP;
  double* Lbuf = Tk;
  double* Ybuf = Lbuf + NP * BS;
  int* prow = (int*)(Ybuf + BS * 3 * NP);
  int* Lidx = (int*)(Tk + TK_FREE);
  int* Sidx = Lidx + NPC;
  double* Gs = Tk;              // NPC x LDC   Gs_k
  double* Fs = Tk + NPC * LDC;  // NPC x LDC   Tss_k
  double* ZP = W;               // NPC x NPC in the first column group of W (row stride LDW): Zs, then Zs Tss_k'
  const int lr = lane >> 3, lc = lane & 7;
  wave_sync();
  for (int idx = lane; idx < NP * LDW; idx += 64) W[idx] = 0.0;
  wave_sync();
  lds_load_matrix(Ts, LDW, NP, NP, T + off, n, n, lane);
  lds_load_matrix(Tk, LD, NP, NP, C + off, n, n, lane);
  wave_sync();
  bool nzT = false, nzC = false;
  if (lane < n)
    for (int r = 0; r < n; ++r) {
      nzT = nzT | (Ts[r * LDW + lane] != 0.0);
      nzC = nzC | (Tk[r * LD + lane] != 0.0);
    }
  const unsigned long long cmT = __ballot(nzT), cmC = __ballot(nzC);
  const int ns = __popcll(cmT), nl = __popcll(cmC);
  took = ns >= 1 && nl >= 1 && ns <= NPC && nl <= NPC;
  const int kf = fz ? fz->k : 0, NPK = 8 * ((kf + 7) / 8);  // (fused: R and the thin products take NPK + NPC columns of group 0)
  if (fz) took = took && kf >= 1 && nl + kf <= NP && NPK + NPC <= NP && kf <= NPC;
  if (!took) return false;
  t_cols = 64 - __clzll((long long)cmT);  // last non-zero column of T + 1
  {
    double Mb[BS][BS];
    blk_load_global<BS>(Mb, B + off, n, n, n, lr, lc);
    // M = B + C T: the contraction runs over the columns of C between its first and its last non-zero one
    const int c_lo = __ffsll((long long)cmC) - 1, c_hi = 64 - __clzll((long long)cmC);
    mm_acc<BS, false>(Mb, Tk + c_lo, LD, Ts + c_lo * LDW, LDW, c_hi - c_lo, lr, lc);
    wave_sync();  // T (third column group of W) and C (Tk) are dead from here
    if (lane < n) {
      const unsigned long long below = (1ull << lane) - 1ull;
      if ((cmC >> lane) & 1ull) Lidx[__popcll(cmC & below)] = lane;
      if ((cmT >> lane) & 1ull) Sidx[__popcll(cmT & below)] = lane;
    }
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        W[(lc * BS + j) * LDW + lr * BS + i] = Mb[i][j];  // M'
        W[(lc * BS + j) * LDW + 2 * NP + lr * BS + i] = 0.0;
      }
    blk_store_lds<BS>(Hb, W + NP, LDW, lr, lc);
  }
  wave_sync();
  if (lane < nl) W[Lidx[lane] * LDW + 2 * NP + lane] = 1.0;  // E_L
  if (fz) {  // Rbar behind E_L
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j)
        if (lc * BS + j < kf) W[(lr * BS + i) * LDW + 2 * NP + nl + lc * BS + j] = fz->Rbar[i][j];
    fz->x_col = nl;
  }
  gauss_jordan_blocked<BS>(W, LDW, n, 3, Lbuf, Ybuf, prow, lane);
  gj_unpermute<BS>(W, LDW, n, 1, 3, prow, lane);
  {
    double Zr[BS][BS];
    blk_zero<BS>(Zr);
    blk_load_lds<BS>(Sb, W + NP, LDW, lr, lc);
    wave_sync();
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) Sb[i][j] = -Sb[i][j];
    blk_store_lds<BS>(Sb, W + NP, LDW, lr, lc);  // H = -M^-T rhs
    blk_store_lds<BS>(Zr, W, LDW, lr, lc);       // the first column group: zero (it takes the compact matrices)
  }
  // C_L (n x nl, row stride LDC) staged in the NP x LD matrix; H[:, St] gathered into the (zeroed) first column group
  lane_loop_batched<8>(n * nl, lane,
                       [&](int idx) {
                         const int r = idx / nl, a = idx - r * nl;
                         return C[off + (size_t)r * n + Lidx[a]];
                       },
                       [&](int idx, double v) {
                         const int r = idx / nl, a = idx - r * nl;
                         Tk[r * LDC + a] = v;
                       });
  for (int idx = lane; idx < NP * (LDC - nl); idx += 64) {  // (the tile reads all NPC columns: zero beyond nl)
    const int r = idx / (LDC - nl), a = nl + idx - r * (LDC - nl);
    Tk[r * LDC + a] = 0.0;
  }
  wave_sync();  // (the zeros of the first column group and H in the second are in place)
  if (fz) {
    // ---- Dbar = -X;  H += Wm (C_L' X) R'  through the (zeroed) first column group: R in its columns [0, NPK), the thin factors in
    //      [NPK, NPK + NPC) ----------------------------------------------------------------------------------------------------
    const double* Xp = W + 2 * NP + nl;
    for (int idx = lane; idx < n * kf; idx += 64) {
      const int r = idx / kf, c = idx - r * kf;
      fz->D_bar[idx] = -Xp[r * LDW + c];
    }
    lane_loop_batched<8>(n * kf, lane, [&](int idx) { return fz->R[idx]; },
                         [&](int idx, double v) {
                           const int r = idx / kf, c = idx - r * kf;
                           W[r * LDW + c] = v;
                         });
    double Y1[BSC][BSC];
    blk_zero<BSC>(Y1);
    mm_acc_ta<BSC>(Y1, Tk, LDC, Xp, LDW, n, lr, lc);  // C_L' X  (nl x k)
#pragma unroll
    for (int i = 0; i < BSC; ++i)
#pragma unroll
      for (int j = 0; j < BSC; ++j) W[(lr * BSC + i) * LDW + NPK + lc * BSC + j] = (lr * BSC + i < nl && lc * BSC + j < kf) ? Y1[i][j] : 0.0;
    wave_sync();
    double Y2[BS][BS];
    blk_zero<BS>(Y2);
    mm_acc<BS, false>(Y2, W + 2 * NP, LDW, W + NPK, LDW, nl, lr, lc);  // Wm (C_L' X)  (n x k; columns beyond k: whatever follows)
    wave_sync();
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j)
        if (lc * BS + j < NPC) W[(lr * BS + i) * LDW + NPK + lc * BS + j] = (lc * BS + j < kf) ? Y2[i][j] : 0.0;
    wave_sync();
    double Hc[BS][BS];
    blk_zero<BS>(Hc);
    mm_acc<BS, true>(Hc, W + NPK, LDW, W, LDW, kf, lr, lc);  // (Wm C_L' X) R'
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) Sb[i][j] += Hc[i][j];
    wave_sync();
    blk_store_lds<BS>(Sb, W + NP, LDW, lr, lc);
    {
      double Zr[BS][BS];
      blk_zero<BS>(Zr);
      blk_store_lds<BS>(Zr, W, LDW, lr, lc);  // the first column group back to zero
    }
    wave_sync();
  }
  for (int idx = lane; idx < n * ns; idx += 64) {
    const int r = idx / ns, b = idx - r * ns;
    W[r * LDW + b] = W[r * LDW + NP + Sidx[b]];
  }
  wave_sync();
  // Gs_0 = -C_L' Wm and Zs_0 = C_L' H[:, St] on the compact tile (contraction over the n rows)
  double Gb0[BSC][BSC];
  double Zb[BSC][BSC];
  blk_zero<BSC>(Gb0);
  blk_zero<BSC>(Zb);
  mm_acc_ta<BSC>(Gb0, Tk, LDC, W + 2 * NP, LDW, n, lr, lc);
  mm_acc_ta<BSC>(Zb, Tk, LDC, W, LDW, n, lr, lc);
#pragma unroll
  for (int i = 0; i < BSC; ++i)
#pragma unroll
    for (int j = 0; j < BSC; ++j) {
      const int a = lr * BSC + i, b = lc * BSC + j;
      Gb0[i][j] = (a < nl && b < nl) ? -Gb0[i][j] : 0.0;
      Zb[i][j] = (a < nl && b < ns) ? Zb[i][j] : 0.0;
    }
  gmax = blk_maxabs<BSC>(Gb0);
  wave_sync();
  for (int idx = lane; idx < 2 * NPC * LDC; idx += 64) Tk[idx] = 0.0;
  for (int idx = lane; idx < n * ns; idx += 64) {  // the gathered H: back to zero (the first column group takes Zs)
    const int r = idx / ns, b = idx - r * ns;
    W[r * LDW + b] = 0.0;
  }
  wave_sync();
  blk_store_lds<BSC>(Gb0, Gs, LDC, lr, lc);
  blk_store_lds<BSC>(Zb, ZP, LDW, lr, lc);
  lane_loop_batched<8>(ns * ns, lane,
                       [&](int idx) {
                         const int i = idx / ns, j = idx - i * ns;
                         return T[off + (size_t)Sidx[i] * n + Sidx[j]];
                       },
                       [&](int idx, double v) {
                         const int i = idx / ns, j = idx - i * ns;
                         Fs[i * LDC + j] = v;
                       });
  wave_sync();
  bool ok = false;
  for (int it = 0; it < LYAP_MAX_DOUBLINGS; ++it) {
    {
      double P1[BSC][BSC];
      blk_zero<BSC>(P1);
      mm_acc<BSC, true>(P1, ZP, LDW, Fs, LDC, ns, lr, lc);  // Zs Tss_k'
      wave_sync();
      blk_store_lds<BSC>(P1, ZP, LDW, lr, lc);
    }
    wave_sync();
    double Ib[BSC][BSC], G2[BSC][BSC], T2[BSC][BSC];
    blk_zero<BSC>(Ib);
    blk_zero<BSC>(G2);
    blk_zero<BSC>(T2);
    mm_acc<BSC, false>(Ib, Gs, LDC, ZP, LDW, nl, lr, lc);  // Gs_k Zs Tss_k'
    mm_acc<BSC, false>(G2, Gs, LDC, Gs, LDC, nl, lr, lc);  // Gs_k^2
    mm_acc<BSC, false>(T2, Fs, LDC, Fs, LDC, ns, lr, lc);  // Tss_k^2
    wave_sync();
#pragma unroll
    for (int i = 0; i < BSC; ++i)
#pragma unroll
      for (int j = 0; j < BSC; ++j) Zb[i][j] += Ib[i][j];
    blk_store_lds<BSC>(Zb, ZP, LDW, lr, lc);
    blk_store_lds<BSC>(G2, Gs, LDC, lr, lc);
    blk_store_lds<BSC>(T2, Fs, LDC, lr, lc);
    const double dmax = blk_maxabs<BSC>(Ib), smax = blk_maxabs<BSC>(Zb);
    wave_sync();
    if (!(dmax == dmax) || !(smax < 1e300)) break;
    if (dmax <= 1e-17 * smax) {
      ok = true;
      break;
    }
    gmax = fmax(gmax, blk_maxabs<BSC>(G2));
  }
  // S = H - (Wm Zs) T[:, St]'
  {
    double Ub[BS][BS];
    blk_zero<BS>(Ub);
    mm_acc<BS, false>(Ub, W + 2 * NP, LDW, ZP, LDW, nl, lr, lc);  // Wm Zs (n x ns; the columns >= ns of the group are zero)
    wave_sync();
    blk_store_lds<BS>(Ub, W, LDW, lr, lc);
    for (int idx = lane; idx < TK_FREE; idx += 64) Tk[idx] = 0.0;
    wave_sync();
    lane_loop_batched<8>(n * ns, lane,
                         [&](int idx) {
                           const int j = idx / ns, k = idx - j * ns;
                           return T[off + (size_t)j * n + Sidx[k]];
                         },
                         [&](int idx, double v) {
                           const int j = idx / ns, k = idx - j * ns;
                           Tk[j * LDC + k] = v;  // (row stride LDC: NP rows end below the index lists)
                         });
    wave_sync();
    double Db[BS][BS];
    blk_zero<BS>(Db);
    mm_acc<BS, true>(Db, W, LDW, Tk, LDC, ns, lr, lc);  // (Wm Zs) T[:, St]'
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) Sb[i][j] -= Db[i][j];
    wave_sync();
    blk_store_lds<BS>(Sb, W + NP, LDW, lr, lc);
    wave_sync();
  }
  return ok;
}

// The fall-back of the second pass: X <- -M^-T (T_bar + C' X T') with an ELIMINATION per sweep instead of powers of an explicit
// G = -M^-T C'.  When M = B + C T is nearly singular (SW-shaped draw 752: cond 3e8, max|G| = 2e7) G is only known to
// cond x eps x |G| ~ 0.6 in absolute terms: its computed powers explode (true |G^16| = 1.7e5, float64 1e25) and even the plain
// fixed point with the explicit G stalls at 3 % -- the eigenvalues of G are the reciprocals of the unstable roots, well
// conditioned as functions of (M, C) but not as functions of the entries of G.  A backward-stable solve per sweep is a sweep
// with a slightly perturbed M: it contracts at rho(G) rho(T) (0.53 per sweep on that draw, 30 sweeps) down to the level of the
// reference's Kronecker LU (5e-9 of exact against 1.5e-8, tools/adjoint_fixed_point_model.py).  Starts from the first
// pass's S (zero when that pass failed) and returns the CORRECTION X - S in Sb and in the second column group of W.
constexpr int ADJ_FP_MAX_SWEEPS = 200;
constexpr double ADJ_FP_RESIDUAL = 1e-4;  // relative Stein residual of the first pass above which the second pass eliminates
// Round 5: M' is eliminated ONCE (sweep 0, multipliers recorded: gauss_jordan_blocked's rec_L) and every later sweep replays that
// elimination on its right-hand side (gj_replay: bit-identical to eliminating [M' | rhs] again -- the same backward-stable solve,
// the same iterates) -- a sweep is eight rank-BS updates and two products instead of a 40-pivot elimination with its panels:
// 97 k -> ~25 k cycles per sweep, the second pass on the batch with draw 752 1.78 -> see profiles/r5/grad_rate.txt.
// `extra`: AdjSmem::refine_extra_doubles of LDS.
template <int BS>
__device__ __forceinline__ bool adj_stein_fixed_point(double* W, double* Tk, double* extra, const double* __restrict__ B,
                                                      const double* __restrict__ C, const double* __restrict__ T,
                                                      const double* __restrict__ T_bar, const double* __restrict__ S0, size_t off,
                                                      int n, double (&Sb)[BS][BS], int lane) {
  constexpr int NP = AdjSmem<BS>::NP, LD = AdjSmem<BS>::LD, LDW = AdjSmem<BS>::LDW;
  double* Ts = W + 2 * NP;
  double* Lbuf = Tk;
  double* Ybuf = Lbuf + NP * BS;
  int* prow = (int*)(Ybuf + BS * 3 * NP);
  double* rec_L = extra;                        // [8][NP][BS]
  double* Ybuf2 = extra + (size_t)8 * NP * BS;  // [BS][NP]
  const int lr = lane >> 3, lc = lane & 7;
  wave_sync();
  lds_load_matrix(W + NP, LDW, NP, NP, S0 + off, n, n, lane);  // X_0
  lds_load_matrix(Ts, LDW, NP, NP, T + off, n, n, lane);
  wave_sync();
  bool ok = false;
  double best = 1e300;
  int since = 0;  // sweeps since the step last shrank by 10 % (the steps of a complex pair of modes are not monotone)
  double Tb0[BS][BS];
  blk_load_global<BS>(Tb0, T_bar + off, n, n, n, lr, lc);
  int* prow2 = (int*)(Ybuf2 + BS * NP);  // the pivot rows, kept outside Tk (which C takes back from sweep 1 on)
  for (int sweep = 0; sweep < ADJ_FP_MAX_SWEEPS; ++sweep) {
    {
      double P1[BS][BS];
      blk_zero<BS>(P1);
      mm_acc<BS, true>(P1, W + NP, LDW, Ts, LDW, n, lr, lc);  // X T'
      wave_sync();
      blk_store_lds<BS>(P1, W, LDW, lr, lc);
    }
    // C in Tk: sweep 0's elimination uses Tk as its scratch, so C is loaded for sweep 0 and once more for sweep 1; it stays after
    if (sweep <= 1) lds_load_matrix(Tk, LD, NP, NP, C + off, n, n, lane);
    wave_sync();
    double Xo[BS][BS];
    blk_load_lds<BS>(Xo, W + NP, LDW, lr, lc);
    double Rr[BS][BS];
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) Rr[i][j] = Tb0[i][j];
    mm_acc_ta<BS>(Rr, Tk, LD, W, LDW, n, lr, lc);  // T_bar + C' X T'
    if (sweep == 0) {
      double Mb[BS][BS];
      blk_load_global<BS>(Mb, B + off, n, n, n, lr, lc);
      mm_acc<BS, false>(Mb, Tk, LD, Ts, LDW, n, lr, lc);  // M = B + C T
      wave_sync();
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) W[(lc * BS + j) * LDW + lr * BS + i] = Mb[i][j];  // M'
      blk_store_lds<BS>(Rr, W + NP, LDW, lr, lc);
      gauss_jordan_blocked<BS>(W, LDW, n, 2, Lbuf, Ybuf, prow, lane, nullptr, rec_L);
      for (int idx = lane; idx < NP; idx += 64) prow2[idx] = prow[idx];  // (prow sits in Tk, which C takes back)
      wave_sync();
    } else {
      wave_sync();  // (every lane has read X and P1)
      blk_store_lds<BS>(Rr, W + NP, LDW, lr, lc);
      gj_replay<BS>(W + NP, LDW, n, rec_L, prow2, Ybuf2, lane);
    }
    gj_unpermute<BS>(W, LDW, n, 1, 2, prow2, lane);
    double Xn[BS][BS], Df[BS][BS];
    blk_load_lds<BS>(Xn, W + NP, LDW, lr, lc);
    wave_sync();
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        Xn[i][j] = -Xn[i][j];
        Df[i][j] = Xn[i][j] - Xo[i][j];
      }
    blk_store_lds<BS>(Xn, W + NP, LDW, lr, lc);
    const double dmax = blk_maxabs<BS>(Df), xmax = blk_maxabs<BS>(Xn);
    wave_sync();
    if (!(dmax == dmax) || !(xmax < 1e300)) break;
    if (dmax < 0.9 * best) {
      best = dmax;
      since = 0;
    } else {
      ++since;
    }
    // converged, or at the noise floor of an ill-conditioned M (the step is small and has stopped shrinking)
    if (dmax <= 1e-15 * xmax || (dmax <= 1e-6 * xmax && since >= 8)) {
      ok = true;
      break;
    }
  }
  {
    double t0[BS][BS];
    blk_load_global<BS>(t0, S0 + off, n, n, n, lr, lc);
    blk_load_lds<BS>(Sb, W + NP, LDW, lr, lc);
    wave_sync();
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) Sb[i][j] -= t0[i][j];
    blk_store_lds<BS>(Sb, W + NP, LDW, lr, lc);
    wave_sync();
  }
  return ok;
}

// REFINE = false: the solve and the outputs.  The doubling sums S = sum_k G^k H F^k; when the powers of G = -(B + C T)^-T C'
// grow before they decay (a non-normal G) it loses digits -- 6e-7 relative on 1 of ~160 random systems where the reference's
// Kronecker LU (shared.py:53-71) keeps 1e-12.  The loss follows max_k max|G^(2^k)| closely (numpy emulation of the iteration
// over 400 random systems: error <~ 5e-14 x growth^2; growth 1e3 -> 6e-8, 2e2 -> 3e-10, below 1e2 never above 4e-11), and
// that maximum costs one wave reduction per doubling: a draw whose growth exceeds ADJ_REFINE_GROWTH is flagged
// DSGE_ST_INTERNAL_RERUN.
// REFINE = true, the OUT-OF-LINE second pass (a launch of its own on the flagged draws only, empty otherwise): the residual
// rho = T_bar + M' S + C' S T' of the first pass's S (read back from A_bar; three products), one step of iterative refinement
// S += solve(rho), added to all three outputs.  Inlined into the first pass the second solve cost every draw 1.1-1.6 KB of
// scratch (round 2); behind a device function call the kernel loses the 256 accumulation registers its 40-wide instance
// spills into; measuring the residual in the first pass cost it 30 % (1.05 -> 1.37 ms per 4096 draws).
constexpr double ADJ_REFINE_GROWTH = 150.0;  // (error <~ 5e-14 x growth^2: 1e-9 at ~140; 3..6 % of the SW-shaped draws are above 100)

// FUSED = true (round 6; the gradient pipeline, diagonal Q): the reverse of the state-space assembly (grad_assemble_kernel's job: the
// pullback of sym(R Q R') and of R = -(B + C T)^-1 D) rides on this kernel's elimination -- see AdjFuse -- and the launch writes all
// of A_bar, B_bar, C_bar, D_bar, q_bar; T_bar is the filter's cotangent alone and is not written.  A draw the compact solve cannot
// take, or whose solve asks for refinement, is left untouched and flagged DSGE_ST_INTERNAL_RERUN for the two-kernel path
// (grad_assemble_kernel and this kernel with only_flag = 1, which visit the flagged draws only); failed draws get zero cotangents.
struct AdjFuseArgs {
  const double* R = nullptr;
  const double* q = nullptr;
  int q_batched = 0;
  const double* Gbar = nullptr;
  int k = 0;
  double* D_bar = nullptr;
  double* q_bar = nullptr;
};

template <int BS, bool REFINE, bool FUSED = false>
__global__ __launch_bounds__(64, (BS <= 3 ? 2 : 1)) void adjoint_kernel(  // (BS = 3: 258 registers without the bound)
    const double* __restrict__ B, const double* __restrict__ C,
                                                      const double* __restrict__ T, const double* __restrict__ T_bar,
                                                      int batch, int n, double* __restrict__ A_bar,
                                                      double* __restrict__ B_bar, double* __restrict__ C_bar,
                                                      int32_t* __restrict__ status, int accumulate, int refine_mode,
                                                      int only_flag = 0, AdjFuseArgs fa = AdjFuseArgs()) {
  static_assert(!(REFINE && FUSED), "the refinement pass belongs to the two-kernel path");
  constexpr int NP = AdjSmem<BS>::NP, LD = AdjSmem<BS>::LD, LDW = AdjSmem<BS>::LDW;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* W = smem;             // [M' | T_bar | C'] -> [. | M^-T T_bar | M^-T C'];  later [W1 | S | G_k]
  double* Tk = W + NP * LDW;    // C at first, then the Gauss-Jordan scratch, then T^(2^k)
  double* Ts = W + 2 * NP;      // T (row stride LDW): before W is filled, and again for the final products
  const int lane = threadIdx.x, lr = lane >> 3, lc = lane & 7;
  for (int draw = blockIdx.x; draw < batch; draw = batch) {  // (one draw per workgroup, grid = batch: see kalman_nt_kernel)
    const size_t off = (size_t)draw * n * n;
    double Sb[BS][BS];
    bool ok = true, flag = false;
    int st_in = 0, kt_cols = 0;
    double gmax = 0.0;
    if constexpr (REFINE) {
      st_in = status[draw];
      if (!(st_in & DSGE_ST_INTERNAL_RERUN)) continue;
      // the residual rho = T_bar + M' S + C' P1, P1 = S T', of the first pass's S
      wave_sync();
      for (int idx = lane; idx < NP * LDW; idx += 64) W[idx] = 0.0;
      wave_sync();
      lds_load_matrix(W + NP, LDW, NP, NP, A_bar + off, n, n, lane);
      lds_load_matrix(Ts, LDW, NP, NP, T + off, n, n, lane);
      lds_load_matrix(Tk, LD, NP, NP, C + off, n, n, lane);
      wave_sync();
      double Rr[BS][BS];
      {
        double Mb[BS][BS];
        blk_load_global<BS>(Mb, B + off, n, n, n, lr, lc);
        mm_acc<BS, false>(Mb, Tk, LD, Ts, LDW, n, lr, lc);  // M = B + C T
        blk_store_lds<BS>(Mb, W, LDW, lr, lc);
      }
      blk_load_global<BS>(Rr, T_bar + off, n, n, n, lr, lc);
      wave_sync();
      mm_acc_ta<BS>(Rr, W, LDW, W + NP, LDW, n, lr, lc);  // + M' S
      {
        double P1[BS][BS];
        blk_zero<BS>(P1);
        mm_acc<BS, true>(P1, W + NP, LDW, Ts, LDW, n, lr, lc);  // S T'
        wave_sync();
        blk_store_lds<BS>(P1, W, LDW, lr, lc);
      }
      wave_sync();
      mm_acc_ta<BS>(Rr, Tk, LD, W, LDW, n, lr, lc);       // + C' P1
      double tmax;
      {
        double t0[BS][BS];
        blk_load_global<BS>(t0, T_bar + off, n, n, n, lr, lc);
        tmax = blk_maxabs<BS>(t0);
      }
      const double rmax = blk_maxabs<BS>(Rr);
      if (rmax <= ADJ_FP_RESIDUAL * tmax)
        ok = adj_stein_solve<BS>(W, Tk, B, C, T, off, n, Rr, Sb, lane, gmax);  // the correction dS (also in W's second group)
      else  // the first pass broke down or is far off: no power of G can be trusted
        ok = adj_stein_fixed_point<BS>(W, Tk, smem + AdjSmem<BS>::bytes / sizeof(double), B, C, T, T_bar, A_bar, off, n, Sb, lane);
    } else if constexpr (FUSED) {
      const int k = fa.k;
      const size_t offk = (size_t)draw * n * k;
      if (status[draw] != 0) {  // failed draw: zero cotangents
        double z[BS][BS];
        blk_zero<BS>(z);
        blk_store_global<BS>(z, A_bar + off, n, n, n, lr, lc);
        blk_store_global<BS>(z, B_bar + off, n, n, n, lr, lc);
        blk_store_global<BS>(z, C_bar + off, n, n, n, lr, lc);
        blk_store_global<BS>(z, fa.D_bar + offk, n, k, k, lr, lc);
        if (fa.q_bar && lane < k) fa.q_bar[(size_t)draw * k + lane] = 0.0;
        continue;
      }
      // ---- GR = Gbar R;  Rbar = 2 GR diag(q);  qbar_j = sum_i R_ij GR_ij  (grad_assemble_kernel's expressions; W is free) ----
      double Rbar[BS][BS];
      {
        wave_sync();
        lds_load_matrix(W, LDW, NP, NP, fa.Gbar + off, n, n, lane);
        lds_load_matrix(W + NP, LDW, NP, NP, fa.R + offk, n, k, lane);
        wave_sync();
        bool nzG = false;
        if (lane < n)
          for (int r = 0; r < n; ++r) nzG = nzG | (W[r * LDW + lane] != 0.0);
        const unsigned long long cmG = __ballot(nzG);
        const int g_lo = cmG ? __ffsll((long long)cmG) - 1 : 0, g_hi = cmG ? 64 - __clzll((long long)cmG) : 0;
        double GR[BS][BS], Rb[BS][BS];
        blk_zero<BS>(GR);
        mm_acc<BS, false>(GR, W + g_lo, LDW, W + NP + g_lo * LDW, LDW, g_hi - g_lo, lr, lc);
        blk_load_lds<BS>(Rb, W + NP, LDW, lr, lc);
        const double* qd = fa.q + (fa.q_batched ? (size_t)draw * k : 0);
#pragma unroll
        for (int j = 0; j < BS; ++j) {
          const int c = lc * BS + j;
          const double qj = (c < k) ? qd[c] : 0.0;
          double colsum = 0.0;
#pragma unroll
          for (int i = 0; i < BS; ++i) {
            Rbar[i][j] = 2.0 * GR[i][j] * qj;
            colsum = fma(Rb[i][j], GR[i][j], colsum);
          }
          colsum += shfl_xor_f64(colsum, 8);
          colsum += shfl_xor_f64(colsum, 16);
          colsum += shfl_xor_f64(colsum, 32);
          if (lr == 0 && c < k) fa.q_bar[(size_t)draw * k + c] = colsum;
        }
      }
      double Hb[BS][BS];
      blk_load_global<BS>(Hb, T_bar + off, n, n, n, lr, lc);
      bool took = false;
      AdjFuse<BS> fz{Rbar, fa.R + offk, k, fa.D_bar + offk, 0};
      if constexpr (AdjCompact<BS>::enabled) ok = adj_stein_solve_compact<BS>(W, Tk, B, C, T, off, n, Hb, Sb, lane, gmax, took, kt_cols, &fz);
      if (!took || !ok || gmax > ADJ_REFINE_GROWTH || refine_mode == 1) {  // (wave-uniform) the two-kernel path takes this draw
        if (lane == 0) status[draw] |= DSGE_ST_INTERNAL_RERUN;
        continue;
      }
      // ---- A_bar = S;  B_bar = -X R' + S T';  C_bar = B_bar T' ------------------------------------------------------------------
      blk_store_global<BS>(Sb, A_bar + off, n, n, n, lr, lc);
      wave_sync();
      lds_load_matrix(Tk, LD, NP, 8 * ((k + 7) / 8), fa.R + offk, n, k, lane);
      wave_sync();
      double Bb[BS][BS], Cb[BS][BS];
      blk_zero<BS>(Bb);
      mm_acc<BS, true>(Bb, W + 2 * NP + fz.x_col, LDW, Tk, LD, k, lr, lc);  // X R'
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) Bb[i][j] = -Bb[i][j];
      wave_sync();  // (X has been consumed by every lane: T takes the third column group)
      lds_load_matrix(Ts, LDW, NP, NP, T + off, n, n, lane);
      wave_sync();
      const int kt_f = (kt_cols > 0 && kt_cols <= n) ? kt_cols : n;
      mm_acc<BS, true>(Bb, W + NP, LDW, Ts, LDW, kt_f, lr, lc);  // + S T'
      blk_store_global<BS>(Bb, B_bar + off, n, n, n, lr, lc);
      blk_store_lds<BS>(Bb, W, LDW, lr, lc);
      wave_sync();
      blk_zero<BS>(Cb);
      mm_acc<BS, true>(Cb, W, LDW, Ts, LDW, kt_f, lr, lc);  // B_bar T'
      blk_store_global<BS>(Cb, C_bar + off, n, n, n, lr, lc);
      continue;
    } else {
      if (BS <= 5 && only_flag) {  // (the second half of the two-kernel path behind a fused launch: the draws that launch left; the
                                   //  fused launch is used up to 40 variables)
        if (!(status[draw] & DSGE_ST_INTERNAL_RERUN)) continue;
        wave_sync();
        if (lane == 0) status[draw] &= ~DSGE_ST_INTERNAL_RERUN;
        wave_sync();
      }
      double Hb[BS][BS];
      blk_load_global<BS>(Hb, T_bar + off, n, n, n, lr, lc);
      bool took = false;
      if constexpr (AdjCompact<BS>::enabled) ok = adj_stein_solve_compact<BS>(W, Tk, B, C, T, off, n, Hb, Sb, lane, gmax, took, kt_cols);
      if (!took) ok = adj_stein_solve<BS>(W, Tk, B, C, T, off, n, Hb, Sb, lane, gmax);
      // (debug hook: refine_mode 1 = every draw, 2 = none).  A solve that broke down -- the computed powers of G exploded --
      // leaves zeros and goes to the second pass as well, which then takes its elimination-based fall-back
      flag = refine_mode ? refine_mode == 1 : (!ok || gmax > ADJ_REFINE_GROWTH);
      if (!ok && flag) {
        blk_zero<BS>(Sb);
        wave_sync();
        blk_store_lds<BS>(Sb, W + NP, LDW, lr, lc);
        wave_sync();
        ok = true;
      }
    }
    const bool acc_out = REFINE || accumulate;
    if constexpr (REFINE) {
      double t0[BS][BS];
      blk_load_global<BS>(t0, A_bar + off, n, n, n, lr, lc);
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) t0[i][j] += Sb[i][j];
      blk_store_global<BS>(t0, A_bar + off, n, n, n, lr, lc);
    } else {
      blk_store_global<BS>(Sb, A_bar + off, n, n, n, lr, lc);
    }
    wave_sync();
    lds_load_matrix(Ts, LDW, NP, NP, T + off, n, n, lane);  // T again, over the dead G_k
    wave_sync();
    // (the products with T' contract over the columns of T: up to its last non-zero one when the compact solve has found it)
    const int kt_hi = (kt_cols > 0 && kt_cols <= n) ? kt_cols : n;
    {
      double Bb[BS][BS], Cb[BS][BS];
      blk_zero<BS>(Bb);
      mm_acc<BS, true>(Bb, W + NP, LDW, Ts, LDW, kt_hi, lr, lc);  // S T'  (second pass: dS T')
      if (acc_out) {  // gradient pipeline: B_bar, C_bar already hold the cotangents that came through R
        double t0[BS][BS];
        blk_load_global<BS>(t0, B_bar + off, n, n, n, lr, lc);
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) t0[i][j] += Bb[i][j];
        blk_store_global<BS>(t0, B_bar + off, n, n, n, lr, lc);
      } else {
        blk_store_global<BS>(Bb, B_bar + off, n, n, n, lr, lc);
      }
      wave_sync();
      blk_store_lds<BS>(Bb, W, LDW, lr, lc);
      wave_sync();
      blk_zero<BS>(Cb);
      mm_acc<BS, true>(Cb, W, LDW, Ts, LDW, kt_hi, lr, lc);  // S T' T'
      if (acc_out) {
        double t0[BS][BS];
        blk_load_global<BS>(t0, C_bar + off, n, n, n, lr, lc);
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) Cb[i][j] += t0[i][j];
      }
      blk_store_global<BS>(Cb, C_bar + off, n, n, n, lr, lc);
    }
    if constexpr (REFINE) {
      if (lane == 0) status[draw] = (st_in & ~DSGE_ST_INTERNAL_RERUN) | (ok ? 0 : DSGE_ST_NOT_CONVERGED);
    } else if (accumulate) {
      if (lane == 0 && (!ok || flag)) status[draw] |= (ok ? 0 : DSGE_ST_NOT_CONVERGED) | (flag ? DSGE_ST_INTERNAL_RERUN : 0);
    } else if (lane == 0) {
      status[draw] = (ok ? DSGE_ST_OK : DSGE_ST_NOT_CONVERGED) | (flag ? DSGE_ST_INTERNAL_RERUN : 0);
    }
  }
}

// ---------------------------------------------------------------------------------------
// gEcon recursion residual norms (diagnostics of DSGEStateSpace.build_statespace_graph,
// gEconpy/model/statespace.py:1181-1204).  With the state mask s (variables that appear at t-1 and at
// t), M = diag(s):
//   deterministic_norm = || (A + B T + C T M T)[:, s] ||_F     (A' + B R' + C R' P)
//   stochastic_norm    = || B R + C T M R + D ||_F              (B S' + C R' Q + D)
// ---------------------------------------------------------------------------------------
template <int BS>
struct NormSmem {
  static constexpr int NP = Tile<BS>::NP, LD = Tile<BS>::LD;
  static constexpr size_t bytes = sizeof(double) * (size_t)(5 * NP * LD);
};

template <int BS>
__global__ __launch_bounds__(64) void norms_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                    const double* __restrict__ C, const double* __restrict__ D,
                                                    const double* __restrict__ T, const double* __restrict__ R,
                                                    const int32_t* __restrict__ state_mask, int batch, int n, int k,
                                                    double* __restrict__ det_out, double* __restrict__ sto_out) {
  constexpr int NP = NormSmem<BS>::NP, LD = NormSmem<BS>::LD;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Ts = smem;
  double* TMs = Ts + NP * LD;   // T with the non-state columns zeroed
  double* Ws = TMs + NP * LD;   // T M T, then T M R
  double* Ls = Ws + NP * LD;    // left operand (B, then C)
  double* Rs = Ls + NP * LD;    // R padded to NP columns
  const int lane = threadIdx.x, lr = lane >> 3, lc = lane & 7;
  bool colmask[BS];
#pragma unroll
  for (int j = 0; j < BS; ++j) {
    const int c = lc * BS + j;
    colmask[j] = (c < n) && (state_mask[c] != 0);
  }
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    const size_t off = (size_t)draw * n * n, offk = (size_t)draw * n * k;
    wave_sync();
    lds_load_matrix(Ts, LD, NP, NP, T + off, n, n, lane);
    lds_load_matrix(Rs, LD, NP, NP, R + offk, n, k, lane);
    lds_load_matrix(Ls, LD, NP, NP, B + off, n, n, lane);
    {
      double t[BS][BS];
      blk_load_global<BS>(t, T + off, n, n, n, lr, lc);
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) t[i][j] = colmask[j] ? t[i][j] : 0.0;
      blk_store_lds<BS>(t, TMs, LD, lr, lc);
    }
    wave_sync();
    double W1[BS][BS], W2[BS][BS], E1[BS][BS], E2[BS][BS];
    blk_zero<BS>(W1);
    blk_zero<BS>(W2);
    mm_acc<BS, false>(W1, TMs, LD, Ts, LD, n, lr, lc);  // T M T
    mm_acc<BS, false>(W2, TMs, LD, Rs, LD, n, lr, lc);  // T M R
    blk_load_global<BS>(E1, A + off, n, n, n, lr, lc);
    blk_load_global<BS>(E2, D + offk, n, k, k, lr, lc);
    mm_acc<BS, false>(E1, Ls, LD, Ts, LD, n, lr, lc);   // + B T
    mm_acc<BS, false>(E2, Ls, LD, Rs, LD, n, lr, lc);   // + B R
    wave_sync();
    lds_load_matrix(Ls, LD, NP, NP, C + off, n, n, lane);
    blk_store_lds<BS>(W1, Ws, LD, lr, lc);
    wave_sync();
    mm_acc<BS, false>(E1, Ls, LD, Ws, LD, n, lr, lc);   // + C T M T
    wave_sync();
    blk_store_lds<BS>(W2, Ws, LD, lr, lc);
    wave_sync();
    mm_acc<BS, false>(E2, Ls, LD, Ws, LD, n, lr, lc);   // + C T M R
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        if (colmask[j]) s1 = fma(E1[i][j], E1[i][j], s1);
        s2 = fma(E2[i][j], E2[i][j], s2);
      }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
      det_out[draw] = sqrt(s1);
      sto_out[draw] = sqrt(s2);
    }
  }
}

// ---------------------------------------------------------------------------------------
// Kalman filter log-likelihood, "standard" filter (reference call
// site gEconpy/model/statespace.py:1151-1157, a0 = 0 :812).
//
// Update in square-root-free downdate form.  With F = L L' (Cholesky), G = P Zm' L^-T and
// K = G L^-1 = P Zm' F^-1, the Joseph-form covariance of the reference recursion
//     P+ = sym((I-KZm) P (I-KZm)') + sym(K Hm K') + jit_P I
// equals, exactly in real arithmetic (F = Zm P Zm' + Hm + jit_F I),
//     P+ = P - G G' - jit_V K K' + jit_P I          (jit_V = jit_F; jit_V = 0 is the plain form P - K F K' + jit_P I;
// the jitters, the ln 2pi constant and the masking of d are run-time conventions: FilterConv, dsge_device.hpp)
// which costs O(m^2 p) instead of O(m^3).  Verified against the Joseph form of the oracle to
// ~1e-15 relative on logp (tests/test_device_algorithm_model.py).
// Prediction:  a = T a+,  P = sym(T P+ T') + sym(RQR).
// ---------------------------------------------------------------------------------------
template <int BS>
struct KfSmem {
  static constexpr int NP = Tile<BS>::NP, LD = Tile<BS>::LD;
  // Ts, Ps, Ws (NP x LD), Zs (p x LD), PZt, G, K (NP x pld), F (p x pld), vectors; pld = p|1
  __host__ __device__ static constexpr size_t doubles(int p) {
    return (size_t)3 * NP * LD + (size_t)p * LD + (size_t)3 * NP * (p | 1) + (size_t)p * (p | 1) + 2 * NP +
           6 * DSGE_MAX_P;
  }
  static size_t bytes(int p) { return sizeof(double) * doubles(p); }
};

template <int BS>
__global__ __launch_bounds__(64) void kalman_kernel(
    const double* __restrict__ T, const double* __restrict__ RQR, const double* __restrict__ P0,
    const double* __restrict__ Z, int z_batched, const double* __restrict__ dvec, int d_batched,
    const double* __restrict__ Hdiag, int h_batched, const double* __restrict__ y, int batch, int m, int p,
    int T_len, FilterConv cv, double missing_fill, double* __restrict__ logp_out,
    int32_t* __restrict__ status, int rerun_only) {
  constexpr int NP = KfSmem<BS>::NP, LD = KfSmem<BS>::LD, PMAX = DSGE_MAX_P;
  const int PLD = p | 1;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Ts = smem;
  double* Ps = Ts + NP * LD;
  double* Ws = Ps + NP * LD;
  double* Zs = Ws + NP * LD;
  double* PZt = Zs + p * LD;
  double* Gs = PZt + NP * PLD;
  double* Ks = Gs + NP * PLD;
  double* Fs = Ks + NP * PLD;
  double* av = Fs + p * PLD;  // predicted state a (NP)
  double* af = av + NP;          // filtered state a+ (NP)
  double* vv = af + NP;          // innovation v (PMAX)
  double* wv = vv + PMAX;        // L^-1 v
  double* ww = wv + PMAX;        // 1/0 observation weights
  double* ds = ww + PMAX;        // obs intercept
  double* hs = ds + PMAX;        // diag(H)
  double* ys = hs + PMAX;        // masked y_t
  const int lane = threadIdx.x, lr = lane >> 3, lc = lane & 7;
  const double LN2PI = 1.8378770664093453;

  if (rerun_only && rerun_pass_is_empty(status, batch)) return;
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    if (rerun_only) {
      // second pass after kalman_sel_kernel: only the draws it could not handle
      const int32_t st_in = status[draw];
      if (!(st_in & DSGE_ST_INTERNAL_RERUN)) continue;
      wave_sync();
      if (lane == 0) status[draw] = st_in & ~DSGE_ST_INTERNAL_RERUN;
      if (st_in != DSGE_ST_INTERNAL_RERUN) {  // a failure was recorded on the way here (e.g. Lyapunov)
        if (lane == 0) logp_out[draw] = -INFINITY;
        continue;
      }
    } else if (status && status[draw] != 0) {
      if (lane == 0) logp_out[draw] = -INFINITY;
      continue;
    }
    const size_t off = (size_t)draw * m * m;
    wave_sync();
    for (int idx = lane; idx < (int)KfSmem<BS>::doubles(p); idx += 64) smem[idx] = 0.0;
    wave_sync();
    lds_load_matrix(Ts, LD, NP, NP, T + off, m, m, lane);
    lds_load_matrix(Ps, LD, NP, NP, P0 + off, m, m, lane);
    lds_load_matrix(Zs, LD, p, NP, Z + (z_batched ? (size_t)draw * p * m : 0), p, m, lane);
    if (lane < PMAX) {
      ds[lane] = (dvec && lane < p) ? dvec[(d_batched ? (size_t)draw * p : 0) + lane] : 0.0;
      hs[lane] = (Hdiag && lane < p) ? Hdiag[(h_batched ? (size_t)draw * p : 0) + lane] : 0.0;
    }
    for (int i = lane; i < NP; i += 64) av[i] = 0.0;
    double Qb[BS][BS], Pb[BS][BS];
    blk_load_global<BS>(Qb, RQR + off, m, m, m, lr, lc);
    blk_load_global<BS>(Pb, P0 + off, m, m, m, lr, lc);
    wave_sync();

    double ll_sum = 0.0, ll_comp = 0.0;  // Kahan-compensated sum of ll_t
    for (int t = 0; t < T_len; ++t) {
      // ---- missing-data mask (handle_missing_values): weights, masked y
      int n_obs = 0;
      if (lane < PMAX) {
        double yt = (lane < p) ? y[(size_t)t * p + lane] : 0.0;
        const bool miss = (lane >= p) || (yt != yt) || (yt == missing_fill);
        ww[lane] = miss ? 0.0 : 1.0;
        ys[lane] = miss ? 0.0 : yt;
      }
      {
        double yt = (lane < p) ? y[(size_t)t * p + lane] : missing_fill;
        const bool obs = (lane < p) && (yt == yt) && (yt != missing_fill);
        n_obs = __popcll(__ballot(obs));
      }
      wave_sync();
      // ---- PZt[i][o] = w_o sum_k P[i][k] Z[o][k];  v_o = y_o - (d_o + w_o Z[o] a)
      for (int idx = lane; idx < m * p; idx += 64) {
        const int i = idx / p, o = idx - i * p;
        double s = 0.0;
        for (int kk = 0; kk < m; ++kk) s = fma(Ps[i * LD + kk], Zs[o * LD + kk], s);
        PZt[i * PLD + o] = ww[o] * s;
      }
      if (lane < p) {
        double s = 0.0;
        for (int kk = 0; kk < m; ++kk) s = fma(Zs[lane * LD + kk], av[kk], s);
        vv[lane] = ys[lane] - (((ww[lane] != 0.0 || !cv.mask_d) ? ds[lane] : 0.0) + ww[lane] * s);
      }
      wave_sync();
      // ---- F = Zm PZt + Hm + jit_F I
      for (int idx = lane; idx < p * p; idx += 64) {
        const int o = idx / p, q = idx - o * p;
        double s = 0.0;
        for (int kk = 0; kk < m; ++kk) s = fma(Zs[o * LD + kk], PZt[kk * PLD + q], s);
        s *= ww[o];
        if (o == q) s += ww[o] * hs[o] + cv.jit_F;
        Fs[o * PLD + q] = s;
      }
      wave_sync();
      // ---- Cholesky F = L L' (lower, in place), column by column
      for (int j = 0; j < p; ++j) {
        if (lane == 0) {
          double s = Fs[j * PLD + j];
          for (int q = 0; q < j; ++q) s = fma(-Fs[j * PLD + q], Fs[j * PLD + q], s);
          Fs[j * PLD + j] = sqrt(s);
        }
        wave_sync();
        const int i = j + 1 + lane;
        if (i < p) {
          double s = Fs[i * PLD + j];
          for (int q = 0; q < j; ++q) s = fma(-Fs[i * PLD + q], Fs[j * PLD + q], s);
          Fs[i * PLD + j] = s / Fs[j * PLD + j];
        }
        wave_sync();
      }
      // ---- w = L^-1 v (lane 0), G = PZt L^-T and K = G L^-1 (one row per lane)
      if (lane == 0) {
        for (int o = 0; o < p; ++o) {
          double s = vv[o];
          for (int q = 0; q < o; ++q) s = fma(-Fs[o * PLD + q], wv[q], s);
          wv[o] = s / Fs[o * PLD + o];
        }
      }
      for (int i = lane; i < m; i += 64) {
        // forward substitution: L g = PZt[i,:]'
        for (int o = 0; o < p; ++o) {
          double s = PZt[i * PLD + o];
          for (int q = 0; q < o; ++q) s = fma(-Fs[o * PLD + q], Gs[i * PLD + q], s);
          Gs[i * PLD + o] = s / Fs[o * PLD + o];
        }
        // back substitution: L' k = g
        for (int o = p - 1; o >= 0; --o) {
          double s = Gs[i * PLD + o];
          for (int q = o + 1; q < p; ++q) s = fma(-Fs[q * PLD + o], Ks[i * PLD + q], s);
          Ks[i * PLD + o] = s / Fs[o * PLD + o];
        }
      }
      wave_sync();
      // ---- log-likelihood of the step
      {
        double logdet = 0.0, inner = 0.0;
        for (int o = 0; o < p; ++o) {
          logdet += log(Fs[o * PLD + o]);
          inner = fma(wv[o], wv[o], inner);
        }
        const double ll = (n_obs == 0) ? 0.0 : -0.5 * (cv.ll_terms_step(n_obs, p) * LN2PI + 2.0 * logdet + inner);
        const double yk = ll - ll_comp;
        const double tk = ll_sum + yk;
        ll_comp = (tk - ll_sum) - yk;
        ll_sum = tk;
      }
      // ---- a+ = a + G w
      for (int i = lane; i < m; i += 64) {
        double s = av[i];
        for (int o = 0; o < p; ++o) s = fma(Gs[i * PLD + o], wv[o], s);
        af[i] = s;
      }
      // ---- P+ = P - G G' - jit_V K K' + jit_P I   (register blocks)
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) {
          const int r = lr * BS + i, c = lc * BS + j;
          double gg = 0.0, kk2 = 0.0;
          for (int o = 0; o < p; ++o) {
            gg = fma(Gs[r * PLD + o], Gs[c * PLD + o], gg);
            kk2 = fma(Ks[r * PLD + o], Ks[c * PLD + o], kk2);
          }
          double v = Pb[i][j] - gg - cv.jit_V * kk2;
          if (r == c && r < m) v += cv.jit_P;
          Pb[i][j] = v;
        }
      wave_sync();
      blk_store_lds<BS>(Pb, Ps, LD, lr, lc);  // P+ as the operand of the prediction
      wave_sync();
      // ---- predict: a = T a+;  W = P+ T';  X = T W;  P = sym(X) + RQR
      for (int i = lane; i < m; i += 64) {
        double s = 0.0;
        for (int kk = 0; kk < m; ++kk) s = fma(Ts[i * LD + kk], af[kk], s);
        av[i] = s;
      }
      double Wb[BS][BS];
      blk_zero<BS>(Wb);
      mm_acc<BS, true>(Wb, Ps, LD, Ts, LD, m, lr, lc);
      blk_store_lds<BS>(Wb, Ws, LD, lr, lc);
      wave_sync();
      blk_zero<BS>(Pb);
      mm_acc<BS, false>(Pb, Ts, LD, Ws, LD, m, lr, lc);
      blk_store_lds<BS>(Pb, Ps, LD, lr, lc);  // X (P+ no longer needed)
      wave_sync();
      {
        double Xt[BS][BS];
        blk_load_lds_t<BS>(Xt, Ps, LD, lr, lc);
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) Pb[i][j] = 0.5 * (Pb[i][j] + Xt[i][j]) + Qb[i][j];
      }
      wave_sync();
      blk_store_lds<BS>(Pb, Ps, LD, lr, lc);
      // the wave_sync at the top of the next step orders this store before its readers
    }
    if (lane == 0) {
      const bool finite = (ll_sum == ll_sum) && (fabs(ll_sum) < 1.797e308);
      logp_out[draw] = ll_sum;
      if (!finite && status) status[draw] |= DSGE_ST_FILTER_NONFINITE;
    }
  }
}

// DSGE_SOLVER_FLAG_ZERO_T_ON_FAILURE: park the solver's failure bits of a draw (it carries T = 0) so that the assembly and the
// filter treat it as an ordinary draw; `restore` = 1 ORs them back into the status after the filter.
template <int BLOCK>  // (a template only so that the header can be included by several translation units)
__global__ __launch_bounds__(BLOCK) void status_park_kernel(int32_t* __restrict__ status, int32_t* __restrict__ park, int batch, int restore) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch) return;
  if (restore) {
    status[i] |= park[i];
  } else {
    const int32_t s = status[i], f = s & (DSGE_ST_NOT_CONVERGED | DSGE_ST_NAN);
    park[i] = f;
    if (f && (s & ~f) == 0) status[i] = 0;
  }
}

}  // namespace dsge

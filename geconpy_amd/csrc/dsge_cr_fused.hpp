// Static-variable deflation, cycle reduction on the reduced system and the back-substitution of the static rows in ONE
// launch ("cr_fused_kernel"): the three phases of dsge_cr_deflate.hpp / dsge_cr_compact.hpp with the reduced system handed
// from phase to phase through LDS instead of HBM.
//
//   phase 1  Householder QR of the static columns of B applied to [B_st | B_dy | A_dy | C_dy | D], one column per lane in
//            registers (crd_qr_chunk).  The h top rows go to a per-draw scratch record in global memory (8.5 KB at
//            n = 40, h = 10; read back in phase 3 by the same workgroup, i.e. from L2); the columns of the REDUCED system are
//            written straight into the compact kernel's LDS layout W = [B_dy | A_dy[:,S] C_dy[:,L]] -- the zero columns
//            of A_dy and C_dy are found by a ballot over the column registers, not by a pass over global memory.
//   phase 2  crc_iterate (the loop of cr_compact_kernel, unchanged) and the final solve  [T_dy[:,S] | R_dy] =
//            -A1_hat^-1 [A_dy[:,S] | D_red]; the right-hand side of that solve is the one thing that has to survive the
//            iteration outside the register blocks: it is parked in global scratch (nd x (s + k) doubles, 6 KB).
//   phase 3  back-substitution of the static rows (crd_inflate_prepare / crd_inflate_chunk) with the columns of
//            [T_dy | R_dy] read from LDS, scatter to the caller's variable order.
//
// Against the three launches (SW-shaped 40-variable system, 10 static, 4096 draws): no reduced A, B, C, D, T_dy, R_dy in HBM
// (131 MB written and read back twice), two launches and their tails less.  Draws the fused kernel cannot take -- fewer
// static variables than the bound h, s + l or s + k beyond the reduced tile, a numerically singular R_st -- are flagged
// DSGE_ST_INTERNAL_RERUN and solved by the full-size dense kernel afterwards, exactly as on the three-launch path.
// Limits checked by the launcher: h + 3 nd + k <= 128 with two columns per lane, <= 192 with three (n = 46 .. 64), and
// nd + k <= 64.
#pragma once
#include "dsge_cr_compact.hpp"
#include "dsge_cr_deflate.hpp"

namespace dsge {

template <int BSF, int BSD>
struct CrfSmem {
  static constexpr int NMF = 8 * BSF, NPD = 8 * BSD, LDW = CrcSmem<BSD>::LDW, HM = CRD_HMAX;
  // doubles of the compact kernel's layout (W, Lbuf, Ybuf); phase 1 aliases V (HM x NMF) and phase 3 the inflate arrays
  static constexpr size_t dbl_compact = (size_t)(NPD * LDW + NPD * BSD + BSD * 2 * NPD);
  static constexpr size_t dbl_qr = (size_t)(HM * NMF + HM);
  static constexpr size_t dbl_inflate = CrdInflateLds<NPD>::doubles;
  static constexpr size_t dbl =
      dbl_compact > dbl_qr ? (dbl_compact > dbl_inflate ? dbl_compact : dbl_inflate) : (dbl_qr > dbl_inflate ? dbl_qr : dbl_inflate);
  // index tables: prow (NPD ints, shared with gauss_jordan_blocked); cmap, posS, posL, rsrc (NPD), dyi (64), sti (HM) as
  // bytes (all values < 64): with 32-bit tables the (5, 4) instance needs 20.9 KB -- 7 draws per CU instead of 8
  typedef signed char idx_t;
  static constexpr size_t bytes = sizeof(double) * dbl + sizeof(int) * NPD + ((4 * NPD + 64 + HM + 15) & ~(size_t)15);
};

// per-draw global scratch: top block (crd_top_doubles) and the right-hand side of the final solve, NPD columns of nd
// doubles (column-major; columns [A_dy[:,S] | D_red], s + k <= NPD of them are written)
__host__ __device__ inline size_t crf_rhs_doubles(int nd, int npd) { return (size_t)nd * npd; }

// bit (lane + shift) of the result = bit lane of b, for any shift in (-64, 64)
__device__ __forceinline__ unsigned long long crf_place(unsigned long long b, int shift) {
  return shift >= 0 ? (shift < 64 ? b << shift : 0ull) : (-shift < 64 ? b >> (-shift) : 0ull);
}

// NC = columns of [B_st | B_dy | A_dy | C_dy | D] per lane: 2 for h + 3 nd + k <= 128, 3 up to 192 (n = 46 .. 64)
template <int BSF, int BSD, int NC>
__device__ __forceinline__ void cr_fused_body(const double* __restrict__ A, const double* __restrict__ B,
                                              const double* __restrict__ C, const double* __restrict__ D, int batch, int n,
                                              int k, int h, int max_iter, double tol, double* __restrict__ top,
                                              double* __restrict__ rhs, double* __restrict__ T_out,
                                              double* __restrict__ R_out, int32_t* __restrict__ status,
                                              int32_t* __restrict__ n_iter_out,
                                              unsigned long long* __restrict__ colmask_out) {
  using SM = CrfSmem<BSF, BSD>;
  constexpr int NMF = SM::NMF, NPD = SM::NPD, LDW = SM::LDW, HM = CRD_HMAX;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* W = smem;
  double* G1 = W + NPD;
  double* Lbuf = W + NPD * LDW;
  double* Ybuf = Lbuf + NPD * BSD;
  typedef typename SM::idx_t idx_t;
  int* prow = (int*)(smem + SM::dbl);
  idx_t* cmap = (idx_t*)(prow + NPD);
  idx_t* posS = cmap + NPD;
  idx_t* posL = posS + NPD;
  idx_t* rsrc = posL + NPD;
  idx_t* dyi = rsrc + NPD;
  idx_t* sti = dyi + 64;
  const int lane = threadIdx.x, lr = lane >> 3, lc = lane & 7;
  const int nd = n - h;
  const int draw = blockIdx.x;  // one draw per workgroup (see cr_deflate_kernel)
  if (draw >= batch) return;
  const size_t off = (size_t)draw * n * n, offk = (size_t)draw * n * k;
  auto hand_over = [&]() {  // the full-size kernel solves this draw
    if (lane == 0) {
      status[draw] = DSGE_ST_INTERNAL_RERUN;
      if (colmask_out) colmask_out[draw] = ~0ull;
    }
  };
  if (colmask_out && lane == 0) colmask_out[draw] = ~0ull;  // "not known" until the solve has succeeded

  // ---- phase 1: deflation ----------------------------------------------------------------------------------------------
  unsigned long long smask = crd_static_mask<NMF>(A, C, off, n, lane);
  if (__popcll(smask) < h) return hand_over();
  smask = crd_first_bits(smask, h);
  crd_index_tables(smask, n, lane, dyi, sti);
  double* tp = top + (size_t)draw * crd_top_doubles(n, k, h);
  double* rh = rhs + (size_t)draw * crf_rhs_doubles(nd, NPD);
  int s, l;
  unsigned long long state_cols = 0ull, maskS_red = 0ull, maskL_red = 0ull;  // (red: over the reduced variables)
  {
    double col[NC][NMF];
    bool act[NC];
    crd_qr_chunk<NMF, NC>(A, B, C, D, off, offk, n, k, h, 0, dyi, sti, /*V=*/smem, tp, lane, col, act,
                          /*skip_zero_ac=*/true);
    // block (0 B, 1 A, 2 C, 3 D, -1 none) and column inside the block of this lane's columns 64 q + lane
    int blk[NC], dcol[NC];
    unsigned long long bS = 0ull, bL = 0ull;  // non-zero columns of A_dy (states) / C_dy (leads), over the reduced variables
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      const int c = 64 * q + lane - h;
      blk[q] = -1;
      dcol[q] = 0;
      if (act[q] && c >= 0) {
        blk[q] = (c >= nd) + (c >= 2 * nd) + (c >= 3 * nd);
        dcol[q] = c - blk[q] * nd;
      }
      bool nz = false;  // (rows >= nd of a column are zeros by now)
#pragma unroll
      for (int r = 0; r < NMF; ++r) nz = nz || (col[q][r] != 0.0);
      bS |= crf_place(__ballot(blk[q] == 1 && nz), 64 * q - (h + nd));
      bL |= crf_place(__ballot(blk[q] == 2 && nz), 64 * q - (h + 2 * nd));
    }
    const unsigned long long ndmask = (nd >= 64) ? ~0ull : ((1ull << nd) - 1ull);
    const unsigned long long maskS = bS & ndmask, maskL = bL & ndmask;
    s = __popcll(maskS);
    l = __popcll(maskL);
    maskS_red = maskS;
    maskL_red = maskL;
    if (s + l > NPD || s + k > NPD) return hand_over();
    {  // the non-zero columns of T, in the caller's variable numbering: bit v <=> v is dynamic and a state
      const unsigned long long below = (1ull << lane) - 1ull;
      const bool dyn = (lane < n) && !((smask >> lane) & 1ull);
      const int dred = lane - __popcll(smask & below);
      state_cols = __ballot(dyn && ((maskS >> (dred & 63)) & 1ull));
    }
    wave_sync();  // every lane is done with V
    for (int idx = lane; idx < NPD * LDW; idx += 64) W[idx] = 0.0;
    if (lane < NPD) {
      const unsigned long long below = (1ull << lane) - 1ull;
      const bool isS = (maskS >> lane) & 1ull, isL = (maskL >> lane) & 1ull;
      const int ps = __popcll(maskS & below), pl = s + __popcll(maskL & below);
      posS[lane] = (idx_t)(isS ? ps : -1);
      posL[lane] = (idx_t)(isL ? pl : -1);
      if (isS) cmap[ps] = (idx_t)lane;
      if (isL) cmap[pl] = (idx_t)lane;
    }
    wave_sync();
    // the columns of the reduced system -> W = [B_dy | A_dy[:,S] C_dy[:,L]] (LDS); [A_dy[:,S] | D_red] -> global scratch
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      const int d = dcol[q];
      const unsigned long long below = (1ull << d) - 1ull;
      int lcol = -1, gcol = -1;  // column of W / of the scratch record
      if (blk[q] == 0) {
        lcol = d;
      } else if (blk[q] == 1) {
        if ((maskS >> d) & 1ull) {
          gcol = __popcll(maskS & below);
          lcol = NPD + gcol;
        }
      } else if (blk[q] == 2) {
        if ((maskL >> d) & 1ull) lcol = NPD + s + __popcll(maskL & below);
      } else if (blk[q] == 3) {
        gcol = s + d;
      }
      if (lcol >= 0) {
        double* dst = W + lcol;
#pragma unroll
        for (int r = 0; r < NMF; ++r)
          if (r < nd) dst[r * LDW] = col[q][r];
      }
      if (gcol >= 0) {  // (column-major: a lane writes its column contiguously, unused columns are never touched)
        double* dst = rh + (size_t)gcol * nd;
#pragma unroll
        for (int r = 0; r < NMF; ++r)
          if (r < nd) dst[r] = col[q][r];
      }
    }
  }
  wave_sync();

  // ---- phase 2: cycle reduction on the reduced system ---------------------------------------------------------------------
  double A1[BSD][BSD], Ah[BSD][BSD], Rb[BSD][BSD];
  blk_load_lds<BSD>(A1, W, LDW, lr, lc);
  blk_load_lds<BSD>(Rb, G1, LDW, lr, lc);
  int vS[BSD], vL[BSD];
#pragma unroll
  for (int j = 0; j < BSD; ++j) {
    vS[j] = posS[lc * BSD + j];
    vL[j] = posL[lc * BSD + j];
  }
#pragma unroll
  for (int i = 0; i < BSD; ++i)
#pragma unroll
    for (int j = 0; j < BSD; ++j) Ah[i][j] = A1[i][j];
  wave_sync();
  bool converged, saw_nan;
  int it;
  crc_iterate<BSD>(A1, Ah, Rb, W, Lbuf, Ybuf, prow, cmap, rsrc, posS, posL, nd, s, l, max_iter, tol, 0, lane, nullptr, it,
                   converged, saw_nan);
  if (lane == 0) {
    status[draw] = converged ? DSGE_ST_OK : (DSGE_ST_NOT_CONVERGED | (saw_nan ? DSGE_ST_NAN : 0));
    if (n_iter_out) n_iter_out[draw] = it;
  }
  if (!converged) {  // zero policy matrices, as the full-size kernels write
    for (int idx = lane; idx < n * n; idx += 64) T_out[off + idx] = 0.0;
    for (int idx = lane; idx < n * k; idx += 64) R_out[offk + idx] = 0.0;
    return;
  }
  // [T_dy[:,S] | R_dy] = -A1_hat^-1 [A_dy[:,S] | D_red]  (cycle_reduction.py:181, shared.py:74-75; see cr_compact_body)
  wave_sync();
  blk_store_lds<BSD>(Ah, W, LDW, lr, lc);
  {
    double t[BSD][BSD];
#pragma unroll
    for (int i = 0; i < BSD; ++i)
#pragma unroll
      for (int j = 0; j < BSD; ++j) {
        const int r = lr * BSD + i, c = lc * BSD + j;
        t[i][j] = (r < nd && c < s + k) ? rh[(size_t)c * nd + r] : 0.0;
      }
    blk_store_lds<BSD>(t, G1, LDW, lr, lc);
  }
  gauss_jordan_blocked<BSD>(W, LDW, nd, 2, Lbuf, Ybuf, prow, lane);
  gj_unpermute<BSD>(W, LDW, nd, 1, 2, prow, lane);

  // ---- phase 3: the static rows, scatter to the caller's variable order -------------------------------------------------
  double y[NPD];  // column `lane` of [T_dy | R_dy]: T_dy[:, v] = -X[:, posS(v)] for a state v, zero otherwise
  {
    const int ntot = nd + k;
    int src = -1;
    if (lane < nd)
      src = posS[lane];
    else if (lane < ntot)
      src = s + (lane - nd);
#pragma unroll
    for (int q = 0; q < NPD; ++q) y[q] = (q < nd && src >= 0) ? -G1[q * LDW + src] : 0.0;
  }
  wave_sync();  // the solution is in registers: W becomes the work space of the back-substitution
  const CrdInflateLds<NPD> L(smem);
  if (!crd_inflate_prepare<NPD>(y, lane < nd + k, tp, n, k, h, lane, L, maskL_red)) return hand_over();
  if (lane < n) {  // static columns of T are exact zeros
#pragma unroll
    for (int s2 = 0; s2 < HM; ++s2)
      if (s2 < h) T_out[off + (size_t)lane * n + sti[s2]] = 0.0;
  }
  crd_inflate_chunk<NPD>(0, y, tp, n, k, h, lane, L, dyi, sti, T_out + off, R_out + offk, maskS_red);
  // every other column of T is exactly zero (written so): the filter kernel need not look for them
  if (colmask_out && lane == 0) colmask_out[draw] = state_cols;
}

template <int BSF, int BSD, int NC = 2>
__global__ __launch_bounds__(64) void cr_fused_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                       const double* __restrict__ C, const double* __restrict__ D, int batch,
                                                       int n, int k, int h, int max_iter, double tol,
                                                       double* __restrict__ top, double* __restrict__ rhs,
                                                       double* __restrict__ T_out, double* __restrict__ R_out,
                                                       int32_t* __restrict__ status, int32_t* __restrict__ n_iter_out,
                                                       unsigned long long* __restrict__ colmask_out) {
  cr_fused_body<BSF, BSD, NC>(A, B, C, D, batch, n, k, h, max_iter, tol, top, rhs, T_out, R_out, status, n_iter_out, colmask_out);
}

// the register budget of two waves per SIMD for the 4 x 4 reduced tile (see cr_compact_kernel_occ2)
template <int BSF, int BSD>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void cr_fused_kernel_occ2(
    const double* __restrict__ A, const double* __restrict__ B, const double* __restrict__ C, const double* __restrict__ D,
    int batch, int n, int k, int h, int max_iter, double tol, double* __restrict__ top, double* __restrict__ rhs,
    double* __restrict__ T_out, double* __restrict__ R_out, int32_t* __restrict__ status, int32_t* __restrict__ n_iter_out,
    unsigned long long* __restrict__ colmask_out) {
  cr_fused_body<BSF, BSD, 2>(A, B, C, D, batch, n, k, h, max_iter, tol, top, rhs, T_out, R_out, status, n_iter_out, colmask_out);
}

}  // namespace dsge

// Wave-wide real Householder reflector on LDS-resident matrices: shared by the gensys window kernels
// (dsge_gensys_win.hpp) and the static-variable deflation in front of cycle reduction (dsge_cr_deflate.hpp).
#pragma once
#include "dsge_device.hpp"

namespace dsge {

// ---- real Householder reflector built from column `col` of `src` (rows j..N-1, one row per lane) and applied from the
// left to rows j..N-1 of H (columns h0..h0+nH-1), T (nT columns) and X (nX columns).  dlarfg conventions as in
// householder_left (dsge_gensys.hpp).  The nH + nT + nX columns are dealt one per lane (in passes of 64); a lane walks
// down its column with a private pointer and stride, four rows per trip and no divergent control flow, so the LDS loads
// of a trip are in flight together (the first version -- one conditional load per matrix and row -- spent 300 cycles
// per row).
__device__ __forceinline__ void hh_left_real(double* Hr, int ldH, int h0, int nH, double* Tr, int ldW, int nT, double* Xr,
                                             int ldX, int nX, double* src, int ld_src, int col, int j, int N, int lane) {
  wave_sync();
  const double x = (lane >= j && lane < N) ? src[lane * ld_src + col] : 0.0;
  const double xnorm2 = wave_sum_dpp((lane > j) ? x * x : 0.0);
  if (xnorm2 == 0.0) return;
  const double alpha = readlane_dyn_f64(x, j);
  const double nrm = sqrt(fma(alpha, alpha, xnorm2));
  const double beta = (alpha >= 0.0) ? -nrm : nrm;
  const double tau = (beta - alpha) / beta;
  const double scal = 1.0 / (alpha - beta);
  const int ncols = nH + nT + nX;
  // column c of the virtual matrix [H | T | X] -> base pointer and row stride (inactive lanes walk a valid column of H
  // and store nothing)
  auto column = [&](int c, double*& base, int& ld) -> bool {
    base = Hr + h0;
    ld = ldH;
    if (c >= ncols) return false;
    if (c < nH) {
      base = Hr + h0 + c;
    } else if (c < nH + nT) {
      base = Tr + (c - nH);
      ld = ldW;
    } else {
      base = Xr + (c - nH - nT);
      ld = ldX;
    }
    return true;
  };
  // Round 4: v is not handed out by v_readlane (two VALU instructions per value and row) but read back from the source column
  // in LDS, every lane the same address (a broadcast), unscaled:  v' m = m_j + scal sum_{r > j} x_r m_r,  m_r += x_r (scal w),
  // m_j += w.  The source column is part of the sweep: its owner updates rows r .. r+3 in the trip that has just read them
  // (LDS operations of a wavefront execute in program order), and it is overwritten with (beta, 0, ...) at the end anyway.
  const double* const px0 = src + (size_t)(j + 1) * ld_src + col;
  for (int c0 = 0; c0 < ncols; c0 += 128) {  // two columns per lane and trip: eight own loads + four broadcast loads in flight
    double *bA, *bB;
    int lA, lB;
    column(c0 + lane, bA, lA);
    column(c0 + 64 + lane, bB, lB);
    const double mjA = bA[j * lA], mjB = bB[j * lB];
    double* pA = bA + (j + 1) * lA;
    double* pB = bB + (j + 1) * lB;
    const double* px = px0;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
    int r = j + 1;
    for (; r + 4 <= N; r += 4) {
      const double m0 = pA[0], m1 = pA[lA], m2 = pA[2 * lA], m3 = pA[3 * lA];
      const double q0 = pB[0], q1 = pB[lB], q2 = pB[2 * lB], q3 = pB[3 * lB];
      const double v0 = px[0], v1 = px[ld_src], v2 = px[2 * ld_src], v3 = px[3 * ld_src];
      a0 = fma(v0, m0, a0);
      a1 = fma(v1, m1, a1);
      a2 = fma(v2, m2, a2);
      a3 = fma(v3, m3, a3);
      b0 = fma(v0, q0, b0);
      b1 = fma(v1, q1, b1);
      b2 = fma(v2, q2, b2);
      b3 = fma(v3, q3, b3);
      pA += 4 * lA;
      pB += 4 * lB;
      px += 4 * ld_src;
    }
    for (; r < N; ++r) {
      const double vr = px[0];
      a0 = fma(vr, pA[0], a0);
      b0 = fma(vr, pB[0], b0);
      pA += lA;
      pB += lB;
      px += ld_src;
    }
    const double wA = -tau * fma(scal, (a0 + a1) + (a2 + a3), mjA), wB = -tau * fma(scal, (b0 + b1) + (b2 + b3), mjB);
    const double wsA = scal * wA, wsB = scal * wB;
    pA = bA + (j + 1) * lA;
    pB = bB + (j + 1) * lB;
    px = px0;
    for (r = j + 1; r + 4 <= N; r += 4) {
      double m0 = pA[0], m1 = pA[lA], m2 = pA[2 * lA], m3 = pA[3 * lA];
      double q0 = pB[0], q1 = pB[lB], q2 = pB[2 * lB], q3 = pB[3 * lB];
      const double v0 = px[0], v1 = px[ld_src], v2 = px[2 * ld_src], v3 = px[3 * ld_src];
      m0 = fma(v0, wsA, m0);
      m1 = fma(v1, wsA, m1);
      m2 = fma(v2, wsA, m2);
      m3 = fma(v3, wsA, m3);
      q0 = fma(v0, wsB, q0);
      q1 = fma(v1, wsB, q1);
      q2 = fma(v2, wsB, q2);
      q3 = fma(v3, wsB, q3);
      // (no store masks: a lane beyond the last column walks column 0 of H, computes what that column's owner computes and
      // stores the same values to the same addresses -- an exec-mask branch less per trip, see gw_realqz_sweeps)
      pA[0] = m0;
      pA[lA] = m1;
      pA[2 * lA] = m2;
      pA[3 * lA] = m3;
      pB[0] = q0;
      pB[lB] = q1;
      pB[2 * lB] = q2;
      pB[3 * lB] = q3;
      pA += 4 * lA;
      pB += 4 * lB;
      px += 4 * ld_src;
    }
    for (; r < N; ++r) {
      const double vr = px[0];
      const double m0 = fma(vr, wsA, pA[0]), q0 = fma(vr, wsB, pB[0]);
      pA[0] = m0;
      pB[0] = q0;
      pA += lA;
      pB += lB;
      px += ld_src;
    }
    bA[j * lA] = mjA + wA;  // row j last: when this lane's column IS the source column the rows below were read unscaled above
    bB[j * lB] = mjB + wB;
  }
  wave_sync();
  if (lane >= j && lane < N) src[lane * ld_src + col] = (lane == j) ? beta : 0.0;
  wave_sync();
}

// ---- the same reflector on NW wavefronts of one workgroup (gensys_reduce_kernel, round 4): every wavefront builds the reflector
// for itself from column `col` of `src` (one row per lane: identical arithmetic, no exchange), the nH + nT + nX columns are
// dealt out in NW contiguous shares, one column per lane.  `skip` is the index of the source column inside [H | T | X] (or -1):
// it is left alone by the sweep -- other wavefronts may still be reading it -- and set to (beta, 0, ..., 0) by wavefront 0
// behind the barrier.  Two barriers per reflector.
template <int NW>
__device__ __forceinline__ void hh_left_real_mw(double* Hr, int ldH, int h0, int nH, double* Tr, int ldW, int nT, double* Xr,
                                                int ldX, int nX, double* src, int ld_src, int col, int j, int N, int skip,
                                                int lane, int wv) {
  __syncthreads();
  const double x = (lane >= j && lane < N) ? src[lane * ld_src + col] : 0.0;
  const double xnorm2 = wave_sum_dpp((lane > j) ? x * x : 0.0);
  if (xnorm2 == 0.0) return;  // (the same decision in every wavefront)
  const double alpha = readlane_dyn_f64(x, j);
  const double nrm = sqrt(fma(alpha, alpha, xnorm2));
  const double beta = (alpha >= 0.0) ? -nrm : nrm;
  const double tau = (beta - alpha) / beta;
  const double scal = 1.0 / (alpha - beta);
  const int ncols = nH + nT + nX;
  const int per = (ncols + NW - 1) / NW;
  const int c = wv * per + lane;
  const bool mine = lane < per && c < ncols && c != skip;
  double* bA = Hr + h0;
  int lA = ldH;
  if (mine) {
    if (c < nH) {
      bA = Hr + h0 + c;
    } else if (c < nH + nT) {
      bA = Tr + (c - nH);
      lA = ldW;
    } else {
      bA = Xr + (c - nH - nT);
      lA = ldX;
    }
  }
  // The walk down the rows, eight per trip.  v is NOT handed out by v_readlane (two VALU instructions per value and lane-step:
  // they were most of this routine's instruction stream): the source column itself -- which the sweep leaves alone -- is read
  // back from LDS, every lane the same address (a broadcast, conflict-free), unscaled:
  //   v' m = m_j + scal * sum_{r > j} x_r m_r,      m_r += v_r w = m_r + x_r (scal w)  (r > j),   m_j += w.
  const double* px = src + (size_t)(j + 1) * ld_src + col;
  double* pA = bA + (j + 1) * lA;
  const double mj = bA[j * lA];
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int r = j + 1;
  for (; r + 8 <= N; r += 8) {
    double mr[8], xr[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      mr[u] = pA[u * lA];
      xr[u] = px[u * ld_src];
    }
#pragma unroll
    for (int u = 0; u < 8; u += 4) {
      a0 = fma(xr[u], mr[u], a0);
      a1 = fma(xr[u + 1], mr[u + 1], a1);
      a2 = fma(xr[u + 2], mr[u + 2], a2);
      a3 = fma(xr[u + 3], mr[u + 3], a3);
    }
    pA += 8 * lA;
    px += 8 * ld_src;
  }
  for (; r < N; ++r) {
    a0 = fma(px[0], pA[0], a0);
    pA += lA;
    px += ld_src;
  }
  const double wA = -tau * fma(scal, (a0 + a1) + (a2 + a3), mj);
  const double ws = scal * wA;
  if (mine) bA[j * lA] = mj + wA;
  pA = bA + (j + 1) * lA;
  px = src + (size_t)(j + 1) * ld_src + col;
  for (r = j + 1; r + 8 <= N; r += 8) {
    double mr[8], xr[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      mr[u] = pA[u * lA];
      xr[u] = px[u * ld_src];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) mr[u] = fma(xr[u], ws, mr[u]);
    if (mine) {
#pragma unroll
      for (int u = 0; u < 8; ++u) pA[u * lA] = mr[u];
    }
    pA += 8 * lA;
    px += 8 * ld_src;
  }
  for (; r < N; ++r) {
    const double m0 = fma(px[0], ws, pA[0]);
    if (mine) pA[0] = m0;
    pA += lA;
    px += ld_src;
  }
  __syncthreads();
  if (wv == 0 && lane >= j && lane < N) src[lane * ld_src + col] = (lane == j) ? beta : 0.0;
}

}  // namespace dsge

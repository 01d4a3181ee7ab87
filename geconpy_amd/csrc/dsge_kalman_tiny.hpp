// Kalman-filter log-likelihood for SMALL models: one THREAD per draw, everything in registers.
//
// The wave-per-draw kernels keep a 64-lane wavefront busy with one draw; for the reference's own small
// test models (rbc_linearized.gcn: 8 variables of which 2 are states and 1-3 are observed) the exactly
// reduced filter has u = |S u O| <= 6 variables and a wavefront would idle on 3 x 3 matrices.  Here lane l of
// a wave owns draw 64 b + l: T, G = sym(R Q R') and P restricted to U live in VGPRs as fully unrolled
// U x U arrays, the data y_t are wave-uniform (scalar loads), and there is no LDS traffic and no fence at all.
//
// Same mathematics as kalman_sel_kernel (dsge_kalman2.hpp): exact reduction to U (ordered OBSERVED variables
// first, in observation order, so that the selector picks a compile-time position), stationary P0 by doubling
// on the reduced model, reference update P+ = P - K (M + jit_V K)' + jit_P I (FilterConv), missing data as upstream, the
// steady-state switch with the same tolerance.  Conditions: selector design matrix, p <= PM, u <= U; a draw
// that violates them is flagged DSGE_ST_INTERNAL_RERUN and taken by the wave-per-draw kernels.
#pragma once
#include "dsge_device.hpp"

#include "../../include/dsge_hip.h"

namespace dsge {

template <int U, int PM>
__global__ __launch_bounds__(64) void kalman_tiny_kernel(
    const double* __restrict__ T, const double* __restrict__ RQR, const double* __restrict__ Z, int z_batched,
    const double* __restrict__ dvec, int d_batched, const double* __restrict__ Hdiag, int h_batched,
    const double* __restrict__ y, int batch, int m_full, int p, int T_len, FilterConv cv, double missing_fill,
    double steady_tol, double* __restrict__ logp_out, int32_t* __restrict__ status, int32_t* __restrict__ steady_at) {
  const int draw = blockIdx.x * 64 + threadIdx.x;
  if (draw >= batch) return;
  const int32_t st_in = status[draw];
  if (st_in != 0) {
    logp_out[draw] = -INFINITY;
    return;
  }
  const size_t off = (size_t)draw * m_full * m_full;
  const double* Td = T + off;
  const double* Gd = RQR + off;
  const double* Zg = Z + (z_batched ? (size_t)draw * p * m_full : 0);
  // ---- structure: state columns of T, selected variable of every observation ---------------------
  unsigned long long colmask = 0ull;
  for (int c = 0; c < m_full; ++c) {
    bool nz = false;
    for (int r = 0; r < m_full; ++r) nz = nz || (Td[(size_t)r * m_full + c] != 0.0);
    if (nz) colmask |= 1ull << c;
  }
  bool ok = (p <= PM);
  unsigned long long obsmask = 0ull;
  int ovar[PM];
  double zval[PM];
#pragma unroll
  for (int o = 0; o < PM; ++o) {
    ovar[o] = 0;
    zval[o] = 0.0;
    if (o < p) {
      int cnt = 0;
      for (int c = 0; c < m_full; ++c) {
        const double zl = Zg[(size_t)o * m_full + c];
        if (zl != 0.0) {
          ++cnt;
          ovar[o] = c;
          zval[o] = zl;
        }
      }
      if (cnt != 1 || ((obsmask >> ovar[o]) & 1ull)) ok = false;
      obsmask |= 1ull << ovar[o];
    }
  }
  unsigned long long rest = colmask & ~obsmask;  // states that are not observed
  const int u = __popcll(obsmask) + __popcll(rest);
  ok = ok && (u <= U);
  if (!ok) {
    status[draw] = DSGE_ST_INTERNAL_RERUN;
    return;
  }
  // reduced ordering: observed variables first (observation order), then the remaining states
  int perm[U];
#pragma unroll
  for (int i = 0; i < U; ++i) {
    if (i < PM && i < p) {
      perm[i] = ovar[i];
    } else if (rest) {
      perm[i] = __ffsll((long long)rest) - 1;
      rest &= rest - 1;
    } else {
      perm[i] = -1;
    }
  }
  // p < PM: positions p..PM-1 were filled from `rest` above (i >= p), which is what we want
  double Tc[U][U], G[U][U], P[U][U];
#pragma unroll
  for (int i = 0; i < U; ++i)
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const bool in = perm[i] >= 0 && perm[j] >= 0;
      const size_t g = in ? (size_t)perm[i] * m_full + perm[j] : 0;
      Tc[i][j] = in ? Td[g] : 0.0;
      G[i][j] = in ? Gd[g] : 0.0;
      P[i][j] = G[i][j];
    }
  double dd[PM], hh[PM];
#pragma unroll
  for (int o = 0; o < PM; ++o) {
    dd[o] = (dvec && o < p) ? dvec[(d_batched ? (size_t)draw * p : 0) + o] : 0.0;
    hh[o] = (Hdiag && o < p) ? Hdiag[(h_batched ? (size_t)draw * p : 0) + o] : 0.0;
  }
  // ---- P0 = dlyap(Tu, G) by doubling --------------------------------------------------------------
  {
    double A[U][U];
#pragma unroll
    for (int i = 0; i < U; ++i)
#pragma unroll
      for (int j = 0; j < U; ++j) A[i][j] = Tc[i][j];
    bool lyap_ok = false;
    for (int it = 0; it < 64; ++it) {
      double W[U][U], A2[U][U];
#pragma unroll
      for (int i = 0; i < U; ++i)
#pragma unroll
        for (int j = 0; j < U; ++j) {
          double s0 = 0.0, s1 = 0.0;
#pragma unroll
          for (int k = 0; k < U; ++k) {
            s0 = fma(P[i][k], A[j][k], s0);   // P A'
            s1 = fma(A[i][k], A[k][j], s1);   // A A
          }
          W[i][j] = s0;
          A2[i][j] = s1;
        }
      double dmax = 0.0, pmax = 0.0;
      double X[U][U];
#pragma unroll
      for (int i = 0; i < U; ++i)
#pragma unroll
        for (int j = 0; j < U; ++j) {
          double s0 = 0.0;
#pragma unroll
          for (int k = 0; k < U; ++k) s0 = fma(A[i][k], W[k][j], s0);  // A P A'
          X[i][j] = s0;
        }
#pragma unroll
      for (int i = 0; i < U; ++i)
#pragma unroll
        for (int j = 0; j < U; ++j) {
          const double inc = 0.5 * (X[i][j] + X[j][i]);
          P[i][j] += inc;
          dmax = nanmax(dmax, fabs(inc));
          pmax = nanmax(pmax, fabs(P[i][j]));
          A[i][j] = A2[i][j];
        }
      if (!(dmax == dmax) || !(pmax < 1e300)) break;
      if (dmax <= 1e-17 * pmax) {
        lyap_ok = true;
        break;
      }
    }
    if (!lyap_ok) {
      status[draw] = DSGE_ST_LYAP_FAIL;
      logp_out[draw] = -INFINITY;
      return;
    }
  }
  // ---- filter ------------------------------------------------------------------------------------
  const double LN2PI = 1.8378770664093453, LN2 = 0.6931471805599453;
  double a[U];
#pragma unroll
  for (int i = 0; i < U; ++i) a[i] = 0.0;
  double quad_sum = 0.0, quad_comp = 0.0, ld_mant = 1.0;
  long long ld_exp = 0, n_ll = 0, n_entries = 0;
  double K[U][PM], Fi[PM][PM];
  double step_mant = 1.0;
  int step_exp = 0;
  bool steady = false;
  unsigned steady_mask = 0u;
  int steady_step = -1;
  double Pf_prev[U][U];  // previous filtered covariance (steady-state test)
#pragma unroll
  for (int i = 0; i < U; ++i)
#pragma unroll
    for (int j = 0; j < U; ++j) Pf_prev[i][j] = 0.0;
#pragma unroll
  for (int i = 0; i < U; ++i)
#pragma unroll
    for (int o = 0; o < PM; ++o) K[i][o] = 0.0;
#pragma unroll
  for (int o = 0; o < PM; ++o)
#pragma unroll
    for (int q = 0; q < PM; ++q) Fi[o][q] = 0.0;
  for (int t = 0; t < T_len; ++t) {
    double yv[PM], w[PM];
    unsigned omask = 0u;
#pragma unroll
    for (int o = 0; o < PM; ++o) {
      const double yt = (o < p) ? y[(size_t)t * p + o] : 0.0;  // wave-uniform address: scalar load
      const bool obs = (o < p) && (yt == yt) && (yt != missing_fill);
      w[o] = obs ? 1.0 : 0.0;
      yv[o] = obs ? yt : 0.0;
      omask |= obs ? (1u << o) : 0u;
    }
    const bool any_obs = omask != 0u;
    double v[PM];
#pragma unroll
    for (int o = 0; o < PM; ++o) v[o] = (o < p) ? yv[o] - (((w[o] != 0.0 || !cv.mask_d) ? dd[o] : 0.0) + w[o] * zval[o] * a[o]) : 0.0;
    if (!(steady && omask == steady_mask)) {
      steady = false;
      // M = P Zm' (columns 0..p-1 of P scaled), F = Zm M + Hm + jit I
      double M[U][PM], F[PM][PM];
#pragma unroll
      for (int i = 0; i < U; ++i)
#pragma unroll
        for (int o = 0; o < PM; ++o) M[i][o] = w[o] * zval[o] * P[i][o];
#pragma unroll
      for (int o = 0; o < PM; ++o)
#pragma unroll
        for (int q = 0; q < PM; ++q) {
          if (o < p && q < p)
            F[o][q] = w[o] * zval[o] * M[o][q] + ((o == q) ? (w[o] * hh[o] + cv.jit_F) : 0.0);
          else
            F[o][q] = (o == q) ? 1.0 : 0.0;
        }
      // Finv by Gauss-Jordan (SPD, no pivoting), det F as mantissa * 2^exponent
      step_mant = 1.0;
      step_exp = 0;
#pragma unroll
      for (int j = 0; j < PM; ++j) {
        if (j < p) {
          const double piv = F[j][j];
          const double inv = 1.0 / piv;
          int e;
          step_mant *= frexp(piv, &e);
          step_exp += e;
          double rowj[PM];
#pragma unroll
          for (int q = 0; q < PM; ++q) rowj[q] = F[j][q];
#pragma unroll
          for (int o = 0; o < PM; ++o) {
            if (o == j) continue;
            const double ci = F[o][j] * inv;
#pragma unroll
            for (int q = 0; q < PM; ++q) F[o][q] = (q == j) ? -ci : fma(-ci, rowj[q], F[o][q]);
          }
#pragma unroll
          for (int q = 0; q < PM; ++q) F[j][q] = (q == j) ? inv : rowj[q] * inv;
        }
      }
#pragma unroll
      for (int o = 0; o < PM; ++o)
#pragma unroll
        for (int q = 0; q < PM; ++q) Fi[o][q] = F[o][q];
      // K = M Finv;  P+ = P - K (M + jit K)' + jit I
#pragma unroll
      for (int i = 0; i < U; ++i)
#pragma unroll
        for (int o = 0; o < PM; ++o) {
          double s0 = 0.0;
#pragma unroll
          for (int q = 0; q < PM; ++q) s0 = fma(M[i][q], Fi[q][o], s0);
          K[i][o] = (o < p) ? s0 : 0.0;
        }
      double dmax = 0.0, pscale = 0.0;
#pragma unroll
      for (int i = 0; i < U; ++i)
#pragma unroll
        for (int j = 0; j < U; ++j) {
          pscale = nanmax(pscale, fabs(P[i][j]));
          double s0 = P[i][j];
#pragma unroll
          for (int o = 0; o < PM; ++o) s0 = fma(-K[i][o], fma(cv.jit_V, K[j][o], M[j][o]), s0);
          s0 += (i == j && i < u) ? cv.jit_P : 0.0;
          dmax = nanmax(dmax, fabs(s0 - Pf_prev[i][j]));
          Pf_prev[i][j] = s0;
        }
      // predict: P = sym(T P+ T') + G
      {
        double W[U][U];
#pragma unroll
        for (int i = 0; i < U; ++i)
#pragma unroll
          for (int j = 0; j < U; ++j) {
            double s0 = 0.0;
#pragma unroll
            for (int k = 0; k < U; ++k) s0 = fma(Pf_prev[i][k], Tc[j][k], s0);  // P+ T'
            W[i][j] = s0;
          }
        double X[U][U];
#pragma unroll
        for (int i = 0; i < U; ++i)
#pragma unroll
          for (int j = 0; j < U; ++j) {
            double s0 = 0.0;
#pragma unroll
            for (int k = 0; k < U; ++k) s0 = fma(Tc[i][k], W[k][j], s0);
            X[i][j] = s0;
          }
#pragma unroll
        for (int i = 0; i < U; ++i)
#pragma unroll
          for (int j = 0; j < U; ++j) P[i][j] = 0.5 * (X[i][j] + X[j][i]) + G[i][j];
      }
      if (steady_tol > 0.0 && t > 0 && dmax <= steady_tol * pscale) {
        steady = true;
        steady_mask = omask;
        if (steady_step < 0) steady_step = t + 1;
      }
    }
    // likelihood contribution and mean recursion (K, Finv, det F of the last covariance update)
    if (any_obs) {
      double qd = 0.0;
#pragma unroll
      for (int o = 0; o < PM; ++o)
#pragma unroll
        for (int q = 0; q < PM; ++q) qd = fma(Fi[o][q] * v[o], v[q], qd);
      const double yk = qd - quad_comp;
      const double tk = quad_sum + yk;
      quad_comp = (tk - quad_sum) - yk;
      quad_sum = tk;
      int e;
      ld_mant = frexp(ld_mant * step_mant, &e);
      ld_exp += (long long)e + step_exp;
      ++n_ll;
      n_entries += __popc(omask);
    }
    double ap[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
      double s0 = a[i];
#pragma unroll
      for (int o = 0; o < PM; ++o) s0 = fma(K[i][o], v[o], s0);
      ap[i] = s0;
    }
#pragma unroll
    for (int i = 0; i < U; ++i) {
      double s0 = 0.0;
#pragma unroll
      for (int k = 0; k < U; ++k) s0 = fma(Tc[i][k], ap[k], s0);
      a[i] = s0;
    }
  }
  const double logdet = log(ld_mant) + (double)ld_exp * LN2;
  const double ll = -0.5 * (cv.ll_terms(n_ll, n_entries, p) * LN2PI + logdet + quad_sum);
  logp_out[draw] = ll;
  if (steady_at) steady_at[draw] = steady_step;
  if (!((ll == ll) && (fabs(ll) < 1.797e308))) status[draw] = DSGE_ST_FILTER_NONFINITE;
}

}  // namespace dsge

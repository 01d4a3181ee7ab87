// Filter OUTPUTS of the "standard" Kalman filter, per time step: what `save_kalman_filter_outputs_in_idata` asks
// PyMCStateSpace for (gEconpy/model/statespace.py:1145, 1151-1157: filtered / predicted states and covariances, the
// log-likelihood of every observation) -- the scalar log-likelihood of the estimation loop comes from the fast kernels
// (dsge_kalman_nt.hpp, dsge_kalman2.hpp); this kernel serves the post-estimation pass and favours clarity: one workgroup of
// 256 threads per draw, T, P in LDS, the full recursion of oracle/statespace.py (no steady-state switch, no state reduction),
// same rank-p update algebra as the fast kernels:
//     v = ym - d - Zm a;  F = Zm P Zm' + Hm + jit_F I;  K = P Zm' F^-1;  a+ = a + K v
//     P+ = P - sym(K (P Zm' + jit_V K)') + jit_P I;    ll_t = -1/2 (c ln 2 pi + ln det F + v' F^-1 v)  (0 if all missing)
//     (jit_F, jit_P, jit_V, c and the masking of d: the run-time conventions of FilterConv, dsge_device.hpp)
//     a = T a+;  P = sym(T P+ T') + sym(R Q R')
// Outputs per draw and step t: ll[t]; a_pred[t] = a_{t|t-1}, a_filt[t] = a_{t|t}; the diagonals of P_{t|t-1} and P_{t|t}
// (or, on request, the full matrices).
#pragma once
#include "dsge_device.hpp"

namespace dsge {

constexpr int KO_THREADS = 256, KO_PMAX = 16;

struct KoArgs {
  const double* T;       // [batch][m][m]
  const double* RQR;     // [batch][m][m] sym(R Q R')
  const double* P0;      // [batch][m][m]
  const double* Z;       // [p][m] or [batch][p][m]
  const double* d;       // nullptr, [p] or [batch][p]
  const double* Hdiag;   // nullptr, [p] or [batch][p]
  const double* y;       // [T_len][p]
  double* ll;            // [batch][T_len]
  double* a_pred;        // [batch][T_len][m] or nullptr
  double* a_filt;
  double* p_pred;        // [batch][T_len][m] (diagonals) or [batch][T_len][m][m] (full_cov) or nullptr
  double* p_filt;
  int32_t* status;       // [batch] in/out
  int batch, m, p, T_len, z_batched, d_batched, h_batched, full_cov;
  FilterConv cv;         // third-party conventions of the step (dsge_device.hpp)
  double missing_fill;
};

__host__ __device__ inline size_t ko_lds_doubles(int m, int p) {
  return (size_t)3 * m * m + (size_t)p * m + 2 * (size_t)m * KO_PMAX + 2 * (size_t)m + 2 * KO_PMAX * KO_PMAX + 4 * KO_PMAX + 64;
}

__global__ __launch_bounds__(KO_THREADS) void kalman_outputs_kernel(KoArgs a) {
  constexpr int NT = KO_THREADS, PM = KO_PMAX;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, draw = blockIdx.x, m = a.m, p = a.p, mm = m * m;
  if (draw >= a.batch) return;
  double* ll_o = a.ll + (size_t)draw * a.T_len;
  if (a.status[draw] != 0) {  // failed solve upstream: no filter -- EVERY requested output of the draw is NaN (like a -inf logp)
    for (int t = tid; t < a.T_len; t += NT) ll_o[t] = NAN;
    const size_t nv = (size_t)a.T_len * m, nc = a.full_cov ? nv * m : nv;
    if (a.a_pred) for (size_t i = tid; i < nv; i += NT) a.a_pred[(size_t)draw * nv + i] = NAN;
    if (a.a_filt) for (size_t i = tid; i < nv; i += NT) a.a_filt[(size_t)draw * nv + i] = NAN;
    if (a.p_pred) for (size_t i = tid; i < nc; i += NT) a.p_pred[(size_t)draw * nc + i] = NAN;
    if (a.p_filt) for (size_t i = tid; i < nc; i += NT) a.p_filt[(size_t)draw * nc + i] = NAN;
    return;
  }
  double* pl = smem;
  double* Tm = pl; pl += mm;
  double* P = pl; pl += mm;
  double* W = pl; pl += mm;
  double* Zm = pl; pl += (size_t)p * m;
  double* PZ = pl; pl += (size_t)m * PM;   // [i][o]
  double* Kg = pl; pl += (size_t)m * PM;
  double* av = pl; pl += m;
  double* af = pl; pl += m;
  double* Fm = pl; pl += PM * PM;
  double* Lc = pl; pl += PM * PM;
  double* vv = pl; pl += PM;
  double* dv = pl; pl += PM;
  double* hv = pl; pl += PM;
  double* wv = pl; pl += PM;
  int* imask = (int*)pl;
  const double* Tg = a.T + (size_t)draw * mm;
  const double* Gg = a.RQR + (size_t)draw * mm;
  const double* Zg = a.Z + (a.z_batched ? (size_t)draw * p * m : 0);
  for (int idx = tid; idx < mm; idx += NT) {
    Tm[idx] = Tg[idx];
    P[idx] = a.P0[(size_t)draw * mm + idx];
  }
  for (int idx = tid; idx < p * m; idx += NT) Zm[idx] = Zg[idx];
  for (int i = tid; i < m; i += NT) av[i] = 0.0;
  if (tid < p) {
    dv[tid] = a.d ? a.d[(a.d_batched ? (size_t)draw * p : 0) + tid] : 0.0;
    hv[tid] = a.Hdiag ? a.Hdiag[(a.h_batched ? (size_t)draw * p : 0) + tid] : 0.0;
  }
  __syncthreads();
  const double LN2PI = 1.8378770664093453;
  bool finite = true;
  for (int t = 0; t < a.T_len; ++t) {
    if (tid == 0) {
      int mk = 0;
      for (int o = 0; o < p; ++o) {
        const double yo = a.y[(size_t)t * p + o];
        if (!(yo != yo) && yo != a.missing_fill) mk |= 1 << o;
      }
      imask[0] = mk;
    }
    __syncthreads();
    const int mask = imask[0];
    // predicted state / covariance of this step
    const size_t ot = ((size_t)draw * a.T_len + t) * m;
    if (a.a_pred)
      for (int i = tid; i < m; i += NT) a.a_pred[ot + i] = av[i];
    if (a.p_pred) {
      if (a.full_cov)
        for (int idx = tid; idx < mm; idx += NT) a.p_pred[ot * m + idx] = P[idx];
      else
        for (int i = tid; i < m; i += NT) a.p_pred[ot + i] = P[i * m + i];
    }
    // v, P Zm'
    if (tid < p) {
      const int o = tid;
      double za = 0.0;
      if ((mask >> o) & 1)
        for (int j = 0; j < m; ++j) za = fma(Zm[o * m + j], av[j], za);
      vv[o] = (((mask >> o) & 1) ? a.y[(size_t)t * p + o] : 0.0) - ((((mask >> o) & 1) || !a.cv.mask_d) ? dv[o] : 0.0) - za;
    }
    for (int idx = tid; idx < m * p; idx += NT) {
      const int i = idx / p, o = idx - i * p;
      double acc = 0.0;
      if ((mask >> o) & 1)
        for (int j = 0; j < m; ++j) acc = fma(P[i * m + j], Zm[o * m + j], acc);
      PZ[i * PM + o] = acc;
    }
    __syncthreads();
    if (tid < p * p) {
      const int o = tid / p, o2 = tid - o * p;
      double acc = 0.0;
      if ((mask >> o) & 1)
        for (int j = 0; j < m; ++j) acc = fma(Zm[o * m + j], PZ[j * PM + o2], acc);
      if (o == o2) acc += (((mask >> o) & 1) ? hv[o] : 0.0) + a.cv.jit_F;
      Fm[o * PM + o2] = acc;
    }
    __syncthreads();
    if (tid == 0) {  // Cholesky of F, ln det F, F^-1 v, ll_t
      double ld = 0.0;
      bool okc = true;
      for (int j = 0; j < p; ++j) {
        double ds = Fm[j * PM + j];
        for (int r = 0; r < j; ++r) ds -= Lc[j * PM + r] * Lc[j * PM + r];
        if (!(ds > 0.0)) okc = false;
        const double dj = sqrt(ds);
        Lc[j * PM + j] = dj;
        ld += 2.0 * log(dj);
        for (int i = j + 1; i < p; ++i) {
          double sv = 0.5 * (Fm[i * PM + j] + Fm[j * PM + i]);
          for (int r = 0; r < j; ++r) sv -= Lc[i * PM + r] * Lc[j * PM + r];
          Lc[i * PM + j] = sv / dj;
        }
      }
      double quad = 0.0;
      for (int o = 0; o < p; ++o) {
        double sv = vv[o];
        for (int r = 0; r < o; ++r) sv -= Lc[o * PM + r] * wv[r];
        wv[o] = sv / Lc[o * PM + o];
        quad = fma(wv[o], wv[o], quad);
      }
      const double l = (mask != 0) ? -0.5 * (a.cv.ll_terms_step(__popc(mask), p) * LN2PI + ld + quad) : 0.0;
      ll_o[t] = l;
      if (!okc || !(l == l)) finite = false;
    }
    __syncthreads();
    for (int i = tid; i < m; i += NT) {  // K row i = F^-1 (P Zm')[i]
      double x[PM];
      for (int o = 0; o < p; ++o) {
        double sv = PZ[i * PM + o];
        for (int r = 0; r < o; ++r) sv -= Lc[o * PM + r] * x[r];
        x[o] = sv / Lc[o * PM + o];
      }
      for (int o = p - 1; o >= 0; --o) {
        double sv = x[o];
        for (int r = o + 1; r < p; ++r) sv -= Lc[r * PM + o] * x[r];
        x[o] = sv / Lc[o * PM + o];
      }
      double acc = av[i];
      for (int o = 0; o < p; ++o) {
        Kg[i * PM + o] = x[o];
        acc = fma(x[o], vv[o], acc);
      }
      af[i] = acc;
    }
    __syncthreads();
    for (int idx = tid; idx < mm; idx += NT) {  // P+ -> W
      const int i = idx / m, j = idx - i * m;
      double acc = 0.0;
      for (int o = 0; o < p; ++o) {
        const double ki = Kg[i * PM + o], kj = Kg[j * PM + o];
        acc = fma(ki, fma(a.cv.jit_V, kj, PZ[j * PM + o]), acc);
        acc = fma(kj, fma(a.cv.jit_V, ki, PZ[i * PM + o]), acc);
      }
      W[idx] = P[idx] - 0.5 * acc + (i == j ? a.cv.jit_P : 0.0);
    }
    __syncthreads();
    if (a.a_filt)
      for (int i = tid; i < m; i += NT) a.a_filt[ot + i] = af[i];
    if (a.p_filt) {
      if (a.full_cov)
        for (int idx = tid; idx < mm; idx += NT) a.p_filt[ot * m + idx] = W[idx];
      else
        for (int i = tid; i < m; i += NT) a.p_filt[ot + i] = W[i * m + i];
    }
    // prediction: a = T a+,  P = sym(T P+ T') + sym(R Q R')
    for (int i = tid; i < m; i += NT) {
      double acc = 0.0;
      for (int j = 0; j < m; ++j) acc = fma(Tm[i * m + j], af[j], acc);
      av[i] = acc;
    }
    for (int idx = tid; idx < mm; idx += NT) {  // P <- P+ T'  (P is free: its last reader was the P+ pass)
      const int i = idx / m, j = idx - i * m;
      double acc = 0.0;
      for (int c = 0; c < m; ++c) acc = fma(W[i * m + c], Tm[j * m + c], acc);
      P[idx] = acc;
    }
    __syncthreads();
    for (int idx = tid; idx < mm; idx += NT) {  // W <- T (P+ T')
      const int i = idx / m, j = idx - i * m;
      double acc = 0.0;
      for (int c = 0; c < m; ++c) acc = fma(Tm[i * m + c], P[c * m + j], acc);
      W[idx] = acc;
    }
    __syncthreads();
    for (int idx = tid; idx < mm; idx += NT) {
      const int i = idx / m, j = idx - i * m;
      P[idx] = 0.5 * (W[idx] + W[j * m + i]) + 0.5 * (Gg[idx] + Gg[(size_t)j * m + i]);
    }
    __syncthreads();
  }
  if (tid == 0 && !finite) a.status[draw] |= DSGE_ST_FILTER_NONFINITE;
}

}  // namespace dsge

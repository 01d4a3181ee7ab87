// The tail of the Kalman log-likelihood once the covariance recursion is frozen AND the missing-data mask no longer
// changes: a linear recursion in the mean alone,
//   v_t = c_t - Zw a_t,   a_{t+1} = Phi a_t + Gam c_t      (c_t = w o y_t - d,  Gam = T K,  Phi = T - Gam Zw).
// kalman_sel_kernel runs it step by step (1.1 k cycles per step: a chain of shuffles and reductions); here two steps are ONE
// matrix-vector product
//     [ R v_t ; R v_{t+1} ; a_{t+2} ] = [ -R Zw      |  R        |  0  ]   [ a_t     ]
//                                       [ -R Zw Phi  | -R Zw Gam |  R  ]   [ c_t     ]
//                                       [  Phi^2     |  Phi Gam  | Gam ]   [ c_{t+1} ]
// with R'R = F^-1 (Cholesky), so that v' F^-1 v is a sum of squares each lane accumulates on its own.  Lane i < m owns row i
// of the state block, lanes 32..47 the 16 innovation rows; a row (m + 16 coefficients) lives in registers, the input vector is
// broadcast by v_readlane: about 140 VALU instructions per pair of steps and no LDS traffic.  The set-up (Gam, Phi, R, the
// rows) is done once per draw.  It is a launch of its own because those rows do not fit next to the 256 registers of
// kalman_sel_kernel (an in-kernel version spilled into the full-step loop and lost more than it won); the hand-off record
// (transition, gain, F^-1, state, partial sums: 11 KB per draw) goes through HBM once.
#pragma once
#include "dsge_device.hpp"
#include "dsge_kalman2.hpp"

#include "../../include/dsge_hip.h"

namespace dsge {

// first step from which the missing-data mask stays the same until the end of the sample (0 if it never changes);
// y is shared by all draws, so one wavefront scans it once per call.  *out must be 0 on entry.
__global__ __launch_bounds__(64) void kalman_mask_scan_kernel(const double* __restrict__ y, int p, int T_len,
                                                               double missing_fill, int32_t* __restrict__ out) {
  const int lane = threadIdx.x;
  for (int t = lane; t + 1 < T_len; t += 64) {
    unsigned m0 = 0u, m1 = 0u;
    for (int o = 0; o < p; ++o) {
      const double a = y[(size_t)t * p + o], b = y[(size_t)(t + 1) * p + o];
      m0 |= ((a == a) && (a != missing_fill)) ? (1u << o) : 0u;
      m1 |= ((b == b) && (b != missing_fill)) ? (1u << o) : 0u;
    }
    if (m0 != m1) atomicMax(out, t + 1);
  }
}

__global__ __launch_bounds__(64) void kalman_tail_kernel(const double* __restrict__ rec_all,
                                                          const int32_t* __restrict__ tail_flag,
                                                          const double* __restrict__ y, int batch, int p, int T_len,
                                                          double missing_fill, double* __restrict__ logp_out,
                                                          int32_t* __restrict__ status, int32_t* __restrict__ steady_at,
                                                          FilterConv cv) {
  constexpr int NP = 32, LDM = 33, PS = 10;
  __shared__ __attribute__((aligned(16))) double Tc[NP * LDM];  // transition
  __shared__ __attribute__((aligned(16))) double Ph[NP * LDM];  // Phi = T - Gam Zw
  __shared__ double Ks[NP * PS], Gm[NP * PS];                   // K, Gam = T K
  __shared__ double Fi[64], Rs[64], Lm[8 * LDM];                // F^-1, its Cholesky factor, -R Zw
  __shared__ double av[NP], af[NP], vv[8], zv[8], dd[8];
  __shared__ int zpos[8];
  const int lane = threadIdx.x, fo = lane >> 3, fq = lane & 7;
  const double LN2PI = 1.8378770664093453, LN2 = 0.6931471805599453;
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    if (tail_flag[draw] != 1) continue;
    const double* rec = rec_all + (size_t)draw * KT_REC;
    const double* sc = rec + KT_SC;
    const int m = (int)sc[0], s = (int)sc[1];
    int t = (int)sc[2];
    const unsigned long long omask = (unsigned long long)sc[3];
    const int n_obs = (int)sc[4];
    const double step_mant = sc[5];
    const int step_exp = (int)sc[6];
    double quad_sum = sc[7], quad_comp = sc[8], ld_mant = sc[9];
    long long ld_exp = (long long)sc[10], n_ll = (long long)sc[11];
    wave_sync();
    for (int idx = lane; idx < NP * NP; idx += 64) Tc[(idx >> 5) * LDM + (idx & 31)] = rec[KT_T + idx];
    for (int idx = lane; idx < NP * 8; idx += 64) Ks[(idx >> 3) * PS + (idx & 7)] = rec[KT_K + idx];
    Fi[lane] = rec[KT_FI + lane];
    if (lane < NP) av[lane] = rec[KT_A + lane];
    if (lane < 8) {
      zv[lane] = rec[KT_ZV + lane];
      dd[lane] = rec[KT_DD + lane];
      zpos[lane] = (int)rec[KT_ZP + lane];
    }
    wave_sync();
    double av_reg = (lane < NP) ? av[lane] : 0.0;
    for (int idx = lane; idx < m * 8; idx += 64) {  // Gam = T K
      const int i = idx >> 3, o = idx & 7;
      double g = 0.0;
      for (int k2 = 0; k2 < s; ++k2) g = fma(Tc[i * LDM + k2], Ks[k2 * PS + o], g);
      Gm[i * PS + o] = g;
    }
    // R = upper Cholesky factor of F^-1, in the 8 x 8 lane grid (lane = fo * 8 + fq)
    double rch = Fi[lane];
    bool chol_ok = true;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const double dj = readlane_f64(rch, j * 9);
      if (!(dj > 0.0)) chol_ok = false;
      const double rd = 1.0 / sqrt(dj);
      const double rowq = __shfl(rch, (j << 3) | fq, 64), rowo = __shfl(rch, (j << 3) | fo, 64);
      if (fo == j)
        rch = rowq * rd;
      else if (fo > j)
        rch = fma(-rowo * (rd * rd), rowq, rch);
    }
    if (fq < fo) rch = 0.0;
    Rs[lane] = rch;
    wave_sync();
    if (chol_ok && t + 2 < T_len) {
      for (int idx = lane; idx < NP * LDM; idx += 64) Ph[idx] = 0.0;
      wave_sync();
      for (int idx = lane; idx < m * m; idx += 64) {  // Phi = T - Gam Zw
        const int i = idx / m, c = idx - i * m;
        double val = (c < s) ? Tc[i * LDM + c] : 0.0;
        for (int o = 0; o < p; ++o)
          if (zpos[o] == c && ((omask >> o) & 1ull)) val = fma(-zv[o], Gm[i * PS + o], val);
        Ph[i * LDM + c] = val;
      }
      for (int idx = lane; idx < 8 * LDM; idx += 64) Lm[idx] = 0.0;
      wave_sync();
      if (lane < 8)
        for (int q = lane; q < p; ++q)
          if ((omask >> q) & 1ull) Lm[lane * LDM + zpos[q]] -= Rs[lane * 8 + q] * zv[q];  // row `lane` of -R Zw
      wave_sync();
      const bool vl = lane >= 32 && lane < 48;
      const int vj = (lane - 32) >> 3, vo = lane & 7;
      double coef[NP + 16];
      {
        double Lw[NP];
#pragma unroll
        for (int k2 = 0; k2 < NP; ++k2)
          Lw[k2] = (k2 < m) ? ((lane < m) ? Ph[lane * LDM + k2] : (vl ? Lm[vo * LDM + k2] : 0.0)) : 0.0;
#pragma unroll
        for (int c = 0; c < NP + 16; ++c) coef[c] = 0.0;
        for (int k2 = 0; k2 < m; ++k2) {  // (row of weights) x [Phi | Gam]
          const double lw = (lane < m) ? Ph[lane * LDM + k2] : (vl ? Lm[vo * LDM + k2] : 0.0);
#pragma unroll
          for (int c = 0; c < NP; ++c) coef[c] = fma(lw, Ph[k2 * LDM + c], coef[c]);
#pragma unroll
          for (int o = 0; o < 8; ++o) coef[NP + o] = fma(lw, Gm[k2 * PS + o], coef[NP + o]);
        }
        if (lane < m) {
#pragma unroll
          for (int o = 0; o < 8; ++o) coef[NP + 8 + o] = Gm[lane * PS + o];
        } else if (vl && vj == 1) {
#pragma unroll
          for (int o = 0; o < 8; ++o) coef[NP + 8 + o] = (o < p) ? Rs[vo * 8 + o] : 0.0;
        } else if (vl) {  // first step of the pair: [-R Zw | R | 0]
#pragma unroll
          for (int c = 0; c < NP; ++c) coef[c] = Lw[c];
#pragma unroll
          for (int o = 0; o < 8; ++o) {
            coef[NP + o] = (o < p) ? Rs[vo * 8 + o] : 0.0;
            coef[NP + 8 + o] = 0.0;
          }
        } else {
#pragma unroll
          for (int c = 0; c < NP + 16; ++c) coef[c] = 0.0;
        }
      }
      // the pair loop: steps t+1, t+2 per trip; y of the NEXT trip is in flight during the current one
      const double c_dd = (vl && vo < p) ? dd[vo] : 0.0;
      const bool c_w = vl && vo < p && ((omask >> vo) & 1ull);
      double yb = (vl && vo < p) ? y[(size_t)(t + 1 + vj) * p + vo] : 0.0;
      double qacc = 0.0;
      while (t + 2 < T_len) {
        const double cval = vl ? ((c_w ? ((yb == yb) ? yb : 0.0) : 0.0) - c_dd) : 0.0;
        yb = (vl && vo < p && t + 3 + vj < T_len) ? y[(size_t)(t + 3 + vj) * p + vo] : 0.0;
        double o0 = 0.0, o1 = 0.0;
#pragma unroll
        for (int c = 0; c < NP; c += 2) {
          if (c < m) {
            o0 = fma(coef[c], readlane_f64(av_reg, c), o0);
            o1 = fma(coef[c + 1], readlane_f64(av_reg, c + 1), o1);
          }
        }
#pragma unroll
        for (int o = 0; o < 16; o += 2) {
          o0 = fma(coef[NP + o], readlane_f64(cval, 32 + o), o0);
          o1 = fma(coef[NP + o + 1], readlane_f64(cval, 33 + o), o1);
        }
        const double outv = o0 + o1;
        if (lane < m) av_reg = outv;
        if (vl && n_obs > 0) qacc = fma(outv, outv, qacc);
        if (n_obs > 0) {
          int e;
          ld_mant = frexp(ld_mant * step_mant, &e);
          ld_exp += (long long)e + step_exp;
          ld_mant = frexp(ld_mant * step_mant, &e);
          ld_exp += (long long)e + step_exp;
          n_ll += 2;
        }
        t += 2;
      }
      const double qs = wave_sum_dpp(qacc);
      const double yk = qs - quad_comp;
      const double tk = quad_sum + yk;
      quad_comp = (tk - quad_sum) - yk;
      quad_sum = tk;
    }
    // ---- whatever is left (one step after an even number of pairs; everything if F^-1 lost positive definiteness to
    // rounding): the plain recursion through LDS
    wave_sync();
    if (lane < NP) av[lane] = (lane < m) ? av_reg : 0.0;
    wave_sync();
    while (t + 1 < T_len) {
      ++t;
      if (lane < 8) {
        double v = 0.0;
        if (lane < p) {
          const double yt = y[(size_t)t * p + lane];
          const bool ob = (omask >> lane) & 1ull;
          v = (ob ? ((yt == yt) ? yt : 0.0) : 0.0) - (dd[lane] + (ob ? 1.0 : 0.0) * zv[lane] * av[zpos[lane]]);
        }
        vv[lane] = v;
      }
      wave_sync();
      double part = 0.0;
      if (lane < 8) {
        double w = 0.0;
        for (int q = 0; q < 8; ++q) w = fma(Fi[lane * 8 + q], vv[q], w);
        part = vv[lane] * w;
      }
      const double qp = wave_sum_dpp(part);
      if (n_obs > 0) {
        const double yk = qp - quad_comp;
        const double tk = quad_sum + yk;
        quad_comp = (tk - quad_sum) - yk;
        quad_sum = tk;
        int e;
        ld_mant = frexp(ld_mant * step_mant, &e);
        ld_exp += (long long)e + step_exp;
        ++n_ll;
      }
      if (lane < m) {
        double a = av[lane];
        for (int o = 0; o < 8; ++o) a = fma(Ks[lane * PS + o], vv[o], a);
        af[lane] = a;
      }
      wave_sync();
      if (lane < m) {
        double a = 0.0;
        for (int k2 = 0; k2 < s; ++k2) a = fma(Tc[lane * LDM + k2], af[k2], a);
        av[lane] = a;
      }
      wave_sync();
    }
    if (lane == 0) {
      const double logdet = log(ld_mant) + (double)ld_exp * LN2;
      // (the mask is constant over the tail: every step of it has n_obs observed entries; d arrives masked if cv.mask_d)
      const long long n_entries = (long long)sc[13] + (n_ll - (long long)sc[11]) * n_obs;
      const double ll = -0.5 * (cv.ll_terms(n_ll, n_entries, p) * LN2PI + logdet + quad_sum);
      logp_out[draw] = ll;
      if (steady_at) steady_at[draw] = (int)sc[12];
      if (!((ll == ll) && (fabs(ll) < 1.797e308))) status[draw] |= DSGE_ST_FILTER_NONFINITE;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// kalman_tail4_kernel (round 5): the same tail, FOUR steps per trip, small enough in LDS to keep every draw of a 4096-draw batch
// resident.  Inside the tail the recursion is affine with constant matrices and data known in advance:
//     a' = L a + TK y',   v = y' - Zm a,   L = Tc - TK Zm,   TK = Tc K,   y' = ym - d,   F^-1 = U'U
//     U v_{t+j} = U y'_{t+j} - U Zm L^j a_t - sum_{i<j} U Zm L^(j-1-i) TK y'_{t+i}          (j = 0..3, 8 entries each)
//     a_{t+4}   = L^4 a_t + sum_j L^(3-j) TK y'_{t+j}
// 32 + m independent affine forms of (a_t, y'_t .. y'_{t+3}): lane 32 + 8 j + o owns entry o of U v_{t+j} (its square IS its share
// of that step's quadratic form: no exchange), lane i < m owns a_{t+4}[i].  Per four steps: ONE broadcast of a_t (2 v_readlane per
// entry) and MC + 32 FMAs per lane -- the single-step loop of the filter kernels issues ~150 instructions per STEP on one dependent
// chain.  Every lane builds its own coefficient row: a row vector r pushed through L up to four times, r TK collected on the way.
// MC = tile width of the filter instance that wrote the records (compile time: the rows are register arrays).
// ---------------------------------------------------------------------------------------------------------------------------
template <int MC>
__global__ __launch_bounds__(64, 2) void kalman_tail4_kernel(const double* __restrict__ rec_all, int32_t* __restrict__ tail_flag,
                                                              const double* __restrict__ y, int batch, int p, int T_len,
                                                              double missing_fill, double* __restrict__ logp_out,
                                                              int32_t* __restrict__ status, int32_t* __restrict__ steady_at,
                                                              FilterConv cv, int m_lo) {
  constexpr int LDL = MC + 2, PS = 10;
  __shared__ __attribute__((aligned(16))) double Ls[MC * LDL];   // L
  __shared__ __attribute__((aligned(16))) double TKs[MC * PS];   // TK
  __shared__ __attribute__((aligned(16))) double Ksh[MC * PS];   // K (set-up only)
  __shared__ __attribute__((aligned(16))) double FinvM[64], Us[64], ymb[64];
  __shared__ double zv[8], dd[8], vv[8], yps[8];
  __shared__ int zpos[8];
  const int lane = threadIdx.x;
  const double LN2PI = 1.8378770664093453, LN2 = 0.6931471805599453;
  const int draw = blockIdx.x;
  if (draw >= batch || tail_flag[draw] != 1) return;
  const double* rec = rec_all + (size_t)draw * KT_REC;
  const double* sc = rec + KT_SC;
  const int m = (int)sc[0], s = (int)sc[1];
  if (m > MC || m <= m_lo) return;  // another instance's draw (wave-uniform): this one takes m_lo < m <= MC
  int t = (int)sc[2];
  const unsigned long long omask = (unsigned long long)sc[3];
  const int n_obs = (int)sc[4];
  const double step_mant = sc[5];
  const int step_exp = (int)sc[6];
  double quad_sum = (lane == 0) ? sc[7] - sc[8] : 0.0, quad_comp = 0.0;  // (per-lane shares from here on; lane 0 carries the past)
  const double ld_mant = sc[9];
  const long long ld_exp0 = (long long)sc[10], n_ll0 = (long long)sc[11];
  for (int idx = lane; idx < MC * 8; idx += 64) Ksh[(idx >> 3) * PS + (idx & 7)] = rec[KT_K + idx];
  FinvM[lane] = rec[KT_FI + lane];
  if (lane < 8) {
    zv[lane] = rec[KT_ZV + lane];
    dd[lane] = rec[KT_DD + lane];  // (arrives masked if the convention says so: the mask is constant over the tail)
    zpos[lane] = (int)rec[KT_ZP + lane];
  }
  double av_reg = (lane < MC) ? rec[KT_A + lane] : 0.0;
  wave_sync();
  for (int idx = lane; idx < m * 8; idx += 64) {  // TK = Tc K[S] (rows of Tc straight from the record: used once)
    const int i = idx >> 3, o = idx & 7;
    double g0 = 0.0, g1 = 0.0;
    for (int k2 = 0; k2 < s; ++k2) {
      const double tv = rec[KT_T + i * 32 + k2];
      if (k2 & 1) g1 = fma(tv, Ksh[k2 * PS + o], g1); else g0 = fma(tv, Ksh[k2 * PS + o], g0);
    }
    TKs[i * PS + o] = g0 + g1;
  }
  wave_sync();
  for (int idx = lane; idx < MC * LDL; idx += 64) {  // L = Tc - TK Zm (row stride LDL; rows / columns >= m zero)
    const int i = idx / LDL, c = idx - i * LDL;
    double val = (i < m && c < s) ? rec[KT_T + i * 32 + c] : 0.0;
    if (i < m)
      for (int o = 0; o < p; ++o)
        if (zpos[o] == c && ((omask >> o) & 1ull)) val = fma(-zv[o], TKs[i * PS + o], val);
    Ls[idx] = (i < m && c < m) ? val : 0.0;
  }
  // U = upper Cholesky factor of F^-1 (block diagonal: observed block, 1 / jit_F for missing entries, 1 for the padding)
  double rch = FinvM[lane];
  bool chol_ok = true;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const double dj = readlane_f64(rch, j * 9);
    if (!(dj > 0.0)) chol_ok = false;
    const double rd = 1.0 / sqrt(dj);
    const double rowq = __shfl(rch, (j << 3) | (lane & 7), 64), rowo = __shfl(rch, (j << 3) | (lane >> 3), 64);
    if ((lane >> 3) == j)
      rch = rowq * rd;
    else if ((lane >> 3) > j)
      rch = fma(-rowo * (rd * rd), rowq, rch);
  }
  if ((lane & 7) < (lane >> 3)) rch = 0.0;
  Us[lane] = rch;
  wave_sync();
  int n_steps = 0;  // steps of the tail processed (each contributes ln det F and, through the lanes' shares, its quadratic form)
  if (chol_ok && n_obs > 0) {
    const bool is_s = lane < m, is_v = lane >= 32;
    const int vj = (lane - 32) >> 3, vo = lane & 7;
    const int my_mults = is_s ? 4 : (is_v ? vj : 0);
    double r[MC], dco[4][8];
#pragma unroll
    for (int x = 0; x < MC; ++x) r[x] = (is_s && x == lane) ? 1.0 : 0.0;
    if (is_v) {  // r = -(U Zm)[vo, :]
      for (int q = vo; q < p; ++q)
        if ((omask >> q) & 1ull) {
          const double uv = -Us[vo * 8 + q] * zv[q];
          const int zq = zpos[q];
#pragma unroll
          for (int x = 0; x < MC; ++x) r[x] = (x == zq) ? r[x] + uv : r[x];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 8; ++q) dco[i][q] = (is_v && i == vj) ? Us[vo * 8 + q] : 0.0;
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
      double g[8];  // g = r TK: the coefficient of y' of an earlier step (v-lanes) / of step 3 - k (state lanes)
#pragma unroll
      for (int q = 0; q < 8; ++q) g[q] = 0.0;
#pragma unroll
      for (int yy = 0; yy < MC; ++yy) {  // (unrolled: r[yy] is a register, the row of TK a wave-uniform address)
        if (yy < m) {
          const double2* tk2 = reinterpret_cast<const double2*>(TKs + yy * PS);
#pragma unroll
          for (int q2 = 0; q2 < 4; ++q2) {
            const double2 tv = tk2[q2];
            g[2 * q2] = fma(r[yy], tv.x, g[2 * q2]);
            g[2 * q2 + 1] = fma(r[yy], tv.y, g[2 * q2 + 1]);
          }
        }
        if ((yy & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      const int slot = is_s ? 3 - k : vj - 1 - k;  // which step's y' this coefficient multiplies
      if (slot >= 0 && (is_s || is_v)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i == slot) {
#pragma unroll
            for (int q = 0; q < 8; ++q) dco[i][q] = g[q];
          }
      }
      if (k < my_mults) {  // r <- r L
        double nr[MC];
#pragma unroll
        for (int x = 0; x < MC; ++x) nr[x] = 0.0;
#pragma unroll
        for (int yy = 0; yy < MC; ++yy) {
          if (yy < m) {
            const double2* l2 = reinterpret_cast<const double2*>(Ls + yy * LDL);
#pragma unroll
            for (int x2 = 0; x2 < MC / 2; ++x2) {
              const double2 lv = l2[x2];
              nr[2 * x2] = fma(r[yy], lv.x, nr[2 * x2]);
              nr[2 * x2 + 1] = fma(r[yy], lv.y, nr[2 * x2 + 1]);
            }
          }
          if ((yy & 1) == 1) __builtin_amdgcn_sched_barrier(0);  // (two rows of L in flight, not all MC: the scheduler otherwise hoists every load)
        }
#pragma unroll
        for (int x = 0; x < MC; ++x) r[x] = nr[x];
      }
    }
    double e0 = 0.0;  // constant of the affine form: -sum dco d
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const double dq = (q < p) ? dd[q] : 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i) e0 = fma(-dco[i][q], dq, e0);
    }
    if (!(is_s || is_v)) e0 = 0.0;
    // ---- the block loop: steps t+1 .. t+4 per trip (the mask is constant from here on: only the count of steps decides);
    //      the data of the next trip are in flight during the current one
    const int bi4 = lane >> 3, bq = lane & 7;  // lanes < 32: entry bq of step bi4 of the block
    const int bqc = (bq < p) ? bq : (p > 0 ? p - 1 : 0);
    const bool b_obs = (lane < 32) && (bq < p) && ((omask >> bq) & 1ull);
    double yb = (lane < 32 && t + 1 + bi4 < T_len) ? y[(size_t)(t + 1 + bi4) * p + bqc] : 0.0;
    while (t + 4 < T_len) {
      double* yw = ymb + ((n_steps >> 2) & 1) * 32;
      if (lane < 32) yw[lane] = (b_obs && yb == yb) ? yb : 0.0;
      {
        const int tn = t + 5 + bi4;
        yb = (lane < 32) ? y[(size_t)(tn < T_len ? tn : T_len - 1) * p + bqc] : 0.0;
      }
      wave_sync();
      double o0 = e0, o1 = 0.0;
#pragma unroll
      for (int x = 0; x < MC; x += 2) {
        o0 = fma(r[x], readlane_f64(av_reg, x), o0);
        o1 = fma(r[x + 1], readlane_f64(av_reg, x + 1), o1);
      }
      const double2* yr = reinterpret_cast<const double2*>(yw);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) {
          const double2 yv2 = yr[i * 4 + q2];
          o0 = fma(dco[i][2 * q2], yv2.x, o0);
          o1 = fma(dco[i][2 * q2 + 1], yv2.y, o1);
        }
      const double outv = o0 + o1;
      if (is_v) {  // (U v)[vo] of step t + 1 + vj: its square is this lane's share of that step's quadratic form
        const double yk = outv * outv - quad_comp;
        const double tk = quad_sum + yk;
        quad_comp = (tk - quad_sum) - yk;
        quad_sum = tk;
      }
      av_reg = is_s ? outv : 0.0;
      t += 4;
      n_steps += 4;
    }
  }
  // ---- whatever is left (at most three steps; everything if F^-1 lost positive definiteness to rounding, or nothing is observed):
  //      a' = L a + TK y',  v = y' - Zm a,  v' F^-1 v, one step at a time
  while (t + 1 < T_len) {
    ++t;
    const bool ob_l = (lane < p) && ((omask >> lane) & 1ull);
    const double yt = (lane < p) ? y[(size_t)t * p + lane] : 0.0;
    const double yp = (lane < p) ? ((ob_l && yt == yt) ? yt : 0.0) - dd[lane] : 0.0;
    const int zp_l = (lane < p) ? zpos[lane] : 0;
    const double a_sel = __shfl(av_reg, zp_l, 64);
    const double v_l = (lane < p) ? yp - (ob_l ? zv[lane] * a_sel : 0.0) : 0.0;
    wave_sync();
    if (lane < 8) {
      vv[lane] = v_l;
      yps[lane] = yp;
    }
    wave_sync();
    if (n_obs > 0) {
      double part = 0.0;
      if (lane < 8) {
        double w = 0.0;
        for (int q = 0; q < 8; ++q) w = fma(FinvM[lane * 8 + q], vv[q], w);
        part = vv[lane] * w;
      }
      const double yk = part - quad_comp;
      const double tk = quad_sum + yk;
      quad_comp = (tk - quad_sum) - yk;
      quad_sum = tk;
      ++n_steps;
    }
    double acc = 0.0;  // a' = L a + TK y': row `lane` of L against the broadcast state
    for (int c = 0; c < m; ++c) acc = fma((lane < m) ? Ls[lane * LDL + c] : 0.0, readlane_dyn_f64(av_reg, c), acc);
    for (int q = 0; q < 8; ++q) acc = fma((lane < m) ? TKs[lane * PS + q] : 0.0, yps[q], acc);
    av_reg = (lane < m) ? acc : 0.0;
  }
  const double quad_total = wave_sum_dpp(quad_sum - quad_comp);
  if (lane == 0) {
    const long long n_ll = n_ll0 + n_steps;
    const long long n_entries = (long long)sc[13] + (long long)n_steps * n_obs;
    const double logdet = log(ld_mant) + (double)ld_exp0 * LN2 + (double)n_steps * (log(step_mant) + (double)step_exp * LN2);
    const double ll = -0.5 * (cv.ll_terms(n_ll, n_entries, p) * LN2PI + logdet + quad_total);
    logp_out[draw] = ll;
    if (steady_at) steady_at[draw] = (int)sc[12];
    if (!((ll == ll) && (fabs(ll) < 1.797e308))) status[draw] |= DSGE_ST_FILTER_NONFINITE;
    tail_flag[draw] = 2;  // done: a later pass of the tail kernel (behind a re-run of flagged draws) leaves this draw alone
  }
}

}  // namespace dsge

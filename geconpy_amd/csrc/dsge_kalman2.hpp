// Kalman filter log-likelihood, structure-exploiting fast path ("kalman_sel_kernel").
//
// Same recursion as kalman_kernel (dsge_kernels.hpp), restricted to
//   * p <= 8 observed series,
//   * a selector design matrix: every row of Z has exactly one non-zero entry, in distinct
//     columns (gEconpy's default `_make_design_matrix` branch, statespace.py:282-296),
// and exploiting two exact structural facts:
//   (1) the columns of T that belong to non-state variables are EXACTLY zero when T comes out
//       of cycle reduction (A has exactly-zero columns there and Gauss-Jordan maps a zero
//       right-hand side to exact zeros), so  T P+ T' = Tc P+[S,S] Tc'  with Tc = T[:,S],
//       S = non-zero columns, s = |S|.  Dropping the zero columns removes only additions of
//       +0.0 and is therefore bit-identical to the dense product;
//   (2) with a selector Z, P Z' is a column gather of P and F = Z P Z' + H is a p x p gather.
// The update uses, with Finv = F^-1 (F = Zm P Zm' + Hm + jitter I), K = P Zm' Finv:
//     P+ = P - K (P Zm' + jitter K)' + jitter I
// which equals the reference's Joseph form  sym((I-KZm) P (I-KZm)') + sym(K Hm K') + jitter I
// exactly in real arithmetic (expand with F = Zm P Zm' + Hm + jitter I).
//
// The hints (s_cap = LDS capacity for the compact state dimension; selector Z) are verified
// per draw on the device.  A draw that violates them is left untouched and flagged with
// DSGE_ST_INTERNAL_RERUN; the host then runs the general kalman_kernel on flagged draws only.
//
// The p x p inverse is a Gauss-Jordan sweep held in registers: lane l owns F[l>>3][l&7];
// pivot row/column travel by ds_bpermute, the pivot by v_readlane.  log det F is the product
// of the pivots, accumulated as (mantissa, exponent) over ALL time steps and turned into a
// logarithm once per draw.
#pragma once
#include "dsge_device.hpp"

#include "../../include/dsge_hip.h"

namespace dsge {

constexpr int32_t DSGE_ST_INTERNAL_RERUN = 1 << 30;

template <int BS>
struct Kf2Smem {
  static constexpr int NP = Tile<BS>::NP, LDM = Tile<BS>::LD;
  __host__ __device__ static constexpr int ldt(int s_cap) { return s_cap | 1; }
  // doubles: Tc NP*LDT, Pc s_cap*LDT, Wc s_cap*LDM, PZt/K/V NP*8 each, Fi 64, av NP, afc NP,
  // vv/dd/hh/zv 8 each; ints: zidx 8 (4 doubles)
  __host__ __device__ static constexpr size_t doubles(int s_cap) {
    return (size_t)NP * ldt(s_cap) + (size_t)s_cap * ldt(s_cap) + (size_t)s_cap * LDM + 3 * (size_t)NP * 8 + 64 +
           2 * NP + 32 + 4;
  }
  static size_t bytes(int s_cap) { return sizeof(double) * doubles(s_cap); }
};

__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, src_lane);
  hi = __builtin_amdgcn_readlane(hi, src_lane);
  return __hiloint2double(hi, lo);
}

template <int BS>
__global__ __launch_bounds__(64) void kalman_sel_kernel(
    const double* __restrict__ T, const double* __restrict__ RQR, const double* __restrict__ P0,
    const double* __restrict__ Z, int z_batched, const double* __restrict__ dvec, int d_batched,
    const double* __restrict__ Hdiag, int h_batched, const double* __restrict__ y, int batch, int m, int p,
    int T_len, int s_cap, double jitter, double missing_fill, double* __restrict__ logp_out,
    int32_t* __restrict__ status, long long* __restrict__ dbg) {
  constexpr int NP = Kf2Smem<BS>::NP, LDM = Kf2Smem<BS>::LDM;
  const int LDT = Kf2Smem<BS>::ldt(s_cap);
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* PZt = smem;                // NP x 8     P Z'   (unmasked)     [16-byte aligned block first]
  double* Ks = PZt + NP * 8;         // NP x 8     K = P Zm' Finv
  double* Vs = Ks + NP * 8;          // NP x 8     P Zm' + jitter K
  double* Fi = Vs + NP * 8;          // 8 x 8      Finv
  double* Tc = Fi + 64;              // NP x LDT   compact transition  T[:, S]
  double* Pc = Tc + NP * LDT;        // s_cap x LDT compact P+[S,S]
  double* Wc = Pc + s_cap * LDT;     // s_cap x LDM W = Pc Tc'
  double* av = Wc + s_cap * LDM;     // NP         predicted state
  double* afc = av + NP;             // NP         filtered state, compacted to S
  double* vv = afc + NP;             // 8 innovation
  double* dd = vv + 8;               // 8 obs intercept
  double* hh = dd + 8;               // 8 diag(H)
  double* zv = hh + 8;               // 8 selector values
  int* zidx = (int*)(zv + 8);        // 8 selector columns
  const int lane = threadIdx.x, lr = lane >> 3, lc = lane & 7;
  const int fo = lane >> 3, fq = lane & 7;  // owner of F[fo][fq]
  const double LN2PI = 1.8378770664093453, LN2 = 0.6931471805599453;

  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    if (status[draw] != 0) {
      if (lane == 0) logp_out[draw] = -INFINITY;
      continue;
    }
    const size_t off = (size_t)draw * m * m;
    wave_sync();
    for (int idx = lane; idx < (int)Kf2Smem<BS>::doubles(s_cap); idx += 64) smem[idx] = 0.0;

    // ---- structure of T: non-zero columns S, rank map ---------------------------------
    double Pb[BS][BS];
    blk_load_global<BS>(Pb, T + off, m, m, m, lr, lc);  // T blocks (reused register file)
    unsigned long long colmask = 0ull;
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      bool nz = false;
#pragma unroll
      for (int i = 0; i < BS; ++i) nz = nz || (Pb[i][j] != 0.0);
      unsigned long long b = __ballot(nz);
      b |= b >> 32;
      b |= b >> 16;
      b |= b >> 8;
#pragma unroll
      for (int g = 0; g < 8; ++g)
        if ((b >> g) & 1ull) colmask |= 1ull << (g * BS + j);
    }
    const int s = __popcll(colmask);
    bool ok = (s <= s_cap);
    // ---- selector structure of Z ---------------------------------------------------------
    const double* Zg = Z + (z_batched ? (size_t)draw * p * m : 0);
    unsigned long long used = 0ull;
    for (int o = 0; o < p; ++o) {
      const double zl = (lane < m) ? Zg[(size_t)o * m + lane] : 0.0;
      const unsigned long long b = __ballot(zl != 0.0);
      if (__popcll(b) != 1 || ((used & b) != 0ull)) ok = false;
      used |= b;
      const int idx = b ? (__ffsll((long long)b) - 1) : 0;
      if (lane == 0) {
        zidx[o] = idx;
        zv[o] = Zg[(size_t)o * m + idx];
      }
    }
    if (!ok) {
      if (lane == 0) status[draw] = DSGE_ST_INTERNAL_RERUN;
      continue;
    }
    wave_sync();
    // per-lane constant maps: rank of my rows / columns in S, observation attached to my columns
    int rr[BS], rc[BS], ocol[BS];
    double zcol[BS];
#pragma unroll
    for (int i = 0; i < BS; ++i) {
      const int r = lr * BS + i, c = lc * BS + i;
      rr[i] = ((colmask >> r) & 1ull) ? __popcll(colmask & ((1ull << r) - 1ull)) : -1;
      rc[i] = ((colmask >> c) & 1ull) ? __popcll(colmask & ((1ull << c) - 1ull)) : -1;
      ocol[i] = -1;
      zcol[i] = 0.0;
      for (int o = 0; o < p; ++o)
        if (zidx[o] == c) {
          ocol[i] = o;
          zcol[i] = zv[o];
        }
    }
    // Tc = T[:, S]
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j)
        if (rc[j] >= 0) Tc[(lr * BS + i) * LDT + rc[j]] = Pb[i][j];
    if (lane < 8) {
      dd[lane] = (dvec && lane < p) ? dvec[(d_batched ? (size_t)draw * p : 0) + lane] : 0.0;
      hh[lane] = (Hdiag && lane < p) ? Hdiag[(h_batched ? (size_t)draw * p : 0) + lane] : 0.0;
    }
    double Qb[BS][BS];
    blk_load_global<BS>(Qb, RQR + off, m, m, m, lr, lc);
    blk_load_global<BS>(Pb, P0 + off, m, m, m, lr, lc);
    // P Z' for the first step
#pragma unroll
    for (int j = 0; j < BS; ++j)
      if (ocol[j] >= 0) {
#pragma unroll
        for (int i = 0; i < BS; ++i) PZt[(lr * BS + i) * 8 + ocol[j]] = zcol[j] * Pb[i][j];
      }
    const int my_zidx = (fo < p) ? zidx[fo] : 0;
    const double my_zv = (fo < p) ? zv[fo] : 0.0;
    wave_sync();

    double quad_sum = 0.0, quad_comp = 0.0;  // Kahan sum of v' Finv v over observed steps
    double ld_mant = 1.0;                    // prod of pivots = mant * 2^exp
    long long ld_exp = 0;
    long long n_ll_steps = 0;
    long long ph[6] = {0, 0, 0, 0, 0, 0};
    for (int t = 0; t < T_len; ++t) {
      long long tk0 = dbg ? clock64() : 0;
      // ---- (a) missing-data mask ------------------------------------------------------
      const double yt = (lane < p) ? y[(size_t)t * p + lane] : 0.0;
      const bool obs = (lane < p) && (yt == yt) && (yt != missing_fill);
      const unsigned long long omask = __ballot(obs);
      const int n_obs = __popcll(omask);
      const double wo = (double)((omask >> fo) & 1ull), wq = (double)((omask >> fq) & 1ull);
      // ---- (b) F[fo][fq] and the innovation -------------------------------------------
      double f;
      if (fo < p && fq < p) {
        f = wo * wq * my_zv * PZt[my_zidx * 8 + fq];
        if (fo == fq) f += wo * hh[fo] + jitter;
      } else {
        f = (fo == fq) ? 1.0 : 0.0;
      }
      if (lane < 8) {
        const double wl = (double)((omask >> lane) & 1ull);
        const double ym = obs ? yt : 0.0;
        vv[lane] = (lane < p) ? ym - (dd[lane] + wl * zv[lane] * av[zidx[lane]]) : 0.0;
      }
      // ---- (c) Finv by in-register Gauss-Jordan (SPD: no pivoting) ----------------------
      double step_mant = 1.0;
      int step_exp = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < p) {
          const double piv = readlane_f64(f, j * 9);
          const double rowj = __shfl(f, (j << 3) | fq, 64);
          const double colj = __shfl(f, (fo << 3) | j, 64);
          const double inv = 1.0 / piv;
          const double ci = colj * inv;
          double nf = fma(-ci, rowj, f);
          nf = (fo == j) ? rowj * inv : nf;
          nf = (fq == j) ? -ci : nf;
          nf = (fo == j && fq == j) ? inv : nf;
          f = nf;
          int e;
          step_mant *= frexp(piv, &e);
          step_exp += e;
        }
      }
      Fi[lane] = f;
      wave_sync();  // #1
      if (dbg) {
        const long long tk1 = clock64();
        ph[0] += tk1 - tk0;
        tk0 = tk1;
      }
      // ---- log-likelihood pieces ------------------------------------------------------
      {
        double qp = f * vv[fo] * vv[fq];
        qp = wave_sum(qp);
        if (n_obs > 0) {
          const double yk = qp - quad_comp;
          const double tk = quad_sum + yk;
          quad_comp = (tk - quad_sum) - yk;
          quad_sum = tk;
          int e;
          ld_mant = frexp(ld_mant * step_mant, &e);
          ld_exp += (long long)e + step_exp;
          ++n_ll_steps;
        }
      }
      // ---- (d) K = (P Zm') Finv, V = P Zm' + jitter K, a+ = a + K v (one state per lane) -
      for (int i = lane; i < m; i += 64) {
        double pz[8], kr[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          pz[q] = ((omask >> q) & 1ull) ? PZt[i * 8 + q] : 0.0;
          kr[q] = 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int o = 0; o < 8; ++o) kr[o] = fma(pz[q], Fi[q * 8 + o], kr[o]);
        double afi = av[i];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
          Ks[i * 8 + o] = kr[o];
          Vs[i * 8 + o] = fma(jitter, kr[o], pz[o]);
          afi = fma(kr[o], vv[o], afi);
        }
        const int ri = ((colmask >> i) & 1ull) ? __popcll(colmask & ((1ull << i) - 1ull)) : -1;
        if (ri >= 0) afc[ri] = afi;
      }
      wave_sync();  // #2
      if (dbg) {
        const long long tk1 = clock64();
        ph[1] += tk1 - tk0;
        tk0 = tk1;
      }
      // ---- (e) P+ = P - K V' + jitter I (register blocks), compact copy to LDS -----------
#pragma unroll
      for (int o2 = 0; o2 < 4; ++o2) {
        double2 ka[BS], vb[BS];
#pragma unroll
        for (int i = 0; i < BS; ++i) ka[i] = *reinterpret_cast<const double2*>(&Ks[(lr * BS + i) * 8 + 2 * o2]);
#pragma unroll
        for (int j = 0; j < BS; ++j) vb[j] = *reinterpret_cast<const double2*>(&Vs[(lc * BS + j) * 8 + 2 * o2]);
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) {
            Pb[i][j] = fma(-ka[i].x, vb[j].x, Pb[i][j]);
            Pb[i][j] = fma(-ka[i].y, vb[j].y, Pb[i][j]);
          }
      }
      if (lr == lc) {
#pragma unroll
        for (int i = 0; i < BS; ++i)
          if (lr * BS + i < m) Pb[i][i] += jitter;
      }
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j)
          if (rr[i] >= 0 && rc[j] >= 0) Pc[rr[i] * LDT + rc[j]] = Pb[i][j];
      wave_sync();  // #3
      if (dbg) {
        const long long tk1 = clock64();
        ph[2] += tk1 - tk0;
        tk0 = tk1;
      }
      // ---- (f) predict: a = Tc a+[S];  W = Pc Tc';  X = Tc W;  P = sym(X) + RQR ---------
      for (int i = lane; i < m; i += 64) {
        double sacc = 0.0;
        for (int kk = 0; kk < s; ++kk) sacc = fma(Tc[i * LDT + kk], afc[kk], sacc);
        av[i] = sacc;
      }
      if (lr * BS < s) {
        double Wb[BS][BS];
        blk_zero<BS>(Wb);
        mm_acc<BS, true>(Wb, Pc, LDT, Tc, LDT, s, lr, lc);
        blk_store_lds<BS>(Wb, Wc, LDM, lr, lc);
      }
      wave_sync();  // #4
      if (dbg) {
        const long long tk1 = clock64();
        ph[3] += tk1 - tk0;
        tk0 = tk1;
      }
      {
        double Xb[BS][BS];
        blk_zero<BS>(Xb);
        mm_acc<BS, false>(Xb, Tc, LDT, Wc, LDM, s, lr, lc);
        const int src = (lc << 3) | lr;  // lane holding the transposed block
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) {
            const double xt = __shfl(Xb[j][i], src, 64);
            Pb[i][j] = 0.5 * (Xb[i][j] + xt) + Qb[i][j];
          }
      }
      // ---- P Z' for the next step ---------------------------------------------------------
#pragma unroll
      for (int j = 0; j < BS; ++j)
        if (ocol[j] >= 0) {
#pragma unroll
          for (int i = 0; i < BS; ++i) PZt[(lr * BS + i) * 8 + ocol[j]] = zcol[j] * Pb[i][j];
        }
      wave_sync();  // #5
      if (dbg) {
        const long long tk1 = clock64();
        ph[4] += tk1 - tk0;
        tk0 = tk1;
      }
    }
    if (dbg && draw == 0 && lane == 0)
      for (int k = 0; k < 5; ++k) dbg[k] = ph[k];
    if (lane == 0) {
      const double logdet = log(ld_mant) + (double)ld_exp * LN2;
      const double ll = -0.5 * ((double)n_ll_steps * (double)p * LN2PI + logdet + quad_sum);
      logp_out[draw] = ll;
      if (!((ll == ll) && (fabs(ll) < 1.797e308))) status[draw] |= DSGE_ST_FILTER_NONFINITE;
    }
  }
}

}  // namespace dsge

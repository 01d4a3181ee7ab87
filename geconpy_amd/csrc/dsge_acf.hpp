// Model-implied autocovariance / autocorrelation matrices, batched over draws.
//
// Replaces, per draw, _compute_autocovariance_matrix (gEconpy/model/statistics/covariance.py:133-161) and the
// graph of DSGEStateSpace.sample_autocorrelation_matrices (gEconpy/model/statespace.py:1262-1300):
//   Sigma = dlyap(T, R Q R')            (computed by the assemble kernel's doubling pass)
//   G_k   = T_step^k Sigma,  T_step = T^lag_step,  k = 0..n_lags          (latent states)
//   G_k   = Z (T_step^k Sigma) Z',  G_0 = Z Sigma Z' + H                   (observed series)
//   out_k = G_k / (std std'),  std = sqrt(diag(G_0))                       (if `correlation`)
// One wavefront per draw; T_step and the running G_k live in LDS, each lag is one register-block product.
#pragma once
#include "dsge_device.hpp"

#include "../../include/dsge_hip.h"

namespace dsge {

template <int BS>
struct AcfSmem {
  static constexpr int NP = Tile<BS>::NP, LD = Tile<BS>::LD;
  // Ts, Gs, scratch W (NP x LD each), Zs (16 x LD), Y (NP x 17), std (NP)
  static constexpr size_t bytes = sizeof(double) * (size_t)(3 * NP * LD + DSGE_MAX_P * LD + NP * (DSGE_MAX_P + 1) + NP);
};

template <int BS>
__global__ __launch_bounds__(64) void acf_kernel(const double* __restrict__ T, const double* __restrict__ Sigma,
                                                  const double* __restrict__ Z, const double* __restrict__ Hdiag,
                                                  int batch, int m, int p, int n_lags, int lag_step, int correlation,
                                                  double* __restrict__ out, const int32_t* __restrict__ status) {
  constexpr int NP = AcfSmem<BS>::NP, LD = AcfSmem<BS>::LD, PL = DSGE_MAX_P + 1;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Ts = smem;             // T_step
  double* Gs = Ts + NP * LD;     // T_step^k Sigma
  double* Ws = Gs + NP * LD;     // scratch (powers of T)
  double* Zs = Ws + NP * LD;     // design matrix (observed mode)
  double* Ys = Zs + DSGE_MAX_P * LD;  // G Z'  (NP x p)
  double* sd = Ys + NP * PL;     // 1/std or 1
  const int lane = threadIdx.x, lr = lane >> 3, lc = lane & 7;
  const bool observed = (Z != nullptr);
  const int dim = observed ? p : m;  // size of one output matrix
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    const size_t off = (size_t)draw * m * m;
    double* o = out + (size_t)draw * (n_lags + 1) * dim * dim;
    if (status && status[draw] != 0) {  // Lyapunov failed (rho(T) >= 1): no stationary distribution
      for (int idx = lane; idx < (n_lags + 1) * dim * dim; idx += 64) o[idx] = __longlong_as_double(0x7ff8000000000000ll);
      continue;
    }
    wave_sync();
    lds_load_matrix(Ts, LD, NP, NP, T + off, m, m, lane);
    lds_load_matrix(Gs, LD, NP, NP, Sigma + off, m, m, lane);
    if (observed) lds_load_matrix(Zs, LD, DSGE_MAX_P, NP, Z, p, m, lane);
    wave_sync();
    // T_step = T^lag_step
    if (lag_step > 1) {
      double Tb[BS][BS];
      blk_load_lds<BS>(Tb, Ts, LD, lr, lc);
      blk_store_lds<BS>(Tb, Ws, LD, lr, lc);  // Ws = T (right factor), Ts accumulates
      wave_sync();
      for (int e = 1; e < lag_step; ++e) {
        double acc[BS][BS];
        blk_zero<BS>(acc);
        mm_acc<BS, false>(acc, Ts, LD, Ws, LD, m, lr, lc);
        wave_sync();
        blk_store_lds<BS>(acc, Ts, LD, lr, lc);
        wave_sync();
      }
    }
    for (int kx = 0; kx <= n_lags; ++kx) {
      if (kx > 0) {  // G <- T_step G
        double acc[BS][BS];
        blk_zero<BS>(acc);
        mm_acc<BS, false>(acc, Ts, LD, Gs, LD, m, lr, lc);
        wave_sync();
        blk_store_lds<BS>(acc, Gs, LD, lr, lc);
        wave_sync();
      }
      double* ok = o + (size_t)kx * dim * dim;
      if (!observed) {
        if (kx == 0) {
          if (lane < NP) sd[lane] = (correlation && lane < m) ? 1.0 / sqrt(Gs[lane * LD + lane]) : 1.0;
          for (int l2 = lane + 64; l2 < NP; l2 += 64) sd[l2] = (correlation && l2 < m) ? 1.0 / sqrt(Gs[l2 * LD + l2]) : 1.0;
          wave_sync();
        }
        for (int idx = lane; idx < m * m; idx += 64) {
          const int r = idx / m, c = idx - r * m;
          ok[idx] = Gs[r * LD + c] * sd[r] * sd[c];  // coalesced store
        }
      } else {
        // Y = G Z' (m x p), then O = Z Y (p x p) [+ H at lag 0]
        for (int idx = lane; idx < m * p; idx += 64) {
          const int i = idx / p, b = idx - i * p;
          double s0 = 0.0;
          for (int j = 0; j < m; ++j) s0 = fma(Gs[i * LD + j], Zs[b * LD + j], s0);
          Ys[i * PL + b] = s0;
        }
        wave_sync();
        double val[4];  // p*p <= 256 entries: at most 4 per lane
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int idx = lane + 64 * q;
          val[q] = 0.0;
          if (idx < p * p) {
            const int a = idx / p, b = idx - a * p;
            double s0 = 0.0;
            for (int i = 0; i < m; ++i) s0 = fma(Zs[a * LD + i], Ys[i * PL + b], s0);
            if (kx == 0 && a == b && Hdiag) s0 += Hdiag[a];
            val[q] = s0;
          }
        }
        if (kx == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int idx = lane + 64 * q;
            if (idx < p * p && idx / p == idx % p) sd[idx / p] = correlation ? 1.0 / sqrt(val[q]) : 1.0;
          }
        }
        wave_sync();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int idx = lane + 64 * q;
          if (idx < p * p) ok[idx] = val[q] * sd[idx / p] * sd[idx % p];
        }
      }
    }
  }
}

}  // namespace dsge

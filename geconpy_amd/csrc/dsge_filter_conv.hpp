// Run-time conventions of the "standard" Kalman filter step, shared by host launchers and device kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace dsge {

// Conventions of the "standard" filter step that belong to third-party code (pymc_extras' StandardFilter, reached from
// gEconpy/model/statespace.py:1143-1157) and cannot be pinned in this image: every one is a RUN-TIME value, passed by value to
// every filter kernel (dsge_options.ll_constant / jitter_F / jitter_P / mask_d / joseph, include/dsge_hip.h), so that matching a real
// install is a configuration change, never a kernel edit.  oracle.FilterConventions holds the same switches.
struct FilterConv {
  double jit_F;  // added to the diagonal of F = Zm P Zm' + Hm
  double jit_P;  // added to the diagonal of the filtered covariance P+
  double jit_V;  // P+ = P - K (P Zm' + jit_V K)' + jit_P I: jit_F with the Joseph form (expand it with K F = P Zm'), 0 for P - K F K'
  int ll_mode;   // constant of ll_t: 0 = p ln 2pi (full observation dimension), 1 = (#observed entries) ln 2pi, 2 = one ln 2pi
  int mask_d;    // non-zero: the observation intercept d is zeroed on missing entries
  // number of ln(2 pi) terms of the whole sample, from the steps with at least one observed entry and their observed entries
  __host__ __device__ double ll_terms(long long n_ll_steps, long long n_obs_entries, int p) const {
    return ll_mode == 0 ? (double)n_ll_steps * (double)p : (ll_mode == 1 ? (double)n_obs_entries : (double)n_ll_steps);
  }
  // the same for one step
  __host__ __device__ double ll_terms_step(int n_obs, int p) const {
    return ll_mode == 0 ? (double)p : (ll_mode == 1 ? (double)n_obs : 1.0);
  }
};

}  // namespace dsge

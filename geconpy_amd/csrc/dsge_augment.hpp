// State augmentation of the solved policy function (DSGEStateSpace.make_symbolic_graph,
// gEconpy/model/statespace.py:781-786): un-permute T, R to the user's variable order
// (_setup_policy_matrices, :217-220), then append the deterministic cumulator / observation-lag chains
//     T_aug = [ T  0 ]      R_aug = [ R ]          (_augment_transition :598-650,
//             [ F  C ]              [ 0 ]           _append_obs_lag_block :652-694, _augment_selection :696-723)
// where every entry of [F C] is a structural 1.0: F copies a model variable into the first slot of its
// chain, C shifts a chain by one slot.  The chains are constants of the model, so they arrive as a list
// of (row, col) "links"; the kernel is pure index arithmetic, one workgroup per draw.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dsge {

__global__ __launch_bounds__(256) void augment_kernel(const double* __restrict__ T, const double* __restrict__ R,
                                                      int batch, int n, int k, int m,
                                                      const int32_t* __restrict__ inv_var_order, int n_links,
                                                      const int32_t* __restrict__ link_rows,
                                                      const int32_t* __restrict__ link_cols,
                                                      double* __restrict__ T_aug, double* __restrict__ R_aug) {
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    const double* Td = T + (size_t)draw * n * n;
    const double* Rd = R + (size_t)draw * n * k;
    double* Ta = T_aug + (size_t)draw * m * m;
    double* Ra = R_aug + (size_t)draw * m * k;
    for (int idx = threadIdx.x; idx < m * m; idx += 256) {
      const int r = idx / m, c = idx - r * m;
      double v = 0.0;
      if (r < n && c < n) {
        const int sr = inv_var_order ? inv_var_order[r] : r, sc = inv_var_order ? inv_var_order[c] : c;
        v = Td[(size_t)sr * n + sc];
      }
      Ta[idx] = v;
    }
    for (int idx = threadIdx.x; idx < m * k; idx += 256) {
      const int r = idx / k, c = idx - r * k;
      Ra[idx] = (r < n) ? Rd[(size_t)(inv_var_order ? inv_var_order[r] : r) * k + c] : 0.0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_links; i += 256) Ta[(size_t)link_rows[i] * m + link_cols[i]] = 1.0;
    __syncthreads();
  }
}

// ---- dense design matrix on the gradient path (round 3) ---------------------------------------------------------------------
// The reverse sweep (kalman_grad_kernel) is written for a selector Z.  A dense Z (observation equations, statespace.py:298-332)
// becomes one by carrying the observed combinations as p extra variables  o_t = Z x_t = Z T x_{t-1} + Z R eps_t :
//     T_aug = [ T    0 ]      R_aug = [ R   ]      Z_aug = [ 0  I_p ]      (E = [I; Z]:  T_aug = E T [I 0],  R_aug = E R)
//             [ Z T  0 ]              [ Z R ]
// The new variables are observed non-states (nobody depends on o_{t-1}), the likelihood is the same function (F = Z P Z' + H +
// jitter, P x Z' = P_xo; the jitter on the o block of P+ does not propagate: its columns of T_aug are zero), and the
// cotangents come back by the transposed maps:
//     T_bar = E' T_aug_bar[:, :n],   G_bar = E' G_aug_bar E   (G = R Q R', G_aug = E G E'),
//     Z_bar = T_aug_bar[o, :n] T' + ((G_aug_bar + G_aug_bar') E G)[o, :].
__global__ __launch_bounds__(256) void dense_z_augment_kernel(const double* __restrict__ T, const double* __restrict__ R,
                                                              const double* __restrict__ Z, int z_batched, int batch, int n,
                                                              int k, int p, double* __restrict__ T_aug,
                                                              double* __restrict__ R_aug) {
  const int m = n + p;
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    const double* Td = T + (size_t)draw * n * n;
    const double* Rd = R + (size_t)draw * n * k;
    const double* Zd = Z + (z_batched ? (size_t)draw * p * n : 0);
    double* Ta = T_aug + (size_t)draw * m * m;
    double* Ra = R_aug + (size_t)draw * m * k;
    for (int idx = threadIdx.x; idx < m * m; idx += 256) {
      const int r = idx / m, c = idx - r * m;
      double v = 0.0;
      if (c < n) {
        if (r < n) {
          v = Td[(size_t)r * n + c];
        } else {
          for (int i = 0; i < n; ++i) v = fma(Zd[(size_t)(r - n) * n + i], Td[(size_t)i * n + c], v);
        }
      }
      Ta[idx] = v;
    }
    for (int idx = threadIdx.x; idx < m * k; idx += 256) {
      const int r = idx / k, c = idx - r * k;
      double v = 0.0;
      if (r < n) {
        v = Rd[(size_t)r * k + c];
      } else {
        for (int i = 0; i < n; ++i) v = fma(Zd[(size_t)(r - n) * n + i], Rd[(size_t)i * k + c], v);
      }
      Ra[idx] = v;
    }
  }
}

// Z_aug = [0 I_p] (p x m), shared by the batch
__global__ void dense_z_selector_kernel(double* __restrict__ Z_aug, int n, int p) {
  const int m = n + p;
  for (int idx = threadIdx.x; idx < p * m; idx += blockDim.x) Z_aug[idx] = ((idx % m) == n + idx / m) ? 1.0 : 0.0;
}

// T_bar, G_bar (n x n, WRITTEN) and Z_bar (p x n per draw, optional) from the cotangents of the augmented model; G_aug = the
// augmented sym(R Q R') the forward side used (its leading n x n block is G).  Failed draws (status != 0) get zeros.
__global__ __launch_bounds__(256) void dense_z_deaugment_kernel(const double* __restrict__ Tbar_a, const double* __restrict__ Gbar_a,
                                                                const double* __restrict__ T, const double* __restrict__ G_aug,
                                                                const double* __restrict__ Z, int z_batched,
                                                                const int32_t* __restrict__ status, int batch, int n, int p,
                                                                double* __restrict__ Tbar, double* __restrict__ Gbar,
                                                                double* __restrict__ Z_bar) {
  const int m = n + p;
  extern __shared__ double sm[];  // (Gbar_a + Gbar_a') E, rows o: p x n
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    const double* Ta = Tbar_a + (size_t)draw * m * m;
    const double* Ga = Gbar_a + (size_t)draw * m * m;
    const double* Td = T + (size_t)draw * n * n;
    const double* Gd = G_aug + (size_t)draw * m * m;
    const double* Zd = Z + (z_batched ? (size_t)draw * p * n : 0);
    double* Tb = Tbar + (size_t)draw * n * n;
    double* Gb = Gbar + (size_t)draw * n * n;
    const bool dead = status && status[draw] != 0;
    for (int idx = threadIdx.x; idx < n * n; idx += 256) {
      const int i = idx / n, j = idx - i * n;
      double tv = 0.0, gv = 0.0;
      if (!dead) {
        tv = Ta[(size_t)i * m + j];
        gv = Ga[(size_t)i * m + j];
        for (int o = 0; o < p; ++o) {
          const double zi = Zd[(size_t)o * n + i], zj = Zd[(size_t)o * n + j];
          tv = fma(zi, Ta[(size_t)(n + o) * m + j], tv);
          double acc = Ga[(size_t)(n + o) * m + j];
          for (int o2 = 0; o2 < p; ++o2) acc = fma(Ga[(size_t)(n + o) * m + n + o2], Zd[(size_t)o2 * n + j], acc);
          gv = fma(zi, acc, gv);
          gv = fma(Ga[(size_t)i * m + n + o], zj, gv);
        }
      }
      Tb[idx] = tv;
      Gb[idx] = gv;
    }
    if (Z_bar) {
      __syncthreads();
      for (int idx = threadIdx.x; idx < p * n; idx += 256) {
        const int o = idx / n, i = idx - o * n;
        double acc = Ga[(size_t)(n + o) * m + i] + Ga[(size_t)i * m + n + o];
        for (int o2 = 0; o2 < p; ++o2)
          acc = fma(Ga[(size_t)(n + o) * m + n + o2] + Ga[(size_t)(n + o2) * m + n + o], Zd[(size_t)o2 * n + i], acc);
        sm[idx] = acc;
      }
      __syncthreads();
      double* Zb = Z_bar + (size_t)draw * p * n;
      for (int idx = threadIdx.x; idx < p * n; idx += 256) {
        const int o = idx / n, j = idx - o * n;
        double v = 0.0;
        if (!dead) {
          for (int c = 0; c < n; ++c) v = fma(Ta[(size_t)(n + o) * m + c], Td[(size_t)j * n + c], v);
          for (int i = 0; i < n; ++i) v = fma(sm[o * n + i], Gd[(size_t)i * m + j], v);
        }
        Zb[idx] = v;
      }
      __syncthreads();
    }
  }
}

}  // namespace dsge

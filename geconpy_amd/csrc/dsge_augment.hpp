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

}  // namespace dsge

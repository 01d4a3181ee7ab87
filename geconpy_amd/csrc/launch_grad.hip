// Launchers of the gradient kernels: Kalman reverse sweep and the reverse of the state-space assembly.
#include "dsge_host.hpp"
#include "dsge_kalman_grad.hpp"

namespace dsge_host {

// tile of the reduced filter: from the hint (states + observed variables) when given, else the model size
static int grad_tile(int u_hint, int m) { return tile_bs((u_hint > 0 && u_hint < m) ? u_hint : m); }

size_t kalman_grad_store_doubles_per_draw(int u_hint, int m, int T_len) {
  const size_t np = 8 * (size_t)grad_tile(u_hint, m);
  return (size_t)T_len * (np * np + np * 8 + 128 + np + 2) + np * np;  // per step: P+, K, F^-1, F, a_t, source index, previous source; + P_0
}

int launch_kalman_grad(const double* T, const double* RQR, const double* Z, int z_batched, const double* d, int d_batched,
                       const double* Hdiag, int h_batched, const double* y, int batch, int m, int p, int T_len,
                       double jitter, double missing_fill, int u_hint, double* store, double* logp, int32_t* status,
                       double* Tbar, double* Gbar, double* dbar, double* hbar, hipStream_t st, const int32_t* order_key,
                       int32_t* order_buf) {
  const int bs = grad_tile(u_hint, m);
  int rc = DSGE_ERR_INVALID;
  const int32_t* order = nullptr;
  if (order_key && order_buf && batch >= 512) {
    hipLaunchKernelGGL(dsge::kalman_order_kernel<1024>, dim3(1), dim3(1024), 0, st, order_key, batch, order_buf);
    HIP_TRY(hipGetLastError());
    order = order_buf;
  }
  DISPATCH_BS(bs, 8, {
    const size_t lds = dsge::KgSmem<BS>::bytes;
    if (lds > LDS_LIMIT) return fail(DSGE_ERR_INVALID, "gradient kernel: model too large for the 160 KB LDS");
    rc = set_lds(dsge::kalman_grad_kernel<BS>, lds);
    if (rc == DSGE_SUCCESS) {
      hipLaunchKernelGGL(dsge::kalman_grad_kernel<BS>, dim3(batch), dim3(64), lds, st, T, RQR, Z, z_batched, d, d_batched,
                         Hdiag, h_batched, y, batch, m, p, T_len, filter_conv(jitter), missing_fill, opt().kalman_steady_tol, store, logp, status, Tbar,
                         Gbar,
                         dbar, hbar, g_kalman_dbg, order);
      HIP_TRY(hipGetLastError());
    }
  });
  return rc;
}

int launch_grad_assemble(const double* B, const double* C, const double* T, const double* R, const double* q,
                         int q_batched, const double* Gbar, int batch, int n, int k, const int32_t* status, double* Tbar,
                         double* B_bar, double* C_bar, double* D_bar, double* q_bar, hipStream_t st, const double* Rbar_in) {
  const int bs = tile_bs(n);
  int rc = DSGE_ERR_INVALID;
  DISPATCH_BS(bs, 7, {
    rc = set_lds(dsge::grad_assemble_kernel<BS>, dsge::GaSmem<BS>::bytes);
    if (rc == DSGE_SUCCESS) {
      hipLaunchKernelGGL(dsge::grad_assemble_kernel<BS>, dim3(batch), dim3(64), dsge::GaSmem<BS>::bytes, st, B, C, T, R, q,
                         q_batched, Gbar, batch, n, k, status, Tbar, B_bar, C_bar, D_bar, q_bar, Rbar_in);
      HIP_TRY(hipGetLastError());
    }
  });
  return rc;
}

}  // namespace dsge_host

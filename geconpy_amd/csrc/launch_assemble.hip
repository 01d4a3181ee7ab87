// Launchers of the assembly kernels: selection matrix / residual / Lyapunov, policy adjoints, gEcon norms.
#include <cstdlib>
#include "dsge_host.hpp"
#include "dsge_kernels.hpp"
#include "dsge_acf.hpp"
#include "dsge_augment.hpp"

namespace dsge_host {

int launch_assemble(const double* A, const double* B, const double* C, const double* D, const double* T,
                    const double* R_in, const double* Q, int q_mode, int batch, int n, int k, double* R_out,
                    double* resid_out, double* RQR_out, double* P0_out, int32_t* status, int do_sel, int do_lyap,
                    hipStream_t st, const int32_t* only_marked) {
  const int bs = tile_bs(n);
  int rc = DSGE_ERR_INVALID;
  DISPATCH_BS(bs, 8, {
    rc = set_lds(dsge::assemble_kernel<BS>, dsge::AsmSmem<BS>::bytes);
    // sym(R Q R') alone needs the two column groups of W only (see the kernel)
    // ... and the selection without a doubling iteration M1 + W (the Gauss-Jordan scratch moves into M1)
    const size_t lds = (!do_sel && do_lyap == 2) ? sizeof(double) * dsge::AsmSmem<BS>::NP * dsge::AsmSmem<BS>::LDW
                       : (do_sel && (do_lyap == 0 || do_lyap == 2))
                           ? sizeof(double) * dsge::AsmSmem<BS>::NP * (dsge::AsmSmem<BS>::LD + dsge::AsmSmem<BS>::LDW)
                           : dsge::AsmSmem<BS>::bytes;
    if (rc == DSGE_SUCCESS) {
      hipLaunchKernelGGL(dsge::assemble_kernel<BS>, dim3(do_lyap == 3 ? rerun_grid(batch) : batch), dim3(64), lds, st, A, B, C, D, T, R_in, Q, q_mode, batch, n, k, R_out, resid_out, RQR_out, P0_out, status, do_sel, do_lyap, only_marked);
      HIP_TRY(hipGetLastError());
    }
  });
  return rc;
}

int launch_rqr(const double* R, const double* q, int q_batched, int batch, int n, int k, const int32_t* status,
               double* RQR_out, hipStream_t st, int rerun_only) {
  const size_t lds = sizeof(double) * (size_t)n * ((k + 1) & ~1);
  hipLaunchKernelGGL(dsge::rqr_kernel<dsge::RQR_KMAX>, dim3(rerun_only ? rerun_grid(batch) : batch), dim3(64), lds, st, R, q,
                     q_batched, batch, n, k, status, RQR_out, rerun_only);
  HIP_TRY(hipGetLastError());
  return DSGE_SUCCESS;
}

int g_adj_refine_mode = 0;

// The gradient pipeline's fused launch (adjoint_kernel<BS, false, true>: reverse of the assembly + policy adjoints on one elimination)
// followed by the two-kernel path over the draws it flagged -- normally none: two launches that find nothing to do.
int launch_adjoint_fused(const double* B, const double* C, const double* T, const double* R, const double* q, int q_batched,
                         const double* Gbar, double* Tbar, int batch, int n, int k, double* Ab, double* Bb, double* Cb, double* Db,
                         double* qb, int32_t* status, hipStream_t st) {
  const int bs = tile_bs(n);
  int rc = DSGE_ERR_INVALID;
  DISPATCH_BS(bs, 7, {
    rc = set_lds(dsge::adjoint_kernel<BS, false, true>, dsge::AdjSmem<BS>::bytes);
    if (rc == DSGE_SUCCESS) {
      dsge::AdjFuseArgs fa;
      fa.R = R;
      fa.q = q;
      fa.q_batched = q_batched;
      fa.Gbar = Gbar;
      fa.k = k;
      fa.D_bar = Db;
      fa.q_bar = qb;
      hipLaunchKernelGGL((dsge::adjoint_kernel<BS, false, true>), dim3(batch), dim3(64), dsge::AdjSmem<BS>::bytes, st, B, C, T,
                         (const double*)Tbar, batch, n, Ab, Bb, Cb, status, 0, g_adj_refine_mode, 0, fa);
      HIP_TRY(hipGetLastError());
    }
  });
  if (rc != DSGE_SUCCESS) return rc;
  if ((rc = launch_grad_assemble(B, C, T, R, q, q_batched, Gbar, batch, n, k, status, Tbar, Bb, Cb, Db, qb, st, nullptr, 1))) return rc;
  return launch_adjoint(B, C, T, Tbar, batch, n, Ab, Bb, Cb, status, st, 1, 1);
}

int launch_adjoint(const double* B, const double* C, const double* T, const double* Tbar, int batch, int n, double* Ab,
                   double* Bb, double* Cb, int32_t* status, hipStream_t st, int accumulate, int only_flag) {
  const int bs = tile_bs(n);
  int rc = DSGE_ERR_INVALID;
  DISPATCH_BS(bs, 7, {
    rc = set_lds(dsge::adjoint_kernel<BS, false>, dsge::AdjSmem<BS>::bytes);
    if (rc == DSGE_SUCCESS) rc = set_lds(dsge::adjoint_kernel<BS, true>, dsge::AdjSmem<BS>::bytes_refine);
    if (rc == DSGE_SUCCESS) {
      hipLaunchKernelGGL((dsge::adjoint_kernel<BS, false>), dim3(batch), dim3(64), dsge::AdjSmem<BS>::bytes, st, B, C, T, Tbar,
                         batch, n, Ab, Bb, Cb, status, accumulate, g_adj_refine_mode, only_flag, dsge::AdjFuseArgs());
      HIP_TRY(hipGetLastError());
      // second pass: one step of iterative refinement for the draws whose Stein residual the first pass flagged (normally none)
      // (one workgroup per draw: a flagged draw costs a whole solve, two of them behind each other in one workgroup twice that)
      hipLaunchKernelGGL((dsge::adjoint_kernel<BS, true>), dim3(batch), dim3(64), dsge::AdjSmem<BS>::bytes_refine, st, B, C,
                         T, Tbar, batch, n, Ab, Bb, Cb, status, 1, 0, 0, dsge::AdjFuseArgs());
      HIP_TRY(hipGetLastError());
    }
  });
  return rc;
}

int launch_norms(const double* A, const double* B, const double* C, const double* D, const double* T, const double* R,
                 const int32_t* mask, int batch, int n, int k, double* det, double* sto, hipStream_t st) {
  const int bs = tile_bs(n);
  int rc = DSGE_ERR_INVALID;
  DISPATCH_BS(bs, 8, {
    rc = set_lds(dsge::norms_kernel<BS>, dsge::NormSmem<BS>::bytes);
    if (rc == DSGE_SUCCESS) {
      hipLaunchKernelGGL(dsge::norms_kernel<BS>, dim3(batch), dim3(64), dsge::NormSmem<BS>::bytes, st, A, B, C, D, T, R,
                         mask, batch, n, k, det, sto);
      HIP_TRY(hipGetLastError());
    }
  });
  return rc;
}

int launch_augment(const double* T, const double* R, int batch, int n, int k, int m, const int32_t* inv_var_order,
                   int n_links, const int32_t* link_rows, const int32_t* link_cols, double* T_aug, double* R_aug,
                   hipStream_t st) {
  hipLaunchKernelGGL(dsge::augment_kernel, dim3(batch), dim3(256), 0, st, T, R, batch, n, k, m, inv_var_order, n_links,
                     link_rows, link_cols, T_aug, R_aug);
  HIP_TRY(hipGetLastError());
  return DSGE_SUCCESS;
}

int launch_acf(const double* T, const double* Sigma, const double* Z, const double* Hdiag, int batch, int m, int p,
               int n_lags, int lag_step, int correlation, double* out, const int32_t* status, hipStream_t st) {
  const int bs = tile_bs(m);
  int rc = DSGE_ERR_INVALID;
  DISPATCH_BS(bs, 8, {
    rc = set_lds(dsge::acf_kernel<BS>, dsge::AcfSmem<BS>::bytes);
    if (rc == DSGE_SUCCESS) {
      hipLaunchKernelGGL(dsge::acf_kernel<BS>, dim3(batch), dim3(64), dsge::AcfSmem<BS>::bytes, st, T, Sigma, Z, Hdiag,
                         batch, m, p, n_lags, lag_step, correlation, out, status);
      HIP_TRY(hipGetLastError());
    }
  });
  return rc;
}

int launch_dense_z_augment(const double* T, const double* R, const double* Z, int z_batched, int batch, int n, int k, int p,
                           double* T_aug, double* R_aug, double* Z_aug, hipStream_t st) {
  hipLaunchKernelGGL(dsge::dense_z_augment_kernel, dim3(batch), dim3(256), 0, st, T, R, Z, z_batched, batch, n, k, p, T_aug, R_aug);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(dsge::dense_z_selector_kernel, dim3(1), dim3(256), 0, st, Z_aug, n, p);
  HIP_TRY(hipGetLastError());
  return DSGE_SUCCESS;
}

int launch_dense_z_deaugment(const double* Tbar_a, const double* Gbar_a, const double* T, const double* G_aug, const double* Z,
                             int z_batched, const int32_t* status, int batch, int n, int p, double* Tbar, double* Gbar,
                             double* Z_bar, hipStream_t st) {
  hipLaunchKernelGGL(dsge::dense_z_deaugment_kernel, dim3(batch), dim3(256), sizeof(double) * (size_t)p * n, st, Tbar_a, Gbar_a, T,
                     G_aug, Z, z_batched, status, batch, n, p, Tbar, Gbar, Z_bar);
  HIP_TRY(hipGetLastError());
  return DSGE_SUCCESS;
}

int launch_status_park(int32_t* status, int32_t* park, int batch, int restore, hipStream_t st) {
  hipLaunchKernelGGL(dsge::status_park_kernel<256>, dim3((batch + 255) / 256), dim3(256), 0, st, status, park, batch, restore);
  HIP_TRY(hipGetLastError());
  return DSGE_SUCCESS;
}

}  // namespace dsge_host

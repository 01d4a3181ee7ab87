// Launchers of the solver kernels: cycle reduction and the backward-looking direct solve.
#include "dsge_host.hpp"
#include "dsge_kernels.hpp"
#include "dsge_cr_compact.hpp"

namespace dsge_host {

int g_cr_compact = 1;  // 0 = dense kernel only (tests compare the two paths)
long long* g_cr_dbg = nullptr;  // debug: device int64[8], phase cycles of draw 0 of the compact kernel

int launch_cr(const double* A, const double* B, const double* C, int batch, int n, int max_iter, double tol,
              double* T_out, int32_t* status, int32_t* n_iter, hipStream_t st, int scan_mode, const double* D, int k,
              double* R_out) {
  const int bs = tile_bs(n);
  int rc = DSGE_ERR_INVALID;
  // Column-compact kernel first (zero columns of A and C dropped); it flags the draws whose
  // s + l exceeds the tile, and the dense kernel then runs on exactly those.
  const bool compact = g_cr_compact != 0;
  if (compact) {
    DISPATCH_BS(bs, 8, {
      rc = set_lds(dsge::cr_compact_kernel<BS>, dsge::CrcSmem<BS>::bytes);
      if (rc == DSGE_SUCCESS) {
        hipLaunchKernelGGL(dsge::cr_compact_kernel<BS>, dim3(batch), dim3(64), dsge::CrcSmem<BS>::bytes, st, A, B, C,
                           batch, n, max_iter, tol, T_out, status, n_iter, g_cr_dbg, scan_mode, D, k, R_out);
        HIP_TRY(hipGetLastError());
      }
    });
    if (rc) return rc;
  }
  rc = DSGE_ERR_INVALID;
  DISPATCH_BS(bs, 8, {
    rc = set_lds(dsge::cr_kernel<BS>, dsge::CrSmem<BS>::bytes);
    if (rc == DSGE_SUCCESS) {
      hipLaunchKernelGGL(dsge::cr_kernel<BS>, dim3(batch), dim3(64), dsge::CrSmem<BS>::bytes, st, A, B, C, batch, n,
                         max_iter, tol, T_out, status, n_iter, compact ? 1 : 0, scan_mode, D, k, R_out);
      HIP_TRY(hipGetLastError());
    }
  });
  return rc;
}

int launch_bdirect(const double* A, const double* B, const double* D, int batch, int n, int k, double* T_out,
                   double* R_out, hipStream_t st) {
  const int bs = tile_bs(n);
  int rc = DSGE_ERR_INVALID;
  DISPATCH_BS(bs, 8, {
    rc = set_lds(dsge::bdirect_kernel<BS>, dsge::BdSmem<BS>::bytes);
    if (rc == DSGE_SUCCESS) {
      hipLaunchKernelGGL(dsge::bdirect_kernel<BS>, dim3(batch), dim3(64), dsge::BdSmem<BS>::bytes, st, A, B, D, batch,
                         n, k, T_out, R_out);
      HIP_TRY(hipGetLastError());
    }
  });
  return rc;
}

}  // namespace dsge_host

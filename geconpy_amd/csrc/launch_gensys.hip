// Launcher of the gensys (ordered QZ) kernel.
#include "dsge_host.hpp"
#include "dsge_gensys.hpp"

namespace dsge_host {

// choose the on-chip pencil capacity (n_cap = n + l_cap) for gensys
int gensys_caps(int n, int n_lead_hint, int* n_cap, int* l_cap) {
  int l = (n_lead_hint > 0) ? n_lead_hint : n;
  if (l > n) l = n;
  if (n + l > DSGE_MAX_N_GENSYS) l = DSGE_MAX_N_GENSYS - n;
  while (l >= 1 && dsge::gensys_smem_bytes(n, n + l, l) > LDS_LIMIT) {
    if (n_lead_hint > 0) return fail(DSGE_ERR_INVALID, "gensys: n + n_lead_hint does not fit the 160 KB LDS");
    --l;
  }
  if (l < 1) return fail(DSGE_ERR_INVALID, "gensys: model too large for the on-chip pencil");
  *l_cap = l;
  *n_cap = n + l;
  return DSGE_SUCCESS;
}

int launch_gensys(const double* A, const double* B, const double* C, int batch, int n, double tol, int n_lead_hint,
                  double* T_out, int32_t* eu_out, int32_t* status, hipStream_t st, long long* dbg) {
  int n_cap = 0, l_cap = 0;
  int rc = gensys_caps(n, n_lead_hint, &n_cap, &l_cap);
  if (rc) return rc;
  const size_t lds = dsge::gensys_smem_bytes(n, n_cap, l_cap);
  if ((rc = set_lds(dsge::gensys_kernel, lds))) return rc;
  hipLaunchKernelGGL(dsge::gensys_kernel, dim3(batch), dim3(64), lds, st, A, B, C, batch, n, n_cap, l_cap, tol, T_out,
                     eu_out, status, dbg);
  HIP_TRY(hipGetLastError());
  return DSGE_SUCCESS;
}

}  // namespace dsge_host

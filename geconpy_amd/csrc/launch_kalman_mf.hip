// Launcher of the tile-layout filter kernel (dsge_kalman_mf.hpp): its own translation unit -- seven template instances, compiled
// next to launch_kalman.hip instead of inside it.
#include <type_traits>

#include "dsge_host.hpp"
#include "dsge_kalman_mf.hpp"

namespace dsge_host {

// The covariance in the tile layout of the FP64 matrix core's 4 x 4 x 4 instruction -- downdate and both prediction products as matrix
// issues (63 per full step on the SW-shaped model).  Instances by the caller's hint: KT = tiles of four holding the state block (9 ..
// 20 state variables: KT = 3, 4, 5), TM = KT tiles for the retained variables first and, when observed non-states may add to them
// (n_state_hint + p beyond 4 KT), TM = KT + 2 as a second pass on the draws the first flagged.  The kernel checks every draw and flags
// what does not fit (DSGE_ST_INTERNAL_RERUN) for the caller's cascade.  dsge_options.kalman_mfma = 2 (default); 0: the VALU kernels.
//   rerun_first: 1 = the first instance, too, is a second pass (only draws flagged DSGE_ST_INTERNAL_RERUN).
//   *launched: number of instances launched;  *covers: every instance wanted was launched.  What these instances refuse -- more state
//   variables than the hint, a design matrix that is no selector -- the VALU fast kernels refuse too (their state-block capacity
//   comes from the same hint, and with s <= 20, p <= 8 no draw has more than 28 retained variables): the caller then skips that
//   cascade's empty second passes (~5 us each) and a refused draw goes straight to the general kernel.
int launch_kalman_mf(const double* T, const double* RQR, const double* P0, const double* Z, int z_batched, const double* d,
                     int d_batched, const double* Hdiag, int h_batched, const double* y, int batch, int m, int p, int T_len,
                     dsge::FilterConv cv, double missing_fill, int n_state_hint, double* logp, int32_t* status, hipStream_t st,
                     const int32_t* order, const double* Rsel, const double* qdiag, int q_batched, int k_shocks,
                     const unsigned long long* colmask, int rerun_first, int* launched, bool* covers) {
  *launched = 0;
  *covers = false;
  const size_t r_doubles = Rsel ? (size_t)m * ((k_shocks + 1) & ~1) : 0;
  int n_mf = 0;
  auto launch_mf = [&](auto kt_tag, auto tm_tag, auto dbg_tag) -> int {
    constexpr int KTV = decltype(kt_tag)::value, TMV = decltype(tm_tag)::value;
    constexpr bool DBGV = decltype(dbg_tag)::value;
    using SMF = dsge::KmfSmem<KTV, TMV>;
    if (r_doubles > (size_t)SMF::WT) return DSGE_SUCCESS;  // the staged selection matrix does not fit this instance's W' buffer
    int rc2;
    if ((rc2 = set_lds(dsge::kalman_mf_kernel<KTV, TMV, DBGV>, SMF::bytes))) return rc2;
    hipLaunchKernelGGL((dsge::kalman_mf_kernel<KTV, TMV, DBGV>), dim3(batch), dim3(64), SMF::bytes, st, T, RQR, P0, Z, z_batched, d,
                       d_batched, Hdiag, h_batched, y, batch, m, p, T_len, cv, missing_fill, opt().kalman_steady_tol, logp, status,
                       DBGV ? g_kalman_dbg : (long long*)nullptr, (n_mf > 0 || rerun_first) ? 1 : 0, g_kalman_steady_at, order, Rsel, qdiag, q_batched,
                       k_shocks, colmask);
    HIP_TRY(hipGetLastError());
    ++n_mf;
    return DSGE_SUCCESS;
  };
  using std::integral_constant;
  const int kt = (n_state_hint + 3) / 4;
  const bool wide = n_state_hint + p > 4 * kt && n_state_hint < m;  // observed non-states may exceed the KT x KT instance
  int rc = DSGE_SUCCESS;
  if (g_kalman_dbg) {  // (tools/kalman_phases.py: the stamped instance of the SW-shaped size)
    if (kt == 5) rc = launch_mf(integral_constant<int, 5>{}, integral_constant<int, 5>{}, std::true_type{});
  } else if (kt == 5) {
    rc = launch_mf(integral_constant<int, 5>{}, integral_constant<int, 5>{}, std::false_type{});
    if (!rc && wide) rc = launch_mf(integral_constant<int, 5>{}, integral_constant<int, 7>{}, std::false_type{});
  } else if (kt == 4) {
    rc = launch_mf(integral_constant<int, 4>{}, integral_constant<int, 4>{}, std::false_type{});
    if (!rc && wide) rc = launch_mf(integral_constant<int, 4>{}, integral_constant<int, 6>{}, std::false_type{});
  } else {
    rc = launch_mf(integral_constant<int, 3>{}, integral_constant<int, 3>{}, std::false_type{});
    if (!rc && wide) rc = launch_mf(integral_constant<int, 3>{}, integral_constant<int, 5>{}, std::false_type{});
  }
  if (rc) return rc;
  *launched = n_mf;
  *covers = !g_kalman_dbg && n_mf == (wide ? 2 : 1);
  return DSGE_SUCCESS;
}

}  // namespace dsge_host

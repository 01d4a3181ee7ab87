// Launchers of the gradient kernels: Kalman reverse sweep and the reverse of the state-space assembly.
#include <cstdlib>
#include "dsge_host.hpp"
#include "dsge_kalman_grad.hpp"
#include "dsge_kalman_mf.hpp"
#include "dsge_kalman_nt.hpp"

namespace dsge_host {

// tile of the reduced filter: from the hint (states + observed variables) when given, else the model size
static int grad_tile(int u_hint, int m) { return tile_bs((u_hint > 0 && u_hint < m) ? u_hint : m); }

// the per-draw record of the forward sweep (dsge_kalman_rec.hpp: KgRec<BS>::per_draw) for the tile the launch will use
size_t kalman_grad_store_doubles_per_draw(int u_hint, int m, int T_len) {
  const size_t np = 8 * (size_t)grad_tile(u_hint, m);
  const size_t bs = np / 8;
  // per step: P+, K, F^-1, F, a_t, source index, previous source; + P_0; + the tail kernel's state
  return (size_t)T_len * (np * np + np * 8 + 128 + np + 2) + np * np + (16 + np + 8 + 64 * (bs * bs + bs + 1));
}
static_assert(dsge::KgRec<3>::per_draw(200) == (size_t)200 * (24 * 24 + 24 * 8 + 128 + 24 + 2) + 24 * 24 + (16 + 24 + 8 + 64 * 13) &&
                  dsge::KgRec<7>::per_draw(5) == (size_t)5 * (56 * 56 + 56 * 8 + 128 + 56 + 2) + 56 * 56 + (16 + 56 + 8 + 64 * 57),
              "kalman_grad_store_doubles_per_draw and KgRec describe the same record");

int launch_kalman_grad(const double* T, const double* RQR, const double* Z, int z_batched, const double* d, int d_batched,
                       const double* Hdiag, int h_batched, const double* y, int batch, int m, int p, int T_len,
                       double jitter, double missing_fill, int u_hint, double* store, double* logp, int32_t* status,
                       double* Tbar, double* Gbar, double* dbar, double* hbar, hipStream_t st, int32_t* order_key,
                       int32_t* order_buf) {
  const int bs = grad_tile(u_hint, m);
  int rc = DSGE_ERR_INVALID;
  const int32_t* order = nullptr;
  if (order_key && order_buf && batch >= 512) {
    hipLaunchKernelGGL(dsge::kalman_order_kernel<1024>, dim3(1), dim3(1024), 0, st, order_key, batch, order_buf);
    HIP_TRY(hipGetLastError());
    order = order_buf;
  }
  const dsge::FilterConv cv = filter_conv(jitter);
  const double stol = opt().kalman_steady_tol;
  DISPATCH_BS(bs, 8, {
    const size_t lds = dsge::KgSmem<BS>::bytes;
    if (lds > LDS_LIMIT) return fail(DSGE_ERR_INVALID, "gradient kernel: model too large for the 160 KB LDS");
    // Round 5 (dsge_options.kalman_grad_split): the forward sweep is the logp kernel with record output (two wavefronts per SIMD on
    // the tiles of up to 24 variables, its NT products, its row-per-lane update), the reverse sweep a kernel of its own; what the
    // forward kernel cannot take (it flags DSGE_ST_INTERNAL_RERUN) goes through the one-kernel path afterwards, in the same call.
    // Not under the phase-stamp hook (tools/grad_phases.py reads the one-kernel path's stamps).
    const size_t lds_f = dsge::KntSmem<BS, 8 * BS>::bytes(8 * BS);
    // (under the phase-stamp hook the split path runs too when DSGE_DBG_SPLIT_PHASES is set: the reverse sweep's stamps of draw 0 then
    //  come from kalman_grad_kernel<BS, true>; tools/grad_phases.py reads the one-kernel path's stamps otherwise)
    static const bool dbg_split = std::getenv("DSGE_DBG_SPLIT_PHASES") != nullptr;
    const bool split = opt().kalman_grad_split != 0 && (!g_kalman_dbg || dbg_split) && p <= 8 && lds_f <= LDS_LIMIT;
    if (split) {
      rc = set_lds(dsge::kalman_nt_kernel<BS, false, 8 * BS, false, true>, lds_f);
      if (rc == DSGE_SUCCESS) rc = set_lds(dsge::kalman_grad_kernel<BS, true>, lds);
      if (rc == DSGE_SUCCESS) rc = set_lds(dsge::kalman_grad_kernel<BS, false>, lds);
      // Round 6: on the 24-wide tile the forward sweep is the tile-layout filter with record output (kalman_mf_kernel<5, 5, .., REC>:
      // up to 20 retained variables, selector Z; P+ recorded as its upper tiles, which the reverse sweep scatters); what it cannot
      // take it flags, and kalman_nt_kernel<.., REC> then runs as a second pass over those draws only.
      int nt_rerun = 0;
      // the forward sweep reports every draw's first steady step into the (consumed) key buffer: the reverse launches -- whose time
      // per draw is the number of full steps -- then start their draws in THAT order (round 6; the key of the forward launch, the
      // solver's iteration count, is a proxy for it)
      int32_t* const steps_key = order ? order_key : nullptr;
      if constexpr (BS == 3) {
        using SMF = dsge::KmfSmem<5, 5>;
        if (rc == DSGE_SUCCESS && opt().kalman_mfma == 2 && opt().kalman_nt_products && u_hint > 0 && u_hint <= 20) {
          rc = set_lds(dsge::kalman_mf_kernel<5, 5, false, true, 3>, SMF::bytes);
          if (rc == DSGE_SUCCESS) {
            hipLaunchKernelGGL((dsge::kalman_mf_kernel<5, 5, false, true, 3>), dim3(batch), dim3(64), SMF::bytes, st, T, RQR,
                               (const double*)nullptr, Z, z_batched, d, d_batched, Hdiag, h_batched, y, batch, m, p, T_len, cv,
                               missing_fill, stol, logp, status, (long long*)nullptr, 0, steps_key, order,
                               (const double*)nullptr, (const double*)nullptr, 0, 0, (const unsigned long long*)nullptr, store);
            nt_rerun = 1;
          }
        }
      }
      if (rc == DSGE_SUCCESS) {
        hipLaunchKernelGGL((dsge::kalman_nt_kernel<BS, false, 8 * BS, false, true>), dim3(batch), dim3(64), lds_f, st, T, RQR,
                           (const double*)nullptr, Z, z_batched, d, d_batched, Hdiag, h_batched, y, batch, m, p, T_len, 8 * BS, cv,
                           missing_fill, stol, logp, status, (long long*)nullptr, nt_rerun, steps_key, order,
                           (const double*)nullptr, (const double*)nullptr, 0, 0, (const unsigned long long*)nullptr,
                           (double*)nullptr, (int32_t*)nullptr, (const int32_t*)nullptr, store);
        // the reverse mean side of every draw's LAST steady segment at two wavefronts per SIMD (kalman_grad_tail_kernel); the
        // reverse sweep then starts at that segment's source step (kalman_grad_split = 2; the default is 1: without it)
        if (steps_key) {
          hipLaunchKernelGGL(dsge::kalman_order_kernel<1024>, dim3(1), dim3(1024), 0, st, (const int32_t*)steps_key, batch, order_buf,
                             T_len);
        }
        const int with_tail = opt().kalman_grad_split >= 2;
        if (with_tail)
          hipLaunchKernelGGL((dsge::kalman_grad_tail_kernel<BS>), dim3(batch), dim3(64), 0, st, T, Z, z_batched, d, d_batched, y,
                             batch, m, p, T_len, cv, missing_fill, store, (const int32_t*)status, order);
        hipLaunchKernelGGL((dsge::kalman_grad_kernel<BS, true>), dim3(batch), dim3(64), lds, st, T, RQR, Z, z_batched, d, d_batched,
                           Hdiag, h_batched, y, batch, m, p, T_len, cv, missing_fill, stol, store, logp, status, Tbar, Gbar, dbar,
                           hbar, g_kalman_dbg, order, 0, with_tail);
        hipLaunchKernelGGL((dsge::kalman_grad_kernel<BS, false>), dim3(batch), dim3(64), lds, st, T, RQR, Z, z_batched, d,
                           d_batched, Hdiag, h_batched, y, batch, m, p, T_len, cv, missing_fill, stol, store, logp, status, Tbar, Gbar,
                           dbar, hbar, (long long*)nullptr, (const int32_t*)nullptr, 1, 0);
        HIP_TRY(hipGetLastError());
      }
    } else {
      rc = set_lds(dsge::kalman_grad_kernel<BS, false>, lds);
      if (rc == DSGE_SUCCESS) {
        hipLaunchKernelGGL((dsge::kalman_grad_kernel<BS, false>), dim3(batch), dim3(64), lds, st, T, RQR, Z, z_batched, d, d_batched,
                           Hdiag, h_batched, y, batch, m, p, T_len, cv, missing_fill, stol, store, logp, status, Tbar, Gbar, dbar,
                           hbar, g_kalman_dbg, order, 0, 0);
        HIP_TRY(hipGetLastError());
      }
    }
  });
  return rc;
}

int launch_grad_assemble(const double* B, const double* C, const double* T, const double* R, const double* q,
                         int q_batched, const double* Gbar, int batch, int n, int k, const int32_t* status, double* Tbar,
                         double* B_bar, double* C_bar, double* D_bar, double* q_bar, hipStream_t st, const double* Rbar_in,
                         int only_flag) {
  const int bs = tile_bs(n);
  int rc = DSGE_ERR_INVALID;
  DISPATCH_BS(bs, 7, {
    rc = set_lds(dsge::grad_assemble_kernel<BS>, dsge::GaSmem<BS>::bytes);
    if (rc == DSGE_SUCCESS) {
      hipLaunchKernelGGL(dsge::grad_assemble_kernel<BS>, dim3(batch), dim3(64), dsge::GaSmem<BS>::bytes, st, B, C, T, R, q,
                         q_batched, Gbar, batch, n, k, status, Tbar, B_bar, C_bar, D_bar, q_bar, Rbar_in, only_flag);
      HIP_TRY(hipGetLastError());
    }
  });
  return rc;
}

}  // namespace dsge_host

// Launcher of the second-order path (dsge_second_order.hpp): structure kernels, coefficient / pruned-system set-up,
// stationary covariance and the filter on the FP64 matrix core.
#include "dsge_host.hpp"
#include "dsge_second_order.hpp"

#include <algorithm>
#include <mutex>

namespace dsge_host {

long long* g_so_dbg = nullptr;  // debug: device int64[8], phase cycles of draw 0 of the filter kernel

namespace {
StreamArenaPool g_so_pool;
int so_reserve(size_t bytes, hipStream_t st, void** out) { return g_so_pool.reserve(bytes, st, out); }

template <int MT>
int launch_so_mt(const dsge::SoFilterArgs& fa, const dsge::SoLayout& lay, int nb, hipStream_t st, float* ms) {
  int rc;
  const size_t lds = dsge::SoFilterSmem<MT>::bytes;
  if ((rc = set_lds(dsge::so_lyap_kernel<MT>, lds))) return rc;
  if ((rc = set_lds(dsge::so_filter_kernel<MT>, lds))) return rc;
  EventGuard e[3];
  if (ms)
    for (auto& x : e) HIP_TRY(x.create());
  if (ms) HIP_TRY(hipEventRecord(e[0], st));
  hipLaunchKernelGGL(dsge::so_lyap_kernel<MT>, dim3(nb), dim3(dsge::SO_THREADS), lds, st, fa, lay);
  HIP_TRY(hipGetLastError());
  if (ms) HIP_TRY(hipEventRecord(e[1], st));
  hipLaunchKernelGGL(dsge::so_filter_kernel<MT>, dim3(nb), dim3(dsge::SO_THREADS), lds, st, fa, lay);
  HIP_TRY(hipGetLastError());
  if (ms) {
    HIP_TRY(hipEventRecord(e[2], st));
    HIP_TRY(hipEventSynchronize(e[2]));
    float t1 = 0.f, t2 = 0.f;
    HIP_TRY(hipEventElapsedTime(&t1, e[0], e[1]));
    HIP_TRY(hipEventElapsedTime(&t2, e[1], e[2]));
    ms[1] += t1;  // accumulated over the chunks of a call (launch_second_order zeroes them)
    ms[2] += t2;
  }
  return DSGE_SUCCESS;
}
}  // namespace

int so_tiles(int m) {  // kernel instances built: 2, 4, 7, 10, 13 tiles of 16 per side
  const int need = (m + 15) / 16;
  for (int mt : {2, 4, 7, 10, 13})
    if (need <= mt) return mt;
  return 0;
}

// T, R: first-order solution (device, [batch][n][n], [batch][n][k]); status_io: draws with a non-zero status are skipped
// (logp = -inf).  ms (nullable): durations of the three stages in milliseconds, summed over ALL workspace chunks of the call
// (synchronises after every chunk).
int launch_second_order(const double* B, const double* C, const double* T, const double* R, const int32_t* hess_idx, int nnz,
                        const double* hess_val, const double* q, int q_batched, const double* Z, const double* d,
                        const double* Hdiag, const double* y, int batch, int n, int k, int p, int T_len, double jitter,
                        double missing_fill, const int32_t* S, int s, const int32_t* L, int l, const int32_t* U, int u,
                        double* logp, int32_t* status_io, double* gyy_out, double* gyu_out, double* guu_out, double* gss_out,
                        int32_t* steady_at, int32_t* n_doublings, hipStream_t st, float* ms, const int32_t* order_key) {
  if (s < 1 || s > dsge::SO_MAX_S || u < s || u > 40 || l < 0 || l > 64 || k > dsge::SO_MAX_K || k > s || p > dsge::SO_MAX_P)
    return fail(DSGE_ERR_INVALID, "second order: sizes out of range (1 <= s <= 24, s <= u <= 40, k <= min(s, 12), p <= 8)");
  const int q_ = s * (s + 1) / 2, m = 2 * u + q_, mt = so_tiles(m);
  if (mt == 0) return fail(DSGE_ERR_INVALID, "second order: pruned state 2 u + s (s + 1) / 2 exceeds 208");
  dsge::SoLayout lay;
  lay.init(n, k, s, u, l, p, mt);
  const size_t lds_setup = dsge::so_setup_lds_doubles(lay) * sizeof(double);
  if (lds_setup > LDS_LIMIT) return fail(DSGE_ERR_INVALID, "second order: model too large for the set-up kernel's LDS");
  dsge::SoIdx ix{};
  for (int i = 0; i < s; ++i) ix.S[i] = (uint8_t)S[i];
  for (int i = 0; i < u; ++i) ix.U[i] = (uint8_t)U[i];
  for (int i = 0; i < l; ++i) ix.L[i] = (uint8_t)L[i];
  for (int i = 0; i < s; ++i)
    if (U[i] != S[i]) return fail(DSGE_ERR_INVALID, "second order: the retained variables must list the states first");
  // workspace: chunks of draws so that it stays below ~6 GiB
  const size_t per_draw = lay.total * sizeof(double);
  int chunk = (int)std::min<size_t>((size_t)batch, std::max<size_t>(1, ((size_t)6 << 30) / per_draw));
  int rc;
  void* base = nullptr;
  const size_t head = 4096 + sizeof(double) * 8 * 40;
  if ((rc = so_reserve(head + per_draw * chunk + 2 * sizeof(int32_t) * (size_t)chunk + 256, st, &base))) return rc;
  int32_t* hptr = (int32_t*)base;                      // [n + 1]
  int32_t* flags = hptr + 128;                         // [2]
  double* Zu = (double*)((char*)base + 4096);          // [p][u]
  double* work = (double*)((char*)base + head);
  int32_t* order = (int32_t*)((char*)base + head + per_draw * chunk);
  int32_t* key = order + chunk;
  // Dispatch order of the filter launch (slow draws first; a never-steady draw is 200 full steps = 68 ms of one CU, the
  // average draw 17 ms, four draws per CU at 1024 draws): the number of full steps follows the spectral radius of T (rank
  // correlation 0.99 on the SW-shaped draws, tools/so_order_potential.py; the cycle-reduction iteration count of the
  // first-order kernels: 0.76), so the key is persistence_key_kernel's power-iteration estimate; the set-up kernel raises it
  // to the top for nearly singular M = B + C T, whose covariance never leaves its rounding noise.
  const bool use_key = opt().kalman_order && n <= 64;
  (void)order_key;
  HIP_TRY(hipMemsetAsync(flags, 0, 2 * sizeof(int32_t), st));
  hipLaunchKernelGGL(dsge::so_hessptr_kernel, dim3(1), dim3(256), 0, st, hess_idx, nnz, n, k, hptr, flags);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(dsge::so_design_kernel, dim3(1), dim3(256), 0, st, Z, p, n, ix, u, Zu, flags);
  HIP_TRY(hipGetLastError());
  if ((rc = set_lds(dsge::so_setup_kernel, lds_setup))) return rc;
  if (ms) ms[0] = ms[1] = ms[2] = 0.f;
  for (int c0 = 0; c0 < batch; c0 += chunk) {
    const int nb = std::min(chunk, batch - c0);
    dsge::SoSetupArgs sa{};
    sa.B = B + (size_t)c0 * n * n;
    sa.C = C + (size_t)c0 * n * n;
    sa.T = T + (size_t)c0 * n * n;
    sa.R = R + (size_t)c0 * n * k;
    sa.hess_ptr = hptr;
    sa.hess_idx = hess_idx;
    sa.hess_val = hess_val + (size_t)c0 * nnz;
    sa.q = q + (q_batched ? (size_t)c0 * k : 0);
    sa.ix = ix;
    sa.flags = flags;
    sa.work = work;
    sa.status = status_io + c0;
    sa.gyy_out = gyy_out ? gyy_out + (size_t)c0 * n * s * s : nullptr;
    sa.gyu_out = gyu_out ? gyu_out + (size_t)c0 * n * s * k : nullptr;
    sa.guu_out = guu_out ? guu_out + (size_t)c0 * n * k * k : nullptr;
    sa.gss_out = gss_out ? gss_out + (size_t)c0 * n : nullptr;
    sa.batch = nb;
    sa.nnz = nnz;
    sa.q_batched = q_batched;
    sa.order_key = nullptr;
    if (use_key && nb >= 512) {
      if ((rc = launch_persistence_key(T + (size_t)c0 * n * n, status_io + c0, nb, n, key, st))) return rc;
      sa.order_key = key;
    }
    const bool time_it = ms != nullptr;
    EventGuard e0, e1;
    if (time_it) {
      HIP_TRY(e0.create());
      HIP_TRY(e1.create());
      HIP_TRY(hipEventRecord(e0, st));
    }
    hipLaunchKernelGGL(dsge::so_setup_kernel, dim3(nb), dim3(dsge::SO_SETUP_THREADS), lds_setup, st, sa, lay);
    HIP_TRY(hipGetLastError());
    if (time_it) HIP_TRY(hipEventRecord(e1, st));
    dsge::SoFilterArgs fa{};
    fa.work = work;
    fa.Zu = Zu;
    fa.d = d;
    fa.Hdiag = Hdiag;
    fa.y = y;
    fa.logp = logp + c0;
    fa.status = status_io + c0;
    fa.order = nullptr;
    if (sa.order_key) {  // slow draws first (descending key)
      hipLaunchKernelGGL(dsge::kalman_order_kernel<1024>, dim3(1), dim3(1024), 0, st, key, nb, order);
      HIP_TRY(hipGetLastError());
      fa.order = order;
    }
    fa.steady_at = steady_at ? steady_at + c0 : nullptr;
    fa.n_doublings = n_doublings ? n_doublings + c0 : nullptr;
    fa.phases = (c0 == 0) ? g_so_dbg : nullptr;
    fa.batch = nb;
    fa.T_len = T_len;
    fa.cv = filter_conv(jitter);
    fa.missing_fill = missing_fill;
    // The scale-free test of the second-order filter ((dP_ij)^2 <= tol^2 P_ii P_jj for EVERY entry of a 207 x 207 matrix whose
    // entries are sums of 207 products) meets its own rounding noise at ~1e-14: with the option's default the covariance of
    // one draw in twenty never "stops" (profiles/r3/so_steady_tol_sweep.txt).  100 x the option value (1e-12 by default) moves
    // logp by 3.6e-14 relative against the full recursion -- the level the first-order kernels' max-norm test has at 1e-14.
    fa.steady_tol = 100.0 * opt().kalman_steady_tol;
    float* msl = time_it ? ms : nullptr;
    switch (mt) {
      case 2: rc = launch_so_mt<2>(fa, lay, nb, st, msl); break;
      case 4: rc = launch_so_mt<4>(fa, lay, nb, st, msl); break;
      case 7: rc = launch_so_mt<7>(fa, lay, nb, st, msl); break;
      case 10: rc = launch_so_mt<10>(fa, lay, nb, st, msl); break;
      case 13: rc = launch_so_mt<13>(fa, lay, nb, st, msl); break;
      default: rc = fail(DSGE_ERR_INVALID, "second order: no kernel instance");
    }
    if (rc) return rc;
    if (time_it) {
      float t0 = 0.f;
      HIP_TRY(hipEventElapsedTime(&t0, e0, e1));
      ms[0] += t0;
    }
  }
  return DSGE_SUCCESS;
}

}  // namespace dsge_host

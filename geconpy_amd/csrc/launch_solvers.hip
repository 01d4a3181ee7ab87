// Launchers of the solver kernels: cycle reduction and the backward-looking direct solve.
#include "dsge_host.hpp"
#include "dsge_kernels.hpp"
#include "dsge_cr_compact.hpp"
#include "dsge_cr_deflate.hpp"
#include "dsge_cr_fused.hpp"
#include "dsge_cr_wide.hpp"

#include <algorithm>
#include <mutex>

namespace dsge_host {

long long* g_cr_dbg = nullptr;  // debug: device int64[8], phase cycles of draw 0 of the compact kernel


int launch_cr(const double* A, const double* B, const double* C, int batch, int n, int max_iter, double tol,
              double* T_out, int32_t* status, int32_t* n_iter, hipStream_t st, int scan_mode, const double* D, int k,
              double* R_out) {
  const int bs = tile_bs(n);
  int rc = DSGE_ERR_INVALID;
  // Column-compact kernel first (zero columns of A and C dropped); it flags the draws whose
  // s + l exceeds the tile, and the dense kernel then runs on exactly those.
  const bool compact = opt().cr_compact != 0;
  if (compact && n > 48 && opt().cr_four_waves) {
    // 49..64 variables: four wavefronts per draw (dsge_cr_wide.hpp); it flags what it cannot take like the compact kernel
    if ((rc = set_lds(dsge::cr_wide_kernel, dsge::CrwSmem::bytes))) return rc;
    hipLaunchKernelGGL(dsge::cr_wide_kernel, dim3(batch), dim3(256), dsge::CrwSmem::bytes, st, A, B, C, batch, n, max_iter, tol,
                       T_out, status, n_iter, scan_mode, D, k, R_out);
    HIP_TRY(hipGetLastError());
  } else if (compact) {
    DISPATCH_BS(bs, 8, {
      if (BS == 4 && opt().cr_two_waves) {
        rc = set_lds(dsge::cr_compact_kernel_occ2<4>, dsge::CrcSmem<4>::bytes);
        if (rc == DSGE_SUCCESS) {
          hipLaunchKernelGGL(dsge::cr_compact_kernel_occ2<4>, dim3(batch), dim3(64), dsge::CrcSmem<4>::bytes, st, A, B, C,
                             batch, n, max_iter, tol, T_out, status, n_iter, g_cr_dbg, scan_mode, D, k, R_out);
          HIP_TRY(hipGetLastError());
        }
      } else {
        rc = set_lds(dsge::cr_compact_kernel<BS>, dsge::CrcSmem<BS>::bytes);
        if (rc == DSGE_SUCCESS) {
          hipLaunchKernelGGL(dsge::cr_compact_kernel<BS>, dim3(batch), dim3(64), dsge::CrcSmem<BS>::bytes, st, A, B, C,
                             batch, n, max_iter, tol, T_out, status, n_iter, g_cr_dbg, scan_mode, D, k, R_out);
          HIP_TRY(hipGetLastError());
        }
      }
    });
    if (rc) return rc;
  }
  rc = DSGE_ERR_INVALID;
  DISPATCH_BS(bs, 8, {
    rc = set_lds(dsge::cr_kernel<BS>, dsge::CrSmem<BS>::bytes);
    if (rc == DSGE_SUCCESS) {
      hipLaunchKernelGGL(dsge::cr_kernel<BS>, dim3(compact ? rerun_grid(batch) : batch), dim3(64), dsge::CrSmem<BS>::bytes, st, A,
                         B, C, batch, n, max_iter, tol, T_out, status, n_iter, compact ? 1 : 0, scan_mode, D, k, R_out);
      HIP_TRY(hipGetLastError());
    }
  });
  return rc;
}

// ---- cycle reduction behind the static-variable deflation (dsge_cr_deflate.hpp) -----------------------------------------

namespace {
std::mutex g_defl_mutex;
int g_static_hint[DSGE_MAX_N + 2];  // per model size n: lower bound of the number of static variables; 0 = not measured yet
unsigned g_static_calls[DSGE_MAX_N + 2];
bool g_static_hint_init = false;    // (stored as h + 1)
StreamArenaPool g_defl_pool;
int defl_reserve(size_t bytes, hipStream_t st, void** out) { return g_defl_pool.reserve(bytes, st, out); }
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
}  // namespace

void cr_deflation_reset() {
  std::lock_guard<std::mutex> lk(g_defl_mutex);
  g_static_hint_init = false;
}

// deflation + cycle reduction + inflation in one launch; *done = 0 if no kernel instance covers (n, n - h)
namespace {
template <int BSF, int BSD, int NC>
int launch_cr_fused_inst(const double* A, const double* B, const double* C, const double* D, int batch, int n, int k, int h,
                         int max_iter, double tol, double* top, double* rhs, double* T_out, double* R_out, int32_t* status,
                         int32_t* n_iter, hipStream_t st, unsigned long long* colmask) {
  using SM = dsge::CrfSmem<BSF, BSD>;
  int rc;
  if constexpr (BSD == 4 && NC == 2) {
    if (opt().cr_two_waves) {
      if ((rc = set_lds(dsge::cr_fused_kernel_occ2<BSF, BSD>, SM::bytes))) return rc;
      hipLaunchKernelGGL((dsge::cr_fused_kernel_occ2<BSF, BSD>), dim3(batch), dim3(64), SM::bytes, st, A, B, C, D, batch, n, k, h,
                         max_iter, tol, top, rhs, T_out, R_out, status, n_iter, colmask);
      HIP_TRY(hipGetLastError());
      return DSGE_SUCCESS;
    }
  }
  if ((rc = set_lds(dsge::cr_fused_kernel<BSF, BSD, NC>, SM::bytes))) return rc;
  hipLaunchKernelGGL((dsge::cr_fused_kernel<BSF, BSD, NC>), dim3(batch), dim3(64), SM::bytes, st, A, B, C, D, batch, n, k, h,
                     max_iter, tol, top, rhs, T_out, R_out, status, n_iter, colmask);
  HIP_TRY(hipGetLastError());
  return DSGE_SUCCESS;
}
}  // namespace

int launch_cr_fused(const double* A, const double* B, const double* C, const double* D, int batch, int n, int k, int h,
                    int max_iter, double tol, double* T_out, double* R_out, int32_t* status, int32_t* n_iter, hipStream_t st,
                    int* done, unsigned long long* colmask) {
  *done = 0;
  const int nd = n - h, bsf = tile_bs(n), bsd = tile_bs(nd);
  const int ncol = h + 3 * nd + k, nc = ncol <= 128 ? 2 : 3;  // columns of [B_st | B_dy | A_dy | C_dy | D] per lane
  if (ncol > 192 || nd + k > 64) return DSGE_SUCCESS;
  // instances built: (full tile, reduced tile) x columns per lane
  const bool have = nc == 2 ? ((bsf == 3 && (bsd == 2 || bsd == 3)) || (bsf == 4 && (bsd == 3 || bsd == 4)) ||
                               (bsf == 5 && bsd == 4) || (bsf == 6 && (bsd == 4 || bsd == 5)))
                            : ((bsf == 6 && bsd == 5) || (bsf == 7 && (bsd == 5 || bsd == 6)) || (bsf == 8 && bsd == 6));
  if (!have) return DSGE_SUCCESS;
  int rc;
  void* base = nullptr;
  const size_t tops = (size_t)batch * dsge::crd_top_doubles(n, k, h), rhss = (size_t)batch * dsge::crf_rhs_doubles(nd, 8 * bsd);
  if ((rc = defl_reserve(al256(tops * 8) + al256(rhss * 8) + 4096, st, &base))) return rc;
  double* top = (double*)base;
  double* rhs = (double*)((char*)base + al256(tops * 8));
#define FUSED_CASE(F, D_, N_)                                                                                          \
  if (bsf == F && bsd == D_ && nc == N_)                                                                               \
    rc = launch_cr_fused_inst<F, D_, N_>(A, B, C, D, batch, n, k, h, max_iter, tol, top, rhs, T_out, R_out, status, n_iter, st, \
                                         colmask)
  rc = DSGE_ERR_INVALID;
  FUSED_CASE(3, 2, 2);
  FUSED_CASE(3, 3, 2);
  FUSED_CASE(4, 3, 2);
  FUSED_CASE(4, 4, 2);
  FUSED_CASE(5, 4, 2);
  FUSED_CASE(6, 4, 2);
  FUSED_CASE(6, 5, 2);
  FUSED_CASE(6, 5, 3);
  FUSED_CASE(7, 5, 3);
  FUSED_CASE(7, 6, 3);
  FUSED_CASE(8, 6, 3);
#undef FUSED_CASE
  if (rc) return rc;
  // draws the fused kernel could not take (status == DSGE_ST_INTERNAL_RERUN): full-size dense kernel on exactly those
  rc = DSGE_ERR_INVALID;
  DISPATCH_BS(bsf, 8, {
    rc = set_lds(dsge::cr_kernel<BS>, dsge::CrSmem<BS>::bytes);
    if (rc == DSGE_SUCCESS) {
      hipLaunchKernelGGL(dsge::cr_kernel<BS>, dim3(rerun_grid(batch)), dim3(64), dsge::CrSmem<BS>::bytes, st, A, B, C, batch, n,
                         max_iter, tol, T_out, status, n_iter, 1, 0, D, k, R_out);
      HIP_TRY(hipGetLastError());
    }
  });
  if (rc) return rc;
  *done = 1;
  return DSGE_SUCCESS;
}

// *used = 0: nothing done (deflation off, too few static variables, ...): the caller runs launch_cr on the full system.
int launch_cr_deflated(const double* A, const double* B, const double* C, const double* D, int batch, int n, int k,
                       int max_iter, double tol, double* T_out, double* R_out, int32_t* status, int32_t* n_iter,
                       hipStream_t st, int* used, unsigned long long* colmask) {
  *used = 0;
  if (!opt().cr_deflation || !D || !R_out || n < 8 || n > 64 || batch < 1) return DSGE_SUCCESS;
  int rc;
  void* base = nullptr;
  int h;
  if (opt().n_static_hint >= 0) {
    // the caller's bound (a property of the model, computed once on the host like n_state_hint): nothing is measured, no
    // read-back, no synchronisation -- the call only enqueues; a draw with fewer static variables is caught on the device
    h = std::min(opt().n_static_hint, (int)dsge::CRD_HMAX);
  } else {
    std::lock_guard<std::mutex> lk(g_defl_mutex);
    if (!g_static_hint_init) {
      for (auto& x : g_static_hint) x = 0;
      for (auto& x : g_static_calls) x = 0;
      g_static_hint_init = true;
    }
    // n_static_hint = -1: first batch of this model size measures (one small launch and a 4-byte read-back, which
    // synchronises the stream); every 256th call after that measures again and keeps the minimum, so that an
    // unrepresentative first batch corrects itself.  Callers that need a pure enqueue pass the hint.
    // (the measurement synchronises the stream: never while the caller is capturing it into a graph -- the periodic renewal is
    //  skipped then, and a FIRST call under capture has nothing to go by: it runs the full-size solve.  Graph capture of this
    //  path wants n_static_hint, include/dsge_hip.h)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = (hipStreamIsCapturing(st, &cap) == hipSuccess) && cap != hipStreamCaptureStatusNone;
    if (capturing && g_static_hint[n] == 0) return DSGE_SUCCESS;
    const bool remeasure = !capturing && g_static_hint[n] != 0 && (++g_static_calls[n] & 255) == 0;
    if (g_static_hint[n] == 0 || remeasure) {
      if ((rc = defl_reserve(256, st, &base))) return rc;
      int32_t hmin = n;
      HIP_TRY(hipMemcpyAsync(base, &hmin, sizeof(hmin), hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(dsge::cr_static_scan_kernel, dim3(batch < 4096 ? batch : 4096), dim3(64), 0, st, A, C, batch, n,
                         (int32_t*)base);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(&hmin, base, sizeof(hmin), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      g_static_hint[n] = (remeasure && g_static_hint[n] - 1 < hmin) ? g_static_hint[n] : hmin + 1;
    }
    h = std::min(g_static_hint[n] - 1, (int)dsge::CRD_HMAX);
  }
  const int nd = n - h;
  // worth it only when the reduced system drops to a smaller register-block tile or loses a fifth of its variables
  if (h < 1 || nd < 4 || (tile_bs(nd) == tile_bs(n) && 5 * h < n)) return DSGE_SUCCESS;
  // D_red (nd x k) rides through the reduced cycle reduction as ONE column group of its tile: more shocks than the reduced
  // tile is wide (possible only when most variables are static) do not fit -- full-size solve then
  if (k > 8 * tile_bs(nd)) return DSGE_SUCCESS;
  if (opt().cr_fused_deflation) {  // (launch_cr_fused checks the sizes: h + 3 nd + k <= 192, nd + k <= 64)
    // one launch (dsge_cr_fused.hpp); the (full tile, reduced tile) pairs built are the ones the deflation test above lets
    // through for n <= 48
    int done = 0;
    if ((rc = launch_cr_fused(A, B, C, D, batch, n, k, h, max_iter, tol, T_out, R_out, status, n_iter, st, &done, colmask)))
      return rc;
    if (done) {
      *used = colmask ? 2 : 1;  // 2: colmask[draw] holds the non-zero columns of T (or ~0: not known)
      return DSGE_SUCCESS;
    }
  }
  const size_t lds1 = dsge::crd_deflate_smem(8 * tile_bs(n)), lds2 = dsge::crd_inflate_smem(8 * tile_bs(nd));
  const size_t ndd = (size_t)batch * nd * nd, ndk = (size_t)batch * nd * k, tops = (size_t)batch * dsge::crd_top_doubles(n, k, h);
  if ((rc = defl_reserve(4 * al256(ndd * 8) + 2 * al256(ndk * 8) + al256(tops * 8) + al256((size_t)batch * 4) + 4096, st,
                         &base)))
    return rc;
  char* p = (char*)base;
  double* Ared = (double*)p; p += al256(ndd * 8);
  double* Bred = (double*)p; p += al256(ndd * 8);
  double* Cred = (double*)p; p += al256(ndd * 8);
  double* Tdy = (double*)p; p += al256(ndd * 8);
  double* Dred = (double*)p; p += al256(ndk * 8);
  double* Rdy = (double*)p; p += al256(ndk * 8);
  double* top = (double*)p; p += al256(tops * 8);
  int32_t* flag = (int32_t*)p;
  rc = DSGE_ERR_INVALID;
  DISPATCH_BS(tile_bs(n), 8, {
    rc = DSGE_SUCCESS;
    hipLaunchKernelGGL(dsge::cr_deflate_kernel<BS>, dim3(batch), dim3(64), lds1, st, A, B, C, D, batch, n, k, h, Ared, Bred,
                       Cred, Dred, top, flag);
  });
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  if ((rc = launch_cr(Ared, Bred, Cred, batch, nd, max_iter, tol, Tdy, status, n_iter, st, 0, Dred, k, Rdy))) return rc;
  rc = DSGE_ERR_INVALID;
  DISPATCH_BS(tile_bs(nd), 8, {
    rc = DSGE_SUCCESS;
    hipLaunchKernelGGL(dsge::cr_inflate_kernel<BS>, dim3(batch), dim3(64), lds2, st, (const double*)Tdy, (const double*)Rdy,
                       (const double*)top, (const int32_t*)flag, batch, n, k, h, status, T_out, R_out);
  });
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  // draws the deflation could not take (fewer static variables than the bound, singular R_st): full-size dense kernel on
  // exactly those (status == DSGE_ST_INTERNAL_RERUN)
  {
    const int bs = tile_bs(n);
    rc = DSGE_ERR_INVALID;
    DISPATCH_BS(bs, 8, {
      rc = set_lds(dsge::cr_kernel<BS>, dsge::CrSmem<BS>::bytes);
      if (rc == DSGE_SUCCESS) {
        hipLaunchKernelGGL(dsge::cr_kernel<BS>, dim3(rerun_grid(batch)), dim3(64), dsge::CrSmem<BS>::bytes, st, A, B, C, batch,
                           n, max_iter, tol, T_out, status, n_iter, 1, 0, D, k, R_out);
        HIP_TRY(hipGetLastError());
      }
    });
    if (rc) return rc;
  }
  *used = 1;
  return DSGE_SUCCESS;
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

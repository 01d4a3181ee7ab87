// Launchers of the 65 .. 96-variable kernels (dsge_big.hpp): cycle reduction, selection matrix, and the gather that hands the
// model restricted to its filtered variables to the n <= 64 filter kernels.
#include "dsge_big.hpp"
#include "dsge_host.hpp"

namespace dsge_host {

long long* g_big_dbg = nullptr;  // debug: device int64[8], phase cycles of workgroup 0's first draw (dsge_debug_big_phases)

namespace {
StreamArenaPool g_big_pool;
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
// draws (one workgroup and one 300-440 KB workspace each) per launch of the solver.  About one workgroup per CU is resident
// (156 KB of LDS), so 1024 draws are four residency rounds of the chip: enough to hide a launch's straggler tail, and the
// arena -- per (device, stream), kept for the life of the stream -- stays at 0.33 GB (n <= 80) / 0.46 GB (n <= 96) instead of twice
// that (footprint table in INTEGRATION.md)
constexpr int BIG_GRID_MAX = 1024;
// Threads per draw: the elimination is a chain of short dependent phases separated by barriers, so a step's duration is set by
// instruction latency, not by arithmetic -- more wavefronts per SIMD hide it better.  Measured cycles per pivot step at
// n = 80 / 96 (tools/big_phases.py): 256 threads 4.5 k / 5.3 k; 640 / 512 threads (below) see DESIGN.md.
using Cfg80 = dsge::BigCfg<80, 5, 2>;  // (two block columns = four pivots per round, BigCfg<80, 5, 2, 2>: 272 bytes of scratch at the
                                        //  168-register cap of ten wavefronts, 260 k instead of 241 k cycles per elimination)
using Cfg96 = dsge::BigCfg<96, 6, 3>;
}  // namespace

bool big_size(int n) { return n > 64 && n <= DSGE_MAX_N_BIG; }

int launch_cr_big(const double* A, const double* B, const double* C, int batch, int n, int max_iter, double tol, double* T_out,
                  int32_t* status, int32_t* n_iter, hipStream_t st, int scan_mode, const double* D, int k, double* R_out) {
  if (!big_size(n)) return fail(DSGE_ERR_INVALID, "launch_cr_big: n out of range");
  if (R_out && (!D || k < 1 || k > n)) return fail(DSGE_ERR_INVALID, "launch_cr_big: R_out requires D and 1 <= k <= n");
  const int grid = batch < BIG_GRID_MAX ? batch : BIG_GRID_MAX;
  int rc;
  void* base = nullptr;
  // one workgroup per draw, at most BIG_GRID_MAX draws (= workspaces) per launch
#define BIG_CR(CFG)                                                                                                            \
  do {                                                                                                                         \
    if ((rc = g_big_pool.reserve(al256((size_t)grid * CFG::ws_doubles * 8), st, &base))) return rc;                            \
    if ((rc = set_lds(dsge::cr_big_kernel<CFG>, CFG::lds_bytes))) return rc;                                                   \
    for (int c0 = 0; c0 < batch; c0 += grid) {                                                                                 \
      const int nb = batch - c0 < grid ? batch - c0 : grid;                                                                    \
      const size_t o2 = (size_t)c0 * n * n, ok = (size_t)c0 * n * k;                                                           \
      hipLaunchKernelGGL(dsge::cr_big_kernel<CFG>, dim3(nb), dim3(CFG::NT), CFG::lds_bytes, st, A + o2, B + o2, C + o2, nb, n,  \
                         max_iter, tol, (double*)base, T_out + o2, status + c0, n_iter ? n_iter + c0 : nullptr, scan_mode,     \
                         D ? D + ok : nullptr, k, R_out ? R_out + ok : nullptr, c0 == 0 ? g_big_dbg : nullptr);                \
    }                                                                                                                          \
  } while (0)
  if (n <= 80)
    BIG_CR(Cfg80);
  else
    BIG_CR(Cfg96);
#undef BIG_CR
  HIP_TRY(hipGetLastError());
  return DSGE_SUCCESS;
}

int launch_selection_big(const double* A, const double* B, const double* C, const double* D, const double* T, int batch, int n,
                         int k, double* R_out, double* resid_out, const int32_t* status, hipStream_t st) {
  if (!big_size(n)) return fail(DSGE_ERR_INVALID, "launch_selection_big: n out of range");
  if (k < 1 || k > n) return fail(DSGE_ERR_INVALID, "launch_selection_big: k out of range (1..n)");
  const int grid = batch < BIG_GRID_MAX ? batch : BIG_GRID_MAX;
  int rc;
  void* base = nullptr;
#define BIG_SEL(CFG)                                                                                                        \
  do {                                                                                                                      \
    if ((rc = g_big_pool.reserve(al256((size_t)grid * CFG::ws_doubles * 8), st, &base))) return rc;                         \
    if ((rc = set_lds(dsge::selection_big_kernel<CFG>, CFG::lds_bytes))) return rc;                                         \
    hipLaunchKernelGGL(dsge::selection_big_kernel<CFG>, dim3(grid), dim3(CFG::NT), CFG::lds_bytes, st, A, B, C, D, T, batch, \
                       n, k, (double*)base, R_out, resid_out, status);                                                      \
  } while (0)
  if (n <= 80)
    BIG_SEL(Cfg80);
  else
    BIG_SEL(Cfg96);
#undef BIG_SEL
  HIP_TRY(hipGetLastError());
  return DSGE_SUCCESS;
}

// gensys for 65 .. 96 variables (gensys_certify_big_kernel): the doubling iteration, then the certificate of eu = [1, 1, 0] with
// R = -(B + C T)^-1 D from the same elimination.  T_out: the solvent (zeroed for draws without a verdict).
int launch_gensys_big(const double* A, const double* B, const double* C, const double* D, int batch, int n, int k, double tol,
                      double* T_out, double* R_out, int32_t* eu_out, int32_t* status, int32_t* n_iter, hipStream_t st) {
  if (!big_size(n)) return fail(DSGE_ERR_INVALID, "launch_gensys_big: n out of range");
  if (R_out && (!D || k < 1 || k > n)) return fail(DSGE_ERR_INVALID, "launch_gensys_big: R_out requires D and 1 <= k <= n");
  int rc;
  // (quadratic convergence: the tolerance only decides the last iteration -- as launch_gensys_doubling)
  if ((rc = launch_cr_big(A, B, C, batch, n, 50, 1e-9, T_out, status, n_iter, st, 0, nullptr, 0, nullptr))) return rc;
  const int grid = batch < BIG_GRID_MAX ? batch : BIG_GRID_MAX;
  void* base = nullptr;
#define BIG_CERT(CFG)                                                                                                          \
  do {                                                                                                                         \
    if ((rc = g_big_pool.reserve(al256((size_t)grid * CFG::ws_doubles * 8), st, &base))) return rc;                            \
    if ((rc = set_lds(dsge::gensys_certify_big_kernel<CFG>, CFG::lds_bytes))) return rc;                                       \
    hipLaunchKernelGGL(dsge::gensys_certify_big_kernel<CFG>, dim3(grid), dim3(CFG::NT), CFG::lds_bytes, st, B, C, D, T_out, batch, \
                       n, k, tol, (double*)base, R_out, eu_out, status);                                                       \
  } while (0)
  if (n <= 80)
    BIG_CERT(Cfg80);
  else
    BIG_CERT(Cfg96);
#undef BIG_CERT
  HIP_TRY(hipGetLastError());
  return DSGE_SUCCESS;
}

// F = {state variables} u {observed variables} of the batch, measured on the device (one small launch and a 32-byte read-back:
// the call synchronises the stream here).  *u_out = |F|, *ns_out = number of state variables; idx_out: the sorted members.
int big_filtered_variables(const double* A, const double* Z, int z_batched, int batch, int n, int p, hipStream_t st,
                           unsigned char* idx_out, int* u_out, int* ns_out) {
  int rc;
  void* base = nullptr;
  // (the mask lives behind the solver workspace's first bytes: the solver launch that used them is ordered before this one)
  if ((rc = g_big_pool.reserve(256, st, &base))) return rc;
  unsigned long long h[4] = {0ull, 0ull, 0ull, 0ull};
  HIP_TRY(hipMemsetAsync(base, 0, sizeof(h), st));
  hipLaunchKernelGGL(dsge::big_mask_kernel, dim3(batch < 4096 ? batch : 4096), dim3(128), 0, st, A, Z, z_batched, batch, n, p,
                     (unsigned long long*)base);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(h, base, sizeof(h), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  int u = 0, ns = 0;
  for (int j = 0; j < n; ++j) {
    const bool s = (h[j >> 6] >> (j & 63)) & 1ull, o = (h[2 + (j >> 6)] >> (j & 63)) & 1ull;
    if (s) ++ns;
    if (s || o) {
      if (u < 64) idx_out[u] = (unsigned char)j;
      ++u;
    }
  }
  *u_out = u;
  *ns_out = ns;
  return DSGE_SUCCESS;
}

int launch_big_compress(const double* T, const double* R, const double* Z, int z_batched, int batch, int n, int k, int p,
                        const unsigned char* idx, int u, double* T_r, double* R_r, double* Z_r, hipStream_t st) {
  if (u < 1 || u > 64) return fail(DSGE_ERR_INVALID, "launch_big_compress: u out of range");
  dsge::BigIndex F;
  for (int i = 0; i < 64; ++i) F.idx[i] = i < u ? idx[i] : 0;
  F.u = u;
  hipLaunchKernelGGL(dsge::big_compress_kernel, dim3(batch < 4096 ? batch : 4096), dim3(256), 0, st, T, R, Z, z_batched, batch, n,
                     k, p, F, T_r, R_r, Z_r);
  HIP_TRY(hipGetLastError());
  return DSGE_SUCCESS;
}

}  // namespace dsge_host

// Launcher of the gensys (ordered QZ) kernel.
#include <map>
#include <mutex>
#include <utility>

#include "dsge_host.hpp"
#include "dsge_gensys.hpp"
#include "dsge_gensys_win.hpp"
#include "dsge_gensys_pair.hpp"
#include "dsge_gensys_doubling.hpp"

namespace dsge_host {

int gensys_caps(int n, int n_lead_hint, int* n_cap, int* l_cap);

long long* g_gensys_win_dbg = nullptr;
float* g_gensys_stage_ms = nullptr;  // debug: host float[8]; when set, the window path times its launches with HIP events (synchronising)  // debug: device int64[32], phase stamps of draw 0 of the window kernels

namespace {
// One workspace per (device, stream) (StreamArenaPool): the chunked host path keeps two pipelines in flight on two streams, and
// a workspace shared between them would be overwritten by the next chunk's reduce launch while the previous chunk's QZ /
// post launches still read it.
StreamArenaPool g_gw_pool;
constexpr size_t GW_WORKSPACE_LIMIT = (size_t)1 << 30;  // draws are processed in chunks that keep the workspace below 1 GiB
int gw_reserve(size_t bytes, hipStream_t st, void** out) { return g_gw_pool.reserve(bytes, st, out); }
// gensys by spectral division: iteration counts, marks and the list of draws without a certificate (its own arena: the window
// launches that solve those draws reserve -- and may re-allocate -- the one above)
StreamArenaPool g_gd_pool;

// Capacity record of the window path per (device, n, n_lead_hint): {max #lead, max window, max z, min z} of the batch it was measured on
// (dsge_options.gensys_shape_cache).  gensys_shape_kernel measures it on the first call of a model size (a launch, a 16-byte
// read-back and a stream synchronisation); later calls launch with the cached record as pure enqueues.  A draw that exceeds
// the record is flagged by the reduce launch, solved in the same call by the rescue pass (the single-launch kernel on its
// full-size pencil; the cache is not used for models that kernel cannot hold), and reported through `obs`, which every cached
// call copies to pinned host memory behind its launches without waiting: the next call that sees it drops the record and
// measures again.  The record is also measured afresh every 256 calls, so that it can shrink.
struct ShapeRecord {
  int shape[4] = {0, 0, 0, 0};
  int* d_obs = nullptr;  // device: {flag}: a reduce launch met a draw beyond the record
  int* h_obs = nullptr;  // pinned host mirror of d_obs
  int age = 0;
  bool valid = false;
};
std::mutex g_shape_mu;
std::map<std::pair<int, int>, ShapeRecord> g_shape_cache;

}  // namespace
// dsge_forget_measured_shapes(): the next call of every model size measures its capacity record again (buffers are kept)
void gensys_shape_reset() {
  std::lock_guard<std::mutex> lk(g_shape_mu);
  for (auto& kv : g_shape_cache) kv.second.valid = false;
}
namespace {

// bk != nullptr: eigenvalue mode (reduce + QZ + gensys_bk_kernel instead of the post-processing)
struct BkOut {
  double *re, *im;
  int32_t *n_eig, *n_forward, *n_unstable;
};
int launch_gensys_split(const double* A, const double* B, const double* C, int batch, int n, double tol, double* T_out,
                        int32_t* eu_out, int32_t* status, hipStream_t st, int* used, const BkOut* bk = nullptr,
                        int n_lead_hint = 0, int32_t* key_out = nullptr, const int32_t* act = nullptr) {
  // act (device, {count, draw[0], draw[1], ...}): only those draws are solved, in workspace slots 0 .. count-1 (the launches keep
  // their full grids -- the count is not known on the host -- and the idle workgroups return at once); without a chunk loop
  *used = 0;
  void* base = nullptr;
  int rc = gw_reserve(256, st, &base);
  if (rc) return rc;
  int shape[4] = {0, 0, 0, n};
  int* obs_d = nullptr;
  int* obs_h = nullptr;
  bool cached = false;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  int rescue_ncap = 0, rescue_lcap = 0;
  // Cached only under a caller-supplied bound on the number of lead columns (n_lead_hint > 0; a draw beyond the caller's own
  // bound is flagged with or without the cache, include/dsge_hip.h) and when the rescue pass -- the single-launch kernel sized
  // for that bound -- exists: whatever else a later batch brings (fewer zero columns of A: a larger window) it solves in the
  // same call.  Not in eigenvalue mode.
  bool cacheable = opt().gensys_shape_cache && !bk && n_lead_hint > 0;
  if (cacheable) cacheable = gensys_caps(n, n_lead_hint, &rescue_ncap, &rescue_lcap) == DSGE_SUCCESS;
  if (cacheable) {
    std::lock_guard<std::mutex> lk(g_shape_mu);
    auto it = g_shape_cache.find({dev, n * 128 + n_lead_hint});
    if (it != g_shape_cache.end() && it->second.valid) {
      ShapeRecord& r = it->second;
      // (the periodic renewal synchronises the stream: never while the caller is capturing it into a graph)
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      const bool capturing = (hipStreamIsCapturing(st, &cap) == hipSuccess) && cap != hipStreamCaptureStatusNone;
      if (r.h_obs[0] != 0 || (!capturing && ++r.age >= 256)) {
        r.valid = false;  // an earlier call met a draw beyond the record (or the record is old): measure again
      } else {
        for (int i = 0; i < 4; ++i) shape[i] = r.shape[i];
        obs_d = r.d_obs;
        obs_h = r.h_obs;
        cached = true;
      }
    }
  }
  if (!cached) {
    int* shape_d = (int*)base;  // the first 256 bytes of the arena hold the shape record
    HIP_TRY(hipMemcpyAsync(shape_d, shape, sizeof(shape), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(dsge::gensys_shape_kernel, dim3(batch < 4096 ? batch : 4096), dim3(64), 0, st, A, C, batch, n, tol,
                       shape_d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(shape, shape_d, sizeof(shape), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (cacheable) {
      std::lock_guard<std::mutex> lk(g_shape_mu);
      ShapeRecord& r = g_shape_cache[{dev, n * 128 + n_lead_hint}];
      if (!r.d_obs || !r.h_obs) {  // both or neither: a record is published only with its two buffers
        int* d_new = nullptr;
        int* h_new = nullptr;
        HIP_TRY(hipMalloc((void**)&d_new, 4 * sizeof(int)));
        const hipError_t eh = hipHostMalloc((void**)&h_new, 4 * sizeof(int), hipHostMallocDefault);
        if (eh != hipSuccess) {
          (void)hipFree(d_new);
          (void)hipGetLastError();
          return fail(DSGE_ERR_HIP, std::string("hipHostMalloc(shape record): ") + hipGetErrorString(eh));
        }
        r.d_obs = d_new;
        r.h_obs = h_new;
      }
      // (other streams' cached calls of this size may still be in flight with the old record: they only ever raise the flag)
      HIP_TRY(hipMemsetAsync(r.d_obs, 0, 4 * sizeof(int), st));
      for (int i = 0; i < 4; ++i) r.shape[i] = shape[i], r.h_obs[i] = 0;
      r.age = 0;
      r.valid = true;
    }
  }
  dsge::GwCaps cp;
  cp.n = n;
  // n_lead_hint is a promise (include/dsge_hip.h: "upper bound on the number of lead columns"): a draw with more lead columns
  // than the caller's bound is flagged DSGE_ST_GENSYS_TOO_BIG -- by the single-launch kernel, by the window path with a fresh
  // record and by the window path with a cached one alike
  if (n_lead_hint > 0 && shape[0] > n_lead_hint) shape[0] = n_lead_hint;
  cp.lcap = shape[0] < 1 ? 1 : shape[0];
  cp.wcap = shape[1] < 1 ? 1 : shape[1];
  cp.zcap = shape[2] < 1 ? 1 : shape[2];
  cp.scap = (n - shape[3]) < 1 ? 1 : (n - shape[3]);
  if (n + cp.lcap > DSGE_MAX_N_GENSYS) return DSGE_SUCCESS;
  const size_t lds1 = dsge::gw_reduce_smem(cp), lds2 = dsge::gw_qz_smem(cp), lds3 = dsge::gw_post_smem(cp),
               lds_eu = dsge::gw_eu_smem(cp), lds1b = dsge::gw_reduce2_smem(cp);
  if (lds1 > LDS_LIMIT || lds2 > LDS_LIMIT || lds3 > LDS_LIMIT || lds_eu > LDS_LIMIT || lds1b > LDS_LIMIT)
    return DSGE_SUCCESS;
  const dsge::GwOffsets wo = dsge::gw_offsets(cp);
  const size_t per_draw = wo.total * sizeof(double);
  size_t chunk = GW_WORKSPACE_LIMIT / per_draw;
  if (chunk < 256) chunk = 256;
  if (chunk > (size_t)batch) chunk = (size_t)batch;
  if (act && chunk < (size_t)batch) return DSGE_SUCCESS;  // (the caller falls back to its own pass)
  if ((rc = gw_reserve(256 + chunk * per_draw, st, &base))) return rc;
  double* wsp = (double*)((char*)base + 256);
  if ((rc = set_lds(dsge::gensys_reduce_kernel<2>, lds1))) return rc;
  if ((rc = set_lds(dsge::gensys_hesstri_kernel, lds1b))) return rc;
  if ((rc = set_lds(dsge::gensys_qzwin_kernel, lds2))) return rc;
  if ((rc = set_lds(dsge::gensys_post_kernel, lds3))) return rc;
  if ((rc = set_lds(dsge::gensys_eu_kernel, lds_eu))) return rc;
  const bool pairs = opt().gensys_real_stage && opt().gensys_pairs && dsge::gp_fits(cp);
  const size_t lds_pair = dsge::gp_smem(cp), lds_hpair = dsge::gp_hess_smem(cp);
  const bool hess_pairs = pairs && opt().gensys_pairs >= 2 && lds_hpair <= LDS_LIMIT;
  if (hess_pairs && (rc = set_lds(dsge::gensys_hesstri_pair_kernel, lds_hpair))) return rc;
  if (pairs && (rc = set_lds(dsge::gensys_sweeps_pair_kernel<37>, lds_pair))) return rc;
  if (pairs && (rc = set_lds(dsge::gensys_sweeps_pair_kernel<39>, lds_pair))) return rc;
  const size_t nn = (size_t)n * n;
  // stage timing (dsge_debug_gensys_stage_ms): events between the launches of the FIRST chunk
  float* const stage_ms = bk ? nullptr : g_gensys_stage_ms;
  EventGuard ev[8];
  if (stage_ms)
    for (auto& e : ev) HIP_TRY(e.create());
#define GW_EVENT(i)                                          \
  do {                                                       \
    if (stage_ms && c0 == 0) HIP_TRY(hipEventRecord(ev[i], st)); \
  } while (0)
  for (size_t c0 = 0; c0 < (size_t)batch; c0 += chunk) {
    const int nb = (int)((c0 + chunk <= (size_t)batch) ? chunk : (size_t)batch - c0);
    GW_EVENT(0);
    // two wavefronts per draw: the reflectors' columns in two shares (measured: 595 us per 4096 SW-shaped draws on one wavefront,
    // 499 us on two, 548 us on four -- the chain is the walk down the rows, which more column shares do not shorten)
    hipLaunchKernelGGL(dsge::gensys_reduce_kernel<2>, dim3(nb), dim3(128), lds1, st, A + c0 * nn, B + c0 * nn, C + c0 * nn, nb,
                       cp, tol, wsp, g_gensys_win_dbg, obs_d, act);
    GW_EVENT(1);
    if (pairs) {
      if (hess_pairs)
        hipLaunchKernelGGL(dsge::gensys_hesstri_pair_kernel, dim3((nb + 1) / 2), dim3(64), lds_hpair, st, nb, cp, wsp,
                           g_gensys_win_dbg, act);
      else
        hipLaunchKernelGGL(dsge::gensys_hesstri_kernel, dim3(nb), dim3(64), lds1b, st, nb, cp, wsp, g_gensys_win_dbg, 0, 1, act);
      GW_EVENT(2);
      if (dsge::gp_ld(cp) == 37)
        hipLaunchKernelGGL(dsge::gensys_sweeps_pair_kernel<37>, dim3((nb + 1) / 2), dim3(64), lds_pair, st, nb, cp, wsp,
                           g_gensys_win_dbg, act);
      else
        hipLaunchKernelGGL(dsge::gensys_sweeps_pair_kernel<39>, dim3((nb + 1) / 2), dim3(64), lds_pair, st, nb, cp, wsp,
                           g_gensys_win_dbg, act);
    } else {
      hipLaunchKernelGGL(dsge::gensys_hesstri_kernel, dim3(nb), dim3(64), lds1b, st, nb, cp, wsp, g_gensys_win_dbg,
                         opt().gensys_real_stage, 0, act);
    }
    if (!pairs) GW_EVENT(2);
    GW_EVENT(3);
    hipLaunchKernelGGL(dsge::gensys_qzwin_kernel, dim3(nb), dim3(64), lds2, st, nb, cp, tol, wsp, g_gensys_win_dbg,
                       // The BK eigenvalue report keeps the zhgeqz-style iteration for every block: the reference prints
                       // beta / (alpha + tol) (gEconpy/model/perturbation.py:438), which depends on the SIZE of the Schur
                       // diagonal pair and so on the order in which the roots of a block appear on the diagonal -- the
                       // iteration's order is LAPACK's, the closed form's need not be (a root of modulus 390 printed
                       // 1.2e-7 away from the golden); T, eu and the likelihood do not depend on it.
                       (opt().gensys_real_stage && opt().gensys_direct_blocks && !bk) ? 1 : 0, act);
    GW_EVENT(4);
    if (bk)
      hipLaunchKernelGGL(dsge::gensys_bk_kernel, dim3(nb), dim3(64), 0, st, nb, cp, tol, (const double*)wsp,
                         bk->re + c0 * 2 * n, bk->im + c0 * 2 * n, bk->n_eig + c0, bk->n_forward + c0,
                         bk->n_unstable + c0, status + c0);
    else {
      hipLaunchKernelGGL(dsge::gensys_eu_kernel, dim3(nb), dim3(64), lds_eu, st, nb, cp, tol, wsp, g_gensys_win_dbg, act);
      GW_EVENT(5);
      hipLaunchKernelGGL(dsge::gensys_post_kernel, dim3(nb), dim3(dsge::GW_POST_THREADS), lds3, st, nb, cp, tol, (const double*)wsp,
                         T_out + c0 * nn, eu_out + 3 * c0, status + c0, g_gensys_win_dbg, cached ? 1 : 0,
                         key_out ? key_out + c0 : nullptr, act);
    }
    GW_EVENT(6);
    HIP_TRY(hipGetLastError());
  }
#undef GW_EVENT
  if (stage_ms) {  // [0] reduce, [1] Hessenberg-triangular, [2] real sweeps (pair launch; 0 without it), [3] complex QZ + reordering,
                   // [4] eu (or bk), [5] post, [6] = sum, [7] = draws timed
    HIP_TRY(hipEventSynchronize(ev[6]));
    float tot = 0.f;
    for (int i = 0; i < 6; ++i) {  // event i is recorded in front of stage i, event 6 behind the post launch
      float t = 0.f;
      HIP_TRY(hipEventElapsedTime(&t, ev[i], ev[i + 1]));
      stage_ms[i] = t;
      tot += t;
    }
    stage_ms[6] = tot;
    stage_ms[7] = (float)(chunk < (size_t)batch ? chunk : (size_t)batch);
  }
  if (obs_d) HIP_TRY(hipMemcpyAsync(obs_h, obs_d, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
  if (cached) {
    // rescue pass: draws whose shape exceeds the cached record (flagged DSGE_ST_INTERNAL_RERUN by the post launch) go
    // through the single-launch kernel with its full-size pencil; normally there is none and the launch returns at once
    {
      const size_t lds = dsge::gensys_smem_bytes(n, rescue_ncap, rescue_lcap);
      if ((rc = set_lds(dsge::gensys_kernel, lds))) return rc;
      hipLaunchKernelGGL(dsge::gensys_kernel, dim3(rerun_grid(batch)), dim3(64), lds, st, A, B, C, batch, n, rescue_ncap,
                         rescue_lcap, tol, T_out, eu_out, status, (long long*)nullptr, 1);
    }
    hipLaunchKernelGGL(dsge::gensys_rescue_close_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, batch, status, eu_out);
    HIP_TRY(hipGetLastError());
  }
  *used = 1;
  return DSGE_SUCCESS;
}
}  // namespace

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

int launch_gensys_bk(const double* A, const double* B, const double* C, int batch, int n, double tol, double* eig_re,
                     double* eig_im, int32_t* n_eig, int32_t* n_forward, int32_t* n_unstable, int32_t* status,
                     hipStream_t st) {
  BkOut bk{eig_re, eig_im, n_eig, n_forward, n_unstable};
  int used = 0;
  int rc = launch_gensys_split(A, B, C, batch, n, tol, nullptr, nullptr, status, st, &used, &bk);
  if (rc) return rc;
  if (!used) return fail(DSGE_ERR_INVALID, "bk eigenvalues: the pencil does not fit the on-chip window path");
  return DSGE_SUCCESS;
}

// gensys by spectral division (dsge_gensys_doubling.hpp, dsge_options.gensys_doubling): cycle reduction for every draw, a
// certificate of eu = [1, 1, 0] per draw, the ordered QZ (single-launch kernel, flagged draws only) for everything else.
// *done = 0: not applicable (sizes) -- the caller runs the QZ path on the whole batch.
// Round 6, `ov` (optional): the VERDICT -- certificate, compaction, the ordered QZ of the draws without a certificate -- runs on a
// second stream next to what the caller enqueues behind this call on `st` (the filter of every draw with the doubling iteration's T
// and R): ov->st is forked from `st` behind the iteration, works on ov->status (a copy of the iteration's status words: the filter
// reads and writes `status` meanwhile) and leaves in ov->marks which draws it re-solved.  The caller joins, merges the two status
// arrays and re-runs the filter on the marked draws (dsge_api.hip::pipeline).
static int launch_gensys_doubling(const double* A, const double* B, const double* C, const double* D, int k, double* R_tmp,
                                  int batch, int n, double tol, int n_lead_hint, int n_state_hint, double* T_out, int32_t* eu_out,
                                  int32_t* status, hipStream_t st, int32_t* key_out, int* key_written, int* done,
                                  const int32_t** qz_marks, GensysOverlap* ov) {
  *done = 0;
  if (qz_marks) *qz_marks = nullptr;
  if (ov) ov->used = 0;
  int rc;
  int rescue_ncap = 0, rescue_lcap = 0;
  if (gensys_caps(n, n_lead_hint, &rescue_ncap, &rescue_lcap) != DSGE_SUCCESS) return DSGE_SUCCESS;  // no QZ fall-back: not here
  const int bs = tile_bs(n);
  int lcap = n_lead_hint > 0 ? n_lead_hint : n;
  if (lcap > n) lcap = n;
  if (lcap > 32) lcap = 32;
  int scap = (n_state_hint > 0 && n_state_hint < n) ? n_state_hint : n;
  size_t lds = 0;
  DISPATCH_BS(bs, 8, { lds = dsge::gd_lds_doubles<BS>(n, lcap, scap) * sizeof(double); });
  if (lds == 0 || lds > LDS_LIMIT) return DSGE_SUCCESS;
  void* base = nullptr;
  const size_t ints = ((size_t)(batch + 1) * sizeof(int32_t) + 255) / 256 * 256;
  if ((rc = g_gd_pool.reserve(3 * ints, st, &base))) return rc;
  int32_t* it = (int32_t*)base;
  int32_t* mark = (int32_t*)((char*)base + ints);
  int32_t* act = (int32_t*)((char*)base + 2 * ints);  // {count, draws without a certificate in ascending order}
  // the doubling iteration: quadratic convergence, so the stopping tolerance only decides the LAST iteration (T is then
  // accurate to the product of the last iterate's norms, < 1e-18)
  const int max_iter = 50;
  const double tol_cr = 1e-9;
  int deflated = 0;
  if (D && R_tmp && k >= 1) {
    if ((rc = launch_cr_deflated(A, B, C, D, batch, n, k, max_iter, tol_cr, T_out, R_tmp, status, it, st, &deflated, nullptr))) return rc;
  }
  if (!deflated) {
    if ((rc = launch_cr(A, B, C, batch, n, max_iter, tol_cr, T_out, status, it, st))) return rc;
  }
  // the verdict's stream and status words: the caller's own, or the fork (only behind the one-launch iteration, whose R the filter
  // can use at once)
  hipStream_t vst = st;
  int32_t* vstatus = status;
  if (ov && ov->st && ov->status && ov->fork && deflated) {
    HIP_TRY(hipMemcpyAsync(ov->status, status, (size_t)batch * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    if (key_out) {  // (the filter's dispatch key is needed on the caller's stream, before the verdict)
      HIP_TRY(hipMemcpyAsync(key_out, it, (size_t)batch * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
      if (key_written) *key_written = 1;
    }
    HIP_TRY(hipEventRecord(ov->fork, st));
    HIP_TRY(hipStreamWaitEvent(ov->st, ov->fork, 0));
    vst = ov->st;
    vstatus = ov->status;
    ov->used = 1;
    ov->marks = mark;
  }
  rc = DSGE_ERR_INVALID;
  DISPATCH_BS(bs, 8, {
    rc = set_lds(dsge::gensys_certify_kernel<BS>, lds);
    if (rc == DSGE_SUCCESS)
      hipLaunchKernelGGL(dsge::gensys_certify_kernel<BS>, dim3(batch), dim3(64), lds, vst, B, C, (const double*)T_out, batch, n, lcap,
                         scap, tol, eu_out, vstatus, mark);
  });
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  // The ordered QZ for the draws without a certificate.  Window path on the compacted list (its launches keep the grid of the
  // whole batch -- the count stays on the device -- and the idle workgroups return at once: 0.03 ms when nothing is flagged, the QZ
  // path's own rate when everything is); the single-launch kernel where the window path does not apply.
  int used = 0;
  if (opt().gensys_split && opt().gensys_doubling != 2) {
    hipLaunchKernelGGL(dsge::gensys_compact_kernel, dim3(1), dim3(1024), 0, vst, (const int32_t*)mark, batch, act);
    HIP_TRY(hipGetLastError());
    if ((rc = launch_gensys_split(A, B, C, batch, n, tol, T_out, eu_out, vstatus, vst, &used, nullptr, n_lead_hint, nullptr, act)))
      return rc;
  }
  if (!used) {
    const size_t lds_q = dsge::gensys_smem_bytes(n, rescue_ncap, rescue_lcap);
    if ((rc = set_lds(dsge::gensys_kernel, lds_q))) return rc;
    hipLaunchKernelGGL(dsge::gensys_kernel, dim3(rerun_grid(batch)), dim3(64), lds_q, vst, A, B, C, batch, n, rescue_ncap, rescue_lcap,
                       tol, T_out, eu_out, vstatus, (long long*)nullptr, 1);
    hipLaunchKernelGGL(dsge::gensys_rescue_close_kernel, dim3((batch + 255) / 256), dim3(256), 0, vst, batch, vstatus, eu_out);
    HIP_TRY(hipGetLastError());
  }
  if (key_out && vst == st) {  // Kalman dispatch key: the iteration count (grows with the persistence of the model, like the QZ-spectrum key)
    HIP_TRY(hipMemcpyAsync(key_out, it, (size_t)batch * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    if (key_written) *key_written = 1;
  }
  if (qz_marks && deflated) *qz_marks = mark;  // (R_tmp holds R of the certified draws only when the fused cycle reduction ran)
  *done = 1;
  return DSGE_SUCCESS;
}

int launch_gensys(const double* A, const double* B, const double* C, int batch, int n, double tol, int n_lead_hint,
                  double* T_out, int32_t* eu_out, int32_t* status, hipStream_t st, long long* dbg, int32_t* key_out,
                  int* key_written, const double* D, int k, double* R_tmp, int n_state_hint, const int32_t** qz_marks,
                  GensysOverlap* ov) {
  if (key_written) *key_written = 0;
  if (qz_marks) *qz_marks = nullptr;
  if (ov) ov->used = 0;
  int rc;
  if (opt().gensys_doubling && !dbg) {
    int done = 0;
    if ((rc = launch_gensys_doubling(A, B, C, D, k, R_tmp, batch, n, tol, n_lead_hint, n_state_hint, T_out, eu_out, status, st,
                                     key_out, key_written, &done, qz_marks, opt().gensys_doubling == 1 ? ov : nullptr)))
      return rc;
    if (done) return DSGE_SUCCESS;
  }
  // Small pencils (<= 24 KB of LDS in the single-launch kernel, i.e. >= 6 draws per CU already) gain nothing from the
  // window path and would pay for its three launches and the shape read-back: RBC-sized models stay on one launch.
  bool small = false;
  {
    int nc = 0, lc = 0;
    if (gensys_caps(n, n_lead_hint, &nc, &lc) == DSGE_SUCCESS) small = dsge::gensys_smem_bytes(n, nc, lc) <= 24 * 1024;
  }
  if (opt().gensys_split && !dbg && (!small || opt().gensys_split == 2)) {
    int used = 0;
    if ((rc = launch_gensys_split(A, B, C, batch, n, tol, T_out, eu_out, status, st, &used, nullptr, n_lead_hint, key_out)))
      return rc;
    if (used) {
      if (key_written && key_out) *key_written = 1;
      return DSGE_SUCCESS;
    }
  }
  int n_cap = 0, l_cap = 0;
  rc = gensys_caps(n, n_lead_hint, &n_cap, &l_cap);
  if (rc) return rc;
  const size_t lds = dsge::gensys_smem_bytes(n, n_cap, l_cap);
  if ((rc = set_lds(dsge::gensys_kernel, lds))) return rc;
  hipLaunchKernelGGL(dsge::gensys_kernel, dim3(batch), dim3(64), lds, st, A, B, C, batch, n, n_cap, l_cap, tol, T_out,
                     eu_out, status, dbg, 0);
  HIP_TRY(hipGetLastError());
  return DSGE_SUCCESS;
}

// gensys on a caller-supplied pencil (dsge_gensys_pencil_batched): single-launch kernel, everything in LDS
int launch_gensys_pencil(const double* g0, const double* g1, const double* c, const double* psi, const double* pi, int batch,
                         int N, int k, int ell, double tol, double* G1_out, double* C_out, double* impact_out,
                         double* gev_out, int32_t* eu_out, int32_t* status, hipStream_t st,
                         const dsge_gensys_forward* fw) {
  const size_t lds = dsge::gensys_pencil_smem_bytes(N, ell, ell + k + 1);
  if (lds > LDS_LIMIT)
    return fail(DSGE_ERR_TOO_LARGE, "gensys pencil: H, T, Z (N x N complex each) and Q [Pi | Psi | c] do not fit the 160 KB LDS "
                                    "(N <= ~52)");
  int rc;
  if ((rc = set_lds(dsge::gensys_pencil_kernel, lds))) return rc;
  dsge::GensysFwdOut fo{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  if (fw) fo = dsge::GensysFwdOut{fw->f_mat, fw->f_wt, fw->y_wt, fw->loose, fw->n_unstable, fw->pi_raw};
  hipLaunchKernelGGL(dsge::gensys_pencil_kernel, dim3(batch), dim3(64), lds, st, g0, g1, c, psi, pi, batch, N, k, ell, tol,
                     G1_out, C_out, impact_out, gev_out, eu_out, status, fo);
  HIP_TRY(hipGetLastError());
  return DSGE_SUCCESS;
}

// After the overlapped verdict (launch_gensys_doubling with `ov`): the status words of the draws the verdict re-solved.  A draw the
// ordered QZ solved is handed to the filter's second pass (DSGE_ST_INTERNAL_RERUN); a draw it rejected keeps the QZ's status and
// gets logp = -inf (what the filter's first pass writes for a failed draw); unmarked draws keep what the first filter pass left.
__global__ __launch_bounds__(256) void gensys_overlap_merge_kernel(int batch, const int32_t* __restrict__ marks,
                                                                   const int32_t* __restrict__ vstatus, int32_t* __restrict__ status,
                                                                   double* __restrict__ logp) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= batch || marks[i] == 0) return;
  const int32_t v = vstatus[i];
  if (v == 0) {
    status[i] = dsge::DSGE_ST_INTERNAL_RERUN;
  } else {
    status[i] = v;
    logp[i] = -INFINITY;
  }
}

int launch_gensys_overlap_merge(int batch, const int32_t* marks, const int32_t* vstatus, int32_t* status, double* logp,
                                hipStream_t st) {
  hipLaunchKernelGGL(gensys_overlap_merge_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, batch, marks, vstatus, status, logp);
  HIP_TRY(hipGetLastError());
  return DSGE_SUCCESS;
}

}  // namespace dsge_host

// Host-side plumbing shared by the translation units of libdsge_hip.so: error reporting, the tile-size
// dispatch and the prototypes of the kernel launchers.  Each launch_*.hip file instantiates only its
// own kernels, so the library builds as independent (parallel) hipcc jobs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/dsge_hip.h"
#include "dsge_filter_conv.hpp"

namespace dsge_host {

int fail(int code, const std::string& msg);  // records the message for dsge_last_error(), returns code

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      (void)hipGetLastError(); /* clear the sticky error so later calls are not poisoned */        \
      return dsge_host::fail(DSGE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));     \
    }                                                                                              \
  } while (0)

constexpr size_t LDS_LIMIT = 160 * 1024;

// Library-owned device scratch: ONE arena per (device, stream) and pool -- the library is re-entrant per stream:
// two calls enqueued on two streams never share intermediates.  Slots are created on demand (a host thread's twin streams
// release theirs when the thread exits: stream_arenas_release); beyond MAX_SLOTS live (device, stream) pairs the least
// recently used slot is recycled after a device-wide synchronisation, and it changes owner -- the former stream gets a
// fresh slot on its next call, so no two streams ever hold the same memory.
class StreamArenaPool {
 public:
  StreamArenaPool();
  int reserve(size_t bytes, hipStream_t st, void** out);  // the arena of (current device, st), grown to >= bytes
  void release(hipStream_t st);                            // the stream is going away: its slots become free
 private:
  struct Slot {
    void* ptr = nullptr;
    size_t cap = 0;
    int dev = -1;
    hipStream_t stream = nullptr;
    bool used = false;
    unsigned long long stamp = 0;
  };
  static constexpr size_t MAX_SLOTS = 64;
  std::vector<Slot> slots_;
  std::mutex mu_;
  unsigned long long clock_ = 0;
};
void stream_arenas_release(hipStream_t st);  // every pool of the library (dsge_api.hip)
// The two streams the host twins of the CALLING THREAD run on (created on first use per device, destroyed -- and their
// arenas released -- when the thread exits): two host threads in two twins never share a stream, hence never an arena.
int twin_streams(hipStream_t* s0, hipStream_t* s1);

// hipEvent_t that destroys itself: stage-timing events must not leak on the early returns of HIP_TRY
struct EventGuard {
  hipEvent_t e = nullptr;
  EventGuard() = default;
  EventGuard(const EventGuard&) = delete;
  EventGuard& operator=(const EventGuard&) = delete;
  ~EventGuard() {
    if (e) (void)hipEventDestroy(e);
  }
  hipError_t create() { return hipEventCreate(&e); }
  operator hipEvent_t() const { return e; }
};

inline int tile_bs(int n) {
  int bs = (n + 7) / 8;
  return bs < 1 ? 1 : bs;
}

// Grid of a second pass that only takes the draws an earlier kernel flagged (normally none): the workgroups loop over the
// draws, so a quarter of the batch-sized grid costs a few microseconds less per empty pass and still fills the chip.
inline int rerun_grid(int batch) { return batch < 1024 ? batch : 1024; }

template <typename K>
int set_lds(K kernel, size_t bytes) {
  HIP_TRY(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return DSGE_SUCCESS;
}

#define DISPATCH_BS(bs, MAXBS, ...)                                                  \
  switch (bs) {                                                                      \
    case 1: { constexpr int BS = 1; __VA_ARGS__; } break;                            \
    case 2: { constexpr int BS = 2; __VA_ARGS__; } break;                            \
    case 3: { constexpr int BS = 3; __VA_ARGS__; } break;                            \
    case 4: { constexpr int BS = 4; __VA_ARGS__; } break;                            \
    case 5: { constexpr int BS = 5; __VA_ARGS__; } break;                            \
    case 6: { constexpr int BS = 6; __VA_ARGS__; } break;                            \
    case 7: if (MAXBS >= 7) { constexpr int BS = (MAXBS >= 7 ? 7 : 6); __VA_ARGS__; } break; \
    case 8: if (MAXBS >= 8) { constexpr int BS = (MAXBS >= 8 ? 8 : 6); __VA_ARGS__; } break; \
    default: break;                                                                  \
  }

// ---- launchers (launch_solvers.hip, launch_assemble.hip, launch_kalman.hip, launch_gensys.hip) ----
int launch_cr(const double* A, const double* B, const double* C, int batch, int n, int max_iter, double tol,
              double* T_out, int32_t* status, int32_t* n_iter, hipStream_t st, int scan_mode = 0,
              const double* D = nullptr, int k = 0, double* R_out = nullptr);  // D, R_out: also R = -A1_hat^-1 D
// launch_big.hip (dsge_big.hpp): models with 65 .. DSGE_MAX_N_BIG variables, one workgroup per draw
bool big_size(int n);
int launch_cr_big(const double* A, const double* B, const double* C, int batch, int n, int max_iter, double tol, double* T_out,
                  int32_t* status, int32_t* n_iter, hipStream_t st, int scan_mode, const double* D, int k, double* R_out);
int launch_gensys_big(const double* A, const double* B, const double* C, const double* D, int batch, int n, int k, double tol,
                      double* T_out, double* R_out, int32_t* eu_out, int32_t* status, int32_t* n_iter, hipStream_t st);
int launch_selection_big(const double* A, const double* B, const double* C, const double* D, const double* T, int batch, int n,
                         int k, double* R_out, double* resid_out, const int32_t* status, hipStream_t st);
int big_filtered_variables(const double* A, const double* Z, int z_batched, int batch, int n, int p, hipStream_t st,
                           unsigned char* idx_out, int* u_out, int* ns_out);
int launch_big_compress(const double* T, const double* R, const double* Z, int z_batched, int batch, int n, int k, int p,
                        const unsigned char* idx, int u, double* T_r, double* R_r, double* Z_r, hipStream_t st);
int launch_cr_deflated(const double* A, const double* B, const double* C, const double* D, int batch, int n, int k,
                       int max_iter, double tol, double* T_out, double* R_out, int32_t* status, int32_t* n_iter,
                       hipStream_t st, int* used, unsigned long long* colmask = nullptr);
// (*used = 2: the one-launch kernel ran and colmask[draw] holds the non-zero columns of T_out[draw], ~0 = not known)  // static-variable deflation + cycle reduction on the reduced system
void cr_deflation_reset();
void gensys_shape_reset();  // launch_gensys.hip
int launch_bdirect(const double* A, const double* B, const double* D, int batch, int n, int k, double* T_out,
                   double* R_out, hipStream_t st);
int launch_rqr(const double* R, const double* q, int q_batched, int batch, int n, int k, const int32_t* status,
               double* RQR_out, hipStream_t st, int rerun_only = 0);  // sym(R diag(q) R') alone, k <= RQR_KMAX (dsge_kernels.hpp);
                                                                      // rerun_only: draws flagged DSGE_ST_INTERNAL_RERUN
int launch_assemble(const double* A, const double* B, const double* C, const double* D, const double* T,
                    const double* R_in, const double* Q, int q_mode, int batch, int n, int k, double* R_out,
                    double* resid_out, double* RQR_out, double* P0_out, int32_t* status, int do_sel, int do_lyap,
                    hipStream_t st, const int32_t* only_marked = nullptr);  // only_marked: draws with a non-zero mark (selection only)
int launch_dense_z_augment(const double* T, const double* R, const double* Z, int z_batched, int batch, int n, int k, int p,
                           double* T_aug, double* R_aug, double* Z_aug, hipStream_t st);
int launch_dense_z_deaugment(const double* Tbar_a, const double* Gbar_a, const double* T, const double* G_aug, const double* Z,
                             int z_batched, const int32_t* status, int batch, int n, int p, double* Tbar, double* Gbar,
                             double* Z_bar, hipStream_t st);
int launch_status_park(int32_t* status, int32_t* park, int batch, int restore, hipStream_t st);
int launch_adjoint(const double* B, const double* C, const double* T, const double* Tbar, int batch, int n, double* Ab,
                   double* Bb, double* Cb, int32_t* status, hipStream_t st, int accumulate = 0, int only_flag = 0);
int launch_adjoint_fused(const double* B, const double* C, const double* T, const double* R, const double* q, int q_batched,
                         const double* Gbar, double* Tbar, int batch, int n, int k, double* Ab, double* Bb, double* Cb, double* Db,
                         double* qb, int32_t* status, hipStream_t st);
int launch_norms(const double* A, const double* B, const double* C, const double* D, const double* T, const double* R,
                 const int32_t* mask, int batch, int n, int k, double* det, double* sto, hipStream_t st);
int launch_augment(const double* T, const double* R, int batch, int n, int k, int m, const int32_t* inv_var_order,
                   int n_links, const int32_t* link_rows, const int32_t* link_cols, double* T_aug, double* R_aug,
                   hipStream_t st);
int launch_acf(const double* T, const double* Sigma, const double* Z, const double* Hdiag, int batch, int m, int p,
               int n_lags, int lag_step, int correlation, double* out, const int32_t* status, hipStream_t st);
int launch_kalman(const double* T, double* RQR, double* P0, int p0_valid, const double* Z, int z_batched,
                  const double* d, int d_batched, const double* Hdiag, int h_batched, const double* y, int batch,
                  int m, int p, int T_len, double jitter, double missing_fill, int n_state_hint, int z_selector_hint,
                  double* logp, int32_t* status, hipStream_t st, const int32_t* order_key = nullptr,
                  const double* Rsel = nullptr, const double* qdiag = nullptr, int q_batched = 0, int k_shocks = 0,
                  const unsigned long long* colmask = nullptr,
                  int rerun_all = 0);  // 1: every launch is a second pass -- only the draws flagged DSGE_ST_INTERNAL_RERUN are filtered
// launch_kalman_mf.hip: the tile-layout filter kernel (dsge_kalman_mf.hpp) in front of launch_kalman's cascade
int launch_kalman_mf(const double* T, const double* RQR, const double* P0, const double* Z, int z_batched, const double* d,
                     int d_batched, const double* Hdiag, int h_batched, const double* y, int batch, int m, int p, int T_len,
                     dsge::FilterConv cv, double missing_fill, int n_state_hint, double* logp, int32_t* status, hipStream_t st,
                     const int32_t* order, const double* Rsel, const double* qdiag, int q_batched, int k_shocks,
                     const unsigned long long* colmask, int rerun_first, int* launched, bool* covers);
int launch_kalman_outputs(const double* T, const double* RQR, const double* P0, const double* Z, int z_batched, const double* d,
                          int d_batched, const double* Hdiag, int h_batched, const double* y, int batch, int m, int p, int T_len,
                          double jitter, double missing_fill, double* ll, double* a_pred, double* a_filt, double* p_pred,
                          double* p_filt, int full_cov, int32_t* status, hipStream_t st);
// true if launch_kalman, given the selection matrix R and a diagonal Q (Rsel, qdiag), forms sym(R Q R')[U,U] inside the
// fast filter kernel: the caller then skips the full-size product (RQR is filled for handed-on draws only)
bool kalman_folds_rqr(int m, int p, int k, int n_state_hint, int z_selector_hint);
// launch_grad.hip: reverse sweep of the Kalman filter + reverse of the assembly (dsge_kalman_grad.hpp)
int launch_persistence_key(const double* T, const int32_t* status, int batch, int n, int32_t* key, hipStream_t st);
int launch_kalman_grad(const double* T, const double* RQR, const double* Z, int z_batched, const double* d, int d_batched,
                       const double* Hdiag, int h_batched, const double* y, int batch, int m, int p, int T_len,
                       double jitter, double missing_fill, int u_hint, double* store, double* logp, int32_t* status,
                       double* Tbar, double* Gbar, double* dbar, double* hbar, hipStream_t st,
                       int32_t* order_key = nullptr, int32_t* order_buf = nullptr);
size_t kalman_grad_store_doubles_per_draw(int u_hint, int m, int T_len);
int launch_grad_assemble(const double* B, const double* C, const double* T, const double* R, const double* q,
                         int q_batched, const double* Gbar, int batch, int n, int k, const int32_t* status, double* Tbar,
                         double* B_bar, double* C_bar, double* D_bar, double* q_bar, hipStream_t st,
                         const double* Rbar_in = nullptr, int only_flag = 0);  // Rbar_in: pullback of R = -(C T + B)^-1 D alone (Tbar written)
int gensys_caps(int n, int n_lead_hint, int* n_cap, int* l_cap);
// gensys by spectral division with the verdict next to the filter (round 6; launch_gensys.hip::launch_gensys_doubling)
struct GensysOverlap {
  hipStream_t st = nullptr;       // in: the verdict's stream (a library stream of the calling thread)
  hipEvent_t fork = nullptr;      // in: event the verdict stream waits for (recorded on the caller's stream behind the iteration)
  int32_t* status = nullptr;      // in: [batch] status words the verdict works on
  const int32_t* marks = nullptr; // out: [batch], non-zero = the verdict re-solved (or rejected) the draw
  int used = 0;                   // out: 1 = the verdict was forked
};
int launch_gensys_overlap_merge(int batch, const int32_t* marks, const int32_t* vstatus, int32_t* status, double* logp,
                                hipStream_t st);
int launch_gensys(const double* A, const double* B, const double* C, int batch, int n, double tol, int n_lead_hint,
                  double* T_out, int32_t* eu_out, int32_t* status, hipStream_t st, long long* dbg = nullptr,
                  int32_t* key_out = nullptr, int* key_written = nullptr,  // key_out: Kalman dispatch key from the QZ spectrum
                                                                           // (window path only: *key_written tells)
                  const double* D = nullptr, int k = 0, double* R_tmp = nullptr, int n_state_hint = 0,
                  const int32_t** qz_marks = nullptr,  // *qz_marks: [batch], non-zero = the ordered QZ solved the draw (R_tmp is not its R)
                  GensysOverlap* ov = nullptr);
// (D, k, R_tmp, n_state_hint: only for dsge_options.gensys_doubling -- with them the doubling iteration runs as the one-launch
//  deflated cycle reduction, whose final elimination needs a right-hand side; R_tmp is scratch, [batch][n][k])

int launch_gensys_pencil(const double* g0, const double* g1, const double* c, const double* psi, const double* pi, int batch,
                         int N, int k, int ell, double tol, double* G1_out, double* C_out, double* impact_out,
                         double* gev_out, int32_t* eu_out, int32_t* status, hipStream_t st,
                         const dsge_gensys_forward* fw = nullptr);

int launch_gensys_bk(const double* A, const double* B, const double* C, int batch, int n, double tol, double* eig_re,
                     double* eig_im, int32_t* n_eig, int32_t* n_forward, int32_t* n_unstable, int32_t* status,
                     hipStream_t st);

// launch_second_order.hip (dsge_second_order.hpp); S, L, U: host index lists
int launch_second_order(const double* B, const double* C, const double* T, const double* R, const int32_t* hess_idx, int nnz,
                        const double* hess_val, const double* q, int q_batched, const double* Z, const double* d,
                        const double* Hdiag, const double* y, int batch, int n, int k, int p, int T_len, double jitter,
                        double missing_fill, const int32_t* S, int s, const int32_t* L, int l, const int32_t* U, int u,
                        double* logp, int32_t* status_io, double* gyy_out, double* gyu_out, double* guu_out, double* gss_out,
                        int32_t* steady_at, int32_t* n_doublings, hipStream_t st, float* ms, const int32_t* order_key = nullptr);

extern int g_adj_refine_mode;          // launch_assemble.hip: 0 = residual rule, 1 = refine every draw, 2 = never (debug)
extern long long* g_so_dbg;            // launch_second_order.hip: debug phase counters of the second-order filter kernel
extern long long* g_cr_dbg;            // launch_solvers.hip: debug phase counters of the compact CR kernel
extern long long* g_big_dbg;          // launch_big.hip: debug phase cycles of cr_big_kernel
extern long long* g_kalman_dbg;       // launch_kalman.hip: debug buffer for per-phase cycles of draw 0
extern long long* g_gensys_win_dbg;   // launch_gensys.hip: debug phase stamps of the window kernels (device int64[32])
extern float* g_gensys_stage_ms;      // launch_gensys.hip: debug, host float[8]: launch durations of the window path (dsge_debug_gensys_stage_ms)
extern long long* g_kalman_timeline;  // debug: device int64 [batch][8] {start, end, HW_ID, steady step} per draw (kalman_nt_kernel)
extern int32_t* g_kalman_steady_at;   // debug: device buffer [batch], first steady step per draw (-1 = never)

// Kernel-variant switches.  They are PER CALL: the *_opt entry points carry a dsge_options, which an RAII guard installs
// for the duration of the call on the calling thread (launching is synchronous on the host, every kernel argument is
// passed by value at launch), so two host threads -- two PyMC chains, two streams -- never see each other's settings.
// Calls without options use the compiled-in defaults (g_defaults is never written: ABI 8 removed the dsge_set_* setters).
struct Options {
  int cr_compact = 1;          // 0 = dense cycle-reduction kernel only
  int cr_fused_selection = 1;  // fused pipeline: R from the cycle-reduction kernel's final elimination
  int cr_deflation = 1;        // static-variable deflation in front of cycle reduction
  int cr_two_waves = 1;        // 4 x 4-tile compact kernel built for two waves per SIMD
  int n_static_hint = -1;      // static variables (zero columns of A and C): -1 = measure on the device, >= 0 caller's bound
  int kalman_order = 1;        // Kalman workgroups slow-draws-first (1 = CR iteration count / persistence key, 2 = key, 0 = index)
  int kalman_tiny = 1;         // thread-per-draw kernel for small models
  int kalman_block = 0;        // steady tail handed to kalman_tail_kernel
  int kalman_mfma = 2;         // prediction products on the FP64 matrix core: 2 = 4 x 4 x 4 blocks in the NT kernel (round 6), 1 = 16 x 16 x 4 (round 2, slower), 0 = VALU
  int cr_four_waves = 1;       // n = 49..64: cr_wide_kernel (256 threads per draw) instead of cr_compact_kernel<7|8>
  int cr_fused_deflation = 1;  // deflation + cycle reduction + inflation in one launch (dsge_cr_fused.hpp)
  int kalman_nt_products = 1;  // selector fast path: kalman_nt_kernel (NT prediction products, 16-byte LDS loads); 0 = kalman_sel_kernel
  int pipeline_chunks = 0;     // fused device call in chunks over library-owned streams
  int gensys_split = 1;        // 0 = single-launch gensys kernel, 1 = window path unless small, 2 = always
  int gensys_real_stage = 1;   // window path: real double-shift sweeps in front of the complex single-shift iteration
  double kalman_steady_tol = 1e-14;  // steady-state switch of the fast Kalman kernel (0 = never)
  int gensys_pairs = 1;        // window path: two draws per wavefront in the real double-shift sweeps (dsge_gensys_pair.hpp)
  int gensys_shape_cache = 1;  // window path: capacity record measured once per model size
  int gensys_direct_blocks = 1;  // window path: isolated 2 x 2 blocks triangularised in closed form in front of the complex iteration
  int kalman_narrow = 1;       // fast filter: the SK = 20 instance of the 32-wide tile when the state block fits
  int gensys_doubling = 1;     // gensys by spectral division: cycle reduction + certificate, ordered QZ only for uncertified draws (0: QZ for all)
  int kalman_grad_split = 2;   // gradient: forward sweep by a logp kernel with record output, reverse sweep by kalman_grad_kernel<BS, true>; 2: + kalman_grad_tail_kernel
  int grad_fused_adjoint = 1;  // gradient pipeline: reverse of the assembly + policy adjoints in one launch (internal; DSGE_GRAD_FUSED_ADJOINT=0 switches it off)
  int kalman_head_draws = 0;   // fast filter: this many draws at the head of the dispatch order on the two-wavefront kernel (-1 = all)
  // conventions of the filter step (third party: pymc_extras; include/dsge_hip.h "Filter conventions")
  int ll_constant = DSGE_LL_CONST_P;
  int mask_d = 0;
  int joseph = 1;
  double jitter_F = -1.0;      // < 0: the call's `jitter` argument
  double jitter_P = -1.0;
};
extern const Options g_defaults;
extern thread_local const Options* t_call_options;
inline const Options& opt() { return t_call_options ? *t_call_options : g_defaults; }
// the conventions of the filter step for a call whose `jitter` argument is given: what every filter kernel receives by value
inline dsge::FilterConv filter_conv(double jitter) {
  const Options& o = opt();
  dsge::FilterConv cv;
  cv.jit_F = (o.jitter_F >= 0.0) ? o.jitter_F : jitter;
  cv.jit_P = (o.jitter_P >= 0.0) ? o.jitter_P : jitter;
  cv.jit_V = o.joseph ? cv.jit_F : 0.0;
  cv.ll_mode = o.ll_constant;
  cv.mask_d = o.mask_d ? 1 : 0;
  return cv;
}

}  // namespace dsge_host

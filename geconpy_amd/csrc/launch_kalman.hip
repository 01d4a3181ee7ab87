// Launcher of the Kalman log-likelihood kernels (fast-path tile cascade + general kernel).
#include <mutex>

#include <type_traits>

#include "dsge_host.hpp"
#include "dsge_kalman_out.hpp"
#include "dsge_kernels.hpp"
#include "dsge_kalman2.hpp"
#include "dsge_kalman_nt.hpp"
#include "dsge_kalman_nt2.hpp"
#include "dsge_kalman_tail.hpp"
#include "dsge_kalman_tiny.hpp"

namespace dsge_host {

long long* g_kalman_dbg = nullptr;  // debug: device buffer for per-phase cycles of draw 0
int32_t* g_kalman_steady_at = nullptr;
long long* g_kalman_timeline = nullptr;  // debug: device int64 [batch][8], see kalman_nt_kernel
                       // (experimental: measured SLOWER than the VALU register blocks)

namespace {
// The stream the two-wavefront head launch runs on, next to the caller's stream (fork / join by events): one per host thread and
// device, created on first use.  Two calls of one thread on two caller streams share it -- their heads then run one after the
// other, which is correct (events order them) and rare.
struct HeadStream {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  int dev = -1;
  void release() {  // (idempotent; the owning device is made current for the destroy calls and the caller's restored)
    if (!s && !fork && !join) return;
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    if (dev >= 0) (void)hipSetDevice(dev);
    if (join) (void)hipEventDestroy(join);
    if (fork) (void)hipEventDestroy(fork);
    if (s) (void)hipStreamDestroy(s);
    s = nullptr;
    fork = join = nullptr;
    if (have_cur) (void)hipSetDevice(cur);
  }
  ~HeadStream() { release(); }  // thread exit: the stream and both events go with the thread
};
thread_local HeadStream t_head;
int head_stream(HeadStream** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (t_head.dev != dev || !t_head.s || !t_head.fork || !t_head.join) {
    // another device than last time, or a creation that failed half-way: drop what exists, then create all three -- a failure
    // leaves a partially filled record behind that the next call (or the destructor) releases
    t_head.release();
    t_head.dev = dev;
    HIP_TRY(hipStreamCreateWithFlags(&t_head.s, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&t_head.fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&t_head.join, hipEventDisableTiming));
  }
  *out = &t_head;
  return DSGE_SUCCESS;
}
// hand-off records of the fast kernel for kalman_tail_kernel: one buffer per (device, stream), grown on demand
StreamArenaPool g_tail_pool;
int tail_reserve(size_t bytes, hipStream_t st, void** out) { return g_tail_pool.reserve(bytes, st, out); }
}  // namespace

// dispatch key from the transition matrices themselves (any solver): see persistence_key_kernel
int launch_persistence_key(const double* T, const int32_t* status, int batch, int n, int32_t* key, hipStream_t st) {
  if (n > 64) return fail(DSGE_ERR_INVALID, "persistence key: n > 64");
  hipLaunchKernelGGL(dsge::persistence_key_kernel<64>, dim3(batch), dim3(64), 0, st, T, status, batch, n, key);
  HIP_TRY(hipGetLastError());
  return DSGE_SUCCESS;
}

// p0_valid = 0: P0 is an uninitialised scratch buffer.  The fast kernels then compute the stationary
// covariance themselves (on the reduced model); only the draws that end up in the general kernel get
// their full-size P0 from the assemble kernel's Lyapunov pass (from RQR, which must be valid).
bool kalman_folds_rqr(int m, int p, int k, int n_state_hint, int z_selector_hint) {
  // mirrors the dispatch below: the NT fast kernel takes every draw first (no tiny / tail / MFMA variant in front of it)
  if (!(p <= 8 && z_selector_hint && opt().kalman_nt_products) || opt().kalman_mfma == 1 || k < 1 || k > dsge::RQR_KMAX) return false;
  if (opt().kalman_tiny && p <= 3 && n_state_hint > 0 && n_state_hint + p <= 6) return false;
  if (opt().kalman_block && !opt().kalman_nt_products) return false;  // (the selector kernel's tail instance does not form R Q R')
  const int kp = (k + 1) & ~1;
  const int lo = (n_state_hint > 0 && n_state_hint < m) ? tile_bs(n_state_hint) : tile_bs(m);
  if (lo < 1 || lo > 8) return false;
  const int np = 8 * lo;  // staging area = the W' buffer of the smallest tile tried: NP x (NP + 2) doubles
  return (size_t)m * kp <= (size_t)np * (np + 2);
}

int launch_kalman(const double* T, double* RQR, double* P0, int p0_valid, const double* Z, int z_batched,
                  const double* d, int d_batched, const double* Hdiag, int h_batched, const double* y, int batch,
                  int m, int p, int T_len, double jitter, double missing_fill, int n_state_hint, int z_selector_hint,
                  double* logp, int32_t* status, hipStream_t st, const int32_t* order_key, const double* Rsel,
                  const double* qdiag, int q_batched, int k_shocks, const unsigned long long* colmask, int rerun_all) {
  const int bs = tile_bs(m);
  const dsge::FilterConv cv = filter_conv(jitter);  // the call's jitter + the conventions of dsge_options
  const bool fold = Rsel && qdiag && kalman_folds_rqr(m, p, k_shocks, n_state_hint, z_selector_hint);
  if (Rsel && !fold) return fail(DSGE_ERR_INVALID, "launch_kalman: R given but the filter kernel cannot form R Q R' itself");
  int rc = DSGE_ERR_INVALID;
  // Fast path: selector Z, p <= 8, compact state block of at most s_cap columns.  Draws that
  // violate a hint come back flagged and are re-run by the general kernel below.
  // Fast path (p <= 8): compact state block of at most s_cap columns; selector Z (gathers) or dense
  // Z (one extra product per step).  Draws that violate a hint come back flagged and are re-run by
  // the general kernel below.
  const bool fast = p <= 8;
  bool launched_fast = rerun_all != 0;  // (rerun_all: every launch below is a second pass on the flagged draws)
  // Small models (reduced filter of at most 6 variables, p <= 3, selector Z): one thread per draw, all in
  // registers.  Draws that do not fit are flagged and fall through to the wave-per-draw cascade below.
  if (opt().kalman_tiny && z_selector_hint && p <= 3 && n_state_hint > 0 && n_state_hint + p <= 6 && !rerun_all) {
    const int blocks = (batch + 63) / 64;
    if (n_state_hint + p <= 4) {
      hipLaunchKernelGGL((dsge::kalman_tiny_kernel<4, 3>), dim3(blocks), dim3(64), 0, st, T, RQR, Z, z_batched, d, d_batched,
                         Hdiag, h_batched, y, batch, m, p, T_len, cv, missing_fill, opt().kalman_steady_tol, logp, status,
                         g_kalman_steady_at);
    } else {
      hipLaunchKernelGGL((dsge::kalman_tiny_kernel<6, 3>), dim3(blocks), dim3(64), 0, st, T, RQR, Z, z_batched, d, d_batched,
                         Hdiag, h_batched, y, batch, m, p, T_len, cv, missing_fill, opt().kalman_steady_tol, logp, status,
                         g_kalman_steady_at);
    }
    HIP_TRY(hipGetLastError());
    launched_fast = true;
  }
  // Tail hand-off (selector Z, no hint violation needed: the kernels decide per draw): records, flags and the index from
  // which the missing-data mask of the shared panel y no longer changes.
  double* tail_rec = nullptr;
  int32_t* tail_flag = nullptr;
  int32_t* tail_from = nullptr;
  int32_t* order = nullptr;
  const bool want_tail = fast && z_selector_hint && opt().kalman_block && T_len >= 8;
  const bool want_order = fast && order_key && opt().kalman_order && batch >= 512;
  if (want_tail || want_order) {
    void* base = nullptr;
    const size_t rec_bytes = want_tail ? (size_t)batch * dsge::KT_REC * sizeof(double) : 0;
    const size_t int_bytes = ((2 * (size_t)batch + 1) * sizeof(int32_t) + 255) & ~(size_t)255;
    if ((rc = tail_reserve(rec_bytes + int_bytes, st, &base))) return rc;
    int32_t* ints = (int32_t*)base;  // [batch] flags, [1] scan result, [batch] dispatch order
    if (want_tail) {
      tail_flag = ints;
      tail_from = ints + batch;
      tail_rec = (double*)((char*)base + int_bytes);
      HIP_TRY(hipMemsetAsync(tail_flag, 0, ((size_t)batch + 1) * sizeof(int32_t), st));
      hipLaunchKernelGGL(dsge::kalman_mask_scan_kernel, dim3(1), dim3(64), 0, st, y, p, T_len, missing_fill, tail_from);
    }
    if (want_order) {
      order = ints + batch + 1;
      hipLaunchKernelGGL(dsge::kalman_order_kernel<1024>, dim3(1), dim3(1024), 0, st, order_key, batch, order);
    }
    HIP_TRY(hipGetLastError());
  }
  auto launch_tail = [&](hipStream_t ts) -> int {
    // four steps per trip; one launch per tile width that can have written records (a draw's record names its dimension m: the
    // instance MC takes the draws with m_lo < m <= MC, so that every draw runs with the shortest rows that hold it).
    // kalman_block = 2: the two-step kernel of round 2 (kept for comparison)
    if (opt().kalman_block == 2) {
      hipLaunchKernelGGL(dsge::kalman_tail_kernel, dim3(batch), dim3(64), 0, ts, (const double*)tail_rec,
                         (int32_t*)tail_flag, y, batch, p, T_len, missing_fill, logp, status, g_kalman_steady_at, cv);
    } else {
      const int bs_lo = tile_bs((z_selector_hint && n_state_hint > 0 && n_state_hint < m) ? n_state_hint : m);
      const int bs_hi = tile_bs((z_selector_hint && n_state_hint > 0 && n_state_hint + p < m) ? n_state_hint + p : m);
#define LAUNCH_TAIL4(MCV)                                                                                                     \
  hipLaunchKernelGGL((dsge::kalman_tail4_kernel<MCV>), dim3(batch), dim3(64), 0, ts, (const double*)tail_rec,                 \
                     (int32_t*)tail_flag, y, batch, p, T_len, missing_fill, logp, status, g_kalman_steady_at, cv, m_lo)
      for (int b = bs_lo; b <= bs_hi && b <= 4; ++b) {
        const int m_lo = (b == bs_lo) ? 0 : 8 * (b - 1);  // (the narrowest instance also takes the draws that are smaller than its tile)
        if (b <= 1)
          LAUNCH_TAIL4(8);
        else if (b == 2)
          LAUNCH_TAIL4(16);
        else if (b == 3)
          LAUNCH_TAIL4(24);
        else
          LAUNCH_TAIL4(32);
      }
#undef LAUNCH_TAIL4
    }
    HIP_TRY(hipGetLastError());
    return DSGE_SUCCESS;
  };
  // Round 6: the covariance in the tile layout of the FP64 matrix core's 4 x 4 x 4 instruction (launch_kalman_mf.hip, dsge_kalman_mf.hpp)
  bool mf_covers = false;  // the tile-layout instances launched cover everything the VALU cascade below could take
  if (fast && z_selector_hint && opt().kalman_mfma == 2 && opt().kalman_nt_products && !want_tail && opt().kalman_head_draws == 0 &&
      n_state_hint >= 9 && n_state_hint <= 20 && (!launched_fast || rerun_all)) {
    int launched = 0;
    if ((rc = launch_kalman_mf(T, RQR, p0_valid ? P0 : nullptr, Z, z_batched, d, d_batched, Hdiag, h_batched, y, batch, m, p, T_len, cv,
                               missing_fill, n_state_hint, logp, status, st, order, fold ? Rsel : nullptr, qdiag, q_batched, k_shocks,
                               colmask, rerun_all, &launched, &mf_covers)))
      return rc;
    launched_fast = launched_fast || launched > 0;
  }
  if (fast && !mf_covers) {
    // The fast kernel filters only the variables that matter (states + observed non-states), so its
    // tile size follows that reduced dimension u, not m.  u is only known per draw on the device
    // (n_state_hint <= u <= n_state_hint + p for a selector), so the instances are tried smallest
    // first: each one flags the draws that do not fit it, the next one picks up exactly those.
    // A dense Z may load on every variable: no reduction is assumed for it.
    int tiles[2];
    int n_tiles = 0;
    if (z_selector_hint && n_state_hint > 0 && n_state_hint < m) {
      tiles[n_tiles++] = tile_bs(n_state_hint);
      const int hi = tile_bs(n_state_hint + p < m ? n_state_hint + p : m);
      if (hi != tiles[0]) tiles[n_tiles++] = hi;
    } else {
      tiles[n_tiles++] = tile_bs(m);
    }
    for (int it = 0; it < n_tiles; ++it) {
      const int bs_fast = tiles[it];
      const int rerun = launched_fast ? 1 : 0;
      rc = DSGE_ERR_INVALID;
      DISPATCH_BS(bs_fast, 8, {
        constexpr int NP = 8 * BS;
        int s_cap = (n_state_hint > 0 && n_state_hint < NP) ? ((n_state_hint + BS - 1) / BS) * BS : NP;
        if (s_cap > NP) s_cap = NP;
        if (z_selector_hint) {
          const size_t lds = dsge::Kf2Smem<BS>::bytes(s_cap, false);
          bool done = false;
          if constexpr (BS == 2 || BS == 3) {
            if (opt().kalman_mfma == 1) {  // prediction products on the FP64 matrix core's 16 x 16 x 4 instruction (round 2, slower)
              rc = set_lds(dsge::kalman_sel_kernel<BS, true, true>, lds);
              if (rc == DSGE_SUCCESS) {
                hipLaunchKernelGGL((dsge::kalman_sel_kernel<BS, true, true>), dim3(rerun ? rerun_grid(batch) : batch), dim3(64), lds, st, T, RQR,
                                   p0_valid ? P0 : nullptr, Z, z_batched, d, d_batched, Hdiag, h_batched, y, batch, m, p,
                                   T_len, s_cap, cv, missing_fill, opt().kalman_steady_tol, logp, status, g_kalman_dbg, rerun,
                                   g_kalman_steady_at, nullptr, nullptr, nullptr, order);
                HIP_TRY(hipGetLastError());
                launched_fast = true;
              }
              done = true;
            }
          }
          if (!done && tail_rec && BS <= 4 && !(opt().kalman_nt_products && !g_kalman_dbg)) {
            rc = set_lds(dsge::kalman_sel_kernel<BS, true, false, true>, lds);
            if (rc == DSGE_SUCCESS) {
              hipLaunchKernelGGL((dsge::kalman_sel_kernel<BS, true, false, true>), dim3(rerun ? rerun_grid(batch) : batch), dim3(64), lds, st, T, RQR,
                                 p0_valid ? P0 : nullptr, Z, z_batched, d, d_batched, Hdiag, h_batched, y, batch, m, p, T_len,
                                 s_cap, cv, missing_fill, opt().kalman_steady_tol, logp, status, g_kalman_dbg, rerun,
                                 g_kalman_steady_at, tail_rec, tail_flag, tail_from, order);
              HIP_TRY(hipGetLastError());
              launched_fast = true;
            }
            done = true;
          }
          if (!done && opt().kalman_nt_products) {
            // round-2 fast path: NT prediction products on 16-byte aligned rows (dsge_kalman_nt.hpp)
            // instance by the state-block capacity: SK = 20 (up to 18-20 state variables: rows of the LDS matrices 22 doubles
            // long, the 32-wide tile at exactly 20 KB = eight draws per CU and two waves per SIMD) or the generic SK = NP
            auto launch_nt = [&](auto sk_tag, auto tail_tag) {
              constexpr int SKV = decltype(sk_tag)::value;
              constexpr bool TAILV = decltype(tail_tag)::value;  // hand the steady, constant-mask tails to kalman_tail4_kernel
              const size_t lds_q = dsge::KntSmem<BS, SKV>::bytes(s_cap);
              if constexpr (BS <= 4) {
                if (g_kalman_dbg && opt().kalman_head_draws != 0) {  // tools/kalman_phases.py 2: the two-wavefront kernel, stamped
                  const size_t lds2 = dsge::Knt2Smem<BS, SKV>::bytes(s_cap);
                  rc = set_lds(dsge::kalman_nt2_kernel<BS, SKV, true>, lds2);
                  if (rc == DSGE_SUCCESS)
                    hipLaunchKernelGGL((dsge::kalman_nt2_kernel<BS, SKV, true>), dim3(batch), dim3(128), lds2, st, T, RQR,
                                       p0_valid ? P0 : nullptr, Z, z_batched, d, d_batched, Hdiag, h_batched, y, batch, m, p, T_len,
                                       s_cap, cv, missing_fill, opt().kalman_steady_tol, logp, status, rerun, g_kalman_steady_at,
                                       order, fold ? Rsel : nullptr, qdiag, q_batched, k_shocks, colmask, g_kalman_dbg);
                  return;
                }
              }
              if (g_kalman_dbg) {  // tools/kalman_phases.py: the instance with the phase stamps
                rc = set_lds(dsge::kalman_nt_kernel<BS, true, SKV>, lds_q);
                if (rc == DSGE_SUCCESS)
                  hipLaunchKernelGGL((dsge::kalman_nt_kernel<BS, true, SKV>), dim3(batch), dim3(64), lds_q,
                                     st, T, RQR, p0_valid ? P0 : nullptr, Z, z_batched, d, d_batched, Hdiag, h_batched, y, batch, m,
                                     p, T_len, s_cap, cv, missing_fill, opt().kalman_steady_tol, logp, status, g_kalman_dbg,
                                     rerun, g_kalman_steady_at, order, fold ? Rsel : nullptr, qdiag, q_batched, k_shocks, colmask);
                return;
              }
              // Head of the dispatch order on the two-wavefront kernel (dsge_kalman_nt2.hpp), on a second stream next to the bulk:
              // the launch ends with its slowest draws, and those run a full step in two thirds of the time there.
              int head = 0;
              if constexpr (BS <= 4) {
                head = opt().kalman_head_draws;
                if (rerun) head = 0;
                if (head < 0 || head > batch) head = batch;
                if (head > 0 && head < batch && !order) head = 0;  // (without an order the first indices are no slower than the rest)
                if (head > 0 && dsge::Knt2Smem<BS, SKV>::bytes(s_cap) > LDS_LIMIT) head = 0;
              }
              // The HEAD runs on the caller's stream, where it starts the moment the solver ends; the BULK goes to the library's
              // second stream (fork / join by events).  The other way round the bulk, in stream order right behind the solver, fills
              // every CU's LDS before the cross-stream dependency of the head resolves, and the slow draws start LAST (measured).
              HeadStream* hs = nullptr;
              hipStream_t bulk_st = st;
              if (head > 0 && head < batch) {
                if ((rc = head_stream(&hs))) return;
                if (hipEventRecord(hs->fork, st) != hipSuccess || hipStreamWaitEvent(hs->s, hs->fork, 0) != hipSuccess) {
                  rc = fail(DSGE_ERR_HIP, "kalman head launch: fork failed");
                  return;
                }
                bulk_st = hs->s;
              }
              if constexpr (BS <= 4) {
                if (head > 0) {
                  const size_t lds2 = dsge::Knt2Smem<BS, SKV>::bytes(s_cap);
                  if ((rc = set_lds(dsge::kalman_nt2_kernel<BS, SKV>, lds2))) return;
                  hipLaunchKernelGGL((dsge::kalman_nt2_kernel<BS, SKV>), dim3(head), dim3(128), lds2, st, T, RQR,
                                     p0_valid ? P0 : nullptr, Z, z_batched, d, d_batched, Hdiag, h_batched, y, head, m, p, T_len,
                                     s_cap, cv, missing_fill, opt().kalman_steady_tol, logp, status, rerun, g_kalman_steady_at,
                                     order, fold ? Rsel : nullptr, qdiag, q_batched, k_shocks, colmask, nullptr);
                }
              }
              rc = DSGE_SUCCESS;
              if (head < batch) {
                rc = set_lds(dsge::kalman_nt_kernel<BS, false, SKV, TAILV>, lds_q);
                if (rc == DSGE_SUCCESS)
                  hipLaunchKernelGGL((dsge::kalman_nt_kernel<BS, false, SKV, TAILV>), dim3(batch - head), dim3(64), lds_q,
                                     bulk_st, T, RQR, p0_valid ? P0 : nullptr, Z, z_batched, d, d_batched, Hdiag, h_batched, y,
                                     batch - head, m, p, T_len, s_cap, cv, missing_fill, opt().kalman_steady_tol, logp, status,
                                     g_kalman_timeline, rerun, g_kalman_steady_at, order ? order + head : nullptr, fold ? Rsel : nullptr,
                                     qdiag, q_batched, k_shocks, colmask, TAILV ? tail_rec : nullptr, tail_flag, tail_from);
              }
              if (hs && TAILV && rc == DSGE_SUCCESS) rc = launch_tail(hs->s);  // the bulk's tails next to the head, not behind it
              if (hs) {  // join: everything later on the caller's stream waits for the bulk
                if (hipEventRecord(hs->join, hs->s) != hipSuccess || hipStreamWaitEvent(st, hs->join, 0) != hipSuccess)
                  rc = fail(DSGE_ERR_HIP, "kalman head launch: join failed");
              }
            };
            // (the 32-wide tile only: on the 24-wide one the 20-column instance measured 1.4 % SLOWER on the headline step --
            //  1.588 against 1.566 ms per 4096 draws --, its rows are only four columns shorter.  Not an `if constexpr`: this
            //  is no template)
            constexpr int SK_NARROW = (BS == 4) ? 20 : 8 * BS;
            // (with R folded in, the kernel stages the m x k selection matrix in its W' buffer, which is narrower too)
            const bool stage_fits = !fold || (size_t)m * ((k_shocks + 1) & ~1) <= (size_t)(8 * BS) * (SK_NARROW + 2);
            const bool narrow_i = SK_NARROW < 8 * BS && s_cap <= SK_NARROW && opt().kalman_narrow && stage_fits;
            const bool tail_i = tail_rec != nullptr && !g_kalman_dbg && BS <= 4;  // (the record holds a tile of at most 32 variables)
            if (narrow_i && tail_i)
              launch_nt(std::integral_constant<int, SK_NARROW>{}, std::true_type{});
            else if (narrow_i)
              launch_nt(std::integral_constant<int, SK_NARROW>{}, std::false_type{});
            else if (tail_i)
              launch_nt(std::integral_constant<int, 8 * BS>{}, std::true_type{});
            else
              launch_nt(std::integral_constant<int, 8 * BS>{}, std::false_type{});
            if (rc == DSGE_SUCCESS) {
              HIP_TRY(hipGetLastError());
              launched_fast = true;
            }
            done = true;
          }
          if (!done) {
            rc = set_lds(dsge::kalman_sel_kernel<BS, true>, lds);
            if (rc == DSGE_SUCCESS) {
              hipLaunchKernelGGL((dsge::kalman_sel_kernel<BS, true>), dim3(rerun ? rerun_grid(batch) : batch), dim3(64), lds, st, T, RQR,
                                 p0_valid ? P0 : nullptr, Z, z_batched, d, d_batched, Hdiag, h_batched, y, batch, m, p, T_len,
                                 s_cap, cv, missing_fill, opt().kalman_steady_tol, logp, status, g_kalman_dbg, rerun,
                                 g_kalman_steady_at, nullptr, nullptr, nullptr, order);
              HIP_TRY(hipGetLastError());
              launched_fast = true;
            }
          }
        } else {
          const size_t lds = dsge::Kf2Smem<BS>::bytes(s_cap, true);
          if (lds > LDS_LIMIT) {
            rc = DSGE_SUCCESS;  // does not fit: the general kernel handles everything
          } else {
            rc = set_lds(dsge::kalman_sel_kernel<BS, false>, lds);
            if (rc == DSGE_SUCCESS) {
              hipLaunchKernelGGL((dsge::kalman_sel_kernel<BS, false>), dim3(rerun ? rerun_grid(batch) : batch), dim3(64), lds, st, T, RQR,
                                 p0_valid ? P0 : nullptr, Z, z_batched, d, d_batched, Hdiag, h_batched, y, batch, m, p, T_len, s_cap,
                                 cv, missing_fill, opt().kalman_steady_tol, logp, status, g_kalman_dbg, rerun,
                                 g_kalman_steady_at, nullptr, nullptr, nullptr, order);
              HIP_TRY(hipGetLastError());
              launched_fast = true;
            }
          }
        }
      });
      if (rc) return rc;
    }
  }
  if (tail_rec) {
    if ((rc = launch_tail(st))) return rc;
  }
  if (fold) {  // the general kernel's inputs for the draws the fast kernel handed on: their sym(R Q R') after all
    if ((rc = launch_rqr(Rsel, qdiag, q_batched, batch, m, k_shocks, status, RQR, st, 1))) return rc;
  }
  if (!p0_valid) {
    // full-size P0 for the general kernel: flagged draws only when a fast kernel ran, else every draw
    if ((rc = launch_assemble(nullptr, nullptr, nullptr, nullptr, T, nullptr, nullptr, 0, batch, m, 1, nullptr, nullptr,
                              RQR, P0, status, 0, launched_fast ? 3 : 4, st)))
      return rc;
  }
  rc = DSGE_ERR_INVALID;
  DISPATCH_BS(bs, 8, {
    const size_t lds = dsge::KfSmem<BS>::bytes(p);
    rc = set_lds(dsge::kalman_kernel<BS>, lds);
    if (rc == DSGE_SUCCESS) {
      hipLaunchKernelGGL(dsge::kalman_kernel<BS>, dim3(launched_fast ? rerun_grid(batch) : batch), dim3(64), lds, st, T, RQR, P0, Z,
                         z_batched, d,
                         d_batched, Hdiag, h_batched, y, batch, m, p, T_len, cv, missing_fill, logp, status,
                         launched_fast ? 1 : 0);
      HIP_TRY(hipGetLastError());
    }
  });
  return rc;
}

// per-step filter outputs (dsge_kalman_out.hpp)
int launch_kalman_outputs(const double* T, const double* RQR, const double* P0, const double* Z, int z_batched, const double* d,
                          int d_batched, const double* Hdiag, int h_batched, const double* y, int batch, int m, int p, int T_len,
                          double jitter, double missing_fill, double* ll, double* a_pred, double* a_filt, double* p_pred,
                          double* p_filt, int full_cov, int32_t* status, hipStream_t st) {
  dsge::KoArgs a{};
  a.T = T; a.RQR = RQR; a.P0 = P0; a.Z = Z; a.d = d; a.Hdiag = Hdiag; a.y = y; a.ll = ll; a.a_pred = a_pred; a.a_filt = a_filt;
  a.p_pred = p_pred; a.p_filt = p_filt; a.status = status; a.batch = batch; a.m = m; a.p = p; a.T_len = T_len;
  a.z_batched = z_batched; a.d_batched = d_batched; a.h_batched = h_batched; a.full_cov = full_cov; a.cv = filter_conv(jitter);
  a.missing_fill = missing_fill;
  const size_t lds = dsge::ko_lds_doubles(m, p) * sizeof(double);
  int rc;
  if ((rc = set_lds(dsge::kalman_outputs_kernel, lds))) return rc;
  hipLaunchKernelGGL(dsge::kalman_outputs_kernel, dim3(batch), dim3(dsge::KO_THREADS), lds, st, a);
  HIP_TRY(hipGetLastError());
  return DSGE_SUCCESS;
}

}  // namespace dsge_host

/*
 * dsge_hip.h -- C ABI of libdsge_hip.so, the MI355X (gfx950) batched engine for gEconpy's
 * estimation hot path: first-order perturbation solve + Kalman-filter log-likelihood,
 * evaluated for a batch of parameter draws per call.
 *
 * Conventions (reference conventions cited per entry point):
 *   - every matrix is float64, row-major, contiguous: [batch][rows][cols];
 *     the reference coerces with np.ascontiguousarray(..., float64)
 *     (gEconpy/solvers/gensys.py:505-509,625-628) and asserts C order
 *     (gEconpy/model/model.py:2000);
 *   - A,B,C,D arrive in solver order and T,R are returned in that order; the library
 *     applies no variable permutation (the caller un-permutes exactly as
 *     gEconpy/model/statespace.py:217-220 does);
 *   - the caller owns every input and output buffer; inputs are never written
 *     (gEconpy/solvers/cycle_reduction.py:134-136); scratch is library-owned;
 *   - numerical failure of a draw is a VALUE, never an error return: per-draw status
 *     words / eu codes, zero-filled T, logp = -inf (gensys.py:255-265,
 *     cycle_reduction.py:181, statespace.py:1206-1215).  The int return value is non-zero
 *     only when the CALL is malformed or HIP fails; dsge_last_error() describes it;
 *   - entry points without suffix take DEVICE pointers and a hipStream_t (passed as
 *     void*, NULL = default stream) and only enqueue work; the *_host twins take HOST
 *     pointers, stage through library-owned device buffers and return after completion.
 *
 * The library refuses to run (DSGE_ERR_HIP) when no gfx950 device is present; there is no
 * CPU fallback.
 */
#ifndef DSGE_HIP_H
#define DSGE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSGE_ABI_VERSION 9

/* ABI 8: the process-wide dsge_set_* switches (deprecated at ABI 7) are GONE -- they edited defaults shared by every host
 * thread and stream of the process, which a library called from several PyMC chains must not have.  Every switch is a field
 * of the per-call dsge_options struct (the *_opt entry points, or dsge_options_push / dsge_options_pop around any entry point
 * on the calling thread); dsge_options_init() fills the compiled-in defaults. */
/* ABI 9: no symbol and no struct layout changed; two BEHAVIOURS did, which is what a version is for: (i) dsge_options.kalman_mfma takes
 * the value 2 -- prediction products on the 4 x 4 x 4 FP64 matrix instruction -- and 2 is the default (was 0); (ii) solver = gensys by
 * spectral division (gensys_doubling = 1) certifies a draw only if two scale guards hold that make the reference's absolute-tolerance
 * tests (coincident zeros, rank of Q2 Pi; gEconpy/solvers/gensys.py:243, 276-283) provably pass -- a draw with an equation scaled
 * by <= ~tol now gets the ordered QZ's verdict instead of eu = [1, 1, 0]. */

/* limits of this build */
#define DSGE_MAX_N 64      /* model variables n == Kalman states m */
#define DSGE_MAX_N_CR 64
#define DSGE_MAX_N_GENSYS 64   /* pencil dimension n + #lead columns (further limited by 160 KB LDS) */
#define DSGE_MAX_N_BIG 96      /* cycle reduction (both variants), the selection matrix and the fused solve + Kalman log-likelihood
                                  with a cycle-reduction solver also take 65 .. 96 variables (one workgroup per draw,
                                  csrc/dsge_big.hpp).  The filter then runs on the model restricted to its state and observed
                                  variables, of which there may be at most 64 (else DSGE_ERR_TOO_LARGE); every other entry point
                                  keeps the limits above */
#define DSGE_MAX_P 16      /* observed series */

/* call-level return codes */
#define DSGE_SUCCESS 0
#define DSGE_ERR_INVALID 1     /* bad size / null pointer / unsupported option */
#define DSGE_ERR_HIP 2         /* HIP runtime error or no usable device */
#define DSGE_ERR_TOO_LARGE 3   /* well-formed call whose problem exceeds this entry point's on-chip capacity (the 160 KB of
                                  LDS); nothing was enqueued.  Callers with a second route (the Python wrapper of
                                  solve_policy_function_with_gensys: window-path kernels) branch on this code, never on the
                                  message text */

/* per-draw status bits (int32) */
#define DSGE_ST_OK 0
#define DSGE_ST_NOT_CONVERGED 1   /* cycle reduction hit max_iter / gensys eu != [1,1]      */
#define DSGE_ST_NAN 2             /* NaN met in the solver (cycle_reduction.py:176-177)      */
#define DSGE_ST_LYAP_FAIL 4       /* doubling iteration for P0 did not converge (rho(T)>=1)  */
#define DSGE_ST_FILTER_NONFINITE 8 /* non-finite log-likelihood (F not positive definite...) */
#define DSGE_ST_GENSYS_QZ_FAIL 16  /* QZ iteration did not converge                           */
#define DSGE_ST_GENSYS_TOO_BIG 32  /* n + #lead exceeds the on-chip capacity of the launch     */
#define DSGE_ST_GRAD_UNSUPPORTED 64 /* gradient path: dense design matrix / reduced model exceeds the tile */
#define DSGE_ST_SECOND_ORDER_UNSUPPORTED 128 /* second-order path: input violates the declared model structure */

/* covariance layouts for the Q argument */
#define DSGE_Q_DIAG_SHARED 0    /* Q = diag(q), q: [k]            */
#define DSGE_Q_DIAG_BATCHED 1   /* q: [batch][k]  (statespace.py:240-258, sigma_i^2)          */
#define DSGE_Q_FULL_SHARED 2    /* Q: [k][k]                                                  */
#define DSGE_Q_FULL_BATCHED 3   /* Q: [batch][k][k] (full_covariance, statespace.py:247-251)  */

/* solver selector of the fused entry point (statespace.py:197-207) */
#define DSGE_SOLVER_CYCLE_REDUCTION 0
#define DSGE_SOLVER_GENSYS 1
#define DSGE_SOLVER_BACKWARD_DIRECT 2
#define DSGE_SOLVER_SCAN_CYCLE_REDUCTION 3
/* Flag, OR-ed into the solver code of dsge_solve_kalman_logp_batched (+ _opt / _host twins) with a cycle-reduction solver: what
 * the reference's graph computes when add_solver_success_check is left at its default False (statespace.py:1148, 1210-1215) -- a
 * draw whose cycle reduction did not converge carries T = 0 on (cycle_reduction.py:181), R = -B^-1 D, P0 = R Q R', and gets
 * the FINITE log-likelihood of that system instead of -inf.  status[i] still reports the failure.  Without the flag (default)
 * a failed draw gives logp = -inf, i.e. add_solver_success_check = True.  The flag turns off the shortcuts that take R from
 * the solver's final elimination (R is recomputed from T for every draw) and is ignored by the other solvers. */
#define DSGE_SOLVER_FLAG_ZERO_T_ON_FAILURE 0x100

int dsge_abi_version(void);
const char* dsge_last_error(void);
/* number of visible gfx950 devices (0 if none); never throws */
int dsge_device_count(void);
int dsge_set_device(int device);
/* blocks until all work enqueued on `stream` has finished */
int dsge_stream_synchronize(void* stream);

/*
 * Cycle reduction, batched.  Replaces _cycle_reduction_core / CycleReductionWrapper.perform
 * (gEconpy/solvers/cycle_reduction.py:127-183, :206-210): solves A + B T + C T^2 = 0.
 *   A,B,C : [batch][n][n]      T_out : [batch][n][n] (zeros where not converged, :181)
 *   status: [batch] DSGE_ST_*  n_iter: [batch] iterations used (may be NULL)
 * Stopping rule of the njit variant (:171-177): ||A0||_1 < tol and ||A2||_1 < tol.
 * n <= DSGE_MAX_N_BIG (65 .. 96 variables: one workgroup per draw, csrc/dsge_big.hpp); so has the scan variant below.
 */
int dsge_cycle_reduction_batched(const double* A, const double* B, const double* C, int batch, int n,
                                 int max_iter, double tol, double* T_out, int32_t* status,
                                 int32_t* n_iter, void* stream);
int dsge_cycle_reduction_batched_host(const double* A, const double* B, const double* C, int batch, int n,
                                      int max_iter, double tol, double* T_out, int32_t* status,
                                      int32_t* n_iter);
/*
 * Cycle reduction with the semantics of the pure-pytensor scan variant.  Replaces _scan_cycle_reduction /
 * scan_cycle_reduction (gEconpy/solvers/cycle_reduction.py:246-325): at most max_iter steps, a step is
 * skipped once ||A0||_1 < tol (the A0 norm ONLY, :269-277), 1e-16 is added to the diagonal of A1 / A1_hat
 * in every solve (stabilize, gEconpy/solvers/shared.py:6-9), and T = -(A1_hat + 1e-16 I)^-1 A is formed from
 * whatever iterate the trip count reached (:292) -- there is no convergence flag in the reference.
 *   n_steps : [batch] steps actually taken (the third output of scan_cycle_reduction), may be NULL
 *   status  : 0 unless a NaN appeared (then DSGE_ST_NOT_CONVERGED | DSGE_ST_NAN and T = 0: the reference
 *             would return a NaN matrix; this library never hands NaN policies downstream)
 */
int dsge_scan_cycle_reduction_batched(const double* A, const double* B, const double* C, int batch, int n,
                                      int max_iter, double tol, double* T_out, int32_t* status, int32_t* n_steps,
                                      void* stream);
int dsge_scan_cycle_reduction_batched_host(const double* A, const double* B, const double* C, int batch, int n,
                                           int max_iter, double tol, double* T_out, int32_t* status,
                                           int32_t* n_steps);
/*
 * Per-call options ("re-entrant per stream").  Every kernel-variant switch of the library and every
 * third-party CONVENTION of the filter step is a field of this struct.  Calls that carry a dsge_options -- the *_opt twins of
 * the fused entry points, or any entry point called between dsge_options_push() and dsge_options_pop() on the same host
 * thread -- use exactly those settings: they are installed for the duration of the call on the calling thread only, so two
 * host threads (two PyMC / nutpie chains, two pytensor Ops, two streams) never see each other's settings.  Calls without
 * options use the compiled-in defaults (what dsge_options_init() fills in); there is no process-wide mutable state.
 *   struct_size : sizeof(dsge_options) of the caller's build (set by dsge_options_init; checked)
 *   n_static_hint : number of static variables of the model (columns of A and C both exactly zero), a property of the model
 *                   like n_state_hint, verified per draw on the device; with a value >= 0 the fused call is a pure enqueue
 *                   (stream-capturable); -1 (default) measures it on the device on the first call of a model size: one small
 *                   launch, a 4-byte read-back and a stream synchronisation (dsge_forget_measured_shapes() drops the record)
 *   the kernel-variant switches are documented one by one below the struct.
 *
 * Filter conventions (ABI 8).  The "standard" filter step of the reference lives in third-party pymc_extras
 * (PyMCStateSpace.build_statespace_graph, reached from gEconpy/model/statespace.py:1143-1157); its constants cannot be pinned
 * in the build image, so NONE of them is compiled into a kernel: every filter kernel (fast, selector, tiny, tail, general,
 * per-step outputs, gradient, second order) takes them as run-time values.  oracle.FilterConventions has the same switches,
 * tests/test_gpu_conventions.py checks every combination against it, and tests/test_oracle_kalman.py::test_pymc_extras_pin
 * names the combination a real install matches -- which is then set here, without touching a kernel:
 *   ll_constant : DSGE_LL_CONST_P (default) ll_t = -1/2 (p ln 2pi + ln det F + v' F^-1 v) with p the FULL observation
 *                 dimension, also when entries are missing; DSGE_LL_CONST_OBSERVED: (#observed entries of y_t) ln 2pi;
 *                 DSGE_LL_CONST_ONE: a single ln 2pi per step (older upstream StandardFilter.update)
 *   jitter_F    : added to the diagonal of F = Zm P Zm' + Hm; < 0 (default -1): the `jitter` argument of the call
 *   jitter_P    : added to the diagonal of the filtered covariance P+;  < 0 (default -1): the `jitter` argument of the call
 *                 (the reference passes ONE cov_jitter for both, statespace.py:1144; 0 switches an addition off)
 *   mask_d      : 1: the observation intercept d is zeroed on missing entries; 0 (default): d is NOT masked, a missing entry
 *                 contributes ln(jitter_F) + d_i^2 / jitter_F to ll_t
 *   joseph      : 1 (default): Joseph form P+ = (I - K Zm) P (I - K Zm)' + K Hm K' (+ jitter_P I); 0: P+ = P - K F K'.  The
 *                 two differ by jitter_F K K' (expand with K F = P Zm'): the kernels compute P+ = P - K (P Zm' + j K)' with
 *                 j = jitter_F or 0
 */
#define DSGE_LL_CONST_P 0
#define DSGE_LL_CONST_OBSERVED 1
#define DSGE_LL_CONST_ONE 2
typedef struct dsge_options {
  uint32_t struct_size;
  int32_t cr_compact;
  int32_t cr_fused_selection;
  int32_t cr_deflation;
  int32_t cr_two_waves;
  int32_t n_static_hint;
  int32_t kalman_order;
  int32_t kalman_tiny;
  int32_t kalman_block;
  int32_t kalman_mfma;
  int32_t pipeline_chunks;
  int32_t gensys_split;
  double kalman_steady_tol;
  int32_t kalman_nt_products;
  int32_t cr_fused_deflation;
  int32_t cr_four_waves;
  int32_t gensys_real_stage;
  int32_t gensys_pairs;       /* 1 (default): window path, real double-shift sweeps with TWO draws per wavefront when the window
                                 and #lead are <= 32 (dsge_gensys_pair.hpp); 0: one draw per wavefront (round 3); 2: also the
                                 Hessenberg-triangular launch on two draws per wavefront (measured slower: experiment) */
  int32_t gensys_shape_cache; /* 1 (default): the capacity record of the window path (max #lead, window, deflated roots) is
                                 measured on the first call of a model size and reused -- later calls are pure enqueues; a
                                 draw that exceeds it is flagged DSGE_ST_GENSYS_TOO_BIG and the record grows for the next
                                 call; 0: measured in every call (one small launch + a stream synchronisation) */
  int32_t kalman_narrow;      /* 1 (default): fast filter, 32-wide tile: the instance whose LDS rows hold 20 state
                                 columns instead of the tile width when the model has at most 20 state variables (the 32-wide
                                 tile then takes exactly 20 KB: eight draws per CU); same arithmetic, bit-identical results;
                                 0: generic instance */
  int32_t gensys_direct_blocks; /* 1 (default): window path, behind the real double-shift stage: the isolated 2 x 2 blocks of the
                                 quasi-triangular window are triangularised in closed form (a root of the 2 x 2 pencil, its null
                                 vector, two rotations), checked, and zhgeqz's iteration runs only when something is left for
                                 it; 0: the complex single-shift iteration splits them (round 3) */
  /* ABI 8: conventions of the filter step (see above) */
  int32_t ll_constant;
  int32_t mask_d;
  int32_t joseph;
  int32_t kalman_head_draws;  /* fast filter (selector Z, p <= 8, tiles up to 32 variables): this many draws at the HEAD of the
                                 dispatch order (dsge_options.kalman_order: the draws most likely to run many full covariance
                                 updates) are filtered by kalman_nt2_kernel -- two wavefronts per draw, the measurement update on
                                 one, the two prediction products of the PREDICTED covariance on the other, a rank-p correction
                                 joins them: a full step takes two thirds of the one-wavefront kernel's time (dsge_kalman_nt2.hpp) --
                                 on a library-owned second stream next to the bulk (fork / join by events on the caller's stream).
                                 Same recursion, results agree to rounding (tests compare the two kernels).  0: off; -1: every draw */
  double jitter_F;
  double jitter_P;
  int32_t gensys_doubling;    /* 1 (default): gensys by spectral division (csrc/dsge_gensys_doubling.hpp) -- the doubling iteration
                                 (cycle reduction) computes the solvent T, a per-draw CERTIFICATE (rho(T[S,S]) < 1 and
                                 rho(((B + C T)^-1 C)[L,L]) < 1 by norms of repeated squares: exactly gensys's eu = [1, 1, 0]) is
                                 checked on the device, and every draw WITHOUT it -- not converged, a root within 2e-4 of the unit
                                 circle, more lead / state columns than hinted, a column of C below tol, singular B + C T, |T| >
                                 1e6 -- is solved by the ordered QZ (the window launches on the compacted list of those draws),
                                 which alone issues the non-regular verdicts.  Same eu / status as 0 on every system of the test
                                 and fuzz suites; T agrees with the QZ's to what two float64 algorithms differ by (1e-12 on
                                 well-conditioned draws, cond(B + C T) eps in general).  0: the ordered QZ of the pencil for every
                                 draw (the reference's algorithm, gEconpy/solvers/gensys.py:190-395, operation by operation).
                                 2: as 1 with the single-launch QZ kernel as the fall-back (debug).
                                 Round 6 (ABI 9): under 1 the fused evaluation (dsge_solve_kalman_logp_batched, plain form: no
                                 residual output, diagonal Q folded into the filter, stream not being captured) runs the VERDICT --
                                 certificate, compaction, the QZ of the draws without one -- on a library stream NEXT to the filter
                                 of all draws, joins, and filters the draws the verdict re-solved a second time; a certified draw is
                                 filtered once with the same inputs as in the serial order (identical logp).  3: as 1 with the
                                 verdict on the caller's stream in front of the filter (the round-5 order; for comparison).  The
                                 scale guards of the certificate (csrc/dsge_gensys_doubling.hpp: existence computed exactly from
                                 (B + C T)^-1, a lower bound on the stable block's QZ diagonal) hold under 1, 2 and 3. */
  int32_t kalman_grad_split;  /* 2 (default), 1: the logp + gradient entry points run the FORWARD filter sweep as a logp kernel
                                 itself with record output (the tile-layout kalman_mf_kernel on the 24-wide tile since round 6,
                                 kalman_nt_kernel otherwise: two wavefronts per SIMD) and the reverse sweep as a kernel of its own;
                                 draws the forward kernel cannot take fall back to the one-kernel path in the same call.  2: in
                                 addition the reverse MEAN side of every draw's last steady segment runs as a lean kernel at two
                                 wavefronts per SIMD (kalman_grad_tail_kernel) and the reverse sweep resumes at that segment's source
                                 step: 40 % of the reverse launch's work.  Round 5 measured no gain (that launch then ended with the
                                 200 full reverse steps of a never-steady draw: 3.02 -> 2.62 + 0.41 ms); with the round-6 full step
                                 (matrix-core products and panels: 19.4 k -> 13.6 k cycles) it is 2 % (5.02 -> 4.92 ms per 4096
                                 draws) and the default.  0: forward and reverse sweep in one kernel (rounds 1-4).  Same recursion,
                                 same records. */
  int32_t reserved_[2];
} dsge_options;
/* fills *opt with the compiled-in defaults */
int dsge_options_init(dsge_options* opt);
/* install / remove *opt for the calls THIS host thread makes in between (nests; copied, the caller may free it) */
int dsge_options_push(const dsge_options* opt);
int dsge_options_pop(void);
/* Drops every record the library measured on the device and reuses across calls: the number of static variables per model size
 * (dsge_options.n_static_hint = -1) and the capacity records of the gensys window path (dsge_options.gensys_shape_cache).  The
 * next call of each model size measures again.  For callers that change the model behind a fixed size, and for the tests. */
int dsge_forget_measured_shapes(void);

/* ---- the kernel-variant switches of dsge_options, one by one (none of them changes a result beyond rounding) ---- */
/* dsge_options.cr_compact: Cycle reduction runs on the column-compact form [A[:,S] | C[:,L]] (S, L = non-zero columns of A and C,
 * detected per draw on the device; zero columns only ever contribute +0.0, so T is bit-identical) whenever
 * |S| + |L| <= 8*ceil(n/8); other draws take the dense kernel.  enable = 0 forces the dense kernel for every
 * draw (used by the tests to compare the two). default 1. */
/* dsge_options.cr_fused_selection: In the fused entry points with solver = cycle_reduction and no residual requested, R = -(C T + B)^-1 D is taken
 * from the final elimination of cycle reduction: T = -A1_hat^-1 A and A1_hat -> B + C T (the difference is of the
 * order of the product of the last iterate's norms, < tol^2), so R = -A1_hat^-1 D comes out of the same Gauss-Jordan
 * sweep (agreement with the explicit formula ~1e-13 relative, tests/test_gpu_parity.py).  enable = 0 always uses the
 * explicit formula in the assemble kernel. default 1. */
/* dsge_options.kalman_steady_tol:
 * Steady-state switch of the fast Kalman kernel.  The covariance recursion of a time-invariant model
 * does not depend on the data; once max|P_{t+1|t} - P_{t|t-1}| <= tol * max|P| (and while the
 * missing-data mask stays the same) the kernel reuses F^-1, K and det F and runs only the mean
 * recursion; a step with a different mask resumes the full update.  tol = 0 never switches (the
 * recursion of pymc_extras' "standard" filter step for step); the default 1e-14 is rounding level:
 * logp moves by < 1e-12 relative (tests/test_gpu_parity.py).  tol in [0, 1e-6]. */
/* dsge_options.kalman_tiny: Small models (selector Z, p <= 3, at most 6 filtered variables) are filtered by a thread-per-draw kernel that keeps
 * the whole reduced state space in registers (64 draws per wavefront).  enable = 0 routes every draw through the
 * wave-per-draw kernels (used by the tests to compare the two). default 1. */
/* dsge_options.kalman_nt_products: Selector design matrix, p <= 8: the two prediction products of a full filter step run in "NT" form on 16-byte aligned
 * rows (W stored transposed, even leading dimension, one ds_read_b128 per two k-steps, stages of four k-steps double-
 * buffered; dsge_kalman_nt.hpp): 15.2 k -> 12.6 k cycles per full step on the 18-variable bench model.  enable = 0 keeps
 * the round-1 kernel (kalman_sel_kernel); same arithmetic up to the summation order of the products (tests compare them).
 * default 1. */
/* dsge_options.cr_fused_deflation: Static-variable deflation (cr_deflation) as ONE launch: QR of the static columns, cycle reduction on the reduced
 * system and the back-substitution of the static rows in a single kernel, the reduced system handed over through LDS
 * (dsge_cr_fused.hpp) instead of three launches with the reduced A, B, C, D, T, R in global memory.  Taken when
 * h + 3 (n - h) + k <= 128 and (n - h) + k <= 64; otherwise, and with enable = 0, the three launches run.  Same results
 * (the arithmetic is the same code). default 1. */
/* dsge_options.cr_four_waves: Cycle reduction on systems of 49..64 variables (after the deflation, if any) runs on FOUR wavefronts per draw
 * (cr_wide_kernel, dsge_cr_wide.hpp: 16 x 16 threads with 4 x 4 register blocks, the panel factorisation on one of the
 * wavefronts) instead of one wavefront with 7 x 7 / 8 x 8 blocks that spill.  enable = 0 keeps the one-wavefront kernels
 * (same algorithm; the norms of the stopping rule are summed in a different order).  default 1. */
/* dsge_options.cr_deflation: Fused evaluation with solver = cycle reduction: variables whose columns of A and C are both exactly zero ("static" in
 * Dynare's partition) are eliminated by a Householder QR of their columns of B before the iteration, which then runs on
 * the n - h dynamic variables (30 of 40 on the SW-shaped systems, 20 of 24 on full_nk); their rows of T and R follow by
 * back-substitution (dsge_cr_deflate.hpp).  Same solution (it is unique), (n - h)^3 instead of n^3 work per iteration.
 * h is measured once per model size (a small launch and a 4-byte read-back on the first call) and verified per draw; a
 * draw with fewer static variables is solved by the full-size kernels.  Not used when the caller asks for the iteration
 * counts or the policy residual.  enable = 0 switches it off (dsge_forget_measured_shapes() drops the measured sizes).  Default on. */
/* dsge_options.cr_two_waves: Column-compact cycle reduction on the 32-wide tile (n or n - h in 25..32): the kernel instance built for two waves per SIMD
 * (256 registers + 528 B of scratch instead of 369 registers): same arithmetic, bit-identical results, 10 % faster.
 * enable = 0 launches the one-wave instance.  Default on. */
/* dsge_options.kalman_order: Fused evaluation: the workgroups of the Kalman launch (and of the gradient path's reverse-sweep launch) are dispatched in
 * descending order of a per-draw key (a counting sort on the device).  The launch's makespan is set by its slowest draws --
 * a persistent model reaches the steady state of the covariance recursion late and keeps one wavefront busy for up to T_len
 * full steps -- so the likely slow draws start first instead of wherever their index puts them (3.29 -> 2.86 ms per 4096
 * SW-shaped draws).  mode 1 (default): key = the draw's cycle-reduction iteration count (free; both grow with the persistence
 * of the model), for the other solvers a spectral-radius estimate of T (24 power-iteration steps, persistence_key_kernel);
 * mode 2: always the latter; mode 0: index order.  Results are unaffected: every draw writes its own logp / status. */
/* dsge_options.pipeline_chunks: dsge_solve_kalman_logp_batched (device pointers) runs batches of >= 1024 draws as n_chunks chunks alternating over two
 * library-owned streams, forked from and joined to the caller's stream by events, so that the straggler tail of one
 * chunk's Kalman launch (a draw whose covariance recursion converges late keeps one wavefront busy for up to T_len full
 * steps) overlaps the solver launch of the next chunk.  Results are identical (the kernels are per-draw).  n_chunks < 2:
 * one pass on the caller's stream (the default: on MI355X the chunks' launches did not overlap enough to pay for the
 * extra straggler tails). */
/* dsge_options.kalman_block: Experimental, OFF by default: once the covariance is frozen and the missing-data mask of the shared panel no longer
 * changes, the fast Kalman kernel hands the rest of the sample to kalman_tail_kernel, which runs the (then linear) mean
 * recursion two steps at a time as one matrix-vector product [R v_t; R v_{t+1}; a_{t+2}] = M [a_t; c_t; c_{t+1}], R'R = F^-1,
 * rows in registers.  It removes a quarter of the kernel's work but not its makespan, which is set by the draws that reach
 * the steady state late or never -- measured 3.14 vs 2.85 ms per 4096 draws with the launch's extra 0.26 ms (DESIGN.md
 * 4.3).  enable = 1 switches it on (tests compare both: same logp to 1e-12). */
/* dsge_options.gensys_split: gensys runs as five launches on the active window of the pencil -- structural deflation; real Hessenberg-triangular
 * reduction and complex QZ + reordering on the (N - z) x (N - z) block the deflation leaves, with H and T sharing one LDS
 * array and the accumulated right transformation kept in HBM/L2; existence/uniqueness (Jacobi SVD); post-processing --
 * 4 / 6 / 6-7 / 10 / 2 draws per CU instead of 1 at N = 52.  enable = 1
 * (default): window path unless the pencil is small (single-launch kernel <= 24 KB of LDS: RBC-sized models) or does not
 * fit; 2: window path whenever it fits; 0: single-launch kernel (tests compare both). */
/* dsge_options.gensys_real_stage: Window path of gensys: implicit double-shift QZ sweeps in REAL arithmetic (Moler-Stewart) at the end of the
 * Hessenberg-triangular launch, in front of the complex single-shift iteration that reproduces zhgeqz's logic (which then only
 * splits the remaining 2 x 2 blocks).  An accelerator: every step is an orthogonal equivalence, T and eu are the same to
 * rounding (test_gensys_real_stage_matches_complex_only).  enable = 0: complex iteration only (round 1-2 behaviour).
 * default 1. */
/* dsge_options.kalman_mfma: The two covariance-prediction products of a full filter step (W = P+[S,S] T', X = T W) on the FP64 matrix core.
 *   2 (default, ABI 9): v_mfma_f64_4x4x4f64 -- four independent 4 x 4 x 4 blocks per issue -- inside the fast selector kernel
 *       (kalman_nt_kernel) for the draws whose state block and retained variables both have five tiles of four (17 .. 20
 *       variables: the Smets-Wouters-shaped models; chosen by n_state_hint, checked per draw, every other draw keeps the VALU
 *       products): W in 35 issues, the upper triangle of X in 20, against 432 FMAs and 144 LDS loads per lane.  Same peak as
 *       the VALU, a quarter of the instruction slots: a full step of a lone wavefront 8.0 k -> 7.2 k cycles.
 *   1: the round-2 experiment -- v_mfma_f64_16x16x4_f64 on the 16 x 16 core tile of the 16- and 24-wide selector instances, VALU
 *       for the fringe.  Slower than the VALU register blocks (4.4 vs 3.3 ms per 4096-draw step: the 18-wide reduced model pads
 *       to 32); kept for comparison (tests compare it with the VALU path).
 *   0: VALU products everywhere. */

/* Debug hook: enable != 0 makes the compact cycle-reduction kernel record the shader cycles draw 0 spends in
 * [0] Gauss-Jordan panels, [1] trailing updates, [2] row gather + staging, [3] products, [4] scatter/updates/
 * norms, [5] the final solve, [6] total, [7] = iterations; cycles_out (host int64[8], may be NULL). */
int dsge_debug_cr_phases(int enable, long long* cycles_out);

/*
 * gensys, batched.  Replaces _gensys_setup + _gensys_core as GensysWrapper / gensys_pt use them
 * (gEconpy/solvers/gensys.py:568-614, :190-395, :657-666, :679-683): ordered complex QZ of the
 * (n + #lead)-dimensional pencil, existence / uniqueness codes, T = G1[:n,:n].
 *   A,B,C : [batch][n][n]   D : [batch][n][k] (may be NULL when R_out is NULL)
 *   tol   : lead-column threshold and `realsmall` (gensys.py:223,587)
 *   T_out : [batch][n][n] (zeros on coincident zeros, gensys.py:255-265)
 *   R_out : [batch][n][k] or NULL; R = -(C T + B)^-1 D as gensys_pt computes it (:681)
 *   eu_out: [batch][3] int32  {1,1,0} unique stable solution; {-2,-2,0} coincident zeros;
 *           eu[2] = number of loose endogenous variables; {-3,-3,0} = QZ failed / model too large
 *   status: [batch] 0 iff eu[0] == 1 && eu[1] == 1 (the Op's `success`, gensys.py:663)
 *   n_lead_hint : upper bound on the number of lead columns of C (0 = unknown); sizes the
 *           on-chip pencil.  n + #lead <= DSGE_MAX_N_GENSYS.
 * 65 <= n <= DSGE_MAX_N_BIG (96), dsge_options.gensys_doubling != 0 only: gensys by spectral division with one workgroup per draw
 * (csrc/dsge_big.hpp: the doubling iteration, then gensys_certify_big_kernel -- one elimination of [B + C T | D | C] gives R and
 * G = (B + C T)^-1 C; eu = {1,1,0} iff rho(T[S,S]) < 1 and rho(G[L,L]) < 1 are certified).  There is no ordered QZ at that size: a
 * draw without the certificate (non-regular, or a root within 2e-4 of the unit circle, > 32 lead or > 64 state columns) gets
 * eu = {-3,-3,0}, status DSGE_ST_NOT_CONVERGED | DSGE_ST_GENSYS_TOO_BIG and T = R = 0 -- a failed draw, never a wrong verdict.
 * The fused dsge_solve_kalman_logp_batched takes solver = gensys at these sizes the same way.
 */
int dsge_gensys_batched(const double* A, const double* B, const double* C, const double* D, int batch, int n,
                        int k, double tol, int n_lead_hint, double* T_out, double* R_out, int32_t* eu_out,
                        int32_t* status, void* stream);
int dsge_gensys_batched_host(const double* A, const double* B, const double* C, const double* D, int batch,
                             int n, int k, double tol, int n_lead_hint, double* T_out, double* R_out,
                             int32_t* eu_out, int32_t* status);

/*
 * gensys on a caller-supplied pencil.  Replaces gensys(g0, g1, c, psi, pi, div, tol) (gEconpy/solvers/gensys.py:398-521 ->
 * _gensys_core :190-395), the interactive numpy entry point: Gamma0 y_t = Gamma1 y_{t-1} + c + Psi z_t + Pi eta_t.
 *   g0, g1 : [batch][N][N]   c : [batch][N] or NULL (= 0)   psi : [batch][N][k]   pi : [batch][N][n_eta]
 *   G1_out : [batch][N][N]  (:336-343)
 *   C_out : [batch][N] = Z G0^-1 [T_mat Q c; (A22 - B22)^-1 Q2 c] as in Sims' gensys.m.  The reference omits the G0^-1 on
 *           this output (:345-357: the stable block is not solved), which makes its C for c != 0 depend on the ordered
 *           Schur basis LAPACK happens to return; with c = 0 -- all gEconpy ever passes (:598) -- both are zero.
 *   impact_out : [batch][N][k]  (:359-365, 392)
 *   gev_out : [batch][N][4] = (Re alpha, Im alpha, Re beta, Im beta) per generalized eigenvalue, stable roots first,
 *             LAPACK's normalisation beta real >= 0 (:253; the order within each group is not defined, as in LAPACK)
 *   eu_out : [batch][3], status : [batch] as dsge_gensys_batched
 * fmat, fwt, ywt and loose of the reference's 9-tuple (:367-393): dsge_gensys_pencil_full_batched below.  Only the column
 * space of Pi matters for the outputs above, so Pi is replaced on the device by an orthonormal basis of it (the existence /
 * uniqueness SVDs then share their right singular vectors).  Everything is resident in LDS: N x N complex H, T, Z and
 * N x (n_eta + k + 1) Q [Pi | Psi | c] -- N <= ~52; a larger pencil returns DSGE_ERR_TOO_LARGE.
 */
int dsge_gensys_pencil_batched(const double* g0, const double* g1, const double* c, const double* psi, const double* pi,
                               int batch, int N, int k, int n_eta, double tol, double* G1_out, double* C_out,
                               double* impact_out, double* gev_out, int32_t* eu_out, int32_t* status, void* stream);
int dsge_gensys_pencil_batched_host(const double* g0, const double* g1, const double* c, const double* psi,
                                    const double* pi, int batch, int N, int k, int n_eta, double tol, double* G1_out,
                                    double* C_out, double* impact_out, double* gev_out, int32_t* eu_out,
                                    int32_t* status);

/*
 * The forward-solution part of gensys' 9-tuple (gEconpy/solvers/gensys.py:367-393), formed on the device in the same
 * launch as G1 / C / impact / gev.  Every pointer may be NULL (= not wanted).  Complex arrays are interleaved (re, im).
 *   f_mat      : [batch][N][N][2]  B22^-1 A22 (:371); the leading nu x nu block is valid (row stride N)
 *   f_wt       : [batch][N][k][2]  -B22^-1 Q2 Psi (:372); rows < nu valid
 *   y_wt       : [batch][N][N][2]  Z G0^-1[:, n_stable:] (:374-379); columns < nu valid (row stride N)
 *   loose      : [batch][N][n_eta] Re(Z G0^-1 [Q1 Pi (I - V V^H); 0]) (:383-393), V = right singular vectors of Q2 Pi with
 *                sigma > realsmall
 *   n_unstable : [batch]           nu (0 when there is no solution: every array above is zero-filled then, :258-264)
 *   pi_raw     : 0 = Pi is replaced by an orthonormal basis of its column space (the default of the plain entry point);
 *                1 = Pi is used as given.  With a UNIQUE solution every output depends on Pi only through its column space and
 *                the two agree.  Without one (eu[1] = 0) G1, impact and loose depend on Pi itself -- Phi = (Q1 Pi)(Q2 Pi)^+ --
 *                and only pi_raw = 1 returns the reference's values.  The existence / uniqueness codes are rank decisions
 *                that the device takes with orthonormal columns: exact for pi_raw = 0, and for pi_raw = 1 when Pi has
 *                orthonormal columns (the [0; I] of every gEconpy pencil, gensys.py:606-611).  For any other Pi run two
 *                launches -- matrices from pi_raw = 1, eu / status from pi_raw = 0 (geconpy_amd.batched.gensys_pencil_batched
 *                does; tests/test_gpu_parity.py::test_gensys_forward_outputs_vs_reference[arbitrary_nonunique]).
 * f_mat, f_wt and y_wt are complex and only defined up to the unitary basis of the unstable block that an ordered Schur
 * form happens to produce (LAPACK's differs from this library's: the reference's own values change with the LAPACK build);
 * the products y_wt f_mat^s f_wt -- all the forward solution ever uses -- and the spectrum of f_mat are invariant and are
 * what tests/test_gpu_parity.py compares with the reference's outputs.
 */
typedef struct dsge_gensys_forward {
  double* f_mat;
  double* f_wt;
  double* y_wt;
  double* loose;
  int32_t* n_unstable;
  int32_t pi_raw;
} dsge_gensys_forward;
int dsge_gensys_pencil_full_batched(const double* g0, const double* g1, const double* c, const double* psi,
                                    const double* pi, int batch, int N, int k, int n_eta, double tol, double* G1_out,
                                    double* C_out, double* impact_out, double* gev_out, int32_t* eu_out,
                                    int32_t* status, const dsge_gensys_forward* forward, void* stream);
int dsge_gensys_pencil_full_batched_host(const double* g0, const double* g1, const double* c, const double* psi,
                                         const double* pi, int batch, int N, int k, int n_eta, double tol,
                                         double* G1_out, double* C_out, double* impact_out, double* gev_out,
                                         int32_t* eu_out, int32_t* status, const dsge_gensys_forward* forward);

/*
 * Blanchard-Kahn eigenvalues.  Replaces compute_bk_eigenvalues / check_bk_condition
 * (gEconpy/model/perturbation.py:412-445, :448-565; the graph twin check_bk_condition_pt :586-625 computes the same
 * counts from a dense eig of the regularised pencil): the generalized eigenvalues of the Sims pencil of _gensys_setup
 * (gensys.py:568-614), lambda_i = beta_i / (alpha_i + tol) with LAPACK's normalisation (beta real, non-negative), sorted
 * by ascending modulus, and the two counts the condition compares.  Runs the reduce + QZ launches of the window path.
 *   eig_re, eig_im : [batch][2n]  the first n_eig[i] = n + n_forward[i] entries of a row are valid, the rest 0
 *   n_forward      : [batch]      columns of C with an absolute column sum > tol
 *   n_unstable     : [batch]      eigenvalues with modulus > 1 (an infinite root, alpha = 0, counts: beta / tol)
 *   status         : [batch]      DSGE_ST_OK, or NOT_CONVERGED | GENSYS_QZ_FAIL (n_eig = 0)
 * The condition holds for draw i iff n_forward[i] == n_unstable[i].
 */
int dsge_bk_eigenvalues_batched(const double* A, const double* B, const double* C, int batch, int n, double tol,
                                double* eig_re, double* eig_im, int32_t* n_eig, int32_t* n_forward,
                                int32_t* n_unstable, int32_t* status, void* stream);
int dsge_bk_eigenvalues_batched_host(const double* A, const double* B, const double* C, int batch, int n, double tol,
                                     double* eig_re, double* eig_im, int32_t* n_eig, int32_t* n_forward,
                                     int32_t* n_unstable, int32_t* status);

/* Debug hook: device int32[batch] that later fast-path Kalman launches (and the second-order filter) fill with the first
 * time step that ran in steady-state mode (-1 = never); NULL stops recording. */
int dsge_debug_kalman_steady_steps(int32_t* steady_at_device);
/* Debug hook: timeline_device (device int64 [batch][8], or NULL to switch it off) receives {start, end -- ticks of the 100 MHz
 * wall clock --, HW_REG_HW_ID, first steady step, start of the time loop, time of the first steady step, 0, 0} of every draw the fast selector filter kernel (kalman_nt_kernel) takes: the
 * schedule of a launch (tools/kalman_timeline.py). */
int dsge_debug_kalman_timeline(long long* timeline_device);

/* Debug hook: enable != 0 makes the selector-path Kalman kernel record the shader cycles draw 0 spends
 * in each of its five per-step phases, [5] the cycles spent in steady-state steps, [6] their number and
 * [7] the kernel total; cycles_out (host int64[16], may be NULL) reads them back.  With dsge_options.kalman_head_draws != 0 the
 * two-wavefront kernel is stamped instead: [0..7] the update wavefront (F + elimination, gain, Tc K / Tc V, wait, a', total, closing
 * wait, full steps), [8..15] the product wavefront (W0, X0, -, wait, correction, steady segments, closing wait, full steps). */
int dsge_debug_kalman_phases(int enable, long long* cycles_out);
/* cr_big_kernel (65 .. 96 variables), first draw of workgroup 0; cycles_out: 16 values.  [0] block loads, [1] eliminations, [2] scatters, [3] products, [4] iterations, [5] total, [8..12] inside the eliminations (see dsge_api.hip) */
int dsge_debug_big_phases(int enable, long long* cycles_out);

/* Debug hook: shader-clock stamps of draw 0 at the phase boundaries of the gensys kernel (start,
 * Hessenberg-triangular, QZ, reordering, SVDs/eu, end).  Device pointers; cycles_out: host int64[6]. */
int dsge_debug_gensys_phases(const double* A, const double* B, const double* C, int batch, int n, double tol,
                             int n_lead_hint, double* T_out, int32_t* eu_out, int32_t* status,
                             long long* cycles_out);
/* Debug hook of the window path: enable = 1 arms the stamp buffer (draw 0 of each launch: reduce [0..4] = start, deflation,
 * triangular T22, Hessenberg H22, stored; QZ [8..11] = start, QZ, reordering, stored; eu [16..19] = start, loaded, SVD, end;
 * post [20..26] = start, loaded, rhs, back-substitution, products, non-state rows, T written; [12], [13] = sweep steps and
 * sweeps, accumulated); cycles_out: host int64[32] or NULL. */
int dsge_debug_gensys_window_phases(int enable, long long* cycles_out);
/* Debug hook: enable != 0 makes every later window-path gensys call (of any host thread: debugging only) time its launches
 * with HIP events on the caller's stream -- the call then synchronises -- into an internal record; ms_out (host float[8] or
 * NULL) receives the record of the last such call BEFORE `enable` takes effect: [0] structural deflation (reduce),
 * [1] Hessenberg-triangular reduction, [2] real double-shift sweeps (the two-draws-per-wavefront launch; ~0 when the sweeps run
 * inside [1]), [3] complex QZ + reordering, [4] existence / uniqueness, [5] post-processing, [6] their sum, [7] draws timed. */
int dsge_debug_gensys_stage_ms(int enable, float* ms_out);

/*
 * Shock-impact matrix and policy residual.  Replaces pt_compute_selection_matrix
 * (gEconpy/solvers/shared.py:74-75; numpy twin cycle_reduction.py:395-396) and the residual
 * of DSGEStateSpace._setup_policy_matrices (gEconpy/model/statespace.py:213):
 *   R = -(C T + B)^-1 D            R_out    : [batch][n][k]
 *   resid = sum((A + B T + C T T)^2)  resid_out: [batch] (NULL to skip; then A may be NULL)
 * n <= DSGE_MAX_N_BIG.
 */
int dsge_selection_batched(const double* A, const double* B, const double* C, const double* D,
                           const double* T, int batch, int n, int k, double* R_out, double* resid_out,
                           void* stream);
int dsge_selection_batched_host(const double* A, const double* B, const double* C, const double* D,
                                const double* T, int batch, int n, int k, double* R_out,
                                double* resid_out);

/*
 * Adjoints of the policy function.  Replaces o1_policy_function_adjoints (gEconpy/solvers/shared.py:12-71),
 * the pullback of GensysWrapper / CycleReductionWrapper (gensys.py:668-676, cycle_reduction.py:117-124):
 * given the cotangent T_bar of T it returns A_bar = S, B_bar = S T', C_bar = S T' T' where S solves
 *   (kron(T, C') + kron(I, T'C') + kron(I, B')) vec(S) = -vec(T_bar)   <=>   (B + C T)' S + C' S T' = -T_bar.
 * The reference factorises the n^2 x n^2 Kronecker matrix; here the equivalent Stein equation is solved
 * by a doubling iteration (valid for a determinate solution: rho(T) < 1 and stable-inverse roots).
 *   B, C, T, T_bar, A_bar, B_bar, C_bar : [batch][n][n], n <= 56;  status : [batch] (non-zero = not converged)
 */
int dsge_policy_adjoints_batched(const double* B, const double* C, const double* T, const double* T_bar,
                                 int batch, int n, double* A_bar, double* B_bar, double* C_bar,
                                 int32_t* status, void* stream);
int dsge_policy_adjoints_batched_host(const double* B, const double* C, const double* T, const double* T_bar,
                                      int batch, int n, double* A_bar, double* B_bar, double* C_bar,
                                      int32_t* status);
/* The doubling iteration of the Stein solve loses digits when the powers of G = -(B + C T)^-T C' grow before they decay; the
 * kernel tracks max_k max|G^(2^k)| of every draw and a second pass refines the draws above 100 once (S += solve(residual),
 * residual = T_bar + M' S + C' S T'): the level of the reference's Kronecker LU (shared.py:53-71).  Debug hook for the tests:
 * mode 0 = that rule (default), 1 = refine every draw, 2 = never refine.  Process-wide. */
int dsge_debug_adjoint_refine(int mode);

/*
 * Pullback of the shock-impact matrix.  The reference's R = -solve(C @ T + B, D) is plain differentiable pytensor
 * (pt_compute_selection_matrix, gEconpy/solvers/shared.py:74-75), so pytensor.grad flows through it into B, C, D and T;
 * this is that reverse-mode rule as one launch.  With M = C T + B and the cotangent R_bar of R:
 *   G = -M^-T R_bar,   D_bar = G,   B_bar = G R',   C_bar = G R' T',   T_bar = C' G R'
 *   B, C, T, B_bar, C_bar, T_bar : [batch][n][n]   R, R_bar, D_bar : [batch][n][k]   n <= 56
 * T_bar is the contribution of R only: the caller adds the direct cotangent of T and passes the sum on to
 * dsge_policy_adjoints_batched.
 */
int dsge_selection_adjoints_batched(const double* B, const double* C, const double* T, const double* R,
                                    const double* R_bar, int batch, int n, int k, double* B_bar, double* C_bar,
                                    double* D_bar, double* T_bar, void* stream);
int dsge_selection_adjoints_batched_host(const double* B, const double* C, const double* T, const double* R,
                                         const double* R_bar, int batch, int n, int k, double* B_bar, double* C_bar,
                                         double* D_bar, double* T_bar);

/*
 * gEcon recursion residual norms, the `deterministic_norm` / `stochastic_norm` Deterministics of
 * DSGEStateSpace.build_statespace_graph (gEconpy/model/statespace.py:1181-1204).  With the state mask
 * s (variables that appear at t-1 and at t in some equation, :1186-1193):
 *   deterministic_norm = ||A[:,s] + B T[:,s] + C T[:,s] T[s][:,s]||_F
 *   stochastic_norm    = ||B R + C T[:,s] R[s] + D||_F
 *   state_mask : [n] int32 (non-zero = state), shared by all draws;  outputs: [batch] each.  n <= 56.
 */
int dsge_policy_norms_batched(const double* A, const double* B, const double* C, const double* D,
                              const double* T, const double* R, const int32_t* state_mask, int batch, int n,
                              int k, double* det_norm_out, double* stoch_norm_out, void* stream);
int dsge_policy_norms_batched_host(const double* A, const double* B, const double* C, const double* D,
                                   const double* T, const double* R, const int32_t* state_mask, int batch,
                                   int n, int k, double* det_norm_out, double* stoch_norm_out);

/*
 * Backward-looking direct solve.  Replaces solve_policy_function_with_backward_direct
 * (gEconpy/solvers/backward_looking.py:102-134): T = (-B)^-1 A, R = -B^-1 D.
 */
int dsge_backward_direct_batched(const double* A, const double* B, const double* D, int batch, int n,
                                 int k, double* T_out, double* R_out, void* stream);
int dsge_backward_direct_batched_host(const double* A, const double* B, const double* D, int batch,
                                      int n, int k, double* T_out, double* R_out);

/*
 * Stationary state covariance.  Replaces
 *   P0 = solve_discrete_lyapunov(T, R Q R^T)        (gEconpy/model/statespace.py:814-815)
 * by the doubling (Smith) iteration; also returns sym(R Q R^T).
 *   T : [batch][m][m]  R : [batch][m][k]  Q per q_mode
 *   P0_out, RQR_out : [batch][m][m] (RQR_out may be NULL)   status |= DSGE_ST_LYAP_FAIL
 */
int dsge_lyapunov_batched(const double* T, const double* R, const double* Q, int q_mode, int batch,
                          int m, int k, double* P0_out, double* RQR_out, int32_t* status, void* stream);
int dsge_lyapunov_batched_host(const double* T, const double* R, const double* Q, int q_mode, int batch,
                               int m, int k, double* P0_out, double* RQR_out, int32_t* status);

/*
 * Model-implied autocovariance / autocorrelation matrices.  Replaces _compute_autocovariance_matrix
 * (gEconpy/model/statistics/covariance.py:133-161) and the per-draw graph of
 * DSGEStateSpace.sample_autocorrelation_matrices (gEconpy/model/statespace.py:1262-1300):
 *   Sigma = dlyap(T, R Q R'),  T_step = T^lag_step,  G_k = T_step^k Sigma  (k = 0..n_lags)
 *   Z == NULL : out[draw][k] = G_k                      (m x m, latent states)
 *   Z != NULL : out[draw][k] = Z G_k Z', lag 0 + diag(Hdiag)   (p x p, observed series; Z [p][m] shared)
 *   correlation != 0 : every matrix divided by std std', std = sqrt(diag(out[draw][0]))
 *   acf_out   : [batch][n_lags+1][dim][dim], dim = Z ? p : m
 *   Sigma_out : [batch][m][m] or NULL (device variant: NULL uses library scratch)
 *   status    : [batch] out; DSGE_ST_LYAP_FAIL (rho(T) >= 1) fills that draw's matrices with NaN
 */
int dsge_autocorrelation_batched(const double* T, const double* R, const double* Q, int q_mode, const double* Z,
                                 const double* Hdiag, int batch, int m, int k, int p, int n_lags, int lag_step,
                                 int correlation, double* acf_out, double* Sigma_out, int32_t* status, void* stream);
int dsge_autocorrelation_batched_host(const double* T, const double* R, const double* Q, int q_mode, const double* Z,
                                      const double* Hdiag, int batch, int m, int k, int p, int n_lags, int lag_step,
                                      int correlation, double* acf_out, double* Sigma_out, int32_t* status);

/*
 * Kalman-filter log-likelihood, batched over draws, "standard" filter.  Replaces the scan
 * that PyMCStateSpace.build_statespace_graph builds for DSGEStateSpace
 * (gEconpy/model/statespace.py:1151-1157; recursion restated in SURVEY.md Appendix B.4)
 * with a0 = 0 (statespace.py:812), state intercept c = 0 and P0 = dlyap(T, R Q R^T).
 *   T : [batch][m][m]   R : [batch][m][k]   Q per q_mode
 *   Z : [p][m] (z_batched=0) or [batch][p][m]     (statespace.py:260-332)
 *   d : NULL (=0), [p] or [batch][p]              (statespace.py:334-388)
 *   Hdiag : NULL (=0), [p] or [batch][p]          (statespace.py:800-810)
 *   y : [T_len][p] shared by all draws; NaN or == missing_fill marks a missing entry
 *       (statespace.py:1143)
 *   status_io : [batch] in/out.  A draw whose incoming status is non-zero is skipped and
 *       gets logp = -inf (the -inf Potentials of statespace.py:1206-1215).
 *   logp_out  : [batch]
 * Performance hints (never affect results; each is verified per draw on the device and a draw
 * that violates one is re-run by the general kernel):
 *   n_state_hint    : upper bound on the number of non-zero columns of T (the model's state
 *                     variables, i.e. non-zero columns of A); 0 = unknown
 *   z_selector_hint : non-zero if every row of Z has exactly one non-zero entry, in distinct
 *                     columns (the pure-selector design of statespace.py:282-296)
 */
int dsge_kalman_logp_batched(const double* T, const double* R, const double* Q, int q_mode,
                             const double* Z, int z_batched, const double* d, int d_batched,
                             const double* Hdiag, int h_batched, const double* y, int batch, int m,
                             int k, int p, int T_len, double jitter, double missing_fill,
                             int n_state_hint, int z_selector_hint, double* logp_out,
                             int32_t* status_io, void* stream);
int dsge_kalman_logp_batched_host(const double* T, const double* R, const double* Q, int q_mode,
                                  const double* Z, int z_batched, const double* d, int d_batched,
                                  const double* Hdiag, int h_batched, const double* y, int batch,
                                  int m, int k, int p, int T_len, double jitter, double missing_fill,
                                  int n_state_hint, int z_selector_hint, double* logp_out,
                                  int32_t* status_io);

/*
 * Per-step OUTPUTS of the "standard" filter, batched over draws: what DSGEStateSpace.build_statespace_graph(...,
 * save_kalman_filter_outputs_in_idata=True) registers (gEconpy/model/statespace.py:1145, 1151-1157; pymc_extras
 * filtered / predicted states and covariances, per-observation log-likelihood) -- the post-estimation pass, not the hot loop
 * (full recursion, no steady-state switch, no state reduction).  Arguments as dsge_kalman_logp_batched, plus
 *   ll_out     : [batch][T_len]      ll_t (0 for a step with every entry missing); sum_t ll_t = the logp of the fast kernels
 *   a_pred_out : [batch][T_len][m]   a_{t|t-1} (a_{0|-1} = 0)          a_filt_out : [batch][T_len][m]  a_{t|t}
 *   p_pred_out, p_filt_out : full_cov = 0: [batch][T_len][m] the DIAGONALS of P_{t|t-1}, P_{t|t};
 *                            full_cov != 0: [batch][T_len][m][m] the matrices;  every output except ll_out may be NULL
 *   status_io  : [batch] in/out; a draw with a non-zero incoming status is skipped: EVERY requested output of it is NaN (ll and
 *                the states / covariances); a filter that goes non-finite sets DSGE_ST_FILTER_NONFINITE and leaves the values of
 *                the steps up to that point
 */
int dsge_kalman_filter_outputs_batched(const double* T, const double* R, const double* Q, int q_mode, const double* Z,
                                       int z_batched, const double* d, int d_batched, const double* Hdiag, int h_batched,
                                       const double* y, int batch, int m, int k, int p, int T_len, double jitter,
                                       double missing_fill, double* ll_out, double* a_pred_out, double* a_filt_out,
                                       double* p_pred_out, double* p_filt_out, int full_cov, int32_t* status_io, void* stream);
int dsge_kalman_filter_outputs_batched_host(const double* T, const double* R, const double* Q, int q_mode, const double* Z,
                                            int z_batched, const double* d, int d_batched, const double* Hdiag, int h_batched,
                                            const double* y, int batch, int m, int k, int p, int T_len, double jitter,
                                            double missing_fill, double* ll_out, double* a_pred_out, double* a_filt_out,
                                            double* p_pred_out, double* p_filt_out, int full_cov, int32_t* status_io);

/*
 * Fused evaluation A,B,C,D -> T,R -> P0 -> logp: one call per MCMC step for the whole draw
 * batch (the per-evaluation hot loop of SURVEY.md section 3A; what
 * DSGEStateSpace._setup_policy_matrices + make_symbolic_graph + the filter compute,
 * statespace.py:197-222,725-820,1151-1157).  Intermediate T,R,P0 stay in library scratch
 * unless the optional outputs are given.
 *   solver : DSGE_SOLVER_*; tol/max_iter as configure(...) passes them (statespace.py:835-836)
 *   T_out [batch][n][n], R_out [batch][n][k], resid_out [batch], n_iter_out [batch]: optional
 *   status_out : [batch] (required)    logp_out : [batch] (required)
 * n <= DSGE_MAX_N (64); with a cycle-reduction solver n <= DSGE_MAX_N_BIG (96): the solver then runs with one workgroup per draw
 * and the filter on the model restricted to F = {state variables} u {observed variables} -- exact, because every column of T
 * outside the state variables is zero.  F is measured on the device (the call synchronises `stream` once; the hints are not
 * used); more than 64 variables in F: DSGE_ERR_TOO_LARGE -- F is measured before the solver is launched, so nothing of the
 * call (of the current chunk, when the batch is processed in chunks) has been computed or written.
 */
int dsge_solve_kalman_logp_batched(const double* A, const double* B, const double* C, const double* D,
                                   const double* Q, int q_mode, const double* Z, int z_batched,
                                   const double* d, int d_batched, const double* Hdiag, int h_batched,
                                   const double* y, int batch, int n, int k, int p, int T_len,
                                   int solver, double tol, int max_iter, double jitter,
                                   double missing_fill, int n_state_hint, int z_selector_hint,
                                   int n_lead_hint, double* logp_out, int32_t* status_out,
                                   double* T_out, double* R_out, double* resid_out,
                                   int32_t* n_iter_out, void* stream);
int dsge_solve_kalman_logp_batched_host(const double* A, const double* B, const double* C,
                                        const double* D, const double* Q, int q_mode, const double* Z,
                                        int z_batched, const double* d, int d_batched,
                                        const double* Hdiag, int h_batched, const double* y, int batch,
                                        int n, int k, int p, int T_len, int solver, double tol,
                                        int max_iter, double jitter, double missing_fill,
                                        int n_state_hint, int z_selector_hint, int n_lead_hint,
                                        double* logp_out, int32_t* status_out, double* T_out,
                                        double* R_out, double* resid_out, int32_t* n_iter_out);

/*
 * Fused evaluation with the un-permutation and state augmentation of DSGEStateSpace.make_symbolic_graph
 * (gEconpy/model/statespace.py:781-820): A,B,C,D -> T,R (solver order) -> resid (:213) -> T = T[inv][:,inv],
 * R = R[inv] (:217-220) -> T_aug = [[T,0],[F,C]], R_aug = [R;0] (_augment_transition :598-650,
 * _append_obs_lag_block :652-694, _augment_selection :696-723) -> P0 = dlyap(T_aug, R_aug Q R_aug') (:814-815)
 * -> Kalman logp with the m-dimensional design matrix of _make_design_matrix (:260-332).
 *   m : augmented state dimension, n <= m <= DSGE_MAX_N_BIG (96; round 6).  Beyond 64 the filter runs on the model restricted to
 *       F = {non-zero columns of T_aug} u {observed variables} -- exact: x_t[F] depends on x_{t-1}[F] only, y_t on x_t[F] only --,
 *       measured on the device (the call then synchronises its stream once); |F| > 64 is DSGE_ERR_TOO_LARGE, returned with
 *       T_aug_out, R_aug_out, resid_out and the status words of the solve already written (the exception to "nothing computed":
 *       F is a property of the SOLVED model).  A 40-variable model with 18 states takes 46 chain states.
 *   Z : [p][m] or [batch][p][m]
 *   inv_var_order : [n] int32 or NULL (identity)
 *   link_rows/link_cols : [n_links] int32; T_aug[link_rows[i]][link_cols[i]] = 1.0 -- the unit entries of the
 *       constant blocks F (model variable -> first slot of its cumulator / lag chain) and C (slot -> next slot);
 *       rows in [n, m)
 *   T_aug_out [batch][m][m], R_aug_out [batch][m][k], resid_out [batch] : optional
 * n_state_hint / z_selector_hint refer to the AUGMENTED system (non-zero columns of T_aug, rows of Z).
 */
int dsge_solve_kalman_logp_augmented_batched(const double* A, const double* B, const double* C, const double* D,
                                             const double* Q, int q_mode, const double* Z, int z_batched,
                                             const double* d, int d_batched, const double* Hdiag, int h_batched,
                                             const double* y, int batch, int n, int k, int p, int T_len, int solver,
                                             double tol, int max_iter, double jitter, double missing_fill, int m,
                                             const int32_t* inv_var_order, int n_links, const int32_t* link_rows,
                                             const int32_t* link_cols, int n_state_hint, int z_selector_hint,
                                             int n_lead_hint, double* logp_out, int32_t* status_out, double* T_aug_out,
                                             double* R_aug_out, double* resid_out, void* stream);
int dsge_solve_kalman_logp_augmented_batched_host(const double* A, const double* B, const double* C, const double* D,
                                                  const double* Q, int q_mode, const double* Z, int z_batched,
                                                  const double* d, int d_batched, const double* Hdiag, int h_batched,
                                                  const double* y, int batch, int n, int k, int p, int T_len, int solver,
                                                  double tol, int max_iter, double jitter, double missing_fill, int m,
                                                  const int32_t* inv_var_order, int n_links, const int32_t* link_rows,
                                                  const int32_t* link_cols, int n_state_hint, int z_selector_hint,
                                                  int n_lead_hint, double* logp_out, int32_t* status_out,
                                                  double* T_aug_out, double* R_aug_out, double* resid_out);

/*
 * Fused evaluation WITH reverse-mode gradient: logp and its cotangents with respect to A, B, C, D, the shock
 * variances q, the observation intercept d and the measurement-error variances Hdiag, per draw.  This is what
 * pytensor autodiff produces for the reference's logp graph (solver pullback gensys.py:668-676 /
 * cycle_reduction.py:117-124 = o1_policy_function_adjoints shared.py:12-71; R = -(C T + B)^-1 D shared.py:74-75;
 * P0 = solve_discrete_lyapunov statespace.py:814-815; the filter scan statespace.py:1151-1157), written as four
 * kernels: Kalman reverse sweep on the reduced model (stores every predicted (a_t, P_t) in library scratch, the
 * batch is processed in chunks of <= 16 GiB of scratch), reverse of the assembly, policy-function adjoints.
 *   q, q_batched : q_batched is a DSGE_Q_* mode (the name is historical): 0 / 1 = diagonal variances [k] / [batch][k]
 *       (sigma_i^2, statespace.py:252-258), 2 / 3 = full symmetric Q [k][k] / [batch][k][k] (full_covariance, :247-251)
 *   Z : selector design matrix (one non-zero per row, distinct columns), [p][n] or [batch][p][n];  p <= 8;  n <= 56
 *   A_bar,B_bar,C_bar : [batch][n][n];  D_bar : [batch][n][k];  q_bar : [batch][k] (also for a shared q: sum over
 *       the batch for a joint logp); for a full Q [batch][k][k] = R' Gbar R, the cotangent of ALL k x k entries taken as
 *       independent (symmetric; chain it through the caller's parametrisation of Q);  d_bar, h_bar : [batch][p] or NULL
 *   Contract: the columns of A that are exactly zero (non-state variables) are treated as structurally zero -- T has
 *   exactly-zero columns there for every parameter value, so A_bar is meaningful on the non-zero columns of A only
 *   (the others multiply dA = 0 in any chain rule through the model's Jacobians).  No cotangent is produced for Z, y.
 *   n_filter_hint : upper bound on the number of variables the filter keeps, |S u O| = non-zero columns of A plus
 *       observed non-states (0 = unknown: tile sized for n); it only sizes the tile, a draw that exceeds it is flagged.
 *   A draw whose design matrix is not a selector or whose reduced model exceeds the tile gets
 *   DSGE_ST_GRAD_UNSUPPORTED, logp = NaN and zero cotangents.
 */
int dsge_solve_kalman_logp_grad_batched(const double* A, const double* B, const double* C, const double* D,
                                        const double* q, int q_batched, const double* Z, int z_batched,
                                        const double* d, int d_batched, const double* Hdiag, int h_batched,
                                        const double* y, int batch, int n, int k, int p, int T_len, int solver,
                                        double tol, int max_iter, double jitter, double missing_fill, int n_filter_hint,
                                        int n_lead_hint, double* logp_out, int32_t* status_out, double* A_bar,
                                        double* B_bar, double* C_bar, double* D_bar, double* q_bar, double* d_bar,
                                        double* h_bar, void* stream);
int dsge_solve_kalman_logp_grad_batched_host(const double* A, const double* B, const double* C, const double* D,
                                             const double* q, int q_batched, const double* Z, int z_batched,
                                             const double* d, int d_batched, const double* Hdiag, int h_batched,
                                             const double* y, int batch, int n, int k, int p, int T_len, int solver,
                                             double tol, int max_iter, double jitter, double missing_fill,
                                             int n_filter_hint, int n_lead_hint, double* logp_out, int32_t* status_out,
                                             double* A_bar, double* B_bar, double* C_bar, double* D_bar, double* q_bar,
                                             double* d_bar, double* h_bar);
/*
 * The same with ANY design matrix (observation equations: rows of Z that are linear combinations of the variables,
 * _make_design_matrix, gEconpy/model/statespace.py:298-332, which pytensor differentiates through), and with the cotangent of Z
 * itself for a parameter-dependent design matrix.  The observed combinations o_t = Z x_t are carried as p extra variables
 * (T_aug = [[T, 0], [Z T, 0]], R_aug = [R; Z R], selector on the new variables: the same likelihood function), the reverse sweep
 * runs on that model of n + p <= 56 variables and the cotangents are mapped back (csrc/dsge_augment.hpp).
 *   n_state_hint : number of STATE variables (non-zero columns of A), 0 = unknown; it sizes the filter tile (states + p)
 *   Z_bar        : [batch][p][n] or NULL -- cotangent of every entry of Z (also for a shared Z: sum over the batch for a joint
 *                  logp); everything else as dsge_solve_kalman_logp_grad_batched.  A selector Z is a valid input (slower here).
 */
int dsge_solve_kalman_logp_grad_dense_z_batched(const double* A, const double* B, const double* C, const double* D,
                                                const double* q, int q_batched, const double* Z, int z_batched, const double* d,
                                                int d_batched, const double* Hdiag, int h_batched, const double* y, int batch,
                                                int n, int k, int p, int T_len, int solver, double tol, int max_iter,
                                                double jitter, double missing_fill, int n_state_hint, int n_lead_hint,
                                                double* logp_out, int32_t* status_out, double* A_bar, double* B_bar,
                                                double* C_bar, double* D_bar, double* q_bar, double* d_bar, double* h_bar,
                                                double* Z_bar, void* stream);
int dsge_solve_kalman_logp_grad_dense_z_batched_host(const double* A, const double* B, const double* C, const double* D,
                                                     const double* q, int q_batched, const double* Z, int z_batched,
                                                     const double* d, int d_batched, const double* Hdiag, int h_batched,
                                                     const double* y, int batch, int n, int k, int p, int T_len, int solver,
                                                     double tol, int max_iter, double jitter, double missing_fill,
                                                     int n_state_hint, int n_lead_hint, double* logp_out, int32_t* status_out,
                                                     double* A_bar, double* B_bar, double* C_bar, double* D_bar, double* q_bar,
                                                     double* d_bar, double* h_bar, double* Z_bar);

/*
 * The fused entry points with per-call options (opt == NULL: the compiled-in defaults); otherwise identical to the
 * functions of the same name without the suffix.
 */
int dsge_solve_kalman_logp_batched_opt(const dsge_options* opt, const double* A, const double* B, const double* C,
                                       const double* D, const double* Q, int q_mode, const double* Z, int z_batched,
                                       const double* d, int d_batched, const double* Hdiag, int h_batched,
                                       const double* y, int batch, int n, int k, int p, int T_len, int solver,
                                       double tol, int max_iter, double jitter, double missing_fill, int n_state_hint,
                                       int z_selector_hint, int n_lead_hint, double* logp_out, int32_t* status_out,
                                       double* T_out, double* R_out, double* resid_out, int32_t* n_iter_out,
                                       void* stream);
int dsge_solve_kalman_logp_batched_host_opt(const dsge_options* opt, const double* A, const double* B, const double* C,
                                            const double* D, const double* Q, int q_mode, const double* Z,
                                            int z_batched, const double* d, int d_batched, const double* Hdiag,
                                            int h_batched, const double* y, int batch, int n, int k, int p, int T_len,
                                            int solver, double tol, int max_iter, double jitter, double missing_fill,
                                            int n_state_hint, int z_selector_hint, int n_lead_hint, double* logp_out,
                                            int32_t* status_out, double* T_out, double* R_out, double* resid_out,
                                            int32_t* n_iter_out);
int dsge_solve_kalman_logp_grad_batched_opt(const dsge_options* opt, const double* A, const double* B, const double* C,
                                            const double* D, const double* q, int q_batched, const double* Z,
                                            int z_batched, const double* d, int d_batched, const double* Hdiag,
                                            int h_batched, const double* y, int batch, int n, int k, int p, int T_len,
                                            int solver, double tol, int max_iter, double jitter, double missing_fill,
                                            int n_filter_hint, int n_lead_hint, double* logp_out, int32_t* status_out,
                                            double* A_bar, double* B_bar, double* C_bar, double* D_bar, double* q_bar,
                                            double* d_bar, double* h_bar, void* stream);
int dsge_solve_kalman_logp_grad_batched_host_opt(const dsge_options* opt, const double* A, const double* B,
                                                 const double* C, const double* D, const double* q, int q_batched,
                                                 const double* Z, int z_batched, const double* d, int d_batched,
                                                 const double* Hdiag, int h_batched, const double* y, int batch, int n,
                                                 int k, int p, int T_len, int solver, double tol, int max_iter,
                                                 double jitter, double missing_fill, int n_filter_hint, int n_lead_hint,
                                                 double* logp_out, int32_t* status_out, double* A_bar, double* B_bar,
                                                 double* C_bar, double* D_bar, double* q_bar, double* d_bar,
                                                 double* h_bar);

/*
 * Second-order perturbation + pruned-state-space quasi-likelihood, batched over draws (BASELINE.json configs[4]).
 * The reference has NO such solver -- it raises NotImplementedError for order != 1 (gEconpy/model/perturbation.py:97-98,
 * gEconpy/model/model.py:1433-1434, 1614-1615) -- so this entry point replaces that raise; its results are defined by
 * oracle/second_order.py (Schmitt-Grohe & Uribe 2004; pruning after Kim, Kim, Schaumburg & Sims 2008 / Andreasen,
 * Fernandez-Villaverde & Rubio-Ramirez 2018) and are parity-unpinned against the reference by construction.
 *   A,B,C,D -> T,R by the first-order solver (solver: DSGE_SOLVER_CYCLE_REDUCTION or DSGE_SOLVER_GENSYS), then
 *   y_t = T y- + R u + 1/2 [g_yy (y- (x) y-) + 2 g_yu (y- (x) u) + g_uu (u (x) u) + g_ss]  from the model Hessian, then the
 *   Gaussian ("standard" filter) likelihood of the pruned system on z = [x_f[U]; x_s[U]; vech(x_f[S] x_f[S]')], started
 *   from its stationary mean and covariance, observation y = Z (x_f + x_s) + d.
 *   hess_idx : [nnz][3] int32 DEVICE, (equation, z_a, z_b) with z_a <= z_b over z = [y-; y; y+; u] (3 n + k entries), sorted
 *              by equation: the sparsity pattern of d2F_i / dz_a dz_b, shared by all draws
 *   hess_val : [batch][nnz] the values per draw
 *   q        : shock variances (diagonal Sigma), [k] (q_batched = 0) or [batch][k]
 *   Z [p][n], d [p] or NULL, Hdiag [p] or NULL, y [T_len][p]: shared by the draws; p <= 8
 *   state_idx / lead_idx / ret_idx : HOST int32 arrays (model structure, read during the call): the non-zero columns of A
 *              (n_state <= 24), the non-zero columns of C, and the variables the filter retains = the states followed by the
 *              observed non-states (n_ret <= 40); 2 n_ret + n_state (n_state + 1) / 2 <= 208, k <= min(n_state, 12).
 *              A draw that violates the structure gets DSGE_ST_SECOND_ORDER_UNSUPPORTED and logp = -inf.
 *   T_out, R_out, gyy_out [batch][n][n_state][n_state], gyu_out [batch][n][n_state][k], guu_out [batch][n][k][k],
 *   gss_out [batch][n] : optional
 * The steady-state switch (dsge_options.kalman_steady_tol) applies with a scale-free test, (dP_ij)^2 <= (100 tol)^2 P_ii P_jj
 * for every entry; tol = 0 runs the full recursion.
 *   stage_ms : HOST float[4] or NULL; non-NULL makes the call synchronise and report the durations (ms) of the first-order
 *              solve, the coefficient / pruned-system set-up, the stationary covariance and the filter (last chunk)
 */
int dsge_second_order_logp_batched(const double* A, const double* B, const double* C, const double* D,
                                   const int32_t* hess_idx, int nnz, const double* hess_val, const double* q, int q_batched,
                                   const double* Z, const double* d, const double* Hdiag, const double* y, int batch, int n,
                                   int k, int p, int T_len, int solver, double tol, int max_iter, double jitter,
                                   double missing_fill, const int32_t* state_idx, int n_state, const int32_t* lead_idx,
                                   int n_lead, const int32_t* ret_idx, int n_ret, double* logp_out, int32_t* status_out,
                                   double* T_out, double* R_out, double* gyy_out, double* gyu_out, double* guu_out,
                                   double* gss_out, float* stage_ms, void* stream);
int dsge_second_order_logp_batched_host(const double* A, const double* B, const double* C, const double* D,
                                        const int32_t* hess_idx, int nnz, const double* hess_val, const double* q,
                                        int q_batched, const double* Z, const double* d, const double* Hdiag, const double* y,
                                        int batch, int n, int k, int p, int T_len, int solver, double tol, int max_iter,
                                        double jitter, double missing_fill, const int32_t* state_idx, int n_state,
                                        const int32_t* lead_idx, int n_lead, const int32_t* ret_idx, int n_ret,
                                        double* logp_out, int32_t* status_out, double* T_out, double* R_out, double* gyy_out,
                                        double* gyu_out, double* guu_out, double* gss_out);

/* Debug hook: enable != 0 makes the second-order filter kernel record the shader cycles draw 0 spends in [0] P Z', F, the gain;
 * [1] the pass over Az' (predicted mean, Az K, Az V); [2] the first product; [3] the second product + epilogue; [4] steady
 * steps; [5] = number of full steps, [6] = number of steady steps, [7] = kernel total; cycles_out: host int64[8] or NULL. */
int dsge_debug_second_order_phases(int enable, long long* cycles_out);

/*
 * Timing hook for bench.py: runs `reps` back-to-back launches of the fused pipeline's
 * kernels on `stream` bracketed by hipEvents and reports the average duration of each
 * kernel in milliseconds: ms_out[0] = solver, [1] = selection+Lyapunov, [2] = Kalman.
 * Same arguments as dsge_solve_kalman_logp_batched (device pointers).
 */
int dsge_profile_pipeline(const double* A, const double* B, const double* C, const double* D,
                          const double* Q, int q_mode, const double* Z, int z_batched, const double* d,
                          int d_batched, const double* Hdiag, int h_batched, const double* y, int batch,
                          int n, int k, int p, int T_len, int solver, double tol, int max_iter,
                          double jitter, double missing_fill, int n_state_hint, int z_selector_hint,
                          int n_lead_hint, double* logp_out, int32_t* status_out, int reps, float* ms_out,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DSGE_HIP_H */

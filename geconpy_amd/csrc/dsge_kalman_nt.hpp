// Kalman filter log-likelihood, selector design matrix: "kalman_nt_kernel" -- the fast path of round 2.
//
// Same algorithm as kalman_sel_kernel<BS, true> (dsge_kalman2.hpp: exact reduction to the retained variables in a
// states-first ordering, stationary initial covariance by doubling, p x p inverse by Gauss-Jordan, P+ = P - K (P Zm' +
// jit_V K)' + jit_P I (FilterConv, dsge_device.hpp), steady-state switch with a register-resident mean recursion; the recursion is the one restated in
// SURVEY.md Appendix B.4 for statespace.py:1151-1157), rebuilt twice in round 2:
//
// (1) the two prediction products of a full step -- W = P+[S,S] Tc' and X = Tc W -- in "NT" form: W is stored TRANSPOSED,
//     so both products contract over the second index of two row-major operands,
//       W[i][j] = sum_k P+[i][k] Tc[j][k],        X[r][c] = sum_k Tc[r][k] Wt[c][k],
//     on an EVEN leading dimension (NP + 2: 16-byte aligned rows, one ds_read_b128 per two k-steps), stages of four k-steps
//     (12 loads, 36 FMAs for 3 x 3 blocks), two stages in flight: full step 15.2 k -> 12.6 k cycles.
// (2) the measurement update as the latency chain of a lone wave (tools/latency_probe): lane l works on row l & 7 of F in
//     eight replicated 8-lane groups -- row-per-lane Gauss-Jordan with the pivot row through SGPRs and unscaled rows, the
//     gain K = (P Zm') Finv from registers (Finv symmetric: no round trip through LDS, one fence less), the steady test
//     by ballot, the mean prediction as the spare padding column of the X product, per-lane Kahan shares of the quadratic
//     form, sym(R Q R')[U,U] formed in the prologue from R and q: 12.6 k -> 8.9 k cycles per full step for a lone wave
//     (tools/kalman_phases.py 1), launch 1.10 -> 0.85 ms per 4096 draws.
//
// Measured on the way and NOT used: observation-at-a-time updates, a panel recursion with one row
// per lane, a per-lane Cholesky of F, the predicted-form recursion, raised wave priority for the slowest draws, a two-wave
// build of the 32-wide tile.
#pragma once
#include "dsge_kalman2.hpp"
#include "dsge_kalman_rec.hpp"

namespace dsge {

// SK: compile-time capacity of the state block (columns of Tc / W' / Pc that can be non-zero; a multiple of four, >= the
// run-time s_cap rounded up to the tile's block size).  The rows of the three LDS matrices are SK + 2 doubles long instead of
// NP + 2 -- the columns beyond the state block are structural zeros that nothing reads -- and the steady-state loop keeps SK
// instead of NP entries of its row of T in registers.  SK = NP is the generic instance; SK = 20 (18 states, the SW-shaped
// models) takes the 32-wide tile from 28 KB to exactly 20 KB of LDS: eight draws per CU instead of five (round 4).
template <int BS, int SK = 8 * BS>
struct KntSmem {
  static constexpr int NP = Tile<BS>::NP, LDK = SK + 2, PS = 10;
  static_assert(SK % 4 == 0 && SK <= NP && SK >= 8, "SK: a multiple of four in 8..NP");
  static constexpr int WT = (NP * LDK > NP * PS) ? NP * LDK : NP * PS;  // W' buffer; the V panel aliases it
  // doubles: Tc NP*LDK, Wt WT, Pc s_cap*LDK, PZt, Ks NP*PS each, av NP, af NP, vv/dd/hh/zv 8 each; ints perm NP, zpos 8
  __host__ __device__ static constexpr size_t doubles(int s_cap) {
    return (size_t)NP * LDK + WT + (size_t)s_cap * LDK + 2 * (size_t)NP * PS + 2 * NP + 32 + NP / 2 + 4;
  }
  static size_t bytes(int s_cap) { return sizeof(double) * doubles(s_cap); }
};

// register block -> LDS, columns < CAP only (CAP = 0: all of them)
template <int BS, int CAP>
__device__ __forceinline__ void knt_store_cols(const double (&x)[BS][BS], double* s, int ld, int lr, int lc) {
#pragma unroll
  for (int i = 0; i < BS; ++i)
#pragma unroll
    for (int j = 0; j < BS; ++j)
      if (CAP == 0 || lc * BS + j < CAP) s[(lr * BS + i) * ld + lc * BS + j] = x[i][j];
}

// acc += A[rows lr*BS.., :K] * B[rows lc*BS.., :K]'  -- both operands row-major along k with the even stride LD
// (16-byte aligned rows): one ds_read_b128 per row and k-pair, stages of four k-steps, two stages in flight.
// K is rounded up to a multiple of four: the callers keep the padding columns zero / finite.
// SB (single buffer): stages of TWO k-steps, one stage in flight -- a quarter of the operand registers of the double-buffered
// four-step stages (BS = 4: 32 instead of 128 VGPRs); the two-waves-per-SIMD build of the 32-wide tile needs them.
template <int BS, int LD>
__device__ __forceinline__ void mm_nt_sb(double (&acc)[BS][BS], const double* A, const double* B, int K, int lr, int lc) {
  const double2* ap = reinterpret_cast<const double2*>(A + lr * BS * LD);
  const double2* bp = reinterpret_cast<const double2*>(B + lc * BS * LD);
  constexpr int RS = LD / 2;
  const int nq = (K + 1) >> 1;  // pairs of k
  if (nq <= 0) return;
  double2 a[BS], b[BS], an[BS], bn[BS];
#pragma unroll
  for (int i = 0; i < BS; ++i) a[i] = ap[i * RS];
#pragma unroll
  for (int j = 0; j < BS; ++j) b[j] = bp[j * RS];
  for (int q = 0; q < nq; ++q) {
    const int qn = (q + 1 < nq) ? q + 1 : q;
#pragma unroll
    for (int i = 0; i < BS; ++i) an[i] = ap[i * RS + qn];
#pragma unroll
    for (int j = 0; j < BS; ++j) bn[j] = bp[j * RS + qn];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        acc[i][j] = fma(a[i].x, b[j].x, acc[i][j]);
        acc[i][j] = fma(a[i].y, b[j].y, acc[i][j]);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < BS; ++i) a[i] = an[i];
#pragma unroll
    for (int j = 0; j < BS; ++j) b[j] = bn[j];
  }
}

template <int BS, int LD>
__device__ __forceinline__ void mm_nt(double (&acc)[BS][BS], const double* A, const double* B, int K, int lr, int lc) {
  const double2* ap = reinterpret_cast<const double2*>(A + lr * BS * LD);
  const double2* bp = reinterpret_cast<const double2*>(B + lc * BS * LD);
  constexpr int RS = LD / 2;  // row stride in double2
  double2 a0[BS][2], b0[BS][2], a1[BS][2], b1[BS][2];
#define NT_LOAD(a, b, q)                                                     \
  do {                                                                       \
    _Pragma("unroll") for (int i = 0; i < BS; ++i) {                         \
      a[i][0] = ap[i * RS + 2 * (q)];                                        \
      a[i][1] = ap[i * RS + 2 * (q) + 1];                                    \
    }                                                                        \
    _Pragma("unroll") for (int j = 0; j < BS; ++j) {                         \
      b[j][0] = bp[j * RS + 2 * (q)];                                        \
      b[j][1] = bp[j * RS + 2 * (q) + 1];                                    \
    }                                                                        \
  } while (0)
#define NT_FMA(a, b)                                                         \
  do {                                                                       \
    _Pragma("unroll") for (int i = 0; i < BS; ++i)                           \
      _Pragma("unroll") for (int j = 0; j < BS; ++j) {                       \
        acc[i][j] = fma(a[i][0].x, b[j][0].x, acc[i][j]);                    \
        acc[i][j] = fma(a[i][0].y, b[j][0].y, acc[i][j]);                    \
        acc[i][j] = fma(a[i][1].x, b[j][1].x, acc[i][j]);                    \
        acc[i][j] = fma(a[i][1].y, b[j][1].y, acc[i][j]);                    \
      }                                                                      \
  } while (0)
  const int nq = (K + 3) >> 2;
  if (nq <= 0) return;
  NT_LOAD(a0, b0, 0);
  int q = 0;
  for (; q + 2 <= nq; q += 2) {
    NT_LOAD(a1, b1, q + 1);
    __builtin_amdgcn_sched_barrier(0);
    NT_FMA(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    const int qn = (q + 2 < nq) ? q + 2 : nq - 1;
    NT_LOAD(a0, b0, qn);
    __builtin_amdgcn_sched_barrier(0);
    NT_FMA(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (q < nq) NT_FMA(a0, b0);
#undef NT_LOAD
#undef NT_FMA
}

// DBG = true: the instance tools/kalman_phases.py launches (phase stamps of draw 0 in `dbg`); the product instance carries
// neither the stamps nor their registers.
// TAIL = true: the instance that hands the steady, constant-mask tail of the sample to kalman_tail_kernel (dsge_options.kalman_block)
// REC = true (round 5): the forward sweep of the gradient -- every step also writes its record (dsge_kalman_rec.hpp) for the reverse
// sweep of kalman_grad_kernel<BS, true>; the filter itself is the same code.
template <int BS, bool DBG = false, int SK = 8 * BS, bool TAIL = false, bool REC = false>
// (two wavefronts per SIMD up to the 24-wide tile, and on the 32-wide one for its 20-column instance -- 20 KB of LDS, eight draws
//  per CU; with the 256-register cap it spills 332 bytes per lane, and is still faster: observe_jumps 1.96 -> 2.04 M evals/s.  The
//  generic 32-wide instance (28 KB of LDS) measured SLOWER with the cap: 1.92 against 1.77 ms per 4096 draws at 25 states.)
__global__ __launch_bounds__(64, (BS == 1 ? 3 : ((BS <= KSEL_TWO_WAVES_MAX_BS || (BS == 4 && SK <= 20)) ? 2 : 1))) void kalman_nt_kernel(
    const double* __restrict__ T, const double* __restrict__ RQR, const double* __restrict__ P0,
    const double* __restrict__ Z, int z_batched, const double* __restrict__ dvec, int d_batched,
    const double* __restrict__ Hdiag, int h_batched, const double* __restrict__ y, int batch, int m_full, int p,
    int T_len, int s_cap, FilterConv cv, double missing_fill, double steady_tol, double* __restrict__ logp_out,
    int32_t* __restrict__ status, long long* __restrict__ dbg, int rerun_only, int32_t* __restrict__ steady_at,
    const int32_t* __restrict__ order, const double* __restrict__ Rsel, const double* __restrict__ qdiag, int q_batched,
    int k_shocks, const unsigned long long* __restrict__ colmask_in, double* __restrict__ tail_rec = nullptr,
    int32_t* __restrict__ tail_flag = nullptr, const int32_t* __restrict__ tail_from = nullptr,
    double* __restrict__ rec_store = nullptr) {
  using RC = KgRec<BS>;
  using SM = KntSmem<BS, SK>;
  constexpr int NP = SM::NP, LDK = SM::LDK, PS = SM::PS;
  constexpr bool NARROW = SK < NP;  // rows of Tc / W' / Pc shorter than the tile: stores beyond column SK - 1 are skipped
  // LEAN (single-buffered products, for a two-waves-per-SIMD build of the 32-wide tile) is OFF: measured in round 4 on the
  // observe_jumps leg (25 filtered variables): 2.49 -> 2.79 ms per step.  The launch ends with the never-steady draw's 200 full
  // steps on a lone wavefront, so the full step's latency sets it, not the occupancy; 304 B of spills and the shorter product
  // stages make that step slower.
  constexpr bool LEAN = false;
#define KNT_MM(acc, A_, B_, K_)                            \
  do {                                                     \
    if constexpr (LEAN)                                    \
      mm_nt_sb<BS, LDK>(acc, A_, B_, K_, lr, lc);          \
    else                                                   \
      mm_nt<BS, LDK>(acc, A_, B_, K_, lr, lc);             \
  } while (0)
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Tc = smem;                 // NP x LDK    transition, states-first ordering (columns >= s exactly zero)
  double* Wt = Tc + NP * LDK;        // WT doubles  W' : Wt[j][k] = (P+[S,S] Tc')[k][j]            (phase f)
  double* Vs = Wt;                   //   alias: NP x PS  -V = -(P Zm' + jitter K)                 (phases d, e)
  double* Pc = Wt + SM::WT;          // s_cap x LDK P+ restricted to the state block
  double* PZt = Pc + s_cap * LDK;    // NP x PS     (predicted P) Z', unmasked
  double* Ks = PZt + NP * PS;        // NP x PS     K = P Zm' Finv  (kept through the prediction: the steady loop reads it)
  double* av = Ks + NP * PS;         // NP          predicted state
  double* af = av + NP;              // NP          filtered state
  double* vv = af + NP;              // 8 innovation
  double* dd = vv + 8;               // 8 obs intercept
  double* hh = dd + 8;               // 8 diag(H)
  double* zv = hh + 8;               // 8 selector values
  int* perm = (int*)(zv + 8);        // NP: position -> original variable (states first)
  int* zpos = perm + NP;             // 8: position of the state each observation selects
  const int lane = threadIdx.x, lr = lane >> 3, lc = lane & 7;
  const int fo = lane >> 3, fq = lane & 7;  // owner of F[fo][fq] at the switch
  const double LN2PI = 1.8378770664093453, LN2 = 0.6931471805599453;

  if (rerun_only && rerun_pass_is_empty(status, batch, order)) return;
  // ONE draw per workgroup (the launcher's grid is the batch, also for a second pass): the "loop" below runs once.  With a real
  // grid-stride loop the compiler hoists everything loop-invariant -- down to the constants of log()'s polynomial in the
  // epilogue -- in front of it and then spills it around the time loop: 148 bytes of scratch per lane, 36 MB of HBM writes per
  // 4096-draw launch for a kernel whose output is 48 KB.
  for (int bi = blockIdx.x; bi < batch; bi = batch) {
    // (readfirstlane: the draw index is wave-uniform, and everything derived from it -- base addresses of a dozen
    // arrays -- then lives in scalar registers instead of being spilled around the time loop)
    const int draw = __builtin_amdgcn_readfirstlane(order ? order[bi] : bi);
    const int32_t st_in = __builtin_amdgcn_readfirstlane(status[draw]);
    // debug (dsge_debug_kalman_timeline): the instance WITHOUT phase stamps gets a [batch][8] int64 record through `dbg` --
    // {start, end (100 MHz wall clock), HW_ID, first steady step, start of the time loop, time of the first steady step} per draw
    long long tl_start = 0;
    if constexpr (!DBG) {
      if (dbg) tl_start = (long long)wall_clock64();
    }
    if (rerun_only) {
      if (st_in != DSGE_ST_INTERNAL_RERUN) continue;
    } else if (st_in != 0) {
      if (lane == 0) logp_out[draw] = -INFINITY;
      continue;
    }
    const size_t off = (size_t)draw * m_full * m_full;
    wave_sync();
    for (int idx = lane; idx < (int)SM::doubles(s_cap); idx += 64) smem[idx] = 0.0;

    // ---- exact state-space reduction to U = S u O, states first (as kalman_sel_kernel) ----------------------------
    const double* Zg = Z + (z_batched ? (size_t)draw * p * m_full : 0);
    bool is_state = false;
    // the solver may hand over which columns of T are non-zero (it wrote the others as exact zeros): no pass over T then
    const unsigned long long cm_in = colmask_in ? colmask_in[draw] : ~0ull;
    if (cm_in != ~0ull) {
      is_state = (lane < m_full) && ((cm_in >> lane) & 1ull);
    } else {
      const double* tcol = T + off + (lane < m_full ? lane : 0);
      for (int r0 = 0; r0 < m_full; r0 += 8) {
        double tv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) tv[u] = tcol[(size_t)(r0 + u < m_full ? r0 + u : m_full - 1) * m_full];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) is_state |= (tv[u] != 0.0);
      }
      is_state = is_state && (lane < m_full);
    }
    const unsigned long long colmask = __ballot(is_state);
    unsigned long long obsmask = 0ull, used = 0ull;
    bool ok = true;
    // (the p <= 8 rows of Z are requested together -- clamped, unconditional loads -- and kept for the second pass below: a
    //  load per trip, each behind its own ballot, was p round trips to L2 twice over: tools/kalman_timeline.py, prologue)
    double zrow[8];
    {
      const int zl_lane = lane < m_full ? lane : m_full - 1;
#pragma unroll
      for (int o = 0; o < 8; ++o) zrow[o] = Zg[(size_t)(o < p ? o : (p > 0 ? p - 1 : 0)) * m_full + zl_lane];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int o = 0; o < 8; ++o) zrow[o] = (lane < m_full && o < p) ? zrow[o] : 0.0;
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      if (o < p) {
        const unsigned long long b = __ballot(zrow[o] != 0.0);
        if (__popcll(b) != 1 || ((used & b) != 0ull)) ok = false;
        used |= b;
        obsmask |= b;
      }
    }
    const unsigned long long extra = obsmask & ~colmask;  // observed non-states
    const int s = __popcll(colmask);
    const int m = s + __popcll(extra);
    ok = ok && (s <= s_cap) && (m <= NP);
    int my_pos = -1;
    if (lane < m_full) {
      const unsigned long long below = (1ull << lane) - 1ull;
      if ((colmask >> lane) & 1ull)
        my_pos = __popcll(colmask & below);
      else if ((extra >> lane) & 1ull)
        my_pos = s + __popcll(extra & below);
      if (my_pos >= 0 && my_pos < NP) perm[my_pos] = lane;
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      if (o < p && zrow[o] != 0.0) {
        zpos[o] = (my_pos >= 0 && my_pos < NP) ? my_pos : 0;
        zv[o] = zrow[o];
      }
    }
    if (!ok) {
      if (lane == 0) status[draw] = DSGE_ST_INTERNAL_RERUN;
      continue;
    }
    if (rerun_only && lane == 0) status[draw] = 0;
    wave_sync();
    int pr[BS], pcx[BS], ocol[BS];
    double zcol[BS];
#pragma unroll
    for (int i = 0; i < BS; ++i) {
      const int r = lr * BS + i, c = lc * BS + i;
      pr[i] = (r < m) ? perm[r] : -1;
      pcx[i] = (c < m) ? perm[c] : -1;
      ocol[i] = -1;
      zcol[i] = 0.0;
      for (int o = 0; o < p; ++o)
        if (zpos[o] == c) {
          ocol[i] = o;
          zcol[i] = zv[o];
        }
    }
    double Qb[BS][BS], Pb[BS][BS], Tb0[BS][BS];
    if (Rsel) {
      // sym(R diag(q) R')[U,U] from the selection matrix itself (m_full x k_shocks, staged in the W' buffer: the launcher
      // checked that it fits): the 40 x 40 product launch and its 51 MB round trip through HBM are gone for the draws
      // this kernel takes.  Same expression and summation order as rqr_kernel, so the block is bit-identical to it.
      const int kp = (k_shocks + 1) & ~1;
      const double* Rg = Rsel + (size_t)draw * m_full * k_shocks;
      for (int idx = lane; idx < m_full * k_shocks; idx += 64) {
        const int i = idx / k_shocks, c = idx - i * k_shocks;
        Wt[i * kp + c] = Rg[idx];
      }
      const double* qd = qdiag + (q_batched ? (size_t)draw * k_shocks : 0);
      wave_sync();
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) {
          const bool in = pr[i] >= 0 && pcx[j] >= 0;
          const double2* ri = reinterpret_cast<const double2*>(Wt + (in ? pr[i] : 0) * kp);
          const double2* rj = reinterpret_cast<const double2*>(Wt + (in ? pcx[j] : 0) * kp);
          double a0 = 0.0, a1 = 0.0;
          for (int c2 = 0; 2 * c2 < kp; ++c2) {
            const double2 ti = ri[c2], tj = rj[c2];
            a0 = fma(ti.x * tj.x, qd[2 * c2], a0);
            a1 = fma(ti.y * tj.y, (2 * c2 + 1 < k_shocks) ? qd[2 * c2 + 1] : 0.0, a1);
          }
          Qb[i][j] = in ? a0 + a1 : 0.0;
        }
      wave_sync();
      for (int idx = lane; idx < m_full * kp; idx += 64) Wt[idx] = 0.0;  // (the filter relies on zero padding)
    }
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        const bool in = pr[i] >= 0 && pcx[j] >= 0;
        const size_t g = in ? (size_t)pr[i] * m_full + pcx[j] : 0;
        const double tv = in ? T[off + g] : 0.0;
        if (!Rsel) Qb[i][j] = in ? RQR[off + g] : 0.0;
        Pb[i][j] = (in && P0) ? P0[off + g] : 0.0;
        Tb0[i][j] = tv;
        if (!NARROW || lc * BS + j < SK) Tc[(lr * BS + i) * LDK + lc * BS + j] = tv;  // (columns >= s of T are exact zeros)
      }
    const bool in_state_block = (lr * BS < s) && (lc * BS < s);
    const bool w_rows = lr * BS < s;  // this lane's rows of W = P+[S,S] Tc' exist
    if (!P0) {
      // ---- P0 = dlyap(T, RQR)[U,U] by doubling on the reduced model (statespace.py:814-815) ---------------------
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) Pb[i][j] = Qb[i][j];
      bool lyap_ok = false;
      for (int itl = 0; itl < 64; ++itl) {
        wave_sync();
        if (in_state_block) blk_store_lds<BS>(Pb, Pc, LDK, lr, lc);
        wave_sync();
        if (w_rows) {
          double Wb[BS][BS];
          blk_zero<BS>(Wb);
          KNT_MM(Wb, Pc, Tc, s);  // P[S,S] A_k'
#pragma unroll
          for (int i = 0; i < BS; ++i)
#pragma unroll
            for (int j = 0; j < BS; ++j) Wt[(lc * BS + j) * LDK + lr * BS + i] = Wb[i][j];
        }
        double Ab[BS][BS];
        blk_zero<BS>(Ab);
        mm_acc_p<BS, false, LDK, LDK>(Ab, Tc, Tc, s, lr, lc);  // A_k[:,S] A_k[S,:]
        wave_sync();
        double Xb[BS][BS];
        blk_zero<BS>(Xb);
        KNT_MM(Xb, Tc, Wt, s);
        wave_sync();
        knt_store_cols<BS, NARROW ? SK : 0>(Ab, Tc, LDK, lr, lc);
        const int src = (lc << 3) | lr;
        double dmax = 0.0, pmax = 0.0;
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) {
            const double xt = __shfl(Xb[j][i], src, 64);
            const double dlt = 0.5 * (Xb[i][j] + xt);
            Pb[i][j] += dlt;
            dmax = nanmax(dmax, fabs(dlt));
            pmax = nanmax(pmax, fabs(Pb[i][j]));
          }
        dmax = wave_nanmax(dmax);
        pmax = wave_nanmax(pmax);
        if (!(dmax == dmax) || !(pmax < 1e300)) break;
        if (dmax <= 1e-17 * pmax) {
          lyap_ok = true;
          break;
        }
      }
      wave_sync();
      knt_store_cols<BS, NARROW ? SK : 0>(Tb0, Tc, LDK, lr, lc);
      for (int idx = lane; idx < SM::WT; idx += 64) Wt[idx] = 0.0;  // (the filter relies on zero padding)
      if (!lyap_ok) {
        if (lane == 0) {
          status[draw] |= DSGE_ST_LYAP_FAIL;
          logp_out[draw] = -INFINITY;
        }
        continue;
      }
    }
    if (lane < 8) {
      dd[lane] = (dvec && lane < p) ? dvec[(d_batched ? (size_t)draw * p : 0) + lane] : 0.0;
      hh[lane] = (Hdiag && lane < p) ? Hdiag[(h_batched ? (size_t)draw * p : 0) + lane] : 0.0;
    }
    // P Z' panel of the predicted covariance (unmasked): written by the lanes that own an observed column
#define STORE_PZT()                                                                              \
  do {                                                                                           \
    _Pragma("unroll") for (int j = 0; j < BS; ++j) if (ocol[j] >= 0) {                           \
      _Pragma("unroll") for (int i = 0; i < BS; ++i) PZt[(lr * BS + i) * PS + ocol[j]] = zcol[j] * Pb[i][j]; \
    }                                                                                            \
  } while (0)
    STORE_PZT();
    const int v_zpos = (lane < p) ? zpos[lane] : 0;
    const double v_zv = (lane < p) ? zv[lane] : 0.0, v_dd = (lane < 8) ? dd[lane & 7] : 0.0;
    wave_sync();

    const int r8 = lane & 7, g8 = lane >> 3;  // the update runs on 8 replicas of an 8-lane group: lane -> row r8 of F
    const int r_zpos = (r8 < p) ? zpos[r8] : 0;
    const double r_zv = (r8 < p) ? zv[r8] : 0.0, r_dd = dd[r8], r_hh = hh[r8];
    double jit_d[BS];  // jitter on the diagonal of P+ (rows < m of the diagonal lanes)
#pragma unroll
    for (int i = 0; i < BS; ++i) jit_d[i] = (lr == lc && lr * BS + i < m) ? cv.jit_P : 0.0;
    const bool fold_a = m < NP;  // a spare padding column: the mean prediction rides along in the X product

    double quad_sum = 0.0, quad_comp = 0.0;  // Kahan sum of v' Finv v over observed steps
    double ld_mant = 1.0;                    // prod of pivots = mant * 2^exp
    int ld_exp = 0;
    int n_ll_steps = 0, n_obs_entries = 0;  // steps with >= 1 observed entry, and their observed entries (FilterConv::ll_terms)
    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long tk_start = DBG ? clock64() : 0;
    long long tl_loop = 0, tl_steady = 0;  // debug timeline: start of the time loop, first steady step
    if constexpr (!DBG) {
      if (dbg) tl_loop = (long long)wall_clock64();
    }
    int steady_step = -1;
    bool handed_off = false;
    double* const rec_d = REC ? rec_store + (size_t)draw * RC::per_draw(T_len) : nullptr;  // this draw's record
    int seg_src = -1;                                                                       // source step of the running segment
    if constexpr (REC) {  // P_0 (register blocks -> row-major NP x NP behind the last step)
      double* p0s = rec_d + (size_t)T_len * RC::STEP;
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) p0s[(lr * BS + i) * NP + lc * BS + j] = Pb[i][j];
      if (lane == 0) rec_d[RC::tail_state_off(T_len) + RC::TS_LAYOUT] = 0.0;  // P+ of the full steps: lane-major register blocks
    }
    // y_t is fetched one step ahead by an UNCONDITIONAL, branch-free load (clamped indices; lanes with r8 >= p and the value
    // past the last step are never used -- every use is guarded by `obs`): under a condition the compiler sank the load to
    // the top of the step that needs it and waited for it there (s_waitcnt vmcnt(0) right behind the load, every step)
    const int r8c = (r8 < p) ? r8 : (p > 0 ? p - 1 : 0);
    double yt_next = (T_len > 0) ? y[r8c] : 0.0;
    for (int t = 0; t < T_len; ++t) {
      long long tk0 = DBG ? clock64() : 0;
      // ---- (a) missing-data mask; every LDS operand of the update is requested up front ------------------------------
      const double yt = yt_next;
      yt_next = y[(size_t)((t + 1 < T_len) ? t + 1 : t) * p + r8c];
      const bool obs = (r8 < p) && (yt == yt) && (yt != missing_fill);
      const unsigned long long omask = __ballot(obs) & 0xffull;
      const int n_obs = __popcll(omask);
      double* const sg = REC ? rec_d + (size_t)t * RC::STEP : nullptr;  // the record of this (full) step
      if constexpr (REC) {
        if (lane < NP) sg[RC::OFF_A + lane] = av[lane];
        if (lane == 0) {
          sg[RC::OFF_SRC] = (double)t;
          sg[RC::OFF_PREV] = (double)seg_src;
        }
        seg_src = t;
      }
      double2 fr2[4], pz2[BS][4];
      double pzo[BS], avi[BS];
#pragma unroll
      for (int q2 = 0; q2 < 4; ++q2) fr2[q2] = *reinterpret_cast<const double2*>(&PZt[r_zpos * PS + 2 * q2]);
      const double a_sel = av[r_zpos];
#pragma unroll
      for (int ps = 0; ps < BS; ++ps) {
        const int i = g8 + 8 * ps;
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) pz2[ps][q2] = *reinterpret_cast<const double2*>(&PZt[i * PS + 2 * q2]);
        pzo[ps] = PZt[i * PS + r8];
        avi[ps] = av[i];
      }
      __builtin_amdgcn_sched_barrier(0);
      bool steady = false;
      double pm = 0.0;  // max |P_{t|t-1}| (wave-uniform), reduced while the loads are in flight.  P is positive
                        // semi-definite: its largest entry is on the diagonal, which the lanes lr == lc hold
      if (steady_tol > 0.0) {
        double pscale = 0.0;
#pragma unroll
        for (int i = 0; i < BS; ++i) pscale = fmax(pscale, fabs(Pb[i][i]));
        if (lr != lc) pscale = 0.0;
        // exact max of non-negative doubles: high words first, then the low words of the lanes that attain it
        const unsigned hi = (unsigned)__double2hiint(pscale), lo = (unsigned)__double2loint(pscale);
        const unsigned mhi = wave_max_u32(hi);
        const unsigned mlo = wave_max_u32(hi == mhi ? lo : 0u);
        pm = __hiloint2double((int)mhi, (int)mlo);
      }
      // ---- (b) innovation v[r8] and row r8 of F = Zm P Zm' + Hm + jitter I, replicated over the eight 8-lane groups
      //      (identity rows / columns for missing observations and for r8 >= p) ---------------------------------------
      const double c_r = obs ? r_zv : 0.0;
      const double v_r = (obs ? yt : 0.0) - (((obs || !cv.mask_d) ? r_dd : 0.0) + c_r * a_sel);
      const double dg = (r8 < p) ? ((obs ? r_hh : 0.0) + cv.jit_F) : 1.0;
      if (lane < 8) vv[lane] = v_r;  // broadcast of v for the quadratic form: through LDS, behind the elimination
      asm volatile("" ::: "memory");  // (the double2 reads below must not be hoisted above this store)
      double fr[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const double tq = (q & 1) ? fr2[q >> 1].y : fr2[q >> 1].x;
        const double wq = ((omask >> q) & 1ull) ? 1.0 : 0.0;  // wave-uniform
        const double f = (c_r * tq) * wq;
        fr[q] = (q == r8) ? f + dg : f;
      }
      if constexpr (REC) {
        if (lane < 8) {
#pragma unroll
          for (int q = 0; q < 8; ++q) sg[RC::OFF_F + lane * 8 + q] = fr[q];
        }
      }
      // ---- (c) Finv by Gauss-Jordan, one row per lane (SPD: no pivoting); the pivot row arrives through SGPRs.  The
      //      rows are left unscaled during the elimination (row j keeps its pivot: one fma per entry and lane, no
      //      per-entry select); Finv[r][:] = inv_own * fr[:] afterwards -------------------------------------------------
      double step_mant = 1.0, inv_own = 1.0;
      int step_exp = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < p) {
          double rowj[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) rowj[q] = readlane_f64(fr[q], j);
          const double inv = fast_rcp(rowj[j]);
          const bool is_j = (r8 == j);
          const double ci = is_j ? 0.0 : fr[j] * inv;  // row j itself is left alone
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (q != j) fr[q] = fma(-ci, rowj[q], fr[q]);
          fr[j] = is_j ? 1.0 : -ci;  // column j of the identity block, stored in place
          inv_own = is_j ? inv : inv_own;
          int e;
          step_mant *= frexp(rowj[j], &e);
          step_exp += e;
        }
      }
      if constexpr (REC) {
        if (lane < 8) {
#pragma unroll
          for (int q = 0; q < 8; ++q) sg[RC::OFF_FI + lane * 8 + q] = fr[q] * inv_own;
        }
      }
      {
        const double2* vv2 = reinterpret_cast<const double2*>(vv);
        double w0 = 0.0, w1 = 0.0;
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) {
          const double2 vq = vv2[q2];
          w0 = fma(fr[2 * q2], vq.x, w0);
          w1 = fma(fr[2 * q2 + 1], vq.y, w1);
        }
        if (n_obs > 0) {
          // lane r < 8 keeps its own share v_r (Finv v)_r of the quadratic forms (Kahan); the lanes are added up once
          const double yk = ((lane < 8) ? (v_r * inv_own) * (w0 + w1) : 0.0) - quad_comp;
          const double tk = quad_sum + yk;
          quad_comp = (tk - quad_sum) - yk;
          quad_sum = tk;
          int e;
          ld_mant = frexp(ld_mant * step_mant, &e);
          ld_exp += e + step_exp;
          ++n_ll_steps;
          n_obs_entries += n_obs;
        }
      }
      if constexpr (DBG) {
        const long long tk1 = clock64();
        ph[0] += tk1 - tk0;
        tk0 = tk1;
      }
      // ---- (d) K = (P Zm') Finv (Finv symmetric: lane (i, o) contracts row i of P Zm' with ITS row o of Finv),
      //          V = P Zm' + jitter K, a+ = a + K v ---------------------------------------------------------------
#pragma unroll
      for (int ps = 0; ps < BS; ++ps) {
        const int i = g8 + 8 * ps;
        // F is block diagonal (observed block, identity for the rest) and so is Finv, exactly: masking the result for a
        // missing observation o = r8 is all the masking the gain needs
        double k0 = 0.0, k1 = 0.0;
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) {
          k0 = fma(pz2[ps][q2].x, fr[2 * q2], k0);
          k1 = fma(pz2[ps][q2].y, fr[2 * q2 + 1], k1);
        }
        const double kk = obs ? (k0 + k1) * inv_own : 0.0;
        Ks[i * PS + r8] = kk;
        Vs[i * PS + r8] = -fma(cv.jit_V, kk, obs ? pzo[ps] : 0.0);  // stored negated: the downdate is a plain fma chain
        double part = kk * v_r;
        part += dpp_move_f64<0xB1, 0xf>(part);   // quad_perm [1,0,3,2]
        part += dpp_move_f64<0x4E, 0xf>(part);   // quad_perm [2,3,0,1]
        part += dpp_move_f64<0x141, 0xf>(part);  // row_half_mirror: the other quad of the 8-lane group
        if (r8 == 0) af[i] = avi[ps] + part;
      }
      wave_sync();  // #2
      if constexpr (REC) {
        for (int idx = lane; idx < NP * 8; idx += 64) sg[RC::OFF_K + idx] = Ks[(idx >> 3) * PS + (idx & 7)];
      }
      if constexpr (DBG) {
        const long long tk1 = clock64();
        ph[1] += tk1 - tk0;
        tk0 = tk1;
      }
      // ---- (e) P+ = P - K V' + jitter I (register blocks); state block -> LDS -------------
      {
        double2 ka[BS], vb[BS], kan[BS], vbn[BS];
#pragma unroll
        for (int i = 0; i < BS; ++i) ka[i] = *reinterpret_cast<const double2*>(&Ks[(lr * BS + i) * PS]);
#pragma unroll
        for (int j = 0; j < BS; ++j) vb[j] = *reinterpret_cast<const double2*>(&Vs[(lc * BS + j) * PS]);
#pragma unroll
        for (int o2 = 0; o2 < 4; ++o2) {
          if (o2 < 3) {
#pragma unroll
            for (int i = 0; i < BS; ++i) kan[i] = *reinterpret_cast<const double2*>(&Ks[(lr * BS + i) * PS + 2 * (o2 + 1)]);
#pragma unroll
            for (int j = 0; j < BS; ++j) vbn[j] = *reinterpret_cast<const double2*>(&Vs[(lc * BS + j) * PS + 2 * (o2 + 1)]);
          }
#pragma unroll
          for (int i = 0; i < BS; ++i)
#pragma unroll
            for (int j = 0; j < BS; ++j) {
              Pb[i][j] = fma(ka[i].x, vb[j].x, Pb[i][j]);
              Pb[i][j] = fma(ka[i].y, vb[j].y, Pb[i][j]);
            }
#pragma unroll
          for (int i = 0; i < BS; ++i) ka[i] = kan[i];
#pragma unroll
          for (int j = 0; j < BS; ++j) vb[j] = vbn[j];
        }
      }
#pragma unroll
      for (int i = 0; i < BS; ++i) Pb[i][i] += jit_d[i];
      if constexpr (REC) {  // P+ straight from the register blocks, lane-major (BS^2 coalesced stores)
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) sg[(size_t)(i * BS + j) * 64 + lane] = Pb[i][j];
      }
      if (steady_tol > 0.0) {
        double dmax = 0.0;
        if (in_state_block) {
#pragma unroll
          for (int i = 0; i < BS; ++i)
#pragma unroll
            for (int j = 0; j < BS; ++j) dmax = fmax(dmax, fabs(Pb[i][j] - Pc[(lr * BS + i) * LDK + lc * BS + j]));
        }
        // max|dP+[S,S]| <= tol * max|P|  <=>  no lane violates it (a NaN in P ends in a non-finite logp either way)
        steady = (t > 0) && (__ballot(!(dmax <= steady_tol * pm)) == 0ull);
      }
      if (in_state_block) blk_store_lds<BS>(Pb, Pc, LDK, lr, lc);
      wave_sync();  // #3
      if constexpr (DBG) {
        const long long tk1 = clock64();
        ph[2] += tk1 - tk0;
        tk0 = tk1;
      }
      // ---- (f) predict: W = P+[S,S] Tc' (stored transposed);  X = Tc W;  P = sym(X) + RQR.  The mean a = Tc a+[:s] is
      //      the last column of X when the tile has a spare padding column (a+ rides along as row NP-1 of W'): no extra flops ------
      double af_l = 0.0;
      if (fold_a) {
        af_l = af[lane < NP ? lane : 0];
      } else if (lane < m) {
        const double2* trow2 = reinterpret_cast<const double2*>(Tc + lane * LDK);
        const double2* af2 = reinterpret_cast<const double2*>(af);
        double s0 = 0.0, s1 = 0.0;
        for (int kk = 0; 2 * kk < s; ++kk) {  // (columns >= s of Tc are zero: an odd s reads one harmless extra term)
          const double2 tv2 = trow2[kk], fv = af2[kk];
          s0 = fma(tv2.x, fv.x, s0);
          s1 = fma(tv2.y, fv.y, s1);
        }
        av[lane] = s0 + s1;
      }
      if (w_rows) {
        double Wb[BS][BS];
        blk_zero<BS>(Wb);
        KNT_MM(Wb, Pc, Tc, s);
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) Wt[(lc * BS + j) * LDK + lr * BS + i] = Wb[i][j];
      }
      if (fold_a && lane < SK) Wt[(NP - 1) * LDK + lane] = af_l;  // after the W stores (same wave: LDS keeps program order);
                                                                  // the X product contracts over k < s <= SK only
      wave_sync();  // #4
      if constexpr (DBG) {
        const long long tk1 = clock64();
        ph[3] += tk1 - tk0;
        tk0 = tk1;
      }
      {
        double Xb[BS][BS];
        blk_zero<BS>(Xb);
        KNT_MM(Xb, Tc, Wt, s);
        if (fold_a && lc == 7) {  // column NP-1 of X = Tc a+: hand it to the LDS copy of the mean, restore the padding
#pragma unroll
          for (int i = 0; i < BS; ++i) {
            av[lr * BS + i] = Xb[i][BS - 1];
            Xb[i][BS - 1] = 0.0;
          }
        }
        const int src = (lc << 3) | lr;  // lane holding the transposed block
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) {
            const double xt = __shfl(Xb[j][i], src, 64);
            Pb[i][j] = 0.5 * (Xb[i][j] + xt) + Qb[i][j];
          }
      }
      STORE_PZT();
      wave_sync();  // #5
      if constexpr (DBG) {
        const long long tk1 = clock64();
        ph[4] += tk1 - tk0;
        tk0 = tk1;
      }
      if (!steady) continue;
      // ==== steady-state steps: mean recursion only, while the missing-data mask stays the same (register-only) ====
      if constexpr (!DBG) {
        if (dbg && steady_step < 0) tl_steady = (long long)wall_clock64();
      }
      if (steady_step < 0) steady_step = t + 1;
      if constexpr (TAIL) {
        // ---- hand-off: the covariance is frozen AND the missing-data mask no longer changes until the end of the sample
        //      (t >= *tail_from, kalman_mask_scan_kernel): the rest is a linear recursion in the mean, which kalman_tail_kernel runs in
        //      blocks of steps with registers of its own.  This kernel writes the record and leaves the time loop.
        if (tail_rec && t + 2 < T_len && t >= *tail_from) {
          const double quad_now = wave_sum_dpp(quad_sum - quad_comp);
          double* rec = tail_rec + (size_t)draw * KT_REC;
          for (int idx = lane; idx < NP * NP; idx += 64) {
            const int i = idx / NP, c = idx - i * NP;
            rec[KT_T + i * 32 + c] = (c < s && c < SK) ? Tc[i * LDK + c] : 0.0;
          }
          for (int idx = lane; idx < NP * 8; idx += 64) rec[KT_K + idx] = Ks[(idx >> 3) * PS + (idx & 7)];
          if (lane < 8) {
#pragma unroll
            for (int q = 0; q < 8; ++q) rec[KT_FI + lane * 8 + q] = fr[q] * inv_own;
          }
          if (lane < NP) rec[KT_A + lane] = (lane < m) ? av[lane] : 0.0;
          if (lane < 8) {
            rec[KT_ZV + lane] = (lane < p) ? zv[lane] : 0.0;
            rec[KT_DD + lane] = (lane < p && (((omask >> lane) & 1ull) || !cv.mask_d)) ? dd[lane] : 0.0;
            rec[KT_ZP + lane] = (lane < p) ? (double)zpos[lane] : 0.0;
          }
          if (lane == 0) {
            double* sc = rec + KT_SC;
            sc[0] = (double)m;
            sc[1] = (double)s;
            sc[2] = (double)t;
            sc[3] = (double)omask;
            sc[4] = (double)n_obs;
            sc[5] = step_mant;
            sc[6] = (double)step_exp;
            sc[7] = quad_now;
            sc[8] = 0.0;
            sc[9] = ld_mant;
            sc[10] = (double)ld_exp;
            sc[11] = (double)n_ll_steps;
            sc[12] = (double)steady_step;
            sc[13] = (double)n_obs_entries;
            tail_flag[draw] = 1;
          }
          handed_off = true;
          break;
        }
      }
      {
        double trow[SK], finv_row[8], kr_ss[8];
        double av_reg = (lane < m) ? av[lane] : 0.0;
#pragma unroll
        for (int kk = 0; kk < SK; ++kk) trow[kk] = (lane < NP) ? Tc[lane * LDK + kk] : 0.0;  // columns >= s are zero
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          finv_row[q] = (lane < 8) ? fr[q] * inv_own : 0.0;  // lane r < 8 still holds row r of Finv (unscaled)
          kr_ss[q] = (lane < m) ? Ks[lane * PS + q] : 0.0;
        }
        // (Round 5, measured and NOT used -- profiles/r5/steady_loop_ab.txt: (i) y_t staged through LDS in chunks of NP / 8 steps, so that
        //  no step waits for the load of its successor's y -- the compiler copies the register loaded one step ahead into the
        //  loop-carried one at the back edge behind an s_waitcnt vmcnt(0) --, and (ii) the two broadcasts of a step, v and a+, as
        //  one ds_write + SK / 2 + 4 uniform ds_read_b128 instead of 2 (SK + 8) v_readlane_b32: 80 VALU instructions per step
        //  instead of 156, and 0.549 us per steady step instead of 0.495 at two wavefronts per SIMD.  The step is a latency chain --
        //  a_sel -> v -> K v -> T a+ --, not an issue-bound loop: the LDS round trips lengthen it by more than the fewer
        //  instructions shorten it.)
        while (t + 1 < T_len) {
          const double yt_s = yt_next;
          const bool obs_s = (lane < p) && (yt_s == yt_s) && (yt_s != missing_fill);
          if (__ballot(obs_s) != omask) break;
          ++t;
          yt_next = y[(size_t)((t + 1 < T_len) ? t + 1 : t) * p + r8c];
          if constexpr (REC) {
            double* sgs = rec_d + (size_t)t * RC::STEP;
            if (lane < NP) sgs[RC::OFF_A + lane] = av_reg;
            if (lane == 0) sgs[RC::OFF_SRC] = (double)seg_src;
          }
          const double av_sel = __shfl(av_reg, v_zpos, 64);
          double v_s = 0.0;
          if (lane < p) v_s = (obs_s ? yt_s : 0.0) - (((obs_s || !cv.mask_d) ? v_dd : 0.0) + (obs_s ? 1.0 : 0.0) * (v_zv * av_sel));
          double vsc[8];
#pragma unroll
          for (int o = 0; o < 8; ++o) vsc[o] = readlane_f64(v_s, o);
          double w0 = 0.0, w1 = 0.0, a0 = av_reg, a1 = 0.0;
#pragma unroll
          for (int o = 0; o < 8; o += 2) {
            w0 = fma(finv_row[o], vsc[o], w0);
            w1 = fma(finv_row[o + 1], vsc[o + 1], w1);
            a0 = fma(kr_ss[o], vsc[o], a0);
            a1 = fma(kr_ss[o + 1], vsc[o + 1], a1);
          }
          if (n_obs > 0) {
            const double yk = v_s * (w0 + w1) - quad_comp;  // lanes >= 8 hold finv_row = 0
            const double tk = quad_sum + yk;
            quad_comp = (tk - quad_sum) - yk;
            quad_sum = tk;
            int e;
            ld_mant = frexp(ld_mant * step_mant, &e);
            ld_exp += e + step_exp;
            ++n_ll_steps;
            n_obs_entries += n_obs;
          }
          const double afi = a0 + a1;
          double s0 = 0.0, s1 = 0.0;
#pragma unroll
          for (int kk = 0; kk < SK; kk += 2) {
            s0 = fma(trow[kk], readlane_f64(afi, kk), s0);
            s1 = fma(trow[kk + 1], readlane_f64(afi, kk + 1), s1);
          }
          av_reg = (lane < m) ? s0 + s1 : 0.0;
          if constexpr (DBG) ++ph[6];
        }
        if (lane < m) av[lane] = av_reg;  // hand the predicted state back to the LDS copy
      }
      wave_sync();
      if constexpr (DBG) ph[5] += clock64() - tk0;
    }
#undef STORE_PZT
#undef KNT_MM
    if (DBG && dbg && draw == 0 && lane == 0) {
      ph[7] = clock64() - tk_start;
      for (int k = 0; k < 8; ++k) dbg[k] = ph[k];
    }
    const double quad_total = wave_sum_dpp(quad_sum - quad_comp);  // lanes 0..7 hold the shares, the others zero
    if (lane == 0 && !handed_off) {
      const double logdet = log(ld_mant) + (double)ld_exp * LN2;
      const double ll = -0.5 * (cv.ll_terms(n_ll_steps, n_obs_entries, p) * LN2PI + logdet + quad_total);
      logp_out[draw] = ll;
      if (steady_at) steady_at[draw] = steady_step;
      if constexpr (!DBG) {
        if (dbg) {
          dbg[8 * (size_t)draw] = tl_start;
          dbg[8 * (size_t)draw + 1] = (long long)wall_clock64();
          dbg[8 * (size_t)draw + 2] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_REG_HW_ID
          dbg[8 * (size_t)draw + 3] = steady_step;
          dbg[8 * (size_t)draw + 4] = tl_loop;
          dbg[8 * (size_t)draw + 5] = tl_steady;
        }
      }
      if (!((ll == ll) && (fabs(ll) < 1.797e308))) status[draw] |= DSGE_ST_FILTER_NONFINITE;
    }
  }
}

}  // namespace dsge

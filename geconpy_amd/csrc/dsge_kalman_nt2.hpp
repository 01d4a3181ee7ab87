// Kalman filter log-likelihood, selector design matrix, TWO wavefronts per draw: "kalman_nt2_kernel" (round 5).
//
// Why.  A launch of kalman_nt_kernel ends with its slowest draw: a model whose covariance recursion never reaches its fixed point
// within the sample runs T_len full steps, and a full step of a lone wavefront is one dependent chain of 8.9 k cycles
// (F + Gauss-Jordan 3.1 k -> gain 0.6 k -> P+ downdate 1.3 k -> W product 1.8 k -> X product 2.4 k): the other 4095 draws of the
// bench need 0.45 ms of the chip, that draw alone 0.75 ms.  The chain is serial only because the products are taken of the
// FILTERED covariance.  Written for the PREDICTED one,
//     P_{t+1} = Tc P+ Tc' + Q,   P+ = P - K V' + jit_P I,   V = P Zm' + jit_V K
//             = Tc P[S,S] Tc'  -  (Tc K)(Tc V)'  +  (jit_P Tc Tc' + Q),
// the two big products need nothing of the measurement update and run NEXT TO it:
//     wavefront B (products): G = Tc (P Z')[S] (thin; flag to A), W0 = P[S,S] Tc', X0 = Tc W0, Xs = sym(X0) + J
//     wavefront A (update):   mask, F, Gauss-Jordan, K, a+, then TK = Tc K = G Finv, TV = Tc V = G + jit_V TK (rows of G, as the gain)
//     barrier;  B: P_{t+1} = Xs - TK TV', steady test, P Z' panel;  A: a_{t+1} = Tc a+
//     barrier.
// Same recursion, same conventions (FilterConv), same steady-state switch; the rank-p correction replaces the P+ downdate
// (same flops), the thin products TK, TV are the price (2 m s p flops).  Full step ~6 k cycles instead of 8.9 k.
//
// Used for the HEAD of the dispatch order only (dsge_options.kalman_head_draws: the draws most likely to be slow, launched on a
// library-owned second stream next to the one-wavefront kernel that takes the bulk): at 256 registers per wavefront a CU holds
// four two-wavefront draws instead of eight one-wavefront ones, so the bulk is better off where it is.  The tests run whole
// batches through it (kalman_head_draws = -1) against the oracle and against kalman_nt_kernel.
#pragma once
#include "dsge_kalman_nt.hpp"

namespace dsge {

template <int BS, int SK = 8 * BS>
struct Knt2Smem {
  static constexpr int NP = Tile<BS>::NP, LDK = SK + 2, PS = 10;
  static_assert(SK % 4 == 0 && SK <= NP && SK >= 8, "SK: a multiple of four in 8..NP");
  // doubles: Tc, Wt NP*LDK each; Pc s_cap*LDK; PZt, Ks, Vs, TKs, TVs NP*PS each; av, af NP; vv, dd, hh, zv 8 each; flags 8;
  // ints perm NP, zpos 8
  __host__ __device__ static constexpr size_t doubles(int s_cap) {
    return 2 * (size_t)NP * LDK + (size_t)s_cap * LDK + 8 * (size_t)LDK + 5 * (size_t)NP * PS + 2 * NP + 40 + NP / 2 + 4;
  }
  static size_t bytes(int s_cap) { return sizeof(double) * doubles(s_cap); }
};

// fence among the lanes of ONE wavefront: LDS operations of a wavefront are executed in program order, so all the fence has to do is
// keep the compiler from moving them (no s_barrier: the other wavefront of the workgroup is somewhere else entirely)
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// DBG: the instance tools/kalman_phases.py launches (per-phase shader cycles of draw 0 of both wavefronts in `dbg`, 16 values)
template <int BS, int SK = 8 * BS, bool DBG = false>
__global__ __launch_bounds__(128, 2) void kalman_nt2_kernel(
    const double* __restrict__ T, const double* __restrict__ RQR, const double* __restrict__ P0,
    const double* __restrict__ Z, int z_batched, const double* __restrict__ dvec, int d_batched,
    const double* __restrict__ Hdiag, int h_batched, const double* __restrict__ y, int batch, int m_full, int p,
    int T_len, int s_cap, FilterConv cv, double missing_fill, double steady_tol, double* __restrict__ logp_out,
    int32_t* __restrict__ status, int rerun_only, int32_t* __restrict__ steady_at, const int32_t* __restrict__ order,
    const double* __restrict__ Rsel, const double* __restrict__ qdiag, int q_batched, int k_shocks,
    const unsigned long long* __restrict__ colmask_in, long long* __restrict__ dbg) {
  using SM = Knt2Smem<BS, SK>;
  constexpr int NP = SM::NP, LDK = SM::LDK, PS = SM::PS;
  constexpr bool NARROW = SK < NP;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Tc = smem;                 // NP x LDK    transition, states-first ordering (columns >= s exactly zero)
  double* Wt = Tc + NP * LDK;        // NP x LDK    W' : Wt[j][k] = (P[S,S] Tc')[k][j]; prologue: staging of R
  double* Pc = Wt + NP * LDK;        // s_cap x LDK predicted P restricted to the state block
  double* ZPt = Pc + s_cap * LDK;    // 8 x LDK     Z P[:, S] (the panel below, transposed, state columns): B's operand for G
  double* PZt = ZPt + 8 * LDK;       // NP x PS     (predicted P) Z', unmasked                       B -> A
  double* Ks = PZt + NP * PS;        // NP x PS     K = P Zm' Finv                                    A
  double* Gs = Ks + NP * PS;         // NP x PS     G = Tc (P Z')[S], unmasked                        B -> A (flag)
  double* TKs = Gs + NP * PS;        // NP x PS     Tc K                                              A -> B
  double* TVs = TKs + NP * PS;       // NP x PS     -(Tc V)  (negated: the correction is a plain fma chain)  A -> B
  double* av = TVs + NP * PS;        // NP          predicted state                                   A
  double* af = av + NP;              // NP          filtered state                                    A
  double* vv = af + NP;              // 8 innovation
  double* dd = vv + 8;               // 8 obs intercept
  double* hh = dd + 8;               // 8 diag(H)
  double* zv = hh + 8;               // 8 selector values
  int* flags = (int*)(zv + 8);       // [0] steady (B -> A), [1] next time step (A -> B), [2] lyap_ok (B -> A), [3] G of step t is ready: t + 1 (B -> A)
  int* perm = flags + 16;            // NP: position -> original variable (states first)
  int* zpos = perm + NP;             // 8: position of the state each observation selects
  const int tid = threadIdx.x;
  const int lane = tid & 63, lr = lane >> 3, lc = lane & 7;
  const bool isA = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;  // wave-uniform role
  const double LN2PI = 1.8378770664093453, LN2 = 0.6931471805599453;

  const int bi = blockIdx.x;
  if (bi >= batch) return;
  const int draw = __builtin_amdgcn_readfirstlane(order ? order[bi] : bi);
  const int32_t st_in = __builtin_amdgcn_readfirstlane(status[draw]);
  if (rerun_only) {
    if (st_in != DSGE_ST_INTERNAL_RERUN) return;
  } else if (st_in != 0) {
    if (tid == 0) logp_out[draw] = -INFINITY;
    return;
  }
  const size_t off = (size_t)draw * m_full * m_full;
  for (int idx = tid; idx < (int)SM::doubles(s_cap); idx += 128) smem[idx] = 0.0;
  __syncthreads();

  // ---- exact state-space reduction to U = S u O, states first (as kalman_nt_kernel; both wavefronts compute the same tables
  //      and write the same values) ---------------------------------------------------------------------------------------
  const double* Zg = Z + (z_batched ? (size_t)draw * p * m_full : 0);
  bool is_state = false;
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
  for (int o = 0; o < p; ++o) {
    const double zl = (lane < m_full) ? Zg[(size_t)o * m_full + lane] : 0.0;
    const unsigned long long b = __ballot(zl != 0.0);
    if (__popcll(b) != 1 || ((used & b) != 0ull)) ok = false;
    used |= b;
    obsmask |= b;
  }
  const unsigned long long extra = obsmask & ~colmask;
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
  for (int o = 0; o < p; ++o) {
    const double zl = (lane < m_full) ? Zg[(size_t)o * m_full + lane] : 0.0;
    if (zl != 0.0) {
      zpos[o] = (my_pos >= 0 && my_pos < NP) ? my_pos : 0;
      zv[o] = zl;
    }
  }
  if (!ok) {  // (wave-uniform, the same in both wavefronts)
    if (tid == 0) status[draw] = DSGE_ST_INTERNAL_RERUN;
    return;
  }
  if (rerun_only && tid == 0) status[draw] = 0;
  if (tid < 8) {
    dd[tid] = (dvec && tid < p) ? dvec[(d_batched ? (size_t)draw * p : 0) + tid] : 0.0;
    hh[tid] = (Hdiag && tid < p) ? Hdiag[(h_batched ? (size_t)draw * p : 0) + tid] : 0.0;
  }
  __syncthreads();

  // ================================================================================================================
  // wavefront B: register blocks of Q, P; prologue (R Q R', Tc, P0 by doubling)
  // ================================================================================================================
  int pr[BS], pcx[BS], ocol[BS];
  double zcol[BS];
  double Jb[BS][BS], Pb[BS][BS];
  const bool in_state_block = (lr * BS < s) && (lc * BS < s);
  const bool w_rows = lr * BS < s;
#define STORE_PZT2()                                                                             \
  do {                                                                                           \
    _Pragma("unroll") for (int j = 0; j < BS; ++j) if (ocol[j] >= 0) {                           \
      _Pragma("unroll") for (int i = 0; i < BS; ++i) {                                           \
        const double pzv = zcol[j] * Pb[i][j];                                                   \
        PZt[(lr * BS + i) * PS + ocol[j]] = pzv;                                                 \
        if (!NARROW || lr * BS + i < SK) ZPt[ocol[j] * LDK + lr * BS + i] = pzv;                 \
      }                                                                                          \
    }                                                                                            \
  } while (0)
  if (!isA) {
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
    double Qb[BS][BS], Tb0[BS][BS];
    if (Rsel) {  // sym(R diag(q) R')[U,U] from the selection matrix (same expression and summation order as rqr_kernel)
      const int kp = (k_shocks + 1) & ~1;
      const double* Rg = Rsel + (size_t)draw * m_full * k_shocks;
      for (int idx = lane; idx < m_full * k_shocks; idx += 64) {
        const int i = idx / k_shocks, c = idx - i * k_shocks;
        Wt[i * kp + c] = Rg[idx];
      }
      const double* qd = qdiag + (q_batched ? (size_t)draw * k_shocks : 0);
      wave_fence();
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
      wave_fence();
      for (int idx = lane; idx < m_full * kp; idx += 64) Wt[idx] = 0.0;
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
        if (!NARROW || lc * BS + j < SK) Tc[(lr * BS + i) * LDK + lc * BS + j] = tv;
      }
    bool lyap_ok = true;
    if (!P0) {  // P0 = dlyap(T, RQR)[U,U] by doubling on the reduced model (statespace.py:814-815), as kalman_nt_kernel
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) Pb[i][j] = Qb[i][j];
      lyap_ok = false;
      for (int itl = 0; itl < 64; ++itl) {
        wave_fence();
        if (in_state_block) blk_store_lds<BS>(Pb, Pc, LDK, lr, lc);
        wave_fence();
        if (w_rows) {
          double Wb[BS][BS];
          blk_zero<BS>(Wb);
          mm_nt<BS, LDK>(Wb, Pc, Tc, s, lr, lc);
#pragma unroll
          for (int i = 0; i < BS; ++i)
#pragma unroll
            for (int j = 0; j < BS; ++j) Wt[(lc * BS + j) * LDK + lr * BS + i] = Wb[i][j];
        }
        double Ab[BS][BS];
        blk_zero<BS>(Ab);
        mm_acc_p<BS, false, LDK, LDK>(Ab, Tc, Tc, s, lr, lc);
        wave_fence();
        double Xb[BS][BS];
        blk_zero<BS>(Xb);
        mm_nt<BS, LDK>(Xb, Tc, Wt, s, lr, lc);
        wave_fence();
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
      wave_fence();
      knt_store_cols<BS, NARROW ? SK : 0>(Tb0, Tc, LDK, lr, lc);
      for (int idx = lane; idx < NP * LDK; idx += 64) Wt[idx] = 0.0;
    }
    // J = jit_P (Tc Tc')[U,U] + Q: the constant of the predicted-form recursion (Tc has its non-zero columns in the state block)
    wave_fence();
    {
      double TTb[BS][BS];
      blk_zero<BS>(TTb);
      mm_nt<BS, LDK>(TTb, Tc, Tc, s, lr, lc);
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) Jb[i][j] = fma(cv.jit_P, TTb[i][j], Qb[i][j]);
    }
    if (in_state_block) blk_store_lds<BS>(Pb, Pc, LDK, lr, lc);
    STORE_PZT2();
    if (lane == 0) flags[2] = lyap_ok ? 1 : 0;
  }
  __syncthreads();
  if (flags[2] == 0) {  // (uniform)
    if (tid == 0) {
      status[draw] |= DSGE_ST_LYAP_FAIL;
      logp_out[draw] = -INFINITY;
    }
    return;
  }

  // ================================================================================================================
  // wavefront A: per-lane constants of the measurement update (8 replicas of an 8-lane group: lane -> row r8 of F)
  // ================================================================================================================
  const int r8 = lane & 7, g8 = lane >> 3;
  const int r_zpos = (r8 < p) ? zpos[r8] : 0;
  const double r_zv = (r8 < p) ? zv[r8] : 0.0, r_dd = dd[r8], r_hh = hh[r8];
  const int v_zpos = (lane < p) ? zpos[lane] : 0;
  const double v_zv = (lane < p) ? zv[lane] : 0.0, v_dd = (lane < 8) ? dd[lane & 7] : 0.0;
  double quad_sum = 0.0, quad_comp = 0.0, ld_mant = 1.0;
  int ld_exp = 0, n_ll_steps = 0, n_obs_entries = 0, steady_step = -1;
  const int r8c = (r8 < p) ? r8 : (p > 0 ? p - 1 : 0);
  double yt_next = (isA && T_len > 0) ? y[r8c] : 0.0;
  // what the steady loop reuses of A's last full step
  double fr[8], inv_own = 1.0, step_mant = 1.0;
  int step_exp = 0, n_obs = 0;
  unsigned long long omask = 0ull;
#pragma unroll
  for (int q = 0; q < 8; ++q) fr[q] = 0.0;

  long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long tk_start = DBG ? clock64() : 0;
  int t = 0;
  while (t < T_len) {
    long long tk0 = DBG ? clock64() : 0;
    // both wavefronts: flags[0] = the covariance has reached its fixed point at the end of the previous step
    const int steady_now = flags[0];
    if (steady_now) {
      // ==== steady-state steps: mean recursion only, while the missing-data mask stays the same (wavefront A, registers) ====
      if (isA) {
        if (steady_step < 0) steady_step = t;
        double trow[SK], finv_row[8], kr_ss[8];
        double av_reg = (lane < m) ? av[lane] : 0.0;
#pragma unroll
        for (int kk = 0; kk < SK; ++kk) trow[kk] = (lane < NP) ? Tc[lane * LDK + kk] : 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          finv_row[q] = (lane < 8) ? fr[q] * inv_own : 0.0;
          kr_ss[q] = (lane < m) ? Ks[lane * PS + q] : 0.0;
        }
        while (t < T_len) {
          const double yt_s = yt_next;
          const bool obs_s = (lane < p) && (yt_s == yt_s) && (yt_s != missing_fill);
          if (__ballot(obs_s) != omask) break;
          yt_next = y[(size_t)((t + 1 < T_len) ? t + 1 : t) * p + r8c];
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
          ++t;
        }
        if (lane < m) av[lane] = av_reg;
        if (lane == 0) {
          flags[0] = 0;  // a step with another mask: the full update resumes from the current covariance
          flags[1] = t;
        }
      }
      __syncthreads();
      t = flags[1];
      __syncthreads();  // (flags[1] is read by everybody before anybody can write it again)
      if constexpr (DBG) ph[5] += clock64() - tk0;
      continue;
    }

    // ==== full step ====
    if (isA) {
      // ---- (a) missing-data mask; every LDS operand of the update is requested up front
      const double yt = yt_next;
      yt_next = y[(size_t)((t + 1 < T_len) ? t + 1 : t) * p + r8c];
      const bool obs = (r8 < p) && (yt == yt) && (yt != missing_fill);
      omask = __ballot(obs) & 0xffull;
      n_obs = __popcll(omask);
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
      // ---- (b) innovation v[r8] and row r8 of F = Zm P Zm' + Hm + jit_F I
      const double c_r = obs ? r_zv : 0.0;
      const double v_r = (obs ? yt : 0.0) - (((obs || !cv.mask_d) ? r_dd : 0.0) + c_r * a_sel);
      const double dg = (r8 < p) ? ((obs ? r_hh : 0.0) + cv.jit_F) : 1.0;
      if (lane < 8) vv[lane] = v_r;
      asm volatile("" ::: "memory");
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const double tq = (q & 1) ? fr2[q >> 1].y : fr2[q >> 1].x;
        const double wq = ((omask >> q) & 1ull) ? 1.0 : 0.0;
        const double f = (c_r * tq) * wq;
        fr[q] = (q == r8) ? f + dg : f;
      }
      // ---- (c) Finv by Gauss-Jordan, one row per lane, rows left unscaled (as kalman_nt_kernel)
      step_mant = 1.0;
      inv_own = 1.0;
      step_exp = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < p) {
          double rowj[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) rowj[q] = readlane_f64(fr[q], j);
          const double inv = fast_rcp(rowj[j]);
          const bool is_j = (r8 == j);
          const double ci = is_j ? 0.0 : fr[j] * inv;
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (q != j) fr[q] = fma(-ci, rowj[q], fr[q]);
          fr[j] = is_j ? 1.0 : -ci;
          inv_own = is_j ? inv : inv_own;
          int e;
          step_mant *= frexp(rowj[j], &e);
          step_exp += e;
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
      // ---- (d) K = (P Zm') Finv, V = P Zm' + jit_V K, a+ = a + K v
#pragma unroll
      for (int ps = 0; ps < BS; ++ps) {
        const int i = g8 + 8 * ps;
        double k0 = 0.0, k1 = 0.0;
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) {
          k0 = fma(pz2[ps][q2].x, fr[2 * q2], k0);
          k1 = fma(pz2[ps][q2].y, fr[2 * q2 + 1], k1);
        }
        const double kk = obs ? (k0 + k1) * inv_own : 0.0;
        Ks[i * PS + r8] = kk;
        double part = kk * v_r;
        part += dpp_move_f64<0xB1, 0xf>(part);
        part += dpp_move_f64<0x4E, 0xf>(part);
        part += dpp_move_f64<0x141, 0xf>(part);
        if (r8 == 0) af[i] = avi[ps] + part;
      }
      if constexpr (DBG) {
        const long long tk1 = clock64();
        ph[1] += tk1 - tk0;
        tk0 = tk1;
      }
      // ---- (d') TK = Tc K = G Finv and TV = Tc V = G + jit_V TK, from B's panel G = Tc (P Z')[S] (unmasked): the same contraction
      //      as the gain, on rows of G.  B formed G at the head of this step -- long ago -- and raised flags[3]
      while (__hip_atomic_load(&flags[3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != t + 1) __builtin_amdgcn_s_sleep(1);
      {
        double2 gz2[BS][4];
        double gzo[BS];
#pragma unroll
        for (int ps = 0; ps < BS; ++ps) {
          const int i = g8 + 8 * ps;
#pragma unroll
          for (int q2 = 0; q2 < 4; ++q2) gz2[ps][q2] = *reinterpret_cast<const double2*>(&Gs[i * PS + 2 * q2]);
          gzo[ps] = Gs[i * PS + r8];
        }
#pragma unroll
        for (int ps = 0; ps < BS; ++ps) {
          const int i = g8 + 8 * ps;
          double k0 = 0.0, k1 = 0.0;
#pragma unroll
          for (int q2 = 0; q2 < 4; ++q2) {
            k0 = fma(gz2[ps][q2].x, fr[2 * q2], k0);
            k1 = fma(gz2[ps][q2].y, fr[2 * q2 + 1], k1);
          }
          const double tkk = obs ? (k0 + k1) * inv_own : 0.0;
          TKs[i * PS + r8] = tkk;
          TVs[i * PS + r8] = -fma(cv.jit_V, tkk, obs ? gzo[ps] : 0.0);
        }
      }
      if constexpr (DBG) {
        const long long tk1 = clock64();
        ph[2] += tk1 - tk0;
        tk0 = tk1;
      }
    } else {
      // ---- wavefront B: first G = Tc (P Z')[S] (24 x 18 by 18 x 8, lane (lr, lc): BS rows of column lc) for A's thin products --
      //      A needs it behind its elimination, 3 k cycles from now; flags[3] tells it (release / acquire, no barrier)
      {
        // both operands row-major along k (rows of Tc, row lc of Z P): one ds_read_b128 per row and k-pair; the loads of a half of
        // the k range are requested together, in front of its FMAs
        double g0[BS], g1[BS];
#pragma unroll
        for (int i = 0; i < BS; ++i) g0[i] = g1[i] = 0.0;
        constexpr int NQ = SK / 2, NH = (NQ + 1) / 2;  // k-pairs, per half
        const double2* tp = reinterpret_cast<const double2*>(Tc + lr * BS * LDK);
        const double2* zp = reinterpret_cast<const double2*>(ZPt + lc * LDK);
#pragma unroll
        for (int h0 = 0; h0 < NQ; h0 += NH) {
          double2 ta[BS][NH], zb[NH];
#pragma unroll
          for (int q = 0; q < NH; ++q) {
            const int qq = (h0 + q < NQ) ? h0 + q : NQ - 1;
            zb[q] = zp[qq];
#pragma unroll
            for (int i = 0; i < BS; ++i) ta[i][q] = tp[i * (LDK / 2) + qq];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < NH; ++q)
            if (h0 + q < NQ) {
#pragma unroll
              for (int i = 0; i < BS; ++i) {
                g0[i] = fma(ta[i][q].x, zb[q].x, g0[i]);
                g1[i] = fma(ta[i][q].y, zb[q].y, g1[i]);
              }
            }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < BS; ++i) Gs[(lr * BS + i) * PS + lc] = g0[i] + g1[i];
        wave_fence();
        if (lane == 0) __hip_atomic_store(&flags[3], t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if constexpr (DBG) {
        const long long tk1 = clock64();
        ph[2] += tk1 - tk0;
        tk0 = tk1;
      }
      // ---- W0 = P[S,S] Tc' (stored transposed), X0 = Tc W0 -- of the PREDICTED covariance
      if (w_rows) {
        double Wb[BS][BS];
        blk_zero<BS>(Wb);
        mm_nt<BS, LDK>(Wb, Pc, Tc, s, lr, lc);
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) Wt[(lc * BS + j) * LDK + lr * BS + i] = Wb[i][j];
      }
      wave_fence();
      if constexpr (DBG) {
        const long long tk1 = clock64();
        ph[0] += tk1 - tk0;
        tk0 = tk1;
      }
    }
    double Xb[BS][BS], Xs[BS][BS];
    if (!isA) {
      blk_zero<BS>(Xb);
      mm_nt<BS, LDK>(Xb, Tc, Wt, s, lr, lc);
      // Xs = sym(X0) + J while A is still eliminating: behind the barrier only the rank-p term is left
      const int src = (lc << 3) | lr;
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) {
          const double xt = __shfl(Xb[j][i], src, 64);
          const double xs = 0.5 * (Xb[i][j] + xt) + Jb[i][j];
          Xs[i][j] = xs;
        }
      if constexpr (DBG) {
        const long long tk1 = clock64();
        ph[1] += tk1 - tk0;
        tk0 = tk1;
      }
    }
    __syncthreads();  // K, TK, TV, a+ (A) and Xs (B, registers) are ready
    if constexpr (DBG) {  // (wait at the barrier: [3])
      const long long tk1 = clock64();
      ph[3] += tk1 - tk0;
      tk0 = tk1;
    }
    if (isA) {
      // a_{t+1} = Tc a+[:s] while B corrects the covariance
      if (lane < m) {
        const double2* trow2 = reinterpret_cast<const double2*>(Tc + lane * LDK);
        const double2* af2 = reinterpret_cast<const double2*>(af);
        double s0 = 0.0, s1 = 0.0;
        for (int kk = 0; 2 * kk < s; ++kk) {
          const double2 tv2 = trow2[kk], fv = af2[kk];
          s0 = fma(tv2.x, fv.x, s0);
          s1 = fma(tv2.y, fv.y, s1);
        }
        av[lane] = s0 + s1;
      }
    } else {
      // ---- P_{t+1} = Xs - TK TV' (the rank-p term is symmetric up to rounding; the next step's sym(X0) takes that out again);
      //      steady test against P_t; state block and P Z' panel to LDS
      {
        double2 ka[BS], vb[BS], kan[BS], vbn[BS];
#pragma unroll
        for (int i = 0; i < BS; ++i) ka[i] = *reinterpret_cast<const double2*>(&TKs[(lr * BS + i) * PS]);
#pragma unroll
        for (int j = 0; j < BS; ++j) vb[j] = *reinterpret_cast<const double2*>(&TVs[(lc * BS + j) * PS]);
#pragma unroll
        for (int o2 = 0; o2 < 4; ++o2) {
          if (o2 < 3) {
#pragma unroll
            for (int i = 0; i < BS; ++i) kan[i] = *reinterpret_cast<const double2*>(&TKs[(lr * BS + i) * PS + 2 * (o2 + 1)]);
#pragma unroll
            for (int j = 0; j < BS; ++j) vbn[j] = *reinterpret_cast<const double2*>(&TVs[(lc * BS + j) * PS + 2 * (o2 + 1)]);
          }
#pragma unroll
          for (int i = 0; i < BS; ++i)
#pragma unroll
            for (int j = 0; j < BS; ++j) {
              Xs[i][j] = fma(ka[i].x, vb[j].x, Xs[i][j]);
              Xs[i][j] = fma(ka[i].y, vb[j].y, Xs[i][j]);
            }
#pragma unroll
          for (int i = 0; i < BS; ++i) ka[i] = kan[i];
#pragma unroll
          for (int j = 0; j < BS; ++j) vb[j] = vbn[j];
        }
      }
      double pm = 0.0;
      if (steady_tol > 0.0) {
        double pscale = 0.0;
#pragma unroll
        for (int i = 0; i < BS; ++i) pscale = fmax(pscale, fabs(Pb[i][i]));
        if (lr != lc) pscale = 0.0;
        const unsigned hi = (unsigned)__double2hiint(pscale), lo = (unsigned)__double2loint(pscale);
        const unsigned mhi = wave_max_u32(hi);
        const unsigned mlo = wave_max_u32(hi == mhi ? lo : 0u);
        pm = __hiloint2double((int)mhi, (int)mlo);
      }
      double dmax = 0.0;
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) {
          dmax = fmax(dmax, fabs(Xs[i][j] - Pb[i][j]));
          Pb[i][j] = Xs[i][j];
        }
      bool steady = false;
      if (steady_tol > 0.0) {
        // max|P_{t+1|t} - P_{t|t-1}| over the state block <= tol max|P| (the other blocks are functions of it)
        const bool viol = in_state_block && !(dmax <= steady_tol * pm);
        steady = (t > 0) && (__ballot(viol) == 0ull);
      }
      if (in_state_block) blk_store_lds<BS>(Pb, Pc, LDK, lr, lc);
      STORE_PZT2();
      if (lane == 0) flags[0] = steady ? 1 : 0;
    }
    if constexpr (DBG) {  // ([4]: A: a_{t+1}; B: the correction, steady test, stores)
      const long long tk1 = clock64();
      ph[4] += tk1 - tk0;
      tk0 = tk1;
    }
    ++t;
    __syncthreads();  // P Z', Pc, a_{t+1}, flags of the next step
    if constexpr (DBG) {
      ph[6] += clock64() - tk0;  // wait at the closing barrier
      ph[7] += 1;
    }
  }
#undef STORE_PZT2
  if (DBG && dbg && draw == 0 && lane == 0) {
    long long* o = dbg + (isA ? 0 : 8);
    for (int k2 = 0; k2 < 8; ++k2) o[k2] = ph[k2];
    if (isA) o[5] = clock64() - tk_start;  // (A: total of the time loop in [5]; steady cycles are in B's [5])
  }
  if (isA) {
    const double quad_total = wave_sum_dpp(quad_sum - quad_comp);
    if (lane == 0) {
      const double logdet = log(ld_mant) + (double)ld_exp * LN2;
      const double ll = -0.5 * (cv.ll_terms(n_ll_steps, n_obs_entries, p) * LN2PI + logdet + quad_total);
      logp_out[draw] = ll;
      if (steady_at) steady_at[draw] = steady_step;
      if (!((ll == ll) && (fabs(ll) < 1.797e308))) status[draw] |= DSGE_ST_FILTER_NONFINITE;
    }
  }
}

}  // namespace dsge

// Reverse-mode gradient of the Kalman-filter log-likelihood (second half).
//
// The reference obtains d logp / d theta by pytensor autodiff through solve_discrete_lyapunov and the
// pymc_extras filter scan (gEconpy/model/statespace.py:814-815, 1151-1157).  This kernel is that reverse
// sweep written out by hand for ONE draw per wavefront, on the exactly reduced model of kalman_sel_kernel
// (U = states + observed variables, states first; selector design matrix, p <= 8):
//
//   forward  (t = 0..T_len-1), storing the predicted (a_t, P_t) of every step in a global scratch:
//     M = P Zm',  F = Zm M + Hm + jit_F I,  K = M F^-1,  v = ym - d - Zm a,  a+ = a + K v,
//     P+ = P - K (M + jit_V K)' + jit_P I,  ll_t = -1/2 (c ln 2pi + ln det F + v' F^-1 v),
//     (jit_F, jit_P, jit_V, c and the masking of d are the run-time conventions of FilterConv, dsge_device.hpp: constants of
//      the sweep -- of the formulas below only Kbar carries one, jit_V)
//     a' = T a+,  P' = sym(T P+ T') + G                                  
//   reverse  (t = T_len-1..0), cotangents (abar, Pbar) of the predicted moments of step t+1:
//     Tbar += abar a+' + 2 Pbar T P+,   Gbar += Pbar,   a+bar = T' abar,   P+bar = T' Pbar T
//     vbar  = -lam F^-1 v + K' a+bar
//     Y     = P+bar K
//     Kbar  = a+bar v' - 2 Y (F + jit_V I)
//     Mbar  = Kbar F^-1
//     Fbar  = -lam/2 (F^-1 - F^-1 v v' F^-1) - K' Y - K' Mbar
//     Mbar += Zm' Fbar,   hbar += w o diag(Fbar)
//     Pbar  = sym(P+bar + Mbar Zm),   abar = a+bar - Zm' vbar,   dbar -= vbar
//   (lam = 1 unless every entry of y_t is missing), then the stationary initial covariance
//   P0 = dlyap(T, G):   S = dlyap(T', Pbar_0) by doubling,   Gbar += S,   Tbar += 2 S T P0.
// Outputs: logp, Tbar (cotangent of T; rows U, columns S -- the columns of T outside S are structurally
// zero functions of the parameters), Gbar (cotangent of sym(R Q R'), block [U,U]), dbar, hbar.
// Zbar is not produced (the selector design matrix is a constant of the model, statespace.py:282-296).
//
// Steady segments: the forward sweep freezes K, F^-1 and ln det F of the segment's source step, so the reverse sweep
// collects the cotangents the segment's steps send to those shared quantities (Kacc, Qacc, nlam below) and runs the
// covariance part of the step ONCE per segment, at the source step; a steady step costs two matrix-vector products.
// Every matrix lives in LDS, products go through mm_acc register blocks.
#pragma once
#include "dsge_device.hpp"
#include "dsge_kalman2.hpp"
#include "dsge_kalman_rec.hpp"
#include "dsge_mfma4.hpp"

#include "../../include/dsge_hip.h"

namespace dsge {

// sum_k a[k sa] b[k sb] on four accumulators, the loads of a trip requested together: a loop bounded by the run-time u is not
// unrolled, and with one accumulator every term waited for its own pair of LDS loads.
__device__ __forceinline__ double kg_dot4(const double* a, int sa, const double* b, int sb, int K) {
  double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
  int k = 0;
  for (; k + 4 <= K; k += 4) {
    const double a0 = a[k * sa], a1 = a[(k + 1) * sa], a2 = a[(k + 2) * sa], a3 = a[(k + 3) * sa];
    const double b0 = b[k * sb], b1 = b[(k + 1) * sb], b2 = b[(k + 2) * sb], b3 = b[(k + 3) * sb];
    c0 = fma(a0, b0, c0);
    c1 = fma(a1, b1, c1);
    c2 = fma(a2, b2, c2);
    c3 = fma(a3, b3, c3);
  }
  for (; k < K; ++k) c0 = fma(a[k * sa], b[k * sb], c0);
  return (c0 + c1) + (c2 + c3);
}

#ifndef KG_MFMA_PRODUCTS
#define KG_MFMA_PRODUCTS 1
#endif
#ifndef KG_TAIL_WAVES
#define KG_TAIL_WAVES 2
#endif

template <int BS>
struct KgSmem {
  static constexpr int NP = Tile<BS>::NP, LDM = Tile<BS>::LD, PS = 9;
  // 5 NP x LDM matrices (Tc, P, Pb, X1, X2), 5 NP x PS panels (M, K, Kb, Mb, Y),
  // 4 8x8 (F, Fi, Fb, tmp), 6 NP vectors, 10 8-vectors, ints perm NP + zpos 8
  __host__ __device__ static constexpr size_t doubles() {
    return 5 * (size_t)NP * LDM + 5 * (size_t)NP * PS + 4 * 64 + 6 * NP + 10 * 8 + NP / 2 + 4;
  }
  static constexpr size_t bytes = sizeof(double) * doubles();
};

// dst (LDS, NP x LDM) = or += alpha * A B   (TB: A B').  dst must not alias A or B; caller fences.
template <int BS, bool TB>
__device__ __forceinline__ void kg_mm(double* dst, const double* A, const double* B, int K, double alpha, bool accumulate,
                                      int lr, int lc) {
  constexpr int LDM = Tile<BS>::LD;
  double acc[BS][BS];
  blk_zero<BS>(acc);
  mm_acc<BS, TB>(acc, A, LDM, B, LDM, K, lr, lc);
#pragma unroll
  for (int i = 0; i < BS; ++i)
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      double* d = &dst[(lr * BS + i) * LDM + lc * BS + j];
      *d = accumulate ? fma(alpha, acc[i][j], *d) : alpha * acc[i][j];
    }
}

__device__ __forceinline__ double yt_or_zero(double yt) { return (yt == yt) ? yt : 0.0; }
__device__ __forceinline__ bool kg_in_u(int r, int c, int u) { return r < u && c < u; }

// The three u x u products of a full reverse step on the FP64 matrix core (round 6):  X2 = Pbar T,  t2 = (Pbar T) P+  (left in
// the Pbar buffer, which is dead from here to the end of the step),  P+bar = T' (Pbar T)  (symmetric: upper tiles, stored with
// their mirror images).  TMU = tiles of four that cover the u retained variables; the operands stay where the kernel keeps them
// (NP x LDM, odd LDM: mfma4_strided reads element by element).  Everything outside u x u reads as zero -- Tc, Pbar and the staged
// P+ are zero there -- and is written as zero.  Register blocks on the 8 x 8 lane grid pad 18 variables to 24 (1.8 x the flops, and
// the block products are bound by their LDS operand traffic: 8.4 k cycles per step for the three); tiles of four pad to 20 and the
// symmetric product computes 15 of its 25 tiles.
template <int BS, int TMU>
__device__ __forceinline__ void kg_cov_products_mf(double* __restrict__ Pb, const double* __restrict__ Tc,
                                                   const double* __restrict__ X1, double* __restrict__ X2,
                                                   double* __restrict__ Ps, int u, int lane) {
  constexpr int NP = Tile<BS>::NP, LDM = Tile<BS>::LD, DUMP = (NP - 1) * LDM + LDM - 1;  // (column LDM - 1: padding nobody reads)
  using MP = Mfma4Map<TMU, TMU>;
  using UX = Mfma4Upper<TMU>;
  const int blk = (lane >> 2) & 3, i4 = lane & 3, kq = lane >> 4;
  mfma4_strided<TMU, TMU, TMU, LDM, 1, LDM, 1>(Pb, Tc, lane, [&](int g, double d) {
    const int at = (4 * MP::ta(g, blk) + kq) * LDM + 4 * MP::tb(g, blk) + i4;
    X2[MP::live(g, blk) ? at : DUMP] = d;
  });
  wave_sync();
  mfma4_strided<TMU, TMU, TMU, LDM, 1, LDM, 1>(X2, X1, lane, [&](int g, double d) {
    const int r = 4 * MP::ta(g, blk) + kq, c = 4 * MP::tb(g, blk) + i4;
    Pb[MP::live(g, blk) ? r * LDM + c : DUMP] = (r < u && c < u) ? d : 0.0;
  });
  int rowa[UX::NG], rowb[UX::NG];
  mfma4_upper_rows<TMU>(lane, rowa, rowb);
  mfma4_strided_upper<TMU, TMU, 1, LDM, LDM, 1>(Tc, X2, lane, rowa, rowb, [&](int g, double d) {
    // this lane's element: row 4 ta + kq, column 4 tb + i4 (rowa / rowb carry 4 ta + i4, 4 tb + i4)
    const int r = rowa[g] - i4 + kq, c = rowb[g];
    int ta, tb;
    bool live;
    UX::tile(g, blk, ta, tb, live);
    const bool st = live && r <= c;
    Ps[st ? r * LDM + c : DUMP] = d;
    Ps[st ? c * LDM + r : DUMP] = d;
  });
  wave_sync();
}

// (Measured, round 5: capping the SPLIT instance at 256 registers for a second wavefront per SIMD -- it holds 308 -- spills 52
// dwords into the mean-side loop: the reverse launch 3.0 -> 4.5 ms at unchanged residency; LDS, 36.6 KB, would have to shrink
// below 27 KB as well before a second wavefront fits.)
// One doubling step of  S = dlyap(A', Pbar_0)  on the matrix core (round 6):  S += A' (S A)  on the upper tiles with the mirror image,
// A <- A A.  S in Pb (symmetric), A in X1, X2 scratch; everything outside u x u is and stays zero.  Returns max |increment|, max |S|.
template <int BS, int TMU>
__device__ __forceinline__ void kg_dlyap_step_mf(double* __restrict__ Pb, double* __restrict__ X1, double* __restrict__ X2, int u,
                                                 int lane, double& dmax, double& smax) {
  constexpr int NP = Tile<BS>::NP, LDM = Tile<BS>::LD, DUMP = (NP - 1) * LDM + LDM - 1;
  using MP = Mfma4Map<TMU, TMU>;
  using UX = Mfma4Upper<TMU>;
  const int blk = (lane >> 2) & 3, i4 = lane & 3, kq = lane >> 4;
  double a2r[MP::NG];
  mfma4_strided<TMU, TMU, TMU, LDM, 1, LDM, 1>(Pb, X1, lane, [&](int g, double d) {  // S A
    const int at = (4 * MP::ta(g, blk) + kq) * LDM + 4 * MP::tb(g, blk) + i4;
    X2[MP::live(g, blk) ? at : DUMP] = d;
  });
  mfma4_strided<TMU, TMU, TMU, LDM, 1, LDM, 1>(X1, X1, lane, [&](int g, double d) { a2r[g] = d; });  // A A
  wave_sync();
  int rowa[UX::NG], rowb[UX::NG];
  mfma4_upper_rows<TMU>(lane, rowa, rowb);
  double dm = 0.0, sm = 0.0;
  mfma4_strided_upper<TMU, TMU, 1, LDM, LDM, 1>(X1, X2, lane, rowa, rowb, [&](int g, double d) {  // A' (S A), upper tiles
    const int r = rowa[g] - i4 + kq, c = rowb[g];
    int ta, tb;
    bool live;
    UX::tile(g, blk, ta, tb, live);
    const bool own = live && r <= c && c < u;
    const double nv = Pb[own ? r * LDM + c : DUMP] + d;
    Pb[own ? r * LDM + c : DUMP] = own ? nv : 0.0;
    Pb[own ? c * LDM + r : DUMP] = own ? nv : 0.0;
    dm = nanmax(dm, own ? fabs(d) : 0.0);
    sm = nanmax(sm, own ? fabs(nv) : 0.0);
  });
  wave_sync();
#pragma unroll
  for (int g = 0; g < MP::NG; ++g) {
    const int at = (4 * MP::ta(g, blk) + kq) * LDM + 4 * MP::tb(g, blk) + i4;
    X1[MP::live(g, blk) ? at : DUMP] = a2r[g];
  }
  dmax = wave_nanmax(dm);
  smax = wave_nanmax(sm);
  wave_sync();
}

// SPLIT = true (round 5): the reverse sweep ALONE -- the forward sweep ran as kalman_nt_kernel<BS, .., REC = true>
// (dsge_kalman_nt.hpp: the logp kernel's full step at two wavefronts per SIMD instead of this kernel's at one), which wrote the
// records (dsge_kalman_rec.hpp), logp and the status words.  rerun_only (SPLIT = false): only the draws that launch flagged
// DSGE_ST_INTERNAL_RERUN (it could not take them) are processed, from scratch.
template <int BS, bool SPLIT = false>
__global__ __launch_bounds__(64) void kalman_grad_kernel(
    const double* __restrict__ T, const double* __restrict__ RQR, const double* __restrict__ Z, int z_batched,
    const double* __restrict__ dvec, int d_batched, const double* __restrict__ Hdiag, int h_batched,
    const double* __restrict__ y, int batch, int m_full, int p, int T_len, FilterConv cv, double missing_fill,
    double steady_tol, double* __restrict__ store, double* __restrict__ logp_out, int32_t* __restrict__ status,
    double* __restrict__ Tbar_out, double* __restrict__ Gbar_out, double* __restrict__ dbar_out,
    double* __restrict__ hbar_out, long long* __restrict__ dbg, const int32_t* __restrict__ order, int rerun_only,
    int tail_valid) {
  constexpr int NP = KgSmem<BS>::NP, LDM = KgSmem<BS>::LDM, PS = KgSmem<BS>::PS;
  constexpr bool KG_MF = KG_MFMA_PRODUCTS && BS <= 4;  // the reverse step's three products on the matrix core (kg_cov_products_mf)
  // doubles stored per time step: for a FULL step the results of its covariance update -- P+ (NP x NP), K (NP x 8), F^-1,
  // F (8 x 8 each) -- so that the reverse sweep loads them instead of repeating the update; for every step a_t and the
  // index of the step whose covariance it shares.  The initial covariance P_0 sits behind the last step.
  using RC = KgRec<BS>;  // (dsge_kalman_rec.hpp: shared with the forward sweep of kalman_nt_kernel<.., REC>)
  constexpr size_t OFF_K = RC::OFF_K, OFF_FI = RC::OFF_FI, OFF_F = RC::OFF_F, OFF_A = RC::OFF_A, OFF_SRC = RC::OFF_SRC,
                   OFF_PREV = RC::OFF_PREV, STEP = RC::STEP;  // OFF_PREV: source of the PREVIOUS segment
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Tc = smem;              // transition, states-first ordering (columns >= s are zero)
  double* Ps = Tc + NP * LDM;     // predicted covariance of the current step
  double* Pb = Ps + NP * LDM;     // cotangent of the predicted covariance
  double* X1 = Pb + NP * LDM;     // P+ / scratch  (the cotangents of T and G accumulate in register blocks)
  double* X2 = X1 + NP * LDM;
  double* Mp = X2 + NP * LDM;     // M = P Zm'          NP x PS panels
  double* Gs = Pb;                // forward sweep only: G[U,U] lives in the (not yet used) cotangent buffer
  double* Kp = Mp + NP * PS;      // K
  double* Kb = Kp + NP * PS;      // Kbar
  double* Mb = Kb + NP * PS;      // Mbar
  double* Yp = Mb + NP * PS;      // Y = P+bar K
  double* Fs = Yp + NP * PS;      // F
  double* Fi = Fs + 64;           // F^-1
  double* Fb = Fi + 64;           // Fbar
  double* Ft = Fb + 64;           // scratch
  double* av = Ft + 64;           // a (predicted)
  double* ap = av + NP;           // a+
  double* ab = ap + NP;           // abar
  double* apb = ab + NP;          // a+bar
  double* t1 = apb + NP;
  double* t2 = t1 + NP;
  double* vv = t2 + NP;           // v
  double* vb = vv + 8;            // vbar
  double* fiv = vb + 8;           // F^-1 v
  double* dd = fiv + 8;
  double* hh = dd + 8;
  double* zv = hh + 8;
  double* ww = zv + 8;            // observation weights of the step
  double* db = ww + 8;            // dbar (accumulated)
  double* hb = db + 8;            // hbar (accumulated)
  double* sp = hb + 8;            // spare
  int* perm = (int*)(sp + 8);
  int* zpos = perm + NP;
  const int lane = threadIdx.x, lr = lane >> 3, lc = lane & 7;
  const int fo = lane >> 3, fq = lane & 7;
  const double LN2PI = 1.8378770664093453;

  long long qh[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // debug (draw 0), inside FULL steps: fwd update_cov + record, fwd mean, fwd predict | rev record, rev mean side, rev products, rev panels, rev Pbar
  long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // debug (draw 0): setup+P0, fwd full, fwd steady, rev full, rev steady, tail, #full, #steady
  // (one draw per workgroup, grid = batch: no grid-stride loop for the compiler to hoist loop invariants out of and spill
  //  them around the time loops -- see kalman_nt_kernel)
  for (int bi = blockIdx.x; bi < batch; bi = batch) {
    const int draw = order ? order[bi] : bi;  // likely slow draws first (kalman_order_kernel), see kalman_sel_kernel
    const bool tm = dbg && draw == 0;
    long long tk0 = tm ? clock64() : 0;
    const size_t off = (size_t)draw * m_full * m_full;
    double* Tbo = Tbar_out + off;
    double* Gbo = Gbar_out + off;
    if (rerun_only) {  // (wave-uniform)
      if (status[draw] != DSGE_ST_INTERNAL_RERUN) continue;
      wave_sync();
      if (lane == 0) status[draw] = 0;
      wave_sync();
    }
    for (int idx = lane; idx < m_full * m_full; idx += 64) {
      Tbo[idx] = 0.0;
      Gbo[idx] = 0.0;
    }
    if (lane < p) {
      if (dbar_out) dbar_out[(size_t)draw * p + lane] = 0.0;
      if (hbar_out) hbar_out[(size_t)draw * p + lane] = 0.0;
    }
    if (status[draw] != 0) {
      if (lane == 0) logp_out[draw] = -INFINITY;
      continue;
    }
    wave_sync();
    for (int idx = lane; idx < (int)KgSmem<BS>::doubles(); idx += 64) smem[idx] = 0.0;
    // ---- reduction to U = S u O, states first (same construction as kalman_sel_kernel) -------------
    const double* Zg = Z + (z_batched ? (size_t)draw * p * m_full : 0);
    bool is_state = false;
    {  // eight unconditional loads in flight per trip (a short-circuit || here is one round trip per row)
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
    bool ok = p <= 8;
    for (int o = 0; o < p; ++o) {
      const double zl = (lane < m_full) ? Zg[(size_t)o * m_full + lane] : 0.0;
      const unsigned long long b = __ballot(zl != 0.0);
      if (__popcll(b) != 1 || ((used & b) != 0ull)) ok = false;
      used |= b;
      obsmask |= b;
    }
    const unsigned long long extra = obsmask & ~colmask;
    const int s = __popcll(colmask);
    const int u = s + __popcll(extra);
    ok = ok && (u <= NP);
    if (!ok) {  // dense design matrix or model too large for this tile: gradient not available here
      if (lane == 0) {
        status[draw] |= DSGE_ST_GRAD_UNSUPPORTED;
        logp_out[draw] = __longlong_as_double(0x7ff8000000000000ll);
      }
      continue;
    }
    int my_pos = -1;
    if (lane < m_full) {
      const unsigned long long below = (1ull << lane) - 1ull;
      if ((colmask >> lane) & 1ull)
        my_pos = __popcll(colmask & below);
      else if ((extra >> lane) & 1ull)
        my_pos = s + __popcll(extra & below);
      if (my_pos >= 0) perm[my_pos] = lane;
    }
    for (int o = 0; o < p; ++o) {
      const double zl = (lane < m_full) ? Zg[(size_t)o * m_full + lane] : 0.0;
      if (zl != 0.0) {
        zpos[o] = my_pos;
        zv[o] = zl;
      }
    }
    if (lane < 8) {
      dd[lane] = (dvec && lane < p) ? dvec[(d_batched ? (size_t)draw * p : 0) + lane] : 0.0;
      hh[lane] = (Hdiag && lane < p) ? Hdiag[(h_batched ? (size_t)draw * p : 0) + lane] : 0.0;
    }
    wave_sync();
    // Tc <- T[U,U];  Gs <- G[U,U]
    for (int idx = lane; idx < u * u; idx += 64) {
      const int i = idx / u, j = idx - i * u;
      const size_t g = (size_t)perm[i] * m_full + perm[j];
      const double tv = T[off + g];
      Tc[i * LDM + j] = tv;
      Gs[i * LDM + j] = RQR[off + g];
    }
    wave_sync();
    // per-draw record: T_len step records + P_0 (NP x NP) behind them -- kalman_grad_store_doubles_per_draw() on the host.
    // (Round 1 strode by T_len * STEP only: the P_0 of draw d sat on the step-0 record of draw d + 1, a cross-workgroup
    // race that made the cotangent of T non-repeatable for batches of ~100 draws and more.)
    double* st = store + (size_t)draw * RC::per_draw(T_len);
    const int lane_kernel = lane;
    // mask of step t -> ww; returns the ballot
    // (y_t, and in the reverse sweep a_t and the source index, are fetched ONE STEP AHEAD: a steady step is a few hundred
    // cycles of arithmetic, a dependent global load is a few thousand)
    auto load_mask = [&](double yt) -> unsigned long long {
      const bool obs = (lane < p) && (yt == yt) && (yt != missing_fill);
      const unsigned long long omask = __ballot(obs);
      if (lane < 8) ww[lane] = (lane < p && obs) ? 1.0 : 0.0;
      return omask;
    };
    if constexpr (!SPLIT) {
    // ---- P0 = dlyap(Tu, G) by doubling: P <- P + A P A', A <- A A  (A in X1) ---------------------
    for (int idx = lane; idx < NP * LDM; idx += 64) {
      Ps[idx] = Gs[idx];
      X1[idx] = Tc[idx];
    }
    wave_sync();
    bool lyap_ok = false;
    for (int itl = 0; itl < 64; ++itl) {
      kg_mm<BS, true>(X2, Ps, X1, u, 1.0, false, lr, lc);   // P A'
      wave_sync();
      double inc[BS][BS], a2[BS][BS];
      blk_zero<BS>(inc);
      blk_zero<BS>(a2);
      mm_acc<BS, false>(inc, X1, LDM, X2, LDM, u, lr, lc);  // A P A'
      mm_acc<BS, false>(a2, X1, LDM, X1, LDM, u, lr, lc);   // A A
      wave_sync();
      double dmax = 0.0, pmax = 0.0;
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) {
          double* pp = &Ps[(lr * BS + i) * LDM + lc * BS + j];
          *pp += inc[i][j];
          dmax = nanmax(dmax, fabs(inc[i][j]));
          pmax = nanmax(pmax, fabs(*pp));
          X1[(lr * BS + i) * LDM + lc * BS + j] = a2[i][j];
        }
      dmax = wave_nanmax(dmax);
      pmax = wave_nanmax(pmax);
      wave_sync();
      if (!(dmax == dmax) || !(pmax < 1e300)) break;
      if (dmax <= 1e-17 * pmax) {
        lyap_ok = true;
        break;
      }
    }
    if (!lyap_ok) {
      if (lane == 0) {
        status[draw] |= DSGE_ST_LYAP_FAIL;
        logp_out[draw] = -INFINITY;
      }
      continue;
    }
    // symmetrise P0 (the doubling keeps it symmetric up to rounding)
    {
      double a[BS][BS], b[BS][BS];
      blk_load_lds<BS>(a, Ps, LDM, lr, lc);
      blk_load_lds_t<BS>(b, Ps, LDM, lr, lc);
      wave_sync();
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) Ps[(lr * BS + i) * LDM + lc * BS + j] = 0.5 * (a[i][j] + b[i][j]);
    }
    wave_sync();

    // ---- the measurement update, split in its data-independent and data-dependent halves --------------
    // update_cov: from Ps and the mask weights ww -> Mp, Fs, Fi, Kp, X1 = P+; returns ln det F.
    auto update_cov = [&](double* rec) -> double {
      for (int idx = lane; idx < u * 8; idx += 64) {
        const int i = idx >> 3, o = idx & 7;
        Mp[i * PS + o] = (o < p) ? ww[o] * zv[o] * Ps[i * LDM + zpos[o]] : 0.0;
      }
      wave_sync();
      double f;
      if (fo < p && fq < p) {
        f = ww[fo] * zv[fo] * Mp[zpos[fo] * PS + fq];
        if (fo == fq) f += ww[fo] * hh[fo] + cv.jit_F;
      } else {
        f = (fo == fq) ? 1.0 : 0.0;
      }
      Fs[lane] = f;
      double det_m = 1.0;  // det F = det_m * 2^det_e (one logarithm per update instead of one per pivot)
      int det_e = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < p) {
          const double piv = readlane_f64(f, j * 9);
          const double rowj = __shfl(f, (j << 3) | fq, 64);
          const double colj = __shfl(f, (fo << 3) | j, 64);
          const double inv = fast_rcp(piv);
          const double ci = colj * inv;
          double nf = fma(-ci, rowj, f);
          nf = (fo == j) ? rowj * inv : nf;
          nf = (fq == j) ? -ci : nf;
          nf = (fo == j && fq == j) ? inv : nf;
          f = nf;
          int e;
          det_m *= frexp(piv, &e);
          det_e += e;
        }
      }
      Fi[lane] = f;
      wave_sync();
      for (int idx = lane; idx < u * 8; idx += 64) {
        const int i = idx >> 3, o = idx & 7;
        double sk = 0.0;
        for (int q = 0; q < 8; ++q) sk = fma(Mp[i * PS + q], Fi[q * 8 + o], sk);
        Kp[i * PS + o] = (o < p) ? sk : 0.0;
      }
      wave_sync();
      // P+ = P - K (M + jit K)' + jit I on this lane's BS x BS block (rows of K, rows of M + jit K in registers: a rank-8
      // update of nine entries; the entry-per-lane loop it replaces -- kept for the tiles whose register blocks leave no
      // room for 32 BS more registers -- ran six trips of 24 dependent LDS reads and an integer division each).  The block
      // goes to X1 for the prediction products and -- `rec` = the step's record -- straight from the registers to HBM in
      // lane-major order (entry (i, j) of lane l at rec[(i BS + j) 64 + l]: BS^2 coalesced stores, no staging pass through
      // LDS); the reverse sweep reads it back in the same order.
      if constexpr (BS <= 4) {
        double kr[BS][8], mk[BS][8];
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int o = 0; o < 8; ++o) {
            kr[i][o] = Kp[(lr * BS + i) * PS + o];
            mk[i][o] = fma(cv.jit_V, Kp[(lc * BS + i) * PS + o], Mp[(lc * BS + i) * PS + o]);
          }
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) {
            const int r = lr * BS + i, c = lc * BS + j;
            double sp2 = Ps[r * LDM + c];
#pragma unroll
            for (int o = 0; o < 8; ++o) sp2 = fma(-kr[i][o], mk[j][o], sp2);
            sp2 += (r == c && r < u) ? cv.jit_P : 0.0;
            X1[r * LDM + c] = sp2;
            rec[(i * BS + j) * 64 + lane] = sp2;
          }
      } else {
        for (int idx = lane; idx < u * u; idx += 64) {
          const int i = idx / u, j = idx - i * u;
          double sp2 = Ps[i * LDM + j];
          for (int o = 0; o < 8; ++o) sp2 = fma(-Kp[i * PS + o], fma(cv.jit_V, Kp[j * PS + o], Mp[j * PS + o]), sp2);
          X1[i * LDM + j] = sp2 + ((i == j) ? cv.jit_P : 0.0);
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) rec[(i * BS + j) * 64 + lane] = X1[(lr * BS + i) * LDM + lc * BS + j];
      }
      wave_sync();
      return log(det_m) + (double)det_e * 0.6931471805599453;
    };
    // update_mean: from av, y_t, ww, Kp, Fi -> vv, fiv, ap; returns v' F^-1 v.  (Fences on entry and exit.)
    auto update_mean = [&](double yt) -> double {
      wave_sync();
      if (lane < p) vv[lane] = ww[lane] * yt_or_zero(yt) - (((ww[lane] != 0.0 || !cv.mask_d) ? dd[lane] : 0.0) + ww[lane] * zv[lane] * av[zpos[lane]]);
      if (lane >= p && lane < 8) vv[lane] = 0.0;
      wave_sync();
      if (lane < 8) {
        double sfv = 0.0;
        for (int q = 0; q < 8; ++q) sfv = fma(Fi[lane * 8 + q], vv[q], sfv);
        fiv[lane] = (lane < p) ? sfv : 0.0;
      }
      if (lane < u) {
        double sa = av[lane];
        for (int o = 0; o < 8; ++o) sa = fma(Kp[lane * PS + o], vv[o], sa);
        ap[lane] = sa;
      }
      wave_sync();
      double quad = 0.0;
      for (int q = 0; q < 8; ++q) quad = fma(vv[q], fiv[q], quad);
      return quad;
    };

    // ---- forward sweep.  Step t stores (P_t, a_t, src_t): src_t = t for a full step; once the predicted
    // covariance has stopped moving (same rounding-level criterion as kalman_sel_kernel) and while the mask stays
    // the same, the steps are "steady": only a_t is stored and src_t names the step whose covariance they share.
    if (tm) {
      const long long tk1 = clock64();
      ph[0] += tk1 - tk0;
      tk0 = tk1;
    }
    double ll_acc = 0.0;
    long long n_ll = 0, n_entries = 0;  // (FilterConv::ll_terms)
    bool steady = false;
    unsigned long long smask = 0ull;
    int seg_src = -1;
    double seg_logdet = 0.0;
    double yt_next = (lane < p && T_len > 0) ? y[lane] : 0.0;
    for (int t = 0; t < T_len; ++t) {
      // (lane index re-derived from an opaque copy per step: the addresses computed from it are recomputed, not hoisted in front of
      //  the time loop and kept -- or spilled -- across it; see kalman_nt_kernel / crc_iterate)
      int lane = lane_kernel;
      asm volatile("" : "+v"(lane));
      const int lr = lane >> 3, lc = lane & 7, fo = lane >> 3, fq = lane & 7;
      (void)lr; (void)lc; (void)fo; (void)fq;
      double* sg = st + (size_t)t * STEP;
      const double yt = yt_next;
      yt_next = (lane < p && t + 1 < T_len) ? y[(size_t)(t + 1) * p + lane] : 0.0;
      const unsigned long long omask = load_mask(yt);
      const double lam = (omask != 0ull) ? 1.0 : 0.0;
      const bool light = steady && (omask == smask);
      if (light) {
        // ==== steady steps in registers (as in kalman_sel_kernel): this lane's row of Tc and of K, rows of F^-1 in lanes
        // 0..7, the state exchanged by v_readlane; per step one global store of a_t and no LDS traffic.  The loop runs
        // until the missing-data mask changes; the reverse sweep recomputes v, F^-1 v and a+ of these steps from a_t.
        double trow[NP], krow[8], firow[8];
#pragma unroll
        for (int kk = 0; kk < NP; ++kk) trow[kk] = (lane < u) ? Tc[lane * LDM + kk] : 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          krow[q] = (lane < u) ? Kp[lane * PS + q] : 0.0;
          firow[q] = (lane < 8) ? Fi[lane * 8 + q] : 0.0;
        }
        double a_reg = (lane < NP) ? av[lane] : 0.0;
        const double w_l = (lane < p && ((smask >> lane) & 1ull)) ? 1.0 : 0.0;
        const double v_dd = (lane < p && (w_l != 0.0 || !cv.mask_d)) ? dd[lane] : 0.0, v_zv = (lane < p) ? zv[lane] : 0.0;  // (constant mask)
        const int v_zpos = (lane < p) ? zpos[lane] : 0;
        double yc = yt;
        long long n_ss = 0;
        for (;;) {
          double* sgs = st + (size_t)t * STEP;
          if (lane < NP) sgs[OFF_A + lane] = a_reg;
          if (lane == 0) sgs[OFF_SRC] = (double)seg_src;
          const double a_sel = __shfl(a_reg, v_zpos, 64);
          const double v_s = (lane < p) ? w_l * yt_or_zero(yc) - (v_dd + w_l * v_zv * a_sel) : 0.0;
          double vsc[8];
#pragma unroll
          for (int o = 0; o < 8; ++o) vsc[o] = readlane_f64(v_s, o);
          double w0 = 0.0, w1 = 0.0, a0 = a_reg, a1 = 0.0;
#pragma unroll
          for (int o = 0; o < 8; o += 2) {
            w0 = fma(firow[o], vsc[o], w0);
            w1 = fma(firow[o + 1], vsc[o + 1], w1);
            a0 = fma(krow[o], vsc[o], a0);
            a1 = fma(krow[o + 1], vsc[o + 1], a1);
          }
          double part = v_s * (w0 + w1);  // lanes >= 8: firow = 0
          part += dpp_move_f64<0x111, 0xf>(part);
          part += dpp_move_f64<0x112, 0xf>(part);
          part += dpp_move_f64<0x114, 0xf>(part);
          const double quad_s = readlane_f64(part, 7);
          ll_acc += lam * (seg_logdet + quad_s);
          n_ll += (lam != 0.0);
          n_entries += __popcll(smask);
          const double apl = a0 + a1;
          double s0 = 0.0, s1 = 0.0;
#pragma unroll
          for (int kk = 0; kk < NP; kk += 2) {
            s0 = fma(trow[kk], readlane_f64(apl, kk), s0);
            s1 = fma(trow[kk + 1], readlane_f64(apl, kk + 1), s1);
          }
          a_reg = (lane < u) ? s0 + s1 : 0.0;
          ++n_ss;
          if (t + 1 >= T_len) break;
          const bool obs_n = (lane < p) && (yt_next == yt_next) && (yt_next != missing_fill);
          if (__ballot(obs_n) != smask) break;
          ++t;
          yc = yt_next;
          yt_next = (lane < p && t + 1 < T_len) ? y[(size_t)(t + 1) * p + lane] : 0.0;
        }
        if (lane < NP) av[lane] = a_reg;
        wave_sync();
        if (tm) {
          const long long tk1 = clock64();
          ph[2] += tk1 - tk0;
          ph[7] += n_ss;
          tk0 = tk1;
        }
        continue;
      }
      if (lane < NP) sg[OFF_A + lane] = av[lane];
      if (!light) {
        steady = false;
        if (lane == 0) sg[OFF_PREV] = (double)seg_src;  // (-1 for the first step) lets the reverse sweep fetch that record early
        seg_src = t;
        if (t == 0) {  // P_0, for the adjoint of the stationary initial covariance at the end of the reverse sweep
          double* p0s = st + (size_t)T_len * STEP;
          for (int idx = lane; idx < NP * NP; idx += 64) p0s[idx] = Ps[(idx / NP) * LDM + (idx % NP)];
        }
        seg_logdet = update_cov(sg);  // (stores P+ into the record)
        for (int idx = lane; idx < NP * 8; idx += 64) sg[OFF_K + idx] = Kp[(idx >> 3) * PS + (idx & 7)];
        sg[OFF_FI + lane] = Fi[lane];
        sg[OFF_F + lane] = Fs[lane];
      }
      long long tq = tm ? clock64() : 0;
      if (tm) qh[0] += tq - tk0;
      if (lane == 0) sg[OFF_SRC] = (double)seg_src;
      const double quad = update_mean(yt);
      ll_acc += lam * (seg_logdet + quad);
      n_ll += (lam != 0.0);
      n_entries += __popcll(omask);
      if (tm) {
        const long long t_ = clock64();
        qh[1] += t_ - tq;
        tq = t_;
      }
      // predict: a = T a+
      if (lane < u) {
        t1[lane] = kg_dot4(Tc + lane * LDM, 1, ap, 1, u);
      }
      if (!light) {  // P = sym(T P+ T') + G, with the steady-state test against the outgoing P_t
        kg_mm<BS, true>(X2, X1, Tc, u, 1.0, false, lr, lc);  // P+ T'
        wave_sync();
        double xb[BS][BS], xt[BS][BS];
        blk_zero<BS>(xb);
        mm_acc<BS, false>(xb, Tc, LDM, X2, LDM, u, lr, lc);  // T P+ T'
        const int src = (lc << 3) | lr;
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) xt[i][j] = __shfl(xb[j][i], src, 64);
        double dmax = 0.0, pmax = 0.0;
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) {
            double* pp = &Ps[(lr * BS + i) * LDM + lc * BS + j];
            const double nv = 0.5 * (xb[i][j] + xt[i][j]) + Gs[(lr * BS + i) * LDM + lc * BS + j];
            dmax = nanmax(dmax, fabs(nv - *pp));
            pmax = nanmax(pmax, fabs(nv));
            *pp = nv;
          }
        dmax = wave_nanmax(dmax);
        pmax = wave_nanmax(pmax);
        if (steady_tol > 0.0 && dmax <= steady_tol * pmax) {  // P_{t+1} = P_t to rounding: later steps reuse step t's update
          steady = true;
          smask = omask;
        }
      }
      wave_sync();
      if (lane < NP) av[lane] = (lane < u) ? t1[lane] : 0.0;
      wave_sync();
      if (tm) qh[2] += clock64() - tq;
      if (tm) {
        const long long tk1 = clock64();
        ph[light ? 2 : 1] += tk1 - tk0;
        ph[light ? 7 : 6] += 1;
        tk0 = tk1;
      }
    }
    const double logp = -0.5 * (cv.ll_terms(n_ll, n_entries, p) * LN2PI + ll_acc);
    if (lane == 0) {
      logp_out[draw] = logp;
      if (!((logp == logp) && (fabs(logp) < 1.797e308))) status[draw] |= DSGE_ST_FILTER_NONFINITE;
    }
    }  // (!SPLIT: the forward sweep)

    // ---- reverse sweep ------------------------------------------------------------------------------
    wave_sync();
    for (int idx = lane; idx < NP * LDM; idx += 64) Pb[idx] = 0.0;  // (Gs aliases Pb: the forward sweep is over)
    double TbR[BS][BS], GbR[BS][BS];  // cotangents of T and G, accumulated in register blocks
    blk_zero<BS>(TbR);
    blk_zero<BS>(GbR);
    if (lane < NP) ab[lane] = 0.0;
    if (lane < 8) {
      db[lane] = 0.0;
      hb[lane] = 0.0;
    }
    wave_sync();
    int cur_src = -1;  // step whose covariance-side quantities (Mp, Fs, Fi, Kp, X1 = P+) are in LDS
    // which observation selects this lane's variable: p dependent LDS reads -- once per draw, not once per segment (a never-steady
    // draw has 200 segments of one step)
    int my_o_draw = -1;
    double my_zv_draw = 0.0;
    for (int o = 0; o < p; ++o)
      if (zpos[o] == lane_kernel) {
        my_o_draw = o;
        my_zv_draw = zv[o];
      }
    // and the observation (if any) that selects row (lane + 64 k) >> 3 of the panels (the reverse step's Mbar += Zm' Fbar)
    int orow_draw[BS];
    double ozv_draw[BS];
#pragma unroll
    for (int k2 = 0; k2 < BS; ++k2) {
      orow_draw[k2] = -1;
      ozv_draw[k2] = 0.0;
      for (int o = 0; o < p; ++o)
        if (zpos[o] == ((lane_kernel + 64 * k2) >> 3)) {
          orow_draw[k2] = o;
          ozv_draw[k2] = zv[o];
        }
    }
    // likewise the draw's constants of the mean side: d, z and the selected position of observation `lane`, and column `lane` of Tc
    // (48 registers across the whole sweep: the kernel runs one wavefront per SIMD, they are free)
    const double dd_draw = (lane_kernel < p) ? dd[lane_kernel] : 0.0, zv_draw = (lane_kernel < p) ? zv[lane_kernel] : 0.0;
    const int zpos_draw = (lane_kernel < p) ? zpos[lane_kernel] : 0;
    double tcol[NP];
#pragma unroll
    for (int kk = 0; kk < NP; ++kk) tcol[kk] = (lane_kernel < u && kk < u) ? Tc[kk * LDM + lane_kernel] : 0.0;
    // selector entries of the observations this lane meets in the panel algebra of the covariance side: row lane >> 3, column lane & 7
    const int zpos_f = zpos[lane_kernel >> 3], zpos_q = zpos[lane_kernel & 7];
    const double zv_f = zv[lane_kernel >> 3], zv_q = zv[lane_kernel & 7];
    // Cotangents that the steps of one steady segment send to the quantities they SHARE (the forward sweep freezes K, F^-1
    // and ln det F of the segment's source step, so this is the exact reverse of what was executed): they are collected in
    // registers and pulled back through the covariance update once, at the source step.  A steady step therefore costs two
    // matrix-vector products instead of the three u x u products of a full reverse step.
    //   Kacc  = sum_t a+bar_t v_t'           (u x 8 panel: element (i, o) = idx = lane + 64 k -> register k)
    //   Qacc  = sum_t lam_t (F^-1 v_t)(F^-1 v_t)'   (8 x 8: lane = fo * 8 + fq)
    //   nlam  = sum_t lam_t                  (number of log-determinants the segment contributed)
    double Kacc[BS], Qacc = 0.0, nlam = 0.0;
#pragma unroll
    for (int k2 = 0; k2 < BS; ++k2) Kacc[k2] = 0.0;
    // the record of the NEXT segment's source step (P+, K, F^-1, F: NPF + NKF + 2 doubles per lane) is fetched while the
    // current segment is processed: it was written ~1e5 cycles ago and comes from HBM
    constexpr int NPF = (NP * NP + 63) / 64, NKF = (NP * 8 + 63) / 64;
    double pf_p[NPF], pf_k[NKF], pf_fi = 0.0, pf_f = 0.0;
    int pf_src = -2;
#pragma unroll
    for (int k2 = 0; k2 < NPF; ++k2) pf_p[k2] = 0.0;
#pragma unroll
    for (int k2 = 0; k2 < NKF; ++k2) pf_k[k2] = 0.0;
    // how the forward sweep left P+ of its full steps (KgRec::TS_LAYOUT): register blocks, or -- kalman_mf_kernel<.., REC> -- the
    // upper 4 x 4 tiles of a (2 BS - 1)-tile matrix, element g of lane l at [g 64 + l]: scattered into X1 with its mirror image
    constexpr int TMF = 2 * BS - 1, NGF = Mfma4Upper<TMF>::NG;
    bool rec_tiles = false;
    int x1a[NGF], x1b[NGF];
    if constexpr (SPLIT && KG_MF) {
      rec_tiles = (int)st[RC::tail_state_off(T_len) + RC::TS_LAYOUT] == RC::LAYOUT_TILES + TMF;  // (wave-uniform)
      constexpr int X1_DUMP = (NP - 1) * LDM + LDM - 1;
#pragma unroll
      for (int g = 0; g < NGF; ++g) {
        int ta, tb;
        bool live;
        Mfma4Upper<TMF>::tile(g, (lane_kernel >> 2) & 3, ta, tb, live);
        const int r = 4 * ta + (lane_kernel >> 4), c = 4 * tb + (lane_kernel & 3);
        const bool in = live && r <= c && c < u;
        x1a[g] = in ? r * LDM + c : X1_DUMP;
        x1b[g] = in ? c * LDM + r : X1_DUMP;
      }
    }
    double src_next = 0.0, av_next = 0.0, yr_next = 0.0;
    int t_first = T_len - 1;
    bool resumed = false;  // the steady tail's mean side came from kalman_grad_tail_kernel: this sweep starts AT its source step
    if constexpr (SPLIT) {
      if (tail_valid) {
        const double* ts = st + RC::tail_state_off(T_len);
        const int n_done = (int)ts[0];
        if (n_done > 0) {  // (wave-uniform)
          resumed = true;
          t_first = T_len - 1 - n_done;
          nlam = ts[1];
          if (lane < NP) ab[lane] = ts[RC::TS_AB + lane];
          if (lane < 8) db[lane] = ts[RC::TS_DB + lane];
#pragma unroll
          for (int i = 0; i < BS; ++i)
#pragma unroll
            for (int j = 0; j < BS; ++j) TbR[i][j] = ts[RC::TS_TB + (size_t)(i * BS + j) * 64 + lane];
#pragma unroll
          for (int k2 = 0; k2 < BS; ++k2) Kacc[k2] = ts[RC::TS_KA + (size_t)k2 * 64 + lane];
          Qacc = ts[RC::TS_QA + lane];
          wave_sync();
        }
      }
    }
    if (T_len > 0) {
      const double* sg0 = st + (size_t)t_first * STEP;
      src_next = sg0[OFF_SRC];
      if (lane < NP) av_next = sg0[OFF_A + lane];
      if (lane < p) yr_next = y[(size_t)t_first * p + lane];
    }
    for (int t = t_first; t >= 0; --t) {
      int lane = lane_kernel;
      asm volatile("" : "+v"(lane));
      const int lr = lane >> 3, lc = lane & 7, fo = lane >> 3, fq = lane & 7;
      (void)lr; (void)lc; (void)fo; (void)fq;
      int src_t = (int)src_next;
      double a_cur = av_next;
      if (lane < NP) av[lane] = a_cur;
      double yt = yr_next;
      if (t > 0) {
        const double* sgp = st + (size_t)(t - 1) * STEP;
        src_next = sgp[OFF_SRC];
        if (lane < NP) av_next = sgp[OFF_A + lane];
        yr_next = (lane < p) ? y[(size_t)(t - 1) * p + lane] : 0.0;
      }
      const unsigned long long omask = load_mask(yt);
      const double lam = (omask != 0ull) ? 1.0 : 0.0;
      if (src_t != cur_src) {  // a full step, or the first (last in time) step of a steady segment
        const double* sp_ = st + (size_t)src_t * STEP;  // the source step's covariance update, as the forward sweep left it
        wave_sync();
        if (pf_src == src_t) {
          if (rec_tiles) {
            if constexpr (SPLIT && KG_MF) {
#pragma unroll
              for (int g = 0; g < NGF; ++g) {
                X1[x1a[g]] = pf_p[g];
                X1[x1b[g]] = pf_p[g];
              }
            }
          } else {
#pragma unroll
          for (int k2 = 0; k2 < NPF; ++k2)  // (lane-major record: entry (k2 / BS, k2 % BS) of this lane's block)
            X1[(lr * BS + k2 / BS) * LDM + lc * BS + k2 % BS] = kg_in_u(lr * BS + k2 / BS, lc * BS + k2 % BS, u) ? pf_p[k2] : 0.0;
          }
#pragma unroll
          for (int k2 = 0; k2 < NKF; ++k2) {
            const int idx = lane + 64 * k2;
            if (idx < NP * 8) Kp[(idx >> 3) * PS + (idx & 7)] = pf_k[k2];
          }
          Fi[lane] = pf_fi;
          Fs[lane] = pf_f;
        } else {
          if (rec_tiles) {
            if constexpr (SPLIT && KG_MF) {
              double tv[NGF];
#pragma unroll
              for (int g = 0; g < NGF; ++g) tv[g] = sp_[lane + 64 * g];
#pragma unroll
              for (int g = 0; g < NGF; ++g) {
                X1[x1a[g]] = tv[g];
                X1[x1b[g]] = tv[g];
              }
            }
          } else {
#pragma unroll
          for (int k2 = 0; k2 < NPF; ++k2)
            X1[(lr * BS + k2 / BS) * LDM + lc * BS + k2 % BS] = kg_in_u(lr * BS + k2 / BS, lc * BS + k2 % BS, u) ? sp_[lane + 64 * k2] : 0.0;
          }
          for (int idx = lane; idx < NP * 8; idx += 64) Kp[(idx >> 3) * PS + (idx & 7)] = sp_[OFF_K + idx];
          Fi[lane] = sp_[OFF_FI + lane];
          Fs[lane] = sp_[OFF_F + lane];
        }
        {  // start fetching the record this sweep needs next
          const int prev = (int)sp_[OFF_PREV];
          pf_src = prev;
          if (prev >= 0) {
            const double* pp_ = st + (size_t)prev * STEP;
#pragma unroll
            for (int k2 = 0; k2 < NPF; ++k2) {
              const int idx = lane + 64 * k2;
              pf_p[k2] = (idx < NP * NP) ? pp_[idx] : 0.0;
            }
#pragma unroll
            for (int k2 = 0; k2 < NKF; ++k2) {
              const int idx = lane + 64 * k2;
              pf_k[k2] = (idx < NP * 8) ? pp_[OFF_K + idx] : 0.0;
            }
            pf_fi = pp_[OFF_FI + lane];
            pf_f = pp_[OFF_F + lane];
          }
        }
        wave_sync();
        if (tm) qh[3] += clock64() - tk0;
        cur_src = src_t;
        if (!resumed) {  // (a resumed segment arrives with the sums of its steady steps)
#pragma unroll
          for (int k2 = 0; k2 < BS; ++k2) Kacc[k2] = 0.0;
          Qacc = 0.0;
          nlam = 0.0;
        }
        resumed = false;
      }
      {
        // ==== the mean side of every step of the segment (its source step included) in registers: column `lane` of Tc (a+bar = T' abar), row `lane` of K (a+), rows of
        // F^-1 and columns of K in lanes 0..7 (F^-1 v, K' a+bar); vectors are exchanged by v_readlane / ds_bpermute, no fence.
        double krow[8], firow[8], kcol[NP];  // (tcol: column `lane` of Tc, loaded once per draw in front of the sweep)
#pragma unroll
        for (int kk = 0; kk < NP; ++kk) kcol[kk] = (lane < 8 && kk < u) ? Kp[kk * PS + lane] : 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          krow[q] = (lane < u) ? Kp[lane * PS + q] : 0.0;
          firow[q] = (lane < 8) ? Fi[lane * 8 + q] : 0.0;
        }
        const double w_l = (lane < p && ((omask >> lane) & 1ull)) ? 1.0 : 0.0;
        const bool d_live = (w_l != 0.0 || !cv.mask_d);  // d enters v on this entry (the mask is constant over the segment)
        const double v_dd = d_live ? dd_draw : 0.0, v_zv = zv_draw;
        const int v_zpos = zpos_draw;
        const int my_o = my_o_draw;  // the observation (if any) that selects this lane's variable (a constant of the draw)
        const double my_wz = (my_o >= 0) ? (((omask >> my_o) & 1ull) ? 1.0 : 0.0) * my_zv_draw : 0.0;
        const int my_os = (my_o >= 0) ? my_o : 0;
        double ab_reg = (lane < NP) ? ab[lane] : 0.0, db_reg = 0.0;
        // one steady step of the reverse sweep from the stored a_t (a_in) and y_t (y_in)
        auto steady_step = [&](const double a_in, const double y_in) __attribute__((always_inline)) {
          // v, F^-1 v, a+ of step t from the stored a_t
          const double a_sel = __shfl(a_in, v_zpos, 64);
          const double v_s = (lane < p) ? w_l * yt_or_zero(y_in) - (v_dd + w_l * v_zv * a_sel) : 0.0;
          double vsc[8];
#pragma unroll
          for (int o = 0; o < 8; ++o) vsc[o] = readlane_f64(v_s, o);
          double w0 = 0.0, w1 = 0.0, a0 = (lane < u) ? a_in : 0.0, a1 = 0.0;
#pragma unroll
          for (int o = 0; o < 8; o += 2) {
            w0 = fma(firow[o], vsc[o], w0);
            w1 = fma(firow[o + 1], vsc[o + 1], w1);
            a0 = fma(krow[o], vsc[o], a0);
            a1 = fma(krow[o + 1], vsc[o + 1], a1);
          }
          const double fiv_l = (lane < p) ? w0 + w1 : 0.0;  // F^-1 v (lanes 0..7)
          const double ap_l = a0 + a1;                       // a+ (lanes < u)
          // a+bar = T' abar
          double s0 = 0.0, s1 = 0.0;
#pragma unroll
          for (int kk = 0; kk < NP; kk += 2) {
            s0 = fma(tcol[kk], readlane_f64(ab_reg, kk), s0);
            s1 = fma(tcol[kk + 1], readlane_f64(ab_reg, kk + 1), s1);
          }
          const double apb_l = (lane < u) ? s0 + s1 : 0.0;
          // Tbar += abar a+'
#pragma unroll
          for (int i = 0; i < BS; ++i) {
            const double abi = __shfl(ab_reg, lr * BS + i, 64);
#pragma unroll
            for (int j = 0; j < BS; ++j) TbR[i][j] = fma(abi, __shfl(ap_l, lc * BS + j, 64), TbR[i][j]);
          }
          // vbar = -lam F^-1 v + K' a+bar (lanes 0..7)
          double q0 = -lam * fiv_l, q1 = 0.0;
#pragma unroll
          for (int kk = 0; kk < NP; kk += 2) {
            q0 = fma(kcol[kk], readlane_f64(apb_l, kk), q0);
            q1 = fma(kcol[kk + 1], readlane_f64(apb_l, kk + 1), q1);
          }
          const double vb_l = (lane < p) ? q0 + q1 : 0.0;
          // shared-quantity cotangents
          const double v_mine = __shfl(v_s, lane & 7, 64);
#pragma unroll
          for (int k2 = 0; k2 < BS; ++k2) {
            const int i = (lane + 64 * k2) >> 3;
            const double api = __shfl(apb_l, i & 63, 64);
            if (i < u) Kacc[k2] = fma(api, v_mine, Kacc[k2]);
          }
          Qacc = fma(lam * __shfl(fiv_l, fo, 64), __shfl(fiv_l, fq, 64), Qacc);
          nlam += lam;
          // abar = a+bar - Zm' vbar;  dbar -= vbar
          ab_reg = apb_l - my_wz * __shfl(vb_l, my_os, 64);
          db_reg -= d_live ? vb_l : 0.0;
        };
        // The steps' records (a_t, y_t) come from HBM.  Fetching them one step ahead into registers (`cur = next; next =
        // load`) does not work: the compiler loads into a temporary, copies it into the loop-carried register right away and
        // waits for the load it has just issued (s_waitcnt vmcnt(0) in every step); two or three register sets taking
        // turns end in the same copies at the back edge.  So a steady segment stages CH steps at a time through LDS (X2 is
        // scratch, dead here): the NEXT chunk is in flight during the CH steps of the current one, its registers are
        // defined and consumed in the same trip of the chunk loop, and a step reads its a_t / y_t from LDS -- no global
        // load and no vmcnt wait inside the step loop (measured: 4.6 k -> 3.9 k cycles per step; the rest is the step's own
        // 112 v_readlane + 28 ds_bpermute + ~90 FP64 instructions in one dependent chain).  Every step of a segment
        // carries the SAME source index (the forward sweep wrote seg_src into all of them), so the loop ends at
        // t == src_t; the step before the source step belongs to the previous segment, whose source the source record
        // names (OFF_PREV, in pf_src since the segment's first step).  Loads are unconditional and branch-free (clamped
        // indices; unused lanes are never read): a load under a condition leaves the number of loads in flight unknown to
        // the compiler, which then waits for all of them.
        const int lane_a = (lane < NP) ? lane : NP - 1, lane_y = (lane < p) ? lane : (p > 0 ? p - 1 : 0);
        {
          const bool single = (t == src_t);  // a full step: its record came with the outer loop's own fetch; (av_next, yr_next) stay valid
          constexpr int CH = (BS == 1) ? 4 : 8, NCA = (CH * NP + 63) / 64;
          double* SA = X2;            // [CH][NP]
          double* SY = X2 + CH * NP;  // [CH][8]
          double ra[NCA], ry;
          auto chunk_load = [&](int t_hi) __attribute__((always_inline)) {
#pragma unroll
            for (int k2 = 0; k2 < NCA; ++k2) {
              const int idx = (lane + 64 * k2 < CH * NP) ? lane + 64 * k2 : CH * NP - 1;
              const int jj = idx / NP, el = idx - jj * NP;
              const int tt = (t_hi - jj > 0) ? t_hi - jj : 0;
              ra[k2] = st[(size_t)tt * STEP + OFF_A + el];
            }
            const int jy = ((lane >> 3) < CH) ? (lane >> 3) : CH - 1, oy = ((lane & 7) < p) ? (lane & 7) : (p > 0 ? p - 1 : 0);
            const int ty = (t_hi - jy > 0) ? t_hi - jy : 0;
            ry = y[(size_t)ty * p + oy];
          };
          auto chunk_store = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int k2 = 0; k2 < NCA; ++k2)
              if (lane + 64 * k2 < CH * NP) SA[lane + 64 * k2] = ra[k2];
            if (lane < CH * 8) SY[lane] = ry;
          };
          if (!single) {
            chunk_load(t);
            chunk_store();
          }
          for (;;) {
            if (!single) chunk_load(t - CH);
            bool done = false;
#pragma unroll 1
            for (int j = 0; j < CH; ++j) {
              const double a_in = single ? a_cur : SA[j * NP + lane_a], y_in = single ? yt : SY[j * 8 + (lane & 7)];
              steady_step(a_in, y_in);
              if (t == src_t) {
                done = true;
                break;
              }
              --t;
            }
            if (done) break;
            chunk_store();
          }
          if (!single) {  // the step before the source step, for the outer loop
            const int tp = (t > 0) ? t - 1 : 0;
            av_next = st[(size_t)tp * STEP + OFF_A + lane_a];
            yr_next = y[(size_t)tp * p + lane_y];
          }
        }
        src_next = (double)pf_src;
        // t == src_t now: the cotangent of a_t goes back to LDS for the next segment (or the end of the sweep)
        if (lane < NP) ab[lane] = ab_reg;
        if (lane < 8) db[lane] += db_reg;
        wave_sync();
        if (tm) {
          const long long tk1 = clock64();
          ph[4] += tk1 - tk0;
          tk0 = tk1;
        }
      }
      long long tr = tm ? clock64() : 0;
      if (t == src_t) {
        // ---- covariance side, once per segment: Pb is the cotangent of the predicted covariance P_{src+1}, which the
        // steady steps after src never touched; X1 = P+ of the source step.
        // (round 5, measured: the three products with four k-steps per stage and two stages in flight -- 8 (BS + BS) operand
        //  registers, free at one wavefront per SIMD -- make the launch SLOWER, 2.62 -> 2.75 ms: not used)
        if constexpr (KG_MF) {
          {
            double pb[BS][BS];
            blk_load_lds<BS>(pb, Pb, LDM, lr, lc);
#pragma unroll
            for (int i = 0; i < BS; ++i)
#pragma unroll
              for (int j = 0; j < BS; ++j) GbR[i][j] += pb[i][j];  // Gbar += Pbar
          }
          wave_sync();  // (Pbar is overwritten below)
          if (u <= 4 * (2 * BS - 1))
            kg_cov_products_mf<BS, 2 * BS - 1>(Pb, Tc, X1, X2, Ps, u, lane);
          else
            kg_cov_products_mf<BS, 2 * BS>(Pb, Tc, X1, X2, Ps, u, lane);
          {
            double t2[BS][BS];
            blk_load_lds<BS>(t2, Pb, LDM, lr, lc);
#pragma unroll
            for (int i = 0; i < BS; ++i)
#pragma unroll
              for (int j = 0; j < BS; ++j) TbR[i][j] = fma(2.0, t2[i][j], TbR[i][j]);  // Tbar += 2 Pbar T P+
          }
        } else {
          kg_mm<BS, false>(X2, Pb, Tc, u, 1.0, false, lr, lc);   // Pbar T
          wave_sync();
          {
            double pb[BS][BS], t2[BS][BS], pp[BS][BS];
            blk_load_lds<BS>(pb, Pb, LDM, lr, lc);
            blk_zero<BS>(t2);
            blk_zero<BS>(pp);
            mm_acc<BS, false>(t2, X2, LDM, X1, LDM, u, lr, lc);  // (Pbar T) P+
            mm_acc_ta<BS>(pp, Tc, LDM, X2, LDM, u, lr, lc);      // P+bar = T' (Pbar T)
#pragma unroll
            for (int i = 0; i < BS; ++i)
#pragma unroll
              for (int j = 0; j < BS; ++j) {
                GbR[i][j] += pb[i][j];                        // Gbar += Pbar
                TbR[i][j] = fma(2.0, t2[i][j], TbR[i][j]);    // Tbar += 2 Pbar T P+
              }
            blk_store_lds<BS>(pp, Ps, LDM, lr, lc);  // Ps now holds P+bar (P_t itself is no longer needed)
          }
          wave_sync();
        }
        if (tm) {
          const long long t_ = clock64();
          qh[5] += t_ - tr;
          tr = t_;
        }
        if constexpr (KG_MF) {
          // ---- the panel algebra with the two contractions over the retained variables on the matrix core (round 6) ----
          constexpr int PDUMP = (NP - 1) * PS + PS - 1;
          const bool small = u <= 4 * (2 * BS - 1);
          {  // Y = P+bar K
            const int blk = (lane >> 2) & 3, i4 = lane & 3, kq = lane >> 4;
            auto y_sink = [&](auto mp, int g, double d) {
              using MPY = decltype(mp);
              const int at = (4 * MPY::ta(g, blk) + kq) * PS + 4 * MPY::tb(g, blk) + i4;
              Yp[MPY::live(g, blk) ? at : PDUMP] = d;
            };
            if (small)
              mfma4_strided<2 * BS - 1, 2 * BS - 1, 2, LDM, 1, PS, 1>(Ps, Kp, lane, [&](int g, double d) { y_sink(Mfma4Map<2 * BS - 1, 2>{}, g, d); });
            else
              mfma4_strided<2 * BS, 2 * BS, 2, LDM, 1, PS, 1>(Ps, Kp, lane, [&](int g, double d) { y_sink(Mfma4Map<2 * BS, 2>{}, g, d); });
          }
          wave_sync();
          // Kbar = sum_t a+bar_t v_t' - 2 Y (F + jit I);  Mbar = Kbar F^-1: row i of Kbar sits in the eight lanes of this lane's group
          // (idx = lane + 64 k: i = idx >> 3, o = idx & 7 = fq) -- exchanged by shuffles, no LDS round trip;  Kb <- Y + Mbar
          double fcol[8], ficol[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            fcol[q] = -2.0 * (Fs[q * 8 + fq] + ((q == fq) ? cv.jit_V : 0.0));
            ficol[q] = Fi[q * 8 + fq];
          }
          double yrow[BS][8];
#pragma unroll
          for (int k2 = 0; k2 < BS; ++k2)
#pragma unroll
            for (int q = 0; q < 8; ++q) yrow[k2][q] = Yp[((lane + 64 * k2) >> 3) * PS + q];
#pragma unroll
          for (int k2 = 0; k2 < BS; ++k2) {
            const int i = (lane + 64 * k2) >> 3;
            double sk = Kacc[k2];
#pragma unroll
            for (int q = 0; q < 8; ++q) sk = fma(yrow[k2][q], fcol[q], sk);
            const double kb = (i < u && fq < p) ? sk : 0.0;
            double sm0 = 0.0, sm1 = 0.0;
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
              sm0 = fma(__shfl(kb, (lane & 56) + q, 64), ficol[q], sm0);
              sm1 = fma(__shfl(kb, (lane & 56) + q + 1, 64), ficol[q + 1], sm1);
            }
            const double mb = (i < u && fq < p) ? sm0 + sm1 : 0.0;
            Mb[i * PS + fq] = mb;
            Kb[i * PS + fq] = (i < u) ? yrow[k2][fq] + mb : 0.0;
          }
          wave_sync();
          {  // K' (Y + Mbar), 8 x 8
            const int blk = (lane >> 2) & 3, i4 = lane & 3, kq = lane >> 4;
            auto f_sink = [&](int g, double d) {
              using MPF = Mfma4Map<2, 2>;
              const int at = (4 * MPF::ta(g, blk) + kq) * 8 + 4 * MPF::tb(g, blk) + i4;
              *(MPF::live(g, blk) ? Ft + at : sp) = d;  // (blocks 2, 3 recompute tile 0: into the spare vector)
            };
            if (small)
              mfma4_strided<2 * BS - 1, 2, 2, 1, PS, PS, 1>(Kp, Kb, lane, f_sink);
            else
              mfma4_strided<2 * BS, 2, 2, 1, PS, PS, 1>(Kp, Kb, lane, f_sink);
          }
          wave_sync();
          // Fbar = -1/2 (nlam F^-1 - sum_t lam_t fiv_t fiv_t') - K' (Y + Mbar)      (lane = fo*8 + fq)
          const double sf = (fo < p && fq < p) ? -0.5 * (nlam * Fi[lane] - Qacc) - Ft[lane] : 0.0;
          Fb[lane] = sf;
          const double w_q = ((omask >> fq) & 1ull) ? 1.0 : 0.0;
          {  // hbar += w o diag(Fbar)   (lane < 8: fq = lane)
            const double fdiag = __shfl(sf, (lane * 9) & 63, 64);
            if (lane < 8 && lane < p) hb[lane] = fma(w_q, fdiag, hb[lane]);
          }
          wave_sync();
          // Mbar += Zm' Fbar (row zpos[o] gets w zv Fbar[o,:]) folded into  Pbar = sym(P+bar + Mbar Zm):  column zpos[o] += w zv Mbar[:,o]
#pragma unroll
          for (int k2 = 0; k2 < BS; ++k2) {
            const int i = (lane + 64 * k2) >> 3;
            const int oi = orow_draw[k2];
            const double wz_i = (oi >= 0 && ((omask >> (oi & 7)) & 1ull)) ? ozv_draw[k2] : 0.0;
            const double mbv = fma(wz_i, Fb[(oi & 7) * 8 + fq], Mb[i * PS + fq]);
            if (i < u && fq < p) Ps[i * LDM + zpos_q] = fma(w_q * zv_q, mbv, Ps[i * LDM + zpos_q]);
          }
          wave_sync();
        } else {
          for (int idx = lane; idx < u * 8; idx += 64) {  // Y = P+bar K
            const int i = idx >> 3, o = idx & 7;
            Yp[i * PS + o] = kg_dot4(Ps + i * LDM, 1, Kp + o, PS, u);
          }
          wave_sync();
          // (idx = lane + 64 k: the column o = idx & 7 = fq is the same in every trip -- column fq of F + jit I and of F^-1 are read
          //  once, with compile-time loop bounds)
          double fcol[8], ficol[8];
  #pragma unroll
          for (int q = 0; q < 8; ++q) {
            fcol[q] = -2.0 * (Fs[q * 8 + fq] + ((q == fq) ? cv.jit_V : 0.0));
            ficol[q] = Fi[q * 8 + fq];
          }
  #pragma unroll
          for (int k2 = 0; k2 < BS; ++k2) {  // Kbar = sum_t a+bar_t v_t' - 2 Y (F + jit I)
            const int idx = lane + 64 * k2, i = idx >> 3;
            if (i < u) {
              double sk = Kacc[k2];
  #pragma unroll
              for (int q = 0; q < 8; ++q) sk = fma(Yp[i * PS + q], fcol[q], sk);
              Kb[i * PS + fq] = (fq < p) ? sk : 0.0;
            }
          }
          wave_sync();
          for (int idx = lane; idx < u * 8; idx += 64) {  // Mbar = Kbar F^-1
            const int i = idx >> 3;
            double sm = 0.0;
  #pragma unroll
            for (int q = 0; q < 8; ++q) sm = fma(Kb[i * PS + q], ficol[q], sm);
            Mb[i * PS + fq] = (fq < p) ? sm : 0.0;
          }
          wave_sync();
          {  // Fbar = -1/2 (nlam F^-1 - sum_t lam_t fiv_t fiv_t') - K' Y - K' Mbar      (lane = fo*8 + fq)
            double sf = 0.0;
            if (fo < p && fq < p) {
              // (over all NP rows with a compile-time bound: rows >= u of K are zero, of Y and Mbar zero or stale but finite;
              // four accumulators, the 72 LDS reads requested together -- the loop over u waited for three reads per term)
              double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
              constexpr int FB_UNROLL = BS <= 4 ? NP / 4 : 2;
  #pragma unroll FB_UNROLL
              for (int i = 0; i < NP; i += 4) {
                s0 = fma(-Kp[i * PS + fo], Yp[i * PS + fq] + Mb[i * PS + fq], s0);
                s1 = fma(-Kp[(i + 1) * PS + fo], Yp[(i + 1) * PS + fq] + Mb[(i + 1) * PS + fq], s1);
                s2 = fma(-Kp[(i + 2) * PS + fo], Yp[(i + 2) * PS + fq] + Mb[(i + 2) * PS + fq], s2);
                s3 = fma(-Kp[(i + 3) * PS + fo], Yp[(i + 3) * PS + fq] + Mb[(i + 3) * PS + fq], s3);
              }
              sf = -0.5 * (nlam * Fi[lane] - Qacc) + ((s0 + s1) + (s2 + s3));
            }
            Fb[lane] = sf;
          }
          wave_sync();
          // (mask weights from the ballot, selector entries from the draw's registers: no dependent LDS reads of the 8-vectors)
          const double w_f = ((omask >> fo) & 1ull) ? 1.0 : 0.0, w_q = ((omask >> fq) & 1ull) ? 1.0 : 0.0;
          if (lane < 8 && lane < p) hb[lane] = fma(w_q, Fb[lane * 9], hb[lane]);  // hbar += w o diag(Fbar)   (lane < 8: fq = lane)
          if (fo < p && fq < p) {  // Mbar += Zm' Fbar: row zpos[fo] (distinct per fo) gets w zv Fbar[fo,:]
            Mb[zpos_f * PS + fq] = fma(w_f * zv_f, Fb[lane], Mb[zpos_f * PS + fq]);
          }
          wave_sync();
          // Pbar = sym(P+bar + Mbar Zm):  column zpos[o] += w zv Mbar[:,o]   (idx = lane + 64 k: o = idx & 7 = fq in every trip)
          for (int idx = lane; idx < u * 8; idx += 64) {
            const int i = idx >> 3;
            if (fq < p) Ps[i * LDM + zpos_q] = fma(w_q * zv_q, Mb[i * PS + fq], Ps[i * LDM + zpos_q]);
          }
          wave_sync();
        }
        if (tm) {
          const long long t_ = clock64();
          qh[6] += t_ - tr;
          tr = t_;
        }
        {  // (register blocks: the entry-per-lane loop divided by u and read two dependent entries per trip)
          double a[BS][BS], b[BS][BS];
          blk_load_lds<BS>(a, Ps, LDM, lr, lc);
          blk_load_lds_t<BS>(b, Ps, LDM, lr, lc);
#pragma unroll
          for (int i = 0; i < BS; ++i)
#pragma unroll
            for (int j = 0; j < BS; ++j) {
              const int r = lr * BS + i, c = lc * BS + j;
              if (r < u && c < u) Pb[r * LDM + c] = 0.5 * (a[i][j] + b[i][j]);
            }
        }
      }
      wave_sync();
      if (tm) {
        const long long tk1 = clock64();
        qh[7] += tk1 - tr;
        ph[3] += tk1 - tk0;
        tk0 = tk1;
      }
    }
    // ---- initial covariance: S = dlyap(T', Pbar_0) by doubling; Gbar += S; Tbar += 2 S T P0 ---------
    // S in Pb, A = T^(2^k) in X1
    for (int idx = lane; idx < NP * LDM; idx += 64) X1[idx] = Tc[idx];
    wave_sync();
    for (int itl = 0; itl < 64; ++itl) {
      if constexpr (KG_MF) {
        double dmax, smax;
        if (u <= 4 * (2 * BS - 1))
          kg_dlyap_step_mf<BS, 2 * BS - 1>(Pb, X1, X2, u, lane, dmax, smax);
        else
          kg_dlyap_step_mf<BS, 2 * BS>(Pb, X1, X2, u, lane, dmax, smax);
        if (!(dmax == dmax) || dmax <= 1e-17 * smax || smax == 0.0) break;
        continue;
      }
      kg_mm<BS, false>(X2, Pb, X1, u, 1.0, false, lr, lc);  // S A
      wave_sync();
      double inc[BS][BS], a2[BS][BS];
      blk_zero<BS>(inc);
      blk_zero<BS>(a2);
      mm_acc_ta<BS>(inc, X1, LDM, X2, LDM, u, lr, lc);      // A' S A
      mm_acc<BS, false>(a2, X1, LDM, X1, LDM, u, lr, lc);
      wave_sync();
      double dmax = 0.0, smax = 0.0;
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) {
          const int e = (lr * BS + i) * LDM + lc * BS + j;
          Pb[e] += inc[i][j];
          dmax = nanmax(dmax, fabs(inc[i][j]));
          smax = nanmax(smax, fabs(Pb[e]));
          X1[e] = a2[i][j];
        }
      dmax = wave_nanmax(dmax);
      smax = wave_nanmax(smax);
      wave_sync();
      if (!(dmax == dmax) || dmax <= 1e-17 * smax || smax == 0.0) break;
    }
    for (int idx = lane; idx < NP * NP; idx += 64) Ps[(idx / NP) * LDM + (idx % NP)] = st[(size_t)T_len * STEP + idx];  // P0
    wave_sync();
    kg_mm<BS, false>(X2, Tc, Ps, u, 1.0, false, lr, lc);  // T P0
    wave_sync();
    {
      double sb[BS][BS], t2[BS][BS];
      blk_load_lds<BS>(sb, Pb, LDM, lr, lc);
      blk_zero<BS>(t2);
      mm_acc<BS, false>(t2, Pb, LDM, X2, LDM, u, lr, lc);  // S T P0
      // ---- scatter to the caller's variable order --------------------------------------------------
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) {
          const int r = lr * BS + i, c = lc * BS + j;
          if (r < u && c < u) {
            const size_t g = (size_t)perm[r] * m_full + perm[c];
            if (c < s) Tbo[g] = fma(2.0, t2[i][j], TbR[i][j]);  // Tbar += 2 S T P0
            Gbo[g] = GbR[i][j] + sb[i][j];                      // Gbar += S
          }
        }
    }
    if (lane < p) {
      if (dbar_out) dbar_out[(size_t)draw * p + lane] = db[lane];
      if (hbar_out) hbar_out[(size_t)draw * p + lane] = hb[lane];
    }
    if (tm) {
      ph[5] += clock64() - tk0;
      if (lane == 0)
        for (int k2 = 0; k2 < 8; ++k2) {
          dbg[k2] = ph[k2];
          dbg[8 + k2] = qh[k2];
        }
    }
  }
}


// ---- the reverse MEAN side of the last steady segment, as a lean kernel of its own (round 5) ---------------------------------------
// In the reverse sweep of kalman_grad_kernel<BS, true> half of a typical draw's time is the mean side of its steady tail -- 167 of
// 200 steps on the SW-shaped workload, two matrix-vector products and three outer-product accumulations each -- at ONE wavefront per
// SIMD (the kernel holds 308 registers and 36 KB of LDS for the covariance side).  That part needs none of the matrices: column
// `lane` of T, row / column `lane` of K, rows of F^-1 in registers, a_t and y_t staged through 2.5 KB of LDS.  At the END of the
// sample the cotangent of the state is zero, so the last segment's steady steps depend on nothing the covariance side computes:
// this kernel runs them (t = T_len - 1 down to src + 1, src = the segment's source step) at two wavefronts per SIMD and hands
// abar, dbar and the accumulators (Tbar, Kacc, Qacc, nlam) to the reverse sweep, which resumes AT the source step.
// Same arithmetic as the steady_step of the reverse sweep, operation by operation.
template <int BS>
__global__ __launch_bounds__(64, KG_TAIL_WAVES) void kalman_grad_tail_kernel(
    const double* __restrict__ T, const double* __restrict__ Z, int z_batched, const double* __restrict__ dvec, int d_batched,
    const double* __restrict__ y, int batch, int m_full, int p, int T_len, FilterConv cv, double missing_fill,
    double* __restrict__ store, const int32_t* __restrict__ status, const int32_t* __restrict__ order) {
  using RC = KgRec<BS>;
  constexpr int NP = 8 * BS;
  constexpr int CH = (BS == 1) ? 4 : 8, NCA = (CH * NP + 63) / 64;
  __shared__ double SA[CH * NP], SY[CH * 8], zv[8], dd[8];
  __shared__ int perm[NP], zpos[8];
  const int lane = threadIdx.x, lr = lane >> 3, lc = lane & 7, fo = lane >> 3, fq = lane & 7;
  for (int bi = blockIdx.x; bi < batch; bi = batch) {
    const int draw = order ? order[bi] : bi;
    double* st = store + (size_t)draw * RC::per_draw(T_len);
    double* ts = st + RC::tail_state_off(T_len);
    if (lane == 0) ts[0] = 0.0;  // "nothing processed" unless the loop below runs
    if (status[draw] != 0 || T_len < 2) continue;
    const int src = (int)st[(size_t)(T_len - 1) * RC::STEP + RC::OFF_SRC];
    if (src >= T_len - 1 || src < 0) continue;  // the sample ends with a full step: no steady tail
    const size_t off = (size_t)draw * m_full * m_full;
    // ---- reduction to U = S u O, states first (as kalman_grad_kernel) ---------------------------------------------------------
    const double* Zg = Z + (z_batched ? (size_t)draw * p * m_full : 0);
    bool is_state = false;
    {
      const double* tcol_g = T + off + (lane < m_full ? lane : 0);
      for (int r0 = 0; r0 < m_full; r0 += 8) {
        double tv[8];
#pragma unroll
        for (int u8 = 0; u8 < 8; ++u8) tv[u8] = tcol_g[(size_t)(r0 + u8 < m_full ? r0 + u8 : m_full - 1) * m_full];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u8 = 0; u8 < 8; ++u8) is_state |= (tv[u8] != 0.0);
      }
      is_state = is_state && (lane < m_full);
    }
    const unsigned long long colmask = __ballot(is_state);
    unsigned long long obsmask = 0ull;
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
    for (int o = 0; o < 8; ++o)
      if (o < p) obsmask |= __ballot(zrow[o] != 0.0);
    const unsigned long long extra = obsmask & ~colmask;
    const int s = __popcll(colmask);
    const int u = s + __popcll(extra);
    int my_pos = -1;
    wave_sync();
    if (lane < m_full) {
      const unsigned long long below = (1ull << lane) - 1ull;
      if ((colmask >> lane) & 1ull)
        my_pos = __popcll(colmask & below);
      else if ((extra >> lane) & 1ull)
        my_pos = s + __popcll(extra & below);
      if (my_pos >= 0 && my_pos < NP) perm[my_pos] = lane;
    }
#pragma unroll
    for (int o = 0; o < 8; ++o)
      if (o < p && zrow[o] != 0.0) {
        zpos[o] = (my_pos >= 0 && my_pos < NP) ? my_pos : 0;
        zv[o] = zrow[o];
      }
    if (lane < 8) dd[lane] = (dvec && lane < p) ? dvec[(d_batched ? (size_t)draw * p : 0) + lane] : 0.0;
    wave_sync();
    // ---- the segment's constants in registers: column `lane` of T[U,U], row / column `lane` of K, rows of F^-1 ------------------
    const double* rs = st + (size_t)src * RC::STEP;  // the source step's record
    double tcol[NP], krow[8], firow[8], kcol[NP];
    const int pl = (lane < u) ? perm[lane] : 0;
#pragma unroll
    for (int kk = 0; kk < NP; ++kk) {
      const int pk = (kk < u) ? perm[kk < NP ? kk : 0] : 0;
      tcol[kk] = T[off + (size_t)pk * m_full + pl];
      kcol[kk] = rs[RC::OFF_K + (size_t)kk * 8 + (lane & 7)];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < NP; ++kk) {
      tcol[kk] = (lane < u && kk < u) ? tcol[kk] : 0.0;
      kcol[kk] = (lane < 8 && kk < u) ? kcol[kk] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      krow[q] = rs[RC::OFF_K + (size_t)(lane < NP ? lane : 0) * 8 + q];
      firow[q] = rs[RC::OFF_FI + (size_t)(lane & 7) * 8 + q];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      krow[q] = (lane < u) ? krow[q] : 0.0;
      firow[q] = (lane < 8) ? firow[q] : 0.0;
    }
    // the mask of the segment (constant over it: the forward sweep ends a segment where the mask changes)
    const double y_last = (lane < p) ? y[(size_t)(T_len - 1) * p + lane] : 0.0;
    const unsigned long long omask = __ballot((lane < p) && (y_last == y_last) && (y_last != missing_fill));
    const double lam = (omask != 0ull) ? 1.0 : 0.0;
    const double w_l = (lane < p && ((omask >> lane) & 1ull)) ? 1.0 : 0.0;
    const bool d_live = (w_l != 0.0 || !cv.mask_d);
    const double v_dd = (lane < p && d_live) ? dd[lane] : 0.0, v_zv = (lane < p) ? zv[lane] : 0.0;
    const int v_zpos = (lane < p) ? zpos[lane] : 0;
    int my_o = -1;
    for (int o = 0; o < p; ++o)
      if (zpos[o] == lane) my_o = o;
    const double my_wz = (my_o >= 0) ? (((omask >> my_o) & 1ull) ? 1.0 : 0.0) * zv[my_o] : 0.0;
    const int my_os = (my_o >= 0) ? my_o : 0;
    double ab_reg = 0.0, db_reg = 0.0, TbR[BS][BS], Kacc[BS], Qacc = 0.0, nlam = 0.0;
    blk_zero<BS>(TbR);
#pragma unroll
    for (int k2 = 0; k2 < BS; ++k2) Kacc[k2] = 0.0;
    auto steady_step = [&](const double a_in, const double y_in) __attribute__((always_inline)) {
      const double a_sel = __shfl(a_in, v_zpos, 64);
      const double v_s = (lane < p) ? w_l * yt_or_zero(y_in) - (v_dd + w_l * v_zv * a_sel) : 0.0;
      double vsc[8];
#pragma unroll
      for (int o = 0; o < 8; ++o) vsc[o] = readlane_f64(v_s, o);
      double w0 = 0.0, w1 = 0.0, a0 = (lane < u) ? a_in : 0.0, a1 = 0.0;
#pragma unroll
      for (int o = 0; o < 8; o += 2) {
        w0 = fma(firow[o], vsc[o], w0);
        w1 = fma(firow[o + 1], vsc[o + 1], w1);
        a0 = fma(krow[o], vsc[o], a0);
        a1 = fma(krow[o + 1], vsc[o + 1], a1);
      }
      const double fiv_l = (lane < p) ? w0 + w1 : 0.0;
      const double ap_l = a0 + a1;
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int kk = 0; kk < NP; kk += 2) {
        s0 = fma(tcol[kk], readlane_f64(ab_reg, kk), s0);
        s1 = fma(tcol[kk + 1], readlane_f64(ab_reg, kk + 1), s1);
      }
      const double apb_l = (lane < u) ? s0 + s1 : 0.0;
#pragma unroll
      for (int i = 0; i < BS; ++i) {
        const double abi = __shfl(ab_reg, lr * BS + i, 64);
#pragma unroll
        for (int j = 0; j < BS; ++j) TbR[i][j] = fma(abi, __shfl(ap_l, lc * BS + j, 64), TbR[i][j]);
      }
      double q0 = -lam * fiv_l, q1 = 0.0;
#pragma unroll
      for (int kk = 0; kk < NP; kk += 2) {
        q0 = fma(kcol[kk], readlane_f64(apb_l, kk), q0);
        q1 = fma(kcol[kk + 1], readlane_f64(apb_l, kk + 1), q1);
      }
      const double vb_l = (lane < p) ? q0 + q1 : 0.0;
      const double v_mine = __shfl(v_s, lane & 7, 64);
#pragma unroll
      for (int k2 = 0; k2 < BS; ++k2) {
        const int i = (lane + 64 * k2) >> 3;
        const double api = __shfl(apb_l, i & 63, 64);
        if (i < u) Kacc[k2] = fma(api, v_mine, Kacc[k2]);
      }
      Qacc = fma(lam * __shfl(fiv_l, fo, 64), __shfl(fiv_l, fq, 64), Qacc);
      nlam += lam;
      ab_reg = apb_l - my_wz * __shfl(vb_l, my_os, 64);
      db_reg -= d_live ? vb_l : 0.0;
    };
    // ---- the steps t = T_len - 1 .. src + 1, a chunk of CH of them staged through LDS at a time (see the reverse sweep) ---------
    const int lane_a = (lane < NP) ? lane : NP - 1;
    int t = T_len - 1;
    double ra[NCA], ry;
    auto chunk_load = [&](int t_hi) __attribute__((always_inline)) {
#pragma unroll
      for (int k2 = 0; k2 < NCA; ++k2) {
        const int idx = (lane + 64 * k2 < CH * NP) ? lane + 64 * k2 : CH * NP - 1;
        const int jj = idx / NP, el = idx - jj * NP;
        const int tt = (t_hi - jj > 0) ? t_hi - jj : 0;
        ra[k2] = st[(size_t)tt * RC::STEP + RC::OFF_A + el];
      }
      const int jy = ((lane >> 3) < CH) ? (lane >> 3) : CH - 1, oy = ((lane & 7) < p) ? (lane & 7) : (p > 0 ? p - 1 : 0);
      const int ty = (t_hi - jy > 0) ? t_hi - jy : 0;
      ry = y[(size_t)ty * p + oy];
    };
    auto chunk_store = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int k2 = 0; k2 < NCA; ++k2)
        if (lane + 64 * k2 < CH * NP) SA[lane + 64 * k2] = ra[k2];
      if (lane < CH * 8) SY[lane] = ry;
    };
    chunk_load(t);
    chunk_store();
    wave_sync();
    for (;;) {
      chunk_load(t - CH);
      bool done = false;
#pragma unroll 1
      for (int j = 0; j < CH; ++j) {
        steady_step(SA[j * NP + lane_a], SY[j * 8 + (lane & 7)]);
        --t;
        if (t == src) {
          done = true;
          break;
        }
      }
      if (done) break;
      wave_sync();
      chunk_store();
      wave_sync();
    }
    // ---- the state for the reverse sweep, which resumes at t = src -----------------------------------------------------------------
    if (lane == 0) {
      ts[0] = (double)(T_len - 1 - src);
      ts[1] = nlam;
    }
    if (lane < NP) ts[RC::TS_AB + lane] = ab_reg;
    if (lane < 8) ts[RC::TS_DB + lane] = db_reg;
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) ts[RC::TS_TB + (size_t)(i * BS + j) * 64 + lane] = TbR[i][j];
#pragma unroll
    for (int k2 = 0; k2 < BS; ++k2) ts[RC::TS_KA + (size_t)k2 * 64 + lane] = Kacc[k2];
    ts[RC::TS_QA + lane] = Qacc;
  }
}

}  // namespace dsge

namespace dsge {

// -------------------------------------------------------------------------------------------------------
// Reverse of the state-space assembly (full model size n), between the Kalman reverse sweep and the
// policy-function adjoints:
//   G = sym(R Q R'), Q = diag(q):   Rbar = 2 Gbar R Q,   qbar_j = (R' Gbar R)_jj;   full Q:  Qbar = R' Gbar R  (k x k)
//   R = -M^-1 D,  M = B + C T  (gEconpy/solvers/shared.py:74-75):
//       X = M^-T Rbar,   Dbar = -X,   Mbar = -X R',   Bbar = Mbar,   Cbar = Mbar T',   Tbar += C' Mbar
// Tbar then goes through adjoint_kernel (shared.py:12-71), which ADDS its B, C cotangents to these.
// -------------------------------------------------------------------------------------------------------
template <int BS>
struct GaSmem {
  static constexpr int NP = Tile<BS>::NP, LD = Tile<BS>::LD, LDW = 2 * NP + 1;
  // Ts, Cs, Ms, Rs (NP x LD), W (NP x LDW); the Gauss-Jordan scratch lives in Ms, which is dead between the product
  // Gbar R and the store of Mbar: 78.4 instead of 83.4 KB at n = 40 -- two draws per CU instead of ONE
  static constexpr size_t bytes = sizeof(double) * (size_t)(4 * NP * LD + NP * LDW);
  static_assert(NP * BS + BS * 2 * NP + NP / 2 <= NP * LD, "Gauss-Jordan scratch must fit an NP x LD matrix");
};

template <int BS>
__global__ __launch_bounds__(64) void grad_assemble_kernel(
    const double* __restrict__ B, const double* __restrict__ C, const double* __restrict__ T,
    const double* __restrict__ R, const double* __restrict__ q, int q_batched, const double* __restrict__ Gbar,
    int batch, int n, int k, const int32_t* __restrict__ status, double* __restrict__ Tbar,
    double* __restrict__ B_bar, double* __restrict__ C_bar, double* __restrict__ D_bar, double* __restrict__ q_bar,
    const double* __restrict__ Rbar_in = nullptr, int only_flag = 0) {
  // only_flag (round 6): behind the fused launch (adjoint_kernel<BS, false, true>) -- only the draws that launch flagged
  // DSGE_ST_INTERNAL_RERUN are visited (the flag stays for the adjoint pass that follows); everything else is already done
  // Rbar_in != nullptr: the pullback of R = -(C T + B)^-1 D ALONE (pt_compute_selection_matrix, shared.py:74-75) --
  // the cotangent of R arrives directly, nothing is known about Q, and Tbar is WRITTEN (= C' Mbar) instead of accumulated.
  constexpr int NP = GaSmem<BS>::NP, LD = GaSmem<BS>::LD, LDW = GaSmem<BS>::LDW;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Ts = smem;
  double* Cs = Ts + NP * LD;   // C, later C'
  double* Ms = Cs + NP * LD;   // Gbar, later Mbar
  double* Rs = Ms + NP * LD;   // R (n x k, zero padded), later X
  double* W = Rs + NP * LD;    // [M' | Rbar]
  double* Lbuf = Ms;           // (Gauss-Jordan scratch: Ms is dead while the solve runs)
  double* Ybuf = Lbuf + NP * BS;
  int* prow = (int*)(Ybuf + BS * 2 * NP);
  const int lane = threadIdx.x, lr = lane >> 3, lc = lane & 7;
  for (int draw = blockIdx.x; draw < batch; draw = batch) {  // (one draw per workgroup, grid = batch)
    const size_t off = (size_t)draw * n * n, offk = (size_t)draw * n * k;
    if (only_flag) {
      if (!(status[draw] & DSGE_ST_INTERNAL_RERUN)) continue;
    } else if (status && status[draw] != 0) {  // failed draw: zero cotangents
      double z[BS][BS];
      blk_zero<BS>(z);
      blk_store_global<BS>(z, B_bar + off, n, n, n, lr, lc);
      blk_store_global<BS>(z, C_bar + off, n, n, n, lr, lc);
      blk_store_global<BS>(z, D_bar + offk, n, k, k, lr, lc);
      if (q_bar && q_batched >= 2 && !Rbar_in) {
        for (int idx = lane; idx < k * k; idx += 64) q_bar[(size_t)draw * k * k + idx] = 0.0;
      } else if (q_bar && lane < k) {
        q_bar[(size_t)draw * k + lane] = 0.0;
      }
      if (Rbar_in) blk_store_global<BS>(z, Tbar + off, n, n, n, lr, lc);
      continue;
    }
    wave_sync();
    for (int idx = lane; idx < NP * LDW; idx += 64) W[idx] = 0.0;
    lds_load_matrix(Ts, LD, NP, NP, T + off, n, n, lane);
    lds_load_matrix(Cs, LD, NP, NP, C + off, n, n, lane);
    if (!Rbar_in) lds_load_matrix(Ms, LD, NP, NP, Gbar + off, n, n, lane);
    lds_load_matrix(Rs, LD, NP, NP, R + offk, n, k, lane);
    wave_sync();
    // the ranges of non-zero columns of T (states), C (variables with a lead) and Gbar (the filter's retained variables): the
    // products below contract over them (SW-shaped: 18, 12 and 18 of 40 terms)
    int t_hi = n, c_lo = 0, c_hi = n, g_lo = 0, g_hi = n;
    {
      bool nzT = false, nzC = false, nzG = false;
      if (lane < n)
        for (int r = 0; r < n; ++r) {
          nzT = nzT | (Ts[r * LD + lane] != 0.0);
          nzC = nzC | (Cs[r * LD + lane] != 0.0);
          nzG = nzG | (Ms[r * LD + lane] != 0.0);
        }
      const unsigned long long cmT = __ballot(nzT), cmC = __ballot(nzC), cmG = __ballot(nzG);
      t_hi = cmT ? 64 - __clzll((long long)cmT) : 0;
      c_lo = cmC ? __ffsll((long long)cmC) - 1 : 0;
      c_hi = cmC ? 64 - __clzll((long long)cmC) : 0;
      if (!Rbar_in) {
        g_lo = cmG ? __ffsll((long long)cmG) - 1 : 0;
        g_hi = cmG ? 64 - __clzll((long long)cmG) : 0;
      }
    }
    // q_batched is the q_mode of include/dsge_hip.h: 0 / 1 = diag(q) shared / per draw, 2 / 3 = full k x k Q shared / per draw
    const bool qfull = !Rbar_in && q_batched >= 2;
    const double* qd = Rbar_in ? nullptr
                               : (qfull ? q + (q_batched == 3 ? (size_t)draw * k * k : 0) : q + (q_batched ? (size_t)draw * k : 0));
    // GR = Gbar R  (n x k);  Rbar = 2 GR Q;  qbar_j = sum_i R_ij GR_ij
    double GR[BS][BS], Rb[BS][BS];
    blk_zero<BS>(GR);
    double Rbar[BS][BS];
    if (Rbar_in) {
      blk_load_global<BS>(Rbar, Rbar_in + offk, n, k, k, lr, lc);
    } else {
    mm_acc<BS, false>(GR, Ms + g_lo, LD, Rs + g_lo * LD, LD, g_hi - g_lo, lr, lc);
    if (qfull) {
      // full shock covariance (full_covariance, statespace.py:247-251): G = R Q R', Q symmetric ->
      //   Rbar = 2 (Gbar R) Q,   Qbar = R' (Gbar R)   (k x k, symmetric because Gbar is)
      // GR and Q go through the two (still unused) column groups of W
      blk_store_lds<BS>(GR, W, LDW, lr, lc);
      lds_load_matrix(W + NP, LDW, NP, NP, qd, k, k, lane);
      wave_sync();
      blk_zero<BS>(Rbar);
      mm_acc<BS, false>(Rbar, W, LDW, W + NP, LDW, k, lr, lc);
      double Qb[BS][BS];
      blk_zero<BS>(Qb);
      mm_acc_ta<BS>(Qb, Rs, LD, W, LDW, n, lr, lc);
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) Rbar[i][j] *= 2.0;
      blk_store_global<BS>(Qb, q_bar + (size_t)draw * k * k, k, k, k, lr, lc);
      wave_sync();
      for (int idx = lane; idx < NP * LDW; idx += 64) W[idx] = 0.0;
      wave_sync();
    } else {
    blk_load_lds<BS>(Rb, Rs, LD, lr, lc);
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      const int c = lc * BS + j;
      const double qj = (c < k) ? qd[c] : 0.0;
      double colsum = 0.0;
#pragma unroll
      for (int i = 0; i < BS; ++i) {
        Rbar[i][j] = 2.0 * GR[i][j] * qj;
        colsum = fma(Rb[i][j], GR[i][j], colsum);
      }
      colsum += shfl_xor_f64(colsum, 8);
      colsum += shfl_xor_f64(colsum, 16);
      colsum += shfl_xor_f64(colsum, 32);
      if (lr == 0 && c < k) q_bar[(size_t)draw * k + c] = colsum;
    }
    }
    }
    // M = B + C T; W = [M' | Rbar]
    {
      double Mb[BS][BS];
      blk_load_global<BS>(Mb, B + off, n, n, n, lr, lc);
      mm_acc<BS, false>(Mb, Cs + c_lo, LD, Ts + c_lo * LD, LD, c_hi - c_lo, lr, lc);
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) W[(lc * BS + j) * LDW + lr * BS + i] = Mb[i][j];
      blk_store_lds<BS>(Rbar, W + NP, LDW, lr, lc);
    }
    wave_sync();  // (Gbar in Ms has been consumed by every lane)
    gauss_jordan_blocked<BS>(W, LDW, n, 2, Lbuf, Ybuf, prow, lane);
    gj_unpermute<BS>(W, LDW, n, 1, 2, prow, lane);
    double Xb[BS][BS];
    blk_load_lds<BS>(Xb, W + NP, LDW, lr, lc);  // X = M^-T Rbar  (columns >= k are zero)
    blk_store_global<BS>(Xb, D_bar + offk, n, k, k, lr, lc, -1.0);
    // Mbar = -X R'
    wave_sync();
    blk_store_lds<BS>(Xb, W, LDW, lr, lc);  // X -> dead group 0
    {
      double Ct[BS][BS];
      blk_load_lds_t<BS>(Ct, Cs, LD, lr, lc);
      wave_sync();
      blk_store_lds<BS>(Ct, Cs, LD, lr, lc);  // Cs <- C'
    }
    wave_sync();
    double Mbar[BS][BS];
    blk_zero<BS>(Mbar);
    mm_acc<BS, true>(Mbar, W, LDW, Rs, LD, k, lr, lc);  // X R'
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) Mbar[i][j] = -Mbar[i][j];
    blk_store_global<BS>(Mbar, B_bar + off, n, n, n, lr, lc);
    wave_sync();
    blk_store_lds<BS>(Mbar, Ms, LD, lr, lc);
    wave_sync();
    double Cb[BS][BS], Tb[BS][BS];
    blk_zero<BS>(Cb);
    mm_acc<BS, true>(Cb, Ms, LD, Ts, LD, t_hi, lr, lc);  // Mbar T'
    blk_store_global<BS>(Cb, C_bar + off, n, n, n, lr, lc);
    if (Rbar_in)
      blk_zero<BS>(Tb);
    else
      blk_load_global<BS>(Tb, Tbar + off, n, n, n, lr, lc);
    mm_acc<BS, false>(Tb, Cs, LD, Ms, LD, n, lr, lc);  // Tbar += C' Mbar
    blk_store_global<BS>(Tb, Tbar + off, n, n, n, lr, lc);
  }
}

}  // namespace dsge

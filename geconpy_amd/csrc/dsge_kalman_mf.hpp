// Kalman filter log-likelihood, selector design matrix, covariance in the TILE LAYOUT of the FP64 matrix core: "kalman_mf_kernel"
// (round 6).
//
// Same algorithm, conventions and missing-data rules as kalman_nt_kernel (dsge_kalman_nt.hpp: exact reduction to the retained
// variables U = states u observed non-states, states first; stationary initial covariance by doubling; p x p inverse by row-per-lane
// Gauss-Jordan; P+ = P - K (P Zm' + jit_V K)' + jit_P I; steady-state switch with a register-resident mean recursion; the recursion
// of SURVEY.md Appendix B.4 for gEconpy/model/statespace.py:1151-1157) -- phases (a)-(d) and the steady loop are that kernel's code.
// What changed is where the covariance lives.  There: 3 x 3 register blocks on an 8 x 8 lane grid (24-wide tile for 18 variables:
// 2.4 x the flops), 432 FMAs + 144 LDS loads per lane for the two prediction products, 72 FMAs for the downdate, a shuffle
// transposition to symmetrise.  Here: the symmetric matrices P, Q are held as their UPPER 4 x 4 tiles in the result layout of
// v_mfma_f64_4x4x4f64 (dsge_mfma4.hpp; lane l of group g holds element (4 ta + (l >> 4), 4 tb + (l & 3)) of tile (ta, tb) =
// Mfma4Upper<TM>::tile(g, (l >> 2) & 3)): FOUR registers for 18 x 18 (5 x 5 tiles, 15 upper).  Per full step
//   (e) P+ = P + K (-V)'           8 issues with C = P (the rank-p downdate as a product),
//   (f) W  = P+[S,S] Tc'          35 issues, stored transposed,
//       P' = Tc W + Q             20 issues with C = Q, upper tiles only: symmetric by construction, no transposition,
// and the mean prediction rides along as column m of the last tile.  63 issues of 256 FMAs instead of ~500 FMAs of 64 plus their
// operand loads: tools/mfma_probe/product_probe.hip measured the two products alone at 0.61 x the VALU form's launch time.
// The register blocks are gone, so are their 96 operand registers.
//
// Capacity: s <= 4 KT state variables, m <= 4 TM retained variables, p <= 8, selector Z; a draw beyond it is flagged
// (DSGE_ST_INTERNAL_RERUN) for the next instance of the launcher's cascade, as in the other fast kernels.
#pragma once
#include <type_traits>

#include "dsge_kalman_nt.hpp"
#include "dsge_kalman_rec.hpp"
#include "dsge_mfma4.hpp"

namespace dsge {

template <int KT, int TM>
struct KmfSmem {
  static_assert(KT >= 1 && KT <= TM && TM <= 8, "tiles");
  static constexpr int NS = 4 * KT, NM = 4 * TM;  // padded state block, padded retained variables
  static constexpr int LDK = NS + 2;              // rows of Tc / W' / Pc (even: 16-byte aligned rows; + 2: conflict-free b128 reads)
  static constexpr int NR = 8 * ((NM + 7) / 8);   // rows of the update's panels (the update walks rows g8 + 8 ps)
  static constexpr int RP = NR / 8, PS = 10;
  static constexpr int WT = NM * LDK > NR * PS ? NM * LDK : NR * PS;  // W' buffer; the -V panel and the staged R alias it
  // doubles: Tc NM*LDK, Wt WT, Pc NS*LDK, PZt, Ks NR*PS each, av, af NR each, vv/dd/hh/zv 8 each; ints perm NR, zpos 8
  static constexpr size_t doubles = (size_t)NM * LDK + WT + (size_t)NS * LDK + 2 * (size_t)NR * PS + 2 * NR + 4 + 44 + NR / 2 + 4;
  static constexpr size_t bytes = sizeof(double) * doubles;
};

#ifndef KMF_WAVES
#define KMF_WAVES 2
#endif
// REC = true (round 6): the forward sweep of the gradient -- every step also writes its record for the reverse sweep
// (dsge_kalman_rec.hpp, KgRec<RBS>: the record of the 8 RBS-wide tile the reverse sweep runs on; 4 TM <= 8 RBS).  K, F, F^-1, a_t
// and the segment links as kalman_nt_kernel<.., REC> writes them; P+ in the TILE layout -- element g of lane l at [g 64 + l], the
// upper tiles straight from the registers, four coalesced stores for 18 variables instead of nine -- which the record announces in
// its layout word (KgRec::TS_LAYOUT) and the reverse sweep scatters into its LDS square.
template <int KT, int TM, bool DBG, bool REC = false, int RBS = (4 * TM + 7) / 8>
__global__ __launch_bounds__(64, KMF_WAVES) void kalman_mf_kernel(
    const double* __restrict__ T, const double* __restrict__ RQR, const double* __restrict__ P0, const double* __restrict__ Z,
    int z_batched, const double* __restrict__ dvec, int d_batched, const double* __restrict__ Hdiag, int h_batched,
    const double* __restrict__ y, int batch, int m_full, int p, int T_len, FilterConv cv, double missing_fill, double steady_tol,
    double* __restrict__ logp_out, int32_t* __restrict__ status, long long* __restrict__ dbg, int rerun_only,
    int32_t* __restrict__ steady_at, const int32_t* __restrict__ order, const double* __restrict__ Rsel,
    const double* __restrict__ qdiag, int q_batched, int k_shocks, const unsigned long long* __restrict__ colmask_in,
    double* __restrict__ rec_store = nullptr) {
  using SM = KmfSmem<KT, TM>;
  using RC = KgRec<RBS>;
  static_assert(!REC || (4 * TM <= 8 * RBS && Mfma4Upper<TM>::NG * 64 <= 8 * RBS * 8 * RBS), "the record's tile holds the filter's");
  using UX = Mfma4Upper<TM>;
  using MW = Mfma4Map<KT, TM>;
  constexpr int NS = SM::NS, NM = SM::NM, LDK = SM::LDK, NR = SM::NR, RP = SM::RP, PS = SM::PS, NG = UX::NG;
  // more than five groups of tiles (TM = 7: 28 upper tiles): the per-element tables of where a value is stored would push the kernel
  // past 256 registers -- the Pc offsets are recomputed where they are used and the selector values come from an LDS table
  constexpr bool LEAN = NG > 5;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Tc = smem;                // NM x LDK   transition, states-first ordering (columns >= s exactly zero)
  double* Wt = Tc + NM * LDK;       // WT         W' : Wt[j][k] = (P+[S,S] Tc')[k][j]
  double* Vs = Wt;                  //   alias: NR x PS  -V = -(P Zm' + jit_V K)     (phases d, e)
  double* Pc = Wt + SM::WT;         // NS x LDK   P+ restricted to the state block, full square
  double* PZt = Pc + NS * LDK;      // NR x PS    (predicted P) Z', unmasked
  double* Ks = PZt + NR * PS;       // NR x PS    K = P Zm' Finv
  double* av = Ks + NR * PS;        // NR + 2     predicted state (+ the dump slot of the mean's store)
  double* avd = av;
  double* af = av + NR + 2;         // NR + 2     filtered state (+ a dump slot)
  double* vv = af + NR + 2;         // 8 innovation (+ 2: dump slot)
  double* dd = vv + 10;             // 8 obs intercept
  double* hh = dd + 8;              // 8 diag(H)
  double* zv = hh + 8;              // 8 selector values
  double* zvt = zv + 8;             // 10: selector value by observation index, [8] = 0 (LEAN: the P Z' stores look their factor up)
  int* perm = (int*)(zvt + 10);     // NR: position -> original variable (states first)
  int* zpos = perm + NR;            // 8: position of the state each observation selects
  constexpr int W_DUMP = (NM - 1) * LDK + LDK - 1;   // padding slots that nothing reads: targets of the stores of dead blocks
  constexpr int TC_DUMP = (NM - 1) * LDK + LDK - 1;
  constexpr int PC_DUMP = (NS - 1) * LDK + LDK - 1;
  constexpr int PZ_DUMP = (NR - 1) * PS + PS - 1;
  const int lane = threadIdx.x;
  const int blk = (lane >> 2) & 3, i4 = lane & 3, kq = lane >> 4;
  const double LN2PI = 1.8378770664093453, LN2 = 0.6931471805599453;

  if (rerun_only && rerun_pass_is_empty(status, batch, order)) return;
  for (int bi = blockIdx.x; bi < batch; bi = batch) {  // (one draw per workgroup: see kalman_nt_kernel)
    const int draw = __builtin_amdgcn_readfirstlane(order ? order[bi] : bi);
    const int32_t st_in = __builtin_amdgcn_readfirstlane(status[draw]);
    if (rerun_only) {
      if (st_in != DSGE_ST_INTERNAL_RERUN) continue;
    } else if (st_in != 0) {
      if (lane == 0) logp_out[draw] = -INFINITY;
      continue;
    }
    const size_t off = (size_t)draw * m_full * m_full;
    wave_sync();
    for (int idx = lane; idx < (int)SM::doubles; idx += 64) smem[idx] = 0.0;

    // ---- exact state-space reduction to U = S u O, states first (kalman_nt_kernel's code) ---------------------------------
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
    ok = ok && (s <= NS) && (m <= NM) && (s >= 1);
    int my_pos = -1;
    if (lane < m_full) {
      const unsigned long long below = (1ull << lane) - 1ull;
      if ((colmask >> lane) & 1ull)
        my_pos = __popcll(colmask & below);
      else if ((extra >> lane) & 1ull)
        my_pos = s + __popcll(extra & below);
      if (my_pos >= 0 && my_pos < NR) perm[my_pos] = lane;
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      if (o < p && zrow[o] != 0.0) {
        zpos[o] = (my_pos >= 0 && my_pos < NR) ? my_pos : 0;
        zv[o] = zrow[o];
      }
    }
    if (!ok) {
      if (lane == 0) status[draw] = DSGE_ST_INTERNAL_RERUN;
      continue;
    }
    if (rerun_only && lane == 0) status[draw] = 0;
    wave_sync();

    // ---- this lane's elements of the symmetric matrices: (rr[g], cc[g]) of the upper tiles, and what they feed ---------------
    int rr[NG], cc[NG];
    bool inm[NG], up[NG];    // inside the m x m matrix; on or above the diagonal (the copy that is stored / mirrored)
    constexpr int NT_ = LEAN ? 1 : NG;
    int pcw0_[NT_], pcw1_[NT_];  // where the element goes in Pc (and its mirror image), or the dump slot
    int pzw0[NG], pzw1[NG];      // where it goes in the P Z' panel: row r at the observation selecting c, row c at the one selecting r
                                 // (LEAN: shifted left by four, the observation's index into zvt -- 8 = none -- in the low bits)
    double pzv0_[NT_], pzv1_[NT_];
    if (lane < 9) zvt[lane] = (lane < p) ? zv[lane] : 0.0;  // selector values by observation, zvt[8] = 0
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      int ta, tb;
      bool live;
      UX::tile(g, blk, ta, tb, live);
      const int r = 4 * ta + kq, c = 4 * tb + i4;
      rr[g] = r;
      cc[g] = c;
      inm[g] = live && r < m && c < m;
      up[g] = live && r <= c;
      int oc = -1, orr = -1;
      double zc = 0.0, zr = 0.0;
      for (int o = 0; o < p; ++o) {
        if (zpos[o] == c) {
          oc = o;
          zc = zv[o];
        }
        if (zpos[o] == r) {
          orr = o;
          zr = zv[o];
        }
      }
      const bool w0 = up[g] && inm[g] && oc >= 0, w1 = up[g] && inm[g] && r < c && orr >= 0;
      if constexpr (LEAN) {
        pzw0[g] = ((w0 ? r * PS + oc : PZ_DUMP) << 4) | (w0 ? oc : 8);
        pzw1[g] = ((w1 ? c * PS + orr : PZ_DUMP) << 4) | (w1 ? orr : 8);
      } else {
        const bool in_pc = up[g] && c < NS;  // (r <= c < NS)
        pcw0_[g] = in_pc ? r * LDK + c : PC_DUMP;
        pcw1_[g] = (in_pc && r < c) ? c * LDK + r : PC_DUMP;
        pzw0[g] = w0 ? r * PS + oc : PZ_DUMP;
        pzw1[g] = w1 ? c * PS + orr : PZ_DUMP;
        pzv0_[g] = zc;
        pzv1_[g] = zr;
      }
    }
    int rowa[NG], rowb[NG];  // operand rows of this lane's block in the three upper-tile products (computed once: mfma4_nt_upper_acc)
    mfma4_upper_rows<TM>(lane, rowa, rowb);
    auto pcw0 = [&](int g) {
      if constexpr (LEAN) return (up[g] && cc[g] < NS) ? rr[g] * LDK + cc[g] : PC_DUMP;
      else return pcw0_[g];
    };
    auto pcw1 = [&](int g) {
      if constexpr (LEAN) return (up[g] && cc[g] < NS && rr[g] < cc[g]) ? cc[g] * LDK + rr[g] : PC_DUMP;
      else return pcw1_[g];
    };
    wave_sync();

    // ---- T -> Tc (gathered: position -> original variable), Q and P in tile layout -----------------------------------------
    auto load_T = [&]() {
      lane_loop_batched<8>(m * s, lane,
                           [&](int idx) {
                             const int r = idx / s, c = idx - r * s;
                             return T[off + (size_t)perm[r] * m_full + perm[c]];
                           },
                           [&](int idx, double v) {
                             const int r = idx / s, c = idx - r * s;
                             Tc[r * LDK + c] = v;
                           });
    };
    double Qt[NG], Pt[NG];
    if (Rsel) {
      // sym(R diag(q) R')[U,U] from the selection matrix itself (m_full x k_shocks staged in the W' buffer; the launcher checked
      // that it fits), rqr_kernel's expression and summation order
      const int kp = (k_shocks + 1) & ~1;
      const double* Rg = Rsel + (size_t)draw * m_full * k_shocks;
      for (int idx = lane; idx < m_full * k_shocks; idx += 64) {
        const int i = idx / k_shocks, c = idx - i * k_shocks;
        Wt[i * kp + c] = Rg[idx];
      }
      const double* qd = qdiag + (q_batched ? (size_t)draw * k_shocks : 0);
      wave_sync();
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const double2* ri = reinterpret_cast<const double2*>(Wt + (inm[g] ? perm[rr[g]] : 0) * kp);
        const double2* rj = reinterpret_cast<const double2*>(Wt + (inm[g] ? perm[cc[g]] : 0) * kp);
        double a0 = 0.0, a1 = 0.0;
        for (int c2 = 0; 2 * c2 < kp; ++c2) {
          const double2 ti = ri[c2], tj = rj[c2];
          a0 = fma(ti.x * tj.x, qd[2 * c2], a0);
          a1 = fma(ti.y * tj.y, (2 * c2 + 1 < k_shocks) ? qd[2 * c2 + 1] : 0.0, a1);
        }
        Qt[g] = inm[g] ? a0 + a1 : 0.0;
      }
      wave_sync();
      for (int idx = lane; idx < m_full * kp; idx += 64) Wt[idx] = 0.0;  // (the filter relies on zero padding)
    } else {
#pragma unroll
      for (int g = 0; g < NG; ++g) Qt[g] = inm[g] ? RQR[off + (size_t)perm[rr[g]] * m_full + perm[cc[g]]] : 0.0;
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) Pt[g] = (inm[g] && P0) ? P0[off + (size_t)perm[rr[g]] * m_full + perm[cc[g]]] : 0.0;
    load_T();
    wave_sync();
    // full square of the state block from the upper elements: Pc[r][c] and Pc[c][r]
#define KMF_STORE_PC()                        \
  do {                                        \
    _Pragma("unroll") for (int g = 0; g < NG; ++g) { \
      Pc[pcw0(g)] = Pt[g];                    \
      Pc[pcw1(g)] = Pt[g];                    \
    }                                         \
  } while (0)
    if (!P0) {
      // ---- P0 = dlyap(T, RQR)[U,U] by doubling on the reduced model (statespace.py:814-815): P += A_k P A_k', A_{k+1} = A_k^2 ----
#pragma unroll
      for (int g = 0; g < NG; ++g) Pt[g] = Qt[g];
      bool lyap_ok = false;
      for (int itl = 0; itl < 64; ++itl) {
        wave_sync();
        KMF_STORE_PC();
        wave_sync();
        mfma4_nt<KT, KT, TM, LDK>(Pc, Tc, lane, [&](int g, double d) {  // W = P[S,S] A_k', transposed
          const int at = (4 * MW::tb(g, blk) + i4) * LDK + 4 * MW::ta(g, blk) + kq;
          Wt[MW::live(g, blk) ? at : W_DUMP] = d;
        });
        using MA = Mfma4Map<TM, KT>;
        double a2r[MA::NG];
        mfma4_nn<KT, TM, KT, LDK>(Tc, Tc, lane, [&](int g, double d) { a2r[g] = d; });  // A_k[:,S] A_k[S,:]
        wave_sync();
        double dmax = 0.0, pmax = 0.0;
        mfma4_nt_upper_acc<KT, TM, LDK>(Tc, Wt, lane, rowa, rowb, [](int) { return 0.0; }, [&](int g, double d) {
          const double dlt = inm[g] ? d : 0.0;
          Pt[g] += dlt;
          dmax = nanmax(dmax, fabs(dlt));
          pmax = nanmax(pmax, fabs(Pt[g]));
        });
        wave_sync();
#pragma unroll
        for (int g = 0; g < MA::NG; ++g) {
          const int at = (4 * MA::ta(g, blk) + kq) * LDK + 4 * MA::tb(g, blk) + i4;
          Tc[MA::live(g, blk) ? at : TC_DUMP] = a2r[g];
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
      for (int idx = lane; idx < NM * LDK; idx += 64) Tc[idx] = 0.0;
      for (int idx = lane; idx < SM::WT; idx += 64) Wt[idx] = 0.0;  // (the filter relies on zero padding)
      wave_sync();
      load_T();
      if (!lyap_ok) {
        if (lane == 0) {
          status[draw] |= DSGE_ST_LYAP_FAIL;
          logp_out[draw] = -INFINITY;
        }
        continue;
      }
    }
    double* const rec_d = REC ? rec_store + (size_t)draw * RC::per_draw(T_len) : nullptr;  // this draw's record
    int seg_src = -1;                                                                       // source step of the running segment
    if constexpr (REC) {  // P_0, row-major NP x NP behind the last step (zero outside m x m), and the layout word
      constexpr int RNP = 8 * RBS;
      double* p0s = rec_d + (size_t)T_len * RC::STEP;
      for (int idx = lane; idx < RNP * RNP; idx += 64) {
        const int r = idx / RNP, c = idx - r * RNP;
        if (!(r < m && c < m)) p0s[idx] = 0.0;
      }
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (up[g] && inm[g]) {
          p0s[rr[g] * RNP + cc[g]] = Pt[g];
          p0s[cc[g] * RNP + rr[g]] = Pt[g];
        }
      }
      if (lane == 0) rec_d[RC::tail_state_off(T_len) + RC::TS_LAYOUT] = (double)(RC::LAYOUT_TILES + TM);
    }
    if (lane < 8) {
      dd[lane] = (dvec && lane < p) ? dvec[(d_batched ? (size_t)draw * p : 0) + lane] : 0.0;
      hh[lane] = (Hdiag && lane < p) ? Hdiag[(h_batched ? (size_t)draw * p : 0) + lane] : 0.0;
    }
    // P Z' panel of the predicted covariance (unmasked), from the upper elements and their mirror images
#define KMF_STORE_PZT()                                 \
  do {                                                  \
    _Pragma("unroll") for (int g = 0; g < NG; ++g) {    \
      if constexpr (LEAN) {                             \
        PZt[pzw0[g] >> 4] = zvt[pzw0[g] & 15] * Pt[g];  \
        PZt[pzw1[g] >> 4] = zvt[pzw1[g] & 15] * Pt[g];  \
      } else {                                          \
        PZt[pzw0[g]] = pzv0_[g] * Pt[g];                \
        PZt[pzw1[g]] = pzv1_[g] * Pt[g];                \
      }                                                 \
    }                                                   \
  } while (0)
    KMF_STORE_PZT();
    const int v_zpos = (lane < p) ? zpos[lane] : 0;
    const double v_zv = (lane < p) ? zv[lane] : 0.0, v_dd = (lane < 8) ? dd[lane & 7] : 0.0;
    wave_sync();

    const int r8 = lane & 7, g8 = lane >> 3;  // the update runs on 8 replicas of an 8-lane group: lane -> row r8 of F
    const int r_zpos = (r8 < p) ? zpos[r8] : 0;
    const double r_zv = (r8 < p) ? zv[r8] : 0.0, r_dd = dd[r8], r_hh = hh[r8];
    const bool fold_a = m < NM;  // a spare padding column: the mean prediction rides along in the X product as column m

    double quad_sum = 0.0, quad_comp = 0.0;  // Kahan sum of v' Finv v over observed steps
    double ld_mant = 1.0;                    // prod of pivots = mant * 2^exp
    int ld_exp = 0;
    int n_ll_steps = 0, n_obs_entries = 0;
    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long tk_start = DBG ? clock64() : 0;
    int steady_step = -1;
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
        if (lane < 8 * RBS) sg[RC::OFF_A + lane] = (lane < NR) ? av[lane] : 0.0;
        if (lane == 0) {
          sg[RC::OFF_SRC] = (double)t;
          sg[RC::OFF_PREV] = (double)seg_src;
        }
        seg_src = t;
      }
      double2 fr2[4], pz2[RP][4];
      double pzo[RP], avi[RP];
#pragma unroll
      for (int q2 = 0; q2 < 4; ++q2) fr2[q2] = *reinterpret_cast<const double2*>(&PZt[r_zpos * PS + 2 * q2]);
      const double a_sel = av[r_zpos];
      auto load_panel = [&]() {
#pragma unroll
        for (int ps = 0; ps < RP; ++ps) {
          const int i = g8 + 8 * ps;
#pragma unroll
          for (int q2 = 0; q2 < 4; ++q2) pz2[ps][q2] = *reinterpret_cast<const double2*>(&PZt[i * PS + 2 * q2]);
          pzo[ps] = PZt[i * PS + r8];
          avi[ps] = av[i];
        }
      };
      if constexpr (!LEAN) load_panel();  // (LEAN: requested in phase (d) -- 72 registers less across the elimination)
      __builtin_amdgcn_sched_barrier(0);
      bool steady = false;
      double pm = 0.0;  // max |P_{t|t-1}| = the largest diagonal entry (P is positive semi-definite)
      if (steady_tol > 0.0) {
        double pscale = 0.0;
#pragma unroll
        for (int g = 0; g < NG; ++g) pscale = fmax(pscale, (rr[g] == cc[g]) ? fabs(Pt[g]) : 0.0);
        const unsigned hi = (unsigned)__double2hiint(pscale), lo = (unsigned)__double2loint(pscale);
        const unsigned mhi = wave_max_u32(hi);
        const unsigned mlo = wave_max_u32(hi == mhi ? lo : 0u);
        pm = __hiloint2double((int)mhi, (int)mlo);
      }
      // ---- (b) innovation v[r8] and row r8 of F = Zm P Zm' + Hm + jitter I, replicated over the eight 8-lane groups ----------
      const double c_r = obs ? r_zv : 0.0;
      const double v_r = (obs ? yt : 0.0) - (((obs || !cv.mask_d) ? r_dd : 0.0) + c_r * a_sel);
      const double dg = (r8 < p) ? ((obs ? r_hh : 0.0) + cv.jit_F) : 1.0;
      vv[lane < 8 ? lane : 8] = v_r;  // (entry 8: a slot for the other lanes -- no branch)
      asm volatile("" ::: "memory");
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
      // ---- (c) Finv by Gauss-Jordan, one row per lane (SPD: no pivoting); the pivot row arrives through SGPRs ----------------
      double step_mant = 1.0, inv_own = 1.0;
      int step_exp = 0;
      // (p >= 7: pivots 0..6 unconditionally, pivot 7 only when there is an eighth observation: a row beyond p -- like the row of a missing
      //  observation -- is a row of the identity, pivot 1, multipliers 0, mantissa 0.5 x 2^1: a no-op.  Seven branches `j < p` cut the
      //  chain into basic blocks the scheduler could not overlap; one uniform branch at the end does not)
      auto pivot = [&](auto jt) {
        constexpr int j = decltype(jt)::value;
        double rowj[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) rowj[q] = readlane_f64(fr[q], j);
        // (ONE Newton step on v_rcp_f64: 2.2e-15 relative, tools/latency_probe/rcp_probe.hip -- the multipliers and the final
        //  row scaling of a 7 x 7 inverse whose conditioning costs more digits than that; the determinant takes the pivots
        //  themselves.  Two dependent FMAs less on the chain of every pivot)
        double inv = __builtin_amdgcn_rcp(rowj[j]);
        inv = inv * fma(-rowj[j], inv, 2.0);
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
      };
      if (p >= 7) {
        pivot(std::integral_constant<int, 0>{});
        pivot(std::integral_constant<int, 1>{});
        pivot(std::integral_constant<int, 2>{});
        pivot(std::integral_constant<int, 3>{});
        pivot(std::integral_constant<int, 4>{});
        pivot(std::integral_constant<int, 5>{});
        pivot(std::integral_constant<int, 6>{});
        if (p > 7) pivot(std::integral_constant<int, 7>{});
      } else {  // (few observables: the no-op pivots would cost more than the branches)
        pivot(std::integral_constant<int, 0>{});
        if (p > 1) pivot(std::integral_constant<int, 1>{});
        if (p > 2) pivot(std::integral_constant<int, 2>{});
        if (p > 3) pivot(std::integral_constant<int, 3>{});
        if (p > 4) pivot(std::integral_constant<int, 4>{});
        if (p > 5) pivot(std::integral_constant<int, 5>{});
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
        {  // (a step without any observed entry contributes nothing: selects on the wave-uniform condition instead of a branch,
           //  which would end the basic block between the elimination and the gain)
          const bool any = n_obs > 0;
          const double yk = ((lane < 8) ? (v_r * inv_own) * (w0 + w1) : 0.0) - quad_comp;
          const double tk = quad_sum + yk;
          quad_comp = any ? (tk - quad_sum) - yk : quad_comp;
          quad_sum = any ? tk : quad_sum;
          int e;
          const double lm = frexp(ld_mant * step_mant, &e);
          ld_mant = any ? lm : ld_mant;
          ld_exp += any ? e + step_exp : 0;
          n_ll_steps += any ? 1 : 0;
          n_obs_entries += n_obs;
        }
      }
      if constexpr (DBG) {
        const long long tk1 = clock64();
        ph[0] += tk1 - tk0;
        tk0 = tk1;
      }
      // ---- (d) K = (P Zm') Finv, V = P Zm' + jit_V K (stored negated), a+ = a + K v -----------------------------------------
      if constexpr (LEAN) load_panel();
#pragma unroll
      for (int ps = 0; ps < RP; ++ps) {
        const int i = g8 + 8 * ps;
        double k0 = 0.0, k1 = 0.0;
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) {
          k0 = fma(pz2[ps][q2].x, fr[2 * q2], k0);
          k1 = fma(pz2[ps][q2].y, fr[2 * q2 + 1], k1);
        }
        const double kk = obs ? (k0 + k1) * inv_own : 0.0;
        Ks[i * PS + r8] = kk;
        Vs[i * PS + r8] = -fma(cv.jit_V, kk, obs ? pzo[ps] : 0.0);
        double part = kk * v_r;
        part += dpp_move_f64<0xB1, 0xf>(part);
        part += dpp_move_f64<0x4E, 0xf>(part);
        part += dpp_move_f64<0x141, 0xf>(part);
        af[(r8 == 0) ? i : NR] = avi[ps] + part;  // (entry NR: a slot for the lanes that own nothing -- no branch)
      }
      wave_sync();  // #2
      if constexpr (REC) {
        for (int idx = lane; idx < 8 * RBS * 8; idx += 64) sg[RC::OFF_K + idx] = ((idx >> 3) < NR) ? Ks[(idx >> 3) * PS + (idx & 7)] : 0.0;
      }
      if constexpr (DBG) {
        const long long tk1 = clock64();
        ph[1] += tk1 - tk0;
        tk0 = tk1;
      }
      // ---- (e) P+ = P + K (-V)' + jit_P I on the matrix core (C = P), then the state block -> Pc (full square) ---------------
      {
        double dmax = 0.0;
        mfma4_nt_upper_acc<2, TM, PS>(Ks, Vs, lane, rowa, rowb, [&](int g) { return Pt[g]; },
                                      [&](int g, double d) {
                                        const double pn = inm[g] ? d + ((rr[g] == cc[g]) ? cv.jit_P : 0.0) : 0.0;
                                        Pt[g] = pn;
                                      });
        if constexpr (REC) {  // P+ in tile layout, straight from the registers
#pragma unroll
          for (int g = 0; g < NG; ++g) sg[(size_t)g * 64 + lane] = Pt[g];
        }
        if (steady_tol > 0.0) {
          // (the old values are read UNCONDITIONALLY and together -- volatile: under the predicate the compiler turned each read into
          //  a branch with its own wait, four LDS round trips in a row)
          if constexpr (!LEAN) {
            double oldv[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) oldv[g] = *reinterpret_cast<const volatile double*>(&Pc[pcw0(g)]);
#pragma unroll
            for (int g = 0; g < NG; ++g) dmax = fmax(dmax, (pcw0(g) != PC_DUMP) ? fabs(Pt[g] - oldv[g]) : 0.0);
          } else {  // (seven groups: one at a time, the registers are needed elsewhere)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
              const int at = pcw0(g);
              const double old = *reinterpret_cast<const volatile double*>(&Pc[at]);
              dmax = fmax(dmax, (at != PC_DUMP) ? fabs(Pt[g] - old) : 0.0);
            }
          }
          steady = (t > 0) && (__ballot(!(dmax <= steady_tol * pm)) == 0ull);
        }
        // (no fence between the reads of the old state block and its stores: same element type, and LDS executes a wavefront's
        //  accesses in program order)
        KMF_STORE_PC();
      }
      // the filtered mean as row m of W': column m of X is then Tc a+ (rows >= m of Tc are zero: row m of W' feeds nothing else)
      double af_l = 0.0;
      if (fold_a) {
        af_l = af[lane < NR ? lane : 0];
      } else if (lane < m) {
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
      wave_sync();  // #3
      if constexpr (DBG) {
        const long long tk1 = clock64();
        ph[2] += tk1 - tk0;
        tk0 = tk1;
      }
      // ---- (f) predict: W = P+[S,S] Tc' (stored transposed);  P' = Tc W + Q (upper tiles, C = Q) ---------------------------------
      mfma4_nt<KT, KT, TM, LDK>(Pc, Tc, lane, [&](int g, double d) {
        const int at = (4 * MW::tb(g, blk) + i4) * LDK + 4 * MW::ta(g, blk) + kq;
        Wt[MW::live(g, blk) ? at : W_DUMP] = d;
      });
      Wt[(fold_a && lane < NS) ? m * LDK + lane : W_DUMP] = af_l;  // (after the W stores: LDS keeps a wavefront's program order)
      wave_sync();  // #4
      if constexpr (DBG) {
        const long long tk1 = clock64();
        ph[3] += tk1 - tk0;
        tk0 = tk1;
      }
      mfma4_nt_upper_acc<KT, TM, LDK>(Tc, Wt, lane, rowa, rowb, [&](int g) { return Qt[g]; },
                                      [&](int g, double d) {
                                        // column m: the predicted mean (Q is zero there); everything else outside m x m: padding
                                        const bool is_mean = fold_a && cc[g] == m && rr[g] < m && up[g];
                                        avd[is_mean ? rr[g] : NR] = is_mean ? d : 0.0;  // (entry NR: a slot of its own for the rest)
                                        Pt[g] = inm[g] ? d : 0.0;
                                      });
      KMF_STORE_PZT();
      wave_sync();  // #5
      if constexpr (DBG) {
        const long long tk1 = clock64();
        ph[4] += tk1 - tk0;
        tk0 = tk1;
      }
      if (!steady) continue;
      // ==== steady-state steps: mean recursion only, while the missing-data mask stays the same (kalman_nt_kernel's loop) ====
      if (steady_step < 0) steady_step = t + 1;
      {
        double trow[NS], finv_row[8], kr_ss[8];
        double av_reg = (lane < m) ? av[lane] : 0.0;
#pragma unroll
        for (int kk = 0; kk < NS; ++kk) trow[kk] = (lane < NM) ? Tc[lane * LDK + kk] : 0.0;  // columns >= s are zero
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          finv_row[q] = (lane < 8) ? fr[q] * inv_own : 0.0;
          kr_ss[q] = (lane < m) ? Ks[lane * PS + q] : 0.0;
        }
        while (t + 1 < T_len) {
          const double yt_s = yt_next;
          const bool obs_s = (lane < p) && (yt_s == yt_s) && (yt_s != missing_fill);
          if (__ballot(obs_s) != omask) break;
          ++t;
          yt_next = y[(size_t)((t + 1 < T_len) ? t + 1 : t) * p + r8c];
          if constexpr (REC) {
            double* sgs = rec_d + (size_t)t * RC::STEP;
            if (lane < 8 * RBS) sgs[RC::OFF_A + lane] = av_reg;
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
          for (int kk = 0; kk < NS; kk += 2) {
            s0 = fma(trow[kk], readlane_f64(afi, kk), s0);
            s1 = fma(trow[kk + 1], readlane_f64(afi, kk + 1), s1);
          }
          av_reg = (lane < m) ? s0 + s1 : 0.0;
          if constexpr (DBG) ++ph[6];
        }
        if (lane < m) av[lane] = av_reg;
      }
      wave_sync();
      if constexpr (DBG) ph[5] += clock64() - tk0;
    }
#undef KMF_STORE_PZT
#undef KMF_STORE_PC
    if (DBG && dbg && draw == 0 && lane == 0) {
      ph[7] = clock64() - tk_start;
      for (int k = 0; k < 8; ++k) dbg[k] = ph[k];
    }
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

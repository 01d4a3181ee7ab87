// Kalman filter log-likelihood, structure-exploiting fast path ("kalman_sel_kernel").
//
// Same recursion as kalman_kernel (dsge_kernels.hpp), restricted to
//   * p <= 8 observed series,
//   * a selector design matrix: every row of Z has exactly one non-zero entry, in distinct
//     columns (gEconpy's default `_make_design_matrix` branch, statespace.py:282-296),
// and exploiting two exact structural facts:
//   (1) the columns of T that belong to non-state variables are EXACTLY zero when T comes out
//       of cycle reduction (A has exactly-zero columns there and Gauss-Jordan maps a zero
//       right-hand side to exact zeros), so  T P+ T' = Tc P+[S,S] Tc'  with Tc = T[:,S],
//       S = non-zero columns, s = |S|.  Dropping the zero columns removes only additions of
//       +0.0 and is therefore bit-identical to the dense product;
//   (2) with a selector Z, P Z' is a column gather of P and F = Z P Z' + H is a p x p gather.
// The update uses, with Finv = F^-1 (F = Zm P Zm' + Hm + jit_F I), K = P Zm' Finv:
//     P+ = P - K (P Zm' + jit_V K)' + jit_P I
// which, with jit_V = jit_F, equals the Joseph form  sym((I-KZm) P (I-KZm)') + sym(K Hm K') + jit_P I
// exactly in real arithmetic (expand with K F = P Zm'), and with jit_V = 0 the plain form P - K F K' + jit_P I.
// jit_F, jit_P, jit_V, the ln 2pi constant and the masking of d are run-time conventions (FilterConv, dsge_device.hpp).
//
// The hints (s_cap = LDS capacity for the compact state dimension; selector Z) are verified
// per draw on the device.  A draw that violates them is left untouched and flagged with
// DSGE_ST_INTERNAL_RERUN; the host then runs the general kalman_kernel on flagged draws only.
//
// The p x p inverse is a Gauss-Jordan sweep held in registers: lane l owns F[l>>3][l&7];
// pivot row/column travel by ds_bpermute, the pivot by v_readlane.  log det F is the product
// of the pivots, accumulated as (mantissa, exponent) over ALL time steps and turned into a
// logarithm once per draw.
#pragma once
#include "dsge_device.hpp"

#include "../../include/dsge_hip.h"

namespace dsge {

template <int BS>
struct Kf2Smem {
  static constexpr int NP = Tile<BS>::NP, LDM = Tile<BS>::LD;
  // doubles: PZt/K/V NP*8 each, Fi 64, Tc NP*LDM, Pc s_cap*LDM, Wc s_cap*LDM, av NP, af NP,
  // vv/dd/hh/zv 8 each, trash 64; ints: perm NP, zpos 8
  __host__ __device__ static constexpr size_t doubles(int s_cap, bool dense_z = false) {
    return 3 * (size_t)NP * 10 + 64 + (size_t)NP * LDM + 2 * (size_t)s_cap * LDM + 2 * NP + 32 + 64 + NP / 2 + 4 +
           (dense_z ? (size_t)NP * LDM + 8 * (size_t)LDM : 0);  // dense Z: full P and Z (8 x LDM) in LDS
  }
  static size_t bytes(int s_cap, bool dense_z = false) { return sizeof(double) * doubles(s_cap, dense_z); }
};

// hand-off record of kalman_sel_kernel -> kalman_tail_kernel (doubles per draw)
constexpr int KT_T = 0;              // 32 x 32   transition, states-first ordering
constexpr int KT_K = KT_T + 1024;    // 32 x 8    gain
constexpr int KT_FI = KT_K + 256;    // 8 x 8     F^-1
constexpr int KT_A = KT_FI + 64;     // 32        predicted state of the next step
constexpr int KT_ZV = KT_A + 32;     // 8         selector values
constexpr int KT_DD = KT_ZV + 8;     // 8         observation intercept
constexpr int KT_ZP = KT_DD + 8;     // 8         selected positions (as doubles)
constexpr int KT_SC = KT_ZP + 8;     // 16        m, s, t, mask, n_obs, step_mant, step_exp, quad_sum, quad_comp, ld_mant, ld_exp, n_ll, steady_step, n_obs_entries
constexpr int KT_REC = KT_SC + 16;

__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, src_lane);
  hi = __builtin_amdgcn_readlane(hi, src_lane);
  return __hiloint2double(hi, lo);
}

// Register-blocked product with compile-time strides, software-pipelined by hand: two operand
// register sets ping-pong, the LDS loads of step k+1 are issued BEFORE the 25 FMAs of step k and
// scheduling barriers keep hipcc from sinking them below the FMAs (it otherwise does, exposing the
// full LDS latency every step).
template <int BS, bool TB, int LDA, int LDB>
__device__ __forceinline__ void mm_acc_p(double (&acc)[BS][BS], const double* A, const double* B, int K, int lr,
                                         int lc) {
  const double* a0p = A + lr * BS * LDA;
  const double* b0p = TB ? (B + lc * BS * LDB) : (B + lc * BS);
  double a0[BS], b0[BS], a1[BS], b1[BS];
#define MM_LOAD(a, b, k)                                                              \
  do {                                                                                \
    _Pragma("unroll") for (int i = 0; i < BS; ++i) a[i] = a0p[i * LDA + (k)];         \
    _Pragma("unroll") for (int j = 0; j < BS; ++j) b[j] = TB ? b0p[j * LDB + (k)] : b0p[(k)*LDB + j]; \
  } while (0)
#define MM_FMA(a, b)                                                                  \
  do {                                                                                \
    _Pragma("unroll") for (int i = 0; i < BS; ++i)                                    \
      _Pragma("unroll") for (int j = 0; j < BS; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]); \
  } while (0)
  if (K <= 0) return;
  MM_LOAD(a0, b0, 0);
  int k = 0;
  for (; k + 2 <= K; k += 2) {
    MM_LOAD(a1, b1, k + 1);
    __builtin_amdgcn_sched_barrier(0);
    MM_FMA(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    const int kn = (k + 2 < K) ? k + 2 : K - 1;
    MM_LOAD(a0, b0, kn);
    __builtin_amdgcn_sched_barrier(0);
    MM_FMA(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (k < K) MM_FMA(a0, b0);
#undef MM_LOAD
#undef MM_FMA
}

typedef double v4f64 __attribute__((ext_vector_type(4)));

// One 16 x 16 output tile of a product on the FP64 matrix core: K4/4 x v_mfma_f64_16x16x4_f64.
//   TB = true : D = A[0:16, 0:K] B[0:16, 0:K]'      TB = false : D = A[0:16, 0:K] B[0:K, 0:16]
// Fragment layout (probed on gfx950, tools/mfma_probe): lane l supplies A[l & 15][k0 + (l >> 4)] and
// B[k0 + (l >> 4)][l & 15]; it receives D[(l >> 4) + 4 r][l & 15], r = 0..3.  Rows >= a_rows / b_rows and
// k >= K are fed as zeros (the operands are read with predicates, nothing outside the buffers is touched).
template <bool TB, int KMAX>
__device__ __forceinline__ v4f64 mfma_tile_16(const double* A, int lda, int a_rows, const double* B, int ldb, int b_rows,
                                              int K, int lane) {
  const int r16 = lane & 15, kq = lane >> 4;
  // every operand of the (at most KMAX / 4) steps is loaded up front from a clamped, always valid address and zeroed
  // by a select afterwards (no exec-mask branches); two accumulators halve the dependent MFMA chain
  const int ra = r16 < a_rows ? r16 : a_rows - 1, rb = r16 < b_rows ? r16 : b_rows - 1;
  double a[KMAX / 4], b[KMAX / 4];
#pragma unroll
  for (int q = 0; q < KMAX / 4; ++q) {
    const int k = 4 * q + kq, kc = k < K ? k : K - 1;
    const double av = A[ra * lda + kc];
    const double bv = TB ? B[rb * ldb + kc] : B[kc * ldb + r16];
    a[q] = (k < K && r16 < a_rows) ? av : 0.0;
    b[q] = (k < K && (!TB || r16 < b_rows)) ? bv : 0.0;
  }
  v4f64 c0 = {0.0, 0.0, 0.0, 0.0}, c1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int q = 0; q < KMAX / 4; q += 2) {
    if (4 * q < K) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], c0, 0, 0, 0);
    if (q + 1 < KMAX / 4 && 4 * (q + 1) < K) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q + 1], b[q + 1], c1, 0, 0, 0);
  }
  return c0 + c1;
}

// Elements of an (nr x nc) product outside the 16 x 16 core tile, on the VALU: rows >= 16 (all columns) and
// rows < 16 with columns >= 16.  dst[i][j] = sum_{k<K} A[i][k] * (TB ? B[j][k] : B[k][j]).
template <bool TB>
__device__ __forceinline__ void product_fringe(double* dst, int ldd, const double* A, int lda, const double* B, int ldb,
                                               int nr, int nc, int K, int lane) {
  const int r_hi = nr > 16 ? nr - 16 : 0, c_hi = nc > 16 ? nc - 16 : 0, r_lo = nr < 16 ? nr : 16;
  const int n1 = r_hi * nc, n2 = r_lo * c_hi;
  for (int idx = lane; idx < n1 + n2; idx += 64) {
    int i, j;
    if (idx < n1) {
      i = 16 + idx / nc;
      j = idx - (i - 16) * nc;
    } else {
      const int e = idx - n1;
      i = e / c_hi;
      j = 16 + (e - i * c_hi);
    }
    double s0 = 0.0, s1 = 0.0;
    int k = 0;
    for (; k + 1 < K; k += 2) {
      s0 = fma(A[i * lda + k], TB ? B[j * ldb + k] : B[k * ldb + j], s0);
      s1 = fma(A[i * lda + k + 1], TB ? B[j * ldb + k + 1] : B[(k + 1) * ldb + j], s1);
    }
    if (k < K) s0 = fma(A[i * lda + k], TB ? B[j * ldb + k] : B[k * ldb + j], s0);
    dst[i * ldd + j] = s0 + s1;
  }
}

// SEL = true : selector design matrix (P Z' and F are gathers)
// SEL = false: dense Z (p <= 8): P Z' is a register-block product against the full P kept in LDS
// MF = true : the two prediction products of every full step run on the FP64 matrix core (16 x 16 core tile by
//             v_mfma_f64_16x16x4_f64, the <= 8 fringe rows/columns on the VALU); needs 16 <= NP <= 24 (BS = 2, 3)
// Two waves per SIMD (a 256-register budget) pay off up to the 24-wide tile; the 32-wide one spills 440 B at that budget and
// is faster with one wave per SIMD (n = 56: 3.3 -> 2.7 ms per 4096 draws); for the 24-wide tile one wave per SIMD makes the
// step only 8 % shorter and the launch 30 % longer.
constexpr int KSEL_TWO_WAVES_MAX_BS = 3;
// TAIL = true : the instance that can hand the steady, constant-mask tail of the sample to kalman_tail_kernel (its own
//               instance because the hand-off code costs registers -- and spills -- in a kernel that sits at the 256 limit)
template <int BS, bool SEL, bool MF = false, bool TAIL = false>
__global__ __launch_bounds__(64, (BS == 1 ? 3 : (BS <= KSEL_TWO_WAVES_MAX_BS ? 2 : 1))) void kalman_sel_kernel(
    const double* __restrict__ T, const double* __restrict__ RQR, const double* __restrict__ P0,
    const double* __restrict__ Z, int z_batched, const double* __restrict__ dvec, int d_batched,
    const double* __restrict__ Hdiag, int h_batched, const double* __restrict__ y, int batch, int m_full, int p,
    int T_len, int s_cap, FilterConv cv, double missing_fill, double steady_tol, double* __restrict__ logp_out,
    int32_t* __restrict__ status, long long* __restrict__ dbg, int rerun_only, int32_t* __restrict__ steady_at,
    double* __restrict__ tail_rec, int32_t* __restrict__ tail_flag, const int32_t* __restrict__ tail_from,
    const int32_t* __restrict__ order) {
  constexpr int NP = Kf2Smem<BS>::NP, LDM = Kf2Smem<BS>::LDM;
  constexpr int PS = 10;  // row stride of the NP x 8 panels: 80 B keeps 16-byte alignment and spreads rows over all banks
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* PZt = smem;                // NP x 8     P Z'   (unmasked)     [16-byte aligned block first]
  double* Ks = PZt + NP * PS;        // NP x 8     K = P Zm' Finv
  double* Vs = Ks + NP * PS;         // NP x 8     P Zm' + jitter K
  double* Fi = Vs + NP * PS;         // 8 x 8      Finv
  double* Tc = Fi + 64;              // NP x LDM   transition in the states-first ordering (columns < s)
  double* Pc = Tc + NP * LDM;        // s_cap x LDM  P+ restricted to the state block
  double* Wc = Pc + s_cap * LDM;     // s_cap x LDM  W = Pc Tc'
  double* av = Wc + s_cap * LDM;     // NP         predicted state
  double* af = av + NP;              // NP         filtered state
  double* vv = af + NP;              // 8 innovation
  double* dd = vv + 8;               // 8 obs intercept
  double* hh = dd + 8;               // 8 diag(H)
  double* zv = hh + 8;               // 8 selector values
  double* trash = zv + 8;            // 64: sink for the stores of lanes that own no observed column
  int* perm = (int*)(trash + 64);    // NP: position -> original state index (states first)
  int* zpos = perm + NP;             // 8: position of the state each observation selects
  double* Pf = (double*)(zpos + 8);  // dense Z only: NP x LDM full P (operand of P Z')
  double* Zs = Pf + NP * LDM;        // dense Z only: 8 x LDM design matrix in the states-first ordering
  const int lane = threadIdx.x, lr = lane >> 3, lc = lane & 7;
  const int fo = lane >> 3, fq = lane & 7;  // owner of F[fo][fq]
  const double LN2PI = 1.8378770664093453, LN2 = 0.6931471805599453;

  // `order` (optional): workgroup b processes draw order[b] -- the launch's makespan is set by its slowest draws (late or
  // no steady state: up to T_len full steps on one wavefront), so the caller dispatches the likely slow ones first
  // (kalman_order_kernel).  Results do not depend on the order: every draw writes logp[draw], status[draw].
  if (rerun_only && rerun_pass_is_empty(status, batch, order)) return;
  for (int bi = blockIdx.x; bi < batch; bi += gridDim.x) {
    const int draw = order ? order[bi] : bi;
    const int32_t st_in = status[draw];
    if (rerun_only) {
      // later pass of the tile-size cascade: only the draws an earlier (smaller) instance flagged
      if (st_in != DSGE_ST_INTERNAL_RERUN) continue;
    } else if (st_in != 0) {
      if (lane == 0) logp_out[draw] = -INFINITY;
      continue;
    }
    const size_t off = (size_t)draw * m_full * m_full;
    wave_sync();
    for (int idx = lane; idx < (int)Kf2Smem<BS>::doubles(s_cap, !SEL); idx += 64) smem[idx] = 0.0;

    // ---- exact state-space reduction.  S = variables with a non-zero column in T (the states),
    // O = variables some observation loads on.  A variable outside U = S u O neither feeds back into
    // the recursion (its column of T is exactly zero) nor is observed, so dropping it changes nothing
    // in the retained entries of a, P, F, K: the filter runs on the u = |U| retained variables, ordered
    // states first, then the observed non-states.  (One lane per original variable, m_full <= 64.)
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
    bool ok = true;
    for (int o = 0; o < p; ++o) {
      const double zl = (lane < m_full) ? Zg[(size_t)o * m_full + lane] : 0.0;
      const unsigned long long b = __ballot(zl != 0.0);
      if (SEL && (__popcll(b) != 1 || ((used & b) != 0ull))) ok = false;
      used |= b;
      obsmask |= b;
    }
    const unsigned long long extra = obsmask & ~colmask;  // observed non-states
    const int s = __popcll(colmask);
    const int m = s + __popcll(extra);  // dimension of the reduced filter
    ok = ok && (s <= s_cap) && (m <= NP);
    int my_pos = -1;  // position of original variable `lane` in the reduced ordering (-1 = dropped)
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
      if (SEL) {
        if (zl != 0.0) {  // the single owner lane of this observation
          zpos[o] = (my_pos >= 0 && my_pos < NP) ? my_pos : 0;
          zv[o] = zl;
        }
      } else if (my_pos >= 0 && my_pos < NP) {
        Zs[o * LDM + my_pos] = zl;  // design matrix restricted to the retained variables
      }
    }
    if (!ok) {
      if (lane == 0) status[draw] = DSGE_ST_INTERNAL_RERUN;
      continue;
    }
    if (rerun_only && lane == 0) status[draw] = 0;
    wave_sync();
    // ---- load everything in the states-first ordering (a consistent permutation of the state
    // vector leaves the likelihood unchanged): element (r,c) <- original (perm[r], perm[c])
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
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        const bool in = pr[i] >= 0 && pcx[j] >= 0;
        const size_t g = in ? (size_t)pr[i] * m_full + pcx[j] : 0;
        const double tv = in ? T[off + g] : 0.0;
        Qb[i][j] = in ? RQR[off + g] : 0.0;
        Pb[i][j] = (in && P0) ? P0[off + g] : 0.0;
        Tb0[i][j] = tv;
        Tc[(lr * BS + i) * LDM + lc * BS + j] = tv;  // columns >= s are exactly zero
      }
    const bool in_state_block = (lr * BS < s) && (lc * BS < s);
    if (!P0) {
      // ---- stationary covariance of the REDUCED model, P0 = dlyap(T, RQR')[U,U], by doubling:
      //   P <- P + A_k[:,S] P[S,S] A_k[:,S]',  A_{k+1} = A_k[:,S] A_k[S,:],  A_0 = T[U,:]
      // (statespace.py:814-815; the recursion closes on U because the columns of T outside S vanish:
      // (T M T')[U,U] = T[U,S] M[S,S] T[U,S]').  Same products as the prediction step below.
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) Pb[i][j] = Qb[i][j];
      bool lyap_ok = false;
      for (int itl = 0; itl < 64; ++itl) {
        wave_sync();
        if (in_state_block) blk_store_lds<BS>(Pb, Pc, LDM, lr, lc);
        wave_sync();
        if (lr * BS < s) {
          double Wb[BS][BS];
          blk_zero<BS>(Wb);
          mm_acc_p<BS, true, LDM, LDM>(Wb, Pc, Tc, s, lr, lc);  // P[S,S] A_k'
          blk_store_lds<BS>(Wb, Wc, LDM, lr, lc);
        }
        double Ab[BS][BS];
        blk_zero<BS>(Ab);
        mm_acc_p<BS, false, LDM, LDM>(Ab, Tc, Tc, s, lr, lc);  // A_k[:,S] A_k[S,:] (states come first)
        wave_sync();
        double Xb[BS][BS];
        blk_zero<BS>(Xb);
        mm_acc_p<BS, false, LDM, LDM>(Xb, Tc, Wc, s, lr, lc);
        wave_sync();
        blk_store_lds<BS>(Ab, Tc, LDM, lr, lc);
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
        if (!(dmax == dmax) || !(pmax < 1e300)) break;  // NaN / overflow: rho(T) >= 1
        if (dmax <= 1e-17 * pmax) {
          lyap_ok = true;
          break;
        }
      }
      wave_sync();
      blk_store_lds<BS>(Tb0, Tc, LDM, lr, lc);  // the transition itself again
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
    // destination of my P Z' contributions: PZt[row][obs] for observed columns, a private sink else
    int pz_dst[BS];
#pragma unroll
    for (int j = 0; j < BS; ++j) pz_dst[j] = ocol[j];
    // P Z' for the coming step: a column gather (selector) or a product against the full P (dense Z)
#define STORE_PZT()                                                                                   \
  do {                                                                                                \
    if (SEL) {                                                                                        \
      _Pragma("unroll") for (int j = 0; j < BS; ++j) _Pragma("unroll") for (int i = 0; i < BS; ++i) { \
        double* dst = (pz_dst[j] >= 0) ? &PZt[(lr * BS + i) * PS + pz_dst[j]] : &trash[lane];          \
        *dst = zcol[j] * Pb[i][j];                                                                    \
      }                                                                                               \
    } else {                                                                                          \
      blk_store_lds<BS>(Pb, Pf, LDM, lr, lc);                                                         \
      wave_sync();                                                                                    \
      if (lc < p) {                                                                                   \
        double pacc[BS];                                                                              \
        _Pragma("unroll") for (int i = 0; i < BS; ++i) pacc[i] = 0.0;                                 \
        for (int jj = 0; jj < m; ++jj) {                                                              \
          const double zz = Zs[lc * LDM + jj];                                                        \
          _Pragma("unroll") for (int i = 0; i < BS; ++i)                                              \
              pacc[i] = fma(Pf[(lr * BS + i) * LDM + jj], zz, pacc[i]);                               \
        }                                                                                             \
        _Pragma("unroll") for (int i = 0; i < BS; ++i) PZt[(lr * BS + i) * PS + lc] = pacc[i];        \
      }                                                                                               \
    }                                                                                                 \
  } while (0)
    STORE_PZT();
    const int my_zpos = (fo < p) ? zpos[fo] : 0;
    const double my_zv = (fo < p) ? zv[fo] : 0.0;
    const int v_zpos = (lane < p) ? zpos[lane] : 0;
    const double v_zv = (lane < p) ? zv[lane] : 0.0, v_dd = (lane < 8) ? dd[lane & 7] : 0.0;
    wave_sync();

    double quad_sum = 0.0, quad_comp = 0.0;  // Kahan sum of v' Finv v over observed steps
    double ld_mant = 1.0;                    // prod of pivots = mant * 2^exp
    long long ld_exp = 0;
    long long n_ll_steps = 0, n_obs_entries = 0;  // (FilterConv::ll_terms)
    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long tk_start = dbg ? clock64() : 0;
    // ---- steady-state switch.  The covariance recursion P_{t+1|t} = f(P_{t|t-1}; mask_t) does not
    // depend on the data; once it has reached its fixed point for the current missing-data mask
    // (max|P_{t+1|t} - P_{t|t-1}| <= steady_tol * max|P|, i.e. rounding level for the default 1e-14),
    // F^-1, K and det F of the last full step are reused and only the O(m p + m s) mean recursion
    // runs.  A step whose mask differs falls back to the full update from the current P.
    double kr_ss[8], av_reg = 0.0;
    int steady_step = -1;
    bool handed_off = false;  // the tail kernel finishes the draw and writes its logp
    // selector Z and a small tile: the steady-state steps run out of registers only
    constexpr bool REG_SS = SEL && (BS <= 4);
#pragma unroll
    for (int o = 0; o < 8; ++o) kr_ss[o] = 0.0;
    double yt_next = (lane < p && T_len > 0) ? y[lane] : 0.0;
    for (int t = 0; t < T_len; ++t) {
      long long tk0 = dbg ? clock64() : 0;
      // ---- (a) missing-data mask ------------------------------------------------------
      const double yt = yt_next;  // fetched one step ahead: the global-load latency is off the chain
      yt_next = (lane < p && t + 1 < T_len) ? y[(size_t)(t + 1) * p + lane] : 0.0;
      const bool obs = (lane < p) && (yt == yt) && (yt != missing_fill);
      const unsigned long long omask = __ballot(obs);
      const int n_obs = __popcll(omask);
      bool steady = false;
      // scale of the steady-state test: max |P_{t|t-1}| (the filtered block P+[S,S] that is compared can be
      // orders of magnitude smaller than P when the observations are precise; its rounding noise is not)
      double pscale = 0.0;
      if (steady_tol > 0.0) {
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) pscale = nanmax(pscale, fabs(Pb[i][j]));
      }
      // ---- (b) the innovation ------------------------------------------------------------
      double v_own = 0.0;
      if (lane < p) {
        double za;
        if (SEL) {
          za = v_zv * av[v_zpos];
        } else {
          za = 0.0;
          for (int jj = 0; jj < m; ++jj) za = fma(Zs[lane * LDM + jj], av[jj], za);
        }
        v_own = (obs ? yt : 0.0) - (((obs || !cv.mask_d) ? v_dd : 0.0) + (obs ? 1.0 : 0.0) * za);
      }
      if (lane < 8) vv[lane] = v_own;
      // ---- (b') F[fo][fq]: lane (fo,fq) of the 8 x 8 grid ---------------------------------------
      double step_mant = 1.0;
      int step_exp = 0;
      double f;
      {
        const double wo = (double)((omask >> fo) & 1ull), wq = (double)((omask >> fq) & 1ull);
        if (fo < p && fq < p) {
          if (SEL) {
            f = wo * wq * my_zv * PZt[my_zpos * PS + fq];
          } else {
            double fa = 0.0, fb = 0.0;  // F = Z (P Z'): two chains over the state index
            int jj = 0;
            for (; jj + 1 < m; jj += 2) {
              fa = fma(Zs[fo * LDM + jj], PZt[jj * PS + fq], fa);
              fb = fma(Zs[fo * LDM + jj + 1], PZt[(jj + 1) * PS + fq], fb);
            }
            if (jj < m) fa = fma(Zs[fo * LDM + jj], PZt[jj * PS + fq], fa);
            f = wo * wq * (fa + fb);
          }
          if (fo == fq) f += wo * hh[fo] + cv.jit_F;
        } else {
          f = (fo == fq) ? 1.0 : 0.0;
        }
      }
      // ---- (c) Finv by in-register Gauss-Jordan (SPD: no pivoting).  (A one-row-per-lane variant that moves the
      // pivot row through SGPRs instead of two ds_bpermute was measured slower: 4.1 k vs 3.3 k cycles.) ----
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
          step_mant *= frexp(piv, &e);
          step_exp += e;
        }
      }
      Fi[lane] = f;
      // v' Finv v: the innovation entries come from lanes 0..7 by shuffle
      const double qp = wave_sum_dpp(f * __shfl(v_own, fo, 64) * __shfl(v_own, fq, 64));
      if (n_obs > 0) {
        const double yk = qp - quad_comp;
        const double tk = quad_sum + yk;
        quad_comp = (tk - quad_sum) - yk;
        quad_sum = tk;
        int e;
        ld_mant = frexp(ld_mant * step_mant, &e);
        ld_exp += (long long)e + step_exp;
        ++n_ll_steps;
        n_obs_entries += n_obs;
      }
      wave_sync();  // #1
      if (dbg) {
        const long long tk1 = clock64();
        ph[0] += tk1 - tk0;
        tk0 = tk1;
      }
      // ---- (d) K = (P Zm') Finv, V = P Zm' + jitter K, a+ = a + K v -----------------------------
      // all 64 lanes: lane = (row within a group of 16, observation pair); the four lanes of a row add their
      // contributions to a+ with two quad_perm DPP steps
      {
        const int o2 = lane & 3, i16 = lane >> 2;
        double2 fi2[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) fi2[q] = *reinterpret_cast<const double2*>(&Fi[q * 8 + 2 * o2]);
        const double2 vp = *reinterpret_cast<const double2*>(&vv[2 * o2]);
        const bool on0 = (omask >> (2 * o2)) & 1ull, on1 = (omask >> (2 * o2 + 1)) & 1ull;
#pragma unroll
        for (int pass = 0; pass < (NP + 15) / 16; ++pass) {
          const int i = i16 + 16 * pass;
          const bool row_ok = i < m;
          const int ir = row_ok ? i : 0;
          double k0 = 0.0, k1 = 0.0;
#pragma unroll
          for (int q2 = 0; q2 < 4; ++q2) {
            const double2 t2 = *reinterpret_cast<const double2*>(&PZt[ir * PS + 2 * q2]);
            const double pa = ((omask >> (2 * q2)) & 1ull) ? t2.x : 0.0;
            const double pb2 = ((omask >> (2 * q2 + 1)) & 1ull) ? t2.y : 0.0;
            k0 = fma(pa, fi2[2 * q2].x, k0);
            k1 = fma(pa, fi2[2 * q2].y, k1);
            k0 = fma(pb2, fi2[2 * q2 + 1].x, k0);
            k1 = fma(pb2, fi2[2 * q2 + 1].y, k1);
          }
          const double2 pzp = *reinterpret_cast<const double2*>(&PZt[ir * PS + 2 * o2]);
          double part = fma(k0, vp.x, k1 * vp.y);
          part += dpp_move_f64<0xB1, 0xf>(part);  // quad_perm [1,0,3,2]
          part += dpp_move_f64<0x4E, 0xf>(part);  // quad_perm [2,3,0,1]
          if (row_ok) {
            *reinterpret_cast<double2*>(&Ks[i * PS + 2 * o2]) = double2{k0, k1};
            *reinterpret_cast<double2*>(&Vs[i * PS + 2 * o2]) =
                double2{fma(cv.jit_V, k0, on0 ? pzp.x : 0.0), fma(cv.jit_V, k1, on1 ? pzp.y : 0.0)};
            if (o2 == 0) af[i] = av[i] + part;
          }
        }
      }
      wave_sync();  // #2
      if (dbg) {
        const long long tk1 = clock64();
        ph[1] += tk1 - tk0;
        tk0 = tk1;
      }
      // ---- (e) P+ = P - K V' + jitter I (register blocks); state block -> LDS -------------
      if constexpr (MF) {  // the MFMA path reuses the K / V panels as scratch in (f): keep this lane's row of K
#pragma unroll
        for (int q = 0; q < 8; ++q) kr_ss[q] = (lane < m) ? Ks[lane * PS + q] : 0.0;
      }
      {
        // two-stage software pipeline over the four observation pairs
        double2 ka[BS], vb[BS], kan[BS], vbn[BS];
#pragma unroll
        for (int i = 0; i < BS; ++i) ka[i] = *reinterpret_cast<const double2*>(&Ks[(lr * BS + i) * PS]);
#pragma unroll
        for (int j = 0; j < BS; ++j) vb[j] = *reinterpret_cast<const double2*>(&Vs[(lc * BS + j) * PS]);
#pragma unroll
        for (int o2 = 0; o2 < 4; ++o2) {
          if (o2 < 3) {
#pragma unroll
            for (int i = 0; i < BS; ++i)
              kan[i] = *reinterpret_cast<const double2*>(&Ks[(lr * BS + i) * PS + 2 * (o2 + 1)]);
#pragma unroll
            for (int j = 0; j < BS; ++j)
              vbn[j] = *reinterpret_cast<const double2*>(&Vs[(lc * BS + j) * PS + 2 * (o2 + 1)]);
          }
#pragma unroll
          for (int i = 0; i < BS; ++i)
#pragma unroll
            for (int j = 0; j < BS; ++j) {
              Pb[i][j] = fma(-ka[i].x, vb[j].x, Pb[i][j]);
              Pb[i][j] = fma(-ka[i].y, vb[j].y, Pb[i][j]);
            }
#pragma unroll
          for (int i = 0; i < BS; ++i) ka[i] = kan[i];
#pragma unroll
          for (int j = 0; j < BS; ++j) vb[j] = vbn[j];
        }
      }
      if (lr == lc) {
#pragma unroll
        for (int i = 0; i < BS; ++i)
          if (lr * BS + i < m) Pb[i][i] += cv.jit_P;
      }
      // steady-state test on the filtered state block: P_{t+1|t} depends on P+ only through P+[S,S], so the
      // covariance recursion has reached its fixed point once that block stops moving.  The previous
      // step's block is still in LDS (Pc): compare before overwriting it.
      if (steady_tol > 0.0) {
        double dmax = 0.0;
        if (in_state_block) {
#pragma unroll
          for (int i = 0; i < BS; ++i)
#pragma unroll
            for (int j = 0; j < BS; ++j)
              dmax = nanmax(dmax, fabs(Pb[i][j] - Pc[(lr * BS + i) * LDM + lc * BS + j]));
        }
        // non-negative doubles order like their bit patterns; a NaN compares above everything
        const double dm = __longlong_as_double((long long)wave_max_u64((unsigned long long)__double_as_longlong(dmax)));
        const double pm = __longlong_as_double((long long)wave_max_u64((unsigned long long)__double_as_longlong(pscale)));
        steady = (t > 0) && (dm <= steady_tol * pm);
      }
      if (in_state_block) blk_store_lds<BS>(Pb, Pc, LDM, lr, lc);
      wave_sync();  // #3
      if (dbg) {
        const long long tk1 = clock64();
        ph[2] += tk1 - tk0;
        tk0 = tk1;
      }
      // ---- (f) predict: a = Tc a+[:s];  W = Pc Tc';  X = Tc W;  P = sym(X) + RQR ---------
      if (lane < m) {
        double s0 = 0.0, s1 = 0.0;
        int kk = 0;
        for (; kk + 1 < s; kk += 2) {
          s0 = fma(Tc[lane * LDM + kk], af[kk], s0);
          s1 = fma(Tc[lane * LDM + kk + 1], af[kk + 1], s1);
        }
        if (kk < s) s0 = fma(Tc[lane * LDM + kk], af[kk], s0);
        av[lane] = s0 + s1;
        av_reg = s0 + s1;
      }
      if constexpr (MF) {
        // W = Pc Tc' (s x m) -> Wc: core tile on the matrix core, fringe on the VALU
        {
          const v4f64 c = mfma_tile_16<true, NP>(Pc, LDM, s, Tc, LDM, m, s, lane);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = (lane >> 4) + 4 * r;
            if (row < s) Wc[row * LDM + (lane & 15)] = c[r];
          }
          product_fringe<true>(Wc, LDM, Pc, LDM, Tc, LDM, s, m, s, lane);
        }
        wave_sync();  // #4
        if (dbg) {
          const long long tk1 = clock64();
          ph[3] += tk1 - tk0;
          tk0 = tk1;
        }
        // X = Tc W (m x m) -> Xs, which reuses the dead P Z' / K / V panels
        double* Xs = PZt;
        {
          const v4f64 c = mfma_tile_16<false, NP>(Tc, LDM, m, Wc, LDM, s, s, lane);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = (lane >> 4) + 4 * r;
            if (row < m) Xs[row * LDM + (lane & 15)] = c[r];
          }
          // fringe: only the rows >= 16 (all columns); X = T P+ T' is symmetric, so the entries (r < 16, c >= 16) are
          // read from their mirror images below (sym() only averages rounding noise anyway)
          product_fringe<false>(Xs, LDM, Tc, LDM, Wc, LDM, m, m < 16 ? m : 16, s, lane);
          if (m > 16) {
            const int nrest = (m - 16) * (m - 16);
            for (int idx = lane; idx < nrest; idx += 64) {
              const int i = 16 + idx / (m - 16), j = 16 + idx % (m - 16);
              double s0 = 0.0;
              for (int k = 0; k < s; ++k) s0 = fma(Tc[i * LDM + k], Wc[k * LDM + j], s0);
              Xs[i * LDM + j] = s0;
            }
          }
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) {
            const int r = lr * BS + i, c = lc * BS + j;
            const bool in = r < m && c < m;
            // of a mixed pair (one index < 16, the other >= 16) only the element with the row >= 16 was computed:
            // it stands for itself and for its mirror image
            const bool mixed = (r < 16) != (c < 16);
            const int r1 = mixed ? (r > c ? r : c) : r, c1 = mixed ? (r > c ? c : r) : c;
            const double x = in ? Xs[r1 * LDM + c1] : 0.0, xt = in ? (mixed ? x : Xs[c1 * LDM + r1]) : 0.0;
            Pb[i][j] = 0.5 * (x + xt) + Qb[i][j];
          }
        wave_sync();
        for (int idx = lane; idx < 3 * NP * PS; idx += 64) PZt[idx] = 0.0;  // the panels again (P Z' needs its zeros)
        wave_sync();
      } else {
      if (lr * BS < s) {
        double Wb[BS][BS];
        blk_zero<BS>(Wb);
        mm_acc_p<BS, true, LDM, LDM>(Wb, Pc, Tc, s, lr, lc);
        blk_store_lds<BS>(Wb, Wc, LDM, lr, lc);
      }
      wave_sync();  // #4
      if (dbg) {
        const long long tk1 = clock64();
        ph[3] += tk1 - tk0;
        tk0 = tk1;
      }
      {
        double Xb[BS][BS];
        blk_zero<BS>(Xb);
        mm_acc_p<BS, false, LDM, LDM>(Xb, Tc, Wc, s, lr, lc);
        const int src = (lc << 3) | lr;  // lane holding the transposed block
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) {
            const double xt = __shfl(Xb[j][i], src, 64);
            Pb[i][j] = 0.5 * (Xb[i][j] + xt) + Qb[i][j];
          }
      }
      }
      // ---- P Z' for the next step -------------------------------------------------------------
      STORE_PZT();
      wave_sync();  // #5
      if (dbg) {
        const long long tk1 = clock64();
        ph[4] += tk1 - tk0;
        tk0 = tk1;
      }
      if (!steady) continue;
      // ==== steady-state steps: mean recursion only, while the missing-data mask stays the same ====
      if (steady_step < 0) steady_step = t + 1;
      if constexpr (REG_SS && !MF && TAIL) {
        // ---- hand-off: once the covariance is frozen AND the missing-data mask no longer changes until the end of the
        // sample (t >= *tail_from, found by kalman_mask_scan_kernel), the remaining steps are a linear recursion in the
        // mean alone.  kalman_tail_kernel runs it two steps at a time as one matrix-vector product per pair -- in a launch of
        // its own, because the rows it keeps in registers do not fit next to this kernel's 256 (dsge_kalman_tail.hpp).
        // This kernel only writes the record and leaves the time loop.
        if (tail_rec && t + 2 < T_len && t >= *tail_from) {
          double* rec = tail_rec + (size_t)draw * KT_REC;
          for (int idx = lane; idx < NP * NP; idx += 64) {
            const int i = idx / NP, c = idx - i * NP;
            rec[KT_T + i * 32 + c] = (c < s) ? Tc[i * LDM + c] : 0.0;
          }
          for (int idx = lane; idx < NP * 8; idx += 64) rec[KT_K + idx] = Ks[(idx >> 3) * PS + (idx & 7)];
          rec[KT_FI + lane] = Fi[lane];
          if (lane < NP) rec[KT_A + lane] = (lane < m) ? av_reg : 0.0;
          if (lane < 8) {
            rec[KT_ZV + lane] = (lane < p) ? zv[lane] : 0.0;
            rec[KT_DD + lane] = (lane < p && (((omask >> lane) & 1ull) || !cv.mask_d)) ? dd[lane] : 0.0;  // (constant mask from here on)
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
            sc[7] = quad_sum;
            sc[8] = quad_comp;
            sc[9] = ld_mant;
            sc[10] = (double)ld_exp;
            sc[11] = (double)n_ll_steps;
            sc[12] = (double)steady_step;
            sc[13] = (double)n_obs_entries;
            tail_flag[draw] = 1;
          }
          handed_off = true;
          if (dbg) ph[5] += clock64() - tk0;
          break;
        }
      }
      if constexpr (REG_SS) {
        // register-only: this lane's row of Tc, rows of F^-1 in lanes 0..7, no LDS memory traffic
        double trow[NP], finv_row[8];
#pragma unroll
        for (int kk = 0; kk < NP; ++kk) trow[kk] = (lane < NP) ? Tc[lane * LDM + kk] : 0.0;  // columns >= s are zero
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          finv_row[q] = (lane < 8) ? Fi[lane * 8 + q] : 0.0;
          if constexpr (!MF) kr_ss[q] = (lane < m) ? Ks[lane * PS + q] : 0.0;
        }
        while (t + 1 < T_len) {
          const double yt_s = yt_next;
          const bool obs_s = (lane < p) && (yt_s == yt_s) && (yt_s != missing_fill);
          if (__ballot(obs_s) != omask) break;
          ++t;
          yt_next = (lane < p && t + 1 < T_len) ? y[(size_t)(t + 1) * p + lane] : 0.0;
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
          double part = v_s * (w0 + w1);  // lanes >= 8 hold finv_row = 0
          part += dpp_move_f64<0x111, 0xf>(part);
          part += dpp_move_f64<0x112, 0xf>(part);
          part += dpp_move_f64<0x114, 0xf>(part);
          const double qp = readlane_f64(part, 7);
          if (n_obs > 0) {
            const double yk = qp - quad_comp;
            const double tk = quad_sum + yk;
            quad_comp = (tk - quad_sum) - yk;
            quad_sum = tk;
            int e;
            ld_mant = frexp(ld_mant * step_mant, &e);
            ld_exp += (long long)e + step_exp;
            ++n_ll_steps;
            n_obs_entries += n_obs;
          }
          const double afi = a0 + a1;
          double s0 = 0.0, s1 = 0.0;
#pragma unroll
          for (int kk = 0; kk < NP; kk += 2) {
            s0 = fma(trow[kk], readlane_f64(afi, kk), s0);
            s1 = fma(trow[kk + 1], readlane_f64(afi, kk + 1), s1);
          }
          av_reg = (lane < m) ? s0 + s1 : 0.0;
          if (dbg) ++ph[6];
        }
        if (lane < m) av[lane] = av_reg;  // hand the predicted state back to the LDS copy
        wave_sync();
      } else {
        if constexpr (!MF) {
#pragma unroll
          for (int q = 0; q < 8; ++q) kr_ss[q] = (lane < m) ? Ks[lane * PS + q] : 0.0;
        }
        while (t + 1 < T_len) {
          const double yt_s = yt_next;
          const bool obs_s = (lane < p) && (yt_s == yt_s) && (yt_s != missing_fill);
          if (__ballot(obs_s) != omask) break;
          ++t;
          yt_next = (lane < p && t + 1 < T_len) ? y[(size_t)(t + 1) * p + lane] : 0.0;
          double v_s = 0.0;
          if (lane < p) {
            double za;
            if (SEL) {
              za = v_zv * av[v_zpos];
            } else {
              za = 0.0;
              for (int jj = 0; jj < m; ++jj) za = fma(Zs[lane * LDM + jj], av[jj], za);
            }
            v_s = (obs_s ? yt_s : 0.0) - (((obs_s || !cv.mask_d) ? v_dd : 0.0) + (obs_s ? 1.0 : 0.0) * za);
          }
          const double qp_s = wave_sum_dpp(f * __shfl(v_s, fo, 64) * __shfl(v_s, fq, 64));
          if (n_obs > 0) {
            const double yk = qp_s - quad_comp;
            const double tk = quad_sum + yk;
            quad_comp = (tk - quad_sum) - yk;
            quad_sum = tk;
            int e;
            ld_mant = frexp(ld_mant * step_mant, &e);
            ld_exp += (long long)e + step_exp;
            ++n_ll_steps;
            n_obs_entries += n_obs;
          }
          double afi = (lane < m) ? av[lane] : 0.0;
#pragma unroll
          for (int o = 0; o < 8; ++o) afi = fma(kr_ss[o], readlane_f64(v_s, o), afi);
          if (lane < m) af[lane] = afi;
          wave_sync();
          if (lane < m) {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int kk = 0;
            for (; kk + 3 < s; kk += 4) {
              s0 = fma(Tc[lane * LDM + kk], af[kk], s0);
              s1 = fma(Tc[lane * LDM + kk + 1], af[kk + 1], s1);
              s2 = fma(Tc[lane * LDM + kk + 2], af[kk + 2], s2);
              s3 = fma(Tc[lane * LDM + kk + 3], af[kk + 3], s3);
            }
            for (; kk < s; ++kk) s0 = fma(Tc[lane * LDM + kk], af[kk], s0);
            av[lane] = (s0 + s1) + (s2 + s3);
          }
          wave_sync();
          if (dbg) ++ph[6];
        }
      }
      if (dbg) ph[5] += clock64() - tk0;
    }
#undef STORE_PZT
    if (dbg && draw == 0 && lane == 0) {
      ph[7] = clock64() - tk_start;
      for (int k = 0; k < 8; ++k) dbg[k] = ph[k];
    }
    if (lane == 0 && !handed_off) {
      const double logdet = log(ld_mant) + (double)ld_exp * LN2;
      const double ll = -0.5 * (cv.ll_terms(n_ll_steps, n_obs_entries, p) * LN2PI + logdet + quad_sum);
      logp_out[draw] = ll;
      if (steady_at) steady_at[draw] = steady_step;
      if (!((ll == ll) && (fabs(ll) < 1.797e308))) status[draw] |= DSGE_ST_FILTER_NONFINITE;
    }
  }
}

}  // namespace dsge

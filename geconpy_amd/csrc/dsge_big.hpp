// Models with 65 .. 96 variables (round 4): one WORKGROUP per draw instead of one wavefront.
//
// The kernels of dsge_kernels.hpp / dsge_cr_*.hpp give every lane one row or one 8 x 8 register block of an n <= 64 matrix and
// keep all operands of cycle reduction in 64 lanes' registers and LDS.  The reference has no size limit
// (gEconpy/model/statespace.py:822-839; packaged examples of the sims_2024 class sit at the edge of 64), so this file restates
// the same recursion -- _cycle_reduction_core, gEconpy/solvers/cycle_reduction.py:127-183; the scan variant :246-294 -- for
// padded sizes NP = 80 and 96 with 640 / 512 threads per draw:
//
//   * elimination  X = A1^-1 [A0 | A2]  (:151-153, np.linalg.solve): Gauss-Jordan with partial pivoting on the augmented
//     n x 3n matrix, every thread holding a TR x TC register block of each of the three matrices (3 x 96 x 96 doubles = 221 KB:
//     registers are the only on-chip memory that holds them).  Pivoting is IMPLICIT -- rows never move; pivot column j's row is
//     broadcast through LDS, every other row eliminates against it, and the solution's row order is restored by the scatter
//     that writes X -- so a step costs two barriers, ~14 LDS reads and 3 TR TC FMAs per thread;
//   * the four products A0 X0, A0 X2, A2 X0, A2 X2 (:155-169) on the FP64 matrix core (v_mfma_f64_16x16x4_f64: 16 x 16 tiles,
//     K streamed from two LDS panels of NP x (NP + 2) doubles -- the odd-ish stride makes the 16 x 4 A fragment and the 4 x 16
//     B fragment of a half-wave land on 32 distinct bank pairs), tiles dealt round-robin to the wavefronts, the update of
//     A1, A1_hat (read-modify-write in the workgroup's L2-resident workspace) and the induced 1-norms of the stopping rule
//     (:171-179) in the epilogues;
//   * T = -A1_hat^-1 A and, from the same elimination, R = -A1_hat^-1 D (= -(C T + B)^-1 D at convergence, as in cr_kernel).
//
// The filter does not grow with n: T has non-zero columns only for state variables, so the state-space
// model restricted to F = {states} u {observed variables} is EXACT, and with |F| <= 64 the existing filter kernels run on the
// gathered model (big_compress_kernel).  |F| > 64 is refused with DSGE_ERR_TOO_LARGE.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dsge_hip.h"
#include "dsge_device.hpp"

namespace dsge {

typedef double big_v4f64 __attribute__((ext_vector_type(4)));

template <int NP_, int TR_, int TC_, int NBLK_ = 1>
struct BigCfg {
  static constexpr int NP = NP_, TR = TR_, TC = TC_, PB = NBLK_ * TC_;  // PB: pivots per elimination round
  static constexpr int GY = NP / TR, GX = NP / TC, NT = GY * GX, NW = NT / 64;
  static constexpr int LD = NP + 2, MT = NP / 16;
  static_assert(NP % 16 == 0 && NP % TR == 0 && NP % TC == 0 && NT % 64 == 0 && NT <= 1024 && NP <= 128, "tile grid");
  // LDS (doubles): left panel, right panel (NP x LD each) | column-sum partials (MT x NP) | reduction scratch (64) | the
  // elimination's broadcast buffers (the left panel is filled with -A0 BEFORE the elimination, from the register blocks).
  static constexpr int TPW = (MT * MT + NW - 1) / NW;  // 16 x 16 product tiles per wavefront
  // Workspace matrices have the row stride of the LDS panels (LD), so that a panel copy is LINEAR -- piece tid + q NT of the
  // matrix goes to piece tid + q NT of the panel: one base register and immediate offsets, no index arithmetic to hoist and
  // spill (the NP-stride version kept 70 loop-invariant addresses alive: 592 bytes of scratch per lane at NP = 96).
  static constexpr int MAT = NP * LD;                              // doubles per workspace matrix
  static constexpr int PF = (MAT / 2 + NT - 1) / NT;               // double2 per thread of one panel (prefetch registers)
  static constexpr int OFF_R = NP * LD, OFF_PART = 2 * NP * LD, OFF_RED = OFF_PART + MT * NP, OFF_EL = OFF_RED + 64;
  static constexpr size_t lds_bytes = (size_t)(OFF_EL + NP / 2 + 2 + 8) * 8;
  static constexpr size_t ws_doubles = 6 * (size_t)MAT;  // A0, A1, A2, A1_hat, X0, X2 per workgroup
  // elimination: pivcol int[NP] and the round's pivot rows int[TC] in the tail; panel [NP][TC], coefficients [NP][TC] and the
  // round's pivot rows [TC][3 NP] alias the RIGHT panel (free during an elimination; the left one already holds -A0)
  static constexpr int E_PIV = OFF_EL, E_PS = OFF_EL + NP / 2 + 2;
  static constexpr int E_PANEL = OFF_R, E_CV = OFF_R + NP * PB, E_ROW = OFF_R + 2 * NP * PB;
  static_assert(5 * NP * PB <= NP * LD && PB <= 8 && NP % PB == 0, "elimination buffers fit the right panel");
};

// ---- register blocks <-> the workgroup's padded NP x NP workspace matrices ---------------------------------------------------------
template <class Cfg>
__device__ __forceinline__ void big_tile_load(double (&t)[Cfg::TR][Cfg::TC], const double* __restrict__ G, int r0, int c0) {
#pragma unroll
  for (int i = 0; i < Cfg::TR; ++i)
#pragma unroll
    for (int jc = 0; jc < Cfg::TC; ++jc) t[i][jc] = G[(r0 + i) * Cfg::LD + c0 + jc];
}

// workspace matrix -> LDS panel (same layout), 16 bytes per thread and trip
template <class Cfg>
__device__ __forceinline__ void big_panel_load(double* __restrict__ buf, const double* __restrict__ G, int tid) {
  const double2* g2 = reinterpret_cast<const double2*>(G);
  double2* b2 = reinterpret_cast<double2*>(buf);
  for (int idx = tid; idx < Cfg::MAT / 2; idx += Cfg::NT) b2[idx] = g2[idx];
}

// ---- Gauss-Jordan elimination of [t1 | t0 | t2] with implicit partial pivoting, one block column (TC pivots) per round ----------------
// On exit t1 is (a row permutation of) the identity and physical row r of t0, t2 holds row pivcol[r] of t1_in^-1 [t0_in | t2_in];
// pivcol (int[NP], LDS) is valid after the closing barrier.  A zero pivot divides by zero: the NaN / Inf flows into the
// stopping rule's norm like LAPACK's singular-matrix error does in the reference (cycle_reduction.py:176-177).
//
// [First version, one pivot per round: column j published + arg-max by LDS atomic, barrier, pivot row published, barrier, rank-1
//  update -- 3.7 k cycles per pivot at n = 80 for 30 FMAs per thread, all of it barrier waits and the latency of short dependent
//  phases on ten wavefronts.]  A round now takes the TC columns of one block column:
//    1. their owners publish the n x TC panel;                                                                       barrier
//    2. wavefront 0 alone -- two rows per lane, no barrier inside -- eliminates the panel: per pivot a DPP arg-max over the rows
//       not used yet, the pivot row's panel entries by v_readlane, and for EVERY row i its coefficients over the round's pivot
//       rows:   row_i(new) = delta_i row_i(old) + sum_t cv_i[t] row_{p_t}(old),   delta_i = 0 for the round's pivot rows
//       (selecting p at step s: cv_p[s] += 1, cv_p *= 1 / pivot;  every other row: cv_i -= x_i[s] cv_p);            barrier
//    3. the owners of the TC pivot rows publish them (old values, all three matrices);                              barrier
//    4. every thread applies the rank-TC update to its register blocks.
// Three barriers and one serial section per TC pivots instead of two barriers per pivot.
__device__ __forceinline__ unsigned long long big_wave_max_u64(unsigned long long k) {
  unsigned lo = (unsigned)k, hi = (unsigned)(k >> 32);
#define BIG_DPP_STEP(CTRL)                                                                        \
  do {                                                                                            \
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, 0xf, 0xf, false); \
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, 0xf, 0xf, false); \
    const bool gt = ohi > hi || (ohi == hi && olo > lo);                                          \
    lo = gt ? olo : lo;                                                                           \
    hi = gt ? ohi : hi;                                                                           \
  } while (0)
  BIG_DPP_STEP(0xB1);   // quad_perm [1, 0, 3, 2]
  BIG_DPP_STEP(0x4E);   // quad_perm [2, 3, 0, 1]
  BIG_DPP_STEP(0x141);  // row_half_mirror
  BIG_DPP_STEP(0x140);  // row_mirror: every lane of a row of 16 holds the row's maximum
#undef BIG_DPP_STEP
  unsigned long long m = 0ull;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned long long v = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, 16 * r) << 32) |
                                 (unsigned)__builtin_amdgcn_readlane((int)lo, 16 * r);
    m = v > m ? v : m;
  }
  return m;
}
__device__ __forceinline__ double big_readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

template <class Cfg>
__device__ __forceinline__ void big_eliminate(double (&t1)[Cfg::TR][Cfg::TC], double (&t0)[Cfg::TR][Cfg::TC],
                                              double (&t2)[Cfg::TR][Cfg::TC], int n, double* __restrict__ lds, int r0, int c0,
                                              int tid, long long* __restrict__ prof = nullptr) {
  constexpr int NP = Cfg::NP, TR = Cfg::TR, TC = Cfg::TC;
  long long pt = prof ? clock64() : 0, pa[5] = {0, 0, 0, 0, 0};
#define EL_STAMP(K)                    \
  do {                                 \
    if (prof) {                        \
      const long long t_b = clock64(); \
      pa[K] += t_b - pt;               \
      pt = t_b;                        \
    }                                  \
  } while (0)
  constexpr int PB = Cfg::PB;          // pivots per round: NBLK block columns
  double* panel = lds + Cfg::E_PANEL;  // [NP][PB]
  double* cvb = lds + Cfg::E_CV;       // [NP][PB]
  double* rowb = lds + Cfg::E_ROW;     // [PB][3 NP]
  int* pivcol = reinterpret_cast<int*>(lds + Cfg::E_PIV);
  int* psb = reinterpret_cast<int*>(lds + Cfg::E_PS);  // [PB] pivot rows of the round
  const int lane = tid & 63;
  const bool w0 = tid < 64;
  bool used_a = false, used_b = false;  // wavefront 0: rows lane, lane + 64 have been pivot rows
  __syncthreads();                      // (the right panel, which these buffers alias, is no longer read)
  for (int j0 = 0; j0 < n; j0 += PB) {
    const int bs = (n - j0) < PB ? (n - j0) : PB;
    // 1. the panel
    if (c0 >= j0 && c0 < j0 + PB) {
      const int pc = c0 - j0;
#pragma unroll
      for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int jc = 0; jc < TC; ++jc) panel[(r0 + i) * PB + pc + jc] = t1[i][jc];
    }
    EL_STAMP(0);
    __syncthreads();
    EL_STAMP(1);
    // 2. wavefront 0 eliminates the panel
    if (w0) {
      const int ra = lane, rb = lane + 64;
      double xa[PB], xb[PB], ca[PB], cb[PB];
#pragma unroll
      for (int c = 0; c < PB; ++c) {
        xa[c] = panel[ra * PB + c];  // (n > 64: every lane has a first row)
        xb[c] = (rb < n) ? panel[rb * PB + c] : 0.0;
        ca[c] = 0.0;
        cb[c] = 0.0;
      }
#pragma unroll
      for (int sidx = 0; sidx < PB; ++sidx) {
        if (sidx < bs) {
          const unsigned long long ka =
              used_a ? 0ull : (((unsigned long long)__double_as_longlong(fabs(xa[sidx])) & ~127ull) | (unsigned long long)(127 - ra));
          const unsigned long long kb = (used_b || rb >= n)
                                            ? 0ull
                                            : (((unsigned long long)__double_as_longlong(fabs(xb[sidx])) & ~127ull) |
                                               (unsigned long long)(127 - rb));
          const unsigned long long km = big_wave_max_u64(ka > kb ? ka : kb);
          const int p = 127 - (int)(km & 127ull);
          const int pl = p & 63;
          const bool hislot = p >= 64;
          double xr[PB], cr[PB];
#pragma unroll
          for (int c = 0; c < PB; ++c) {
            xr[c] = big_readlane_f64(hislot ? xb[c] : xa[c], pl);
            cr[c] = big_readlane_f64(hislot ? cb[c] : ca[c], pl);
          }
          const double dinv = 1.0 / xr[sidx];
          cr[sidx] += 1.0;
#pragma unroll
          for (int c = 0; c < PB; ++c) {
            xr[c] *= dinv;
            cr[c] *= dinv;
          }
          const bool is_a = (ra == p), is_b = (rb == p);
          const double fa = xa[sidx], fb = xb[sidx];
#pragma unroll
          for (int c = 0; c < PB; ++c) {
            xa[c] = is_a ? xr[c] : fma(-fa, xr[c], xa[c]);
            ca[c] = is_a ? cr[c] : fma(-fa, cr[c], ca[c]);
            xb[c] = is_b ? xr[c] : fma(-fb, xr[c], xb[c]);
            cb[c] = is_b ? cr[c] : fma(-fb, cr[c], cb[c]);
          }
          used_a = used_a || is_a;
          used_b = used_b || is_b;
          if (lane == 0) {
            psb[sidx] = p;
            pivcol[p] = j0 + sidx;
          }
        }
      }
#pragma unroll
      for (int c = 0; c < PB; ++c) {
        cvb[ra * PB + c] = ca[c];
        if (rb < n) cvb[rb * PB + c] = cb[c];
      }
    }
    EL_STAMP(2);
    __syncthreads();
    EL_STAMP(3);
    // 3. the round's pivot rows (old values)
    int ps[PB];
#pragma unroll
    for (int t = 0; t < PB; ++t) ps[t] = (t < bs) ? psb[t] : -1;
    unsigned keep = 0u;  // bit i: row r0 + i is not one of the round's pivot rows (delta = 1)
#pragma unroll
    for (int i = 0; i < TR; ++i) {
      bool piv = false;
#pragma unroll
      for (int t = 0; t < PB; ++t)
        if (r0 + i == ps[t]) {
          piv = true;
#pragma unroll
          for (int jc = 0; jc < TC; ++jc) {
            rowb[t * 3 * NP + c0 + jc] = t1[i][jc];
            rowb[t * 3 * NP + NP + c0 + jc] = t0[i][jc];
            rowb[t * 3 * NP + 2 * NP + c0 + jc] = t2[i][jc];
          }
        }
      keep |= piv ? 0u : (1u << i);
    }
    __syncthreads();
    // 4. rank-bs update, one matrix after the other (the pivot rows of one matrix in registers at a time)
    double cv[TR][PB];
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
      for (int t = 0; t < PB; ++t) cv[i][t] = (r0 + i < n && t < bs) ? cvb[(r0 + i) * PB + t] : 0.0;
    auto update = [&](double (&tm)[TR][TC], int mofs) {
      double q[PB][TC];
#pragma unroll
      for (int t = 0; t < PB; ++t)
#pragma unroll
        for (int jc = 0; jc < TC; ++jc) q[t][jc] = (t < bs) ? rowb[t * 3 * NP + mofs + c0 + jc] : 0.0;  // (short last round: not published)
#pragma unroll
      for (int i = 0; i < TR; ++i) {
        const bool kp = (keep >> i) & 1u;
#pragma unroll
        for (int jc = 0; jc < TC; ++jc) {
          double acc = kp ? tm[i][jc] : 0.0;
#pragma unroll
          for (int t = 0; t < PB; ++t) acc = fma(cv[i][t], q[t][jc], acc);
          tm[i][jc] = acc;
        }
      }
    };
    update(t1, 0);
    update(t0, NP);
    update(t2, 2 * NP);
    EL_STAMP(4);
  }
  __syncthreads();  // pivcol complete; the buffers may be reused
  if (prof && tid == 0)
    for (int q = 0; q < 5; ++q) prof[q] += pa[q];
#undef EL_STAMP
}

// solution rows back in order: physical row r -> row pivcol[r] of the NP x NP workspace matrix (rows >= n stay zero)
template <class Cfg>
__device__ __forceinline__ void big_scatter_rows(const double (&t)[Cfg::TR][Cfg::TC], double* __restrict__ G, int n,
                                                 const double* __restrict__ lds, int r0, int c0) {
  const int* pivcol = reinterpret_cast<const int*>(lds + Cfg::E_PIV);
#pragma unroll
  for (int i = 0; i < Cfg::TR; ++i) {
    const int r = r0 + i;
    if (r < n) {
      int q = pivcol[r];
      q = q < 0 ? 0 : (q >= n ? n - 1 : q);  // (NaN input: the search may have left garbage; stay inside the matrix)
#pragma unroll
      for (int jc = 0; jc < Cfg::TC; ++jc) G[q * Cfg::LD + c0 + jc] = t[i][jc];
    }
  }
}

// ---- C = L R on the matrix core: L, R = the two LDS panels; epi(tile row, tile col, row0, col, acc): acc[e] is element
// (row0 + 4 e, col) (layout of v_mfma_f64_16x16x4_f64 probed in tools/mfma_probe) ----------------------------------------------------
template <class Cfg, class Init, class Epi>
__device__ __forceinline__ void big_gemm(const double* __restrict__ bufL, const double* __restrict__ bufR, int tid, Init init,
                                         Epi epi) {
  constexpr int MT = Cfg::MT, LD = Cfg::LD, NP = Cfg::NP;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
  for (int s = 0; s < Cfg::TPW; ++s) {  // (slot s of this wavefront: tile wave + s NW)
    const int t = wave + s * Cfg::NW;
    if (t < MT * MT) {
      const int ti = t / MT, tj = t - ti * MT;
      big_v4f64 acc = init(s);
      const double* pa = bufL + (16 * ti + (lane & 15)) * LD + (lane >> 4);
      const double* pb = bufR + (lane >> 4) * LD + 16 * tj + (lane & 15);
#pragma unroll 4
      for (int k4 = 0; k4 < NP / 4; ++k4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * k4], pb[4 * k4 * LD], acc, 0, 0, 0);
      epi(s, ti, 16 * ti + (lane >> 4), 16 * tj + (lane & 15), acc);
    }
  }
}
struct BigZero {
  __device__ __forceinline__ big_v4f64 operator()(int) const { return big_v4f64{0.0, 0.0, 0.0, 0.0}; }
};

// a panel on its way from the workspace to LDS, parked in registers while a product runs
template <class Cfg>
struct BigPanelRegs {
  double2 v[Cfg::PF];
  __device__ __forceinline__ void fetch(const double* __restrict__ G, int tid) {
    const double2* g2 = reinterpret_cast<const double2*>(G) + tid;
#pragma unroll
    for (int q = 0; q < Cfg::PF; ++q)
      if ((q + 1) * Cfg::NT <= Cfg::MAT / 2 || tid + q * Cfg::NT < Cfg::MAT / 2) v[q] = g2[q * Cfg::NT];
  }
  __device__ __forceinline__ void store(double* __restrict__ buf, int tid, double sign) const {
    double2* b2 = reinterpret_cast<double2*>(buf) + tid;
#pragma unroll
    for (int q = 0; q < Cfg::PF; ++q)
      if ((q + 1) * Cfg::NT <= Cfg::MAT / 2 || tid + q * Cfg::NT < Cfg::MAT / 2)
        b2[q * Cfg::NT] = double2{sign * v[q].x, sign * v[q].y};
  }
};

// column sums of |tile| into the partial array (deterministic: one writer per (tile row, column))
template <class Cfg>
__device__ __forceinline__ void big_colsum_part(double* __restrict__ part, int ti, int col, big_v4f64 v, int lane) {
  double s = fabs(v[0]) + fabs(v[1]) + fabs(v[2]) + fabs(v[3]);
  s += shfl_xor_f64(s, 16);
  s += shfl_xor_f64(s, 32);
  if (lane < 16) part[ti * Cfg::NP + col] = s;
}

// NaN-propagating maximum over the workgroup; `red`: NW doubles of LDS.  Barriers inside (entry and exit).
template <class Cfg>
__device__ __forceinline__ double big_block_nanmax(double v, double* __restrict__ red, int tid) {
  v = wave_nanmax(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double m = red[0];
#pragma unroll
  for (int w = 1; w < Cfg::NW; ++w) m = nanmax(m, red[w]);
  return m;
}
template <class Cfg>
__device__ __forceinline__ double big_block_sum(double v, double* __restrict__ red, int tid) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double s = red[0];
#pragma unroll
  for (int w = 1; w < Cfg::NW; ++w) s += red[w];
  return s;
}

// induced 1-norm from the partial column sums of the last product (barrier before reading them)
template <class Cfg>
__device__ __forceinline__ double big_norm1(const double* __restrict__ part, double* __restrict__ red, int tid) {
  __syncthreads();
  double m = 0.0;
  for (int c = tid; c < Cfg::NP; c += Cfg::NT) {
    double cs = 0.0;
#pragma unroll
    for (int ti = 0; ti < Cfg::MT; ++ti) cs += part[ti * Cfg::NP + c];
    m = nanmax(m, cs);
  }
  return big_block_nanmax<Cfg>(m, red, tid);
}

// ---- cycle reduction (njit rule; scan_mode: the scan variant's rule and 1e-16 stabilisation) ----------------------------------------
template <class Cfg>
__global__ __launch_bounds__(Cfg::NT) void cr_big_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                         const double* __restrict__ C, int batch, int n, int max_iter, double tol,
                                                         double* __restrict__ ws, double* __restrict__ T_out,
                                                         int32_t* __restrict__ status, int32_t* __restrict__ n_iter_out,
                                                         int scan_mode, const double* __restrict__ D, int k,
                                                         double* __restrict__ R_out, long long* __restrict__ dbg) {
  constexpr int NP = Cfg::NP, TR = Cfg::TR, TC = Cfg::TC, NT = Cfg::NT;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* bufL = lds;
  double* bufR = lds + Cfg::OFF_R;
  double* part = lds + Cfg::OFF_PART;
  // debug (dsge_debug_big_phases): shader cycles of workgroup 0's first draw in [0] tile loads, [1] eliminations, [2] scatters,
  // [3] the four products; [4] iterations, [5] total
  const bool prof = dbg != nullptr && blockIdx.x == 0;
  long long t_a = 0, acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0, t_begin = 0;
#define BIG_STAMP(ACC)                  \
  do {                                  \
    if (prof) {                         \
      const long long t_b = clock64();  \
      ACC += t_b - t_a;                 \
      t_a = t_b;                        \
    }                                   \
  } while (0)
  double* red = lds + Cfg::OFF_RED;
  // Thread coordinates are re-derived from an OPAQUE copy of the thread index at the head of every phase (BIG_COORDS): the
  // compiler otherwise hoists every address that depends on them out of the iteration and the draw loops and keeps ~70 of them
  // alive through the elimination, whose register blocks then spill (544 bytes of scratch per lane at NP = 96).
  // (tx fastest.  ty fastest -- the GY owners of a block column in ONE wavefront, the pivot-candidate code skipped by all others --
  //  measured slower: 310 k instead of 246 k cycles per elimination at n = 80)
#define BIG_COORDS()                                                                              \
  int tid = threadIdx.x;                                                                          \
  asm volatile("" : "+v"(tid));                                                                   \
  const int lane = tid & 63;                                                                      \
  const int ty = tid / Cfg::GX, tx = tid - ty * Cfg::GX, r0 = ty * TR, c0 = tx * TC;             \
  (void)lane;                                                                                     \
  (void)r0;                                                                                       \
  (void)c0
  double* W = ws + (size_t)blockIdx.x * Cfg::ws_doubles;
  constexpr int LD = Cfg::LD, MAT = Cfg::MAT;
  double *A0g = W, *A1g = W + MAT, *A2g = W + 2 * MAT, *Ahg = W + 3 * MAT, *X0g = W + 4 * MAT, *X2g = W + 5 * MAT;
  for (int idx = threadIdx.x; idx < MAT; idx += NT) {  // (rows >= n of X0, X2 are never written again)
    X0g[idx] = 0.0;
    X2g[idx] = 0.0;
  }
  // (one draw per workgroup -- the launcher cuts the batch into launches of at most BIG_GRID_MAX draws --: no grid-stride loop
  //  for the compiler to hoist loop invariants out of and spill them; see kalman_nt_kernel)
  for (int draw = blockIdx.x; draw < batch; draw = batch) {
    const size_t off = (size_t)draw * n * n;
    __syncthreads();
    for (int idx = threadIdx.x; idx < MAT; idx += NT) {
      const int r = idx / LD, c = idx - r * LD;
      const bool in = r < n && c < n;
      const size_t g = off + (size_t)r * n + c;
      const double b = in ? B[g] : 0.0;
      A0g[idx] = in ? A[g] : 0.0;
      A1g[idx] = b;
      Ahg[idx] = b;
      A2g[idx] = in ? C[g] : 0.0;
    }
    for (int idx = threadIdx.x; idx < NP * Cfg::LD; idx += NT) bufR[idx] = 0.0;  // (rows >= n of X0 in the panel are never written)
    __syncthreads();
    bool converged = false, saw_nan = false;
    int it = 0;
    double t1[TR][TC], t0[TR][TC], t2[TR][TC];
    if (prof) t_begin = t_a = clock64();
    for (; it < max_iter;) {
      {
      BIG_COORDS();
      big_tile_load<Cfg>(t1, A1g, r0, c0);
      big_tile_load<Cfg>(t0, A0g, r0, c0);
      big_tile_load<Cfg>(t2, A2g, r0, c0);
      if (scan_mode) {  // stabilize(A1): 1e-16 on the diagonal of the solve only (shared.py:6-9)
#pragma unroll
        for (int i = 0; i < TR; ++i)
#pragma unroll
          for (int jc = 0; jc < TC; ++jc)
            if (r0 + i == c0 + jc && r0 + i < n) t1[i][jc] += 1e-16;
      }
#pragma unroll
      for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int jc = 0; jc < TC; ++jc) bufL[(r0 + i) * Cfg::LD + c0 + jc] = -t0[i][jc];  // the left panel of the first two products
      BIG_STAMP(acc0);
      big_eliminate<Cfg>(t1, t0, t2, n, lds, r0, c0, tid, (prof && draw == 0) ? dbg + 8 : nullptr);
      BIG_STAMP(acc1);
      big_scatter_rows<Cfg>(t0, X0g, n, lds, r0, c0);
      big_scatter_rows<Cfg>(t2, X2g, n, lds, r0, c0);
      {  // X0 straight into the right panel as well (rows >= n of the panel: zero since the draw's set-up)
        const int* pivcol = reinterpret_cast<const int*>(lds + Cfg::E_PIV);
#pragma unroll
        for (int i = 0; i < TR; ++i)
          if (r0 + i < n) {
            int q = pivcol[r0 + i];
            q = q < 0 ? 0 : (q >= n ? n - 1 : q);
#pragma unroll
            for (int jc = 0; jc < TC; ++jc) bufR[q * Cfg::LD + c0 + jc] = t0[i][jc];
          }
      }
      __syncthreads();
      BIG_STAMP(acc2);
      }
      BIG_COORDS();
      // The four products.  The left panel holds -A0, then -A2, so that every product IS the updated matrix: A0 := (-A0) X0,
      // A1 := A1 + (-A0) X2 + (-A2) X0, A2 := (-A2) X2, A1_hat := A1_hat + (-A2) X0.  A1 and A1_hat stay in this wavefront's
      // accumulator tiles from iteration to iteration (no read-modify-write through the workspace); the next panel is fetched
      // into registers while the current product runs.
      BigPanelRegs<Cfg> pf;
      pf.fetch(X2g, tid);
      // A1 and A1_hat in product-tile layout (slot s of this wavefront: tile wave + s NW; element (row0 + 4 e, col)), requested
      // now and first used two products later.  [Kept in registers from iteration to iteration they cost the elimination its
      // registers: 592 bytes of scratch per lane at NP = 96 and an elimination twice as slow.]
      big_v4f64 a1acc[Cfg::TPW], ahacc[Cfg::TPW];
      {
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int sl = 0; sl < Cfg::TPW; ++sl) {
          const int t = wave + sl * Cfg::NW, ti = t / Cfg::MT, tj = t - ti * Cfg::MT;
          if (t < Cfg::MT * Cfg::MT) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int o = (16 * ti + (lane >> 4) + 4 * e) * LD + 16 * tj + (lane & 15);
              a1acc[sl][e] = A1g[o];
              ahacc[sl][e] = Ahg[o];
            }
          }
        }
      }
      big_gemm<Cfg>(bufL, bufR, tid, BigZero(), [&](int, int ti, int row0, int col, big_v4f64 v) {  // A0 := -A0 X0
#pragma unroll
        for (int e = 0; e < 4; ++e) A0g[(row0 + 4 * e) * LD + col] = v[e];
        big_colsum_part<Cfg>(part, ti, col, v, lane);
      });
      const double nrm0 = big_norm1<Cfg>(part, red, tid);
      pf.store(bufR, tid, 1.0);  // X2
      __syncthreads();
      pf.fetch(A2g, tid);
      big_gemm<Cfg>(bufL, bufR, tid, [&](int sl) { return a1acc[sl]; },
                    [&](int sl, int, int, int, big_v4f64 v) { a1acc[sl] = v; });  // A1 -= A0 X2
      __syncthreads();
      pf.store(bufL, tid, -1.0);  // -A2
      __syncthreads();
      pf.fetch(X0g, tid);
      big_gemm<Cfg>(bufL, bufR, tid, BigZero(), [&](int, int ti, int row0, int col, big_v4f64 v) {  // A2 := -A2 X2
#pragma unroll
        for (int e = 0; e < 4; ++e) A2g[(row0 + 4 * e) * LD + col] = v[e];
        big_colsum_part<Cfg>(part, ti, col, v, lane);
      });
      const double nrm2 = big_norm1<Cfg>(part, red, tid);
      pf.store(bufR, tid, 1.0);  // X0
      __syncthreads();
      big_gemm<Cfg>(bufL, bufR, tid, BigZero(), [&](int sl, int, int row0, int col, big_v4f64 v) {  // A1, A1_hat -= A2 X0
        a1acc[sl] += v;
        ahacc[sl] += v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int o = (row0 + 4 * e) * LD + col;
          A1g[o] = a1acc[sl][e];
          Ahg[o] = ahacc[sl][e];
        }
      });
      __syncthreads();
      BIG_STAMP(acc3);
      ++it;
      if (nrm0 < tol) {
        if (nrm2 < tol || scan_mode) {  // the scan variant tests the A0 norm only (cycle_reduction.py:269-277)
          converged = true;
          break;
        }
      } else if (nrm0 != nrm0) {
        saw_nan = true;
        break;
      }
    }
    const bool solve_T = converged || (scan_mode && !saw_nan);
    const bool want_R = (R_out != nullptr) && !scan_mode;
    const size_t offk = (size_t)draw * n * k;
    if (solve_T) {
      // T = -A1_hat^-1 A  (cycle_reduction.py:181); with D the same elimination gives R = -A1_hat^-1 D
      BIG_COORDS();
      big_tile_load<Cfg>(t1, Ahg, r0, c0);
#pragma unroll
      for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int jc = 0; jc < TC; ++jc) {
          const int r = r0 + i, c = c0 + jc;
          if (scan_mode && r == c && r < n) t1[i][jc] += 1e-16;
          t0[i][jc] = (r < n && c < n) ? A[off + (size_t)r * n + c] : 0.0;
          t2[i][jc] = (want_R && r < n && c < k) ? D[offk + (size_t)r * k + c] : 0.0;
        }
      big_eliminate<Cfg>(t1, t0, t2, n, lds, r0, c0, tid);
      const int* pivcol = reinterpret_cast<const int*>(lds + Cfg::E_PIV);
#pragma unroll
      for (int i = 0; i < TR; ++i) {
        const int r = r0 + i;
        if (r < n) {
          int q = pivcol[r];
          q = q < 0 ? 0 : (q >= n ? n - 1 : q);
#pragma unroll
          for (int jc = 0; jc < TC; ++jc) {
            const int c = c0 + jc;
            if (c < n) T_out[off + (size_t)q * n + c] = -t0[i][jc];
            if (want_R && c < k) R_out[offk + (size_t)q * k + c] = -t2[i][jc];
          }
        }
      }
    } else {
      for (int idx = threadIdx.x; idx < n * n; idx += NT) T_out[off + idx] = 0.0;
      if (R_out != nullptr && !scan_mode)
        for (int idx = threadIdx.x; idx < n * k; idx += NT) R_out[offk + idx] = 0.0;
    }
    if (threadIdx.x == 0) {
      status[draw] = solve_T ? DSGE_ST_OK : (DSGE_ST_NOT_CONVERGED | (saw_nan ? DSGE_ST_NAN : 0));
      if (n_iter_out) n_iter_out[draw] = it;
      if (prof && draw == 0) {
        dbg[0] = acc0;
        dbg[1] = acc1;
        dbg[2] = acc2;
        dbg[3] = acc3;
        dbg[4] = it;
        dbg[5] = clock64() - t_begin;
      }
    }
  }
#undef BIG_STAMP
#undef BIG_COORDS
}

// ---- selection matrix and policy residual for a given T (assemble_kernel's do_selection part, n > 64):
//      R = -(C T + B)^-1 D (shared.py:74-75), resid = sum((A + (B + C T) T)^2); failed draws: R = 0, resid = inf ------------------------
template <class Cfg>
__global__ __launch_bounds__(Cfg::NT) void selection_big_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                                const double* __restrict__ C, const double* __restrict__ D,
                                                                const double* __restrict__ T, int batch, int n, int k,
                                                                double* __restrict__ ws, double* __restrict__ R_out,
                                                                double* __restrict__ resid_out,
                                                                const int32_t* __restrict__ status) {
  constexpr int NP = Cfg::NP, TR = Cfg::TR, TC = Cfg::TC, NT = Cfg::NT;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* bufL = lds;
  double* bufR = lds + Cfg::OFF_R;
  double* red = lds + Cfg::OFF_RED;
  const int tid = threadIdx.x;
  // (tx fastest.  ty fastest -- the GY owners of a block column in ONE wavefront, the pivot-candidate code skipped by all others --
  //  measured slower: 310 k instead of 246 k cycles per elimination at n = 80)
  const int ty = tid / Cfg::GX, tx = tid - ty * Cfg::GX, r0 = ty * TR, c0 = tx * TC;
  double* W = ws + (size_t)blockIdx.x * Cfg::ws_doubles;
  constexpr int LD = Cfg::LD, MAT = Cfg::MAT;
  double *Cg = W, *Tg = W + MAT, *Mg = W + 2 * MAT;
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    const size_t off = (size_t)draw * n * n, offk = (size_t)draw * n * k;
    if (status && status[draw] != 0) {
      if (R_out)
        for (int idx = tid; idx < n * k; idx += NT) R_out[offk + idx] = 0.0;
      if (resid_out && tid == 0) resid_out[draw] = INFINITY;
      continue;
    }
    __syncthreads();
    for (int idx = tid; idx < MAT; idx += NT) {
      const int r = idx / LD, c = idx - r * LD;
      const bool in = r < n && c < n;
      const size_t g = off + (size_t)r * n + c;
      Cg[idx] = in ? C[g] : 0.0;
      Tg[idx] = in ? T[g] : 0.0;
    }
    __syncthreads();
    big_panel_load<Cfg>(bufL, Cg, tid);
    big_panel_load<Cfg>(bufR, Tg, tid);
    __syncthreads();
    big_gemm<Cfg>(bufL, bufR, tid, BigZero(), [&](int, int, int row0, int col, big_v4f64 v) {  // M = B + C T
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = row0 + 4 * e;
        Mg[r * LD + col] = (r < n && col < n) ? B[off + (size_t)r * n + col] + v[e] : 0.0;
      }
    });
    __syncthreads();
    if (resid_out) {  // A + M T
      big_panel_load<Cfg>(bufL, Mg, tid);
      __syncthreads();
      double s = 0.0;
      big_gemm<Cfg>(bufL, bufR, tid, BigZero(), [&](int, int, int row0, int col, big_v4f64 v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = row0 + 4 * e;
          if (r < n && col < n) {
            const double x = A[off + (size_t)r * n + col] + v[e];
            s = fma(x, x, s);
          }
        }
      });
      s = big_block_sum<Cfg>(s, red, tid);
      if (tid == 0) resid_out[draw] = s;
      __syncthreads();
    }
    if (R_out) {
      double t1[TR][TC], t0[TR][TC], t2[TR][TC];
      big_tile_load<Cfg>(t1, Mg, r0, c0);
#pragma unroll
      for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int jc = 0; jc < TC; ++jc) {
          const int r = r0 + i, c = c0 + jc;
          t0[i][jc] = (r < n && c < k) ? D[offk + (size_t)r * k + c] : 0.0;
          t2[i][jc] = 0.0;
        }
      big_eliminate<Cfg>(t1, t0, t2, n, lds, r0, c0, tid);
      const int* pivcol = reinterpret_cast<const int*>(lds + Cfg::E_PIV);
#pragma unroll
      for (int i = 0; i < TR; ++i) {
        const int r = r0 + i;
        if (r < n) {
          int q = pivcol[r];
          q = q < 0 ? 0 : (q >= n ? n - 1 : q);
#pragma unroll
          for (int jc = 0; jc < TC; ++jc)
            if (c0 + jc < k) R_out[offk + (size_t)q * k + c0 + jc] = -t0[i][jc];
        }
      }
    }
  }
}

// ---- gensys for 65 .. 96 variables by spectral division (dsge_gensys_doubling.hpp has the derivation and the n <= 64 kernel):
//      given the solvent T of the doubling iteration, M = B + C T, ONE elimination of [M | D | C] gives R = -M^-1 D and G = M^-1 C;
//      the draw is certified -- eu = [1, 1, 0], gEconpy/solvers/gensys.py:237-250, 282-310 -- iff rho(G[L,L]) < 1 and rho(T[S,S]) < 1
//      (L: columns of C with sum|C_ij| > tol, gensys.py:587; S: non-zero columns of T), shown by Frobenius norms of repeated squares
//      (< 1/2 within 12 squarings).  There is NO ordered QZ at this size: a draw without the certificate -- not converged, a root
//      within 2e-4 of the unit circle, a column of C below tol, singular M, |T| > 1e6, more than 32 lead or 64 state columns -- gets
//      eu = [-3, -3, 0], status NOT_CONVERGED | GENSYS_TOO_BIG ("no verdict at this size") and T = R = 0: never a wrong verdict. -----
constexpr int BIG_GD_LCAP = 32, BIG_GD_SCAP = 64, BIG_GD_SQUARINGS = 12;

// rho(P) < 1 certified?  buf: 2 x d x ld doubles (P in the first half); the whole workgroup works, barriers inside, uniform result.
template <class Cfg>
__device__ __forceinline__ bool big_certify_contraction(double* __restrict__ buf, int d, int ld, double* __restrict__ red, int tid) {
  if (d == 0) return true;
  double* cur = buf;
  double* nxt = buf + (size_t)d * ld;
  double fro = 0.0;
  for (int idx = tid; idx < d * d; idx += Cfg::NT) {
    const double v = cur[(idx / d) * ld + (idx % d)];
    fro = fma(v, v, fro);
  }
  fro = big_block_sum<Cfg>(fro, red, tid);
  for (int k = 0; k <= BIG_GD_SQUARINGS; ++k) {
    if (!(fro == fro) || !(fro < 1e300)) return false;
    if (fro < 0.25) return true;
    if (k == BIG_GD_SQUARINGS) break;
    double part = 0.0;
    for (int idx = tid; idx < d * d; idx += Cfg::NT) {
      const int i = idx / d, j = idx - i * d;
      double s0 = 0.0, s1 = 0.0;
      int q = 0;
      for (; q + 1 < d; q += 2) {
        s0 = fma(cur[i * ld + q], cur[q * ld + j], s0);
        s1 = fma(cur[i * ld + q + 1], cur[(q + 1) * ld + j], s1);
      }
      if (q < d) s0 = fma(cur[i * ld + q], cur[q * ld + j], s0);
      const double v = s0 + s1;
      nxt[i * ld + j] = v;
      part = fma(v, v, part);
    }
    fro = big_block_sum<Cfg>(part, red, tid);  // (its barriers order the writes of nxt before the next round's reads)
    double* t = cur;
    cur = nxt;
    nxt = t;
  }
  return false;
}

template <class Cfg>
__global__ __launch_bounds__(Cfg::NT) void gensys_certify_big_kernel(const double* __restrict__ B, const double* __restrict__ C,
                                                                     const double* __restrict__ D, double* __restrict__ T,
                                                                     int batch, int n, int k, double tol, double* __restrict__ ws,
                                                                     double* __restrict__ R_out, int32_t* __restrict__ eu_out,
                                                                     int32_t* __restrict__ status) {
  constexpr int NP = Cfg::NP, TR = Cfg::TR, TC = Cfg::TC, NT = Cfg::NT;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* bufL = lds;
  double* bufR = lds + Cfg::OFF_R;
  double* red = lds + Cfg::OFF_RED;
  // index maps in the column-sum partials (unused here): position of a column in L / in S (or -1), the members of S, counters
  int* posL = reinterpret_cast<int*>(lds + Cfg::OFF_PART);
  int* posS = posL + NP;
  int* sidx = posS + NP;
  int* cnt = sidx + NP;  // {#lead, #state, a column that rules the certificate out}
  static_assert((size_t)Cfg::MT * NP * 2 >= 3 * NP + 4 && NP <= 128, "index maps fit the column-sum partials");
  // after the elimination both panels are free: G[L,L] and its square, T[S,S] and its square
  constexpr int LDG = BIG_GD_LCAP | 1, LDS_ = BIG_GD_SCAP | 1;
  static_assert(2 * BIG_GD_LCAP * LDG + 2 * BIG_GD_SCAP * LDS_ <= 2 * NP * Cfg::LD, "certificate buffers fit the two panels");
  double* Gm = lds;
  double* Ts = lds + 2 * BIG_GD_LCAP * LDG;
  double* Ml = lds + Cfg::OFF_R;  // M^-1 for the scale guards (second panel; overlaps Ts, which is filled after them)
  static_assert(2 * BIG_GD_LCAP * LDG <= Cfg::OFF_R, "G[L,L] and its square stay clear of the inverse");
  const int tid = threadIdx.x;
  const int ty = tid / Cfg::GX, tx = tid - ty * Cfg::GX, r0 = ty * TR, c0 = tx * TC;
  double* W = ws + (size_t)blockIdx.x * Cfg::ws_doubles;
  constexpr int LD = Cfg::LD, MAT = Cfg::MAT;
  double *Cg = W, *Tg = W + MAT, *Mg = W + 2 * MAT;
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    const size_t off = (size_t)draw * n * n, offk = (size_t)draw * n * k;
    __syncthreads();
    const bool conv = status[draw] == 0;  // (read by every thread before thread 0 writes it at the end of the draw)
    bool ok = conv;
    if (conv) {  // (uniform)
      // lead and state columns; what rules the certificate out: a column gensys drops although the iteration used it
      // (0 < sum <= tol), |T| beyond 1e6 (gensys's existence test is a tolerance there), NaN
      if (tid == 0) cnt[2] = 0;
      __syncthreads();
      if (tid < NP) {
        double csum = 0.0, tmax = 0.0;
        bool tnz = false;
        if (tid < n) {
#pragma unroll 8
          for (int r = 0; r < n; ++r) {  // (no short-circuit: the loads of eight rows are requested together)
            const double cvv = C[off + (size_t)r * n + tid], tv = T[off + (size_t)r * n + tid];
            csum += fabs(cvv);
            tnz |= (tv != 0.0);
            tmax = fmax(tmax, fabs(tv));
          }
        }
        posL[tid] = (tid < n && csum > tol) ? 1 : 0;
        posS[tid] = (tid < n && tnz) ? 1 : 0;
        if (tid < n && ((csum > 0.0 && !(csum > tol)) || !(csum == csum) || !(tmax < 1e6))) cnt[2] = 1;
      }
      __syncthreads();
      if (tid < 64) {  // flags -> positions (each lane reads and rewrites its own two entries)
        const bool hi = tid + 64 < NP;
        const bool l0 = posL[tid] != 0, l1 = hi && posL[hi ? tid + 64 : tid] != 0;
        const bool s0 = posS[tid] != 0, s1 = hi && posS[hi ? tid + 64 : tid] != 0;
        const unsigned long long ml0 = __ballot(l0), ml1 = __ballot(l1), ms0 = __ballot(s0), ms1 = __ballot(s1);
        const unsigned long long below = (1ull << tid) - 1ull;
        const int pl0 = __popcll(ml0 & below), pl1 = __popcll(ml0) + __popcll(ml1 & below);
        const int ps0 = __popcll(ms0 & below), ps1 = __popcll(ms0) + __popcll(ms1 & below);
        posL[tid] = l0 ? pl0 : -1;
        posS[tid] = s0 ? ps0 : -1;
        if (s0) sidx[ps0] = tid;
        if (hi) {
          posL[tid + 64] = l1 ? pl1 : -1;
          posS[tid + 64] = s1 ? ps1 : -1;
          if (s1) sidx[ps1] = tid + 64;
        }
        if (tid == 0) {
          cnt[0] = __popcll(ml0) + __popcll(ml1);
          cnt[1] = __popcll(ms0) + __popcll(ms1);
        }
      }
      __syncthreads();
      const int l = cnt[0], sN = cnt[1];
      ok = cnt[2] == 0 && l <= BIG_GD_LCAP && sN <= BIG_GD_SCAP;
      // M = B + C T (as selection_big_kernel)
      for (int idx = tid; idx < MAT; idx += NT) {
        const int r = idx / LD, c = idx - r * LD;
        const bool in = r < n && c < n;
        const size_t g = off + (size_t)r * n + c;
        Cg[idx] = in ? C[g] : 0.0;
        Tg[idx] = in ? T[g] : 0.0;
      }
      __syncthreads();
      big_panel_load<Cfg>(bufL, Cg, tid);
      big_panel_load<Cfg>(bufR, Tg, tid);
      __syncthreads();
      big_gemm<Cfg>(bufL, bufR, tid, BigZero(), [&](int, int, int row0, int col, big_v4f64 v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = row0 + 4 * e;
          Mg[r * LD + col] = (r < n && col < n) ? B[off + (size_t)r * n + col] + v[e] : 0.0;
        }
      });
      __syncthreads();
      // [M | D | I] -> [. | M^-1 D | M^-1]   (round 6: the identity in the third operand slot -- the scale guards need the inverse;
      //  G[L,L] = N_L C[:,L] is a small product afterwards)
      double t1[TR][TC], t0[TR][TC], t2[TR][TC];
      big_tile_load<Cfg>(t1, Mg, r0, c0);
#pragma unroll
      for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int jc = 0; jc < TC; ++jc) {
          const int r = r0 + i, c = c0 + jc;
          t0[i][jc] = (D && r < n && c < k) ? D[offk + (size_t)r * k + c] : 0.0;
          t2[i][jc] = (r == c && r < n) ? 1.0 : 0.0;
        }
      big_eliminate<Cfg>(t1, t0, t2, n, lds, r0, c0, tid);
      const int* pivcol = reinterpret_cast<const int*>(lds + Cfg::E_PIV);
      int qrow[TR];
#pragma unroll
      for (int i = 0; i < TR; ++i) {
        int q = (r0 + i < n) ? pivcol[r0 + i] : 0;
        qrow[i] = q < 0 ? 0 : (q >= n ? n - 1 : q);
      }
      __syncthreads();  // (every thread is out of the elimination: its panels may be overwritten)
      // M^-1 in natural row order into the second LDS panel (both panels are free behind the elimination; the certificate's T[S,S]
      // buffer, which overlaps it, is filled after the guards): the guards and G[L,L] read it from there -- from the L2-resident
      // workspace every access was a microsecond under the launch's own traffic, 60 us per draw.
      // R = -M^-1 D is written now (and zeroed below for a draw that ends without the certificate), so that no register tile
      // lives across the guards
#pragma unroll
      for (int i = 0; i < TR; ++i)
        if (r0 + i < n) {
#pragma unroll
          for (int jc = 0; jc < TC; ++jc) {
            if (c0 + jc < n) Ml[qrow[i] * LD + c0 + jc] = t2[i][jc];
            if (R_out && c0 + jc < k) R_out[offk + (size_t)qrow[i] * k + c0 + jc] = ok ? -t0[i][jc] : 0.0;
          }
        }
      // G[L,L] (and, behind the guards, T[S,S]) into the first panel
      for (int idx = tid; idx < 2 * BIG_GD_LCAP * LDG; idx += NT) lds[idx] = 0.0;
      __syncthreads();
      if (ok) {  // (uniform)
        int gt = threadIdx.x;  // (opaque copy: nothing derived from it is hoisted across the elimination, see BIG_COORDS)
        asm volatile("" : "+v"(gt));
        // ---- the scale guards of dsge_gensys_doubling.hpp (derivation there): (E) existence = sigma_max(N_L) against 1 / tol, exactly
        // the singular values of Q2 Pi; (Z) no coincident-zero pair in the stable block: ||M^-1 (I + N_L'N_L)^-1/2||_F^2 <=
        // ||M^-1||_F^2 - |M^-1 w|^2 / (tau + |w|^2) for w = N_L'N_L v, tau = |N_L v|^2, v from the power iteration
        double* vb = Gm + BIG_GD_LCAP * LDG;  // 128 doubles of the certificate's second buffer (unused until the squarings)
        double* tb = vb + 128;                // 32
        double mi2 = 0.0, nl2 = 0.0, tl2 = 0.0;
        int* lrow = cnt + 4;  // the lead rows in order (the index maps' area has room: 10 NP ints against 4 NP + 4 used)
        if (gt < n && posL[gt] >= 0) lrow[posL[gt]] = gt;
        for (int idx = gt; idx < n * n; idx += NT) {
          const int r = idx / n, c = idx - r * n;
          const double v = Ml[r * LD + c];
          mi2 = fma(v, v, mi2);
          if (posL[r] >= 0) {
            nl2 = fma(v, v, nl2);
            const double tv = Tg[r * LD + c];
            tl2 = fma(tv, tv, tl2);
          }
        }
        __syncthreads();
        // G[L,L] = N_L C[:,L]: one element per thread, the dot product in batches of eight (16 loads of the L2-resident arrays in
        // flight: a scalar loop under its condition was a chain of load latencies, 0.4 ms per 1024 draws at n = 80)
        for (int idx = gt; idx < l * l; idx += NT) {
          const int a_ = idx / l, b_ = idx - a_ * l;
          const double* mr = Ml + lrow[a_] * LD;
          const double* cc_ = Cg + lrow[b_];
          double g0 = 0.0, g1 = 0.0;
          int i = 0;
          for (; i + 8 <= n; i += 8) {
            double mv[8], cv8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              mv[e] = mr[i + e];
              cv8[e] = cc_[(i + e) * LD];
            }
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              g0 = fma(mv[e], cv8[e], g0);
              g1 = fma(mv[e + 1], cv8[e + 1], g1);
            }
          }
          for (; i < n; ++i) g0 = fma(mr[i], cc_[i * LD], g0);
          Gm[a_ * LDG + b_] = g0 + g1;
        }
        mi2 = big_block_sum<Cfg>(mi2, red, gt);
        nl2 = big_block_sum<Cfg>(nl2, red, gt);
        tl2 = big_block_sum<Cfg>(tl2, red, gt);
        // power iteration from the vector of ones (any v gives a valid bound; two steps).  The row products -- t = N_L v, z = M^-1 w --
        // run one ROW PER WAVEFRONT with the lanes along the row (coalesced reads of the L2-resident inverse, a DPP sum per row): one
        // thread per row walked its row alone, n cache lines one after the other, 12 times per draw (n = 80: + 0.4 ms per 1024 draws)
        const int gw = gt >> 6, gl = gt & 63;
        constexpr int NWV = NT / 64;
        double tau = 0.0, w2 = 0.0;
        for (int it = 0; it < 2; ++it) {  // (two steps: a direction that matters is ahead of the rest by orders of magnitude)
          double vj = 0.0;
          if (it == 0) vj = (gt < n) ? 1.0 : 0.0;
          else if (gt < n) vj = vb[gt];
          const double nv = big_block_sum<Cfg>(vj * vj, red, gt);
          vj = nv > 0.0 ? vj * (1.0 / sqrt(nv)) : 0.0;
          __syncthreads();
          if (gt < n) vb[gt] = vj;
          __syncthreads();
          double ta2 = 0.0;
          for (int r = gw; r < n; r += NWV) {
            const int pa = posL[r];
            if (pa >= 0) {  // (wave-uniform)
              double part = 0.0;
              for (int i = gl; i < n; i += 64) part = fma(Ml[r * LD + i], vb[i], part);
              part = wave_sum(part);
              if (gl == 0) {
                tb[pa] = part;
                ta2 = fma(part, part, ta2);
              }
            }
          }
          tau = big_block_sum<Cfg>(ta2, red, gt);
          __syncthreads();
          double wj = 0.0;
          if (gt < n) {  // w = N_L' t over the list of lead rows, eight loads in flight
            const double* mc = Ml + gt;
            double w1 = 0.0;
            int a_ = 0;
            for (; a_ + 8 <= l; a_ += 8) {
              double mv[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) mv[e] = mc[lrow[a_ + e] * LD];
#pragma unroll
              for (int e = 0; e < 8; e += 2) {
                wj = fma(mv[e], tb[a_ + e], wj);
                w1 = fma(mv[e + 1], tb[a_ + e + 1], w1);
              }
            }
            for (; a_ < l; ++a_) wj = fma(mc[lrow[a_] * LD], tb[a_], wj);
            wj += w1;
          }
          w2 = big_block_sum<Cfg>(wj * wj, red, gt);
          __syncthreads();
          if (gt < n) vb[gt] = wj;  // (un-normalised w on exit: the bound uses w itself)
          __syncthreads();
        }
        double zj = 0.0;
        for (int r = gw; r < n; r += NWV) {
          double part = 0.0;
          for (int i = gl; i < n; i += 64) part = fma(Ml[r * LD + i], vb[i], part);
          part = wave_sum(part);
          if (gl == 0) zj = fma(part, part, zj);
        }
        const double z2 = big_block_sum<Cfg>(zj, red, gt);
        const double rs = tol > 0.0 ? tol : 2.220446049250313e-16;
        const double m2 = 1.5625 * rs * rs;
        const double den = tau + w2;
        const double cut = den > 0.0 ? z2 / den : 0.0;
        const double mw2 = fmax(mi2 - cut, 0.0) + 1e-10 * mi2;
        ok = ((1.0 + nl2) * m2 < 1.0) && ((1.0 + tl2) * mw2 * m2 < 1.0);
        __syncthreads();
        for (int idx = tid; idx < 160; idx += NT) vb[idx] = 0.0;  // (the squarings' second buffer again)
        for (int idx = tid; idx < 2 * BIG_GD_SCAP * LDS_; idx += NT) Ts[idx] = 0.0;  // (over the inverse: the guards are done)
        __syncthreads();
        for (int idx = tid; idx < sN * sN; idx += NT) {
          const int i = idx / sN, j = idx - i * sN;
          Ts[i * LDS_ + j] = T[off + (size_t)sidx[i] * n + sidx[j]];
        }
      }
      __syncthreads();
      if (ok) ok = big_certify_contraction<Cfg>(Gm, l, LDG, red, tid);
      if (ok) ok = big_certify_contraction<Cfg>(Ts, sN, LDS_, red, tid);
      if (R_out && !ok) {
        __syncthreads();
        for (int idx = tid; idx < n * k; idx += NT) R_out[offk + idx] = 0.0;
      }
    } else if (R_out) {
      for (int idx = tid; idx < n * k; idx += NT) R_out[offk + idx] = 0.0;
    }
    if (!ok)  // no verdict: nobody gets to use the solvent
      for (int idx = tid; idx < n * n; idx += NT) T[off + idx] = 0.0;
    if (tid == 0) {
      eu_out[3 * draw] = ok ? 1 : -3;
      eu_out[3 * draw + 1] = ok ? 1 : -3;
      eu_out[3 * draw + 2] = 0;
      if (!ok) status[draw] = (conv ? DSGE_ST_NOT_CONVERGED : status[draw]) | DSGE_ST_GENSYS_TOO_BIG;
    }
  }
}

// ---- the filtered variables of the batch: bit j of mask[0..1] = some draw has a non-zero column j in A (a state variable:
// T = -(B + C T)^-1 A has a non-zero column exactly there) or in Z (an observed variable) -------------------------------------------
__global__ __launch_bounds__(128) void big_mask_kernel(const double* __restrict__ A, const double* __restrict__ Z, int z_batched,
                                                       int batch, int n, int p, unsigned long long* __restrict__ mask) {
  const int j = threadIdx.x;
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    bool st = false, ob = false;
    if (j < n) {
      const double* a = A + (size_t)draw * n * n;
      for (int r = 0; r < n; ++r) st = st || (a[(size_t)r * n + j] != 0.0);
      if (z_batched || draw == 0) {
        const double* z = Z + (z_batched ? (size_t)draw * p * n : 0);
        for (int r = 0; r < p; ++r) ob = ob || (z[(size_t)r * n + j] != 0.0);
      }
    }
    const unsigned long long bs = __ballot(st), bo = __ballot(ob);
    if ((threadIdx.x & 63) == 0) {
      const int w = threadIdx.x >> 6;
      if (bs) atomicOr(&mask[w], bs);      // [0..1]: states
      if (bo) atomicOr(&mask[2 + w], bo);  // [2..3]: observed
    }
  }
}

struct BigIndex {
  uint8_t idx[64];
  int u;
};

// gathered model: T_r = T[F, F], R_r = R[F, :], Z_r = Z[:, F]
__global__ __launch_bounds__(256) void big_compress_kernel(const double* __restrict__ T, const double* __restrict__ R,
                                                           const double* __restrict__ Z, int z_batched, int batch, int n, int k,
                                                           int p, BigIndex F, double* __restrict__ T_r, double* __restrict__ R_r,
                                                           double* __restrict__ Z_r) {
  const int u = F.u, tid = threadIdx.x;
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    const double* t = T + (size_t)draw * n * n;
    const double* r = R + (size_t)draw * n * k;
    for (int idx = tid; idx < u * u; idx += 256) {
      const int i = idx / u, j = idx - i * u;
      T_r[(size_t)draw * u * u + idx] = t[(size_t)F.idx[i] * n + F.idx[j]];
    }
    for (int idx = tid; idx < u * k; idx += 256) {
      const int i = idx / k, c = idx - i * k;
      R_r[(size_t)draw * u * k + idx] = r[(size_t)F.idx[i] * k + c];
    }
    if (z_batched || draw == 0) {
      const double* z = Z + (z_batched ? (size_t)draw * p * n : 0);
      double* zr = Z_r + (z_batched ? (size_t)draw * p * u : 0);
      for (int idx = tid; idx < p * u; idx += 256) {
        const int i = idx / u, j = idx - i * u;
        zr[idx] = z[(size_t)i * n + F.idx[j]];
      }
    }
  }
}

}  // namespace dsge

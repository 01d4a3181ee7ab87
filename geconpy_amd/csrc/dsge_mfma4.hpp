// Small dense NT-form products on the FP64 matrix core's 4 x 4 x 4 instruction (round 6).
//
// v_mfma_f64_4x4x4f64 multiplies FOUR independent 4 x 4 x 4 blocks per issue: 256 FMAs in one instruction slot (16.4 cycles
// back to back, 20.7 in a dependent chain; v_fma_f64: 64 FMAs per slot of 4.6 -- tools/mfma_probe/mfma4_rate.hip).  Same peak as the
// VALU, a quarter of the instructions -- and a lone wavefront is bound by its instruction slots.  The
// 16 x 16 x 4 instruction was measured SLOWER on the 18-wide reduced model (round 2: the fragment pads 18 -> 32); 4-wide blocks pad
// 18 -> 20.  Operand / result layout, probed on the device with one-hot operands (tools/mfma_probe/mfma4_layout.hip):
//     lane l:  blk = (l >> 2) & 3     A_blk[i = l & 3][k = l >> 4]     B_blk[k = l >> 4][j = l & 3]     D_blk[i = l >> 4][j = l & 3]
//
// mfma4_nt<KT, TA, TB, LD>:  D[r][c] = sum_k A[r][k] B[c][k]  (both operands row-major along k, even leading dimension LD,
// 16-byte aligned rows -- the layout of the filter's Tc / W' / Pc), r < 4 TA, c < 4 TB, k < 4 KT, everything compile-time so that
// every LDS access is one per-lane base register plus an immediate offset.
//   * tiles: "band 0" = row tiles 0..3, one per block, column tile g in group g (A operands are the same for all TB groups: loaded
//     once); "band 1" = the row tiles 4..TA-1 (TA <= 8), packed 4 / 2 / 1 column tiles per group.
//   * the contraction index is PERMUTED inside the instruction so that one ds_read_b128 feeds two issues: within a pair of k-tiles
//     (8 columns) lane group kq = l >> 4 takes k = 8 c + 2 kq (first issue) and 8 c + 2 kq + 1 (second); an odd last tile is one
//     issue with k = 4 (KT - 1) + kq.  Two accumulators per group alternate (independent issues: 16 instead of 21 cycles each).
//   * operands are read UNPREDICATED: rows up to 4 TA (4 TB) and columns up to 4 KT must be readable, finite, and zero where the
//     mathematical operand ends.
// The consumer gets every group's element through `sink(g, d)`; Mfma4Map tells which tile that is.
#pragma once
#include <hip/hip_runtime.h>

namespace dsge {

template <int TA, int TB>
struct Mfma4Map {
  static_assert(TA >= 1 && TA <= 8 && TB >= 1 && TB <= 8, "1..8 tiles per side");
  static constexpr int G0 = TB;                                  // groups of band 0
  static constexpr int R1 = TA > 4 ? TA - 4 : 0;                 // row tiles of band 1
  static constexpr int RP = R1 == 0 ? 1 : (R1 == 1 ? 1 : (R1 == 2 ? 2 : 4));  // blocks per column tile in band 1
  static constexpr int CPG = 4 / RP;                             // column tiles per group in band 1
  static constexpr int G1 = R1 == 0 ? 0 : (TB + CPG - 1) / CPG;  // groups of band 1
  static constexpr int NG = G0 + G1;
  // row / column tile of this lane's block in group g (g < G0: band 0), and whether the block is a real tile
  __device__ static __forceinline__ int ta(int g, int blk) { return g < G0 ? blk : 4 + (blk % RP); }
  __device__ static __forceinline__ int tb(int g, int blk) { return g < G0 ? g : CPG * (g - G0) + blk / RP; }
  __device__ static __forceinline__ bool live(int g, int blk) { return ta(g, blk) < TA && tb(g, blk) < TB; }
};

template <int KT, int TA, int TB, int LD, class Sink>
__device__ __forceinline__ void mfma4_nt(const double* __restrict__ A, const double* __restrict__ B, int lane, Sink&& sink) {
  using MP = Mfma4Map<TA, TB>;
  constexpr int NPAIR = KT / 2, ODD = KT & 1, NOP = NPAIR + ODD;
  const int blk = (lane >> 2) & 3, i4 = lane & 3, kq = lane >> 4;
  // ---- band 0 ----
  {
    const double* ap = A + (4 * (blk < TA ? blk : 0) + i4) * LD;  // (a block beyond TA recomputes row tile 0: never stored)
    const double* bp = B + i4 * LD;
    double2 a2[NPAIR > 0 ? NPAIR : 1];
    double a1 = 0.0;
#pragma unroll
    for (int c = 0; c < NPAIR; ++c) a2[c] = *reinterpret_cast<const double2*>(ap + 8 * c + 2 * kq);
    if (ODD) a1 = ap[4 * (KT - 1) + kq];
    double2 b2[2][NPAIR > 0 ? NPAIR : 1];
    double b1[2] = {0.0, 0.0};
#pragma unroll
    for (int c = 0; c < NPAIR; ++c) b2[0][c] = *reinterpret_cast<const double2*>(bp + 8 * c + 2 * kq);
    if (ODD) b1[0] = bp[4 * (KT - 1) + kq];
#pragma unroll
    for (int g = 0; g < MP::G0; ++g) {
      const int cur = g & 1, nxt = cur ^ 1;
      if (g + 1 < MP::G0) {  // operands of the next group are requested before this group's issues
#pragma unroll
        for (int c = 0; c < NPAIR; ++c) b2[nxt][c] = *reinterpret_cast<const double2*>(bp + (g + 1) * 4 * LD + 8 * c + 2 * kq);
        if (ODD) b1[nxt] = bp[(g + 1) * 4 * LD + 4 * (KT - 1) + kq];
      }
      __builtin_amdgcn_sched_barrier(0);
      double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
      for (int c = 0; c < NPAIR; ++c) {
        acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a2[c].x, b2[cur][c].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a2[c].y, b2[cur][c].y, acc1, 0, 0, 0);
      }
      if (ODD) acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b1[cur], acc0, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      sink(g, NOP > 1 ? acc0 + acc1 : acc0);
    }
  }
  // ---- band 1 ----
  if constexpr (MP::G1 > 0) {
    const int ta1 = 4 + (blk % MP::RP), tb1 = blk / MP::RP;
    const double* ap = A + (4 * (ta1 < TA ? ta1 : 0) + i4) * LD;
    double2 a2[NPAIR > 0 ? NPAIR : 1];
    double a1 = 0.0;
#pragma unroll
    for (int c = 0; c < NPAIR; ++c) a2[c] = *reinterpret_cast<const double2*>(ap + 8 * c + 2 * kq);
    if (ODD) a1 = ap[4 * (KT - 1) + kq];
#pragma unroll
    for (int g = 0; g < MP::G1; ++g) {
      // (column tiles beyond TB: clamp to tile 0, never stored.  tb1 + CPG g < TB is lane-dependent only in the last group)
      const int tbg = (MP::CPG * g + tb1 < TB) ? MP::CPG * g + tb1 : 0;
      const double* bp = B + (4 * tbg + i4) * LD;
      double2 b2[NPAIR > 0 ? NPAIR : 1];
      double b1 = 0.0;
#pragma unroll
      for (int c = 0; c < NPAIR; ++c) b2[c] = *reinterpret_cast<const double2*>(bp + 8 * c + 2 * kq);
      if (ODD) b1 = bp[4 * (KT - 1) + kq];
      double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
      for (int c = 0; c < NPAIR; ++c) {
        acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a2[c].x, b2[c].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a2[c].y, b2[c].y, acc1, 0, 0, 0);
      }
      if (ODD) acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b1, acc0, 0, 0, 0);
      sink(MP::G0 + g, NOP > 1 ? acc0 + acc1 : acc0);
    }
  }
}

// ---- the upper triangle only (D symmetric by construction of the operands: X = Tc (P Tc')): tiles (ta <= tb), 4 per group --------
template <int TM>
struct Mfma4Upper {
  static constexpr int NT = TM * (TM + 1) / 2, NG = (NT + 3) / 4;
  // tile number t -> (ta, tb), columns first: t = tb (tb + 1) / 2 + ta
  __host__ __device__ static constexpr int tb_of(int t) {
    int tb = 0;
    while ((tb + 1) * (tb + 2) / 2 <= t) ++tb;
    return tb;
  }
  __host__ __device__ static constexpr int ta_of(int t) { return t - tb_of(t) * (tb_of(t) + 1) / 2; }
  // per-lane tile of group g (g is a constant after unrolling: the tile numbers fold; a block beyond the last tile repeats tile 0
  // and is not live): chains of selects on blk, no table in memory
  __device__ static __forceinline__ void tile(int g, int blk, int& ta, int& tb, bool& live) {
    const int t0 = 4 * g, t1 = 4 * g + 1, t2 = 4 * g + 2, t3 = 4 * g + 3;
    const int c0 = t0 < NT ? t0 : 0, c1 = t1 < NT ? t1 : 0, c2 = t2 < NT ? t2 : 0, c3 = t3 < NT ? t3 : 0;
    ta = blk == 0 ? ta_of(c0) : (blk == 1 ? ta_of(c1) : (blk == 2 ? ta_of(c2) : ta_of(c3)));
    tb = blk == 0 ? tb_of(c0) : (blk == 1 ? tb_of(c1) : (blk == 2 ? tb_of(c2) : tb_of(c3)));
    live = blk == 0 ? (t0 < NT) : (blk == 1 ? (t1 < NT) : (blk == 2 ? (t2 < NT) : (t3 < NT)));
  }
};

// D[r][c] (+ init(g)) for the tiles ta <= tb only; sink(g, d): this lane's element is D[4 ta + (l >> 4)][4 tb + (l & 3)] of the tile
// (ta, tb) = Mfma4Upper<TM>::tile(g, blk).  The caller passes the operand rows of its block per group -- rowa[g] = 4 ta + (l & 3),
// rowb[g] = 4 tb + (l & 3) -- computed ONCE (the select chains of tile() compile to branches: recomputed per call they cost more
// than the issues they feed).  The operands of group g + 1 are requested before the issues of group g.  init(g): the accumulator's
// starting value (the lane's element of C in D = A B' + C).
template <int KT, int TM, int LD, class Init, class Sink>
__device__ __forceinline__ void mfma4_nt_upper_acc(const double* __restrict__ A, const double* __restrict__ B, int lane,
                                                   const int (&rowa)[Mfma4Upper<TM>::NG], const int (&rowb)[Mfma4Upper<TM>::NG],
                                                   Init&& init, Sink&& sink) {
  using UX = Mfma4Upper<TM>;
  constexpr int NPAIR = KT / 2, ODD = KT & 1, NG = UX::NG;
  const int kq = lane >> 4;
  double2 a2[2][NPAIR > 0 ? NPAIR : 1], b2[2][NPAIR > 0 ? NPAIR : 1];
  double a1[2] = {0.0, 0.0}, b1[2] = {0.0, 0.0};
#define MFMA4_UP_LOAD(G_, S_)                                                                \
  do {                                                                                       \
    const double* ap_ = A + rowa[G_] * LD;                                                   \
    const double* bp_ = B + rowb[G_] * LD;                                                   \
    _Pragma("unroll") for (int c = 0; c < NPAIR; ++c) {                                      \
      a2[S_][c] = *reinterpret_cast<const double2*>(ap_ + 8 * c + 2 * kq);                   \
      b2[S_][c] = *reinterpret_cast<const double2*>(bp_ + 8 * c + 2 * kq);                   \
    }                                                                                        \
    if (ODD) {                                                                               \
      a1[S_] = ap_[4 * (KT - 1) + kq];                                                       \
      b1[S_] = bp_[4 * (KT - 1) + kq];                                                       \
    }                                                                                        \
  } while (0)
  MFMA4_UP_LOAD(0, 0);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int cur = g & 1, nxt = cur ^ 1;
    if (g + 1 < NG) MFMA4_UP_LOAD(g + 1, nxt);
    double acc0 = init(g), acc1 = 0.0;  // (two chains: the issues of a pair never wait for each other)
#pragma unroll
    for (int c = 0; c < NPAIR; ++c) {
      acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a2[cur][c].x, b2[cur][c].x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a2[cur][c].y, b2[cur][c].y, acc1, 0, 0, 0);
    }
    if (ODD) acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a1[cur], b1[cur], acc0, 0, 0, 0);
    sink(g, NPAIR > 0 ? acc0 + acc1 : acc0);
  }
#undef MFMA4_UP_LOAD
}

// operand rows of this lane's block for every group of the upper-tile map (see mfma4_nt_upper_acc)
template <int TM>
__device__ __forceinline__ void mfma4_upper_rows(int lane, int (&rowa)[Mfma4Upper<TM>::NG], int (&rowb)[Mfma4Upper<TM>::NG]) {
  const int blk = (lane >> 2) & 3, i4 = lane & 3;
#pragma unroll
  for (int g = 0; g < Mfma4Upper<TM>::NG; ++g) {
    int ta, tb;
    bool live;
    Mfma4Upper<TM>::tile(g, blk, ta, tb, live);
    rowa[g] = 4 * ta + i4;
    rowb[g] = 4 * tb + i4;
  }
}

// D = A B (NN form: B row-major along its COLUMNS, B[k][c]) for all TA x TB tiles: the doubling iteration's A_k[:,S] A_k[S,:].
// Same tile map and k permutation as mfma4_nt; the B operand of a k-pair is two ds_read_b64 (rows 8 c + 2 kq and 8 c + 2 kq + 1).
template <int KT, int TA, int TB, int LD, class Sink>
__device__ __forceinline__ void mfma4_nn(const double* __restrict__ A, const double* __restrict__ B, int lane, Sink&& sink) {
  using MP = Mfma4Map<TA, TB>;
  constexpr int NPAIR = KT / 2, ODD = KT & 1, NOP = NPAIR + ODD;
  const int blk = (lane >> 2) & 3, i4 = lane & 3, kq = lane >> 4;
#pragma unroll
  for (int g = 0; g < MP::NG; ++g) {
    const int ta = MP::ta(g, blk), tb = MP::tb(g, blk);
    const int tac = ta < TA ? ta : 0, tbc = tb < TB ? tb : 0;
    const double* ap = A + (4 * tac + i4) * LD;
    const double* bp = B + 4 * tbc + i4;
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int c = 0; c < NPAIR; ++c) {
      const double2 a2 = *reinterpret_cast<const double2*>(ap + 8 * c + 2 * kq);
      const double bx = bp[(8 * c + 2 * kq) * LD], by = bp[(8 * c + 2 * kq + 1) * LD];
      acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a2.x, bx, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a2.y, by, acc1, 0, 0, 0);
    }
    if (ODD) acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(ap[4 * (KT - 1) + kq], bp[(4 * (KT - 1) + kq) * LD], acc0, 0, 0, 0);
    sink(g, NOP > 1 ? acc0 + acc1 : acc0);
  }
}

// D[r][c] for the upper tiles, no accumulator input; sink(g, d, ta, tb, live) (convenience form: computes the rows itself)
template <int KT, int TM, int LD, class Sink>
__device__ __forceinline__ void mfma4_nt_upper(const double* __restrict__ A, const double* __restrict__ B, int lane, Sink&& sink) {
  using UX = Mfma4Upper<TM>;
  int rowa[UX::NG], rowb[UX::NG];
  mfma4_upper_rows<TM>(lane, rowa, rowb);
  const int blk = (lane >> 2) & 3;
  mfma4_nt_upper_acc<KT, TM, LD>(A, B, lane, rowa, rowb, [](int) { return 0.0; }, [&](int g, double d) {
    int ta, tb;
    bool live;
    UX::tile(g, blk, ta, tb, live);
    sink(g, d, ta, tb, live);
  });
}

// ---- strided operands (round 6, the gradient's reverse sweep): D[r][c] = sum_k A[r ARS + k ACS] B[k BRS + c BCS] ----------------
// Any leading dimension and either orientation of each operand (a transposed operand swaps its two strides), one ds_read_b64 per
// operand element -- the LDS bytes per issue are those of the b128 forms above, in twice the instructions.  Same tile maps, same
// two alternating accumulators; the B operands of the next group are requested before the issues of the current one.  Rows up to
// 4 TA / columns up to 4 TB / contraction indices up to 4 KT are read unpredicated.
template <int KT, int TA, int TB, int ARS, int ACS, int BRS, int BCS, class Sink>
__device__ __forceinline__ void mfma4_strided(const double* __restrict__ A, const double* __restrict__ B, int lane, Sink&& sink) {
  using MP = Mfma4Map<TA, TB>;
  const int blk = (lane >> 2) & 3, i4 = lane & 3, kq = lane >> 4;
  auto issue = [&](const double (&a)[KT], const double (&b)[KT]) {
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      if (kt & 1)
        acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[kt], b[kt], acc1, 0, 0, 0);
      else
        acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[kt], b[kt], acc0, 0, 0, 0);
    }
    return KT > 1 ? acc0 + acc1 : acc0;
  };
  {  // ---- band 0: row tile = block, column tile = group (the A operands are the same for every group) ----
    const double* ap = A + (4 * (blk < TA ? blk : 0) + i4) * ARS + kq * ACS;
    const double* bp = B + kq * BRS + i4 * BCS;
    double a[KT], b[2][KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      a[kt] = ap[4 * kt * ACS];
      b[0][kt] = bp[4 * kt * BRS];
    }
#pragma unroll
    for (int g = 0; g < MP::G0; ++g) {
      const int cur = g & 1, nxt = cur ^ 1;
      if (g + 1 < MP::G0) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) b[nxt][kt] = bp[4 * kt * BRS + (g + 1) * 4 * BCS];
      }
      sink(g, issue(a, b[cur]));
    }
  }
  if constexpr (MP::G1 > 0) {  // ---- band 1: row tiles 4 .. TA - 1 ----
    const int ta1 = 4 + (blk % MP::RP), tb1 = blk / MP::RP;
    const double* ap = A + (4 * (ta1 < TA ? ta1 : 0) + i4) * ARS + kq * ACS;
    double a[KT], b[2][KT];
    auto load_b = [&](int g, double (&bb)[KT]) {
      const int tbg = (MP::CPG * g + tb1 < TB) ? MP::CPG * g + tb1 : 0;  // (beyond TB: tile 0, never stored)
      const double* bp = B + kq * BRS + (4 * tbg + i4) * BCS;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) bb[kt] = bp[4 * kt * BRS];
    };
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) a[kt] = ap[4 * kt * ACS];
    load_b(0, b[0]);
#pragma unroll
    for (int g = 0; g < MP::G1; ++g) {
      const int cur = g & 1, nxt = cur ^ 1;
      if (g + 1 < MP::G1) load_b(g + 1, b[nxt]);
      sink(MP::G0 + g, issue(a, b[cur]));
    }
  }
}

// the upper tiles only (ta <= tb; rowa / rowb from mfma4_upper_rows): sink(g, d), d = D[4 ta + (l >> 4)][4 tb + (l & 3)]
template <int KT, int TM, int ARS, int ACS, int BRS, int BCS, class Sink>
__device__ __forceinline__ void mfma4_strided_upper(const double* __restrict__ A, const double* __restrict__ B, int lane,
                                                    const int (&rowa)[Mfma4Upper<TM>::NG], const int (&rowb)[Mfma4Upper<TM>::NG],
                                                    Sink&& sink) {
  constexpr int NG = Mfma4Upper<TM>::NG;
  const int kq = lane >> 4;
  double a[2][KT], b[2][KT];
  auto load = [&](int g, double (&aa)[KT], double (&bb)[KT]) {
    const double* ap = A + rowa[g] * ARS + kq * ACS;
    const double* bp = B + kq * BRS + rowb[g] * BCS;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      aa[kt] = ap[4 * kt * ACS];
      bb[kt] = bp[4 * kt * BRS];
    }
  };
  load(0, a[0], b[0]);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int cur = g & 1, nxt = cur ^ 1;
    if (g + 1 < NG) load(g + 1, a[nxt], b[nxt]);
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      if (kt & 1)
        acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[cur][kt], b[cur][kt], acc1, 0, 0, 0);
      else
        acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[cur][kt], b[cur][kt], acc0, 0, 0, 0);
    }
    sink(g, KT > 1 ? acc0 + acc1 : acc0);
  }
}

}  // namespace dsge

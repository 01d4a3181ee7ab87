// Static-variable deflation in front of cycle reduction.  A variable whose columns of A and C are both exactly zero (it
// appears neither lagged nor led: a "static" variable in Dynare's partition) only enters through B.  A Householder QR of
// those h columns of B, applied to the whole system,
//     Q' [B_st | B_dy | A_dy | C_dy | D] = [ R_st  Btop  Atop  Ctop  Dtop ]     h rows
//                                          [  0    Bred  Ared  Cred  Dred ]     n - h rows
// leaves a quadratic matrix equation  Ared + Bred T_dy + Cred T_dy^2 = 0  in the n - h dynamic variables alone, and the
// static rows follow by back-substitution:
//     T_st = -R_st^-1 ((Btop + Ctop T_dy) T_dy + Atop),      R_st = -R_st^-1 ((Btop + Ctop T_dy) R_dy + Dtop).
// The solution is the one cycle reduction finds on the full system (it is unique); the iteration works on (n - h)^3
// instead of n^3 -- 30 instead of 40 variables on the SW-shaped systems (10 static), 20 of 24 on full_nk, 6 of 12 on the
// two-block RBC golden.  tests/device_models/cr_deflation_model.py restates the algebra in numpy.
// Three launches: cr_deflate_kernel (QR, one wavefront per draw), the existing cycle-reduction kernels on the reduced
// system, cr_inflate_kernel (static rows, scatter to the caller's variable order).  Both kernels keep ONE COLUMN PER LANE
// IN REGISTERS (the reflectors / the small left factors are broadcast from LDS), load it with all rows in flight and
// store it coalesced: the first version -- the whole system in LDS, wave-wide reflectors on it -- took 0.44 + 0.13 ms
// per 4096 SW-shaped draws, a third of what the smaller iteration saved.  `h` is a lower bound of the number of
// static variables that the host passes in (measured once per model size, cr_static_scan_kernel); every draw verifies it and
// a draw with fewer static variables is flagged and solved by the full-size kernels afterwards.
#pragma once
#include "dsge_device.hpp"

#include "../../include/dsge_hip.h"

namespace dsge {

constexpr int CRD_HMAX = 16;  // static variables deflated at most (register-resident rows of B_st / of the solutions)

// per-draw record of the top block: h x (h + 3 nd + k) doubles + 2 (selection mask)
__host__ __device__ inline size_t crd_top_doubles(int n, int k, int h) {
  return (size_t)h * (h + 3 * (n - h) + k) + 2;
}
// dynamic LDS: tile = 8 * BS of the full system (deflate) / of the reduced system (inflate)
__host__ __device__ inline size_t crd_deflate_smem(int tile) {
  return (size_t)(CRD_HMAX * tile + CRD_HMAX) * 8 + (64 + CRD_HMAX) * 4;
}
__host__ __device__ inline size_t crd_inflate_smem(int tile) {
  return (size_t)(2 * CRD_HMAX * tile + CRD_HMAX * CRD_HMAX + CRD_HMAX) * 8 + (64 + CRD_HMAX) * 4;
}

// minimum over the batch of the number of static variables (*out must hold n on entry)
__global__ __launch_bounds__(64) void cr_static_scan_kernel(const double* __restrict__ A, const double* __restrict__ C,
                                                             int batch, int n, int32_t* __restrict__ out) {
  const int lane = threadIdx.x;
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    const size_t off = (size_t)draw * n * n;
    int nz = 0;
    if (lane < n) {
#pragma unroll 8
      for (int i = 0; i < n; ++i)
        nz |= ((A[off + (size_t)i * n + lane] != 0.0) | (C[off + (size_t)i * n + lane] != 0.0)) ? 1 : 0;
    }
    const int h = n - __popcll(__ballot(nz != 0));
    if (lane == 0 && h < __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMin(out, h);
  }
}

// index tables of a selection mask: sti[s] = s-th static variable, dyi[d] = d-th dynamic variable
template <typename IT>
__device__ __forceinline__ void crd_index_tables(unsigned long long smask, int n, int lane, IT* dyi, IT* sti) {
  wave_sync();
  if (lane < n) {
    const unsigned long long bj = 1ull << lane;
    const int below = __popcll(smask & (bj - 1ull));
    if (smask & bj)
      sti[below] = (IT)lane;
    else
      dyi[lane - below] = (IT)lane;
  }
  wave_sync();
}

// which variables are static: lane j looks down column j of A and C (all rows in flight); bit j of the result
template <int NM>
__device__ __forceinline__ unsigned long long crd_static_mask(const double* __restrict__ A, const double* __restrict__ C,
                                                              size_t off, int n, int lane) {
  int nz = 0;
  {
    // unconditional loads (clamped indices): conditional ones compile to one exec-masked block per load
    const int cl = lane < n ? lane : n - 1;
    const double* ap = A + off + cl;
    const double* cp = C + off + cl;
    double av[NM], cv[NM];
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      const int ri = i < n ? i : n - 1;
      av[i] = ap[(size_t)ri * n];
      cv[i] = cp[(size_t)ri * n];
    }
    // (without the barrier the scheduler, short of registers, pairs every load with its compare: 40 serial round
    // trips to HBM, 54k cycles)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NM; ++i) nz |= ((av[i] != 0.0) | (cv[i] != 0.0)) ? 1 : 0;
  }
  const unsigned long long nmask = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
  return ~__ballot(nz != 0) & nmask;
}

// the first h set bits of a mask
__device__ __forceinline__ unsigned long long crd_first_bits(unsigned long long mask, int h) {
  unsigned long long keep = 0ull, rest = mask;
  for (int c = 0; c < h; ++c) {
    const unsigned long long low = rest & (~rest + 1ull);
    keep |= low;
    rest ^= low;
  }
  return keep;
}

// source of column cv of [B_st | B_dy | A_dy | C_dy | D] (h + 3 nd + k columns): pointer to its first row and row stride;
// false for cv beyond the last column (the pointer then walks a valid column)
template <typename IT>
__device__ __forceinline__ bool crd_col_source(int cv, const double* __restrict__ A, const double* __restrict__ B,
                                               const double* __restrict__ C, const double* __restrict__ D, size_t off,
                                               size_t offk, int n, int k, int h, const IT* dyi, const IT* sti,
                                               const double*& src, int& ss) {
  const int nd = n - h;
  src = B + off;
  ss = n;
  if (cv >= h + 3 * nd + k) return false;
  if (cv < h) {
    src = B + off + sti[cv];
    return true;
  }
  const int c = cv - h;
  if (c >= 3 * nd) {
    src = D + offk + (c - 3 * nd);
    ss = k;
    return true;
  }
  const int blk = (c >= nd) + (c >= 2 * nd);
  src = (blk == 0 ? B : (blk == 1 ? A : C)) + off + dyi[c - blk * nd];
  return true;
}

// Q' applied to the 64 NC columns c0 .. c0 + 64 NC - 1 of [B_st | B_dy | A_dy | C_dy | D]: NC columns per lane (column
// c0 + 64 q + lane in col[q]), in registers, all rows of a column loaded in flight.  The first h columns (lanes 0..h-1 of
// the first chunk) are B_st itself: lane j publishes pivot column j to LDS (V), from where every lane (and a later chunk,
// for systems with more columns than one pass holds) reads it as a broadcast; no wave reductions anywhere.  Rows 0..h-1 of
// Q'[...] go to `tp` (h x ncols, row-major); on return col[q] holds the rows of the REDUCED system (rows 0..nd-1, zeros
// below).
template <int NM, int NC, typename IT>
__device__ __forceinline__ void crd_qr_chunk(const double* __restrict__ A, const double* __restrict__ B,
                                             const double* __restrict__ C, const double* __restrict__ D, size_t off,
                                             size_t offk, int n, int k, int h, int c0, const IT* dyi, const IT* sti,
                                             double* V, double* __restrict__ tp, int lane, double (&col)[NC][NM],
                                             bool (&act)[NC], bool skip_zero_ac = false) {
  const int ncols = h + 3 * (n - h) + k;
  const double* src[NC];
  int ss[NC], cv[NC];
#pragma unroll
  for (int q = 0; q < NC; ++q) {
    cv[q] = c0 + 64 * q + lane;
    act[q] = crd_col_source(cv[q], A, B, C, D, off, offk, n, k, h, dyi, sti, src[q], ss[q]);
  }
#pragma unroll
  for (int r = 0; r < NM; ++r) {  // unconditional loads (inactive lanes walk a valid column, rows are clamped)
    const int rr = r < n ? r : n - 1;
#pragma unroll
    for (int q = 0; q < NC; ++q) col[q][r] = src[q][(size_t)rr * ss[q]];
  }
  __builtin_amdgcn_sched_barrier(0);  // all NC * NM loads in flight before the first select waits for one
#pragma unroll
  for (int r = 0; r < NM; ++r)
#pragma unroll
    for (int q = 0; q < NC; ++q) col[q][r] = (r < n && act[q]) ? col[q][r] : 0.0;
  // skip_zero_ac: an all-zero column of A_dy or C_dy stays all-zero under the reflectors; its h top entries are not
  // written (the reader masks them: crd_inflate_* with the column masks) -- 30 of the 107 columns on the SW-shaped system
  bool wr[NC];
#pragma unroll
  for (int q = 0; q < NC; ++q) {
    wr[q] = act[q];
    if (skip_zero_ac) {
      bool nz = false;
#pragma unroll
      for (int r = 0; r < NM; ++r) nz = nz || (col[q][r] != 0.0);
      const int nd_ = n - h;
      const bool ac = cv[q] >= h + nd_ && cv[q] < h + 3 * nd_;
      wr[q] = act[q] && (nz || !ac);
    }
  }
  // Row j is final once reflector j has been applied: it goes straight to the top block and the columns are shifted
  // up by one row, so that the pivot is always register row 0 (no index-dependent selects, which cost a scalar lane
  // mask each) and what is left after h reflectors is the reduced system, rows 0..nd-1.
  // The reflector is never materialised: with u = column j below the pivot (published to LDS by lane j of the first
  // chunk and read back as broadcasts -- 2 x NM v_readlane per sweep cost more than the whole arithmetic),
  // v = [1; scal u], so v'x = x_0 + scal u'x and the raw dots u'x ride along with the norm sweep; the second sweep
  // updates and shifts.  Registers: the columns only.
  for (int j = 0; j < h; ++j) {
    double xn2 = 0.0, alpha;
    double d0[NC], d1[NC];
    if (c0 == 0) {  // publish the pivot column (lane j of the first chunk)
      if (lane == j) {
        double2* vw = reinterpret_cast<double2*>(V + j * NM);
#pragma unroll
        for (int r2 = 0; r2 < NM / 2; ++r2) vw[r2] = make_double2(col[0][2 * r2], col[0][2 * r2 + 1]);
      }
      wave_sync();
    }
    const double2* vj2 = reinterpret_cast<const double2*>(V + j * NM);
    {
      const double2 t0 = vj2[0];
      alpha = t0.x;
      xn2 = t0.y * t0.y;
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        d0[q] = t0.y * col[q][1];
        d1[q] = 0.0;
      }
#pragma unroll
      for (int r2 = 1; r2 < NM / 2; ++r2) {
        const double2 t = vj2[r2];
        xn2 = fma(t.x, t.x, xn2);
#pragma unroll
        for (int q = 0; q < NC; ++q) d1[q] = fma(t.x, col[q][2 * r2], d1[q]);
        xn2 = fma(t.y, t.y, xn2);
#pragma unroll
        for (int q = 0; q < NC; ++q) d0[q] = fma(t.y, col[q][2 * r2 + 1], d0[q]);
      }
    }
    double beta = alpha, scal = 0.0, tj = 0.0;
    if (xn2 != 0.0) {  // dlarfg
      const double nrm = sqrt(fma(alpha, alpha, xn2));
      beta = (alpha >= 0.0) ? -nrm : nrm;
      tj = (beta - alpha) / beta;
      scal = 1.0 / (alpha - beta);
    }
    // w = -tau v'x; the pivot column (beta e_0 exactly) and the columns left of it (all zeros by now) are set directly
    const bool left = (c0 == 0) && (lane <= j);
    const double2 t0 = vj2[0];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      const bool lf = (q == 0) && left;
      const double w = lf ? 0.0 : -tj * fma(scal, d0[q] + d1[q], col[q][0]);
      const double topv = lf ? ((lane == j) ? beta : 0.0) : col[q][0] + w;
      if (wr[q]) tp[(size_t)j * ncols + cv[q]] = topv;
      const double ws = w * scal;
      // update and shift up by one row
      col[q][0] = lf ? 0.0 : fma(t0.y, ws, col[q][1]);
#pragma unroll
      for (int r2 = 1; r2 < NM / 2; ++r2) {
        const double2 t = vj2[r2];
        col[q][2 * r2 - 1] = lf ? 0.0 : fma(t.x, ws, col[q][2 * r2]);
        col[q][2 * r2] = lf ? 0.0 : fma(t.y, ws, col[q][2 * r2 + 1]);
      }
      col[q][NM - 1] = 0.0;
    }
  }
}

template <int BS>
__global__ __launch_bounds__(64) void cr_deflate_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                         const double* __restrict__ C, const double* __restrict__ D,
                                                         int batch, int n, int k, int h, double* __restrict__ Ared,
                                                         double* __restrict__ Bred, double* __restrict__ Cred,
                                                         double* __restrict__ Dred, double* __restrict__ top,
                                                         int32_t* __restrict__ flag) {
  constexpr int NM = 8 * BS, HM = CRD_HMAX;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* V = smem;             // HM x NM: pivot column j (shifted: pivot in slot 0), for the second column chunk
  double* tau_s = V + HM * NM;  // HM (unused)
  int* dyi = (int*)(tau_s + HM);
  int* sti = dyi + 64;
  const int lane = threadIdx.x;
  const int nd = n - h, nv = 3 * nd + k, ncols = h + nv;
  const size_t top_stride = crd_top_doubles(n, k, h);
  {  // one draw per workgroup (grid = batch): a grid-stride loop here makes the compiler hoist every address and
     // comparison of the unrolled row loops out of it, into scalar registers it then has to spill
    const int draw = blockIdx.x;
    if (draw >= batch) return;
    const size_t off = (size_t)draw * n * n, offk = (size_t)draw * n * k;
    const size_t offr = (size_t)draw * nd * nd, offrk = (size_t)draw * nd * k;
    unsigned long long smask = crd_static_mask<NM>(A, C, off, n, lane);
    if (__popcll(smask) < h) {
      // fewer static variables than assumed: hand the draw to the full-size kernels; the reduced system gets a harmless
      // stand-in (B = I, A = C = D = 0) so that the cycle-reduction launch has something finite to chew on
      for (int idx = lane; idx < nd * nd; idx += 64) {
        Ared[offr + idx] = 0.0;
        Cred[offr + idx] = 0.0;
        Bred[offr + idx] = (idx / nd == idx % nd) ? 1.0 : 0.0;
      }
      for (int idx = lane; idx < nd * k; idx += 64) Dred[offrk + idx] = 0.0;
      if (lane == 0) flag[draw] = 1;
      return;
    }
    smask = crd_first_bits(smask, h);  // (the others stay in the dynamic block with their zero columns)
    crd_index_tables(smask, n, lane, dyi, sti);
    double* tp = top + (size_t)draw * top_stride;
    const int ntotc = h + nv;
    const bool multi = ntotc > 128;
    for (int c0 = 0; c0 < ntotc; c0 += 128) {
      double col[2][NM];
      bool act[2];
      crd_qr_chunk<NM, 2>(A, B, C, D, off, offk, n, k, h, c0, dyi, sti, V, tp, lane, col, act);
      if (multi) wave_sync();
      // everything but B_st has rows in the reduced system
      auto destination = [&](int cv, double*& dst, int& ds) -> bool {
        dst = Bred + offr;
        ds = nd;
        if (cv < h || cv >= ntotc) return false;
        const int c = cv - h;
        if (c >= 3 * nd) {
          dst = Dred + offrk + (c - 3 * nd);
          ds = k;
          return true;
        }
        const int blk = (c >= nd) + (c >= 2 * nd);
        dst = (blk == 0 ? Bred : (blk == 1 ? Ared : Cred)) + offr + (c - blk * nd);
        return true;
      };
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        double* dst;
        int ds;
        if (destination(c0 + 64 * q + lane, dst, ds)) {
#pragma unroll
          for (int r = 0; r < NM; ++r)
            if (r < nd) dst[(size_t)r * ds] = col[q][r];
        }
      }
    }
    if (lane == 0) {
      tp[(size_t)h * ncols] = (double)(unsigned)(smask & 0xffffffffull);
      tp[(size_t)h * ncols + 1] = (double)(unsigned)(smask >> 32);
      flag[draw] = 0;
    }
    wave_sync();  // the tables and V are rewritten by the next draw
  }
}

// ---- back-substitution of the static rows (the "inflation"), shared by cr_inflate_kernel and the fused kernel ------------
// LDS arrays (zero padded): Ct (HM x NMD) = Ctop, G1s (HM x NMD) = Btop + Ctop T_dy, Rs (HM x HM) = R_st, rinv (HM) = 1 / diag.
template <int NMD>
struct CrdInflateLds {
  static constexpr int HM = CRD_HMAX;
  static constexpr size_t doubles = (size_t)(2 * HM * NMD + HM * HM + HM);
  double *Ct, *G1s, *Rs, *rinv;
  __device__ __forceinline__ explicit CrdInflateLds(double* base)
      : Ct(base), G1s(base + HM * NMD), Rs(base + 2 * HM * NMD), rinv(base + 2 * HM * NMD + HM * HM) {}
};

// y = column `lane` of [T_dy | R_dy] as loaded (rows >= nd and inactive lanes, y_act = false, are cleared here, behind the
// scheduling barrier, so that the caller's loads and the ones below are all in flight together).  Loads the top block,
// builds the LDS arrays; false if R_st is numerically singular (the verdict is then left to the full-size kernels).
template <int NMD>
__device__ __forceinline__ bool crd_inflate_prepare(double (&y)[NMD], bool y_act, const double* __restrict__ tp, int n, int k,
                                                    int h, int lane, const CrdInflateLds<NMD>& L,
                                                    unsigned long long cmask = ~0ull) {  // bit d: column d of Ctop was stored
  constexpr int HM = CRD_HMAX;
  const int nd = n - h, ncols = h + 3 * nd + k;
  double g[HM], ctv[HM], rsv[(HM * HM + 63) / 64];
  const int ln = lane < nd ? lane : nd - 1;  // column of T_dy / Btop / Ctop this lane looks at (clamped)
#pragma unroll
  for (int i = 0; i < HM; ++i) {
    const size_t row = (size_t)(i < h ? i : h - 1) * ncols;
    g[i] = tp[row + h + ln];
    ctv[i] = tp[row + h + 2 * nd + ln];
  }
#pragma unroll
  for (int u = 0; u < (HM * HM + 63) / 64; ++u) {
    const int idx = u * 64 + lane, i = idx / HM, q = idx % HM;
    rsv[u] = tp[(size_t)(i < h ? i : h - 1) * ncols + (q < h ? q : h - 1)];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < NMD; ++q) y[q] = (q < nd && y_act) ? y[q] : 0.0;
#pragma unroll
  for (int i = 0; i < HM; ++i) {
    g[i] = (i < h && lane < nd) ? g[i] : 0.0;
    if (lane < NMD) L.Ct[i * NMD + lane] = (i < h && lane < nd && ((cmask >> (lane & 63)) & 1ull)) ? ctv[i] : 0.0;
  }
#pragma unroll
  for (int u = 0; u < (HM * HM + 63) / 64; ++u) {
    const int idx = u * 64 + lane, i = idx / HM, q = idx % HM;
    if (idx < HM * HM) L.Rs[idx] = (i < h && q < h) ? rsv[u] : 0.0;
  }
  wave_sync();
  bool bad = false;
  if (lane < HM) {
    const double dg = L.Rs[lane * HM + lane];
    bad = (lane < h) && !(fabs(dg) > 1e-300);
    L.rinv[lane] = (lane < h && !bad) ? 1.0 / dg : 0.0;
  }
  if (__ballot(bad) != 0ull) return false;
  {  // G1 = Btop + Ctop T_dy, column `lane`
#pragma unroll
    for (int i = 0; i < HM; ++i) {
      if (i < h) {
        const double2* cr = reinterpret_cast<const double2*>(L.Ct + i * NMD);
        double e0 = 0.0, e1 = 0.0;
#pragma unroll
        for (int q2 = 0; q2 < NMD / 2; ++q2) {
          const double2 t = cr[q2];
          e0 = fma(t.x, y[2 * q2], e0);
          e1 = fma(t.y, y[2 * q2 + 1], e1);
        }
        g[i] += e0 + e1;
      }
    }
#pragma unroll
    for (int i = 0; i < HM; ++i)
      if (lane < NMD) L.G1s[i * NMD + lane] = (lane < nd) ? g[i] : 0.0;
  }
  wave_sync();
  return true;
}

// Column c = c0 + lane of [T_dy | R_dy] (y: its rows 0..nd-1) -> the static rows by back-substitution, and the whole column
// scattered to the caller's variable order.
template <int NMD, typename IT>
__device__ __forceinline__ void crd_inflate_chunk(int c0, const double (&y)[NMD], const double* __restrict__ tp, int n, int k,
                                                  int h, int lane, const CrdInflateLds<NMD>& L, const IT* dyi, const IT* sti,
                                                  double* __restrict__ Tg, double* __restrict__ Rg,
                                                  unsigned long long amask = ~0ull) {  // bit d: column d of Atop was stored
  constexpr int HM = CRD_HMAX;
  const int nd = n - h, ncols = h + 3 * nd + k, ntot = nd + k;
  const int c = c0 + lane;
  const bool act = c < ntot;
  double x[HM];
  {  // right-hand sides [G1 T_dy + Atop | G1 R_dy + Dtop], column c
    const int cc = act ? c : 0;
    const int tc = h + nd + cc + ((cc >= nd) ? nd : 0);
#pragma unroll
    for (int i = 0; i < HM; ++i) x[i] = tp[(size_t)(i < h ? i : h - 1) * ncols + tc];
    __builtin_amdgcn_sched_barrier(0);
    const bool stored = act && (c >= nd || ((amask >> (c & 63)) & 1ull));
#pragma unroll
    for (int i = 0; i < HM; ++i) x[i] = (i < h && stored) ? x[i] : 0.0;
#pragma unroll
    for (int i = 0; i < HM; ++i) {
      if (i < h) {
        const double2* gr = reinterpret_cast<const double2*>(L.G1s + i * NMD);
        double e0 = 0.0, e1 = 0.0;
#pragma unroll
        for (int q2 = 0; q2 < NMD / 2; ++q2) {
          const double2 t = gr[q2];
          e0 = fma(t.x, y[2 * q2], e0);
          e1 = fma(t.y, y[2 * q2 + 1], e1);
        }
        x[i] += e0 + e1;
      }
    }
  }
  // back-substitution with R_st (zero padded: rows >= h come out as zeros)
#pragma unroll
  for (int i = HM - 1; i >= 0; --i) {
    if (i < h) {
      double acc = x[i];
#pragma unroll
      for (int q = i + 1; q < HM; ++q) acc = fma(-L.Rs[i * HM + q], x[q], acc);
      x[i] = acc * L.rinv[i];
    }
  }
  // scatter to the caller's variable order
  double* dcol = (c < nd) ? (Tg + dyi[act && c < nd ? c : 0]) : (Rg + (c - nd));
  const int ds = (c < nd) ? n : k;
  if (act) {
#pragma unroll
    for (int s2 = 0; s2 < HM; ++s2)
      if (s2 < h) dcol[(size_t)sti[s2] * ds] = -x[s2];
#pragma unroll
    for (int q = 0; q < NMD; ++q)
      if (q < nd) dcol[(size_t)dyi[q] * ds] = y[q];
  }
}

// BSD: tile of the REDUCED system (8 * BSD >= n - h)
template <int BSD>
__global__ __launch_bounds__(64) void cr_inflate_kernel(const double* __restrict__ Tdy, const double* __restrict__ Rdy,
                                                         const double* __restrict__ top, const int32_t* __restrict__ flag,
                                                         int batch, int n, int k, int h, int32_t* __restrict__ status,
                                                         double* __restrict__ T_out, double* __restrict__ R_out) {
  constexpr int NMD = 8 * BSD, HM = CRD_HMAX;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const CrdInflateLds<NMD> L(smem);
  int* dyi = (int*)(smem + CrdInflateLds<NMD>::doubles);
  int* sti = dyi + 64;
  const int lane = threadIdx.x;
  const int nd = n - h, ncols = h + 3 * nd + k, ntot = nd + k;
  const size_t top_stride = crd_top_doubles(n, k, h);
  {  // one draw per workgroup (grid = batch): a grid-stride loop here makes the compiler hoist every address and
     // comparison of the unrolled row loops out of it, into scalar registers it then has to spill
    const int draw = blockIdx.x;
    if (draw >= batch) return;
    const size_t off = (size_t)draw * n * n, offk = (size_t)draw * n * k;
    const size_t offr = (size_t)draw * nd * nd, offrk = (size_t)draw * nd * k;
    if (flag[draw] != 0) {  // not deflated: the full-size kernels solve it next
      if (lane == 0) status[draw] = DSGE_ST_INTERNAL_RERUN;
      return;
    }
    if (status[draw] != 0) {  // the reduced cycle reduction failed: zero policy matrices, as the full-size kernels write
      for (int idx = lane; idx < n * n; idx += 64) T_out[off + idx] = 0.0;
      for (int idx = lane; idx < n * k; idx += 64) R_out[offk + idx] = 0.0;
      return;
    }
    const double* tp = top + (size_t)draw * top_stride;
    const unsigned long long smask = (unsigned long long)(unsigned)tp[(size_t)h * ncols] |
                                     ((unsigned long long)(unsigned)tp[(size_t)h * ncols + 1] << 32);
    crd_index_tables(smask, n, lane, dyi, sti);
    // Every global load of the prologue is issued before the first one is waited for (unconditional loads from clamped
    // addresses, selects behind a scheduling barrier): the kernel is 1.7 k vector instructions per draw and spent 69 % of
    // its 63 k cycles waiting on one load at a time.
    double y[NMD];
    auto load_column = [&](int c) {  // raw (unconditional loads from clamped addresses)
      const bool act = c < ntot;
      const double* src = (act && c >= nd) ? (Rdy + offrk + (c - nd)) : (Tdy + offr + (c < nd ? c : 0));
      const int ss = (act && c >= nd) ? k : nd;
#pragma unroll
      for (int q = 0; q < NMD; ++q) y[q] = src[(size_t)(q < nd ? q : nd - 1) * ss];
    };
    load_column(lane);
    if (!crd_inflate_prepare<NMD>(y, lane < ntot, tp, n, k, h, lane, L)) {
      if (lane == 0) status[draw] = DSGE_ST_INTERNAL_RERUN;
      return;
    }
    // static columns of T are exact zeros
    if (lane < n) {
#pragma unroll
      for (int s2 = 0; s2 < HM; ++s2)
        if (s2 < h) T_out[off + (size_t)lane * n + sti[s2]] = 0.0;
    }
    for (int c0 = 0; c0 < ntot; c0 += 64) {
      if (c0 > 0) {
        load_column(c0 + lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NMD; ++q) y[q] = (q < nd && c0 + lane < ntot) ? y[q] : 0.0;
      }
      crd_inflate_chunk<NMD>(c0, y, tp, n, k, h, lane, L, dyi, sti, T_out + off, R_out + offk);
    }
    wave_sync();  // the LDS tables are rewritten by the next draw
  }
}

}  // namespace dsge

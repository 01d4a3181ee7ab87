// gensys in five launches on the ACTIVE WINDOW of the pencil (same mathematics as dsge_gensys.hpp, which stays as the
// single-launch fallback): the monolithic kernel keeps H, T (N x N complex), Q Pi and Z[:n] in LDS -- 140 KB at N = 52,
// one wavefront per CU.  Here
//   * the z structurally deflated roots (zero columns of A, permuted to the front; a QR of those columns of G0) leave the
//     chip after the reflector phase: rows < z of the pencil are never touched by a row rotation again, and the column
//     rotations they would receive are applied once, at the end, through the accumulated right transformation M;
//   * the reduction to Hessenberg-triangular form runs in REAL storage (the pencil is real until the first complex shift);
//   * the complex QZ iteration + reordering work on the w x w window (w = N - z) only: H22, T22, M (accumulated Z22), X2;
//   * the post-processing uses  Z[:n, :] = P diag(I_z, M)  (P = the column permutation), so that
//       T[state rows]     = Re(M[:, :ns2] Yb Ms^H)[:s']                    (Ms = M[:s'], s' = n - z state variables)
//       T[non-state rows] = R0^-1 (T12[:, :s'] - H12 Re(M1 Yb Ms^H) - X1 Re(Bm B22 Ms2^H))
//     with Yb = A11w^-1 [B11w, B12w - Phi_b B22] the window part of the reference's G0^-1 [Tmat BB] (gensys.py:322-343);
//     tests/device_models/gensys_window_model.py restates this algebra in numpy and checks it against the oracle.
// LDS per draw at N = 52, z = 22: 40 KB (deflation), 25 KB (Hessenberg-triangular), 23 KB (QZ: H and T share one array, M stays in HBM / L2), 15 KB (eu), 72 KB (post) => 4 / 6 / 6-7 / 10 / 2 wavefronts per CU, each on its own
// SIMD.  The launches hand the window over through a library-owned HBM workspace (88 KB per draw, read and written once).
#pragma once
#include "dsge_gensys.hpp"
#include "dsge_householder.hpp"

namespace dsge {

struct GwCaps {
  int n, lcap, wcap, zcap, scap;  // variables; max #lead; max window N - z; max z; max #state variables n - z
};

struct GwOffsets {  // per draw, in doubles
  size_t meta, R0, H12, T12, X1, HR, TR, XR, MRE, HC, TC, MC, XC, BM, PHI, total;
};

__host__ __device__ inline GwOffsets gw_offsets(const GwCaps& c) {
  GwOffsets o;
  size_t p = 0;
  const size_t ww = (size_t)c.wcap * c.wcap, wl = (size_t)c.wcap * c.lcap;
  o.meta = p, p += 8;
  o.R0 = p, p += (size_t)c.zcap * c.zcap;
  o.H12 = p, p += (size_t)c.zcap * c.wcap;
  o.T12 = p, p += (size_t)c.zcap * c.scap;
  o.X1 = p, p += (size_t)c.zcap * c.lcap;
  o.HR = p, p += ww;
  o.TR = p, p += ww;
  o.XR = p, p += wl;
  o.MRE = p, p += ww;  // the accumulated right transformation while it is REAL (two-draws-per-wavefront route): transposed,
                       // element (row, col) at [col * wcap + row] -- half the bytes per streamed column of the complex MC
  p = (p + 1) & ~(size_t)1;  // complex arrays: 16-byte aligned
  o.HC = p, p += 2 * ww;
  o.TC = p, p += 2 * ww;
  o.MC = p, p += 2 * ww;
  o.XC = p, p += 2 * wl;
  o.BM = p, p += 2 * wl;   // Bm (#lead x nu), leading dimension wcap
  o.PHI = p, p += 2 * ww;  // Phi_b (ns2 x nu), leading dimension wcap
  o.total = (p + 1) & ~(size_t)1;
  return o;
}

// meta (int32[16] at the head of a draw's workspace)
enum {
  GW_N = 0, GW_ELL = 1, GW_Z = 2, GW_FLAG = 3, GW_CONV = 4, GW_NS2 = 5, GW_MASK_LO = 6, GW_MASK_HI = 7,
  GW_EU0 = 8, GW_EU1 = 9, GW_EU2 = 10, GW_HAVE_T = 11  // written by the eu launch, read by the post launch
};

// reduce launch A (deflation): the full real pencil H (N x N), the non-zero columns of T (N x w), X (N x #lead);
// reduce launch B (Hessenberg-triangular reduction of the window): H22, T22, Zr (w x w) and X2 (w x #lead).
__host__ __device__ inline size_t gw_reduce_smem(const GwCaps& c) {
  const int Ncap = c.n + c.lcap;
  return ((size_t)Ncap * (Ncap | 1) + (size_t)Ncap * (c.wcap | 1) + (size_t)Ncap * (c.lcap | 1)) * 8 + 64 * 4;
}
__host__ __device__ inline size_t gw_reduce2_smem(const GwCaps& c) {  // [H | X] and T; M stays in the HBM workspace
  return ((size_t)c.wcap * ((c.wcap + c.lcap) | 1) + (size_t)c.wcap * (c.wcap | 1)) * 8;
}
__host__ __device__ inline size_t gw_qz_smem(const GwCaps& c) {  // H and T share one array; M stays in HBM / L2 (GsLayout)
  return ((size_t)c.wcap * ((c.wcap + 4) | 1) + (size_t)c.wcap * (c.lcap | 1)) * 16;
}
__host__ __device__ inline size_t gw_eu_smem(const GwCaps& c) {  // X2, V2 + singular values (Bm is read back from the workspace)
  return ((size_t)c.wcap * (c.lcap | 1) + (size_t)c.lcap * (c.lcap | 1)) * 16 + 128 * 8;
}
__host__ __device__ inline size_t gw_post_smem(const GwCaps& c) {
  const size_t cplx = (size_t)3 * c.wcap * (c.wcap | 1) + (size_t)c.lcap * (c.wcap | 1);
  const size_t real = (size_t)(c.wcap + c.lcap) * (c.scap | 1) + (size_t)c.zcap * (c.scap | 1) +
                      (size_t)c.zcap * (c.zcap | 1) + (size_t)c.zcap * ((c.wcap + c.lcap) | 1);
  return cplx * 16 + real * 8;
}

// act != nullptr: the launch works on the first act[0] slots of the workspace only, and slot i belongs to the caller's draw
// act[1 + i] (gensys by spectral division: the draws without a certificate, compacted by gensys_compact_kernel)
__device__ __forceinline__ int gw_active_count(int batch, const int32_t* act) {
  if (!act) return batch;
  const int na = __builtin_amdgcn_readfirstlane(act[0]);
  return na < batch ? na : batch;
}

#define GW_STAMP(k)                                                         \
  do {                                                                     \
    if (dbg && draw == 0 && lane == 0) dbg[k] = (long long)clock64();        \
  } while (0)

// ---- pre-pass: the shape of the batch (max #lead, max window, max / min z) ----------------------------------------
__global__ __launch_bounds__(64) void gensys_shape_kernel(const double* __restrict__ A, const double* __restrict__ C,
                                                           int batch, int n, double tol, int* __restrict__ out) {
  const int lane = threadIdx.x;
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    const size_t off = (size_t)draw * n * n;
    double cs = 0.0;
    int anz = 0;
    if (lane < n) {
#pragma unroll 8
      for (int i = 0; i < n; ++i) {  // no short-circuit: the loads of a trip are independent
        cs += fabs(C[off + (size_t)i * n + lane]);
        anz |= (A[off + (size_t)i * n + lane] != 0.0) ? 1 : 0;
      }
    }
    const int ell = __popcll(__ballot(lane < n && cs > tol));
    const int z = n - __popcll(__ballot(anz != 0));
    if (lane == 0) {
      // the four records are monotone: a (possibly stale) plain read filters out almost every atomic -- 16 k atomics on four
      // addresses took 0.15 ms
      if (ell > __atomic_load_n(out + 0, __ATOMIC_RELAXED)) atomicMax(out + 0, ell);
      if (n - z + ell > __atomic_load_n(out + 1, __ATOMIC_RELAXED)) atomicMax(out + 1, n - z + ell);
      if (z > __atomic_load_n(out + 2, __ATOMIC_RELAXED)) atomicMax(out + 2, z);
      if (z < __atomic_load_n(out + 3, __ATOMIC_RELAXED)) atomicMin(out + 3, z);
    }
  }
}

__device__ __forceinline__ void rot2r(double& x, double& y, double c, double s) {
  const double tx = fma(c, x, s * y);
  y = fma(c, y, -s * x);
  x = tx;
}

// ---- launch 1a: pencil and structural deflation (real).  Rows < z are final afterwards and leave the chip (R0, H12,
// T12[:, :s'], X1); the window (rows / columns >= z of H and T, rows >= z of X) is handed to launch 1b, which needs a
// third of this launch's LDS and therefore runs at a higher occupancy.
template <int NW>
__global__ __launch_bounds__(64 * NW) void gensys_reduce_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                            const double* __restrict__ C, int batch, GwCaps cp,
                                                            double tol, double* __restrict__ ws,
                                                            long long* __restrict__ dbg, int* __restrict__ obs,
                                                            const int32_t* __restrict__ act) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  batch = gw_active_count(batch, act);
  // NW wavefronts per draw (round 4: 40 KB of LDS allow four draws per CU, and with one wavefront each the launch was four
  // rounds of a 230 k-cycle chain): loads and stores are spread over all threads, the lead / zero-column masks are computed by
  // every wavefront for itself, the reflectors run on hh_left_real_mw
  constexpr int NT = 64 * NW;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = cp.n, Ncap = cp.n + cp.lcap;
  const int ldH = Ncap | 1, ldW = cp.wcap | 1, ldX = cp.lcap | 1;
  double* Hr = smem;
  double* Tr = Hr + (size_t)Ncap * ldH;
  double* Xr = Tr + (size_t)Ncap * ldW;
  int* lead = reinterpret_cast<int*>(Xr + (size_t)Ncap * ldX);
  const size_t total = (size_t)Ncap * ldH + (size_t)Ncap * ldW + (size_t)Ncap * ldX;
  const GwOffsets wo = gw_offsets(cp);

  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {  // (draw: the slot of the workspace; act: the draw behind it)
    const size_t off = (size_t)(act ? act[1 + draw] : draw) * n * n;
    const double* Ag = A + off;
    const double* Bg = B + off;
    const double* Cg = C + off;
    double* wd = ws + (size_t)draw * wo.total;
    int* meta = reinterpret_cast<int*>(wd + wo.meta);
    wave_sync();
    GW_STAMP(14);  // (debug: start of the draw)
    for (size_t idx = tid; idx < total; idx += NT) smem[idx] = 0.0;
    // lead columns (gensys.py:580-589) and the zero columns of A
    int ell = 0;
    unsigned long long a_colmask = 0ull;
    {
      double cs = 0.0;
      int anz = 0;
      if (lane < n) {
#pragma unroll 8
        for (int i = 0; i < n; ++i) {
          cs += fabs(Cg[(size_t)i * n + lane]);
          anz |= (Ag[(size_t)i * n + lane] != 0.0) ? 1 : 0;
        }
      }
      a_colmask = __ballot(anz != 0);
      const unsigned long long lm = __ballot(lane < n && cs > tol);
      ell = __popcll(lm);
      if (wv == 0 && lane < n && ((lm >> lane) & 1ull)) lead[__popcll(lm & ((1ull << lane) - 1ull))] = lane;
    }
    const int N = n + ell;
    const unsigned long long nmask = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
    const unsigned long long zmask = ~a_colmask & nmask;
    const int z = __popcll(zmask);
    const int w = N - z, sp = n - z;
    const bool too_big = (ell > cp.lcap) || (w > cp.wcap) || (z > cp.zcap) || (sp > cp.scap) || (N > 64);
    if (tid == 0) {
      meta[GW_N] = N;
      meta[GW_ELL] = ell;
      meta[GW_Z] = z;
      meta[GW_FLAG] = too_big ? 1 : 0;
      meta[GW_CONV] = 0;
      meta[GW_NS2] = 0;
      meta[GW_MASK_LO] = (int)(unsigned)(a_colmask & 0xffffffffull);
      meta[GW_MASK_HI] = (int)(unsigned)(a_colmask >> 32);
    }
    if (too_big) {
      // the capacity record the launcher cached for this model size is too small for this draw (gensys_shape_cache): tell
      // the next call (rare: no atomic traffic on the normal path)
      if (obs && tid == 0) obs[0] = 1;
      continue;
    }
    wave_sync();
#define COLPOS(c) ((((c) < n) && ((zmask >> (c)) & 1ull)) ? __popcll(zmask & ((1ull << (c)) - 1ull)) \
                                                         : (z + (c) - __popcll(zmask & (((c) >= 64) ? ~0ull : ((1ull << (c)) - 1ull)))))
    // pencil (gensys.py:591-614) by index arithmetic; T holds columns >= z only (the others are exactly zero)
    lane_loop_batched<4, NT>(
        n * n, tid, [&](int idx) { return double2{Bg[idx], Ag[idx]}; },
        [&](int idx, double2 v) {
          const int i = idx / n, j = idx - i * n;
          const int pj = COLPOS(j);
          Hr[i * ldH + pj] = -v.x;
          if (pj >= z) Tr[i * ldW + pj - z] = v.y;
        });
    lane_loop_batched<8, NT>(
        n * ell, tid,
        [&](int idx) {
          const int i = idx / ell, a = idx - i * ell;
          return Cg[(size_t)i * n + lead[a]];
        },
        [&](int idx, double v) {
          const int i = idx / ell, a = idx - i * ell;
          Hr[i * ldH + n + a] = -v;
        });
    if (wv == 0 && lane < ell) {
      const int lc0 = lead[lane];
      Hr[(n + lane) * ldH + COLPOS(lc0)] = 1.0;
      Tr[(n + lane) * ldW + n + lane - z] = 1.0;
      Xr[(n + lane) * ldX + lane] = 1.0;
    }
#undef COLPOS
    wave_sync();

    GW_STAMP(0);
    // ---- structural deflation: QR of the z zero-columns-of-A columns of G0
    for (int j = 0; j < z; ++j) hh_left_real_mw<NW>(Hr, ldH, j, N - j, Tr, ldW, w, Xr, ldX, ell, Hr, ldH, j, j, N, 0, lane, wv);
    wave_sync();
    GW_STAMP(7);  // (debug: end of the reflectors)
    // rows < z are final: R0, H12, T12[:, :s'], X1 leave the chip
    for (int idx = tid; idx < z * z; idx += NT) {
      const int i = idx / z, j = idx - i * z;
      wd[wo.R0 + (size_t)i * cp.zcap + j] = Hr[i * ldH + j];
    }
    for (int idx = tid; idx < z * w; idx += NT) {
      const int i = idx / w, j = idx - i * w;
      wd[wo.H12 + (size_t)i * cp.wcap + j] = Hr[i * ldH + z + j];
    }
    for (int idx = tid; idx < z * sp; idx += NT) {
      const int i = idx / sp, j = idx - i * sp;
      wd[wo.T12 + (size_t)i * cp.scap + j] = Tr[i * ldW + j];
    }
    for (int idx = tid; idx < z * ell; idx += NT) {
      const int i = idx / ell, j = idx - i * ell;
      wd[wo.X1 + (size_t)i * cp.lcap + j] = Xr[i * ldX + j];
    }
    // the window, as it stands after the deflation
    for (int idx = tid; idx < w * w; idx += NT) {
      const int i = idx / w, j = idx - i * w;
      const size_t o = (size_t)i * cp.wcap + j;
      wd[wo.HR + o] = Hr[(z + i) * ldH + z + j];
      wd[wo.TR + o] = Tr[(z + i) * ldW + j];
    }
    for (int idx = tid; idx < w * ell; idx += NT) {
      const int i = idx / ell, j = idx - i * ell;
      wd[wo.XR + (size_t)i * cp.lcap + j] = Xr[(z + i) * ldX + j];
    }
    GW_STAMP(1);
  }
}

// ---- real double-shift accelerator (round 3; model: tests/device_models/gensys_qz_model.py::real_double_shift_stage) ------
// Implicit double-shift QZ sweeps in REAL arithmetic (Moler & Stewart; Golub & Van Loan, Algorithm 7.7.2) on the real
// Hessenberg-triangular window while it is still in this launch's LDS, until every sub-diagonal block has shrunk to 1 x 1 or
// 2 x 2 -- or until anything unusual turns up (a negligible diagonal entry of T inside the active block, 30 sweeps without a
// deflation): the stage then simply stops.  Every transformation is an orthogonal equivalence that keeps the
// Hessenberg-triangular form, so whatever it leaves is a valid input of the complex single-shift iteration of the next launch,
// which owns all of zhgeqz's deflation logic and only has to split the remaining 2 x 2 blocks (12 sweep steps instead of
// 1284 on the SW-shaped window).  A real step is a 3-row reflector from the left and a 3- and a 2-column reflector from the
// right -- about half the FP64 operations of a complex step's two rotations -- and advances two shifts: ~780 steps.
// It runs at the end of the reduction launch (gensys_hesstri_kernel), on that launch's LDS image: [H | X] (w x (w + #lead)) and
// T only -- 18 KB on the SW-shaped window, 8 draws per CU; the step is a chain (two LDS round trips, three reflector
// generations) and VALU-issue bound across the chip, so it lives on occupancy and on its instruction count.  The accumulated
// right transformation M is NOT on the chip: lane = row of M, consecutive steps work on columns k..k+2 and k+1..k+3, so two
// columns are carried in registers, the finished column is stored and the next one prefetched a step ahead (M lives
// transposed in the draw's HBM workspace, as for the complex iteration: a column is one run, one row per lane).
// Mapping: left transformations with one column of [H | X] per lane (42 lanes busy) and one column of T per lane; right
// transformations with one row of H and one row of T per lane; the vectors that define the reflectors travel through
// registers (v_readlane).  Entries outside the bands are exact zeros and stay exact zeros under the reflectors: no masks.
// experiment switch: 1 = skip the rotations whose entry is already zero (two wave-uniform branches per pair), 0 = branch-free
#ifndef GW_HESS_SKIP
#define GW_HESS_SKIP 0
#endif
struct GwHouse {
  double v1, v2, tau, beta;
};
// (I - tau v v') [x y z]' = [beta 0 0]', v = [1 v1 v2]; tau = 0 when y = z = 0.  The chain is what a sweep step waits for, three
// times: 1/||.|| from v_rsq_f64 + two Newton steps, ONE reciprocal (1/(x - beta), v_rcp_f64 + two Newton steps) and
// tau = (beta - x)/beta = (|x| + ||.||)/||.|| as a product -- ~150 dependent cycles instead of ~350 with sqrt() and two divisions.
__device__ __forceinline__ GwHouse gw_house3(double x, double y, double z) {
  // branch-free (a branch would split the sweep step into basic blocks the scheduler cannot interleave) and with ONE select:
  // a vector with y = z = 0, or below 1e-140 in norm, gets tau = 0 (v stays finite, the reflector is the identity; the
  // caller zeroes the targeted entries explicitly anyway).
  GwHouse h;
  const double xn2 = fma(y, y, z * z);
  const double s2 = fma(x, x, xn2);
  const bool ok = xn2 > 0.0 && s2 > 1e-280;
  const double s2c = ok ? s2 : 1.0;
  double rn = __builtin_amdgcn_rsq(s2c);
  rn = rn * fma(-0.5 * s2c * rn, rn, 1.5);
  rn = rn * fma(-0.5 * s2c * rn, rn, 1.5);
  const double nrm = s2c * rn;
  const double d = fabs(x) + nrm;                    // |x - beta|
  const double inv = copysign(fast_rcp(d), x);       // 1 / (x - beta)
  h.beta = ok ? -copysign(nrm, x) : x;
  h.v1 = y * inv;
  h.v2 = z * inv;
  h.tau = ok ? d * rn : 0.0;
  return h;
}

// The sweeps: [H | X] in hb (X(i, j) at column wcap + j, row stride ldH), T in tb (row stride ldW), M transposed and complex
// (real content) in the draw's workspace: element (row, col) at MR[col * mcol + 2 * row].
__device__ __forceinline__ void gw_realqz_sweeps(double* hb, int ldH, double* tb, int ldW, double* MR, size_t mcol, int wcap, int w,
                                                 int ell, int lane, long long* cnt) {
  if (w < 3 || w + ell > 64) return;  // nothing to gain / the packed lane map does not fit: the complex iteration does it all
  constexpr double ULPD = 2.220446049250313e-16, SAFMIN = 2.2250738585072014e-308;
  wave_sync();
  // M transposed and complex (real content): element (row, col) at MR[2 * (col * wcap + row)]
  const bool wa = lane < w, la = lane < w + ell;
  const int cw = min(lane, w - 1);                                        // column of T / row of H, T, M (loads)
  const int ca = lane < w ? lane : (la ? wcap + lane - w : 0);          // column of [H | X]
  double btol;
  {
    double ss = 0.0;
    if (wa)
      for (int i = 0; i <= lane; ++i) {
        const double t = tb[i * ldW + lane];
        ss = fma(t, t, ss);
      }
    btol = fmax(SAFMIN, ULPD * sqrt(wave_sum(ss)));
  }
  int ilast = w - 1, it = 0;
  long long steps = 0, sweeps = 0;
  const int max_total = 40 * w;
  for (int guard = 0; guard < max_total && ilast >= 2; ++guard) {
    wave_sync();
    // ---- deflation tests: one sub-diagonal entry per lane
    double hjj = 0.0, hmm = 0.0, hsub = 0.0, tjj = 1.0;
    if (wa) {
      hjj = hb[lane * ldH + lane];
      tjj = tb[lane * ldW + lane];
      if (lane > 0) {
        hsub = hb[lane * ldH + lane - 1];
        hmm = hb[(lane - 1) * ldH + lane - 1];
      }
    }
    const bool sm = wa && lane > 0 && fabs(hsub) <= fmax(SAFMIN, ULPD * (fabs(hjj) + fabs(hmm)));
    const unsigned long long small = __ballot(sm);
    const unsigned long long tzero = __ballot(wa && fabs(tjj) <= btol);
    if ((small >> ilast) & 1ull) {
      if (lane == 0) hb[ilast * ldH + ilast - 1] = 0.0;
      ilast -= 1;
      it = 0;
      continue;
    }
    if ((small >> (ilast - 1)) & 1ull) {
      if (lane == 0) hb[(ilast - 1) * ldH + ilast - 2] = 0.0;
      ilast -= 2;  // a 2 x 2 block: the complex iteration splits it
      it = 0;
      continue;
    }
    int ifirst = 0;
    {
      const unsigned long long below = small & ((1ull << (ilast - 1)) - 1ull);  // bits 1 .. ilast-2
      if (below) {
        ifirst = 63 - __clzll((long long)below);
        if (lane == 0) hb[ifirst * ldH + ifirst - 1] = 0.0;
      }
    }
    {
      const unsigned long long act = ((ilast >= 63) ? ~0ull : ((1ull << (ilast + 1)) - 1ull)) & ~((1ull << ifirst) - 1ull);
      if (tzero & act) break;  // an infinite root inside the active block: zhgeqz's zero chasing lives in the next launch
    }
    if (++it > 30) break;
    wave_sync();
    // the first three columns of M for this sweep (in flight during the shift arithmetic)
    double m0 = MR[(size_t)ifirst * mcol + 2 * cw], m1 = MR[(size_t)(ifirst + 1) * mcol + 2 * cw],
           m2 = MR[(size_t)(ifirst + 2) * mcol + 2 * cw];
    // ---- the first column of (M - s1)(M - s2), M = H T^-1 on the active block, shifts = roots of the trailing 2 x 2 pencil
    double x, y, z;
    {
      const int m = ilast;
      const double p_ = hb[(m - 1) * ldH + m - 1], q_ = hb[(m - 1) * ldH + m], r_ = hb[m * ldH + m - 1], s_ = hb[m * ldH + m];
      const double e_ = tb[(m - 1) * ldW + m - 1], f_ = tb[(m - 1) * ldW + m], g_ = tb[m * ldW + m];
      double tr, det;
      if (it % 10 == 0) {  // exceptional shifts
        const double w_ = 1.5 * (fabs(r_ / e_) + fabs(hb[(m - 1) * ldH + m - 2] / tb[(m - 2) * ldW + m - 2]));
        tr = w_;
        det = w_ * w_;
      } else {
        tr = p_ / e_ + (s_ - r_ * f_ / e_) / g_;
        det = (p_ * s_ - q_ * r_) / (e_ * g_);
      }
      const int k = ifirst;
      const double a11 = hb[k * ldH + k], a12 = hb[k * ldH + k + 1], a21 = hb[(k + 1) * ldH + k],
                   a22 = hb[(k + 1) * ldH + k + 1], a32 = hb[(k + 2) * ldH + k + 1];
      const double b11 = tb[k * ldW + k], b12 = tb[k * ldW + k + 1], b22 = tb[(k + 1) * ldW + k + 1];
      const double m11 = a11 / b11, m21 = a21 / b11;
      const double y2 = m21 / b22;
      const double y1 = (m11 - b12 * y2) / b11;
      x = a11 * y1 + a12 * y2 - tr * m11 + det;
      y = a21 * y1 + a22 * y2 - tr * m21;
      z = a32 * y2;
    }
    if (!(fabs(x) + fabs(y) + fabs(z) < 1e300)) break;  // NaN / overflow in the shift arithmetic: leave it to zhgeqz's logic
    ++sweeps;
    for (int k = ifirst; k <= ilast - 2; ++k) {
      // the column of M the NEXT step brings in (consumed at the end of this one)
      const double m3 = MR[(size_t)min(k + 3, w - 1) * mcol + 2 * cw];
      // ---- left: rows k .. k+2; lane = column of [H | X] and column of T.  Loads unmasked, stores under one mask each.
      double h0 = hb[k * ldH + ca], h1 = hb[(k + 1) * ldH + ca], h2 = hb[(k + 2) * ldH + ca];
      double t0 = tb[k * ldW + cw], t1 = tb[(k + 1) * ldW + cw], t2 = tb[(k + 2) * ldW + cw];
      const GwHouse q = gw_house3(x, y, z);
      {
        const double sh = q.tau * fma(q.v2, h2, fma(q.v1, h1, h0));
        h0 -= sh;
        h1 = fma(-sh, q.v1, h1);
        h2 = fma(-sh, q.v2, h2);
        const double st = q.tau * fma(q.v2, t2, fma(q.v1, t1, t0));
        t0 -= st;
        t1 = fma(-st, q.v1, t1);
        t2 = fma(-st, q.v2, t2);
      }
      if (k > ifirst && ca == k - 1) {
        h0 = q.beta;
        h1 = 0.0;
        h2 = 0.0;
      }
      // (no store masks: a lane without a column of its own walks a clamped one -- ca = 0, cw = w - 1 -- computes bit for bit
      // what that column's owner computes and stores the same values to the same addresses; the special cases therefore test
      // the COLUMN / ROW index, not the lane.  Three exec-mask branches less per step.)
      hb[k * ldH + ca] = h0;
      hb[(k + 1) * ldH + ca] = h1;
      hb[(k + 2) * ldH + ca] = h2;
      tb[k * ldW + cw] = t0;
      tb[(k + 1) * ldW + cw] = t1;
      tb[(k + 2) * ldW + cw] = t2;
      // Rows k+1 and k+2 of T, columns k .. k+2: ONE right reflector per step, as in LAPACK's dhgeqz -- its first column is
      // the null vector of these two rows, i.e. their cross product, so that column k of T is zero below the diagonal; the
      // entry (k+2, k+1) stays and is part of the next step's 3 x 3 block (the last step of the sweep clears the last one).
      // (Round 3 first chased the bulge with two reflectors, 3 and 2 columns wide: one Householder generation, fifteen
      // FP64 operations and one dependent readlane round more per step.)
      const double a0 = readlane_dyn_f64(t1, k), a1 = readlane_dyn_f64(t1, k + 1), a2 = readlane_dyn_f64(t1, k + 2);
      const double b0 = readlane_dyn_f64(t2, k), b1 = readlane_dyn_f64(t2, k + 1), b2 = readlane_dyn_f64(t2, k + 2);
      wave_sync();
      // ---- right: columns k .. k+2; lane = row of H, of T and of M
      double r0 = hb[cw * ldH + k], r1 = hb[cw * ldH + k + 1], r2 = hb[cw * ldH + k + 2];
      double u0 = tb[cw * ldW + k], u1 = tb[cw * ldW + k + 1], u2 = tb[cw * ldW + k + 2];
      {
        const double w0 = fma(a1, b2, -(a2 * b1)), w1 = fma(a2, b0, -(a0 * b2)), w2 = fma(a0, b1, -(a1 * b0));
        const GwHouse g1 = gw_house3(w0, w1, w2);  // v = [1 v1 v2] on columns k, k+1, k+2: (I - tau v v') e1 = w / beta
        const double sh = g1.tau * fma(g1.v2, r2, fma(g1.v1, r1, r0));
        r0 -= sh;
        r1 = fma(-sh, g1.v1, r1);
        r2 = fma(-sh, g1.v2, r2);
        const double st = g1.tau * fma(g1.v2, u2, fma(g1.v1, u1, u0));
        u0 -= st;
        u1 = fma(-st, g1.v1, u1);
        u2 = fma(-st, g1.v2, u2);
        const double sz = g1.tau * fma(g1.v2, m2, fma(g1.v1, m1, m0));
        m0 -= sz;
        m1 = fma(-sz, g1.v1, m1);
        m2 = fma(-sz, g1.v2, m2);
        if (cw == k + 1 || cw == k + 2) u0 = 0.0;
      }
      hb[cw * ldH + k] = r0;
      hb[cw * ldH + k + 1] = r1;
      hb[cw * ldH + k + 2] = r2;
      tb[cw * ldW + k] = u0;
      tb[cw * ldW + k + 1] = u1;
      tb[cw * ldW + k + 2] = u2;
      MR[(size_t)k * mcol + 2 * cw] = m0;  // column k of M is final for this sweep
      x = readlane_dyn_f64(r0, k + 1);
      y = readlane_dyn_f64(r0, k + 2);
      {
        const double z3 = readlane_dyn_f64(r0, min(k + 3, 63));
        z = (k + 3 <= ilast) ? z3 : 0.0;
      }
      m0 = m1;
      m1 = m2;
      m2 = m3;
      wave_sync();
      ++steps;
    }
    {  // ---- the last step of the sweep: two rows, two columns (m0, m1 = columns ilast-1, ilast of M)
      const int k = ilast - 1;
      double h0 = hb[k * ldH + ca], h1 = hb[(k + 1) * ldH + ca];
      double t0 = tb[k * ldW + cw], t1 = tb[(k + 1) * ldW + cw];
      const GwHouse q = gw_house3(x, y, 0.0);
      {
        const double sh = q.tau * fma(q.v1, h1, h0);
        h0 -= sh;
        h1 = fma(-sh, q.v1, h1);
        const double st = q.tau * fma(q.v1, t1, t0);
        t0 -= st;
        t1 = fma(-st, q.v1, t1);
      }
      if (ca == k - 1) {
        h0 = q.beta;
        h1 = 0.0;
      }
      hb[k * ldH + ca] = h0;
      hb[(k + 1) * ldH + ca] = h1;
      tb[k * ldW + cw] = t0;
      tb[(k + 1) * ldW + cw] = t1;
      const double c0 = readlane_dyn_f64(t1, k), c1 = readlane_dyn_f64(t1, k + 1);
      wave_sync();
      double r0 = hb[cw * ldH + k], r1 = hb[cw * ldH + k + 1];
      double u0 = tb[cw * ldW + k], u1 = tb[cw * ldW + k + 1];
      const GwHouse g2 = gw_house3(c1, c0, 0.0);
      const double sh = g2.tau * fma(g2.v1, r0, r1);
      r0 = fma(-sh, g2.v1, r0);
      r1 -= sh;
      const double st = g2.tau * fma(g2.v1, u0, u1);
      u0 = fma(-st, g2.v1, u0);
      u1 -= st;
      const double sz = g2.tau * fma(g2.v1, m0, m1);
      m0 = fma(-sz, g2.v1, m0);
      m1 -= sz;
      if (cw == k + 1) {
        u0 = 0.0;
        u1 = g2.beta;
      }
      hb[cw * ldH + k] = r0;
      hb[cw * ldH + k + 1] = r1;
      tb[cw * ldW + k] = u0;
      tb[cw * ldW + k + 1] = u1;
      MR[(size_t)k * mcol + 2 * cw] = m0;
      MR[(size_t)(k + 1) * mcol + 2 * cw] = m1;
      ++steps;
    }
  }
  wave_sync();
  if (cnt && lane == 0) {
    cnt[0] = steps;
    cnt[1] = sweeps;
  }
}

// branch-free real Givens for the inner loop of the reduction (selects only): c, s, r with [c s; -s c] [f; g] = [r; 0];
// f = g = 0 (or both below 1e-140) gives the identity.  g = 0 gives c = 1, s = 0, r = f exactly.
__device__ __forceinline__ void gw_lartg(double f, double g, double& c, double& s, double& r) {
  const double d2 = fma(f, f, g * g);
  const bool ok = d2 > 1e-280;
  const double d2c = ok ? d2 : 1.0;
  double rn = __builtin_amdgcn_rsq(d2c);
  rn = rn * fma(-0.5 * d2c * rn, rn, 1.5);
  rn = rn * fma(-0.5 * d2c * rn, rn, 1.5);
  const double d = d2c * rn;
  const double cf = fabs(f) * rn;
  c = ok ? cf : 1.0;
  s = ok ? copysign(g * rn, f * g) : 0.0;   // sign(f) g / d
  r = ok ? copysign(d, f) : f;
  // (f = 0 exactly: copysign(., +0) keeps s = |g| / d * sign(g)... handled: f * g = 0 -> s = +|g| rn; r = +d)
}

// ---- launch 1b: Hessenberg-triangular reduction of the window (real): T22 -> upper triangular by reflectors, H22 -> upper
// Hessenberg by Givens pairs, then (real_stage) the real double-shift sweeps above.  On the chip: [H | X] and T only (18 KB on
// the SW-shaped window: 8 draws per CU).  The accumulated right transformation M lives transposed and complex in the draw's
// HBM workspace (as the complex iteration wants it): lane = row of M, the column shared by two consecutive column rotations
// is carried in a register, the next one is prefetched a rotation ahead, each finished column is stored once.
__global__ __launch_bounds__(64) void gensys_hesstri_kernel(int batch, GwCaps cp, double* __restrict__ ws,
                                                             long long* __restrict__ dbg, int real_stage, int m_real,
                                                             const int32_t* __restrict__ act) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  batch = gw_active_count(batch, act);
  const int lane = threadIdx.x;
  const int ldH = (cp.wcap + cp.lcap) | 1, ldW = cp.wcap | 1;
  double* hb = smem;                          // [H | X]: X(i, j) at column wcap + j
  double* tb = hb + (size_t)cp.wcap * ldH;
  double* xb = hb + cp.wcap;
  const int ldX = ldH;
  const GwOffsets wo = gw_offsets(cp);
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    double* wd = ws + (size_t)draw * wo.total;
    const int* meta = reinterpret_cast<const int*>(wd + wo.meta);
    if (meta[GW_FLAG] != 0) continue;
    const int ell = meta[GW_ELL], w = meta[GW_N] - meta[GW_Z];
    // m_real: the pair launch follows and keeps M real (MRE, stride 1) until the complex iteration needs it
    double* MR = wd + (m_real ? wo.MRE : wo.MC);
    const size_t mcol = (m_real ? 1 : 2) * (size_t)cp.wcap;
    const int ms = m_real ? 1 : 2;
    wave_sync();
    GW_STAMP(5);
    lane_loop_batched<4>(
        w * w, lane,
        [&](int idx) {
          const int i = idx / w, j = idx - i * w;
          const size_t o = (size_t)i * cp.wcap + j;
          return double2{wd[wo.HR + o], wd[wo.TR + o]};
        },
        [&](int idx, double2 v) {
          const int i = idx / w, j = idx - i * w;
          hb[i * ldH + j] = v.x;
          tb[i * ldW + j] = v.y;
        });
    lane_loop_batched<8>(
        w * ell, lane,
        [&](int idx) {
          const int i = idx / ell, j = idx - i * ell;
          return wd[wo.XR + (size_t)i * cp.lcap + j];
        },
        [&](int idx, double v) {
          const int i = idx / ell, j = idx - i * ell;
          xb[i * ldX + j] = v;
        });
    if (lane < w) {  // M = I, every lane writing the row it will keep reading and writing (no cross-lane traffic through HBM)
      if (m_real) {
        for (int col = 0; col < w; ++col) MR[(size_t)col * mcol + lane] = (lane == col) ? 1.0 : 0.0;
      } else {
        cx* MC = reinterpret_cast<cx*>(MR);
        for (int col = 0; col < w; ++col) MC[(size_t)col * cp.wcap + lane] = mk(lane == col ? 1.0 : 0.0, 0.0);
      }
    }
    wave_sync();
    // ---- T22 -> upper triangular (reflectors)
    for (int j = 0; j < w - 1; ++j) hh_left_real(hb, ldH, 0, w, tb, ldW, w, xb, ldX, ell, tb, ldW, j, j, w, lane);
    GW_STAMP(2);
    // ---- window: H22 -> upper Hessenberg by Givens pairs (column j prefetched, pivots travelling through registers).
    // Row rotations with one column of [H | X] and one column of T per lane, column rotations with one row of H, T and M per
    // lane; loads with clamped indices, stores unmasked (a lane without a column / row of its own duplicates the clamped one).
    const int cw = min(lane, w - 1);
    const bool packed = w + ell <= 64;
    const int ca = lane < w ? lane : ((packed && lane < w + ell) ? cp.wcap + lane - w : 0);
    for (int j = 0; j < w - 2; ++j) {
      wave_sync();
      const double colv = hb[cw * ldH + j];
      double g = readlane_dyn_f64(colv, w - 1);
      double m_hi = MR[(size_t)(w - 1) * mcol + ms * cw];  // column w-1 of M, this lane's row
      double m_nx = MR[(size_t)(w - 2) * mcol + ms * cw];  // (the partner column of the NEXT rotation is always in flight)
      for (int i = w - 1; i > j + 1; --i) {
        const double m_lo_in = m_nx;
        m_nx = MR[(size_t)max(i - 2, 0) * mcol + ms * cw];
        const double f = readlane_dyn_f64(colv, i - 1);
        if (GW_HESS_SKIP && g == 0.0) {  // nothing to annihilate: the column pair of M moves on unrotated
          MR[(size_t)i * mcol + ms * cw] = m_hi;
          m_hi = m_lo_in;
          g = f;
          continue;
        }
        wave_sync();
        double hx = hb[(i - 1) * ldH + ca], hy = hb[i * ldH + ca];
        double tx = tb[(i - 1) * ldW + cw], ty = tb[i * ldW + cw];
        double c, s, r;
        gw_lartg(f, g, c, s, r);
        rot2r(hx, hy, c, s);
        rot2r(tx, ty, c, s);
        if (ca == j) {
          hx = r;
          hy = 0.0;
        }
        // (no store masks, as in the sweeps: lanes without a column / row of their own duplicate the clamped one bit for bit)
        hb[(i - 1) * ldH + ca] = hx;
        hb[i * ldH + ca] = hy;
        tb[(i - 1) * ldW + cw] = tx;
        tb[i * ldW + cw] = ty;
        if (!packed) {  // (window + #lead > 64: X in a pass of its own)
          for (int c0 = lane; c0 < ell; c0 += 64) {
            double ax = xb[(i - 1) * ldX + c0], ay = xb[i * ldX + c0];
            rot2r(ax, ay, c, s);
            xb[(i - 1) * ldX + c0] = ax;
            xb[i * ldX + c0] = ay;
          }
        }
        g = r;
        const double tii = readlane_dyn_f64(ty, i), tim = readlane_dyn_f64(ty, i - 1);
        double m_lo = m_lo_in;
        if (!GW_HESS_SKIP || tim != 0.0) {
          wave_sync();
          double hx2 = hb[cw * ldH + i], hy2 = hb[cw * ldH + i - 1];
          double tx2 = tb[cw * ldW + i], ty2 = tb[cw * ldW + i - 1];
          double r2;
          gw_lartg(tii, tim, c, s, r2);
          rot2r(hx2, hy2, c, s);
          rot2r(tx2, ty2, c, s);
          rot2r(m_hi, m_lo, c, s);
          if (cw == i) {
            tx2 = r2;
            ty2 = 0.0;
          }
          hb[cw * ldH + i] = hx2;
          hb[cw * ldH + i - 1] = hy2;
          tb[cw * ldW + i] = tx2;
          tb[cw * ldW + i - 1] = ty2;
        }
        MR[(size_t)i * mcol + ms * cw] = m_hi;  // column i of M is final for this j
        m_hi = m_lo;
      }
      MR[(size_t)(j + 1) * mcol + ms * cw] = m_hi;
    }
    wave_sync();
    GW_STAMP(3);
    if (real_stage) gw_realqz_sweeps(hb, ldH, tb, ldW, MR, mcol, cp.wcap, w, ell, lane, (dbg && draw == 0) ? dbg + 27 : nullptr);
    wave_sync();
    GW_STAMP(6);
    for (int idx = lane; idx < w * w; idx += 64) {
      const int i = idx / w, j = idx - i * w;
      const size_t o = (size_t)i * cp.wcap + j;
      wd[wo.HR + o] = hb[i * ldH + j];
      wd[wo.TR + o] = tb[i * ldW + j];
    }
    for (int idx = lane; idx < w * ell; idx += 64) {
      const int i = idx / ell, j = idx - i * ell;
      wd[wo.XR + (size_t)i * cp.lcap + j] = xb[i * ldX + j];
    }
    GW_STAMP(4);
  }
}


// ---- launch 2: complex single-shift QZ + reordering on the window (qz_iterate / reorder_stable_first of dsge_gensys.hpp
// with N := w, ilo := 0; "Ztop" := the accumulated right transformation M, started from the real phase's Zr) -----------
__global__ __launch_bounds__(64) void gensys_qzwin_kernel(int batch, GwCaps cp, double tol, double* __restrict__ ws,
                                                           long long* __restrict__ dbg, int direct_blocks,
                                                           const int32_t* __restrict__ act) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  batch = gw_active_count(batch, act);
  const int lane = threadIdx.x;
  GsLayout L;
  L.ldh = (cp.wcap + 4) | 1;
  L.ldx = cp.lcap | 1;
  L.ldz = cp.wcap;
  L.H = reinterpret_cast<cx*>(smem);
  L.T = L.H;  // packed map: H(i, j) at [i][j + 4], T(i, j) at [j][i]
  L.hoff = 4;
  L.tsi = 1;
  L.tsj = L.ldh;
  L.packed = true;
  L.zglobal = true;  // M: transposed, in the draw's workspace (set per draw)
  L.halfwave = cp.wcap <= 32;
  L.Z = nullptr;
  L.X = L.H + (size_t)cp.wcap * L.ldh;
  L.V1 = L.V2 = L.S3 = nullptr;
  L.s1 = L.s2 = nullptr;
  L.lead = nullptr;
  const double rs = (tol > 0.0) ? tol : 2.220446049250313e-16;
  const GwOffsets wo = gw_offsets(cp);
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    double* wd = ws + (size_t)draw * wo.total;
    int* meta = reinterpret_cast<int*>(wd + wo.meta);
    if (meta[GW_FLAG] != 0) continue;
    const int ell = meta[GW_ELL], w = meta[GW_N] - meta[GW_Z];
    L.N = w;
    L.n = w;
    L.ell = ell;
    L.xw = ell;
    L.Z = reinterpret_cast<cx*>(wd + wo.MC);
    wave_sync();
    for (int idx = lane; idx < w * L.ldh; idx += 64) L.H[idx] = mk(0.0, 0.0);
    wave_sync();
    lane_loop_batched<4>(
        w * w, lane,
        [&](int idx) {
          const int i = idx / w, j = idx - i * w;
          const size_t o = (size_t)i * cp.wcap + j;
          return double2{wd[wo.HR + o], wd[wo.TR + o]};
        },
        [&](int idx, double2 v) {
          const int i = idx / w, j = idx - i * w;
          hput(L, i, j, mk(v.x, 0.0));  // out-of-band entries are exact zeros after the real reduction
          tput(L, i, j, mk(v.y, 0.0));
        });
    lane_loop_batched<8>(
        w * ell, lane,
        [&](int idx) {
          const int i = idx / ell, j = idx - i * ell;
          return wd[wo.XR + (size_t)i * cp.lcap + j];
        },
        [&](int idx, double v) {
          const int i = idx / ell, j = idx - i * ell;
          L.X[i * L.ldx + j] = mk(v, 0.0);
        });
    wave_sync();
    GW_STAMP(8);
    // round 4: the isolated 2 x 2 blocks the real stage leaves are triangularised in closed form; zhgeqz's iteration only
    // runs when something is left for it
    const bool direct = direct_blocks && qz_direct_blocks(L, lane);
    const bool converged = direct ? true : qz_iterate(L, 0, lane, (dbg && draw == 0) ? dbg + 12 : nullptr);
    GW_STAMP(9);
    const int ns2 = converged ? reorder_stable_first(L, rs, lane) : 0;
    wave_sync();
    GW_STAMP(10);
    cx* HC = reinterpret_cast<cx*>(wd + wo.HC);
    cx* TC = reinterpret_cast<cx*>(wd + wo.TC);
    cx* XC = reinterpret_cast<cx*>(wd + wo.XC);
    for (int idx = lane; idx < w * w; idx += 64) {
      const int i = idx / w, j = idx - i * w;
      const size_t o = (size_t)i * cp.wcap + j;
      HC[o] = hget(L, i, j);
      TC[o] = tget(L, i, j);
    }
    for (int idx = lane; idx < w * ell; idx += 64) {
      const int i = idx / ell, j = idx - i * ell;
      XC[(size_t)i * cp.lcap + j] = L.X[i * L.ldx + j];
    }
    if (lane == 0) {
      meta[GW_CONV] = converged ? 1 : 0;
      meta[GW_NS2] = ns2;
    }
    GW_STAMP(11);
  }
}

// ---- one-sided Jacobi SVD with a round-robin (tournament) ordering: the nc/2 disjoint column pairs of a round are
// rotated at the same time, `lpp` lanes per pair (each lane owns the rows sub, sub + lpp, ... of G and V), the three inner
// products are reduced inside the lane group by DPP butterflies.  Same rotation formula, threshold and stopping rule as
// jacobi_svd (dsge_gensys.hpp), which visits the pairs one after the other with four wave-wide sums each; the singular
// values / vectors agree up to order and rounding.  G: nr x nc (row-major, ld ldg), V: nc x nc accumulates the right
// transformation, sig[j] = ||column j|| on exit.
template <int STEP>
__device__ __forceinline__ double group_xor_sum(double v) {
  if constexpr (STEP == 1) return v + dpp_move_f64<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  if constexpr (STEP == 2) return v + dpp_move_f64<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  if constexpr (STEP == 4) return v + dpp_move_f64<0x141, 0xf>(v);  // row_half_mirror (quads already uniform)
  if constexpr (STEP == 8) return v + dpp_move_f64<0x140, 0xf>(v);  // row_mirror (halves already uniform)
  return v + __shfl_xor(v, STEP, 64);
}
__device__ __forceinline__ double group_sum(double v, int lpp) {
  if (lpp > 1) v = group_xor_sum<1>(v);
  if (lpp > 2) v = group_xor_sum<2>(v);
  if (lpp > 4) v = group_xor_sum<4>(v);
  if (lpp > 8) v = group_xor_sum<8>(v);
  if (lpp > 16) v = group_xor_sum<16>(v);
  if (lpp > 32) v = group_xor_sum<32>(v);
  return v;
}

__device__ __forceinline__ void jacobi_svd_rr(cx* G, int ldg, int nr, int nc, cx* V, int ldv, double* sig, int lane) {
  for (int idx = lane; idx < nc * nc; idx += 64) {
    const int i = idx / nc, j = idx - i * nc;
    V[i * ldv + j] = mk(i == j ? 1.0 : 0.0, 0.0);
  }
  wave_sync();
  const int ncp = nc + (nc & 1), npairs = ncp >> 1;
  if (nc >= 2) {
    int lpp = 1;
    while (2 * lpp * npairs <= 64) lpp *= 2;
    const int pid = lane / lpp, sub = lane - pid * lpp;
    for (int sweep = 0; sweep < 60; ++sweep) {
      bool rotated = false;
      for (int t = 0; t < ncp - 1; ++t) {
        // round t of the tournament: pair 0 = (ncp-1, t); pair k = ((t+k) mod (ncp-1), (t-k) mod (ncp-1))
        int p = 0, q = 0;
        bool act = pid < npairs;
        if (act) {
          if (pid == 0) {
            p = t;
            q = ncp - 1;
          } else {
            p = (t + pid) % (ncp - 1);
            q = (t + (ncp - 1) - pid) % (ncp - 1);
            if (p > q) {
              const int tmp = p;
              p = q;
              q = tmp;
            }
          }
          act = q < nc;  // the padding column of an odd nc sits out
        }
        double al = 0.0, be = 0.0, gr = 0.0, gi = 0.0;
        if (act)
          for (int row = sub; row < nr; row += lpp) {
            const cx gp = G[row * ldg + p], gq = G[row * ldg + q];
            al += gp.re * gp.re + gp.im * gp.im;
            be += gq.re * gq.re + gq.im * gq.im;
            gr += gp.re * gq.re + gp.im * gq.im;  // conj(gp) * gq
            gi += gp.re * gq.im - gp.im * gq.re;
          }
        al = group_sum(al, lpp);
        be = group_sum(be, lpp);
        gr = group_sum(gr, lpp);
        gi = group_sum(gi, lpp);
        const double ag2 = fma(gr, gr, gi * gi);
        const bool rot = act && !(ag2 < 1e-290 || ag2 <= 1e-30 * al * be);
        if (__ballot(rot) != 0ull) rotated = true;
        if (rot) {
          const double inv_ag = fast_rsqrt(ag2);
          const cx phc = mk(gr * inv_ag, -gi * inv_ag);  // conj(phase)
          const double zeta = 0.5 * (be - al) * inv_ag;
          const double z2 = fma(zeta, zeta, 1.0);
          double tt;
          if (z2 < 1e280) {
            const double root = z2 * fast_rsqrt(z2);
            tt = ((zeta >= 0.0) ? 1.0 : -1.0) * fast_rcp(fabs(zeta) + root);
          } else {
            tt = 0.5 / zeta;
          }
          const double cs = fast_rsqrt(fma(tt, tt, 1.0)), sn = cs * tt;
          for (int row = sub; row < nr; row += lpp) {
            const cx gp = G[row * ldg + p], gq = G[row * ldg + q] * phc;
            G[row * ldg + p] = cs * gp - sn * gq;
            G[row * ldg + q] = sn * gp + cs * gq;
          }
          for (int row = sub; row < nc; row += lpp) {
            const cx gp = V[row * ldv + p], gq = V[row * ldv + q] * phc;
            V[row * ldv + p] = cs * gp - sn * gq;
            V[row * ldv + q] = sn * gp + cs * gq;
          }
        }
        wave_sync();
      }
      if (!rotated) break;
    }
  }
  // column norms: one column per lane group would need another reduction tree; nc wave sums are cheap enough
  for (int j = 0; j < nc; ++j) {
    double al = 0.0;
    for (int row = lane; row < nr; row += 64) {
      const cx gp = G[row * ldg + j];
      al += gp.re * gp.re + gp.im * gp.im;
    }
    al = wave_sum_dpp(al);
    if (lane == 0) sig[j] = sqrt(al);
  }
  wave_sync();
}

// sum_{k = lo}^{hi-1} term(k) on FOUR accumulators, the operands of a trip requested together: the run-time-bounded inner
// products of the tail kernels waited for their own pair of LDS loads in every term (one accumulator, no unrolling).
template <class F>
__device__ __forceinline__ cx gw_csum4(int lo, int hi, F term) {
  cx s0 = mk(0, 0), s1 = s0, s2 = s0, s3 = s0;
  int k = lo;
  for (; k + 4 <= hi; k += 4) {
    const cx t0 = term(k), t1 = term(k + 1), t2 = term(k + 2), t3 = term(k + 3);
    s0 = s0 + t0;
    s1 = s1 + t1;
    s2 = s2 + t2;
    s3 = s3 + t3;
  }
  for (; k < hi; ++k) s0 = s0 + term(k);
  return (s0 + s1) + (s2 + s3);
}
template <class F>
__device__ __forceinline__ double gw_sum4(int lo, int hi, F term) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int k = lo;
  for (; k + 4 <= hi; k += 4) {
    const double t0 = term(k), t1 = term(k + 1), t2 = term(k + 2), t3 = term(k + 3);
    s0 += t0;
    s1 += t1;
    s2 += t2;
    s3 += t3;
  }
  for (; k < hi; ++k) s0 += term(k);
  return (s0 + s1) + (s2 + s3);
}

// ---- launch 3: existence / uniqueness (gensys.py:267-310): coincident zeros, SVD of Q2 Pi, the eu codes, Bm and Phi_b.
// Needs only X2 (w x #lead), V2 and Bm on the chip (15 KB at N = 52 => 10 draws per CU): the Jacobi sweeps are a chain
// of short dependent steps, so occupancy is what makes them cheap.
__global__ __launch_bounds__(64) void gensys_eu_kernel(int batch, GwCaps cp, double tol, double* __restrict__ ws,
                                                        long long* __restrict__ dbg, const int32_t* __restrict__ act) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  batch = gw_active_count(batch, act);
  const int lane = threadIdx.x;
  const int ldx = cp.lcap | 1;
  cx* Xc = reinterpret_cast<cx*>(smem);
  cx* V2 = Xc + (size_t)cp.wcap * ldx;  // lcap x ldx
  // (Bm, lcap x nu, is not kept on the chip: Phi_b reads it back from the draw's workspace -- L2 -- which takes the launch from
  // 15.7 to 9.7 KB per draw on the SW-shaped window: 16 draws per CU, the whole batch resident at once)
  double* s1 = reinterpret_cast<double*>(V2 + (size_t)cp.lcap * ldx);
  double* s2 = s1 + 64;
  const double rs = (tol > 0.0) ? tol : 2.220446049250313e-16;
  const GwOffsets wo = gw_offsets(cp);
#define PX(i, j) Xc[(i)*ldx + (j)]
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    double* wd = ws + (size_t)draw * wo.total;
    int* meta = reinterpret_cast<int*>(wd + wo.meta);
    int eu0 = 0, eu1 = 0, eu2 = 0, have_T = 0;
    wave_sync();
    GW_STAMP(16);
    if (meta[GW_FLAG] != 0 || meta[GW_CONV] == 0) {
      eu0 = eu1 = -3;
    } else {
      const int N = meta[GW_N], ell = meta[GW_ELL], z = meta[GW_Z], ns2 = meta[GW_NS2];
      const int w = N - z, nu = w - ns2;
      const cx* HC = reinterpret_cast<const cx*>(wd + wo.HC);
      const cx* TC = reinterpret_cast<const cx*>(wd + wo.TC);
      const cx* XC = reinterpret_cast<const cx*>(wd + wo.XC);
      const double* R0 = wd + wo.R0;
      const double* X1 = wd + wo.X1;
      for (int idx = lane; idx < w * ell; idx += 64) {
        const int i = idx / ell, j = idx - i * ell;
        PX(i, j) = XC[(size_t)i * cp.lcap + j];
      }
      wave_sync();
      GW_STAMP(17);
      // coincident zeros (gensys.py:243-244): deflated roots have beta = 0
      const bool zz0 = (lane < z) && (fabs(R0[(size_t)lane * cp.zcap + lane]) < rs);
      const bool zz1 = (lane < w) && (cabs_(HC[(size_t)lane * cp.wcap + lane]) < rs) &&
                       (cabs_(TC[(size_t)lane * cp.wcap + lane]) < rs);
      if (__ballot(zz0 || zz1) != 0ull) {
        eu0 = eu1 = -2;
      } else {
        int r2 = 0, r1 = 0;
        if (nu > 0) {
          jacobi_svd_rr(&PX(ns2, 0), ldx, nu, ell, V2, ldx, s2, lane);
          for (int j = 0; j < ell; ++j) r2 += (s2[j] > rs) ? 1 : 0;
        } else {
          if (lane < ell) s2[lane] = 0.0;
          for (int idx = lane; idx < ell * ell; idx += 64) {
            const int i = idx / ell, j = idx - i * ell;
            V2[i * ldx + j] = mk(i == j ? 1.0 : 0.0, 0.0);
          }
          wave_sync();
        }
        if (r2 >= nu) eu0 = 1;
        GW_STAMP(18);
        // s1_j = || eta1 v_j ||, eta1 = [X1; X2[:ns2]] (CS decomposition argument of dsge_gensys.hpp)
        for (int j = 0; j < ell; ++j) {
          cx g = mk(0, 0), g0 = mk(0, 0);
          if (lane < ns2) g = gw_csum4(0, ell, [&](int cc) { return PX(lane, cc) * V2[cc * ldx + j]; });
          if (lane < z) g0 = gw_csum4(0, ell, [&](int cc) { return X1[(size_t)lane * cp.lcap + cc] * V2[cc * ldx + j]; });
          const double sq = wave_sum_dpp(fma(g.re, g.re, g.im * g.im) + fma(g0.re, g0.re, g0.im * g0.im));
          if (lane == 0) s1[j] = sqrt(sq);
        }
        wave_sync();
        int n_loose = 0;
        for (int j = 0; j < ell; ++j) {
          const bool k1 = s1[j] > rs, k2 = s2[j] > rs;
          r1 += k1 ? 1 : 0;
          n_loose += (k1 && !k2) ? 1 : 0;
        }
        bool unique = true;
        if (r1 > 0) {
          eu2 = n_loose;
          unique = (n_loose == 0);
        }
        if (unique) eu1 = 1;
        // Bm = V2 diag(w_j) G2^H (ell x nu), w_j = [s1_j > rs][s2_j > rs] / s2_j^2
        cx* BMg = reinterpret_cast<cx*>(wd + wo.BM);
        for (int idx = lane; idx < ell * nu; idx += 64) {
          const int cc = idx / nu, u = idx - cc * nu;
          cx acc = mk(0, 0);
          for (int j = 0; j < ell; ++j) {
            if (!(s1[j] > rs && s2[j] > rs)) continue;
            const double wj = 1.0 / (s2[j] * s2[j]);
            acc = acc + V2[cc * ldx + j] * (wj * conj(PX(ns2 + u, j)));
          }
          BMg[(size_t)cc * cp.wcap + u] = acc;
        }
        __threadfence_block();  // Bm is read back below by other lanes of this wavefront
        wave_sync();
        // Phi_b = X2[:ns2] Bm (ns2 x nu)
        cx* PHg = reinterpret_cast<cx*>(wd + wo.PHI);
        for (int idx = lane; idx < ns2 * nu; idx += 64) {
          const int i = idx / nu, u = idx - i * nu;
          PHg[(size_t)i * cp.wcap + u] = gw_csum4(0, ell, [&](int cc) { return PX(i, cc) * BMg[(size_t)cc * cp.wcap + u]; });
        }
        have_T = 1;
      }
    }
    if (lane == 0) {
      meta[GW_EU0] = eu0;
      meta[GW_EU1] = eu1;
      meta[GW_EU2] = eu2;
      meta[GW_HAVE_T] = have_T;
    }
    GW_STAMP(19);
  }
#undef PX
}

// ---- launch 4: T in the window basis (gensys.py:314-343) and the outputs ------------------------------------------------------
// Four wavefronts per draw (round 4): 72 KB of LDS allow two draws per CU, and at one wavefront each the launch was eight rounds
// of a 140 k-cycle chain; the products, loads and the output write are spread over 256 threads, the two back-substitutions
// (one column per lane) stay on the first wavefront.
constexpr int GW_POST_THREADS = 256;
__global__ __launch_bounds__(GW_POST_THREADS) void gensys_post_kernel(int batch, GwCaps cp, double tol, const double* __restrict__ ws,
                                                          double* __restrict__ T_out, int32_t* __restrict__ eu_out,
                                                          int32_t* __restrict__ status, long long* __restrict__ dbg,
                                                          int rescue, int32_t* __restrict__ key_out,
                                                          const int32_t* __restrict__ act) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  batch = gw_active_count(batch, act);
  const int lane = threadIdx.x;  // 0 .. GW_POST_THREADS-1
  const int n = cp.n;
  const int ldh = cp.wcap | 1, lds_ = cp.scap | 1;
  cx* Hc = reinterpret_cast<cx*>(smem);
  cx* Tc = Hc + (size_t)cp.wcap * ldh;
  cx* Mc = Tc + (size_t)cp.wcap * ldh;
  cx* Bm = Mc + (size_t)cp.wcap * ldh;   // lcap x ldh
  double* RR = reinterpret_cast<double*>(Bm + (size_t)cp.lcap * ldh);  // (wcap + lcap) x lds_: [Re(M1 Yb Ms^H); Re(Bm B22 Ms2^H)]
  double* E = RR + (size_t)(cp.wcap + cp.lcap) * lds_;                // zcap x lds_
  const int ldr = cp.zcap | 1, ldq = (cp.wcap + cp.lcap) | 1;
  double* R0s = E + (size_t)cp.zcap * lds_;                           // zcap x ldr   (R0)
  double* HXs = R0s + (size_t)cp.zcap * ldr;                          // zcap x ldq   ([H12 | X1])
  const GwOffsets wo = gw_offsets(cp);
#define PH(i, j) Hc[(i)*ldh + (j)]
#define PT(i, j) Tc[(i)*ldh + (j)]
#define PM(i, j) Mc[(i)*ldh + (j)]
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    const int gdraw = act ? act[1 + draw] : draw;  // (draw: the slot of the workspace; gdraw: the caller's draw)
    const size_t off = (size_t)gdraw * n * n;
    const double* wd = ws + (size_t)draw * wo.total;
    const int* meta = reinterpret_cast<const int*>(wd + wo.meta);
    const int eu0 = meta[GW_EU0], eu1 = meta[GW_EU1], eu2 = meta[GW_EU2];
    int st_extra = 0;
    if (meta[GW_FLAG] != 0)
      st_extra = DSGE_ST_GENSYS_TOO_BIG;
    else if (meta[GW_CONV] == 0)
      st_extra = DSGE_ST_GENSYS_QZ_FAIL;
    const bool have_T = meta[GW_HAVE_T] != 0;
    wave_sync();
    GW_STAMP(20);
    if (have_T) {
      const int N = meta[GW_N], ell = meta[GW_ELL], z = meta[GW_Z], ns2 = meta[GW_NS2];
      const int w = N - z, sp = n - z, nu = w - ns2;
      const unsigned long long a_colmask =
          (unsigned long long)(unsigned)meta[GW_MASK_LO] | ((unsigned long long)(unsigned)meta[GW_MASK_HI] << 32);
      const unsigned long long nmask = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
      const unsigned long long zmask = ~a_colmask & nmask;
      const cx* HC = reinterpret_cast<const cx*>(wd + wo.HC);
      const cx* TC = reinterpret_cast<const cx*>(wd + wo.TC);
      const cx* MC = reinterpret_cast<const cx*>(wd + wo.MC);
      const cx* BMg = reinterpret_cast<const cx*>(wd + wo.BM);
      const cx* PHg = reinterpret_cast<const cx*>(wd + wo.PHI);
      const double* R0 = wd + wo.R0;
      const double* H12 = wd + wo.H12;
      const double* T12 = wd + wo.T12;
      const double* X1 = wd + wo.X1;
      struct cx3 {
        cx h, t, m;
      };
      lane_loop_batched<4, GW_POST_THREADS>(
          w * w, lane,
          [&](int idx) {
            const int i = idx / w, j = idx - i * w;
            const size_t o = (size_t)i * cp.wcap + j;
            return cx3{HC[o], TC[o], MC[(size_t)j * cp.wcap + i]};  // M is stored transposed
          },
          [&](int idx, cx3 v) {
            const int i = idx / w, j = idx - i * w;
            PH(i, j) = v.h;
            PT(i, j) = v.t;
            PM(i, j) = v.m;
          });
      lane_loop_batched<4, GW_POST_THREADS>(
          ell * nu, lane,
          [&](int idx) {
            const int cc = idx / nu, u = idx - cc * nu;
            return BMg[(size_t)cc * cp.wcap + u];
          },
          [&](int idx, auto v) {
            const int cc = idx / nu, u = idx - cc * nu;
            Bm[cc * ldh + u] = v;
          });
      // the real tail's operands: R0 and [H12 | X1] (read once, coalesced; the products below broadcast them from LDS)
      lane_loop_batched<8, GW_POST_THREADS>(
          z * z, lane,
          [&](int idx) {
            const int i = idx / z, j = idx - i * z;
            return R0[(size_t)i * cp.zcap + j];
          },
          [&](int idx, double v) {
            const int i = idx / z, j = idx - i * z;
            R0s[i * ldr + j] = v;
          });
      lane_loop_batched<8, GW_POST_THREADS>(
          z * w, lane,
          [&](int idx) {
            const int i = idx / w, j = idx - i * w;
            return H12[(size_t)i * cp.wcap + j];
          },
          [&](int idx, double v) {
            const int i = idx / w, j = idx - i * w;
            HXs[i * ldq + j] = v;
          });
      lane_loop_batched<8, GW_POST_THREADS>(
          z * ell, lane,
          [&](int idx) {
            const int i = idx / ell, j = idx - i * ell;
            return X1[(size_t)i * cp.lcap + j];
          },
          [&](int idx, double v) {
            const int i = idx / ell, j = idx - i * ell;
            HXs[i * ldq + w + j] = v;
          });
      wave_sync();
      // Phi_b (ns2 x nu) into the free lower-left block of H: Phi_b[i][u] at H[ns2 + u][i]
      for (int idx = lane; idx < ns2 * nu; idx += GW_POST_THREADS) {
        const int i = idx / nu, u = idx - i * nu;
        PH(ns2 + u, i) = PHg[(size_t)i * cp.wcap + u];
      }
      wave_sync();
      GW_STAMP(21);
      // rhs = [B11, B12 - Phi_b B22] in place in T[:ns2, :]
      for (int idx = lane; idx < ns2 * nu; idx += GW_POST_THREADS) {
        const int i = idx / nu, cc = idx - i * nu;
        PT(i, ns2 + cc) = PT(i, ns2 + cc) - gw_csum4(0, cc + 1, [&](int u) { return PH(ns2 + u, i) * PT(ns2 + u, ns2 + cc); });
      }
      wave_sync();
      GW_STAMP(22);
      // Yb = A11w^-1 rhs by back-substitution, one column per lane
      if (lane < w) {
        for (int i = ns2 - 1; i >= 0; --i) {
          const cx acc = PT(i, lane) - gw_csum4(i + 1, ns2, [&](int k2) { return PH(i, k2) * PT(k2, lane); });
          PT(i, lane) = cdiv(acc, PH(i, i));
        }
      }
      wave_sync();
      GW_STAMP(23);
      // Wb = Yb Ms^H (ns2 x s') into H[:ns2, :s'];  BB = B22 Ms2^H (nu x s') into H[ns2:, :s']
      for (int idx = lane; idx < ns2 * sp; idx += GW_POST_THREADS) {
        const int i = idx / sp, cc = idx - i * sp;
        PH(i, cc) = gw_csum4(0, w, [&](int k2) { return PT(i, k2) * conj(PM(cc, k2)); });
      }
      for (int idx = lane; idx < nu * sp; idx += GW_POST_THREADS) {
        const int u = idx / sp, cc = idx - u * sp;
        PH(ns2 + u, cc) = gw_csum4(u, nu, [&](int v) { return PT(ns2 + u, ns2 + v) * conj(PM(cc, ns2 + v)); });
      }
      wave_sync();
      // RR = [Re(M[:, :ns2] Wb); Re(Bm BB)]  ((w + ell) x s', real)
      for (int idx = lane; idx < w * sp; idx += GW_POST_THREADS) {
        const int r = idx / sp, cc = idx - r * sp;
        RR[r * lds_ + cc] = gw_sum4(0, ns2, [&](int i) {
          const cx a = PM(r, i), b = PH(i, cc);
          return fma(a.re, b.re, -(a.im * b.im));
        });
      }
      for (int idx = lane; idx < ell * sp; idx += GW_POST_THREADS) {
        const int a0 = idx / sp, cc = idx - a0 * sp;
        RR[(w + a0) * lds_ + cc] = gw_sum4(0, nu, [&](int u) {
          const cx a = Bm[a0 * ldh + u], b = PH(ns2 + u, cc);
          return fma(a.re, b.re, -(a.im * b.im));
        });
      }
      wave_sync();
      GW_STAMP(24);
      // non-state rows: E = T12[:, :s'] - [H12 | X1] RR, then R0^-1 E by back-substitution (one column per lane)
      for (int idx = lane; idx < z * sp; idx += GW_POST_THREADS) {
        const int p = idx / sp, cc = idx - p * sp;
        E[p * lds_ + cc] = T12[(size_t)p * cp.scap + cc] - gw_sum4(0, w + ell, [&](int k2) { return HXs[p * ldq + k2] * RR[k2 * lds_ + cc]; });
      }
      wave_sync();
      if (lane < sp) {
        for (int p = z - 1; p >= 0; --p) {
          const double acc = E[p * lds_ + lane] - gw_sum4(p + 1, z, [&](int q) { return R0s[p * ldr + q] * E[q * lds_ + lane]; });
          E[p * lds_ + lane] = acc / R0s[p * ldr + p];
        }
      }
      wave_sync();
      GW_STAMP(25);
      // T in the caller's variable order; columns of non-state variables are exact zeros (see dsge_gensys.hpp)
      for (int idx = lane; idx < n * n; idx += GW_POST_THREADS) {
        const int v = idx / n, c = idx - v * n;
        const unsigned long long bc_ = 1ull << c, bv = 1ull << v;
        double val = 0.0;
        if (!(zmask & bc_)) {
          const int jc = c - __popcll(zmask & (bc_ - 1ull));  // index among the state variables
          if (zmask & bv)
            val = E[__popcll(zmask & (bv - 1ull)) * lds_ + jc];
          else
            val = RR[(v - __popcll(zmask & (bv - 1ull))) * lds_ + jc];
        }
        T_out[off + idx] = val;
      }
      GW_STAMP(26);
    } else {
      for (int idx = lane; idx < n * n; idx += GW_POST_THREADS) T_out[off + idx] = 0.0;
    }
    if (key_out && lane < 64) {
      // Dispatch key of the Kalman launch (persistence_key_kernel's scale: 8 * -log2(1 - rho), 0..63) from the spectrum the QZ
      // iteration has just computed: rho = the largest modulus |beta / alpha| among the stable roots of the window (the deflated
      // roots are zero) -- the spectral radius of T itself, where the power iteration only estimates it
      double kq = 0.0;
      if (have_T) {
        const int w_ = meta[GW_N] - meta[GW_Z], ns2_ = meta[GW_NS2];
        const cx* HCd = reinterpret_cast<const cx*>(wd + wo.HC);
        const cx* TCd = reinterpret_cast<const cx*>(wd + wo.TC);
        double mod = 0.0;
        if (lane < ns2_ && lane < w_) mod = cabs_(TCd[(size_t)lane * cp.wcap + lane]) / cabs_(HCd[(size_t)lane * cp.wcap + lane]);
        if (!(mod == mod)) mod = 0.0;
        const double rho = wave_nanmax(mod);
        kq = (rho < 1.0) ? -8.0 * log2(1.0 - rho) : 63.0;
        if (!(kq == kq)) kq = 0.0;
        kq = kq < 0.0 ? 0.0 : (kq > 63.0 ? 63.0 : kq);
      }
      if (lane == 0) key_out[gdraw] = (int32_t)kq;
    }
    if (lane == 0) {
      eu_out[3 * gdraw] = eu0;
      eu_out[3 * gdraw + 1] = eu1;
      eu_out[3 * gdraw + 2] = eu2;
      // rescue: a rescue pass follows (the capacity record came from a cache): it takes the draws that did not fit
      status[gdraw] = (eu0 == 1 && eu1 == 1) ? DSGE_ST_OK
                     : ((rescue && st_extra == DSGE_ST_GENSYS_TOO_BIG) ? DSGE_ST_INTERNAL_RERUN : (DSGE_ST_NOT_CONVERGED | st_extra));
    }
  }
#undef GW_STAMP
#undef PH
#undef PT
#undef PM
}


// after the rescue pass of the cached-capacity route: a draw that is still flagged fits neither the cached window launches nor
// the single-launch kernel
__global__ __launch_bounds__(256) void gensys_rescue_close_kernel(int batch, int32_t* __restrict__ status,
                                                                   int32_t* __restrict__ eu_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < batch && (status[i] & DSGE_ST_INTERNAL_RERUN)) {
    status[i] = DSGE_ST_NOT_CONVERGED | DSGE_ST_GENSYS_TOO_BIG;
    eu_out[3 * i] = eu_out[3 * i + 1] = -3;
    eu_out[3 * i + 2] = 0;
  }
}

// ---- Blanchard-Kahn eigenvalues (compute_bk_eigenvalues, gEconpy/model/perturbation.py:412-445): after the reduce and QZ
// launches the generalized eigenvalues are on the diagonals -- (R0_ii, 0) for the deflated roots, (H_ii, T_ii) for the
// window.  The reference divides LAPACK's (alpha, beta) with beta real and non-negative (zgges normalisation):
// lambda = beta / (alpha + tol); the pair is rotated to that convention here before tol is added.  Output per draw:
// N = n + #lead eigenvalues sorted by ascending modulus (stride 2n), n_forward = #lead, n_unstable = #{|lambda| > 1}.
__global__ __launch_bounds__(64) void gensys_bk_kernel(int batch, GwCaps cp, double tol, const double* __restrict__ ws,
                                                        double* __restrict__ eig_re, double* __restrict__ eig_im,
                                                        int32_t* __restrict__ n_eig, int32_t* __restrict__ n_forward,
                                                        int32_t* __restrict__ n_unstable, int32_t* __restrict__ status) {
  const int lane = threadIdx.x;
  const int n = cp.n;
  const GwOffsets wo = gw_offsets(cp);
  for (int draw = blockIdx.x; draw < batch; draw += gridDim.x) {
    const double* wd = ws + (size_t)draw * wo.total;
    const int* meta = reinterpret_cast<const int*>(wd + wo.meta);
    const size_t o = (size_t)draw * 2 * n;
    const bool ok = meta[GW_FLAG] == 0 && meta[GW_CONV] != 0;
    const int N = meta[GW_N], ell = meta[GW_ELL], z = meta[GW_Z];
    for (int i = lane; i < 2 * n; i += 64) {
      eig_re[o + i] = 0.0;
      eig_im[o + i] = 0.0;
    }
    if (!ok) {
      if (lane == 0) {
        n_eig[draw] = 0;
        n_forward[draw] = ell;
        n_unstable[draw] = 0;
        status[draw] = DSGE_ST_NOT_CONVERGED | (meta[GW_FLAG] ? DSGE_ST_GENSYS_TOO_BIG : DSGE_ST_GENSYS_QZ_FAIL);
      }
      continue;
    }
    cx al = mk(1.0, 0.0), be = mk(0.0, 0.0);
    if (lane < z) {
      al = mk(wd[wo.R0 + (size_t)lane * cp.zcap + lane], 0.0);
    } else if (lane < N) {
      const int i = lane - z;
      al = reinterpret_cast<const cx*>(wd + wo.HC)[(size_t)i * cp.wcap + i];
      be = reinterpret_cast<const cx*>(wd + wo.TC)[(size_t)i * cp.wcap + i];
    }
    const double ab = cabs_(be);
    if (ab > 0.0) {  // rotate the pair so that beta is real and non-negative
      const cx ph = (1.0 / ab) * conj(be);
      al = al * ph;
      be = mk(ab, 0.0);
    }
    // compute_bk_eigenvalues decomposes (-G0, G1) = (Gamma_0, Gamma_1) (perturbation.py:436-437): alpha changes sign
    const cx lam = cdiv(be, mk(tol - al.re, -al.im));
    const double mod = (lane < N) ? cabs_(lam) : 1e308;
    // rank by (modulus, original position): N <= 64, one eigenvalue per lane
    int rank = 0;
    for (int j = 0; j < N; ++j) {
      const double mj = readlane_dyn_f64(mod, j);
      rank += (mj < mod || (mj == mod && j < lane)) ? 1 : 0;
    }
    if (lane < N) {
      eig_re[o + rank] = lam.re;
      eig_im[o + rank] = lam.im;
    }
    const int nun = __popcll(__ballot(lane < N && mod > 1.0));
    if (lane == 0) {
      n_eig[draw] = N;
      n_forward[draw] = ell;
      n_unstable[draw] = nun;
      status[draw] = DSGE_ST_OK;
    }
  }
}

}  // namespace dsge

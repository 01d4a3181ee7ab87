// gensys window path, real double-shift QZ sweeps with TWO draws per wavefront (round 4).
//
// Why: gw_realqz_sweeps (dsge_gensys_win.hpp) is VALU-issue bound -- profiles/r4/batch_scaling.txt: 1024 draws (one wavefront per
// SIMD) take 0.885 ms, 2048 draws (two per SIMD) 1.47 ms, i.e. a second wavefront buys 21 % -- and it issues every instruction
// for 64 lanes while the 30 x 30 window of the SW-shaped pencil fills 30 (T, H rows) to 42 ([H | X] columns) of them.  Here a
// wavefront chases the bulges of two draws at once: lanes 0..31 own draw 2p, lanes 32..63 draw 2p + 1 (window <= 32, #lead <= 32;
// the launcher keeps the one-draw kernel for anything larger).  A sweep step then costs SIX 32-lane passes per PAIR (H, T, X
// columns from the left; H, T, M rows from the right) instead of five 64-lane passes per draw, and the reflector generation,
// the shift arithmetic and the loop control are shared by two draws.
//
// The two draws deflate at different times, so every half carries its own sweep state (ifirst, ilast, k, iteration count) in
// lane-uniform-per-half registers and the loop body is a two-stage state machine:
//   stage A (wave-uniform branch, entered when a half is between sweeps): zhgeqz-style deflation tests of that half, the
//           double-shift vector of a new sweep;
//   stage B: one bulge-chase step at each half's own k.  The last step of a sweep (two rows / two columns) is the general step
//           with the third row and column switched off (z = 0 and b = e3 make v2 = 0 in both reflectors).
// What a one-draw step hands from lane to lane through v_readlane (wave-uniform lane index) goes through LDS here: the six
// entries of T that define the right reflector and the three entries of H that start the next step are read back after the
// stores of the pass that produced them, in the same round trip as the operands of the next pass.
//
// LDS: H and T of a draw share ONE array of w x ((w + 6) | 1) doubles -- H(i, j) at [i][j + 6] (band i <= j + 3: Hessenberg +
// bulge), T(i, j) transposed at [j][i] (band i <= j + 2) -- 8.9 KB at w = 30; X (rows of Q' Pi) and the accumulated right
// transformation M stay in the draw's workspace (L2), one row / column carried in registers and the next prefetched a step
// ahead, as gw_realqz_sweeps does for M.  17.8 KB per wavefront: nine wavefronts = 18 draws per CU, 4096 draws resident at once.
// The arithmetic of a step is that of gw_realqz_sweeps (same reflectors, same shifts, same deflation rules); the pairing changes
// which lane computes what, not what is computed.
#pragma once
#include "dsge_gensys_win.hpp"

namespace dsge {

// LD: compile-time row stride of the packed array (odd, >= wcap + 7: six columns of offset for H, one zero pad column); the
// array has wcap + 2 rows (zero pad rows).  The pad row / column make the third row and column of the LAST step of a sweep at
// the window's edge readable (zeros, written back unchanged because v2 = 0 there) without clamped addresses.
__host__ __device__ inline int gp_ld(const GwCaps& c) { return c.wcap <= 30 ? 37 : 39; }
__host__ __device__ inline size_t gp_smem(const GwCaps& c) { return (size_t)2 * (c.wcap + 2) * gp_ld(c) * 8; }
__host__ __device__ inline bool gp_fits(const GwCaps& c) { return c.wcap <= 32 && c.lcap <= 32 && c.wcap >= 3; }

__device__ __forceinline__ double half_sum(double v) {  // sum over the 32 lanes of a half
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) v += shfl_xor_f64(v, m);
  return v;
}

// LDS loads issued by hand: the result register must not be read before an LDS_WAIT that names it.  (Mixing them with the
// compiler's own ds instructions is safe: LDS operations of a wavefront complete in order, so a counted wait of the compiler can
// only become stricter through the extra operations in flight.)  "memory": the compiler's LDS stores stay in front / behind.
__device__ __forceinline__ unsigned lds_addr(const double* p) { return (unsigned)(size_t)p; }
template <int OFF>
__device__ __forceinline__ double lds_ld(unsigned addr) {
  double v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
#define LDS_WAIT6(n, a, b, c, d, e, f) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f))
#define LDS_WAIT3(n, a, b, c) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a), "+v"(b), "+v"(c))

// gw_house3 with ONE select: x^2 + y^2 + z^2 + 1e-290 never vanishes, so the reciprocals are always taken of positive numbers;
// y = z = 0 then gives v1 = v2 = 0 and tau = 2 -- the reflector diag(-1, 1, 1), as orthogonal as the identity gw_house3 returns
// there -- and only a vector below 1e-140 in norm (tau = 0: identity) needs the select.
__device__ __forceinline__ GwHouse gp_house3(double x, double y, double z) {
  GwHouse h;
  const double s2 = fma(x, x, fma(y, y, fma(z, z, 1e-290)));
  double rn = __builtin_amdgcn_rsq(s2);
  const double hs = -0.5 * s2;
  rn = rn * fma(hs * rn, rn, 1.5);
  rn = rn * fma(hs * rn, rn, 1.5);
  const double nrm = s2 * rn;
  const double d = fabs(x) + nrm;               // |x - beta|
  const double inv = copysign(fast_rcp(d), x);  // 1 / (x - beta)
  h.beta = -copysign(nrm, x);
  h.v1 = y * inv;
  h.v2 = z * inv;
  h.tau = (s2 > 1e-280) ? d * rn : 0.0;
  return h;
}

// Instruction budget (the kernel is VALU-issue bound: time = slots x VALU instructions per slot x 4 cycles x wavefronts per SIMD):
//   * LD is a template constant, so every row / column offset inside a pass is an immediate of the ds instruction;
//   * T(k, c) at [c][k] and H(c, k) at [c][k + 6] are six doubles apart, and so are T(c, k) / H(k, c): TWO per-lane pointers
//     (pA = row c of the array + k, pB = row k + c) address all four 32-lane passes of a step, a third (pk, the diagonal) the six
//     entries of T that define the right reflector and the three of H that start the next step;
//   * lanes outside a pass's band are switched off by the exec mask (one compare) instead of walking clamped duplicates;
//   * the exact zeros / beta a reflector leaves behind are stored by the one lane that owns them, after the pass;
//   * M and X are addressed by 32-bit offsets from the (wave-uniform) workspace base.
template <int LD>
__global__ __launch_bounds__(64) void gensys_sweeps_pair_kernel(int batch, GwCaps cp, double* __restrict__ ws,
                                                                 long long* __restrict__ dbg, const int32_t* __restrict__ act) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  batch = gw_active_count(batch, act);
  constexpr double ULPD = 2.220446049250313e-16, SAFMIN = 2.2250738585072014e-308;
  const int lane = threadIdx.x, h = lane >> 5, l = lane & 31;
  const int rows = cp.wcap + 2;  // two zero pad rows: row k+2 of a last step and the (unused) row k+3 read at the window's edge
  double* P = smem + (size_t)h * rows * LD;
#define PH(i, j) P[(i)*LD + (j) + 6]
#define PT(i, j) P[(j)*LD + (i)]
  const GwOffsets wo = gw_offsets(cp);
  const int npairs = (batch + 1) >> 1;
  const unsigned mcolB = 8u * (unsigned)cp.wcap, xrowB = 8u * (unsigned)cp.lcap;  // bytes: one column of the REAL M (MRE), one row of X
  const char* wsb = reinterpret_cast<const char*>(ws);
  char* wsw = reinterpret_cast<char*>(ws);
#define GLD(off) (*reinterpret_cast<const double*>(wsb + (size_t)(unsigned)(off)))
#define GST(off, v) (*reinterpret_cast<double*>(wsw + (size_t)(unsigned)(off)) = (v))
  for (int pair = blockIdx.x; pair < npairs; pair += gridDim.x) {
    const int draw = 2 * pair + h;
    const bool exists = draw < batch;
    const size_t wdo = (size_t)(exists ? draw : 2 * pair) * wo.total;  // (odd batch: the idle half only ever reads its neighbour)
    double* wd = ws + wdo;
    const int* meta = reinterpret_cast<const int*>(wd + wo.meta);
    const bool valid = exists && meta[GW_FLAG] == 0;
    const int w = valid ? meta[GW_N] - meta[GW_Z] : 0, ell = valid ? meta[GW_ELL] : 0;
    if (dbg && pair == 0 && lane == 0) dbg[3] = (long long)clock64();
    wave_sync();
    for (int idx = l; idx < rows * LD; idx += 32) P[idx] = 0.0;
    wave_sync();
    {  // the Hessenberg-triangular window: four (H, T) pairs in flight per trip
      const int total = w * w;
      const int wdiv = w > 0 ? w : 1;
      for (int base = 0; base < cp.wcap * cp.wcap; base += 4 * 32) {
        double hv[4], tv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = base + u * 32 + l;
          const int ic = idx < total ? idx : 0;
          const int i = ic / wdiv, j = ic - i * wdiv;
          const size_t o = (size_t)i * cp.wcap + j;
          hv[u] = wd[wo.HR + o];
          tv[u] = wd[wo.TR + o];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = base + u * 32 + l;
          if (idx < total) {
            const int i = idx / wdiv, j = idx - i * wdiv;
            if (i <= j + 1) PH(i, j) = hv[u];
            if (i <= j) PT(i, j) = tv[u];
          }
        }
      }
    }
    wave_sync();
    const int cw = min(l, max(w - 1, 0));     // this lane's column of H / T and row of H / T / M (lanes >= w duplicate the last)
    const int cx_ = min(l, max(ell - 1, 0));  // this lane's column of X
    const unsigned offM0 = (unsigned)((wdo + wo.MRE + (size_t)cw) * 8);  // M(cw, col) at offM0 + col * mcolB
    const unsigned offX0 = (unsigned)((wdo + wo.XR + (size_t)cx_) * 8);     // X(row, cx_) at offX0 + row * xrowB
    const unsigned offMmax = offM0 + (unsigned)max(w - 1, 0) * mcolB, offXmax = offX0 + (unsigned)max(w - 1, 0) * xrowB;
    double* const rowc = P + cw * LD;  // array row cw: T(., cw) at rowc[.], H(cw, .) at rowc[. + 6]
    double btol;
    {
      double ss = 0.0;
      if (l < w)
        for (int i = 0; i <= l; ++i) {
          const double t = PT(i, l);
          ss = fma(t, t, ss);
        }
      btol = fmax(SAFMIN, ULPD * sqrt(half_sum(ss)));
    }
    // per-half sweep state (lane-uniform inside a half): k >= 0 chasing at k, -1 between sweeps, -2 finished
    int ilast = w - 1, ifirst = 0, it = 0, guard = 0;
    int k = (valid && w >= 3) ? -1 : -2;
    const int max_total = 40 * w;
    double x = 0.0, y = 0.0, z = 0.0, m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0, xr0 = 0.0, xr1 = 0.0, xr2 = 0.0, x3 = 0.0;
    double *pA = rowc, *pB = P + cw, *pk = P;  // pA = rowc + k, pB = P + k * LD + cw, pk = P + k * (LD + 1)
    unsigned offM = offM0, offX = offX0;       // column k of M, row k of X
    int steps = 0, sweeps = 0;
    long long slots = 0, a_entries = 0, a_cycles = 0;  // debug counters (pair 0)
    while (true) {
      // ---------------- stage A: halves between sweeps -- deflation tests, shifts (repeated until none is left between sweeps)
      while (true) {
        if (k == -1 && guard >= max_total) k = -2;
        if (__ballot(k == -1) == 0ull) break;
        const long long ta0 = dbg ? (long long)clock64() : 0;
        ++a_entries;
        wave_sync();
        const bool need = k == -1;
        double hjj = 0.0, hmm = 0.0, hsub = 0.0, tjj = 1.0;
        if (need && l < w) {
          hjj = PH(l, l);
          tjj = PT(l, l);
          if (l > 0) {
            hsub = PH(l, l - 1);
            hmm = PH(l - 1, l - 1);
          }
        }
        const bool sm = need && l < w && l > 0 && fabs(hsub) <= fmax(SAFMIN, ULPD * (fabs(hjj) + fabs(hmm)));
        const unsigned small = (unsigned)(__ballot(sm) >> (32 * h));
        const unsigned tzero = (unsigned)(__ballot(need && l < w && fabs(tjj) <= btol) >> (32 * h));
        bool start = false;
        if (need) {
          ++guard;
          if ((small >> ilast) & 1u) {
            if (l == 0) PH(ilast, ilast - 1) = 0.0;
            ilast -= 1;
            it = 0;
          } else if ((small >> (ilast - 1)) & 1u) {
            if (l == 0) PH(ilast - 1, ilast - 2) = 0.0;
            ilast -= 2;  // a 2 x 2 block: the complex iteration splits it
            it = 0;
          } else {
            ifirst = 0;
            const unsigned below = small & ((1u << (ilast - 1)) - 1u);  // bits 1 .. ilast-2
            if (below) {
              ifirst = 31 - __clz((int)below);
              if (l == 0) PH(ifirst, ifirst - 1) = 0.0;
            }
            const unsigned act = ((ilast >= 31) ? ~0u : ((1u << (ilast + 1)) - 1u)) & ~((1u << ifirst) - 1u);
            if (tzero & act)
              k = -2;  // an infinite root inside the active block: zhgeqz's zero chasing lives in the next launch
            else if (++it > 30)
              k = -2;
            else
              start = true;
          }
          if (ilast < 2) k = -2;
        }
        if (__ballot(start) != 0ull) {
          wave_sync();
          if (start) {
            // the first column of (M - s1)(M - s2), M = H T^-1 on the active block, shifts = roots of the trailing 2 x 2 pencil.
            // Reciprocals from v_rcp_f64 + two Newton steps: the shifts steer the convergence, they do not enter the accuracy.
            const int m = ilast;
            const double p_ = PH(m - 1, m - 1), q_ = PH(m - 1, m), r_ = PH(m, m - 1), s_ = PH(m, m);
            const double e_ = PT(m - 1, m - 1), f_ = PT(m - 1, m), g_ = PT(m, m);
            const double ie = fast_rcp(e_), ig = fast_rcp(g_);
            double tr, det;
            if (it % 10 == 0) {  // exceptional shifts
              const double w_ = 1.5 * (fabs(r_ * ie) + fabs(PH(m - 1, m - 2) / PT(m - 2, m - 2)));
              tr = w_;
              det = w_ * w_;
            } else {
              tr = p_ * ie + (s_ - r_ * f_ * ie) * ig;
              det = (p_ * s_ - q_ * r_) * (ie * ig);
            }
            const int kf = ifirst;
            double* pf = P + kf * (LD + 1);
            const double a11 = pf[6], a12 = pf[7], a21 = pf[LD + 6], a22 = pf[LD + 7], a32 = pf[2 * LD + 7];
            const double b11 = pf[0], b12 = pf[LD], b22 = pf[LD + 1];
            const double i11 = fast_rcp(b11), i22 = fast_rcp(b22);
            const double m11 = a11 * i11, m21 = a21 * i11;
            const double y2 = m21 * i22;
            const double y1 = (m11 - b12 * y2) * i11;
            x = a11 * y1 + a12 * y2 - tr * m11 + det;
            y = a21 * y1 + a22 * y2 - tr * m21;
            z = a32 * y2;
            if (!(fabs(x) + fabs(y) + fabs(z) < 1e300)) {
              k = -2;  // NaN / overflow in the shift arithmetic: leave it to zhgeqz's logic
            } else {
              k = kf;
              ++sweeps;
              pk = pf;
              pA = rowc + kf;
              pB = P + kf * LD + cw;
              offM = offM0 + (unsigned)kf * mcolB;
              offX = offX0 + (unsigned)kf * xrowB;
              m0 = GLD(offM);
              m1 = GLD(offM + mcolB);
              m2 = GLD(offM + 2 * mcolB);
              xr0 = GLD(offX);
              xr1 = GLD(offX + xrowB);
              xr2 = GLD(offX + 2 * xrowB);
              m3 = GLD(min(offM + 3 * mcolB, offMmax));
              x3 = GLD(min(offX + 3 * xrowB, offXmax));
            }
          }
        }
        if (dbg) a_cycles += (long long)clock64() - ta0;
      }
      // ---------------- stage B: chase steps, one per half and trip, until a half finishes its sweep
      if (__ballot(k >= 0) == 0ull) break;  // (no half is between sweeps after stage A: every half has finished)
      do {
        // Straight-line step: no branch around the arithmetic.  A half that is between sweeps (or finished) runs the same
        // instructions on whatever its frozen pointers address and stores nothing (its band masks are empty: kl, kr), so no
        // value is defined under a divergent branch -- the compiler turned every such definition into a copy at the join, with
        // a wait for the prefetched column in front of it.  The workgroup is one wavefront, whose LDS instructions execute in
        // program order (what lane A stores, a later load of lane B sees): no barrier inside the step.
        const bool act = k >= 0;
        const int ai = act ? 1 : 0;
        const int kl = act ? k : 4096, kr = act ? k : -4096;
        const bool last = k == ilast - 1;
        const double nl = last ? 0.0 : 1.0;  // switches the third row / column off in the last step of a sweep
        // ---- left: rows k .. k+2 of [H | T | X], one column per lane.  The LDS loads of a step are issued by hand (lds_ld) in
        // front of the arithmetic that does not need them and waited for where it does (LDS_WAIT): left to the compiler they sink
        // into the exec-masked store regions, behind the 150-cycle chain of gw_house3.
        const unsigned aA = lds_addr(pA), aB = lds_addr(pB), aK = lds_addr(pk);
        double h0 = lds_ld<6 * 8>(aB), h1 = lds_ld<(LD + 6) * 8>(aB), h2 = lds_ld<(2 * LD + 6) * 8>(aB);
        double t0 = lds_ld<0>(aA), t1 = lds_ld<8>(aA), t2 = lds_ld<16>(aA);
        const GwHouse q = gp_house3(x, y, z * nl);
        LDS_WAIT6(0, h0, h1, h2, t0, t1, t2);
        {
          const double sh = q.tau * fma(q.v2, h2, fma(q.v1, h1, h0));
          h0 -= sh;
          h1 = fma(-sh, q.v1, h1);
          h2 = fma(-sh, q.v2, h2);
          const double st = q.tau * fma(q.v2, t2, fma(q.v1, t1, t0));
          t0 -= st;
          t1 = fma(-st, q.v1, t1);
          t2 = fma(-st, q.v2, t2);
          const double sx = q.tau * fma(q.v2, xr2, fma(q.v1, xr1, xr0));
          xr0 -= sx;
          xr1 = fma(-sx, q.v1, xr1);
          xr2 = fma(-sx, q.v2, xr2);
        }
        if (cw >= kl) {  // columns k .. of H and T (the others hold exact zeros in these rows)
          pB[2 * LD + 6] = h2;
          pB[LD + 6] = h1;
          pB[6] = h0;
          pA[2] = t2;
          pA[1] = t1;
          pA[0] = t0;
        }
        if (cw == kl - 1 && kl > ifirst) {  // column k-1 of H: the reflector's target
          pB[6] = q.beta;
          pB[LD + 6] = 0.0;
          pB[2 * LD + 6] = 0.0;
        }
        // ---- right: columns k .. k+2, one row of H, T, M per lane.  ONE reflector: its first column is the null vector of rows
        // k+1, k+2 of T (their cross product); last step: rows k+1 and e3, i.e. the 2-column reflector of row k+1
        double a0 = lds_ld<8>(aK), a1 = lds_ld<(LD + 1) * 8>(aK), a2 = lds_ld<(2 * LD + 1) * 8>(aK);
        double b0 = lds_ld<16>(aK), b1 = lds_ld<(LD + 2) * 8>(aK), b2 = lds_ld<(2 * LD + 2) * 8>(aK);
        double r0 = lds_ld<6 * 8>(aA), r1 = lds_ld<7 * 8>(aA), r2 = lds_ld<8 * 8>(aA);
        double u0 = lds_ld<0>(aB), u1 = lds_ld<LD * 8>(aB), u2 = lds_ld<2 * LD * 8>(aB);
        LDS_WAIT6(6, a0, a1, a2, b0, b1, b2);
        b0 *= nl;
        b1 *= nl;
        b2 = fma(b2, nl, 1.0 - nl);
        const double w0 = fma(a1, b2, -(a2 * b1)), w1 = fma(a2, b0, -(a0 * b2)), w2 = fma(a0, b1, -(a1 * b0));
        const GwHouse g1 = gp_house3(w0, w1, w2);
        LDS_WAIT6(0, r0, r1, r2, u0, u1, u2);
        {
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
        }
        if (cw <= kr + 3) {  // rows .. k+3 of H (Hessenberg + bulge)
          pA[8] = r2;
          pA[7] = r1;
          pA[6] = r0;
        }
        if (cw <= kr + 2) {  // rows .. k+2 of T
          pB[2 * LD] = u2;
          pB[LD] = u1;
          pB[0] = (cw > kr) ? 0.0 : u0;  // T(k+1, k), T(k+2, k): the reflector's targets
        }
        // the next step's vector (dead after the last step of the sweep: read anyway), requested now and waited for at the bottom
        double xn = lds_ld<(LD + 6) * 8>(aK), yn = lds_ld<(2 * LD + 6) * 8>(aK), z3 = lds_ld<(3 * LD + 6) * 8>(aK);
        // ---- M and X.  Order matters for the memory counter (vmcnt counts loads and stores together and a mix of both can
        // only be waited for as a whole): FIRST take over the column / row prefetched one step ago -- the only thing outstanding
        // besides it are the previous step's stores, a full step old --, THEN store, THEN prefetch for the next step.
        double m2n, x2n;  // (moves pinned here: left to itself the compiler copies m3 at the TOP of the step, i.e. waits for the
                          // prefetch a hundred cycles after issuing it)
        asm volatile("v_mov_b64 %0, %2\n\tv_mov_b64 %1, %3" : "=&v"(m2n), "=&v"(x2n) : "v"(m3), "v"(x3));
        __builtin_amdgcn_sched_barrier(0);
        if (act) {
          // column k of M and row k of X are final for this sweep, column / row k+1 after the LAST step only (with #lead = 0 the
          // X stores go to the unused first column of the draw's X block)
          GST(offM, m0);
          GST(offX, xr0);
          if (last) {
            GST(offM + mcolB, m1);
            GST(offX + xrowB, xr1);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        m3 = GLD(min(offM + 4 * mcolB, offMmax));  // (clamped at the window's edge: only ever multiplied by v2 = 0 there)
        x3 = GLD(min(offX + 4 * xrowB, offXmax));
        m0 = m1;
        m1 = m2;
        m2 = m2n;
        xr0 = xr1;
        xr1 = xr2;
        xr2 = x2n;
        const int kz = k;
        k = (act && last) ? -1 : k + ai;
        {  // the step's pointers follow k (a half between sweeps keeps valid ones: k clamped at 0)
          const int kc = max(k, 0);
          pk = P + kc * (LD + 1);
          pA = rowc + kc;
          pB = P + kc * LD + cw;
          offM = offM0 + (unsigned)kc * mcolB;
          offX = offX0 + (unsigned)kc * xrowB;
        }
        steps += ai;
        ++slots;
        LDS_WAIT3(0, xn, yn, z3);
        x = xn;
        y = yn;
        z = (kz + 3 <= ilast) ? z3 : 0.0;
      } while (__ballot(k == -1) == 0ull);
    }
    wave_sync();
    if (dbg && pair == 0 && lane == 0) {
      dbg[6] = (long long)clock64();
      dbg[27] = steps;
      dbg[28] = sweeps;
      dbg[31] = a_cycles;
    }
    if (dbg && lane == 0) {  // debug: the sum and the maximum of the pairs' step counts (the launch ends with the slowest pair)
      atomicAdd(reinterpret_cast<unsigned long long*>(dbg + 29), (unsigned long long)slots);
      atomicMax(reinterpret_cast<unsigned long long*>(dbg + 30), (unsigned long long)slots);
    }
    __threadfence_block();  // (the M stores of the sweeps are read back below by other lanes of this wavefront)
    {  // the window goes back as it came: full w x w, exact zeros outside the bands; M becomes complex for the next launch
      const int total = w * w;
      const int wdiv = w > 0 ? w : 1;
      cx* MCx = reinterpret_cast<cx*>(wd + wo.MC);
      const double* MRx = wd + wo.MRE;
      for (int idx = l; idx < cp.wcap * cp.wcap; idx += 32) {
        if (idx < total) {
          const int i = idx / wdiv, j = idx - i * wdiv;
          const size_t o = (size_t)i * cp.wcap + j;
          wd[wo.HR + o] = (i <= j + 3) ? PH(i, j) : 0.0;
          wd[wo.TR + o] = (i <= j + 2) ? PT(i, j) : 0.0;
          MCx[o] = mk(MRx[o], 0.0);  // (same transposed indexing in both: [col * wcap + row])
        }
      }
    }
    if (dbg && pair == 0 && lane == 0) dbg[4] = (long long)clock64();
  }
#undef PH
#undef PT
#undef GLD
#undef GST
}

// ---- Hessenberg-triangular reduction with two draws per wavefront (round 4) ----------------------------------------------------
// The launch in front of the sweeps (gensys_hesstri_kernel: T22 -> upper triangular by reflectors, H22 -> upper Hessenberg by
// Givens pairs) is VALU-bound at two waves per SIMD while a 30 x 30 window fills 30 .. 42 of its 64 lanes.  Both phases are
// lockstep loops (bounds depend on the window size only), so two draws of the same shape share one instruction stream: lanes
// 0..31 own draw 2p, lanes 32..63 draw 2p + 1; a pair of different shapes runs one half at a time (the other half masked).
// What the one-draw kernel broadcasts with v_readlane is read back from LDS (same address in every lane of a half):
// the reflector's source column, the pivots f, g of a Givens pair and the two entries of T that define the column rotation.
// Per Givens pair SIX 32-lane passes serve two draws (rows: H, X, T columns; columns: H, T, M rows) instead of five 64-lane
// passes per draw, and the rotation generators are shared.  LDS: [H | X] and T of both draws, 35.5 KB per wavefront at w = 30,
// #lead = 12 (four wavefronts per CU, one per SIMD: the launch is two rounds of a latency-bound chain).  M is kept REAL in
// the workspace (MRE), one row per lane, the shared column of consecutive rotations in a register, as in the one-draw kernel.
// MEASURED AND NOT THE DEFAULT (dsge_options.gensys_pairs = 2 selects it; tools/hess_pair_check.py): 92 VALU instructions per
// Givens pair for TWO draws against 104 per draw in gensys_hesstri_kernel -- 2.3 x fewer -- but 3.26 ms instead of 3.17 ms per
// 4096-draw gensys call: with a full H and T per draw the LDS holds one wavefront per SIMD, every instruction's latency is
// exposed (135 instructions x ~7 cycles per rotation) and the 2048 wavefronts need two rounds; the one-draw kernel runs nine
// draws per CU.  It would need ~4 wavefronts per SIMD to turn the saved instructions into time (H in a packed layout does not
// exist before it is Hessenberg).
__host__ __device__ inline size_t gp_hess_smem(const GwCaps& c) { return 2 * gw_reduce2_smem(c); }

__global__ __launch_bounds__(64) void gensys_hesstri_pair_kernel(int batch, GwCaps cp, double* __restrict__ ws,
                                                                  long long* __restrict__ dbg, const int32_t* __restrict__ act) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  batch = gw_active_count(batch, act);
  const int lane = threadIdx.x, h = lane >> 5, l = lane & 31;
  const int ldH = (cp.wcap + cp.lcap) | 1, ldW = cp.wcap | 1;
  const size_t per_half = (size_t)cp.wcap * ldH + (size_t)cp.wcap * ldW;
  double* hb = smem + (size_t)h * per_half;  // [H | X]: X(i, j) at column wcap + j
  double* tb = hb + (size_t)cp.wcap * ldH;
  const GwOffsets wo = gw_offsets(cp);
  const int npairs = (batch + 1) >> 1;
  for (int pair = blockIdx.x; pair < npairs; pair += gridDim.x) {
    const int draw = 2 * pair + h;
    const bool exists = draw < batch;
    double* wd = ws + (size_t)(exists ? draw : 2 * pair) * wo.total;
    const int* meta = reinterpret_cast<const int*>(wd + wo.meta);
    const bool valid = exists && meta[GW_FLAG] == 0;
    const int w_own = valid ? meta[GW_N] - meta[GW_Z] : 0, ell_own = valid ? meta[GW_ELL] : 0;
    double* MR = wd + wo.MRE;  // element (row, col) at MR[col * wcap + row]
    if (dbg && pair == 0 && lane == 0) dbg[5] = (long long)clock64();
    wave_sync();
    {  // the window as the deflation left it
      const int total = w_own * w_own, wdiv = w_own > 0 ? w_own : 1;
      for (int base = 0; base < cp.wcap * cp.wcap; base += 4 * 32) {
        double hv[4], tv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = base + u * 32 + l;
          const int ic = idx < total ? idx : 0;
          const int i = ic / wdiv, j = ic - i * wdiv;
          const size_t o = (size_t)i * cp.wcap + j;
          hv[u] = wd[wo.HR + o];
          tv[u] = wd[wo.TR + o];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = base + u * 32 + l;
          if (idx < total) {
            const int i = idx / wdiv, j = idx - i * wdiv;
            hb[i * ldH + j] = hv[u];
            tb[i * ldW + j] = tv[u];
          }
        }
      }
      const int totx = w_own * ell_own, ediv = ell_own > 0 ? ell_own : 1;
      for (int idx = l; idx < cp.wcap * cp.lcap; idx += 32)
        if (idx < totx) {
          const int i = idx / ediv, j = idx - i * ediv;
          hb[i * ldH + cp.wcap + j] = wd[wo.XR + (size_t)i * cp.lcap + j];
        }
      if (l < w_own)  // M = I, every lane the row it keeps reading and writing
        for (int col = 0; col < w_own; ++col) MR[(size_t)col * cp.wcap + l] = (l == col) ? 1.0 : 0.0;
    }
    wave_sync();
    // the shapes of the two halves (wave-uniform through readlane)
    const int wA = __builtin_amdgcn_readlane(w_own, 0), wB = __builtin_amdgcn_readlane(w_own, 32);
    const int eA = __builtin_amdgcn_readlane(ell_own, 0), eB = __builtin_amdgcn_readlane(ell_own, 32);
    const bool same = (wA == wB && eA == eB);
    const int nrun = same ? 1 : 2;
    for (int run = 0; run < nrun; ++run) {
      const int w = same ? wA : (run == 0 ? wA : wB), ell = same ? eA : (run == 0 ? eA : eB);
      const bool on = same ? (w_own == w && w >= 1) : (h == run && w >= 1);  // this half takes part in this run
      if (w < 2) continue;
      const int cw = min(l, w - 1);                  // column of H / T, row of H / T / M (lanes >= w duplicate the last)
      const int cxl = cp.wcap + min(l, max(ell - 1, 0));  // column of X inside hb
      // ---- T22 -> upper triangular (reflectors from the left on [H | X] and T) ------------------------------------------
      for (int j = 0; j < w - 1; ++j) {
        const double xv = (l >= j && l < w) ? tb[l * ldW + j] : 0.0;
        const double xnorm2 = half_sum((l > j) ? xv * xv : 0.0);
        if (__ballot(on && xnorm2 != 0.0) == 0ull) continue;
        const double alpha = tb[j * ldW + j];
        const bool live = xnorm2 != 0.0;
        const double nrm = sqrt(fma(alpha, alpha, xnorm2));
        const double beta = live ? ((alpha >= 0.0) ? -nrm : nrm) : alpha;
        const double tau = live ? (beta - alpha) / beta : 0.0;
        const double scal = live ? 1.0 / (alpha - beta) : 0.0;
        const double* px0 = tb + (size_t)(j + 1) * ldW + j;  // the source column below the diagonal (broadcast reads)
#pragma unroll 1
        for (int pass = 0; pass < 3; ++pass) {
          if (pass == 1 && ell == 0) continue;
          double* bA = pass == 0 ? hb + cw : (pass == 1 ? hb + cxl : tb + cw);
          const int lA = pass == 2 ? ldW : ldH;
          const double mj = bA[j * lA];
          double* pA = bA + (j + 1) * lA;
          const double* px = px0;
          double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
          int r = j + 1;
          for (; r + 4 <= w; r += 4) {
            const double m0 = pA[0], m1 = pA[lA], m2 = pA[2 * lA], m3 = pA[3 * lA];
            const double v0 = px[0], v1 = px[ldW], v2 = px[2 * ldW], v3 = px[3 * ldW];
            a0 = fma(v0, m0, a0);
            a1 = fma(v1, m1, a1);
            a2 = fma(v2, m2, a2);
            a3 = fma(v3, m3, a3);
            pA += 4 * lA;
            px += 4 * ldW;
          }
          for (; r < w; ++r) {
            a0 = fma(px[0], pA[0], a0);
            pA += lA;
            px += ldW;
          }
          const double wv = -tau * fma(scal, (a0 + a1) + (a2 + a3), mj);
          const double wsc = scal * wv;
          pA = bA + (j + 1) * lA;
          px = px0;
          for (r = j + 1; r + 4 <= w; r += 4) {
            double m0 = pA[0], m1 = pA[lA], m2 = pA[2 * lA], m3 = pA[3 * lA];
            const double v0 = px[0], v1 = px[ldW], v2 = px[2 * ldW], v3 = px[3 * ldW];
            m0 = fma(v0, wsc, m0);
            m1 = fma(v1, wsc, m1);
            m2 = fma(v2, wsc, m2);
            m3 = fma(v3, wsc, m3);
            if (on) {
              pA[0] = m0;
              pA[lA] = m1;
              pA[2 * lA] = m2;
              pA[3 * lA] = m3;
            }
            pA += 4 * lA;
            px += 4 * ldW;
          }
          for (; r < w; ++r) {
            const double m0 = fma(px[0], wsc, pA[0]);
            if (on) pA[0] = m0;
            pA += lA;
            px += ldW;
          }
          if (on) bA[j * lA] = mj + wv;  // row j last (the source column's rows below were read unscaled above)
        }
        if (on && l >= j && l < w) tb[l * ldW + j] = (l == j) ? beta : 0.0;
      }
      if (dbg && pair == 0 && lane == 0 && run == 0) dbg[2] = (long long)clock64();
      // ---- H22 -> upper Hessenberg by Givens pairs ----------------------------------------------------------------------
      for (int j = 0; j < w - 2; ++j) {
        double g = hb[(w - 1) * ldH + j];
        double m_hi = MR[(size_t)(w - 1) * cp.wcap + cw];
        double m_nx = MR[(size_t)(w - 2) * cp.wcap + cw];
        for (int i = w - 1; i > j + 1; --i) {
          const double m_lo_in = m_nx;
          m_nx = MR[(size_t)max(i - 2, 0) * cp.wcap + cw];
          const double f = hb[(i - 1) * ldH + j];
          double hx = hb[(i - 1) * ldH + cw], hy = hb[i * ldH + cw];
          double ax = hb[(i - 1) * ldH + cxl], ay = hb[i * ldH + cxl];
          double tx = tb[(i - 1) * ldW + cw], ty = tb[i * ldW + cw];
          double c, s, r;
          gw_lartg(f, g, c, s, r);
          rot2r(hx, hy, c, s);
          rot2r(ax, ay, c, s);
          rot2r(tx, ty, c, s);
          if (cw == j) {
            hx = r;
            hy = 0.0;
          }
          if (on) {
            hb[(i - 1) * ldH + cw] = hx;
            hb[i * ldH + cw] = hy;
            tb[(i - 1) * ldW + cw] = tx;
            tb[i * ldW + cw] = ty;
            if (ell > 0) {
              hb[(i - 1) * ldH + cxl] = ax;
              hb[i * ldH + cxl] = ay;
            }
          }
          g = r;
          // the column rotation that restores T(i, i-1) = 0 (read back behind the stores: LDS keeps program order)
          const double tii = tb[i * ldW + i], tim = tb[i * ldW + i - 1];
          double hx2 = hb[cw * ldH + i], hy2 = hb[cw * ldH + i - 1];
          double tx2 = tb[cw * ldW + i], ty2 = tb[cw * ldW + i - 1];
          double r2, m_lo = m_lo_in;
          gw_lartg(tii, tim, c, s, r2);
          rot2r(hx2, hy2, c, s);
          rot2r(tx2, ty2, c, s);
          rot2r(m_hi, m_lo, c, s);
          if (cw == i) {
            tx2 = r2;
            ty2 = 0.0;
          }
          if (on) {
            hb[cw * ldH + i] = hx2;
            hb[cw * ldH + i - 1] = hy2;
            tb[cw * ldW + i] = tx2;
            tb[cw * ldW + i - 1] = ty2;
            MR[(size_t)i * cp.wcap + cw] = m_hi;  // column i of M is final for this j
          }
          m_hi = m_lo;
        }
        if (on) MR[(size_t)(j + 1) * cp.wcap + cw] = m_hi;
      }
    }
    wave_sync();
    if (dbg && pair == 0 && lane == 0) dbg[3] = (long long)clock64();
    {  // hand the window on
      const int total = w_own * w_own, wdiv = w_own > 0 ? w_own : 1;
      for (int idx = l; idx < cp.wcap * cp.wcap; idx += 32)
        if (idx < total) {
          const int i = idx / wdiv, j = idx - i * wdiv;
          const size_t o = (size_t)i * cp.wcap + j;
          wd[wo.HR + o] = hb[i * ldH + j];
          wd[wo.TR + o] = tb[i * ldW + j];
        }
      const int totx = w_own * ell_own, ediv = ell_own > 0 ? ell_own : 1;
      for (int idx = l; idx < cp.wcap * cp.lcap; idx += 32)
        if (idx < totx) {
          const int i = idx / ediv, j = idx - i * ediv;
          wd[wo.XR + (size_t)i * cp.lcap + j] = hb[i * ldH + cp.wcap + j];
        }
    }
    if (dbg && pair == 0 && lane == 0) dbg[4] = (long long)clock64();
  }
}

}  // namespace dsge

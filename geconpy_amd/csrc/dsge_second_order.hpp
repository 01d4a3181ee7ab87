// Second-order perturbation + pruned-state-space quasi-likelihood (BASELINE.json configs[4]).
//
// The reference has NO second-order solver: it raises NotImplementedError at gEconpy/model/perturbation.py:97-98 and
// gEconpy/model/model.py:1433-1434, 1614-1615.  What is built here is the published algorithm (Schmitt-Grohe & Uribe 2004 for
// the coefficients; Kim, Kim, Schaumburg & Sims 2008 / Andreasen, Fernandez-Villaverde & Rubio-Ramirez 2018 for the pruned
// system) in the reduced formulation of oracle/second_order.py (`second_order_solution_reduced`,
// `pruned_state_space_reduced`, `pruned_kalman_logp`), which is what the kernels are checked against -- parity unpinned
// against the reference by construction.
//
// Notation (gEconpy/model/perturbation.py:42-46): F(y-, y, y+, u) = 0, Jacobians A, B, C, D, first order y = T y- + R u.
//   S = state variables (non-zero columns of A / T), s = |S|;  L = forward-looking variables (non-zero columns of C);
//   U = S followed by the observed non-states, u = |U|;  z = [y-; y; y+; u] indexes the Hessian, given as COO entries
//   (equation, z_a <= z_b) sorted by equation -- pattern shared by the draws, one value vector per draw.
//   M = B + C T,  G = M^-1 C (non-zero columns: L),  Ts = T[S, S],  Rs = R[S].
//
// Three kernels, one workgroup per draw, hand-overs through a per-draw workspace in HBM (SoLayout):
//   so_setup_kernel  (256 threads, VALU)   g_yy, g_yu, g_uu, g_ss: M^-1 by Gauss-Jordan, the Hessian contractions, the
//                                          generalised Sylvester equation X + G X (Ts (x) Ts) = M^-1 rhs by DOUBLING on the
//                                          state block (X_i as s x s matrices: X (Ts (x) Ts) is Ts' X_i Ts, only the rows L of
//                                          it are needed), then the pruned system Az', c, the factors of Qz, the stationary
//                                          mean
//   so_lyap_kernel   (512 threads, MFMA)   Qz = Y' L (one product), P0 = dlyap(Az, Qz) by doubling (three products per step)
//   so_filter_kernel (512 threads, MFMA)   the "standard" filter of oracle/statespace.py on the m = 2u + s(s+1)/2 dimensional
//                                          pruned state (207 on the SW-shaped workload): rank-p update on the VALU, the
//                                          prediction Az P+ Az' + Qz as two products on the FP64 matrix core
//                                          (dsge_so_gemm.hpp), steady-state switch as in the first-order kernels
#pragma once
#include "dsge_device.hpp"
#include "dsge_so_gemm.hpp"

namespace dsge {

constexpr int SO_MAX_S = 24, SO_MAX_K = 12, SO_MAX_P = 8;
constexpr int32_t DSGE_ST_SO_UNSUPPORTED = 128;  // (mirrors DSGE_ST_SECOND_ORDER_UNSUPPORTED of dsge_hip.h)

// Per-draw workspace (doubles).  Everything the three kernels hand to each other.
struct SoLayout {
  int n, k, s, u, l, p, q, m, MP, KQ, KQP;
  size_t gyy, gyu, guu, gss, x0, azt, az, cvec, a0, lt, yt, qz, p0, wt, xb, ak, akt, ak2, akt2, pp, qzj, total;
  __host__ __device__ static int pad8(int x) { return (x + 7) & ~7; }
  // mt: tiles of 16 per side of the kernel instance that will run (>= ceil(m / 16))
  __host__ __device__ void init(int n_, int k_, int s_, int u_, int l_, int p_, int mt) {
    n = n_; k = k_; s = s_; u = u_; l = l_; p = p_;
    q = s * (s + 1) / 2;
    m = 2 * u + q;
    MP = 16 * mt;
    KQ = k + s * k + k * (k + 1) / 2;
    KQP = pad8(KQ);
    size_t o = 0;
    auto take = [&](size_t cnt) { size_t r = o; o += (cnt + 1) & ~(size_t)1; return r; };
    gyy = take((size_t)n * s * s);
    gyu = take((size_t)n * s * k);
    guu = take((size_t)n * k * k);
    gss = take(n);
    x0 = take((size_t)n * s * s);
    const size_t mm = (size_t)MP * MP;
    azt = take(mm);   // Az' (row k = column k of Az): the k-major operand of Az X
    az = take(mm);    // Az
    cvec = take(MP);
    a0 = take(MP);
    lt = take((size_t)KQP * MP);
    yt = take((size_t)KQP * MP);
    qz = take(mm);
    p0 = take(mm);
    wt = take(mm);
    xb = take(mm);
    ak = take(mm);
    akt = take(mm);
    ak2 = take(mm);
    akt2 = take(mm);
    pp = take(mm);
    qzj = take(mm);   // Qz + jitter Az Az'
    total = o;
  }
};

// ---- small dense helpers for a workgroup of NT threads on LDS / L2-resident arrays ------------------------------------------

// In-place Gauss-Jordan with partial pivoting on the n x w array Aug (row stride ld, LDS): the leading n x n block becomes the
// identity, the remaining columns X with A X = rhs.  colbuf: n doubles, red: 2 ints (LDS).  Returns false on a zero / NaN pivot.
template <int NT>
__device__ __forceinline__ bool so_gauss_jordan(double* Aug, int ld, int n, int w, double* colbuf, int* red) {
  const int tid = threadIdx.x, lane = tid & 63;
  bool ok = true;
  for (int c = 0; c < n; ++c) {
    __syncthreads();
    if (tid < 64) {  // pivot: largest |entry| of column c among rows >= c
      double best = -1.0;
      int bi = c;
      for (int r = c + lane; r < n; r += 64) {
        const double v = fabs(Aug[r * ld + c]);
        if (v > best || v != v) { best = v; bi = r; }
      }
      for (int mk = 32; mk >= 1; mk >>= 1) {
        const double ob = __shfl_xor(best, mk, 64);
        const int oi = __shfl_xor(bi, mk, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      if (lane == 0) {
        red[0] = bi;
        red[1] = (best > 0.0 && best < 1e300) ? 1 : 0;
      }
    }
    __syncthreads();
    const int pr = red[0];
    if (!red[1]) ok = false;
    if (pr != c)
      for (int j = tid; j < w; j += NT) {
        const double t = Aug[c * ld + j];
        Aug[c * ld + j] = Aug[pr * ld + j];
        Aug[pr * ld + j] = t;
      }
    __syncthreads();
    const double inv = 1.0 / Aug[c * ld + c];
    for (int r = tid; r < n; r += NT) colbuf[r] = Aug[r * ld + c];
    __syncthreads();
    for (int j = tid; j < w; j += NT) Aug[c * ld + j] *= inv;
    __syncthreads();
    for (int idx = tid; idx < n * w; idx += NT) {
      const int r = idx / w, j = idx - r * w;
      if (r != c) Aug[r * ld + j] = fma(-colbuf[r], Aug[c * ld + j], Aug[r * ld + j]);
    }
  }
  __syncthreads();
  return ok;
}

// workgroup maximum of a per-thread value (NaN-propagating); red: NT / 64 doubles of LDS
template <int NT>
__device__ __forceinline__ double so_wg_max(double v, double* red) {
  v = wave_nanmax(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = red[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) r = nanmax(r, red[w]);
  return r;
}

struct SoIdx {  // the model's index sets, passed by value in the kernel arguments
  uint8_t S[SO_MAX_S];   // state variables (non-zero columns of A), ascending
  uint8_t U[40];         // retained variables: S, then the observed non-states
  uint8_t L[64];         // forward-looking variables (non-zero columns of C)
};

// sum_k a[k sa] b[k sb], k < K, on four independent accumulators with the eight loads of a trip requested together: the set-up
// kernel's inner loops are bounded by run-time sizes (s, l, n), which the compiler does not unroll -- one accumulator and one
// dependent load pair per term made every term cost a full LDS (or global) latency.
__device__ __forceinline__ double so_dot4(const double* a, int sa, const double* b, int sb, int K) {
  double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
  int k = 0;
  for (; k + 4 <= K; k += 4) {
    const double a0 = a[(size_t)k * sa], a1 = a[(size_t)(k + 1) * sa], a2 = a[(size_t)(k + 2) * sa], a3 = a[(size_t)(k + 3) * sa];
    const double b0 = b[(size_t)k * sb], b1 = b[(size_t)(k + 1) * sb], b2 = b[(size_t)(k + 2) * sb], b3 = b[(size_t)(k + 3) * sb];
    c0 = fma(a0, b0, c0);
    c1 = fma(a1, b1, c1);
    c2 = fma(a2, b2, c2);
    c3 = fma(a3, b3, c3);
  }
  for (; k < K; ++k) c0 = fma(a[(size_t)k * sa], b[(size_t)k * sb], c0);
  return (c0 + c1) + (c2 + c3);
}

struct SoSetupArgs {
  const double* B;
  const double* C;
  const double* T;
  const double* R;
  const int32_t* hess_ptr;   // [n + 1] offsets into the entry list, by equation (so_hessptr_kernel)
  const int32_t* hess_idx;   // [nnz][3] (equation, z_a <= z_b), sorted by equation
  const double* hess_val;    // [batch][nnz]
  const double* q;           // shock variances, [k] or [batch][k]
  SoIdx ix;
  const int32_t* flags;      // [2] device flags of the structure kernels (non-zero = unsupported input)
  double* work;              // [batch][layout.total]
  int32_t* status;           // [batch] in/out
  double* gyy_out;           // optional caller copies of the coefficients: [batch][n][s][s], [n][s][k], [n][k][k], [n]
  double* gyu_out;
  double* guu_out;
  double* gss_out;
  int32_t* order_key;        // optional [batch] dispatch key of the filter launch (see launch_second_order): bumped here for
                             // nearly singular M = B + C T, whose covariance recursion never leaves its rounding noise
  int batch, nnz, q_batched;
};

constexpr int SO_SETUP_THREADS = 256;

// LDS of the setup kernel (doubles): see the carve in the kernel
__host__ __device__ inline size_t so_setup_lds_doubles(const SoLayout& L) {
  const int n = L.n, k = L.k, s = L.s, l = L.l, mz = 3 * n + k;
  size_t o = 0;
  o += (size_t)n * (2 * n + 1);          // Maug = [M | I] -> [I | M^-1], one spare column
  o += (size_t)n * s + (size_t)n * k;    // Tc = T[:, S], Rm = R
  o += (size_t)mz * s + (size_t)mz * k;  // Zy, Zu
  o += (size_t)n * l;                    // GL = (M^-1 C)[:, L]
  o += (size_t)n * l;                    // Gk
  o += 2 * (size_t)l * s * s;            // Y1, Y2
  o += 3 * (size_t)s * s;                // Ts, Tk, Tk2
  o += (size_t)s * s;                    // Pf
  o += (size_t)l * l;                    // GLL scratch
  o += (size_t)n + 64;                   // colbuf, reductions
  o += 192;                              // three short vectors
  o += 96;                               // index lists
  return o + 64;
}

__global__ __launch_bounds__(SO_SETUP_THREADS) void so_setup_kernel(SoSetupArgs a, SoLayout lay) {
  constexpr int NT = SO_SETUP_THREADS;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, draw = blockIdx.x;
  if (draw >= a.batch) return;
  if (a.status[draw] != 0) return;  // failed first-order solve: nothing to do (logp = -inf downstream)
  const int n = lay.n, k = lay.k, s = lay.s, u = lay.u, l = lay.l, mz = 3 * n + k, ss = s * s;
  double* wk = a.work + (size_t)draw * lay.total;
  const double* Bg = a.B + (size_t)draw * n * n;
  const double* Cg = a.C + (size_t)draw * n * n;
  const double* Tg = a.T + (size_t)draw * n * n;
  const double* Rg = a.R + (size_t)draw * n * k;
  const double* hv = a.hess_val + (size_t)draw * a.nnz;
  const double* qv = a.q + (a.q_batched ? (size_t)draw * k : 0);
  // ---- carve LDS ---------------------------------------------------------------------------------------------------------
  double* p_ = smem;
  auto carve = [&](size_t cnt) { double* r = p_; p_ += cnt; return r; };
  const int ldm = 2 * n + 1;
  double* Maug = carve((size_t)n * ldm);
  double* Tc = carve((size_t)n * s);
  double* Rm = carve((size_t)n * k);
  double* Zy = carve((size_t)mz * s);
  double* Zu = carve((size_t)mz * k);
  double* GL = carve((size_t)n * l);
  double* Gk = carve((size_t)n * l);
  double* Y1 = carve((size_t)l * ss);
  double* Y2 = carve((size_t)l * ss);
  double* Ts = carve(ss);
  double* Tk = carve(ss);
  double* Tk2 = carve(ss);
  double* Pf = carve(ss);
  double* GLL = carve((size_t)l * l);
  double* colbuf = carve(n);
  double* red = carve(32);
  int* ired = (int*)carve(32);
  double* VW = Y2;  // V / W of steps 9 and 10 (l s k doubles, k <= s): Y2 is free between the doubling and step 12
  double* vec3 = carve(192);
  int* Si = (int*)carve(96);  // S (24), U (40), L (64) as ints
  int* Ui = Si + 24;
  int* Li = Ui + 40;
  if (tid < 24) Si[tid] = a.ix.S[tid];
  if (tid < 40) Ui[tid] = a.ix.U[tid];
  if (tid < 64) Li[tid] = a.ix.L[tid];
  __syncthreads();
  if (a.flags[0] | a.flags[1]) {
    if (tid == 0) a.status[draw] |= DSGE_ST_SO_UNSUPPORTED;
    return;
  }
  // ---- 1. T[:, S], R; structure check: T must vanish outside the columns S ------------------------------------------------
  bool bad = false;
  for (int idx = tid; idx < n * n; idx += NT) {
    const int j = idx % n;
    bool in_s = false;
    for (int c = 0; c < s; ++c) in_s = in_s || (Si[c] == j);
    if (!in_s && Tg[idx] != 0.0) bad = true;
  }
  for (int idx = tid; idx < n * s; idx += NT) Tc[idx] = Tg[(size_t)(idx / s) * n + Si[idx % s]];
  for (int idx = tid; idx < n * k; idx += NT) Rm[idx] = Rg[idx];
  __syncthreads();
  for (int idx = tid; idx < ss; idx += NT) Ts[idx] = Tc[Si[idx / s] * s + idx % s];
  // ---- 2. M = B + C T (only the columns S of C T are non-zero) -> Maug = [M | I] ------------------------------------------
  for (int idx = tid; idx < n * n; idx += NT) {
    const int i = idx / n, j = idx - i * n;
    Maug[i * ldm + j] = Bg[idx];
    Maug[i * ldm + n + j] = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
  for (int idx = tid; idx < n * s; idx += NT) {
    const int i = idx / s, c = idx - i * s;
    double acc = 0.0;
    for (int jj = 0; jj < l; ++jj) acc = fma(Cg[(size_t)i * n + Li[jj]], Tc[Li[jj] * s + c], acc);
    Maug[i * ldm + Si[c]] += acc;
  }
  // columns of C outside L must vanish (the caller's L is a model property)
  for (int idx = tid; idx < n * n; idx += NT) {
    const int j = idx % n;
    bool in_l = false;
    for (int c = 0; c < l; ++c) in_l = in_l || (Li[c] == j);
    if (!in_l && Cg[idx] != 0.0) bad = true;
  }
  if (__syncthreads_or(bad ? 1 : 0)) {
    if (tid == 0) a.status[draw] |= DSGE_ST_SO_UNSUPPORTED;
    return;
  }
  // M + C (for g_ss) is needed after M is gone: park it in the workspace's x0 block is too small -> use Az block (free until step 13)
  double* MpC = wk + lay.az;
  for (int idx = tid; idx < n * n; idx += NT) {
    const int i = idx / n, j = idx - i * n;
    MpC[idx] = Maug[i * ldm + j] + Cg[idx];
  }
  double mmax = 0.0;
  for (int idx = tid; idx < n * n; idx += NT) mmax = fmax(mmax, fabs(Maug[(idx / n) * ldm + idx % n]));
  mmax = so_wg_max<NT>(mmax, red);
  bool okm = so_gauss_jordan<NT>(Maug, ldm, n, 2 * n, colbuf, ired);
  const double* Mi = Maug + n;  // M^-1, row stride ldm
  if (a.order_key) {  // max|M| max|M^-1| > 1e6: dispatch it with the slowest draws
    double imax = 0.0;
    for (int idx = tid; idx < n * n; idx += NT) imax = fmax(imax, fabs(Mi[(idx / n) * ldm + idx % n]));
    imax = so_wg_max<NT>(imax, red);
    if (tid == 0 && !(mmax * imax <= 1e6)) a.order_key[draw] = 63;
  }
  // ---- 4. GL = (M^-1 C)[:, L];  5. Zy = [e_S; T[:, S]; (T T)[:, S]; 0],  Zu = [0; R; T R; I] ------------------------------
  for (int idx = tid; idx < n * l; idx += NT) {
    const int i = idx / l, jj = idx - i * l;
    double acc = 0.0;
    for (int r = 0; r < n; ++r) acc = fma(Mi[i * ldm + r], Cg[(size_t)r * n + Li[jj]], acc);
    GL[idx] = acc;
  }
  for (int idx = tid; idx < mz * s; idx += NT) {
    const int r = idx / s, c = idx - r * s;
    double v = 0.0;
    if (r < n) v = (Si[c] == r) ? 1.0 : 0.0;
    else if (r < 2 * n) v = Tc[(r - n) * s + c];
    else if (r < 3 * n) {
      const int i = r - 2 * n;
      for (int b = 0; b < s; ++b) v = fma(Tc[i * s + b], Ts[b * s + c], v);  // (T T)[:, S] = T[:, S] Ts
    }
    Zy[idx] = v;
  }
  for (int idx = tid; idx < mz * k; idx += NT) {
    const int r = idx / k, j = idx - r * k;
    double v = 0.0;
    if (r >= n && r < 2 * n) v = Rm[(r - n) * k + j];
    else if (r >= 2 * n && r < 3 * n) {
      const int i = r - 2 * n;
      for (int b = 0; b < s; ++b) v = fma(Tc[i * s + b], Rm[Si[b] * k + j], v);  // T R = T[:, S] R[S]
    } else if (r >= 3 * n) v = (r - 3 * n == j) ? 1.0 : 0.0;
    Zu[idx] = v;
  }
  __syncthreads();
  // ---- 6. rhs_yy = -H (Zy (x) Zy) -> x0 (global);  7. X = M^-1 rhs -> gyy block (global) ---------------------------------
  double* X0 = wk + lay.x0;
  double* X = wk + lay.gyy;
  for (int idx = tid; idx < n * ss; idx += NT) {
    const int i = idx / ss, cd = idx - i * ss, c = cd / s, d = cd - c * s;
    double acc = 0.0;
    for (int e = a.hess_ptr[i]; e < a.hess_ptr[i + 1]; ++e) {
      const int za = a.hess_idx[3 * e + 1], zb = a.hess_idx[3 * e + 2];
      const double v = hv[e];
      acc = fma(v, Zy[za * s + c] * Zy[zb * s + d], acc);
      if (za != zb) acc = fma(v, Zy[zb * s + c] * Zy[za * s + d], acc);
    }
    X0[idx] = -acc;
  }
  __syncthreads();
  for (int idx = tid; idx < n * ss; idx += NT) {
    const int i = idx / ss, cd = idx - i * ss;
    X[idx] = so_dot4(Mi + i * ldm, 1, X0 + cd, ss, n);
  }
  // ---- 8. doubling: X <- X - G_k (X (T_k (x) T_k)),  sign: X + G X (Ts (x) Ts) = X0  =>  X = sum_j (-G)^j X0 (Ts (x) Ts)^j ------
  for (int idx = tid; idx < n * l; idx += NT) Gk[idx] = -GL[idx];
  for (int idx = tid; idx < ss; idx += NT) Tk[idx] = Ts[idx];
  __syncthreads();
  bool conv = false;
  for (int it = 0; it < 40 && !conv; ++it) {
    // Y1[j] = X[L_j] T_k,  Y2[j] = T_k' Y1[j]
    for (int idx = tid; idx < l * ss; idx += NT) {
      const int jj = idx / ss, ad = idx - jj * ss, a_ = ad / s, d = ad - a_ * s;
      const double* xr = X + (size_t)Li[jj] * ss + a_ * s;
      Y1[idx] = so_dot4(xr, 1, Tk + d, s, s);
    }
    __syncthreads();
    for (int idx = tid; idx < l * ss; idx += NT) {
      const int jj = idx / ss, cd = idx - jj * ss, c = cd / s, d = cd - c * s;
      Y2[idx] = so_dot4(Tk + c, s, Y1 + jj * ss + d, s, s);
    }
    __syncthreads();
    double dmax = 0.0, xmax = 0.0;
    for (int idx = tid; idx < n * ss; idx += NT) {
      const int i = idx / ss, cd = idx - i * ss;
      const double xold = X[idx];
      const double acc = so_dot4(Gk + i * l, 1, Y2 + cd, ss, l);
      const double xn = xold + acc;
      X[idx] = xn;
      dmax = nanmax(dmax, fabs(acc));
      xmax = nanmax(xmax, fabs(xn));
    }
    // G_{k+1}[:, L] = G_k[:, L] G_k[L, L],  T_{k+1} = T_k T_k
    for (int idx = tid; idx < l * l; idx += NT) GLL[idx] = Gk[Li[idx / l] * l + idx % l];
    for (int idx = tid; idx < ss; idx += NT) {
      const int r = idx / s, c = idx - r * s;
      Tk2[idx] = so_dot4(Tk + r * s, 1, Tk + c, s, s);
    }
    dmax = so_wg_max<NT>(dmax, red);
    xmax = so_wg_max<NT>(xmax, red);  // (barriers inside: GLL, Tk2 and X are complete)
    double* Gn = GL;  // (GL is dead after step 8 starts: next G_k goes there, then swapped back)
    for (int idx = tid; idx < n * l; idx += NT) {
      const int i = idx / l, c = idx - i * l;
      Gn[idx] = so_dot4(Gk + i * l, 1, GLL + c, l, l);
    }
    __syncthreads();
    for (int idx = tid; idx < n * l; idx += NT) Gk[idx] = Gn[idx];
    for (int idx = tid; idx < ss; idx += NT) Tk[idx] = Tk2[idx];
    __syncthreads();
    if (!(dmax == dmax) || !(xmax < 1e300)) {
      okm = false;
      break;
    }
    conv = dmax <= 1e-17 * xmax;
  }
  if (!conv) okm = false;
  // GL was overwritten: recompute GL = (M^-1 C)[:, L] is not needed below (only M^-1 and C are)
  // ---- 9. g_yu = -M^-1 [H (Zy (x) Zu) + C g_yy (T (x) R)] -----------------------------------------------------------------
  // V[j][c][q] = sum_{a,b} X[L_j][a][b] Ts[a][c] Rs[b][q]
  for (int idx = tid; idx < l * s * k; idx += NT) {  // Y1[j][a][q] = sum_b X[L_j][a][b] Rs[b][q]
    const int jj = idx / (s * k), aq = idx - jj * s * k, a_ = aq / k, qq = aq - a_ * k;
    const double* xr = X + (size_t)Li[jj] * ss + a_ * s;
    double acc = 0.0;
    for (int b = 0; b < s; ++b) acc = fma(xr[b], Rm[Si[b] * k + qq], acc);
    Y1[idx] = acc;
  }
  __syncthreads();
  for (int idx = tid; idx < l * s * k; idx += NT) {
    const int jj = idx / (s * k), cq = idx - jj * s * k, c = cq / k, qq = cq - c * k;
    VW[idx] = so_dot4(Ts + c, s, Y1 + jj * s * k + qq, k, s);
  }
  __syncthreads();
  double* Gyu = wk + lay.gyu;
  for (int idx = tid; idx < n * s * k; idx += NT) {
    const int i = idx / (s * k), cq = idx - i * s * k, c = cq / k, qq = cq - c * k;
    double acc = 0.0;
    for (int e = a.hess_ptr[i]; e < a.hess_ptr[i + 1]; ++e) {
      const int za = a.hess_idx[3 * e + 1], zb = a.hess_idx[3 * e + 2];
      const double v = hv[e];
      acc = fma(v, Zy[za * s + c] * Zu[zb * k + qq], acc);
      if (za != zb) acc = fma(v, Zy[zb * s + c] * Zu[za * k + qq], acc);
    }
    for (int jj = 0; jj < l; ++jj) acc = fma(Cg[(size_t)i * n + Li[jj]], VW[jj * s * k + cq], acc);
    X0[idx] = acc;  // (x0 block reused: n s k <= n s s is not guaranteed -> sized n s s with s >= k checked by the launcher)
  }
  __syncthreads();
  for (int idx = tid; idx < n * s * k; idx += NT) {
    const int i = idx / (s * k), cq = idx - i * s * k;
    Gyu[idx] = -so_dot4(Mi + i * ldm, 1, X0 + cq, s * k, n);
  }
  __syncthreads();
  // ---- 10. g_uu = -M^-1 [H (Zu (x) Zu) + C g_yy (R (x) R)] ---------------------------------------------------------------
  for (int idx = tid; idx < l * k * k; idx += NT) {  // W[j][p][q] = sum_a Rs[a][p] Y1[j][a][q]  (Y1 of step 9 is still valid)
    const int jj = idx / (k * k), pq = idx - jj * k * k, pp = pq / k, qq = pq - pp * k;
    double acc = 0.0;
    for (int a_ = 0; a_ < s; ++a_) acc = fma(Rm[Si[a_] * k + pp], Y1[jj * s * k + a_ * k + qq], acc);
    VW[idx] = acc;
  }
  __syncthreads();
  double* Guu = wk + lay.guu;
  for (int idx = tid; idx < n * k * k; idx += NT) {
    const int i = idx / (k * k), pq = idx - i * k * k, pp = pq / k, qq = pq - pp * k;
    double acc = 0.0;
    for (int e = a.hess_ptr[i]; e < a.hess_ptr[i + 1]; ++e) {
      const int za = a.hess_idx[3 * e + 1], zb = a.hess_idx[3 * e + 2];
      const double v = hv[e];
      acc = fma(v, Zu[za * k + pp] * Zu[zb * k + qq], acc);
      if (za != zb) acc = fma(v, Zu[zb * k + pp] * Zu[za * k + qq], acc);
    }
    for (int jj = 0; jj < l; ++jj) acc = fma(Cg[(size_t)i * n + Li[jj]], VW[jj * k * k + pq], acc);
    X0[idx] = acc;
  }
  __syncthreads();
  for (int idx = tid; idx < n * k * k; idx += NT) {
    const int i = idx / (k * k), pq = idx - i * k * k;
    Guu[idx] = -so_dot4(Mi + i * ldm, 1, X0 + pq, k * k, n);
  }
  __syncthreads();
  // ---- 11. g_ss = -(M + C)^-1 [C g_uu + H (Zu' (x) Zu')] vec(Sigma),  Zu' = [0; 0; R; 0],  Sigma = diag(q) ------------------
  for (int i = tid; i < n; i += NT) {
    double acc = 0.0;
    for (int jj = 0; jj < l; ++jj) {
      double t = 0.0;
      for (int pp = 0; pp < k; ++pp) t = fma(Guu[(size_t)Li[jj] * k * k + pp * k + pp], qv[pp], t);
      acc = fma(Cg[(size_t)i * n + Li[jj]], t, acc);
    }
    for (int e = a.hess_ptr[i]; e < a.hess_ptr[i + 1]; ++e) {
      const int za = a.hess_idx[3 * e + 1], zb = a.hess_idx[3 * e + 2];
      if (za >= 2 * n && za < 3 * n && zb >= 2 * n && zb < 3 * n) {
        double t = 0.0;
        for (int pp = 0; pp < k; ++pp) t = fma(Rm[(za - 2 * n) * k + pp] * Rm[(zb - 2 * n) * k + pp], qv[pp], t);
        acc = fma(hv[e] * (za != zb ? 2.0 : 1.0), t, acc);
      }
    }
    colbuf[i] = acc;
  }
  __syncthreads();
  // Maug <- [M + C | rhs]: M^-1 is not needed any more
  for (int idx = tid; idx < n * n; idx += NT) Maug[(idx / n) * ldm + idx % n] = MpC[idx];
  for (int i = tid; i < n; i += NT) Maug[i * ldm + n] = colbuf[i];
  okm = so_gauss_jordan<NT>(Maug, ldm, n, n + 1, colbuf, ired) && okm;
  double* Gss = wk + lay.gss;
  for (int i = tid; i < n; i += NT) Gss[i] = -Maug[i * ldm + n];
  __syncthreads();
  // =========================================================================================================================
  // The pruned system on z = [x_f[U]; x_s[U]; w], w_(a<=b) = x_f[S_a] x_f[S_b] (row-major upper triangle)
  // =========================================================================================================================
  const int q_ = lay.q, m = lay.m, MP = lay.MP, KQP = lay.KQP;
  double* AzT = wk + lay.azt;
  double* Az = wk + lay.az;
  double* cvec = wk + lay.cvec;
  double* a0 = wk + lay.a0;
  double* Lt = wk + lay.lt;
  double* Yt = wk + lay.yt;
  for (size_t idx = tid; idx < (size_t)MP * MP; idx += NT) {
    AzT[idx] = 0.0;
    Az[idx] = 0.0;
  }
  for (size_t idx = tid; idx < (size_t)KQP * MP; idx += NT) {
    Lt[idx] = 0.0;
    Yt[idx] = 0.0;
  }
  for (int idx = tid; idx < MP; idx += NT) {
    cvec[idx] = 0.0;
    a0[idx] = 0.0;
  }
  // pair tables (a <= b) in LDS (ints over the dead Zy block)
  int* pa = (int*)Zy;
  int* pb = pa + q_;
  __syncthreads();
  for (int a_ = tid; a_ < s; a_ += NT) {
    int base = a_ * s - a_ * (a_ - 1) / 2;  // number of pairs before row a_
    for (int b = a_; b < s; ++b) {
      pa[base + b - a_] = a_;
      pb[base + b - a_] = b;
    }
  }
  // ---- 12. Pf = dlyap(Ts, Rs Sigma Rs') by doubling --------------------------------------------------------------------
  for (int idx = tid; idx < ss; idx += NT) {
    const int r = idx / s, c = idx - r * s;
    double acc = 0.0;
    for (int pp = 0; pp < k; ++pp) acc = fma(Rm[Si[r] * k + pp] * Rm[Si[c] * k + pp], qv[pp], acc);
    Pf[idx] = acc;
    Y2[idx] = acc;  // Rs Sigma Rs' kept for c_w
    Tk[idx] = Ts[idx];
  }
  __syncthreads();
  bool pconv = false;
  for (int it = 0; it < 40 && !pconv; ++it) {
    for (int idx = tid; idx < ss; idx += NT) {  // Y1 = T_k Pf
      const int r = idx / s, c = idx - r * s;
      double acc = 0.0;
      for (int b = 0; b < s; ++b) acc = fma(Tk[r * s + b], Pf[b * s + c], acc);
      Y1[idx] = acc;
    }
    __syncthreads();
    double dmax = 0.0, pmax = 0.0;
    for (int idx = tid; idx < ss; idx += NT) {
      const int r = idx / s, c = idx - r * s;
      double acc = 0.0, acct = 0.0;
      for (int b = 0; b < s; ++b) {
        acc = fma(Y1[r * s + b], Tk[c * s + b], acc);   // (T_k Pf T_k')[r][c]
        acct = fma(Y1[c * s + b], Tk[r * s + b], acct);  // its transpose entry: symmetrised increment
      }
      const double inc = 0.5 * (acc + acct);
      Tk2[idx] = inc;
      dmax = nanmax(dmax, fabs(inc));
    }
    __syncthreads();
    for (int idx = tid; idx < ss; idx += NT) {
      Pf[idx] += Tk2[idx];
      pmax = nanmax(pmax, fabs(Pf[idx]));
    }
    __syncthreads();
    for (int idx = tid; idx < ss; idx += NT) {
      const int r = idx / s, c = idx - r * s;
      double acc = 0.0;
      for (int b = 0; b < s; ++b) acc = fma(Tk[r * s + b], Tk[b * s + c], acc);
      Tk2[idx] = acc;
    }
    dmax = so_wg_max<NT>(dmax, red);
    pmax = so_wg_max<NT>(pmax, red);
    for (int idx = tid; idx < ss; idx += NT) Tk[idx] = Tk2[idx];
    __syncthreads();
    if (!(dmax == dmax) || !(pmax < 1e300)) break;
    pconv = dmax <= 1e-17 * pmax;
  }
  if (!pconv) okm = false;
  // ---- 13. Az and Az' ------------------------------------------------------------------------------------------------------
  for (int idx = tid; idx < u * s; idx += NT) {  // T[U, S] in the x_f and x_s blocks
    const int i = idx / s, c = idx - i * s;
    const double v = Tc[Ui[i] * s + c];
    Az[(size_t)i * MP + c] = v;
    AzT[(size_t)c * MP + i] = v;
    Az[(size_t)(u + i) * MP + u + c] = v;
    AzT[(size_t)(u + c) * MP + u + i] = v;
  }
  for (int idx = tid; idx < u * q_; idx += NT) {  // 1/2 g_yy on the symmetric half
    const int i = idx / q_, j = idx - i * q_, c = pa[j], d = pb[j];
    const double* g = X + (size_t)Ui[i] * ss;
    const double v = (c == d) ? 0.5 * g[c * s + c] : 0.5 * (g[c * s + d] + g[d * s + c]);
    Az[(size_t)(u + i) * MP + 2 * u + j] = v;
    AzT[(size_t)(2 * u + j) * MP + u + i] = v;
  }
  for (int idx = tid; idx < q_ * q_; idx += NT) {  // Elim (Ts (x) Ts) Dup
    const int i = idx / q_, j = idx - i * q_, a_ = pa[i], b_ = pb[i], c = pa[j], d = pb[j];
    double v = Ts[a_ * s + c] * Ts[b_ * s + d];
    if (c != d) v = fma(Ts[a_ * s + d], Ts[b_ * s + c], v);
    Az[(size_t)(2 * u + i) * MP + 2 * u + j] = v;
    AzT[(size_t)(2 * u + j) * MP + 2 * u + i] = v;
  }
  // ---- 14. c;  15. the stationary mean a0 ----------------------------------------------------------------------------------
  for (int i = tid; i < u; i += NT) {
    double acc = Gss[Ui[i]];
    for (int pp = 0; pp < k; ++pp) acc = fma(Guu[(size_t)Ui[i] * k * k + pp * k + pp], qv[pp], acc);
    cvec[u + i] = 0.5 * acc;
  }
  for (int j = tid; j < q_; j += NT) {
    cvec[2 * u + j] = Y2[pa[j] * s + pb[j]];
    a0[2 * u + j] = Pf[pa[j] * s + pb[j]];
  }
  __syncthreads();
  // b = Gh mean_w + c_xs (u entries);  x_S = (I - Ts)^-1 b_S by the product (I + T)(I + T^2)(I + T^4)... applied to b_S
  double* bv = vec3;
  double* xv = vec3 + 64;
  double* xn = vec3 + 128;
  for (int i = tid; i < u; i += NT) {
    double acc = cvec[u + i];
    for (int j = 0; j < q_; ++j) acc = fma(Az[(size_t)(u + i) * MP + 2 * u + j], a0[2 * u + j], acc);
    bv[i] = acc;
    if (i < s) xv[i] = acc;
  }
  for (int idx = tid; idx < ss; idx += NT) Tk[idx] = Ts[idx];
  __syncthreads();
  for (int it = 0; it < 40; ++it) {
    double dm = 0.0, xm = 0.0;
    for (int i = tid; i < s; i += NT) {
      double acc = 0.0;
      for (int b = 0; b < s; ++b) acc = fma(Tk[i * s + b], xv[b], acc);
      xn[i] = xv[i] + acc;
      dm = nanmax(dm, fabs(acc));
      xm = nanmax(xm, fabs(xn[i]));
    }
    for (int idx = tid; idx < ss; idx += NT) {
      const int r = idx / s, c = idx - r * s;
      double acc = 0.0;
      for (int b = 0; b < s; ++b) acc = fma(Tk[r * s + b], Tk[b * s + c], acc);
      Tk2[idx] = acc;
    }
    dm = so_wg_max<NT>(dm, red);
    xm = so_wg_max<NT>(xm, red);
    for (int i = tid; i < s; i += NT) xv[i] = xn[i];
    for (int idx = tid; idx < ss; idx += NT) Tk[idx] = Tk2[idx];
    __syncthreads();
    if (!(dm == dm) || !(xm < 1e300)) {
      okm = false;
      break;
    }
    if (dm <= 1e-18 * xm) break;
  }
  for (int i = tid; i < u; i += NT) {
    double acc = bv[i];
    for (int c = 0; c < s; ++c) acc = fma(Tc[Ui[i] * s + c], xv[c], acc);
    a0[u + i] = acc;
  }
  // ---- 16. the factors of Qz = sum_r Yt[r][.]' Lt[r][.] ---------------------------------------------------------------------
  //   rows [0, k):            L1 e:              L = [R[U]; 0; 0],                                   Y = q_j L
  //   rows k + a k + j:       L2 (x_f[S_a] e_j): L = [0; g_yu[U, a, j]; Ts[a_, a] Rs[b_, j] + Ts[b_, a] Rs[a_, j]],  Y = q_j sum_b L_(b, j) Pf[b][a]
  //   rows k + s k + (i<=j):  L3 (e_i e_j - .):  L = [0; 1/2 (g_uu[U, i, j] + [i != j] g_uu[U, j, i]); Rs[a_, i] Rs[b_, j] + [i != j] Rs[a_, j] Rs[b_, i]],
  //                                              Y = (i == j ? 2 q_i^2 : q_i q_j) L
  for (int idx = tid; idx < k * u; idx += NT) {
    const int j = idx / u, i = idx - j * u;
    const double v = Rm[Ui[i] * k + j];
    Lt[(size_t)j * MP + i] = v;
    Yt[(size_t)j * MP + i] = qv[j] * v;
  }
  for (int idx = tid; idx < s * k * (u + q_); idx += NT) {
    const int row = idx / (u + q_), col = idx - row * (u + q_), a_ = row / k, j = row - a_ * k;
    double v;
    int mcol;
    if (col < u) {
      v = Gyu[(size_t)Ui[col] * s * k + a_ * k + j];
      mcol = u + col;
    } else {
      const int pi = col - u, x_ = pa[pi], y_ = pb[pi];
      v = Ts[x_ * s + a_] * Rm[Si[y_] * k + j] + Ts[y_ * s + a_] * Rm[Si[x_] * k + j];
      mcol = 2 * u + pi;
    }
    Lt[(size_t)(k + row) * MP + mcol] = v;
  }
  __syncthreads();
  for (int idx = tid; idx < s * k * (u + q_); idx += NT) {
    const int row = idx / (u + q_), col = idx - row * (u + q_), a_ = row / k, j = row - a_ * k;
    const int mcol = col < u ? u + col : 2 * u + (col - u);
    Yt[(size_t)(k + row) * MP + mcol] = qv[j] * so_dot4(Lt + (size_t)(k + j) * MP + mcol, k * MP, Pf + a_, s, s);
  }
  {
    const int r0 = k + s * k;
    for (int idx = tid; idx < (k * (k + 1) / 2) * (u + q_); idx += NT) {
      const int t = idx / (u + q_), col = idx - t * (u + q_);
      int i = 0, rem = t;  // pair t -> (i <= j) of shocks, row-major upper triangle
      while (rem >= k - i) {
        rem -= k - i;
        ++i;
      }
      const int j = i + rem;
      double v;
      int mcol;
      if (col < u) {
        const double* g = Guu + (size_t)Ui[col] * k * k;
        v = (i == j) ? 0.5 * g[i * k + i] : 0.5 * (g[i * k + j] + g[j * k + i]);
        mcol = u + col;
      } else {
        const int pi = col - u, x_ = pa[pi], y_ = pb[pi];
        v = Rm[Si[x_] * k + i] * Rm[Si[y_] * k + j];
        if (i != j) v = fma(Rm[Si[x_] * k + j], Rm[Si[y_] * k + i], v);
        mcol = 2 * u + pi;
      }
      Lt[(size_t)(r0 + t) * MP + mcol] = v;
      Yt[(size_t)(r0 + t) * MP + mcol] = ((i == j) ? 2.0 * qv[i] * qv[i] : qv[i] * qv[j]) * v;
    }
  }
  (void)m;
  if (a.gyy_out)
    for (int idx = tid; idx < n * ss; idx += NT) a.gyy_out[(size_t)draw * n * ss + idx] = X[idx];
  if (a.gyu_out)
    for (int idx = tid; idx < n * s * k; idx += NT) a.gyu_out[(size_t)draw * n * s * k + idx] = Gyu[idx];
  if (a.guu_out)
    for (int idx = tid; idx < n * k * k; idx += NT) a.guu_out[(size_t)draw * n * k * k + idx] = Guu[idx];
  if (a.gss_out)
    for (int idx = tid; idx < n; idx += NT) a.gss_out[(size_t)draw * n + idx] = Gss[idx];
  if (!okm && tid == 0) a.status[draw] |= DSGE_ST_NOT_CONVERGED;
}

// ---- structure kernels (one small workgroup each, once per call) -------------------------------------------------------------
// offsets of the Hessian entries by equation; flag[0] != 0: entries unsorted / out of range / z_a > z_b
__global__ void so_hessptr_kernel(const int32_t* __restrict__ idx, int nnz, int n, int k, int32_t* __restrict__ ptr,
                                  int32_t* __restrict__ flag) {
  __shared__ int cnt[DSGE_MAX_N + 1];
  __shared__ int bad;
  const int tid = threadIdx.x;
  if (tid <= n) cnt[tid] = 0;
  if (tid == 0) bad = 0;
  __syncthreads();
  const int mz = 3 * n + k;
  for (int e = tid; e < nnz; e += blockDim.x) {
    const int i = idx[3 * e], za = idx[3 * e + 1], zb = idx[3 * e + 2];
    if (i < 0 || i >= n || za < 0 || zb >= mz || za > zb || (e > 0 && idx[3 * (e - 1)] > i)) bad = 1;
    else atomicAdd(&cnt[i], 1);
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int i = 0; i < n; ++i) {
      ptr[i] = acc;
      acc += cnt[i];
    }
    ptr[n] = acc;
    flag[0] = bad;
  }
}

// Zu = Z[:, U]; flag[1] != 0: Z has a non-zero (or NaN) entry outside the columns U
__global__ void so_design_kernel(const double* __restrict__ Z, int p, int n, SoIdx ix, int u, double* __restrict__ Zu,
                                 int32_t* __restrict__ flag) {
  __shared__ int bad;
  const int tid = threadIdx.x;
  if (tid == 0) bad = 0;
  __syncthreads();
  for (int idx = tid; idx < p * n; idx += blockDim.x) {
    const int j = idx % n;
    bool in_u = false;
    for (int c = 0; c < u; ++c) in_u = in_u || (ix.U[c] == j);
    if (!in_u && Z[idx] != 0.0) bad = 1;
  }
  for (int idx = tid; idx < p * u; idx += blockDim.x) Zu[idx] = Z[(size_t)(idx / u) * n + ix.U[idx % u]];
  __syncthreads();
  if (tid == 0) flag[1] = bad;
}


// =============================================================================================================================
// Qz and the stationary covariance P0 = dlyap(Az, Qz) (512 threads, products on the matrix core)
// =============================================================================================================================
struct SoFilterArgs {
  double* work;             // [batch][layout.total]
  const double* Zu;         // [p][u]  design matrix restricted to the retained variables (Z[:, U])
  const double* d;          // [p] or nullptr
  const double* Hdiag;      // [p] or nullptr
  const double* y;          // [T_len][p]
  double* logp;             // [batch]
  int32_t* status;          // [batch] in/out
  const int32_t* order;     // [batch] or nullptr: workgroup b filters draw order[b] (slow draws first, kalman_order_kernel)
  int32_t* steady_at;       // [batch] or nullptr (debug: first steady step)
  int32_t* n_doublings;     // [batch] or nullptr (debug)
  long long* phases;        // device int64[8] or nullptr (debug): shader cycles draw 0 spends in [0] P Z', F, gain; [1] the pass
                            // over Az'; [2] first product; [3] second product; [4] steady steps; [5] full steps; [6] steady steps (count); [7] total
  int batch, T_len;
  FilterConv cv;            // third-party conventions of the filter step (dsge_device.hpp)
  double missing_fill, steady_tol;
};

// x <- (Lc Lc')^-1 x (FWD_ONLY: x <- Lc^-1 x) for the p x p Cholesky factor of F, IDENTITY-PADDED to 8 x 8 with the reciprocals of
// its diagonal in Li: compile-time loop bounds, so x stays in registers.  (With loops bounded by the run-time p the register
// array x was addressed dynamically, i.e. lived in scratch memory: every one of the ~50 dependent accesses of a row's two
// triangular solves was a memory round trip -- most of the 120 k cycles of the update phase and of the 60 k of Az K, Az V.)
template <bool FWD_ONLY>
__device__ __forceinline__ void so_chol_solve8(double (&x)[8], const double* __restrict__ Lc, const double* __restrict__ Li) {
#pragma unroll
  for (int o = 0; o < 8; ++o) {  // Lc z = x
    double sv = x[o];
#pragma unroll
    for (int r = 0; r < o; ++r) sv = fma(-Lc[o * 8 + r], x[r], sv);
    x[o] = sv * Li[o];
  }
  if (FWD_ONLY) return;
#pragma unroll
  for (int o = 7; o >= 0; --o) {  // Lc' k = z
    double sv = x[o];
#pragma unroll
    for (int r = o + 1; r < 8; ++r) sv = fma(-Lc[r * 8 + o], x[r], sv);
    x[o] = sv * Li[o];
  }
}

template <int MT>
struct SoFilterSmem {
  static constexpr int MP = 16 * MT;
  // GEMM staging + vectors: a, ap (MP each), K, PZ (MP x 8 each), F, Lc (64 each), v, w, dvec, hvec (8 each), reductions
  // ... + Az K, Az V (MP x 8 each) and the partial sums of the mean prediction (2 MP)
  static constexpr size_t doubles = SoGemmCfg<MT>::LDS_DOUBLES + 2 * MP + 4 * MP * 8 + 2 * MP + 2 * 64 + 4 * 8 + 64 +
                                    3 * SO_MAX_S * SO_MAX_S;  // + Ts, unvech(w), Ts unvech(w) of the structured mean prediction
  static constexpr size_t bytes = doubles * sizeof(double);
};

// add pass of the doubling: P_new = X + base (+ P_old if ACC) for an exactly symmetric X; returns max |P_new -
// P_old| and max |P_new| in dmax / pmax (workgroup-uniform)
template <int NT, bool ACC>
__device__ __forceinline__ void so_sym_update(double* P, const double* Xb, const double* base, int MP, int m, double* red,
                                              double& dmax, double& pmax) {
  double dm = 0.0, pm = 0.0;
  // Xb comes from so_gemm_sym: exactly symmetric, so the (uncoalesced) average with its transpose is gone; four rows of pairs
  // per trip with their loads requested together (padding rows / columns of Xb, P and base are zero and stay zero)
  constexpr int FX = 4;
  const int npairs = m * (MP / 2);
  for (int idx0 = threadIdx.x; idx0 < npairs; idx0 += FX * NT) {
    double2 x[FX], po[FX], bs[FX];
#pragma unroll
    for (int f = 0; f < FX; ++f) {
      const int idx = idx0 + f * NT < npairs ? idx0 + f * NT : idx0;
      x[f] = ((const double2*)Xb)[idx];
      po[f] = ((const double2*)P)[idx];
      if (!ACC) bs[f] = ((const double2*)base)[idx];
    }
#pragma unroll
    for (int f = 0; f < FX; ++f) {
      const int idx = idx0 + f * NT;
      if (idx < npairs) {
        double2 pn;
        pn.x = ACC ? po[f].x + x[f].x : x[f].x + bs[f].x;
        pn.y = ACC ? po[f].y + x[f].y : x[f].y + bs[f].y;
        ((double2*)P)[idx] = pn;
        dm = nanmax(dm, nanmax(fabs(pn.x - po[f].x), fabs(pn.y - po[f].y)));
        pm = nanmax(pm, nanmax(fabs(pn.x), fabs(pn.y)));
      }
    }
  }
  dmax = so_wg_max<NT>(dm, red);
  pmax = so_wg_max<NT>(pm, red);
}

template <int MT>
__global__ __launch_bounds__(SO_THREADS) void so_lyap_kernel(SoFilterArgs a, SoLayout lay) {
  constexpr int NT = SO_THREADS, MP = 16 * MT;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* lds = smem;
  double* red = smem + SoGemmCfg<MT>::LDS_DOUBLES;
  const int tid = threadIdx.x, draw = blockIdx.x;
  if (draw >= a.batch) return;
  if (a.status[draw] != 0) return;
  double* wk = a.work + (size_t)draw * lay.total;
  const int m = lay.m;
  double* Qz = wk + lay.qz;
  double* P = wk + lay.p0;
  double* Wt = wk + lay.wt;
  double* Xb = wk + lay.xb;
  double* Ak = wk + lay.ak;
  double* AkT = wk + lay.akt;
  double* Ak2 = wk + lay.ak2;
  double* AkT2 = wk + lay.akt2;
  // Qz = sym(Y' L)
  so_gemm<MT>(wk + lay.yt, MP, wk + lay.lt, MP, lay.KQP, lds, [&](int r, int c, so_v4f64 v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) Xb[(size_t)(r + 4 * e) * MP + c] = v[e];
  });
  __syncthreads();
  for (int idx = tid; idx < MP * MP; idx += NT) {
    const int i = idx / MP, j = idx - i * MP;
    const double v = (i < m && j < m) ? 0.5 * (Xb[(size_t)i * MP + j] + Xb[(size_t)j * MP + i]) : 0.0;
    Qz[idx] = v;
    P[idx] = v;
    Ak[idx] = wk[lay.az + idx];
    AkT[idx] = wk[lay.azt + idx];
  }
  bool conv = false;
  int it = 0, extra = 0;
  // (one more doubling after the increment has dropped below 1e-17 max |P|: the blocks of P differ by orders of magnitude and
  // the test only sees the largest; the error of the doubling squares with every step)
  for (; it < 48 && extra < 2; ++it) {
    // W' = P A_k' (= (A_k P)': P is symmetric; natural stores),  X = W A_k',  A_{k+1} = A_k A_k (both layouts)
    so_gemm<MT>(P, MP, AkT, MP, MP, lds, [&](int r, int c, so_v4f64 v) {  // (P symmetric: P A_k' = (A_k P)', natural stores)
#pragma unroll
      for (int e = 0; e < 4; ++e) Wt[(size_t)(r + 4 * e) * MP + c] = v[e];
    });
    so_gemm_sym<MT>(Wt, AkT, MP, lds, [&](int r, int c, so_v4f64 v) {  // (A_k P A_k': symmetric, upper tiles + mirror)
#pragma unroll
      for (int e = 0; e < 4; ++e) Xb[(size_t)(r + 4 * e) * MP + c] = v[e];
    });
    so_gemm<MT>(
        AkT, MP, Ak, MP, MP, lds,
        [&](int r, int c, so_v4f64 v) {
#pragma unroll
          for (int e = 0; e < 4; ++e) Ak2[(size_t)(r + 4 * e) * MP + c] = v[e];
        },
        SoZeroInit(),
        [&](int r, int c, so_v4f64 v) {  // (the transposed tiles, with natural store addresses)
#pragma unroll
          for (int e = 0; e < 4; ++e) AkT2[(size_t)(r + 4 * e) * MP + c] = v[e];
        });
    __syncthreads();
    double dmax, pmax;
    so_sym_update<NT, true>(P, Xb, nullptr, MP, m, red, dmax, pmax);
    double* t = Ak; Ak = Ak2; Ak2 = t;
    t = AkT; AkT = AkT2; AkT2 = t;
    if (!(dmax == dmax) || !(pmax < 1e300)) break;
    conv = conv || dmax <= 1e-17 * pmax;
    extra += conv ? 1 : 0;
  }
  // Qz + jitter Az Az' for the filter's prediction step (the jitter of P+ = ... + jitter I, carried through Az . Az')
  {
    const double* AzT = wk + lay.azt;
    double* Qzj = wk + lay.qzj;
    const double jit = a.cv.jit_P;
    so_gemm_sym<MT>(AzT, AzT, MP, lds, [&](int r, int c, so_v4f64 v) {  // (Az Az': a Gram matrix)
#pragma unroll
      for (int e = 0; e < 4; ++e) Qzj[(size_t)(r + 4 * e) * MP + c] = fma(jit, v[e], Qz[(size_t)(r + 4 * e) * MP + c]);
    });
  }
  if (tid == 0) {
    if (!conv) a.status[draw] |= DSGE_ST_LYAP_FAIL;
    if (a.n_doublings) a.n_doublings[draw] = it;
  }
}

// a = Az a+ + c by all the threads: two halves of the sum over k per output, eight loads of Az' in flight per thread
template <int NT>
__device__ __forceinline__ void so_mean_predict(const double* __restrict__ AzT, const double* ap, const double* __restrict__ cvec,
                                                double* av, double* part, int MP, int m) {
  const int tid = threadIdx.x, half = tid >= NT / 2 ? 1 : 0, i = tid - half * (NT / 2);
  const int kh = (m + 1) / 2, k0 = half * kh, k1 = half ? m : kh;
  if (i < m) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    int kx = k0;
#pragma unroll 1
    for (; kx + 16 <= k1; kx += 16) {  // (sixteen loads in flight)
      double z[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) z[e] = AzT[(size_t)(kx + e) * MP + i];
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e & 3] = fma(z[e], ap[kx + e], acc[e & 3]);
    }
#pragma unroll 1
    for (; kx + 8 <= k1; kx += 8) {
      double z[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = AzT[(size_t)(kx + e) * MP + i];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e & 3] = fma(z[e], ap[kx + e], acc[e & 3]);
    }
    for (; kx < k1; ++kx) acc[0] = fma(AzT[(size_t)kx * MP + i], ap[kx], acc[0]);
    part[half * MP + i] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  }
  __syncthreads();
  if (tid < m) av[tid] = cvec[tid] + part[tid] + part[MP + tid];
  __syncthreads();
}

// The same prediction through the STRUCTURE of Az (so_setup_kernel, step 13):
//     x_f' = T[U, S] x_f[S],    x_s' = T[U, S] x_s[S] + 1/2 g_yy[U] w,    w' = vech(Ts unvech(w) Ts') + (constants in c)
// -- 36 x 18 + 36 x 189 entries of Az' and Ts instead of all 207 x 207 (59 KB instead of 343 KB per step: a steady step is
// this prediction and nothing else, and 256 workgroups reading their Az' every 30 k cycles are 6 TB/s of Infinity-Cache
// traffic, i.e. the steady steps were bandwidth-bound).  The w block is two s x s products: Y = Ts X, X' = Y Ts' (X = unvech(w)).
// Ts: LDS copy of T[S, S] (s x s), Xs, Ys: s x s scratch, pr: pr_cap doubles of scratch (the Az K buffer is free here).
template <int NT>
__device__ __forceinline__ void so_mean_predict_structured(const double* __restrict__ AzT, const double* ap,
                                                           const double* __restrict__ cvec, double* av, const double* Ts,
                                                           double* Xs, double* Ys, double* pr, int pr_cap, int MP, int u, int s,
                                                           int q_) {
  const int tid = threadIdx.x;
  const int G = (NT / u < pr_cap / (2 * u)) ? NT / u : pr_cap / (2 * u);  // k-groups: as many as threads and scratch allow
  const int kg = tid / u, io = tid - kg * u, nk = s + q_;
  // ---- phase A: unvech(w), and the partial sums of the x_f and x_s outputs over this thread's share of the state
  for (int idx = tid; idx < s * s; idx += NT) {
    const int c = idx / s, d = idx - c * s, a_ = c < d ? c : d, b_ = c < d ? d : c;
    Xs[idx] = ap[2 * u + a_ * s - (a_ * (a_ - 1)) / 2 + (b_ - a_)];
  }
  if (kg < G) {
    double sf = 0.0, ss0 = 0.0, ss1 = 0.0;
    for (int kf = kg; kf < s; kf += G) sf = fma(AzT[(size_t)kf * MP + io], ap[kf], sf);  // x_f'[io] <- T[U_io, S_k] x_f[S_k]
    for (int k0 = kg; k0 < nk; k0 += 8 * G) {              // x_s'[io]: up to eight loads in flight
      double z[8], xk[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int kk = k0 + e * G, kc = kk < nk ? kk : kg;
        const int row = kc < s ? u + kc : 2 * u + (kc - s);
        z[e] = AzT[(size_t)row * MP + u + io];
        xk[e] = kk < nk ? ap[row] : 0.0;
      }
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        ss0 = fma(z[e], xk[e], ss0);
        ss1 = fma(z[e + 1], xk[e + 1], ss1);
      }
    }
    pr[kg * 2 * u + io] = sf;
    pr[kg * 2 * u + u + io] = ss0 + ss1;
  }
  __syncthreads();
  // ---- phase B: Y = Ts X; the x_f, x_s outputs
  for (int idx = tid; idx < s * s; idx += NT) {
    const int a_ = idx / s, d = idx - a_ * s;
    double acc = 0.0;
    for (int c = 0; c < s; ++c) acc = fma(Ts[a_ * s + c], Xs[c * s + d], acc);
    Ys[idx] = acc;
  }
  if (tid >= NT - 2 * u) {  // (the last 2u threads: the first s^2 are busy with Y)
    const int i = tid - (NT - 2 * u);
    double acc = cvec[i];
    for (int g = 0; g < G; ++g) acc += pr[g * 2 * u + i];
    av[i] = acc;
  }
  __syncthreads();
  // ---- phase C: w' = vech(Y Ts') + c
  for (int j = tid; j < q_; j += NT) {
    // pair j -> (a, b), a <= b, row-major upper triangle
    int a_ = 0, base = 0;
    while (base + (s - a_) <= j) {
      base += s - a_;
      ++a_;
    }
    const int b_ = a_ + (j - base);
    double acc = 0.0;
    for (int d = 0; d < s; ++d) acc = fma(Ys[a_ * s + d], Ts[b_ * s + d], acc);
    av[2 * u + j] = cvec[2 * u + j] + acc;
  }
  __syncthreads();
}

// =============================================================================================================================
// The filter (oracle/statespace.py kalman_filter_logp on the pruned system: a0 = stationary mean, P0 = stationary covariance,
// state intercept c, design [Zu, Zu, 0]).  Update in the rank-p form of the first-order kernels:
//     P+ = P - sym(K (P Z' + jitter K)') + jitter I
// =============================================================================================================================
template <int MT>
__global__ __launch_bounds__(SO_THREADS) void so_filter_kernel(SoFilterArgs a, SoLayout lay) {
  constexpr int NT = SO_THREADS, MP = 16 * MT, PM = SO_MAX_P;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* lds = smem;
  double* pl = smem + SoGemmCfg<MT>::LDS_DOUBLES;
  double* av = pl; pl += MP;        // predicted mean
  double* ap = pl; pl += MP;        // filtered mean
  double* Kg = pl; pl += MP * PM;   // gain, o-major: Kg[o MP + i] = K[i][o] (the operand transform below reads two adjacent i at once)
  double* PZ = pl; pl += MP * PM;   // P Z', then V = P Z' + jitter K, same layout
  double* AK = pl; pl += MP * PM;   // Az K and Az V of the current full step, o-major like Kg
  double* AV = pl; pl += MP * PM;
  double* part = pl; pl += 2 * MP;  // partial sums of the mean prediction
  double* Fm = pl; pl += 64;        // F, then its Cholesky factor (lower)
  double* Lc = pl; pl += 64;
  double* vv = pl; pl += 8;         // innovation
  double* Li = pl; pl += 8;         // reciprocals of the diagonal of Lc (1 beyond p)
  double* dv = pl; pl += 8;
  double* hv = pl; pl += 8;
  double* red = pl; pl += 32;
  int* imask = (int*)pl; pl += 8;   // [0] = current mask, [1] = steady flag, [2] = finite flag
  double* Tss = pl; pl += SO_MAX_S * SO_MAX_S;  // T[S, S] (the x_f block of Az) for the structured mean prediction
  double* Xss = pl; pl += SO_MAX_S * SO_MAX_S;
  double* Yss = pl; pl += SO_MAX_S * SO_MAX_S;
  // The launch's makespan is set by the draws whose covariance recursion reaches its fixed point late (up to T_len full
  // steps against ~60 on average): like the first-order filter, the workgroups take the draws in the caller's order
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= a.batch) return;
  const int draw = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
  if (a.status[draw] != 0) {
    if (tid == 0) a.logp[draw] = -INFINITY;
    return;
  }
  double* wk = a.work + (size_t)draw * lay.total;
  const int m = lay.m, u = lay.u, p = lay.p;
  const double* AzT = wk + lay.azt;
  const double* Qzj = wk + lay.qzj;
  const double* cvec = wk + lay.cvec;
  double* Pc = wk + lay.p0;  // current predicted covariance
  double* Pn = wk + lay.pp;  // the other buffer: written by the prediction, then swapped
  double* Wt = wk + lay.wt;
  const double* Zu = a.Zu;
  for (int i = tid; i < MP; i += NT) av[i] = wk[lay.a0 + i];
  for (int idx = tid; idx < lay.s * lay.s; idx += NT) Tss[idx] = AzT[(size_t)(idx % lay.s) * MP + idx / lay.s];  // Ts[a][c] = Az[a][c]
  if (tid < 8) {
    dv[tid] = (a.d && tid < p) ? a.d[tid] : 0.0;
    hv[tid] = (a.Hdiag && tid < p) ? a.Hdiag[tid] : 0.0;
  }
  for (int idx = tid; idx < MP * PM; idx += NT) {  // (rows beyond m stay zero: the padding of P+ is the padding of P)
    Kg[idx] = 0.0;
    PZ[idx] = 0.0;
  }
  __syncthreads();
  const double LN2PI = 1.8378770664093453;
  double ll_sum = 0.0, logdet = 0.0;  // (thread 0)
  bool steady = false, finite = true;
  int steady_mask = -1, steady_at = -1;
  const bool stamp = a.phases != nullptr && draw == 0 && tid == 0;
  long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long t_begin = stamp ? clock64() : 0;
  double y_next = (tid < p && a.T_len > 0) ? a.y[tid] : 0.0;  // (this thread's entry of the next observation row)
  const int tid_kernel = tid;
  for (int t = 0; t < a.T_len; ++t) {
    // (thread index re-derived from an opaque copy per step: what is computed from it would otherwise be hoisted in front of the
    //  time loop and spilled there -- see crc_iterate / kalman_nt_kernel)
    int tid = tid_kernel;
    asm volatile("" : "+v"(tid));
    long long tk = stamp ? clock64() : 0;
    // ---- missing-data mask of this step (bit o set = observed); the observation row was requested one step ahead --------------
    const double y_now = y_next;
    if (tid < p && t + 1 < a.T_len) y_next = a.y[(size_t)(t + 1) * p + tid];
    if (tid < 64) {
      const unsigned long long ob = __ballot(tid < p && !(y_now != y_now) && y_now != a.missing_fill);
      if (tid == 0) imask[0] = (int)(ob & 0xffull);
    }
    __syncthreads();
    const int mask = imask[0];
    if (steady && mask != steady_mask) steady = false;  // (workgroup-uniform)
    // ---- innovation v = ym - d - Zm a -------------------------------------------------------------------------------------
    if (tid < p) {
      const int o = tid;
      double za = 0.0;
      if ((mask >> o) & 1)
        for (int c = 0; c < u; ++c) za = fma(Zu[o * u + c], av[c] + av[u + c], za);
      const double yo = ((mask >> o) & 1) ? y_now : 0.0;
      vv[o] = yo - ((((mask >> o) & 1) || !a.cv.mask_d) ? dv[o] : 0.0) - za;
    }
    __syncthreads();  // (the innovation is read by every wavefront below, also on the steady path)
    if (!steady) {
      // ---- P Z' (m x p), F = Zm P Zm' + Hm + jitter I ---------------------------------------------------------------------
      // (P is symmetric: column i is read as row entries P[c][i], coalesced over i.  Two groups of 256 threads share the 2u rows:
      //  the pass is a chain of load latencies -- ~10 k cycles each under the launch's own HBM traffic --, not of arithmetic)
      {
        static_assert(NT == 512, "two groups of 256 threads");
        const int g = tid >> 8, i = tid & 255, uh = (u + 1) / 2, cb = g * uh, ce = (cb + uh < u) ? cb + uh : u;
        double acc[PM];
#pragma unroll
        for (int o = 0; o < PM; ++o) acc[o] = 0.0;
        if (i < m) {
          for (int c0 = cb; c0 < ce; c0 += 12) {  // (24 loads in flight)
            double pv[24];
#pragma unroll
            for (int e = 0; e < 12; ++e) {
              const int c = c0 + e < ce ? c0 + e : cb;
              pv[e] = Pc[(size_t)c * MP + i];
              pv[12 + e] = Pc[(size_t)(u + c) * MP + i];
            }
#pragma unroll
            for (int e = 0; e < 12; ++e) {
              const double pc = (c0 + e < ce) ? pv[e] + pv[12 + e] : 0.0;
              const int c = c0 + e < ce ? c0 + e : cb;
#pragma unroll
              for (int o = 0; o < PM; ++o)
                if (o < p) acc[o] = fma(pc, Zu[o * u + c], acc[o]);
            }
          }
          if (g == 1) {
#pragma unroll
            for (int o = 0; o < PM; ++o) AV[o * MP + i] = acc[o];  // (AV is free until the second half of the step)
          }
        }
        __syncthreads();
        if (g == 0 && i < m) {
#pragma unroll
          for (int o = 0; o < PM; ++o) PZ[o * MP + i] = (o < p && ((mask >> o) & 1)) ? acc[o] + AV[o * MP + i] : 0.0;
        }
      }
      __syncthreads();
      if (tid < p * p) {
        const int o = tid / p, o2 = tid - o * p;
        double acc = 0.0;
        if ((mask >> o) & 1)
          for (int c = 0; c < u; ++c) acc = fma(Zu[o * u + c], PZ[o2 * MP + c] + PZ[o2 * MP + u + c], acc);
        if (o == o2) acc += (((mask >> o) & 1) ? hv[o] : 0.0) + a.cv.jit_F;
        Fm[o * 8 + o2] = acc;
      }
      __syncthreads();
      // ---- Cholesky F = Lc Lc' (thread 0; p <= 8), log det F.  In registers with compile-time bounds on the identity-padded
      //      8 x 8 matrix (sym(F) in the leading p x p block): the version that worked on the LDS copy with loops bounded by p
      //      was a chain of ~150 dependent LDS round trips ---------------------------------------------------------------
      if (tid == 0) {
        double L[8][8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j)
            L[i][j] = (i < p && j < p) ? 0.5 * (Fm[i * 8 + j] + Fm[j * 8 + i]) : ((i == j) ? 1.0 : 0.0);
        double ld = 0.0;
        bool okc = true;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          double dsum = L[j][j];
#pragma unroll
          for (int r = 0; r < j; ++r) dsum = fma(-L[j][r], L[j][r], dsum);
          if (!(dsum > 0.0)) okc = false;
          const double dj = sqrt(dsum), rj = 1.0 / dj;
          L[j][j] = dj;
          Li[j] = rj;
          if (j < p) ld += 2.0 * log(dj);
#pragma unroll
          for (int i = j + 1; i < 8; ++i) {
            double sv = L[i][j];
#pragma unroll
            for (int r = 0; r < j; ++r) sv = fma(-L[i][r], L[j][r], sv);
            L[i][j] = sv * rj;
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) Lc[i * 8 + j] = (j <= i) ? L[i][j] : 0.0;
        logdet = ld;
        if (!okc) finite = false;
        imask[2] = okc ? 1 : 0;
      }
      __syncthreads();
      // ---- gain K = P Z' F^-1 (one row per thread: two triangular solves) ---------------------------------------------------
      for (int i = tid; i < m; i += NT) {
        double x[PM];
#pragma unroll
        for (int o = 0; o < PM; ++o) x[o] = (o < p) ? PZ[o * MP + i] : 0.0;
        so_chol_solve8<false>(x, Lc, Li);
#pragma unroll
        for (int o = 0; o < PM; ++o) Kg[o * MP + i] = x[o];
      }
      __syncthreads();
    }
    // ---- F^-1 v, log-likelihood contribution, filtered mean ---------------------------------------------------------------
    if (tid == 0) {
      double x[PM];
#pragma unroll
      for (int o = 0; o < PM; ++o) x[o] = (o < p) ? vv[o] : 0.0;
      so_chol_solve8<true>(x, Lc, Li);
      double quad = 0.0;
#pragma unroll
      for (int o = 0; o < PM; ++o) quad = fma(x[o], x[o], quad);
      if (mask != 0) ll_sum += -0.5 * (a.cv.ll_terms_step(__popc(mask), p) * LN2PI + logdet + quad);
      if (!(quad == quad)) finite = false;
    }
    for (int i = tid; i < m; i += NT) {
      double acc = av[i];
      for (int o = 0; o < p; ++o) acc = fma(Kg[o * MP + i], vv[o], acc);
      ap[i] = acc;
    }
    __syncthreads();
    // ---- predicted mean a = Az a+ + c ----------------------------------------------------------------------------------------
    so_mean_predict_structured<NT>(AzT, ap, cvec, av, Tss, Xss, Yss, AK, MP * PM, MP, u, lay.s, lay.q);
    if (steady) {
      if (stamp) {
        ph[4] += clock64() - tk;
        ph[6] += 1;
      }
      continue;
    }
    if (stamp) {
      const long long tn = clock64();
      ph[0] += tn - tk;
      tk = tn;
    }
    // ---- P = Az P+ Az' + Qz, P+ = P - 1/2 (K V' + V K') + jitter I with V = P Z' + jitter K.  P+ is never formed:
    //        Az P+ Az' = Az P Az' - 1/2 (AK AV' + AV AK') + jitter Az Az',   AK = Az K,  AV = Az V,
    // so the two products on the matrix core are the plain W' = P Az' (P symmetric: natural stores) and X = W Az'; AK and AV
    // need no pass over Az either: Az P Z' is a combination of 2u rows of W', AK = (Az P Z') F^-1, AV = Az P Z' + jitter AK;
    // the rank-2p term, Qz + jitter Az Az' (a constant, from so_lyap_kernel) and the steady-state test sit in the second
    // product's epilogue, which writes the other P buffer.
    so_gemm<MT>(Pc, MP, AzT, MP, MP, lds, [&](int r, int c, so_v4f64 v) {
#pragma unroll
      for (int e = 0; e < 4; ++e) Wt[(size_t)(r + 4 * e) * MP + c] = v[e];
    });
    __syncthreads();
    if (stamp) {
      const long long tn = clock64();
      ph[2] += tn - tk;
      tk = tn;
    }
    {  // (two groups of 256 threads share the 2u rows of W', as above; PZ is free by now and takes the second group's sums)
      const int g = tid >> 8, i = tid & 255, uh = (u + 1) / 2, cb = g * uh, ce = (cb + uh < u) ? cb + uh : u;
      double x[PM];
#pragma unroll
      for (int o = 0; o < PM; ++o) x[o] = 0.0;
      if (i < m) {
        for (int c0 = cb; c0 < ce; c0 += 12) {
          double wv[24];
#pragma unroll
          for (int e = 0; e < 12; ++e) {
            const int c = c0 + e < ce ? c0 + e : cb;
            wv[e] = Wt[(size_t)c * MP + i];
            wv[12 + e] = Wt[(size_t)(u + c) * MP + i];
          }
#pragma unroll
          for (int e = 0; e < 12; ++e) {
            const double wc = (c0 + e < ce) ? wv[e] + wv[12 + e] : 0.0;
            const int c = c0 + e < ce ? c0 + e : cb;
#pragma unroll
            for (int o = 0; o < PM; ++o)
              if (o < p) x[o] = fma(wc, Zu[o * u + c], x[o]);
          }
        }
        if (g == 1) {
#pragma unroll
          for (int o = 0; o < PM; ++o) PZ[o * MP + i] = x[o];
        }
      }
      __syncthreads();
      if (g == 0 && i < MP) {
        if (i < m) {
#pragma unroll
          for (int o = 0; o < PM; ++o) x[o] = (o < p && ((mask >> o) & 1)) ? x[o] + PZ[o * MP + i] : 0.0;
        }
        double apz[PM];
#pragma unroll
        for (int o = 0; o < PM; ++o) apz[o] = x[o];
        so_chol_solve8<false>(x, Lc, Li);  // (Az P Z') F^-1 row by row
#pragma unroll
        for (int o = 0; o < PM; ++o) {
          AK[o * MP + i] = x[o];
          AV[o * MP + i] = fma(a.cv.jit_V, x[o], apz[o]);
        }
      }
    }
    __syncthreads();
    if (stamp) {
      const long long tn = clock64();
      ph[1] += tn - tk;
      tk = tn;
    }
    // (X = W Az' = Az P Az' is symmetric: only the tiles on and above the diagonal are multiplied, the others are mirrored)
    so_gemm_sym<MT>(Wt, AzT, MP, lds, [&](int r, int c, so_v4f64 v) {
#pragma unroll
      for (int e = 0; e < 4; ++e) Pn[(size_t)(r + 4 * e) * MP + c] = v[e];
    });
    for (int i = tid; i < MP; i += NT) part[i] = (i < m) ? Pc[(size_t)i * MP + i] : 1.0;  // diagonal of the previous P
    __syncthreads();
    // fix-up pass (coalesced, two entries per thread and trip): P_new = X + Qz + jitter Az Az' - 1/2 (AK AV' + AV AK'), and the
    // steady-state test, scale-free: (dP_ij)^2 <= tol^2 P_ii P_jj for every entry -- the blocks of the pruned state differ by
    // orders of magnitude (x_f ~ 1e-4, the products ~ 1e-8), a test against max |P| would only see the largest block (and a
    // component without second-order dynamics has variance exactly 0 in P0: no division)
    double dm = 0.0, pm = 0.0;
    const double tol2 = a.steady_tol * a.steady_tol;
    // (three pairs per trip, their nine global loads requested before any of them is used: the pass is a chain of load
    //  latencies otherwise -- 42 dependent trips per thread)
    constexpr int FX = 3;
    for (int idx0 = tid; idx0 < MP * MP / 2; idx0 += FX * NT) {
      double2 x[FX], qz[FX], po[FX];
#pragma unroll
      for (int f = 0; f < FX; ++f) {
        const int idx = idx0 + f * NT < MP * MP / 2 ? idx0 + f * NT : idx0;
        x[f] = ((const double2*)Pn)[idx];
        qz[f] = ((const double2*)Qzj)[idx];
        po[f] = ((const double2*)Pc)[idx];
      }
#pragma unroll
      for (int f = 0; f < FX; ++f) {
        const int idx = idx0 + f * NT;
        if (idx < MP * MP / 2) {
          const int i = idx / (MP / 2), j = 2 * (idx - i * (MP / 2));
          double c0 = 0.0, c1 = 0.0;
#pragma unroll
          for (int o = 0; o < PM; ++o) {
            const double aki = AK[o * MP + i], avi = AV[o * MP + i];
            const double2 akj = *(const double2*)(AK + o * MP + j), avj = *(const double2*)(AV + o * MP + j);
            c0 = fma(aki, avj.x, fma(avi, akj.x, c0));
            c1 = fma(aki, avj.y, fma(avi, akj.y, c1));
          }
          double2 xn;
          xn.x = fma(-0.5, c0, x[f].x) + qz[f].x;
          xn.y = fma(-0.5, c1, x[f].y) + qz[f].y;
          ((double2*)Pn)[idx] = xn;
          const double di = part[i], d0 = xn.x - po[f].x, d1 = xn.y - po[f].y;
          dm = fmax(dm, (d0 * d0 > tol2 * di * part[j] || d1 * d1 > tol2 * di * part[j + 1]) ? 1.0 : 0.0);  // 1 = still moving
          pm = nanmax(pm, nanmax(fabs(xn.x), fabs(xn.y)));
        }
      }
    }
    {
      double* t = Pc;
      Pc = Pn;
      Pn = t;
    }
    if (stamp) {
      ph[3] += clock64() - tk;
      ph[5] += 1;
    }
    const double dmax = so_wg_max<NT>(dm, red), pmax = so_wg_max<NT>(pm, red);
    if (!(pmax < 1e300)) finite = false;  // (uniform; thread 0 keeps the flag that matters; NaN / overflow of P ends here)
    if (a.steady_tol > 0.0 && dmax == 0.0 && pmax < 1e300) {
      steady = true;
      steady_mask = mask;
      if (steady_at < 0) steady_at = t + 1;
    }
  }
  if (tid == 0) {
    const bool okf = finite && (ll_sum == ll_sum) && fabs(ll_sum) < 1e300;
    a.logp[draw] = okf ? ll_sum : -INFINITY;
    if (!okf) a.status[draw] |= DSGE_ST_FILTER_NONFINITE;
    if (a.steady_at) a.steady_at[draw] = steady_at;
    if (stamp) {
      ph[7] = clock64() - t_begin;
      for (int e = 0; e < 8; ++e) a.phases[e] = ph[e];
    }
  }
}

}  // namespace dsge

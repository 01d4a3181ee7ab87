// GPU box: the prediction products of one filter step -- W = P[S,S] Tc' (stored transposed), X = Tc W, P = sym(X) + Q -- on the VALU
// (mm_nt of csrc/dsge_kalman_nt.hpp, the round-2 form) against the 4 x 4 x 4 FP64 matrix instruction (csrc/dsge_mfma4.hpp):
// cycles per step for a lone wavefront and the time of a full-chip launch, and the difference of the results.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/mfma_probe/product_probe tools/mfma_probe/product_probe.hip
#include "../../geconpy_amd/csrc/dsge_kalman_nt.hpp"
#include "../../geconpy_amd/csrc/dsge_mfma4.hpp"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace dsge;
constexpr int BS = 3, NP = 24, SK = 20, LDK = SK + 2;

template <int MODE>
__global__ __launch_bounds__(64, 2) void step_kernel(const double* Tin, const double* Pin, const double* Qin, double* Pout,
                                                     long long* cyc, int s, int m, int iters) {
  __shared__ __attribute__((aligned(16))) double Tc[NP * LDK], Wt[NP * LDK], Pc[NP * LDK];
  const int lane = threadIdx.x, lr = lane >> 3, lc = lane & 7;
  for (int idx = lane; idx < NP * LDK; idx += 64) {
    Tc[idx] = 0.0;
    Wt[idx] = 0.0;
    Pc[idx] = 0.0;
  }
  __syncthreads();
  double Pb[BS][BS], Qb[BS][BS];
  for (int i = 0; i < BS; ++i)
    for (int j = 0; j < BS; ++j) {
      const int r = lr * BS + i, c = lc * BS + j;
      const bool in = r < m && c < m;
      if (in && c < s) Tc[r * LDK + c] = Tin[r * m + c];
      Pb[i][j] = in ? Pin[r * m + c] : 0.0;
      Qb[i][j] = in ? Qin[r * m + c] : 0.0;
    }
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (lr * BS < s && lc * BS < s) blk_store_lds<BS>(Pb, Pc, LDK, lr, lc);
    wave_sync();
    if (MODE == 0) {
      if (lr * BS < s) {
        double Wb[BS][BS];
        blk_zero<BS>(Wb);
        mm_nt<BS, LDK>(Wb, Pc, Tc, s, lr, lc);
        for (int i = 0; i < BS; ++i)
          for (int j = 0; j < BS; ++j) Wt[(lc * BS + j) * LDK + lr * BS + i] = Wb[i][j];
      }
      wave_sync();
      double Xb[BS][BS];
      blk_zero<BS>(Xb);
      mm_nt<BS, LDK>(Xb, Tc, Wt, s, lr, lc);
      const int src = (lc << 3) | lr;
      for (int i = 0; i < BS; ++i)
        for (int j = 0; j < BS; ++j) {
          const double xt = __shfl(Xb[j][i], src, 64);
          Pb[i][j] = 0.5 * (Xb[i][j] + xt) + Qb[i][j];
        }
    } else {
      // W[i][j] = sum_k Pc[i][k] Tc[j][k], i < s, j < m: 5 x 5 tiles, stored transposed.  Stores are UNCONDITIONAL: a block that
      // is no tile writes to a padding slot nothing reads (a predicated store is a branch, and a branch ends the basic block the
      // scheduler can overlap loads and issues in)
      constexpr int KT = 5, TM = 5, DUMP = (NP - 1) * LDK + LDK - 1;
      using MW = Mfma4Map<KT, TM>;
      const int blk = (lane >> 2) & 3, i4 = lane & 3, kq = lane >> 4;
      mfma4_nt<KT, KT, TM, LDK>(Pc, Tc, lane, [&](int g, double d) {
        const int at = (4 * MW::tb(g, blk) + i4) * LDK + 4 * MW::ta(g, blk) + kq;
        Wt[MW::live(g, blk) ? at : DUMP] = d;
      });
      wave_sync();
      // X[r][c] = sum_k Tc[r][k] Wt[c][k]: the 15 upper tiles -> registers first (Wt is an operand), then over Wt
      using UX = Mfma4Upper<TM>;
      double dd[UX::NG];
      int pos[UX::NG];
      mfma4_nt_upper<KT, TM, LDK>(Tc, Wt, lane, [&](int g, double d, int ta, int tb, bool live) {
        dd[g] = d;
        pos[g] = live ? (4 * ta + kq) * LDK + 4 * tb + i4 : DUMP;
      });
      wave_sync();
#pragma unroll
      for (int g = 0; g < UX::NG; ++g) Wt[pos[g]] = dd[g];
      wave_sync();
      for (int i = 0; i < BS; ++i)
        for (int j = 0; j < BS; ++j) {
          const int r = lr * BS + i, c = lc * BS + j;
          const int lo = r < c ? r : c, hi = r < c ? c : r;
          const bool in = hi < m;
          const double xv = Wt[in ? lo * LDK + hi : 0];
          Pb[i][j] = (in ? xv : 0.0) + Qb[i][j];
        }
      wave_sync();
    }
  }
  const long long t1 = clock64();
  if (blockIdx.x == 0) {
    for (int i = 0; i < BS; ++i)
      for (int j = 0; j < BS; ++j) {
        const int r = lr * BS + i, c = lc * BS + j;
        if (r < m && c < m) Pout[r * m + c] = Pb[i][j];
      }
    if (lane == 0) cyc[0] = t1 - t0;
  }
}

int main() {
  const int s = 18, m = 18, iters = 200;
  std::vector<double> T(m * m), P(m * m), Q(m * m);
  unsigned long long x = 88172645463325252ull;
  auto rnd = [&]() {
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    return (double)(x % 2000001) / 1e6 - 1.0;
  };
  for (int i = 0; i < m * m; ++i) T[i] = 0.2 * rnd();
  for (int i = 0; i < m; ++i)
    for (int j = 0; j <= i; ++j) {
      const double v = (i == j) ? 1.0 + 0.1 * rnd() : 0.05 * rnd();
      P[i * m + j] = P[j * m + i] = v;
      Q[i * m + j] = Q[j * m + i] = 0.01 * v;
    }
  double *dT, *dP, *dQ, *dO;
  long long* dC;
  hipMalloc(&dT, m * m * 8);
  hipMalloc(&dP, m * m * 8);
  hipMalloc(&dQ, m * m * 8);
  hipMalloc(&dO, m * m * 8);
  hipMalloc(&dC, 8);
  hipMemcpy(dT, T.data(), m * m * 8, hipMemcpyHostToDevice);
  hipMemcpy(dP, P.data(), m * m * 8, hipMemcpyHostToDevice);
  hipMemcpy(dQ, Q.data(), m * m * 8, hipMemcpyHostToDevice);
  std::vector<double> out[2] = {std::vector<double>(m * m), std::vector<double>(m * m)};
  for (int mode = 0; mode < 2; ++mode)
    for (int grid : {1, 2048, 4096}) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        if (mode == 0)
          hipLaunchKernelGGL(step_kernel<0>, dim3(grid), dim3(64), 0, 0, dT, dP, dQ, dO, dC, s, m, iters);
        else
          hipLaunchKernelGGL(step_kernel<1>, dim3(grid), dim3(64), 0, 0, dT, dP, dQ, dO, dC, s, m, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
      }
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      long long c;
      hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost);
      hipMemcpy(out[mode].data(), dO, m * m * 8, hipMemcpyDeviceToHost);
      printf("%s grid %4d: %7.1f cycles per step (workgroup 0), launch %.3f ms = %.3f us per step and workgroup-slot\n",
             mode ? "mfma 4x4x4" : "valu mm_nt ", grid, (double)c / iters, ms, ms * 1e3 / iters);
    }
  double err = 0, mx = 0;
  for (int i = 0; i < m * m; ++i) {
    err = fmax(err, fabs(out[0][i] - out[1][i]));
    mx = fmax(mx, fabs(out[0][i]));
  }
  printf("max |P_valu - P_mfma| after %d steps = %.3e (max |P| %.3e)\n", iters, err, mx);
  return 0;
}

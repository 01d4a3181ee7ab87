// Experiment / self-test (not product): the workgroup MFMA product of dsge_so_gemm.hpp against the host, and its rate.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -o tools/so_gemm_probe/probe13 tools/so_gemm_probe/probe.hip   (from the repo root)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "../../geconpy_amd/csrc/dsge_so_gemm.hpp"
#ifndef MT_
#define MT_ 13
#endif
using Cfg = dsge::SoGemmCfg<MT_>;
template <int EPI>
__global__ __launch_bounds__(512) void k(const double* A, const double* B, double* C, double* Ct, int K, int reps, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int MP = Cfg::MP;
  const size_t off = (size_t)blockIdx.x * MP * MP;
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r)
    dsge::so_gemm<MT_>(A + off, MP, B + off, MP, K, lds, [&](int row0, int col, dsge::so_v4f64 v) {
      for (int q = 0; q < 4; ++q) {
        if (EPI & 1) C[off + (size_t)(row0 + 4 * q) * MP + col] = v[q];
        if (EPI & 2) Ct[off + (size_t)col * MP + row0 + 4 * q] = v[q];
        if (EPI == 0 && v[q] == 1.2345e300) C[off] = v[q];
      }
    });
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = clock64() - t0;
}
// symmetric product: Bop = Aop (a Gram matrix): the mirrored half must equal the computed one
__global__ __launch_bounds__(512) void ksym(const double* A, double* C, int K, int reps, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int MP = Cfg::MP;
  const size_t off = (size_t)blockIdx.x * MP * MP;
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r)
    dsge::so_gemm_sym<MT_>(A + off, A + off, K, lds, [&](int row0, int col, dsge::so_v4f64 v) {
      for (int q = 0; q < 4; ++q) C[off + (size_t)(row0 + 4 * q) * MP + col] = v[q];
    });
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = clock64() - t0;
}
__global__ __launch_bounds__(512) void ksym2(const double* A, const double* B, double* C, int K) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int MP = Cfg::MP;
  const size_t off = (size_t)blockIdx.x * MP * MP;
  dsge::so_gemm_sym<MT_>(A + off, B + off, K, lds, [&](int row0, int col, dsge::so_v4f64 v) {
    for (int q = 0; q < 4; ++q) C[off + (size_t)(row0 + 4 * q) * MP + col] = v[q];
  });
}
int main(int argc, char** argv) {
  constexpr int MP = Cfg::MP;
  const int nb = argc > 1 ? atoi(argv[1]) : 1024, reps = argc > 2 ? atoi(argv[2]) : 10, K = MP;
  const size_t mat = (size_t)MP * MP;
  std::vector<double> hA(mat * 2), hB(mat * 2), hC(mat * 2), hCt(mat * 2);
  srand(1);
  for (auto& x : hA) x = rand() / (double)RAND_MAX - 0.5;
  for (auto& x : hB) x = rand() / (double)RAND_MAX - 0.3;
  double *A, *B, *C, *Ct;
  hipMalloc(&A, nb * mat * 8); hipMalloc(&B, nb * mat * 8); hipMalloc(&C, nb * mat * 8); hipMalloc(&Ct, nb * mat * 8);
  for (int i = 0; i < nb; ++i) {
    hipMemcpy(A + i * mat, hA.data() + (i & 1) * mat, mat * 8, hipMemcpyHostToDevice);
    hipMemcpy(B + i * mat, hB.data() + (i & 1) * mat, mat * 8, hipMemcpyHostToDevice);
  }
  const size_t lds = Cfg::LDS_DOUBLES * 8;
  long long* cyc; hipMalloc(&cyc, 8);
  hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute((const void*)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k<3>, dim3(nb), dim3(512), lds, 0, A, B, C, Ct, K, 1, cyc);
  hipDeviceSynchronize();
  double err = 0, errt = 0;
  for (int d : {0, 1, nb - 1}) {
    hipMemcpy(hC.data(), C + d * mat, mat * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hCt.data(), Ct + d * mat, mat * 8, hipMemcpyDeviceToHost);
    const double* a = hA.data() + (d & 1) * mat; const double* b = hB.data() + (d & 1) * mat;
    for (int i = 0; i < MP; i += 3) for (int j = 0; j < MP; ++j) {
      double s = 0; for (int kk = 0; kk < K; ++kk) s += a[kk * MP + i] * b[kk * MP + j];
      err = fmax(err, fabs(s - hC[i * MP + j])); errt = fmax(errt, fabs(s - hCt[j * MP + i]));
    }
  }
  printf("MT=%d MP=%d lds=%zu B: max abs err %.3e (transposed store %.3e)\n", MT_, MP, lds, err, errt);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int epi = 0; epi < 4; ++epi)
  for (int nbb : {256, nb}) {
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      if (epi == 0) hipLaunchKernelGGL(k<0>, dim3(nbb), dim3(512), lds, 0, A, B, C, Ct, K, reps, cyc);
      if (epi == 1) hipLaunchKernelGGL(k<1>, dim3(nbb), dim3(512), lds, 0, A, B, C, Ct, K, reps, cyc);
      if (epi == 2) hipLaunchKernelGGL(k<2>, dim3(nbb), dim3(512), lds, 0, A, B, C, Ct, K, reps, cyc);
      if (epi == 3) hipLaunchKernelGGL(k<3>, dim3(nbb), dim3(512), lds, 0, A, B, C, Ct, K, reps, cyc);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    const double fl = 2.0 * MP * MP * K * (double)nbb * reps;
    printf("  epilogue %d (1 = natural store, 2 = transposed store), %d draws x %d products: %.3f ms = %.2f TFLOP/s, %.1f us per product per CU slot; block 0: %.0f cycles per product (ideal MFMA %d)\n",
           epi, nbb, reps, best, fl / best / 1e9, best * 1e3 / reps / ((nbb + 255) / 256), (double)hc / reps, 98 * 64 * (K / 8));
  }
  // ---- symmetric variant
  hipFuncSetAttribute((const void*)ksym, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipMemset(C, 0xff, nb * mat * 8);
  hipLaunchKernelGGL(ksym, dim3(nb), dim3(512), lds, 0, A, C, K, 1, cyc);
  hipDeviceSynchronize();
  double errs = 0, asym = 0;
  for (int d : {0, 1, nb - 1}) {
    hipMemcpy(hC.data(), C + d * mat, mat * 8, hipMemcpyDeviceToHost);
    const double* a = hA.data() + (d & 1) * mat;
    for (int i = 0; i < MP; i += 3) for (int j = 0; j < MP; ++j) {
      double s = 0; for (int kk = 0; kk < K; ++kk) s += a[kk * MP + i] * a[kk * MP + j];
      errs = fmax(errs, fabs(s - hC[i * MP + j]));
    }
    for (int i = 0; i < MP; ++i) for (int j = 0; j < MP; ++j) asym = fmax(asym, fabs(hC[i * MP + j] - hC[j * MP + i]));
  }
  printf("symmetric product: max abs err %.3e, asymmetry %.3e\n", errs, asym);
  {  // distinct operands: tiles on and above the diagonal against the host, the others against the transposed host value
    hipFuncSetAttribute((const void*)ksym2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipMemset(C, 0xff, nb * mat * 8);
    hipLaunchKernelGGL(ksym2, dim3(nb), dim3(512), lds, 0, A, B, C, K);
    hipDeviceSynchronize();
    double e2 = 0;
    for (int d : {0, 1}) {
      hipMemcpy(hC.data(), C + d * mat, mat * 8, hipMemcpyDeviceToHost);
      const double* a = hA.data() + (d & 1) * mat; const double* b = hB.data() + (d & 1) * mat;
      for (int i = 0; i < MP; ++i) for (int j = 0; j < MP; ++j) {
        const bool up = (i / 16 < j / 16) || (i / 16 == j / 16 && i <= j);
        const int ii = up ? i : j, jj = up ? j : i;
        double s = 0; for (int kk = 0; kk < K; ++kk) s += a[kk * MP + ii] * b[kk * MP + jj];
        e2 = fmax(e2, fabs(s - hC[i * MP + j]));
      }
    }
    printf("symmetric routine on distinct operands (upper tiles + their mirrors): max abs err %.3e\n", e2);
    if (!(e2 < 1e-10)) return 1;
  }
  for (int nbb : {256, nb}) {
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(ksym, dim3(nbb), dim3(512), lds, 0, A, C, K, reps, cyc);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("  symmetric, %d draws x %d products: %.3f ms, %.1f us per product per CU slot; block 0: %.0f cycles per product\n",
           nbb, reps, best, best * 1e3 / reps / ((nbb + 255) / 256), (double)hc / reps);
  }
  return err < 1e-10 && errt < 1e-10 && errs < 1e-10 && asym == 0.0 ? 0 : 1;
}

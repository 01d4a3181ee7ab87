// Experiment (not product): cr_wide_kernel variants timed alone on 4096 systems read from a file.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/dsge_hip.h"
#include CRW_HEADER
int main(int argc, char** argv) {
  const int n = atoi(argv[2]), nb = 4096, ndist = 64;
  std::vector<double> h(3 * (size_t)ndist * n * n);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(h.data(), 8, h.size(), f) != h.size()) return 1;
  fclose(f);
  double *A, *B, *C, *T;
  int32_t *st, *it;
  const size_t mat = (size_t)nb * n * n * 8;
  hipMalloc(&A, mat); hipMalloc(&B, mat); hipMalloc(&C, mat); hipMalloc(&T, mat);
  hipMalloc(&st, nb * 4); hipMalloc(&it, nb * 4);
  for (int r = 0; r < nb / ndist; ++r) {
    hipMemcpy((char*)A + (size_t)r * ndist * n * n * 8, h.data(), (size_t)ndist * n * n * 8, hipMemcpyHostToDevice);
    hipMemcpy((char*)B + (size_t)r * ndist * n * n * 8, h.data() + (size_t)ndist * n * n, (size_t)ndist * n * n * 8, hipMemcpyHostToDevice);
    hipMemcpy((char*)C + (size_t)r * ndist * n * n * 8, h.data() + 2 * (size_t)ndist * n * n, (size_t)ndist * n * n * 8, hipMemcpyHostToDevice);
  }
  hipFuncSetAttribute((const void*)dsge::cr_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dsge::CrwSmem::bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    float best = 1e9;
    for (int k = 0; k < 12; ++k) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(dsge::cr_wide_kernel, dim3(nb), dim3(256), dsge::CrwSmem::bytes, 0, A, B, C, nb, n, 1000, 1e-8, T, st, it, 0,
                         (const double*)nullptr, 0, (double*)nullptr);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (k >= 2 && ms < best) best = ms;
    }
    std::vector<int32_t> hs(nb);
    hipMemcpy(hs.data(), st, nb * 4, hipMemcpyDeviceToHost);
    int bad = 0; for (int x : hs) bad += (x != 0);
    printf("%s n=%d: best %.4f ms, status!=0: %d\n", CRW_NAME, n, best, bad);
  }
  return 0;
}

// Layout probe for v_mfma_f64_16x16x4_f64 on gfx950: D = A(16x4) B(4x16) with asymmetric operands.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
__global__ void probe(const double* A, const double* B, double* D) {
  const int l = threadIdx.x;
  const double a = A[(l & 15) * 4 + (l >> 4)];   // A[i = l&15][k = l>>4]
  const double b = B[(l >> 4) * 16 + (l & 15)];  // B[k = l>>4][j = l&15]
  v4f64 c = {0.0, 0.0, 0.0, 0.0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];  // row = (l>>4) + 4 r, col = l&15
}
int main() {
  double hA[64], hB[64], hD[256], ref[256];
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) hA[i * 4 + k] = 1.0 + i * 0.37 - k * 1.13 + (i * k) * 0.01;
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = 0.5 - k * 0.71 + j * 0.29 + (k * j * j) * 0.003;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
  double *dA, *dB, *dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  double err = 0; for (int i = 0; i < 256; ++i) { double e = hD[i] - ref[i]; if (e < 0) e = -e; if (e > err) err = e; }
  printf("max abs err %.3e  (D[3][7] = %.6f ref %.6f)\n", err, hD[3 * 16 + 7], ref[3 * 16 + 7]);
  return err < 1e-12 ? 0 : 1;
}

// GPU box: operand / result layout of v_mfma_f64_4x4x4f64 (4 blocks of 4x4x4) on gfx950, discovered with one-hot operands:
// a = [lane == la], b = [lane == lb], c = 0  ->  which lane of d is 1?  Prints, per lane la of A, the lanes lb of B it meets and the
// output lane.   hipcc --offload-arch=gfx950 -O2 -o tools/mfma_probe/mfma4_layout tools/mfma_probe/mfma4_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* out) {  // out[la * 64 + lb] = output lane or -1
  const int l = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      const unsigned long long hit = __ballot(d != 0.0);
      if (l == 0) out[la * 64 + lb] = hit ? (int)__builtin_ctzll(hit) + 64 * (__builtin_popcountll(hit) - 1) : -1;
    }
}
int main() {
  int* d;
  static int h[4096];
  hipMalloc(&d, sizeof h);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb)
      if (h[la * 64 + lb] >= 0) printf("  B%2d->D%2d", lb, h[la * 64 + lb]);
    printf("\n");
  }
  return 0;
}

// GPU box: issue rate and latency of the FP64 matrix instructions on gfx950, against v_fma_f64.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma4_probe tools/probes/mfma4_probe.hip && tools/probes/mfma4_probe
// Prints cycles per instruction for (a) a dependent chain, (b) 8 independent accumulators, with 1 / 2 / 4 wavefronts per SIMD
// (one workgroup of 256 / 512 / 1024 threads on one CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int N = 4096;

template <int MODE, int IND>
__global__ void probe(double* out, long long* cyc, double seed) {
  double a = seed + threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  double acc[IND];
  v4d acc4[IND];
#pragma unroll
  for (int i = 0; i < IND; ++i) {
    acc[i] = i;
    acc4[i] = v4d{(double)i, 0, 0, 0};
  }
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < N / IND; ++it) {
#pragma unroll
    for (int i = 0; i < IND; ++i) {
      if (MODE == 0) acc[i] = __builtin_fma(a, b, acc[i]);
      if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
      if (MODE == 2) acc4[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc4[i], 0, 0, 0);
    }
  }
  const long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < IND; ++i) s += acc[i] + acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// mixed: MFMA 4x4x4 stream + independent VALU FMA stream in the same wave (do the two pipes overlap for ONE wave?)
template <int NM, int NV>
__global__ void mixed(double* out, long long* cyc, double seed) {
  double a = seed + threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  double m[4] = {0, 1, 2, 3}, v[8] = {0, 1, 2, 3, 4, 5, 6, 7};
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < N / 4; ++it) {
#pragma unroll
    for (int i = 0; i < NM; ++i) m[i & 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m[i & 3], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i & 7] = __builtin_fma(a, b, v[i & 7]);
  }
  const long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 4; ++i) s += m[i];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class K>
void run(const char* name, K kern, int threads, int per_iter) {
  double* out;
  long long* cyc;
  hipMalloc(&out, 8 * 1024 * 8);
  hipMalloc(&cyc, 64);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, cyc, 1.0);
  hipDeviceSynchronize();
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  // clock64 = s_memtime at 100 MHz on gfx9?  report raw and per-instruction; the wall clock scale is printed by the fma line
  printf("%-44s threads %4d: %8lld ticks, %.3f ticks per instruction-slot\n", name, threads, c, (double)c / per_iter);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  for (int threads : {64, 256, 512, 1024}) {
    run("v_fma_f64 dependent", probe<0, 1>, threads, N);
    run("v_fma_f64 8 independent", probe<0, 8>, threads, N);
    run("mfma_f64_4x4x4 dependent", probe<1, 1>, threads, N);
    run("mfma_f64_4x4x4 8 independent", probe<1, 8>, threads, N);
    run("mfma_f64_16x16x4 dependent", probe<2, 1>, threads, N);
    run("mfma_f64_16x16x4 8 independent", probe<2, 8>, threads, N);
    run("mixed 4 mfma4 + 0 fma per iter", mixed<4, 0>, threads, N / 4);
    run("mixed 4 mfma4 + 8 fma per iter", mixed<4, 8>, threads, N / 4);
    run("mixed 4 mfma4 + 16 fma per iter", mixed<4, 16>, threads, N / 4);
    run("mixed 0 mfma4 + 16 fma per iter", mixed<0, 16>, threads, N / 4);
  }
  return 0;
}

// Latency / issue probe for the FP64 wave-level primitives the kernels are built from (gfx950, one wave alone on a CU).
//   hipcc --offload-arch=gfx950 -O3 -o probe probe.hip && ./probe
// Every figure is shader cycles (clock64) per operation of an unrolled sequence of REP operations, measured once with
// each operation depending on the previous one ("dep") and once as independent streams ("indep").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
#define KEEP(x) asm volatile("" : "+v"(x))

__device__ __forceinline__ double readlane_f64(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

__global__ void probe(double* out, long long* cyc, const double* in) {
  __shared__ double lds[1024];
  const int lane = threadIdx.x;
  double a = in[lane], b = in[64 + lane];
  for (int i = lane; i < 1024; i += 64) lds[i] = in[i & 127] * 1e-3;
  __syncthreads();
  int k = 0;
  long long t0, t1;
  double x = a, x2 = b, x3 = a + b, x4 = a - b, x5 = a * 1.5, x6 = b * 1.5, x7 = a * 2.5, x8 = b * 2.5;
#define BEGIN() t0 = clock64()
#define END()                                   \
  t1 = clock64();                               \
  if (lane == 0) cyc[k] = t1 - t0;              \
  ++k
  // 0: dependent fma
  BEGIN();
#pragma unroll
  for (int i = 0; i < REP; ++i) { x = fma(x, a, b); KEEP(x); }
  END();
  // 1: 8 independent fma chains (REP total)
  BEGIN();
#pragma unroll
  for (int i = 0; i < REP / 8; ++i) {
    x = fma(x, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b); x4 = fma(x4, a, b);
    x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b); x8 = fma(x8, a, b);
    KEEP(x); KEEP(x2); KEEP(x3); KEEP(x4); KEEP(x5); KEEP(x6); KEEP(x7); KEEP(x8);
  }
  END();
  // 2: dependent mul, 3: dependent add
  BEGIN();
#pragma unroll
  for (int i = 0; i < REP; ++i) { x = x * a; KEEP(x); }
  END();
  BEGIN();
#pragma unroll
  for (int i = 0; i < REP; ++i) { x = x + b; KEEP(x); }
  END();
  // 4: dependent rcp
  BEGIN();
#pragma unroll
  for (int i = 0; i < REP; ++i) { x = __builtin_amdgcn_rcp(x); KEEP(x); }
  END();
  // 5: dependent readlane_f64 -> fma
  BEGIN();
#pragma unroll
  for (int i = 0; i < REP; ++i) { x = fma(readlane_f64(x, i & 63), a, b); KEEP(x); }
  END();
  // 6: independent readlane_f64 (same source) each feeding an fma of one of two chains
  BEGIN();
#pragma unroll
  for (int i = 0; i < REP; i += 2) {
    x2 = fma(readlane_f64(x, i), a, x2);
    x3 = fma(readlane_f64(x, i + 1), a, x3);
  }
  KEEP(x2); KEEP(x3);
  END();
  // 7: dependent ds_bpermute (f64 = two b32)
  BEGIN();
#pragma unroll
  for (int i = 0; i < REP; ++i) { x = __shfl(x, (lane + 1) & 63, 64); KEEP(x); }
  END();
  // 8: 8 independent bpermutes per round, REP/8 dependent rounds
  BEGIN();
#pragma unroll
  for (int i = 0; i < REP / 8; ++i) {
    x = __shfl(x, (lane + 1) & 63, 64); x2 = __shfl(x2, (lane + 2) & 63, 64); x3 = __shfl(x3, (lane + 3) & 63, 64);
    x4 = __shfl(x4, (lane + 4) & 63, 64); x5 = __shfl(x5, (lane + 5) & 63, 64); x6 = __shfl(x6, (lane + 6) & 63, 64);
    x7 = __shfl(x7, (lane + 7) & 63, 64); x8 = __shfl(x8, (lane + 8) & 63, 64);
    KEEP(x); KEEP(x2); KEEP(x3); KEEP(x4); KEEP(x5); KEEP(x6); KEEP(x7); KEEP(x8);
  }
  END();
  // 9: dependent DPP move (row_shr:1) + add
  BEGIN();
#pragma unroll
  for (int i = 0; i < REP; ++i) { x = x + dpp_f64<0x111>(x); KEEP(x); }
  END();
  // 10: dependent LDS read b64 (address from the value)
  {
    int idx = lane;
    BEGIN();
#pragma unroll
    for (int i = 0; i < REP; ++i) {
      const double v = lds[idx & 1023];
      idx = (idx + 7 + (__double2loint(v) & 1)) & 1023;
      asm volatile("" : "+v"(idx));
    }
    END();
    x += (double)idx;
  }
  // 11: LDS write -> read round trip (dependent): write x, sync, read neighbour
  BEGIN();
#pragma unroll
  for (int i = 0; i < REP; ++i) {
    lds[lane] = x;
    __syncthreads();
    x = lds[(lane + 1) & 63] + 1.0;
    __syncthreads();
    KEEP(x);
  }
  END();
  // 12: 12 ds_read_b128 (broadcast address) then use, dependent rounds (REP/4 rounds)
  {
    const double2* l2 = reinterpret_cast<const double2*>(lds);
    int base = 0;
    BEGIN();
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) {
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < 12; ++q) { const double2 v = l2[base + q]; s += v.x + v.y; }
      base = (base + (__double2loint(s) & 3)) & 255;
      asm volatile("" : "+v"(base));
    }
    END();
    x += (double)base;
  }
  // 13: dependent v_cndmask pair (f64 select)
  BEGIN();
#pragma unroll
  for (int i = 0; i < REP; ++i) { x = (lane & 1) ? x : x2; KEEP(x); x2 = (lane & 2) ? x2 : x; KEEP(x2); }
  END();
  // 14: ballot -> branchless uniform use
  {
    unsigned long long m = 0;
    BEGIN();
#pragma unroll
    for (int i = 0; i < REP; ++i) { m += __ballot(x > (double)(m & 7)); }
    END();
    x += (double)(m & 15);
  }
  // 15: wave max via 6 DPP steps on u64 (as wave_max_u64) -- one reduction = 1 op here, REP/8 dependent reductions
  BEGIN();
#pragma unroll
  for (int i = 0; i < REP / 8; ++i) {
    unsigned long long v = (unsigned long long)__double_as_longlong(x);
#define STEP(C, R)                                                                                        \
  {                                                                                                       \
    int lo = (int)(v & 0xffffffffull), hi = (int)(v >> 32);                                                \
    lo = __builtin_amdgcn_update_dpp(0, lo, C, R, 0xf, false);                                             \
    hi = __builtin_amdgcn_update_dpp(0, hi, C, R, 0xf, false);                                             \
    const unsigned long long t = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;                  \
    v = t > v ? t : v;                                                                                     \
  }
    STEP(0x111, 0xf) STEP(0x112, 0xf) STEP(0x114, 0xf) STEP(0x118, 0xf) STEP(0x142, 0xa) STEP(0x143, 0xc)
    int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffull), 63), hi = __builtin_amdgcn_readlane((int)(v >> 32), 63);
    x = __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo)) + 1.0;
    KEEP(x);
  }
  END();
  // 16: 36 independent fma (a 3x3 block x 4 k) fed by 12 ds_read_b128: one mm_nt stage, REP/8 dependent-free stages
  {
    const double2* l2 = reinterpret_cast<const double2*>(lds);
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    BEGIN();
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      double2 av[3][2], bv[3][2];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        av[i][0] = l2[(lane >> 3) * 39 + i * 13 + 2 * st]; av[i][1] = l2[(lane >> 3) * 39 + i * 13 + 2 * st + 1];
        bv[i][0] = l2[(lane & 7) * 39 + i * 13 + 2 * st]; bv[i][1] = l2[(lane & 7) * 39 + i * 13 + 2 * st + 1];
      }
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          acc[i * 3 + j] = fma(av[i][0].x, bv[j][0].x, acc[i * 3 + j]);
          acc[i * 3 + j] = fma(av[i][0].y, bv[j][0].y, acc[i * 3 + j]);
          acc[i * 3 + j] = fma(av[i][1].x, bv[j][1].x, acc[i * 3 + j]);
          acc[i * 3 + j] = fma(av[i][1].y, bv[j][1].y, acc[i * 3 + j]);
        }
    }
    END();
    for (int i = 0; i < 9; ++i) x += acc[i];
  }
  out[lane] = x + x2 + x3 + x4 + x5 + x6 + x7 + x8;
  if (lane == 0) cyc[63] = k;
}

int main() {
  double *din, *dout;
  long long* dc;
  std::vector<double> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 1.0 + 1e-3 * (i % 17);
  hipMalloc(&din, 1024 * 8); hipMalloc(&dout, 64 * 8); hipMalloc(&dc, 64 * 8);
  hipMemcpy(din, h.data(), 1024 * 8, hipMemcpyHostToDevice);
  long long c[64];
  const char* names[] = {"fma_f64 dependent", "fma_f64 8 independent chains", "mul_f64 dependent", "add_f64 dependent",
                         "rcp_f64 dependent", "readlane_f64 -> fma dependent", "readlane_f64 independent + fma (2 chains)",
                         "ds_bpermute f64 dependent", "ds_bpermute f64, 8 in flight", "dpp row_shr + add dependent",
                         "ds_read_b64 dependent (pointer chase)", "LDS write -> sync -> read round trip",
                         "round of 12 broadcast ds_read_b128 + reduce (per round of 4 counted ops)", "v_cndmask f64 x2 dependent",
                         "ballot -> scalar use dependent", "wave_max_u64 (per reduction / 8 counted ops)",
                         "mm_nt stage: 12 ds_read_b128 + 36 fma (per 8 counted = one stage)"};
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dout, dc, din);
    hipDeviceSynchronize();
  }
  hipMemcpy(c, dc, 64 * 8, hipMemcpyDeviceToHost);
  for (int i = 0; i < (int)c[63]; ++i) printf("%-80s %8.1f cycles/op  (total %lld)\n", names[i], (double)c[i] / REP, c[i]);
  return 0;
}

// Workgroup-level FP64 matrix product on the matrix core, for the second-order path (dsge_second_order.hpp):
//
//     C (M x N) = Aop' Bop,      Aop : [K][lda] (row k holds column k of the LEFT factor),   Bop : [K][ldb]
//
// i.e. both operands arrive "k-major" -- every row of either array is one rank-1 term -- so that the global loads are
// contiguous rows, the LDS layout is the same for both and the v_mfma_f64_16x16x4_f64 fragments (lane l: A[i = l & 15]
// [k = l >> 4], B[k = l >> 4][j = l & 15]; probed in tools/mfma_probe) are read as 16 consecutive doubles per k: with a row
// stride of 16 (mod 32) doubles the two k's a half-wave touches fall on complementary bank halves (conflict-free
// ds_read_b64).  One workgroup of 512 threads = 8 wavefronts = two per SIMD owns the whole product of ONE draw.  The output
// is cut into 2 x 2 blocks of RB x RB tiles of 16 (RB = ceil(MT / 2); 7 at MT = 13) and every block is shared by the two
// wavefronts of one SIMD: wavefront `half` accumulates tiles [25 half, 25 half + 25) of the block's row-major tile list in
// its accumulation registers (25 tiles x 8 registers = 200 of the 256 a wavefront has at two per SIMD).  [First version: one
// wavefront per block with 49 tiles = 392 registers: hipcc keeps MFMA accumulators in the 256 AGPRs only and shuttled the
// other 17 tiles through v_accvgpr_read/write -- 840 instructions and 45 scratch accesses per 98 MFMAs.]  The K dimension
// streams through LDS in chunks of KC rows, double-buffered: the global loads of chunk c + 1 are in flight while chunk c is
// multiplied, one workgroup barrier per chunk.  The pruned second-order filter has a 207-dimensional state -- 13 x 13 tiles of
// 16 fill the fragments: the one place on the path where FP64 MFMA is the right tool.
#pragma once
#include <hip/hip_runtime.h>

namespace dsge {

typedef double so_v4f64 __attribute__((ext_vector_type(4)));

constexpr int SO_THREADS = 512;

template <int MT>
struct SoGemmCfg {
  static constexpr int MP = 16 * MT;                         // padded matrix dimension
  static constexpr int LDSROW = MP + ((MT & 1) ? 0 : 16);    // == 16 (mod 32)
  static constexpr int KC = 8;                               // rows of Aop / Bop per stage
  static constexpr int RB = (MT + 1) / 2;                    // tiles per block side (the larger block)
  static constexpr int NT0 = (RB * RB + 1) / 2;              // tiles of a block the first wavefront of the pair takes
  static constexpr int STAGE = 2 * KC * LDSROW;              // doubles per stage: [A chunk | B chunk]
  static constexpr int LDS_DOUBLES = 2 * STAGE + 32;         // two stages + slack for the (discarded) out-of-range tile reads
  static constexpr int UNITS = KC * MP / 2;                  // double2 units per operand chunk
  static constexpr int NLD = (UNITS + SO_THREADS - 1) / SO_THREADS;  // loads per thread and operand
};

// One wavefront's share of C = Aop' Bop: tiles [HALF NT0, ...) of block (rt0, ct0).  K a multiple of KC (rows beyond the
// data must be zero).  lds: SoGemmCfg<MT>::LDS_DOUBLES doubles, 16-byte aligned.  All 512 threads run it; barriers inside.
template <int MT, int HALF, class Epi>
__device__ __forceinline__ void so_gemm_half(const double* __restrict__ Aop, int lda, const double* __restrict__ Bop, int ldb,
                                             int K, double* lds, int rt0, int nrt, int ct0, int nct, Epi epi) {
  using Cfg = SoGemmCfg<MT>;
  constexpr int MP = Cfg::MP, LDSROW = Cfg::LDSROW, KC = Cfg::KC, RB = Cfg::RB, STAGE = Cfg::STAGE, UNITS = Cfg::UNITS,
                NLD = Cfg::NLD, T0 = HALF ? Cfg::NT0 : 0, NT = HALF ? RB * RB - Cfg::NT0 : Cfg::NT0;
  constexpr int I0 = T0 / RB, I1 = (T0 + NT - 1) / RB;  // block rows this half touches
  const int tid = threadIdx.x, lane = tid & 63;
  so_v4f64 acc[NT > 0 ? NT : 1];  // (NT = 0: a one-tile block's second wavefront only helps with the staging)
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = so_v4f64{0.0, 0.0, 0.0, 0.0};
  double2 ra[NLD], rb[NLD];
  auto load_regs = [&](int k0) {
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      int idx = tid + SO_THREADS * u;
      idx = idx < UNITS ? idx : UNITS - 1;  // (clamped: unconditional loads)
      const int r = idx / (MP / 2), c2 = idx - r * (MP / 2);
      ra[u] = *(const double2*)(Aop + (size_t)(k0 + r) * lda + 2 * c2);
      rb[u] = *(const double2*)(Bop + (size_t)(k0 + r) * ldb + 2 * c2);
    }
  };
  auto store_lds = [&](int stage) {
    double* sa = lds + stage * STAGE;
    double* sb = sa + KC * LDSROW;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int idx = tid + SO_THREADS * u;
      if (idx < UNITS) {
        const int r = idx / (MP / 2), c2 = idx - r * (MP / 2);
        *(double2*)(sa + r * LDSROW + 2 * c2) = ra[u];
        *(double2*)(sb + r * LDSROW + 2 * c2) = rb[u];
      }
    }
  };
  const int nchunks = K / KC;
  const int foff = (lane >> 4) * LDSROW + (lane & 15);
  __syncthreads();  // (the previous user of the staging buffers is done; global results of the previous phase are visible)
  load_regs(0);
  store_lds(0);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    if (c + 1 < nchunks) load_regs((c + 1) * KC);
    const double* sa = lds + (c & 1) * STAGE + foff;
    const double* sb = sa + KC * LDSROW;
#pragma unroll
    for (int kk = 0; kk < KC / 4; ++kk) {
      double a[RB], b[RB];
#pragma unroll
      for (int i = I0; i <= I1; ++i) a[i] = sa[kk * 4 * LDSROW + (rt0 + i) * 16];
#pragma unroll
      for (int j = 0; j < RB; ++j) b[j] = sb[kk * 4 * LDSROW + (ct0 + j) * 16];
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(T0 + t) / RB], b[(T0 + t) % RB], acc[t], 0, 0, 0);
    }
    if (c + 1 < nchunks) store_lds((c + 1) & 1);
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int i = (T0 + t) / RB, j = (T0 + t) % RB;
    if (i < nrt && j < nct) {
#pragma unroll
      for (int r = 0; r < 4; ++r) epi(16 * (rt0 + i) + (lane >> 4) + 4 * r, 16 * (ct0 + j) + (lane & 15), acc[t][r]);
    }
  }
}

// C = Aop' Bop handed to `epi(row, col, value)` element by element (rows / columns < 16 MT): lane l holds, for tile (ti, tj)
// and r = 0..3, the element (16 ti + (l >> 4) + 4 r, 16 tj + (l & 15)).  Every element is delivered exactly once.
template <int MT, class Epi>
__device__ __forceinline__ void so_gemm(const double* __restrict__ Aop, int lda, const double* __restrict__ Bop, int ldb, int K,
                                        double* lds, Epi epi) {
  constexpr int RB = SoGemmCfg<MT>::RB;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (scalar: block offsets and the half are wave-uniform)
  const int pair = wave >> 1, rbk = pair >> 1, cbk = pair & 1;
  const int rt0 = rbk ? RB : 0, nrt = rbk ? MT - RB : RB, ct0 = cbk ? RB : 0, nct = cbk ? MT - RB : RB;
  if (wave & 1)
    so_gemm_half<MT, 1>(Aop, lda, Bop, ldb, K, lds, rt0, nrt, ct0, nct, epi);
  else
    so_gemm_half<MT, 0>(Aop, lda, Bop, ldb, K, lds, rt0, nrt, ct0, nct, epi);
}

}  // namespace dsge

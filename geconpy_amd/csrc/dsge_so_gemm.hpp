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
  static constexpr int LDSROW = MP;                          // linear image of the chunk (direct-to-LDS loads write lane-linear);
                                                             // == 16 (mod 32) for odd MT, two-way conflicts on the small even ones
  static constexpr int KC = 8;                               // rows of Aop / Bop per stage
  static constexpr int RB = (MT + 1) / 2;                    // tiles per block side (the larger block)
  static constexpr int NT0 = (RB * RB + 1) / 2;              // tiles of a block the first wavefront of the pair takes
  static constexpr int STAGE = 2 * KC * LDSROW;              // doubles per stage: [A chunk | B chunk]
  static constexpr int LDS_DOUBLES = (2 * STAGE + 32 > 8 * 16 * 17) ? 2 * STAGE + 32 : 8 * 16 * 17;  // two stages + slack for the
                                                             // (discarded) out-of-range tile reads; at least the eight 16 x 17
                                                             // transposition tiles of so_gemm_sym's mirrored epilogue
  static constexpr int PIECES = KC * MP * 8 / 1024;          // 1 KiB wave-instructions per operand chunk (KC MP 8 bytes = MT KiB)
  static_assert(KC * MP * 8 % 1024 == 0, "operand chunk is a whole number of 1 KiB pieces");
};

// One wavefront's share of C = Aop' Bop: tiles [HALF NT0, ...) of block (rt0, ct0).  K a multiple of KC (rows beyond the
// data must be zero).  lds: SoGemmCfg<MT>::LDS_DOUBLES doubles, 16-byte aligned.  All 512 threads run it; barriers inside.
struct SoZeroInit {
  __device__ __forceinline__ so_v4f64 operator()(int, int) const { return so_v4f64{0.0, 0.0, 0.0, 0.0}; }
};

// init(row0, col): the starting value of a tile fragment (C = init + Aop' Bop): an epilogue that needs a matrix from global
// memory per element pays one memory latency per fragment -- 25 in a row per wavefront -- whereas the fragments' starting
// values are all requested at once, before the first chunk is multiplied.
struct SoNoTranspose {};

template <int MT, int HALF, class Epi, class Init, class EpiT = SoNoTranspose>
__device__ __forceinline__ void so_gemm_half(const double* __restrict__ Aop, int lda, const double* __restrict__ Bop, int ldb,
                                             int K, double* lds, int rt0, int nrt, int ct0, int nct, Epi epi, Init init,
                                             EpiT epit = EpiT()) {
  using Cfg = SoGemmCfg<MT>;
  constexpr int MP = Cfg::MP, LDSROW = Cfg::LDSROW, KC = Cfg::KC, RB = Cfg::RB, STAGE = Cfg::STAGE, PIECES = Cfg::PIECES,
                T0 = HALF ? Cfg::NT0 : 0, NT = HALF ? RB * RB - Cfg::NT0 : Cfg::NT0;
  constexpr int I0 = T0 / RB, I1 = (T0 + NT - 1) / RB;  // block rows this half touches
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));  // (opaque: the fragment addresses are re-derived per product, not hoisted out of the caller's loops)
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  so_v4f64 acc[NT > 0 ? NT : 1];  // (NT = 0: a one-tile block's second wavefront only helps with the staging)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int i = (T0 + t) / RB, j = (T0 + t) % RB;
    acc[t] = (i < nrt && j < nct) ? init(16 * (rt0 + i) + (lane >> 4), 16 * (ct0 + j) + (lane & 15)) : so_v4f64{0.0, 0.0, 0.0, 0.0};
  }
  // Staging: rows [k0, k0 + KC) of both operands straight into LDS (global_load_lds_dwordx4: 16 bytes per lane, 1 KiB per
  // wave-instruction, destination = wave-uniform base + 16 lane; no staging registers -- with 200 accumulator registers per
  // wavefront the register-staged version spilled its prefetch to scratch right behind the loads and exposed the whole memory
  // latency in every chunk).  The operand rows are contiguous in memory (lda = ldb = MP), so a chunk is one linear image of
  // 2 PIECES KiB; the eight wavefronts take the pieces round-robin.
  auto stage_chunk = [&](int stage, int k0) {
    const char* ga = (const char*)(Aop + (size_t)k0 * MP);
    const char* gb = (const char*)(Bop + (size_t)k0 * MP);
    char* ls = (char*)(lds + stage * STAGE);
    for (int t = wave; t < 2 * PIECES; t += SO_THREADS / 64) {
      const char* src = (t < PIECES ? ga + (size_t)t * 1024 : gb + (size_t)(t - PIECES) * 1024) + lane * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(ls + (size_t)t * 1024), 16, 0, 0);
    }
  };
  const int nchunks = K / KC;
  const int foff = (lane >> 4) * LDSROW + (lane & 15);
  __syncthreads();  // (the previous user of the staging buffers is done; global results of the previous phase are visible)
  stage_chunk(0, 0);
  __syncthreads();  // (carries the vmcnt(0) that lands chunk 0)
  for (int c = 0; c < nchunks; ++c) {
    if (c + 1 < nchunks) stage_chunk((c + 1) & 1, (c + 1) * KC);  // in flight while chunk c is multiplied
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
    __syncthreads();  // (everybody is done with stage c & 1; the loads of chunk c + 1 have landed: vmcnt(0) before the barrier)
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int i = (T0 + t) / RB, j = (T0 + t) % RB;
    if (i < nrt && j < nct) epi(16 * (rt0 + i) + (lane >> 4), 16 * (ct0 + j) + (lane & 15), acc[t]);
  }
  if constexpr (!__is_same(EpiT, SoNoTranspose)) {
    // C' as well (A_k^2 in both layouts): every tile once more, transposed through this wavefront's 16 x 17 doubles of the idle
    // staging LDS and handed to `epit` with the coordinates of the transposed tile -- the store then coalesces like the natural
    // one (a store with transposed addresses cost +50 us per product)
    double* tr = lds + wave * (16 * 17);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int i = (T0 + t) / RB, j = (T0 + t) % RB;
      if (i < nrt && j < nct) {
        int lq = lane >> 4, lm = lane & 15;
        asm volatile("" : "+v"(lq), "+v"(lm));
#pragma unroll
        for (int r = 0; r < 4; ++r) tr[(lq + 4 * r) * 17 + lm] = acc[t][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        so_v4f64 vt;
#pragma unroll
        for (int r = 0; r < 4; ++r) vt[r] = tr[lm * 17 + lq + 4 * r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        epit(16 * (ct0 + j) + lq, 16 * (rt0 + i) + lm, vt);
      }
    }
  }
}

// C = Aop' Bop handed to `epi(row0, col, v)` one tile fragment at a time (rows / columns < 16 MT): v[r], r = 0..3, is the element
// (row0 + 4 r, col) -- lane l of tile (ti, tj) has row0 = 16 ti + (l >> 4), col = 16 tj + (l & 15).  Every element is delivered
// exactly once.  (Per fragment, so that an epilogue that also READS can issue its four loads before its four stores.)
template <int MT, class Epi, class Init = SoZeroInit, class EpiT = SoNoTranspose>
__device__ __forceinline__ void so_gemm(const double* __restrict__ Aop, int lda, const double* __restrict__ Bop, int ldb, int K,
                                        double* lds, Epi epi, Init init = Init(), EpiT epit = EpiT()) {
  constexpr int RB = SoGemmCfg<MT>::RB;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (scalar: block offsets and the half are wave-uniform)
  const int pair = wave >> 1, rbk = pair >> 1, cbk = pair & 1;
  const int rt0 = rbk ? RB : 0, nrt = rbk ? MT - RB : RB, ct0 = cbk ? RB : 0, nct = cbk ? MT - RB : RB;
  if (wave & 1)
    so_gemm_half<MT, 1>(Aop, lda, Bop, ldb, K, lds, rt0, nrt, ct0, nct, epi, init, epit);
  else
    so_gemm_half<MT, 0>(Aop, lda, Bop, ldb, K, lds, rt0, nrt, ct0, nct, epi, init, epit);
}

// ---- symmetric product: C = Aop' Bop is KNOWN to be symmetric (X = W Az' = Az P Az', A_k P A_k', Az Az') --------------------------
// Only the 16 x 16 tiles on and above the diagonal are multiplied -- MT (MT + 1) / 2 of MT^2: 91 of 169 at MT = 13 -- and every
// off-diagonal tile is handed to the epilogue twice: as computed, and mirrored (transposed through a wave-private 2 KB of the
// by then idle staging LDS, so that the mirrored store has the SAME coalescing as the natural one; a store with transposed
// addresses cost +50 us per product).  The tile rows are cut into four ranges R0..R3 (4, 3, 3, 3 at MT = 13); the six
// off-diagonal range pairs and the four diagonal triangles are dealt to the eight wavefronts by SO_SYM_GROUPS so that the
// two wavefronts of a SIMD (wave w and w + 4: workgroup waves go round-robin over the SIMDs) carry 21..24 tiles together,
// against 49 for the full product.  Tile lists are compile-time constants: accumulators stay in registers.
struct SoTileList {
  int n;
  int ti[28], tj[28];
};

// group g: 0..5 = range pairs (0,1) (0,2) (0,3) (1,2) (1,3) (2,3); 6..9 = the triangles of R0..R3.  Up to three groups per wave.
constexpr int SO_SYM_GROUPS[8][3] = {
    {0, -1, -1},  // wave 0 (SIMD 0): R0 x R1
    {1, -1, -1},  // wave 1 (SIMD 1): R0 x R2
    {2, -1, -1},  // wave 2 (SIMD 2): R0 x R3
    {4, 9, -1},   // wave 3 (SIMD 3): R1 x R3 + triangle R3
    {6, -1, -1},  // wave 4 (SIMD 0): triangle R0
    {7, 8, -1},   // wave 5 (SIMD 1): triangles R1, R2
    {3, -1, -1},  // wave 6 (SIMD 2): R1 x R2
    {5, -1, -1},  // wave 7 (SIMD 3): R2 x R3
};

template <int MT>
constexpr int so_sym_range_begin(int r) {  // ranges of MT / 4 tiles, the first MT % 4 one longer
  const int base = MT / 4, rem = MT % 4;
  return r * base + (r < rem ? r : rem);
}

template <int MT>
constexpr SoTileList so_sym_tiles(int wave) {
  SoTileList L{};
  constexpr int PA[6] = {0, 0, 0, 1, 1, 2}, PB[6] = {1, 2, 3, 2, 3, 3};
  for (int q = 0; q < 3; ++q) {
    const int g = SO_SYM_GROUPS[wave][q];
    if (g < 0) continue;
    if (g < 6) {
      for (int i = so_sym_range_begin<MT>(PA[g]); i < so_sym_range_begin<MT>(PA[g] + 1); ++i)
        for (int j = so_sym_range_begin<MT>(PB[g]); j < so_sym_range_begin<MT>(PB[g] + 1); ++j) {
          L.ti[L.n] = i;
          L.tj[L.n] = j;
          ++L.n;
        }
    } else {
      const int r = g - 6;
      for (int i = so_sym_range_begin<MT>(r); i < so_sym_range_begin<MT>(r + 1); ++i)
        for (int j = i; j < so_sym_range_begin<MT>(r + 1); ++j) {
          L.ti[L.n] = i;
          L.tj[L.n] = j;
          ++L.n;
        }
    }
  }
  return L;
}

template <int MT, int WAVE, class Epi>
__device__ __forceinline__ void so_gemm_sym_wave(const double* __restrict__ Aop, const double* __restrict__ Bop, int K, double* lds,
                                                 Epi epi) {
  using Cfg = SoGemmCfg<MT>;
  constexpr int MP = Cfg::MP, LDSROW = Cfg::LDSROW, KC = Cfg::KC, STAGE = Cfg::STAGE, PIECES = Cfg::PIECES;
  constexpr SoTileList TL = so_sym_tiles<MT>(WAVE);
  constexpr int NT = TL.n;
  const int tid = threadIdx.x, lane = tid & 63;
  so_v4f64 acc[NT > 0 ? NT : 1];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = so_v4f64{0.0, 0.0, 0.0, 0.0};
  auto stage_chunk = [&](int stage, int k0) {
    const char* ga = (const char*)(Aop + (size_t)k0 * MP);
    const char* gb = (const char*)(Bop + (size_t)k0 * MP);
    char* ls = (char*)(lds + stage * STAGE);
    for (int t = WAVE; t < 2 * PIECES; t += SO_THREADS / 64) {
      const char* src = (t < PIECES ? ga + (size_t)t * 1024 : gb + (size_t)(t - PIECES) * 1024) + lane * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(ls + (size_t)t * 1024), 16, 0, 0);
    }
  };
  const int nchunks = K / KC;
  const int foff = (lane >> 4) * LDSROW + (lane & 15);
  __syncthreads();
  stage_chunk(0, 0);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    if (c + 1 < nchunks) stage_chunk((c + 1) & 1, (c + 1) * KC);
    const double* sa = lds + (c & 1) * STAGE + foff;
    const double* sb = sa + KC * LDSROW;
#pragma unroll
    for (int kk = 0; kk < KC / 4; ++kk) {
      double a[MT], b[MT];  // (only the fragments the tile list names are ever read: the others are dead code)
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = sa[kk * 4 * LDSROW + i * 16];
#pragma unroll
      for (int j = 0; j < MT; ++j) b[j] = sb[kk * 4 * LDSROW + j * 16];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[TL.ti[t]], b[TL.tj[t]], acc[t], 0, 0, 0);
    }
    __syncthreads();
  }
  // epilogue: natural, then mirrored through this wavefront's 16 x 17 doubles of the (idle) staging buffers
  double* tr = lds + WAVE * (16 * 17);
  static_assert(8 * 16 * 17 <= Cfg::LDS_DOUBLES, "transposition scratch fits the staging buffers");
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int i = TL.ti[t], j = TL.tj[t];
    // The store addresses of a tile are built from these two values right here: they are loop-invariant for the caller's
    // time loop, and left alone the compiler computes the 64-bit addresses of all eight wavefronts' tiles -- 600 of them -- in
    // the kernel's prologue and spills them (603 scratch stores, measured).  The empty asm makes them opaque per tile.
    int lq = lane >> 4, lm = lane & 15;
    asm volatile("" : "+v"(lq), "+v"(lm));
#pragma unroll
    for (int r = 0; r < 4; ++r) tr[(lq + 4 * r) * 17 + lm] = acc[t][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    so_v4f64 vt;
#pragma unroll
    for (int r = 0; r < 4; ++r) vt[r] = tr[lm * 17 + lq + 4 * r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (i != j) {
      epi(16 * i + lq, 16 * j + lm, acc[t]);
      epi(16 * j + lq, 16 * i + lm, vt);
    } else {
      // A diagonal tile is symmetrised too, element by element (upper triangle kept).  NOT optional: with the off-diagonal
      // tiles exactly symmetric and the diagonal ones carrying their rounding-level antisymmetric part, the covariance
      // recursion of the pruned filter diverged after ~150 steps on 2 of 1024 SW-shaped draws (the antisymmetric part evolves
      // as N -> Az N Az', a contraction only asymptotically for a non-normal Az; projected onto the diagonal tiles every step it
      // grew by ~1.2 per step -- reproduced in numpy, profiles/r3/so_mirror_stability.txt); all of it mirrored, or none, is stable.
      so_v4f64 vs;
#pragma unroll
      for (int r = 0; r < 4; ++r) vs[r] = (lq + 4 * r <= lm) ? acc[t][r] : vt[r];
      epi(16 * i + lq, 16 * j + lm, vs);
    }
  }
}

// C = Aop' Bop for a product that is symmetric in exact arithmetic; every element of C is delivered to `epi` exactly once (the
// strictly lower tiles as the transposes of the upper ones, the diagonal tiles with their upper triangles mirrored: C comes out
// EXACTLY symmetric).
// lda = ldb = 16 MT as for so_gemm.
template <int MT, class Epi>
__device__ __forceinline__ void so_gemm_sym(const double* __restrict__ Aop, const double* __restrict__ Bop, int K, double* lds,
                                            Epi epi) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  switch (wave) {
    case 0: so_gemm_sym_wave<MT, 0>(Aop, Bop, K, lds, epi); break;
    case 1: so_gemm_sym_wave<MT, 1>(Aop, Bop, K, lds, epi); break;
    case 2: so_gemm_sym_wave<MT, 2>(Aop, Bop, K, lds, epi); break;
    case 3: so_gemm_sym_wave<MT, 3>(Aop, Bop, K, lds, epi); break;
    case 4: so_gemm_sym_wave<MT, 4>(Aop, Bop, K, lds, epi); break;
    case 5: so_gemm_sym_wave<MT, 5>(Aop, Bop, K, lds, epi); break;
    case 6: so_gemm_sym_wave<MT, 6>(Aop, Bop, K, lds, epi); break;
    default: so_gemm_sym_wave<MT, 7>(Aop, Bop, K, lds, epi); break;
  }
}

}  // namespace dsge

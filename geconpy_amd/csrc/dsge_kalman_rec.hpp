// Per-draw record of a forward filter sweep, as the reverse sweep of the gradient reads it (dsge_kalman_grad.hpp).
// Doubles per time step: for a FULL step the results of its covariance update -- P+ (NP x NP, lane-major: entry (i, j) of lane l's
// BS x BS block at [(i BS + j) 64 + l]), K (NP x 8), F^-1 and F (8 x 8, row-major) --; for every step the predicted state a_t and the
// index of the step whose covariance update it shares (its own for a full step), and for a full step the source of the PREVIOUS
// segment (-1 for the first).  The initial covariance P_0 (NP x NP, row-major) sits behind the last step.
// Written by kalman_grad_kernel's own forward sweep or, round 5, by kalman_nt_kernel<.., REC = true> (dsge_kalman_nt.hpp).
#pragma once
#include <cstddef>

namespace dsge {

template <int BS>
struct KgRec {
  static constexpr int NP = 8 * BS;
  static constexpr size_t OFF_K = (size_t)NP * NP, OFF_FI = OFF_K + (size_t)NP * 8, OFF_F = OFF_FI + 64, OFF_A = OFF_F + 64,
                          OFF_SRC = OFF_A + NP, OFF_PREV = OFF_SRC + 1, STEP = OFF_PREV + 1;
  // Behind P_0: the state the tail kernel (kalman_grad_tail_kernel: the reverse mean side of the LAST steady segment, run at two
  // wavefronts per SIMD) hands to the reverse sweep -- [0] number of steps it processed (0: none), [1] nlam; abar (NP), dbar (8),
  // then lane-major register images: Tbar (BS^2 x 64), Kacc (BS x 64), Qacc (64)
  // [2] (TS_LAYOUT) how the forward sweep stored P+ of its full steps: 0 = lane-major register blocks (kalman_nt_kernel<.., REC>,
  // kalman_grad_kernel's own forward sweep), LAYOUT_TILES + TM = upper 4 x 4 tiles of a TM-tile matrix, element g of lane l at
  // [g 64 + l] (kalman_mf_kernel<.., REC>, round 6).  Every forward sweep writes it.
  static constexpr size_t TS_LAYOUT = 2;
  static constexpr int LAYOUT_TILES = 100;
  static constexpr size_t TS_AB = 16, TS_DB = TS_AB + NP, TS_TB = TS_DB + 8, TS_KA = TS_TB + (size_t)BS * BS * 64,
                          TS_QA = TS_KA + (size_t)BS * 64, TAIL_STATE = TS_QA + 64;
  __host__ __device__ static constexpr size_t tail_state_off(int T_len) { return (size_t)T_len * STEP + (size_t)NP * NP; }
  __host__ __device__ static constexpr size_t per_draw(int T_len) { return tail_state_off(T_len) + TAIL_STATE; }
};

}  // namespace dsge

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
  __host__ __device__ static constexpr size_t per_draw(int T_len) { return (size_t)T_len * STEP + (size_t)NP * NP; }
};

}  // namespace dsge

// assemble_kernels.hpp — hand-written gfx950 (CDNA4, wave64) kernels for the Gauss-Newton
// normal-equation assembly path.
//
// One streaming pass per LM iteration: every lane reads ITEMS consecutive correspondences
// from a tiled struct-of-arrays dataset (one wide load per field), evaluates residual,
// analytic Jacobian and robust weight in registers, and keeps the 28 (or 10) running sums
// of  H = Σ w JᵀJ (upper triangle),  g = Σ w Jᵀr,  cost = Σ ρ  in registers for the whole
// grid-stride loop.  The sums leave the lane exactly once: wave64 butterfly → LDS across
// the block's waves → one row of `partials`, then a fixed-order final pass → 28 doubles.
// No atomics, so results are bit-reproducible for a fixed launch geometry.
//
// Bound: HBM bandwidth (120 B fp64 / 60 B fp32 per correspondence, ≈200 VALU ops); the
// contraction is a fixed 6×6 outer product so MFMA is deliberately not used.
//
// Math restated from (reference paths relative to nonlinear_optimizer/):
//   6-DoF NDT     mahalanobis_distance_minimizer/mahalanobis_distance_minimizer_analytic.cc:159-185
//                 (lane form: ..._analytic_simd_various.cc:656-697)
//   3-DoF NDT     mahalanobis_distance_minimizer/mahalanobis_distance_minimizer_analytic_3dof.cc:110-139
//   reprojection  reprojection_error_minimizer/reprojection_error_minimizer_analytic.cc:107-162
//   robust loss   loss_function.h:28-33, 57-66
//
// Split by role (round 4): items → reductions → loop state and hand-offs → launch-per-pass kernel → one-launch loop →
// voxel-indexed layout → the small stand-alone kernels.  Variants that were measured and lost live in
// tools/exp/assemble_variants_r03.hpp, not here.
#pragma once

#include "assemble_misc.hpp"  // includes the others, in the order above

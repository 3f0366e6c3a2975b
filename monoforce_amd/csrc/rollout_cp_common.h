// Component-parallel lane mapping of the rollout kernels (gfx950, wave64) -- shared lane algebra.
//
// The G = 4 mapping (rollout_fwd_kernel.h) gives one contact point to a lane: every 3-vector operation of a step costs three
// instructions, and the 18-float body update is repeated by each of the four lanes.  A launch of the BASELINE shape
// (1024 rollouts x 4 points) is then 64 waves on 1024 SIMDs, each issuing ~300 instructions per step at one per ~4.4 cycles:
// the step time IS the instruction count of one wave.  Here a rollout owns a 16-lane DPP row instead: quad p (lanes 4p..4p+3)
// is contact point p, and inside a quad
//   * lane c = 0, 1, 2 holds COMPONENT c of every 3-vector (x, xd, w, r, v_p, n, F, tau, ...) and ROW c of R; lane 3 mirrors
//     lane 2 (same values, same store addresses -- so stores need no exec mask and the loop stays one basic block);
//   * lane q = 0..3 gathers CELL q of the bilinear footprint (c, f, l, fl): one load per lane instead of two or four.
// Dot products are a multiply and two DPP adds (quad rotations), cross products read their rotated operands through DPP
// operands, sums over the contact points are two row rotations (row_ror:8, row_ror:4).  ~2.2x fewer instructions per wave and
// step; 4 rollouts per wave, so B = 1024 is 256 waves -- still at most one per SIMD up to B = 4096.
#pragma once
#include "mf_common.h"

namespace mf {
namespace cp {

// quad_perm encodings (sel0 | sel1 << 2 | sel2 << 4 | sel3 << 6)
constexpr int kRot1 = 0x09;    // [1,2,0,0]: lane c reads component (c+1)%3; lane 3 = lane 2's choice
constexpr int kRot2 = 0x52;    // [2,0,1,1]: lane c reads component (c+2)%3
constexpr int kB0 = 0x00, kB1 = 0x55, kB2 = 0xAA, kB3 = 0xFF;   // broadcast lane 0 / 1 / 2 / 3 of the quad
constexpr int kXor1 = 0xB1, kXor2 = 0x4E;
constexpr int kN12 = 0xA9;     // [1,2,2,2]: lane 0 <- 1, lane 1 <- 2 (finite differences z_f - z_c, z_l - z_c to components 0, 1)
constexpr int kRor4 = 0x124, kRor8 = 0x128;   // rotate within the 16-lane row

template <int CTRL>
__device__ __forceinline__ float dpp(float v) { return dpp_mov<CTRL>(v); }
template <int CTRL>
__device__ __forceinline__ int dppi(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }

// sum of the three components held by lanes 0..2 of a quad; every lane (3 included, as lane 2's mirror) gets the total
__device__ __forceinline__ float sum3(float v) { return (v + dpp<kRot1>(v)) + dpp<kRot2>(v); }
// a . b over the components: the product is rounded on its own (contracting it into the first add would need the rotated copy
// in a register of its own: one more instruction), then two DPP adds
__device__ __forceinline__ float dot3(float a, float b) {
#pragma clang fp contract(off)
  const float v = a * b;
  return (v + dpp<kRot1>(v)) + dpp<kRot2>(v);
}
// bitwise select through precomputed lane masks (all ones / zero): a plain ternary on the lane role turns into branches
__device__ __forceinline__ float mask_or(float acc, float v, unsigned m) {
  return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, acc) | (__builtin_bit_cast(unsigned, v) & m));
}
// sum over the four lanes of a quad (cell roles)
__device__ __forceinline__ float sum4(float v) { v += dpp<kXor1>(v); return v + dpp<kXor2>(v); }
// sum over the four quads of a row, lane position by lane position (contact points of one rollout)
__device__ __forceinline__ float sum_points(float v) { v += dpp<kRor8>(v); return v + dpp<kRor4>(v); }
// (a x b)_c given the components of a and b in the lanes: a_{c+1} b_{c+2} - a_{c+2} b_{c+1}
__device__ __forceinline__ float cross_c(float a, float b) { return dpp<kRot1>(a) * dpp<kRot2>(b) - dpp<kRot2>(a) * dpp<kRot1>(b); }

}  // namespace cp
}  // namespace mf

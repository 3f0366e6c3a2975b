// Device helpers shared by the forward and backward rollout kernels: the reference's `interpolate_grid` cell lookup.
#pragma once
#include "mf_common.h"

namespace mf {

// Cell indices and fractions of `interpolate_grid` (dphysics.py:419-435), filled by locate_m() in rollout_fwd_kernel.h.
// Index arithmetic is done in int32 after clamping the cell coordinate to +-2^18 (the reference uses int64; results are
// identical while the robot is within 2^18 cells of the map, far beyond which the flat-index clamp pins everything to
// cell 0 / HW-1 anyway).
template <typename S>
struct Cell {
  int ic, i_f, il, ifl;
  S fx, fy;
};

// The four bilinear weights, each one product rounded on its own (also in the FMA-contracted kernels: contraction would turn
// (1 - fx) * y into fma(-fx, y, y) and make a map of ones blend differently from the weights' own sum).
// NB: the x-fraction weights the +y ("left") neighbour and vice versa (dphysics.py:442-445).
template <typename S>
struct BlendW { S w00, w01, w10, w11; };
template <typename S>
__device__ __forceinline__ BlendW<S> blend_weights(const Cell<S>& c) {
#pragma clang fp contract(off)
  const S one = (S)1;
  BlendW<S> w;
  w.w00 = (one - c.fx) * (one - c.fy); w.w01 = (one - c.fx) * c.fy; w.w10 = c.fx * (one - c.fy); w.w11 = c.fx * c.fy;
  return w;
}

template <typename S>
__device__ __forceinline__ S blend(const Cell<S>& c, S vc, S vf, S vl, S vfl) {
  const BlendW<S> w = blend_weights(c);
  return w.w00 * vc + w.w01 * vf + w.w10 * vl + w.w11 * vfl;
}

// blend() of a map of ones (no friction map = cfg.friction = ones, dphysics.py:141,562): the sum of the weights in blend()'s
// order, bit-identical to blend(c, 1, 1, 1, 1) with or without contraction.
template <typename S>
__device__ __forceinline__ S blend_ones(const Cell<S>& c) {
  const BlendW<S> w = blend_weights(c);
  return ((w.w00 + w.w01) + w.w10) + w.w11;
}

// d(blend)/d(fx), d(blend)/d(fy): the only way a query position influences a sampled value (indices are constants).
template <typename S>
__device__ __forceinline__ void blend_grad(const Cell<S>& c, S vc, S vf, S vl, S vfl, S* dfx, S* dfy) {
  const S one = (S)1;
  *dfx = -(one - c.fy) * vc - c.fy * vf + (one - c.fy) * vl + c.fy * vfl;
  *dfy = -(one - c.fx) * vc + (one - c.fx) * vf - c.fx * vl + c.fx * vfl;
}

}  // namespace mf

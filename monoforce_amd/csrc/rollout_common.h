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

// Two maps blended with the same weights in one pass of packed float32 arithmetic (v_pk_mul_f32 / v_pk_fma_f32: two
// independent IEEE operations per instruction, so each half is bit-identical to blend() of that map).
template <typename S>
__device__ __forceinline__ void blend2(const Cell<S>& c, const S (&a)[4], const S (&b)[4], S* ra, S* rb) {
  *ra = blend(c, a[0], a[1], a[2], a[3]);
  *rb = blend(c, b[0], b[1], b[2], b[3]);
}
template <>
__device__ __forceinline__ void blend2<float>(const Cell<float>& c, const float (&a)[4], const float (&b)[4], float* ra, float* rb) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  const BlendW<float> w = blend_weights(c);
  const f2 v0 = {a[0], b[0]}, v1 = {a[1], b[1]}, v2 = {a[2], b[2]}, v3 = {a[3], b[3]};
  const f2 r = w.w00 * v0 + w.w01 * v1 + w.w10 * v2 + w.w11 * v3;      // same operation order as blend()
  *ra = r.x; *rb = r.y;
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

// Device helpers shared by the forward and backward rollout kernels: the reference's `interpolate_grid` cell lookup.
#pragma once
#include "mf_common.h"

namespace mf {

// Cell indices and fractions of `interpolate_grid` (dphysics.py:419-435).  Index arithmetic is done in int32 after
// clamping the cell coordinate to +-2^18 (the reference uses int64; results are identical while the robot is within
// 2^18 cells of the map, far beyond which the flat-index clamp pins everything to cell 0 / HW-1 anyway).
template <typename S>
struct Cell {
  int ic, i_f, il, ifl;
  S fx, fy;
};

template <typename S>
__device__ __forceinline__ Cell<S> locate(S qx, S qy, S d_max, S res, int H, int last) {
  const S lim = (S)262144.0;
  S ux = (qx + d_max) / res;
  S uy = (qy + d_max) / res;
  int ix = (int)mf_clamp(ux, -lim, lim);  // trunc toward zero, like .long()
  int iy = (int)mf_clamp(uy, -lim, lim);
  Cell<S> c;
  c.fx = ux - (S)ix;
  c.fy = uy - (S)iy;
  int base = iy + H * ix;
  c.ic = min(max(base, 0), last);
  c.i_f = min(max(base + H, 0), last);
  c.il = min(max(base + 1, 0), last);
  c.ifl = min(max(base + 1 + H, 0), last);
  return c;
}

template <typename S>
__device__ __forceinline__ S blend(const Cell<S>& c, S vc, S vf, S vl, S vfl) {
  // NB: the x-fraction weights the +y ("left") neighbour and vice versa (dphysics.py:442-445).
  const S one = (S)1;
  return (one - c.fx) * (one - c.fy) * vc + (one - c.fx) * c.fy * vf + c.fx * (one - c.fy) * vl + c.fx * c.fy * vfl;
}


// d(blend)/d(fx), d(blend)/d(fy): the only way a query position influences a sampled value (indices are constants).
template <typename S>
__device__ __forceinline__ void blend_grad(const Cell<S>& c, S vc, S vf, S vl, S vfl, S* dfx, S* dfy) {
  const S one = (S)1;
  *dfx = -(one - c.fy) * vc - c.fy * vf + (one - c.fy) * vl + c.fy * vfl;
  *dfy = -(one - c.fx) * vc + (one - c.fx) * vf - c.fx * vl + c.fx * vfl;
}

}  // namespace mf

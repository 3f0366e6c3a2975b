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

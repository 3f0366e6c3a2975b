// interpolate_grid (dphysics.py:385-455): kernel template shared by interp_grid.hip (exact) and interp_grid_fast.hip.
#pragma once
#include "rollout_fwd_kernel.h"

namespace mf {

template <typename S>
struct InterpArgs {
  int B, N, H, W, map_shared;
  S res, inv_res, d_max;
  const S* grid;
  const S* xq;
  const S* yq;
  S* z;
  S* n;
  int32_t* cells;
  S* frac;
};

template <typename S, bool FAST>
__global__ void __launch_bounds__(256) interp_grid_kernel(const InterpArgs<S> a) {
  using M = Mth<S, FAST>;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)a.B * a.N) return;
  const int b = (int)(i / a.N);
  const int HW = a.H * a.W, last = HW - 1;
  const unsigned moff = a.map_shared ? 0u : (unsigned)b * (unsigned)HW;
  const S one = (S)1;
  const Cell<S> c = locate_m<S, FAST>(a.xq[i], a.yq[i], a.d_max, a.res, a.inv_res, a.H, last);      // (:419-435)
  S z4[4];
  gather4(a.grid, moff, c, last, z4);                                                              // (:438-441)
  a.z[i] = blend(c, z4[0], z4[1], z4[2], z4[3]);                                                   // (:442-445)
  if (a.n) {                                                                                       // (:447-453), as the contact model computes it
    const S gx = M::div(z4[1] - z4[0], a.res), gy = M::div(z4[2] - z4[0], a.res);
    S n0, n1, n2;
    if (M::kReciprocalNorm) {
      const S inl = M::inv_len(gx * gx + gy * gy + one);
      n0 = -gx * inl; n1 = -gy * inl; n2 = inl;
    } else {
      const S nl = mf_max(M::sqrt(gx * gx + gy * gy + one), (S)1e-6);
      n0 = -gx / nl; n1 = -gy / nl; n2 = one / nl;
    }
    a.n[i * 3 + 0] = n0; a.n[i * 3 + 1] = n1; a.n[i * 3 + 2] = n2;
  }
  if (a.cells) { a.cells[i * 4 + 0] = c.ic; a.cells[i * 4 + 1] = c.i_f; a.cells[i * 4 + 2] = c.il; a.cells[i * 4 + 3] = c.ifl; }
  if (a.frac) { a.frac[i * 2 + 0] = c.fx; a.frac[i * 2 + 1] = c.fy; }
}

template <typename S, bool FAST>
void launch_interp(const InterpArgs<S>& a, hipStream_t st) {
  const long long n = (long long)a.B * a.N;
  hipLaunchKernelGGL((interp_grid_kernel<S, FAST>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
}
void launch_interp_fast_f32(const InterpArgs<float>& a, hipStream_t st);

}  // namespace mf

// `DPhysics.interpolate_grid` (dphysics.py:385-455) as an entry point of its own: one thread per query, the device functions of
// the rollout kernels (locate_m, gather4, blend: rollout_fwd_kernel.h / rollout_common.h) and the normal of their contact
// model.  Exact-arithmetic TU (-ffp-contract=off); the fast float32 variant is instantiated in interp_grid_fast.hip.
#include "interp_grid_kernel.h"

namespace mf {

template <typename S>
static int interpolate_grid(const MfInterpDesc* d, const S* grid, const S* xq, const S* yq, S* z_out, S* n_out, int32_t* cells,
                            S* frac, void* stream) {
  MF_REQUIRE(d && grid && xq && yq && z_out, MF_ERR_INVALID, "interpolate_grid: null argument");
  MF_REQUIRE(d->B > 0 && d->N > 0 && d->H >= 2 && d->W > 0, MF_ERR_INVALID, "interpolate_grid: B, N must be positive and the grid at least 2 x 2");
  MF_REQUIRE((long long)d->H * d->W < (1ll << 30) && d->H < (1 << 23), MF_ERR_UNSUPPORTED, "interpolate_grid: grid too large");
  MF_REQUIRE(d->map_shared || (long long)d->B * d->H * d->W * (long long)sizeof(S) < (1ll << 32), MF_ERR_UNSUPPORTED,
             "interpolate_grid: per-row maps of 4 GiB or more in total");
  MF_REQUIRE(d->math_mode == MF_MATH_EXACT || d->math_mode == MF_MATH_FAST, MF_ERR_INVALID, "interpolate_grid: unknown math_mode");
  InterpArgs<S> a;
  a.B = d->B; a.N = d->N; a.H = d->H; a.W = d->W; a.map_shared = d->map_shared;
  a.res = (S)d->grid_res; a.inv_res = (S)(1.0 / (double)(S)d->grid_res); a.d_max = (S)d->d_max;
  a.grid = grid; a.xq = xq; a.yq = yq; a.z = z_out; a.n = n_out; a.cells = cells; a.frac = frac;
  hipStream_t st = (hipStream_t)stream;
  if (sizeof(S) == 4 && d->math_mode == MF_MATH_FAST) launch_interp_fast_f32(*reinterpret_cast<const InterpArgs<float>*>(&a), st);
  else launch_interp<S, false>(a, st);
  hipError_t e = hipGetLastError();
  MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("interpolate_grid launch: ") + hipGetErrorString(e));
  return MF_OK;
}

}  // namespace mf

extern "C" int mf_interpolate_grid_f32(const MfInterpDesc* d, const float* grid, const float* xq, const float* yq, float* z, float* n,
                                       int32_t* cells, float* frac, void* s) {
  return mf::interpolate_grid<float>(d, grid, xq, yq, z, n, cells, frac, s);
}
extern "C" int mf_interpolate_grid_f64(const MfInterpDesc* d, const double* grid, const double* xq, const double* yq, double* z, double* n,
                                       int32_t* cells, double* frac, void* s) {
  return mf::interpolate_grid<double>(d, grid, xq, yq, z, n, cells, frac, s);
}

// estimate_heightmap (/root/reference/monoforce/src/monoforce/cloudproc.py:88-148) on the GPU: per-cell maximum height of a
// point cloud (scatter-amax) + measurement mask, the label-generation step of the reference's datasets.
// Three small launches on one stream: sentinel fill, one thread per point (filters, `torch.bucketize` semantics against the
// caller's bin edges, integer atomicMax on an order-preserving key), finalize (decode, transpose to [x][y], mask).
#include <math.h>

#include "mf_common.h"

namespace mf {

// float -> int key with the same ordering (for atomicMax on integers); INT_MIN is below every finite key
__device__ __forceinline__ int order_key(float v) {
  const int b = __builtin_bit_cast(int, v);
  return b >= 0 ? b : (b ^ 0x7FFFFFFF);
}
__device__ __forceinline__ float order_value(int k) { return __builtin_bit_cast(float, k >= 0 ? k : (k ^ 0x7FFFFFFF)); }

__global__ void __launch_bounds__(256) hm_fill_kernel(int* __restrict__ keys, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keys[i] = INT_MIN;
}

// torch.bucketize(x, bins) - 1 with right = False: (number of edges strictly below x) - 1
__device__ __forceinline__ int bucket(const float* __restrict__ bins, int n, float x, float inv_res) {
  int g = (int)floorf((x - bins[0]) * inv_res);
  g = min(max(g, 0), n - 1);
  while (g + 1 < n && bins[g + 1] < x) ++g;
  while (g >= 0 && !(bins[g] < x)) --g;
  return g;
}

__global__ void __launch_bounds__(256) hm_scatter_kernel(const MfHeightmapDesc d, const float* __restrict__ pts, const float* __restrict__ xb,
                                                         const float* __restrict__ yb, int* __restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.n_points) return;
  const float x = pts[i * 3 + 0], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
  if (isnan(x) || isnan(y) || isnan(z)) return;                      // cloudproc.py:90-91
  if (d.r_min >= 0.0f) {                                              // :93-97 (torch.norm accumulates a float row in double on the host)
    const float dist = (float)sqrt((double)x * (double)x + (double)y * (double)y);
    if (!(dist > d.r_min)) return;
  }
  if (!(x > -d.d_max && x < d.d_max && y > -d.d_max && y < d.d_max && z > d.h_min && z < d.h_max)) return;   // :102-105
  const int ix = bucket(xb, d.nx, x, d.inv_res), iy = bucket(yb, d.ny, y, d.inv_res);                        // :117-119
  if (ix < 0 || iy < 0) return;
  atomicMax(&keys[iy * d.nx + ix], order_key(z));                     // :122-133 scatter_reduce(amax)
}

// out[0][x][y] = max height (0 where nothing was measured), out[1][x][y] = 1 / 0 measurement mask   (:135-146, transposed)
__global__ void __launch_bounds__(256) hm_finalize_kernel(const MfHeightmapDesc d, const int* __restrict__ keys, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;      // output-major: i = x * ny + y
  if (i >= d.nx * d.ny) return;
  const int x = i / d.ny, y = i - x * d.ny;
  const int k = keys[y * d.nx + x];
  const bool got = k != INT_MIN;
  out[i] = got ? order_value(k) : 0.0f;
  out[d.nx * d.ny + i] = got ? 1.0f : 0.0f;
}

}  // namespace mf

extern "C" int mf_estimate_heightmap_f32(const MfHeightmapDesc* d, const float* points, const float* x_bins, const float* y_bins,
                                         int32_t* scratch, float* hm, void* stream) {
  MF_REQUIRE(d && x_bins && y_bins && scratch && hm, MF_ERR_INVALID, "estimate_heightmap: null argument");
  MF_REQUIRE(d->n_points >= 0 && d->nx > 0 && d->ny > 0 && (d->n_points == 0 || points), MF_ERR_INVALID, "estimate_heightmap: bad sizes");
  MF_REQUIRE((long long)d->nx * d->ny < (1ll << 30), MF_ERR_UNSUPPORTED, "estimate_heightmap: grid too large");
  hipStream_t st = (hipStream_t)stream;
  const int cells = d->nx * d->ny;
  hipLaunchKernelGGL(mf::hm_fill_kernel, dim3((cells + 255) / 256), dim3(256), 0, st, scratch, cells);
  if (d->n_points > 0)
    hipLaunchKernelGGL(mf::hm_scatter_kernel, dim3((d->n_points + 255) / 256), dim3(256), 0, st, *d, points, x_bins, y_bins, scratch);
  hipLaunchKernelGGL(mf::hm_finalize_kernel, dim3((cells + 255) / 256), dim3(256), 0, st, *d, scratch, hm);
  hipError_t e = hipGetLastError();
  MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("estimate_heightmap launch: ") + hipGetErrorString(e));
  return MF_OK;
}

// Terrain staging between the BEV heads and the rollout: in ONE pass over the head outputs
//   terrain = geom - diff                                  (lss.py:158, full resolution: the height-map loss reads it)
//   z  = AvgPool2d(k, k)(terrain),  mu = AvgPool2d(k, k)(friction)   (scripts/train.py:93-99, 233-235: coarser physics grid)
//   zmu[cell] = (z, mu)                                    (the interleaved map pair the rollout kernels gather from)
// instead of a subtraction, two pooling launches and the interleave pass.  The backward scatters the rollout's map gradients
// back through the pool and joins them with the height-map loss' gradient of `terrain`:
//   g_geom = g_terrain + up(gz) / k^2,  g_diff = -g_geom,  g_friction = up(gmu) / k^2.
// One thread per pooling window (windows that overhang the map only produce `terrain`, as AvgPool2d drops them).
#include "mf_common.h"

namespace mf {

__global__ void __launch_bounds__(256) terrain_stage_fwd_kernel(const MfStageDesc d, const float* __restrict__ geom, const float* __restrict__ diff,
                                                                const float* __restrict__ fric, float* __restrict__ terrain,
                                                                float* __restrict__ z, float* __restrict__ mu, float2* __restrict__ zmu) {
  const int hw = (d.H + d.k - 1) / d.k, ww = (d.W + d.k - 1) / d.k;     // windows incl. overhanging ones
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.B * hw * ww) return;
  const int b = i / (hw * ww), r = i - b * hw * ww, wi = r / ww, wj = r - wi * ww;
  const int h = d.H / d.k, w = d.W / d.k;                                // pooled size (floor)
  const size_t plane = (size_t)b * d.H * d.W;
  float sz = 0.0f, sm = 0.0f;
  for (int di = 0; di < d.k; ++di) {
    const int y = wi * d.k + di;
    if (y >= d.H) break;
    for (int dj = 0; dj < d.k; ++dj) {
      const int x = wj * d.k + dj;
      if (x >= d.W) break;
      const size_t c = plane + (size_t)y * d.W + x;
      const float t = geom[c] - diff[c];
      if (terrain) terrain[c] = t;
      sz += t;
      sm += fric[c];
    }
  }
  if (wi < h && wj < w) {
    const float inv = 1.0f / (float)(d.k * d.k);
    const size_t o = ((size_t)b * h + wi) * w + wj;
    const float zv = sz * inv, mv = sm * inv;
    z[o] = zv; mu[o] = mv;
    if (zmu) zmu[o] = make_float2(zv, mv);
  }
}

__global__ void __launch_bounds__(256) terrain_stage_bwd_kernel(const MfStageDesc d, const float* __restrict__ g_terrain, const float* __restrict__ gz,
                                                                const float* __restrict__ gmu, float* __restrict__ g_geom,
                                                                float* __restrict__ g_diff, float* __restrict__ g_fric) {
  const int hw = (d.H + d.k - 1) / d.k, ww = (d.W + d.k - 1) / d.k;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.B * hw * ww) return;
  const int b = i / (hw * ww), r = i - b * hw * ww, wi = r / ww, wj = r - wi * ww;
  const int h = d.H / d.k, w = d.W / d.k;
  const bool full = wi < h && wj < w;
  const float inv = 1.0f / (float)(d.k * d.k);
  const size_t o = ((size_t)b * h + (full ? wi : 0)) * w + (full ? wj : 0);
  const float pz = (full && gz) ? gz[o] * inv : 0.0f, pm = (full && gmu) ? gmu[o] * inv : 0.0f;
  const size_t plane = (size_t)b * d.H * d.W;
  for (int di = 0; di < d.k; ++di) {
    const int y = wi * d.k + di;
    if (y >= d.H) break;
    for (int dj = 0; dj < d.k; ++dj) {
      const int x = wj * d.k + dj;
      if (x >= d.W) break;
      const size_t c = plane + (size_t)y * d.W + x;
      const float g = (g_terrain ? g_terrain[c] : 0.0f) + pz;
      g_geom[c] = g;
      g_diff[c] = -g;
      g_fric[c] = pm;
    }
  }
}

}  // namespace mf

static int stage_check(const MfStageDesc* d) {
  MF_REQUIRE(d && d->B > 0 && d->H > 0 && d->W > 0 && d->k >= 1 && d->k <= d->H && d->k <= d->W, MF_ERR_INVALID, "terrain_stage: bad descriptor");
  MF_REQUIRE((long long)d->B * d->H * d->W < (1ll << 31), MF_ERR_UNSUPPORTED, "terrain_stage: maps too large");
  return MF_OK;
}

extern "C" int mf_terrain_stage_fwd_f32(const MfStageDesc* d, const float* geom, const float* diff, const float* friction, float* terrain,
                                        float* z, float* mu, float* zmu, void* stream) {
  int rc = stage_check(d);
  if (rc != MF_OK) return rc;
  MF_REQUIRE(geom && diff && friction && z && mu, MF_ERR_INVALID, "terrain_stage_fwd: null argument");
  MF_REQUIRE(((uintptr_t)zmu & 7) == 0, MF_ERR_INVALID, "terrain_stage_fwd: zmu must be 8-byte aligned");
  const int n = d->B * ((d->H + d->k - 1) / d->k) * ((d->W + d->k - 1) / d->k);
  hipLaunchKernelGGL(mf::terrain_stage_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, *d, geom, diff, friction, terrain, z, mu,
                     (float2*)zmu);
  hipError_t e = hipGetLastError();
  MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("terrain_stage_fwd launch: ") + hipGetErrorString(e));
  return MF_OK;
}

extern "C" int mf_terrain_stage_bwd_f32(const MfStageDesc* d, const float* g_terrain, const float* gz, const float* gmu, float* g_geom, float* g_diff,
                                        float* g_friction, void* stream) {
  int rc = stage_check(d);
  if (rc != MF_OK) return rc;
  MF_REQUIRE(g_geom && g_diff && g_friction, MF_ERR_INVALID, "terrain_stage_bwd: null gradient output");
  const int n = d->B * ((d->H + d->k - 1) / d->k) * ((d->W + d->k - 1) / d->k);
  hipLaunchKernelGGL(mf::terrain_stage_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, *d, g_terrain, gz, gmu, g_geom, g_diff,
                     g_friction);
  hipError_t e = hipGetLastError();
  MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("terrain_stage_bwd launch: ") + hipGetErrorString(e));
  return MF_OK;
}

// Torch-free client of the C ABI (include/monoforce_hip.h): a maintainer's smoke test of the drop-in boundary.
//   hipcc --offload-arch=gfx950 -O2 tools/c_abi_demo.cpp -Imonoforce_amd/../include -Lmonoforce_amd/csrc -lmonoforce_hip \
//         -Wl,-rpath,$PWD/monoforce_amd/csrc -o /tmp/c_abi_demo && /tmp/c_abi_demo
// Rolls 3 robots (4 contact points, 2 tracks) over a flat 32x32 map for 100 steps with the reference's default integrator and
// checks the obvious physics: they settle on the ground, drive forward at the commanded speed, the turning one turns.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "monoforce_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

template <typename T>
static T* to_device(const std::vector<T>& h) {
  T* d = nullptr;
  if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

int main() {
  const int B = 3, T = 100, N = 4, H = 32, W = 32;
  printf("%s\n", mf_version());
  MfRolloutDesc d;
  memset(&d, 0, sizeof d);
  d.B = B; d.T = T; d.N = N; d.H = H; d.W = W; d.n_tracks = 2;
  d.integrator = MF_INTEG_ODEINT_EULER; d.layout = MF_LAYOUT_BATCH_MAJOR; d.map_shared = 1; d.math_mode = MF_MATH_EXACT;
  d.mass = 40.0; d.gravity = 9.81; d.stiffness = 5e4; d.damping = sqrt(4 * 40.0 * 5e4); d.omega_max = 2.0;
  d.grid_res = 0.1; d.d_max = 1.6; d.dt = 0.01; d.robot_size_y = 0.54;
  // inverse inertia of 4 point masses of 10 kg at (+-0.25, +-0.27, -0.1)
  const double Ixx = 40 * (0.27 * 0.27 + 0.01), Iyy = 40 * (0.25 * 0.25 + 0.01), Izz = 40 * (0.25 * 0.25 + 0.27 * 0.27);
  d.Iinv[0] = 1 / Ixx; d.Iinv[4] = 1 / Iyy; d.Iinv[8] = 1 / Izz;
  d.force_stride = mf_rollout_force_stride(&d);
  if (d.force_stride != 4) { printf("unexpected force stride %d\n", d.force_stride); return 1; }

  std::vector<float> z(H * W, 0.0f), ctrl(B * T * 2), ts(T), pts = {0.25f, 0.27f, -0.1f, 0.25f, -0.27f, -0.1f, -0.25f, 0.27f, -0.1f, -0.25f, -0.27f, -0.1f};
  std::vector<int32_t> part = {0, 1, 0, 1};                              // left track = y > 0
  const float v[B] = {0.5f, 1.0f, 0.8f}, w[B] = {0.0f, 0.0f, 0.8f};
  for (int b = 0; b < B; ++b) for (int t = 0; t < T; ++t) { ctrl[(b * T + t) * 2] = v[b]; ctrl[(b * T + t) * 2 + 1] = w[b]; }
  for (int t = 0; t < T; ++t) ts[t] = 5.0f * t / 499.0f;                 // linspace(0, 5, 500)[:T]
  std::vector<float> x0(B * 3, 0.0f), xd0(B * 3, 0.0f), R0(B * 9, 0.0f), w0(B * 3, 0.0f);
  for (int b = 0; b < B; ++b) { xd0[b * 3] = v[b]; w0[b * 3 + 2] = w[b]; R0[b * 9] = R0[b * 9 + 4] = R0[b * 9 + 8] = 1.0f; }

  MfRolloutFwdBufs p;
  memset(&p, 0, sizeof p);
  float *dz = to_device(z), *dc = to_device(ctrl), *dts = to_device(ts), *dp = to_device(pts), *dx0 = to_device(x0), *dxd = to_device(xd0),
        *dR = to_device(R0), *dw = to_device(w0);
  int32_t* dpart = to_device(part);
  float *Xs, *Xds, *Rs, *Om, *Fs, *Ff;
  HIP_OK(hipMalloc(&Xs, B * T * 3 * 4)); HIP_OK(hipMalloc(&Xds, B * T * 3 * 4)); HIP_OK(hipMalloc(&Rs, B * T * 9 * 4));
  HIP_OK(hipMalloc(&Om, B * T * 3 * 4)); HIP_OK(hipMalloc(&Fs, B * T * N * 3 * 4)); HIP_OK(hipMalloc(&Ff, B * T * N * 3 * 4));
  p.z = dz; p.mu = nullptr; p.controls = dc; p.ts = dts; p.points = dp; p.part = dpart; p.x0 = dx0; p.xd0 = dxd; p.R0 = dR; p.w0 = dw;
  p.Xs = Xs; p.Xds = Xds; p.Rs = Rs; p.Omegas = Om; p.Fs = Fs; p.Ff = Ff;

  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  int rc = mf_rollout_fwd_f32(&d, &p, st);
  if (rc != MF_OK) { printf("mf_rollout_fwd_f32 failed (%d): %s\n", rc, mf_last_error()); return 1; }
  HIP_OK(hipStreamSynchronize(st));
  std::vector<float> hX(B * T * 3), hR(B * T * 9);
  HIP_OK(hipMemcpy(hX.data(), Xs, hX.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hR.data(), Rs, hR.size() * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  const float t_end = ts[T - 1];
  for (int b = 0; b < B; ++b) {
    const float* xe = &hX[(b * T + T - 1) * 3];
    const float yaw = atan2f(hR[(b * T + T - 1) * 9 + 3], hR[(b * T + T - 1) * 9]);
    printf("rollout %d: v=%.1f w=%.1f -> x=(%.3f, %.3f, %.3f) yaw=%.3f after %.2f s\n", b, v[b], w[b], xe[0], xe[1], xe[2], yaw, t_end);
    if (!(xe[0] == xe[0]) || fabsf(xe[2] - 0.1f) > 0.03f) ++bad;                        // rests on the ground: body origin 0.1 m above it
    if (w[b] == 0.0f && (fabsf(xe[0] - v[b] * t_end) > 0.15f * v[b] * t_end + 0.02f || fabsf(xe[1]) > 1e-3f)) ++bad;
    if (w[b] != 0.0f && !(yaw > 0.2f && xe[1] > 0.01f)) ++bad;                            // positive yaw rate turns left
  }
  // a deliberately bad descriptor is rejected with a message, not a crash
  d.n_tracks = 3;
  if (mf_rollout_fwd_f32(&d, &p, st) != MF_ERR_INVALID || !strstr(mf_last_error(), "n_tracks")) ++bad;
  printf(bad ? "FAILED (%d checks)\n" : "C ABI demo ok\n", bad);
  return bad ? 1 : 0;
}

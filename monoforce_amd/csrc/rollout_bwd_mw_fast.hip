// Backward rollout of one rollout over several waves (rollout_bwd_mw_kernel.h): float32 fast-math instantiations and the rules
// that send a launch to them.
#include <cstdlib>
#include "rollout_bwd_mw_kernel.h"

namespace mf {

// The launches the multi-wave mapping of choose_lane_map serves (a body of 65..512 points, <= 2048 waves), float32 MF_MATH_FAST,
// default integrator, rigid body.  MF_MW_BWD=0 keeps the general kernel (A/B runs, parity tests of the two against each other).
static bool mw_shape(const MfRolloutDesc* d) {
  static const bool off = getenv("MF_MW_BWD") && atoi(getenv("MF_MW_BWD")) == 0;
  if (off || !d || d->B <= 0 || d->T <= 0 || d->N <= 64 || d->N > 512) return false;
  if (d->math_mode != MF_MATH_FAST || d->integrator != MF_INTEG_ODEINT_EULER || d->has_joints) return false;
  if (d->points_per_lane == 4) return false;
  return choose_lane_map(d->B, d->N, d->points_per_lane == MF_LANES_COMPONENT ? 0 : d->points_per_lane).G > 64;
}
long long mw_record_bytes(const MfRolloutDesc* d) {
  if (!mw_shape(d)) return 0;
  return (long long)d->T * d->B * kMwRecFloats * (long long)sizeof(float);
}
bool use_multiwave_bwd(const MfRolloutDesc* d, const MfRolloutBwdBufs* p) {
  return mw_shape(d) && p->rec != nullptr && !p->joint_angles && !p->loss;
}

int launch_rollout_bwd_mw_f32(const RolloutBwdArgs<float>& a, int G, bool xs_only, hipStream_t st) {
  bool launched = false;
#define MF_CASE(G_)                                                                                              \
  if (!launched && G == G_) {                                                                                    \
    launched = true;                                                                                             \
    if (xs_only) hipLaunchKernelGGL((rollout_bwd_mw_kernel<G_, true>), dim3(a.B), dim3(G_), 0, st, a);           \
    else hipLaunchKernelGGL((rollout_bwd_mw_kernel<G_, false>), dim3(a.B), dim3(G_), 0, st, a);                  \
  }
  MF_CASE(128) MF_CASE(256) MF_CASE(512)
#undef MF_CASE
  MF_REQUIRE(launched, MF_ERR_UNSUPPORTED, "rollout_bwd: no multi-wave kernel for this lane mapping");
  hipError_t e = hipGetLastError();
  MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("rollout_bwd (multi-wave) launch: ") + hipGetErrorString(e));
  return MF_OK;
}

}  // namespace mf

// Backward rollout of one rollout over several waves (rollout_bwd_mw_kernel.h): float32 fast-math instantiations and the rules
// that send a launch to them.
#include <cstdlib>
#include "rollout_bwd_mw_kernel.h"

namespace mf {

// The launches these kernels serve: float32 MF_MATH_FAST, either integrator, rigid body, one point per lane --
//   * bodies of 65..512 points spread over 2 / 4 / 8 waves by choose_lane_map (<= 2048 waves per launch), and
//   * bodies of 5..64 points (8 / 16 / 32 / 64 lanes per rollout) up to two waves per SIMD (the positions-only instantiations hold
//     237 registers; beyond, the forward goes out in chunks and the general kernels take over).  N <= 4 has the component-parallel kernels.
// MF_MW_BWD=0 keeps the general kernel (A/B runs, parity tests of the two against each other).
static bool mw_shape(const MfRolloutDesc* d) {
  static const bool off = getenv("MF_MW_BWD") && atoi(getenv("MF_MW_BWD")) == 0;
  if (off || !d || d->B <= 0 || d->T <= 0 || d->N <= 4 || d->N > 512) return false;
  if (d->math_mode != MF_MATH_FAST || d->has_joints) return false;
  if (d->integrator != MF_INTEG_ODEINT_EULER && d->integrator != MF_INTEG_DYNAMICS) return false;
  if (d->points_per_lane == 4) return false;
  const LaneMap m = choose_lane_map(d->B, d->N, d->points_per_lane == MF_LANES_COMPONENT ? 0 : d->points_per_lane);
  if (m.PPL != 1 || m.G < 8) return false;
  return m.G > 64 || (long long)d->B * m.G <= 2 * device_simds() * 64;      // two waves per SIMD
}
long long mw_record_bytes(const MfRolloutDesc* d, int scalar_bytes) {
  if (!mw_shape(d)) return 0;
  return (long long)d->T * d->B * kMwRecFloats * (long long)scalar_bytes;
}
bool use_multiwave_bwd(const MfRolloutDesc* d, const MfRolloutBwdBufs* p) {
  return mw_shape(d) && p->rec != nullptr && !p->joint_angles && !p->loss;
}

int launch_rollout_bwd_mw_f32(const RolloutBwdArgs<float>& a, int G, int integ, bool xs_only, hipStream_t st) {
  return launch_rollout_bwd_mw_t<float>(a, G, integ, xs_only, st);
}

}  // namespace mf

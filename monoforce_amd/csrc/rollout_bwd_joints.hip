// Backward rollout of an articulated body (non-zero flipper joint angles; update_joints, dphysics.py:326-358): exact-arithmetic
// instantiations for float32 and float64, default lane mappings.  The joint angles are constants of the rollout.
#include "rollout_bwd_kernel.h"

namespace mf {
int launch_rollout_bwd_joints_f32(const RolloutBwdArgs<float>& a, LaneMap m, int integ, int block, hipStream_t st) {
  return launch_rollout_bwd<float, false, true>(a, m, integ, block, st);
}
int launch_rollout_bwd_joints_f64(const RolloutBwdArgs<double>& a, LaneMap m, int integ, int block, hipStream_t st) {
  return launch_rollout_bwd<double, false, true>(a, m, integ, block, st);
}
}  // namespace mf

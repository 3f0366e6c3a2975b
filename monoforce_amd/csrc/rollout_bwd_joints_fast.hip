// Backward rollout of an articulated body (flipper joint angles, robot 'marv'), float32 fast-math instantiations.
#include "rollout_bwd_kernel.h"

namespace mf {
int launch_rollout_bwd_joints_fast_f32(const RolloutBwdArgs<float>& a, LaneMap m, int integ, int block, hipStream_t st) {
  return launch_rollout_bwd<float, true, true>(a, m, integ, block, st);
}
}  // namespace mf

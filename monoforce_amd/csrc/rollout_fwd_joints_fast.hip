// Forward rollout of an articulated body (flipper joint angles, robot 'marv'), float32 fast-math instantiations.
#include "rollout_fwd_kernel.h"

namespace mf {
int launch_rollout_fwd_joints_fast_f32(const RolloutArgs<float>& a, LaneMap m, int integ, int block, hipStream_t st) {
  return launch_rollout_fwd<float, true, true>(a, m, integ, block, st);
}
}  // namespace mf

// Forward rollout, path-cost instantiations (float32 fast math): 16-byte cost rows + decimated poses instead of the full output
// rows -- the trajectory-shooting epilogue of SURVEY 8f row 1 (monoforce_node.py:91, diff_physics.py:263-266).
#include "rollout_fwd_kernel.h"

namespace mf {
int launch_rollout_fwd_cost_f32(const RolloutArgs<float>& a, LaneMap m, int integ, int block, bool project, hipStream_t st) {
  // `project`: the rows carry the third row of the nearest rotation (roll / pitch costs); only the explicit-Euler
  // integrator lets R drift, so dynamics() has no separate instantiation
  if (project && integ == MF_INTEG_ODEINT_EULER) return launch_rollout_fwd<float, true, false, false, 2>(a, m, integ, block, st);
  return launch_rollout_fwd<float, true, false, false, 1>(a, m, integ, block, st);
}
}  // namespace mf

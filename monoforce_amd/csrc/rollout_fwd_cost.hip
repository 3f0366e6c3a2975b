// Forward rollout, path-cost instantiations (float32 fast math): 16-byte cost rows + decimated poses instead of the full output
// rows -- the trajectory-shooting epilogue of SURVEY 8f row 1 (monoforce_node.py:91, diff_physics.py:263-266).
#include "rollout_fwd_kernel.h"

namespace mf {
int launch_rollout_fwd_cost_f32(const RolloutArgs<float>& a, LaneMap m, int integ, int block, hipStream_t st) {
  return launch_rollout_fwd<float, true, false, false, true>(a, m, integ, block, st);
}
}  // namespace mf

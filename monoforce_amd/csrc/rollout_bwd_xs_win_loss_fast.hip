// Backward rollout, float32 fast math: the LDS-window positions-only kernels of rollout_bwd_xs_win_fast.hip with `physics_loss`
// (losses.py:102-127) inside the launch (rollout_bwd_kernel.h LOSS) -- the saturated launches of the 4-point body in a fit / train step.
#include "rollout_bwd_kernel.h"

namespace mf {
int launch_rollout_bwd_xs_win_loss_fast_f32(const RolloutBwdArgs<float>& a, LaneMap m, int integ, int block, bool zmu, bool carry, hipStream_t st) {
  if (carry) return zmu ? launch_rollout_bwd_xs<float, true, true, true, true>(a, m, integ, block, st) : launch_rollout_bwd_xs<float, false, true, true, true>(a, m, integ, block, st);
  return zmu ? launch_rollout_bwd_xs<float, true, true, false, true>(a, m, integ, block, st) : launch_rollout_bwd_xs<float, false, true, false, true>(a, m, integ, block, st);
}
}  // namespace mf

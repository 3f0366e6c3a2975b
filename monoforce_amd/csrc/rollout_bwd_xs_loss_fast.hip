// Backward rollout, float32 fast math: the positions-only one-point-per-lane kernels of rollout_bwd_xs_fast.hip with `physics_loss`
// (losses.py:102-127) inside the launch (rollout_bwd_kernel.h LOSS): dL/dXs is formed at the stamped rows from the forward's own Xs rows,
// the ground truth and the stamp tables -- a saturated fit / train step loses its loss-gradient launch and the dense [T][B][3] gradient.
#include "rollout_bwd_kernel.h"

namespace mf {
int launch_rollout_bwd_xs_loss_fast_f32(const RolloutBwdArgs<float>& a, LaneMap m, int integ, int block, bool zmu, hipStream_t st) {
  return zmu ? launch_rollout_bwd_xs<float, true, false, true, true>(a, m, integ, block, st) : launch_rollout_bwd_xs<float, false, false, true, true>(a, m, integ, block, st);
}
}  // namespace mf

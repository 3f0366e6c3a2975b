// Backward rollout, float32 fast math: the positions-only instantiations of the general one-point-per-lane kernel whose cell gradients
// go through a 128 x 128-cell LDS window per workgroup (rollout_bwd_kernel.h WIN) -- four lanes per rollout, shared map pair, plain or
// interleaved.  The saturated launches of the 4-point body.
#include "rollout_bwd_kernel.h"

namespace mf {
int launch_rollout_bwd_xs_win_fast_f32(const RolloutBwdArgs<float>& a, LaneMap m, int integ, int block, bool zmu, hipStream_t st) {
  return zmu ? launch_rollout_bwd_xs<float, true, true>(a, m, integ, block, st) : launch_rollout_bwd_xs<float, false, true>(a, m, integ, block, st);
}
}  // namespace mf

// Backward rollout, float32 fast math: the positions-only instantiations of the general one-point-per-lane kernel whose cell gradients
// go through a 128 x 128-cell LDS window per workgroup (rollout_bwd_kernel.h WIN) -- four lanes per rollout, shared map pair, plain or
// interleaved.  The saturated launches of the 4-point body.  With the window the accumulator carry-over (half the cell writes for ~55
// instructions per step) pays from two waves per SIMD up only: 16 384 rollouts 0.991 -> 0.949 ms without it, 32 768: 1.406 -> 1.434.
#include "rollout_bwd_kernel.h"

namespace mf {
int launch_rollout_bwd_xs_win_fast_f32(const RolloutBwdArgs<float>& a, LaneMap m, int integ, int block, bool zmu, bool carry, hipStream_t st) {
  if (carry) return zmu ? launch_rollout_bwd_xs<float, true, true, true>(a, m, integ, block, st) : launch_rollout_bwd_xs<float, false, true, true>(a, m, integ, block, st);
  return zmu ? launch_rollout_bwd_xs<float, true, true, false>(a, m, integ, block, st) : launch_rollout_bwd_xs<float, false, true, false>(a, m, integ, block, st);
}
}  // namespace mf

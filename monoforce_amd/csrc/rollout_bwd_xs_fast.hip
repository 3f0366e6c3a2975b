// Backward rollout, float32 fast math: the positions-only (dL/dXs is the only upstream gradient) instantiations of the general
// one-point-per-lane kernel with accumulator carry-over, reading a shared map pair plain or interleaved (rollout_bwd_kernel.h XS_ONLY /
// ZMU) -- the saturated launches of <= 4-point bodies, where the CU's L1 address path is the bound.
#include "rollout_bwd_kernel.h"

namespace mf {
int launch_rollout_bwd_xs_fast_f32(const RolloutBwdArgs<float>& a, LaneMap m, int integ, int block, bool zmu, hipStream_t st) {
  return zmu ? launch_rollout_bwd_xs<float, true>(a, m, integ, block, st) : launch_rollout_bwd_xs<float, false>(a, m, integ, block, st);
}
}  // namespace mf

// Backward rollout, float32 fast-math instantiations with the accumulator carry-over between adjacent cells (half the
// map-gradient atomics; chosen by the host once a launch has enough waves for the atomics to matter).
#include "rollout_bwd_kernel.h"

namespace mf {
int launch_rollout_bwd_carry_fast_f32(const RolloutBwdArgs<float>& a, LaneMap m, int integ, int block, hipStream_t st) {
  return launch_rollout_bwd<float, true, false, true>(a, m, integ, block, st);
}
}  // namespace mf

// Backward rollout, float32 fast-math instantiations (-ffp-contract=fast + hardware rcp / rsq / exp2).
#include "rollout_bwd_kernel.h"

namespace mf {
int launch_rollout_bwd_fast_f32(const RolloutBwdArgs<float>& a, LaneMap m, int integ, int block, hipStream_t st) {
  return launch_rollout_bwd<float, true, false, false>(a, m, integ, block, st);   // plain flush: fewest instructions per step
}
}  // namespace mf

// Forward rollout, float32 fast-math instantiations with the state stores split over the lanes of a group (SPLIT): chosen by the
// host once a launch has a wave for every SIMD, where the number of memory instructions per step bounds the kernel.
#include "rollout_fwd_kernel.h"

namespace mf {
int launch_rollout_fwd_split_fast_f32(const RolloutArgs<float>& a, LaneMap m, int integ, int block, bool forces, hipStream_t st) {
  if (!forces) return launch_rollout_fwd<float, true, false, false, 0, true>(a, m, integ, block, st);
  return launch_rollout_fwd<float, true, false, true, 0, true>(a, m, integ, block, st);
}
}  // namespace mf

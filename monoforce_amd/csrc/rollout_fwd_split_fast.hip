// Forward rollout, float32 fast-math instantiations with the state stores split over the lanes of a group (SPLIT): chosen by the
// host once a launch has a wave for every SIMD, where the number of memory instructions per step bounds the kernel.
#include "rollout_fwd_kernel.h"

namespace mf {
int launch_rollout_fwd_split_fast_f32(const RolloutArgs<float>& a, LaneMap m, int integ, int block, bool forces, hipStream_t st) {
  if (a.rec != nullptr && m.G >= 8 && m.G <= 64 && m.PPL == 1)      // the record of rollout_bwd_mw_kernel.h
    return forces ? launch_rollout_fwd_mw_rec<true, false, true>(a, m, integ, st) : launch_rollout_fwd_mw_rec<false, false, true>(a, m, integ, st);
  if (!forces) return launch_rollout_fwd<float, true, false, false, 0, true>(a, m, integ, block, st);
  return launch_rollout_fwd<float, true, false, true, 0, true>(a, m, integ, block, st);
}
}  // namespace mf

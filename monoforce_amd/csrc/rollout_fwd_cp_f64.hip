// Forward rollout, component-parallel lane mapping (rollout_fwd_cp_kernel.h): the float64 VALIDATION build of the kernels every
// BASELINE configuration runs -- the same source as rollout_fwd_cp_fast.hip's float32 kernels, instantiated on double with exact
// arithmetic (Mth<double, false>), so that tests/ can hold the lane algebra, the two-stream pipeline, the record and the fused
// loss to the float64 oracle over the full 500-step horizon, where float32 trajectories are chaotic.  Selected only explicitly
// (MfRolloutDesc.points_per_lane = MF_LANES_COMPONENT with the _f64 entry points); speed is irrelevant here.
#include "rollout_fwd_cp_kernel.h"

namespace mf {

int launch_rollout_fwd_cp_f64(const RolloutArgs<double>& a, int integ, bool forces, bool zmu, hipStream_t st) {
  return launch_rollout_fwd_cp_t<double>(a, integ, forces, zmu, st);
}

}  // namespace mf

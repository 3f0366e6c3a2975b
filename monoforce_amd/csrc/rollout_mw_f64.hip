// The float64 VALIDATION build of the kernels that serve bodies of 5..512 contact points at small batch sizes -- the reference's own
// operating point (examples/diff_physics.ipynb:199-226: 64 rollouts x 223 points): the recording one-point-per-lane forward
// (rollout_fwd_kernel.h with FAST's structure: software-pipelined one-exchange step, transposed workgroup exchange, 4-scalar record)
// and the record-reading backward (rollout_bwd_mw_kernel.h), instantiated on double with exact arithmetic, so that tests/ can hold
// them to the float64 oracle without a float32 error floor.  Selected only explicitly (points_per_lane = MF_LANES_COMPONENT with the
// _f64 entry points and a record buffer of mf_rollout_record_bytes_f64 bytes); speed is irrelevant here.
#include "rollout_bwd_mw_kernel.h"
#include "rollout_fwd_kernel.h"

namespace mf {

int launch_rollout_fwd_mw_rec_f64(const RolloutArgs<double>& a, LaneMap m, int integ, bool forces, hipStream_t st) {
  if (forces) return launch_rollout_fwd_mw_rec<true, false, false, double>(a, m, integ, st);
  return launch_rollout_fwd_mw_rec<false, false, false, double>(a, m, integ, st);
}

int launch_rollout_bwd_mw_f64(const RolloutBwdArgs<double>& a, int G, int integ, bool xs_only, hipStream_t st) {
  return launch_rollout_bwd_mw_t<double>(a, G, integ, xs_only, st);
}

}  // namespace mf

// Backward rollout, component-parallel lane mapping, `dynamics()` integrator: the float64 VALIDATION build (recompute early / late,
// record read by the computing wave; the streaming form's sixteen / eighteen float64 planes per slot exceed a CU's LDS).  See
// rollout_fwd_cp_f64.hip.
#include "rollout_bwd_cp_kernel.h"

namespace mf {

int launch_rollout_bwd_cp_dynamics_f64(const RolloutBwdArgs<double>& a, bool xs_only, hipStream_t st) {
  return launch_rollout_bwd_cp_variant<double, MF_INTEG_DYNAMICS>(a, xs_only, st);
}

}  // namespace mf

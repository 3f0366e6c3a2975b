// Backward rollout, component-parallel lane mapping, `dynamics()` integrator (semi-implicit Euler + Rodrigues rotation update):
// its own translation unit so the eight variants of each integrator compile side by side.
#include "rollout_bwd_cp_kernel.h"

namespace mf {

int launch_rollout_bwd_cp_dynamics_f32(const RolloutBwdArgs<float>& a, bool xs_only, hipStream_t st) {
  return launch_rollout_bwd_cp_variant<float, MF_INTEG_DYNAMICS>(a, xs_only, st);
}

}  // namespace mf

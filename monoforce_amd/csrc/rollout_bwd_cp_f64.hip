// Backward rollout, component-parallel lane mapping (rollout_bwd_cp_kernel.h): the float64 VALIDATION build, default integrator --
// the same source as the float32 kernels of rollout_bwd_cp_fast.hip / rollout_bwd_cp_stream_fast.hip instantiated on double (exact
// arithmetic): recompute (early / late), record read by the computing wave, and the record STREAMED through a six-slot LDS ring by
// two more waves (a float64 slot is twice the bytes: 123 / 147 KB, one workgroup per CU).  See rollout_fwd_cp_f64.hip.
#include "rollout_bwd_cp_kernel.h"

namespace mf {

void launch_rollout_bwd_cp_stream_f64(const RolloutBwdArgs<double>& a, bool xs_only, unsigned grid, hipStream_t st) {
  constexpr int I = MF_INTEG_ODEINT_EULER;
  const bool gc = a.gcontrols != nullptr;
#define MF_BCPS(XS_, GC_) MF_KLAUNCH((rollout_bwd_cp_kernel<double, I, XS_, GC_, kCpStream, 6, (XS_ ? 2 : 3)>), dim3(grid), dim3(192), 0, st, a)
  if (xs_only) { if (gc) MF_BCPS(true, true); else MF_BCPS(true, false); }
  else         { if (gc) MF_BCPS(false, true); else MF_BCPS(false, false); }
#undef MF_BCPS
}

int launch_rollout_bwd_cp_dynamics_f64(const RolloutBwdArgs<double>& a, bool xs_only, hipStream_t st);      // rollout_bwd_dyn_cp_f64.hip
int launch_rollout_bwd_cp_f64(const RolloutBwdArgs<double>& a, int integ, bool xs_only, hipStream_t st) {
  if (integ == MF_INTEG_DYNAMICS) return launch_rollout_bwd_cp_dynamics_f64(a, xs_only, st);
  return launch_rollout_bwd_cp_variant<double, MF_INTEG_ODEINT_EULER>(a, xs_only, st);
}

}  // namespace mf

// Backward rollout, component-parallel lane mapping, `dynamics()` integrator, from the forward's record STREAMED through LDS by two
// more waves of the workgroup (MODE = kCpStream): the fetching waves also prepare the adjoint-independent half of the Rodrigues
// step's backward (struct Coef, "CoefD" part), the computing wave runs the adjoint recurrence on 16 / 18 planes of coefficients.
#include "rollout_bwd_cp_kernel.h"

namespace mf {

// six-slot ring (96 / 108 KB of LDS: one workgroup per CU, B <= 1024)
void launch_rollout_bwd_cp_stream_dynamics_f32(const RolloutBwdArgs<float>& a, bool xs_only, unsigned grid, hipStream_t st) {
  constexpr int I = MF_INTEG_DYNAMICS;
  const bool gc = a.gcontrols != nullptr;
#define MF_BCPS(XS_, GC_) MF_KLAUNCH((rollout_bwd_cp_kernel<float, I, XS_, GC_, kCpStream, 6>), dim3(grid), dim3(192), 0, st, a)
  if (xs_only) { if (gc) MF_BCPS(true, true); else MF_BCPS(true, false); }
  else         { if (gc) MF_BCPS(false, true); else MF_BCPS(false, false); }
#undef MF_BCPS
}

}  // namespace mf

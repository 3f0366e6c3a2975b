// Backward rollout, component-parallel lane mapping, default integrator, from the forward's per-step record STREAMED through LDS by
// a second wave of the workgroup (MODE = kCpStream, rollout_bwd_cp_kernel.h): a translation unit of its own -- four kernels, and
// the only ones whose computing loop has no memory latency of its own to schedule around.
#include "rollout_bwd_cp_kernel.h"

namespace mf {

void launch_rollout_bwd_cp_stream_f32(const RolloutBwdArgs<float>& a, bool xs_only, unsigned grid, hipStream_t st) {
  constexpr int I = MF_INTEG_ODEINT_EULER;
  const bool gc = a.gcontrols != nullptr;
#define MF_BCPS(XS_, GC_) hipLaunchKernelGGL((rollout_bwd_cp_kernel<I, XS_, GC_, kCpStream>), dim3(grid), dim3(128), 0, st, a)
  if (xs_only) { if (gc) MF_BCPS(true, true); else MF_BCPS(true, false); }
  else         { if (gc) MF_BCPS(false, true); else MF_BCPS(false, false); }
#undef MF_BCPS
}

}  // namespace mf

// Backward rollout, component-parallel lane mapping, default integrator, from the forward's per-step record STREAMED through LDS by
// a second wave of the workgroup (MODE = kCpStream, rollout_bwd_cp_kernel.h): a translation unit of its own -- eight kernels, and
// the only ones whose computing loop has no memory latency of its own to schedule around.
#include "rollout_bwd_cp_kernel.h"

namespace mf {

// Ring size by launch size: eight slots while a CU holds one workgroup (B <= 1024: 80 / 96 KB of its 160 KB LDS), six for two
// workgroups per CU (60 / 72 KB each).
void launch_rollout_bwd_cp_stream_f32(const RolloutBwdArgs<float>& a, bool xs_only, unsigned grid, hipStream_t st) {
  constexpr int I = MF_INTEG_ODEINT_EULER;
  const bool gc = a.gcontrols != nullptr;
  static const unsigned big_ring_max = getenv("MF_CP_STREAM_BIG_RING_MAX_GRID") ? (unsigned)atoi(getenv("MF_CP_STREAM_BIG_RING_MAX_GRID")) : 256u;
#define MF_BCPS(XS_, GC_) do { if (grid <= big_ring_max) hipLaunchKernelGGL((rollout_bwd_cp_kernel<I, XS_, GC_, kCpStream, 8>), dim3(grid), dim3(128), 0, st, a); \
                               else hipLaunchKernelGGL((rollout_bwd_cp_kernel<I, XS_, GC_, kCpStream, 6>), dim3(grid), dim3(128), 0, st, a); } while (0)
  if (xs_only) { if (gc) MF_BCPS(true, true); else MF_BCPS(true, false); }
  else         { if (gc) MF_BCPS(false, true); else MF_BCPS(false, false); }
#undef MF_BCPS
}

}  // namespace mf

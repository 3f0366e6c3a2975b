// Forward rollout, component-parallel lane mapping with the step's two instruction streams on two waves
// (rollout_fwd_cp2_kernel.h): float32 fast-math instantiations, default integrator.  Built like the other *_fast units.
#include "rollout_fwd_cp2_kernel.h"

namespace mf {

// Workgroups (= 4 rollouts = two waves) up to which the two-wave kernel replaces the one-wave one: while every wave still finds a
// SIMD of its own would be <= 512; the default is 0 = never: the kernel is an experiment that did not pay (rollout_fwd_cp2_kernel.h).
static long long cp2_max_wgs() {
  static const long long v = getenv("MF_CP2_MAX_WGS") ? atoll(getenv("MF_CP2_MAX_WGS")) : 0;
  return v;
}

bool use_two_wave_forward(const MfRolloutDesc* d) {
  return d->integrator == MF_INTEG_ODEINT_EULER && ((long long)d->B + 3) / 4 <= cp2_max_wgs();
}

int launch_rollout_fwd_cp2_f32(const RolloutArgs<float>& a, bool forces, bool zmu, hipStream_t st) {
  const unsigned grid = (unsigned)(((long long)a.B + 3) / 4);
  const bool rec = a.rec != nullptr;
#define MF_CP2(FORCES_, ZMU_) do { if (rec) hipLaunchKernelGGL((rollout_fwd_cp2_kernel<FORCES_, ZMU_, true>), dim3(grid), dim3(128), 0, st, a); \
                                   else hipLaunchKernelGGL((rollout_fwd_cp2_kernel<FORCES_, ZMU_, false>), dim3(grid), dim3(128), 0, st, a); } while (0)
  if (forces) { if (zmu) MF_CP2(true, true); else MF_CP2(true, false); }
  else        { if (zmu) MF_CP2(false, true); else MF_CP2(false, false); }
#undef MF_CP2
  hipError_t e = hipGetLastError();
  MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("rollout_fwd (component-parallel, two waves) launch: ") + hipGetErrorString(e));
  return MF_OK;
}

}  // namespace mf

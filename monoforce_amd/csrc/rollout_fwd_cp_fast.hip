// Forward rollout, component-parallel lane mapping (rollout_fwd_cp_kernel.h): float32 fast-math instantiations and the
// host-side choice between this mapping and the one-point-per-lane kernels.  Built with FMA contraction like the other
// *_fast units.
#include "rollout_fwd_cp_kernel.h"

namespace mf {

// Waves of a component-parallel launch up to which it beats the one-point-per-lane mapping (4 rollouts per wave here, 16 there).
// Measured, forward with all six outputs, N = 4 (tools/ab_cp.py; ms component-parallel vs one point per lane): B = 256 0.16 / 0.29,
// 1024 0.16 / 0.29, 2048 0.17 / 0.29, 4096 0.18 / 0.30 (43 % of the HBM roofline), 8192 0.41 / 0.32 -- so up to 1024 waves, one per
// SIMD.  MF_CP_MAX_WAVES overrides (tuning / A-B runs; 0 disables the mapping).
static long long cp_max_waves() {
  static const long long v = getenv("MF_CP_MAX_WAVES") ? atoll(getenv("MF_CP_MAX_WAVES")) : -1;
  return v >= 0 ? v : device_simds();      // one wave per SIMD (MI355X: 1024)
}

bool use_component_parallel(const MfRolloutDesc* d, const MfRolloutFwdBufs* p, int scalar_bytes) {
  if (d->math_mode != MF_MATH_FAST || d->N > 4 || p->joint_angles || p->cost_rows) return false;
  if (d->points_per_lane != 0 && d->points_per_lane != MF_LANES_COMPONENT) return false;   // an explicit other mapping
  const long long waves = ((long long)d->B + 3) / 4;
  if (d->points_per_lane == 0 && waves > cp_max_waves()) return false;
  // 32-bit byte offsets into every output and into the controls
  const long long fs = d->force_stride ? d->force_stride : d->N;
  const long long row = fs * 3 > 9 ? fs * 3 : 9;
  if ((long long)d->T * d->B * row * scalar_bytes >= (1ll << 32)) return false;
  if (fs < 4) return false;   // quads of absent points write their (zero) slots
  return true;
}

int launch_rollout_fwd_cp_f32(const RolloutArgs<float>& a, int integ, bool forces, bool zmu, hipStream_t st) {
  return launch_rollout_fwd_cp_t<float>(a, integ, forces, zmu, st);
}

}  // namespace mf

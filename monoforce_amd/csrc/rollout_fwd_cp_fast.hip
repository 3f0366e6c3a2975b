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
  static const long long v = getenv("MF_CP_MAX_WAVES") ? atoll(getenv("MF_CP_MAX_WAVES")) : 1024;
  return v;
}

bool use_component_parallel(const MfRolloutDesc* d, const MfRolloutFwdBufs* p) {
  if (d->math_mode != MF_MATH_FAST || d->N > 4 || p->joint_angles || p->cost_rows) return false;
  if (d->points_per_lane != 0 && d->points_per_lane != MF_LANES_COMPONENT) return false;   // an explicit other mapping
  const long long waves = ((long long)d->B + 3) / 4;
  if (d->points_per_lane == 0 && waves > cp_max_waves()) return false;
  // 32-bit byte offsets into every output and into the controls
  const long long fs = d->force_stride ? d->force_stride : d->N;
  const long long row = fs * 3 > 9 ? fs * 3 : 9;
  if ((long long)d->T * d->B * row * 4 >= (1ll << 32)) return false;
  if (fs < 4) return false;   // quads of absent points write their (zero) slots
  return true;
}

int launch_rollout_fwd_cp_f32(const RolloutArgs<float>& a, int integ, bool forces, bool zmu, hipStream_t st) {
  const int block = 64;   // one wave = 4 rollouts per workgroup: B = 1024 puts one wave on each of the 256 CUs
  const long long threads = (long long)a.B * 16;
  const unsigned grid = (unsigned)((threads + block - 1) / block);
  const bool rec = a.rec != nullptr;
  if (a.loss_gt) {      // fused physics loss: default integrator, states only (the host checked)
    constexpr int I = MF_INTEG_ODEINT_EULER;
    if (rec) { if (zmu) hipLaunchKernelGGL((rollout_fwd_cp_kernel<I, false, true, true, true>), dim3(grid), dim3(block), 0, st, a);
               else hipLaunchKernelGGL((rollout_fwd_cp_kernel<I, false, false, true, true>), dim3(grid), dim3(block), 0, st, a); }
    else     { if (zmu) hipLaunchKernelGGL((rollout_fwd_cp_kernel<I, false, true, false, true>), dim3(grid), dim3(block), 0, st, a);
               else hipLaunchKernelGGL((rollout_fwd_cp_kernel<I, false, false, false, true>), dim3(grid), dim3(block), 0, st, a); }
    hipError_t e = hipGetLastError();
    MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("rollout_fwd (component-parallel, fused loss) launch: ") + hipGetErrorString(e));
    return MF_OK;
  }
#define MF_CP(INTEG_, FORCES_, ZMU_) do { if (rec) hipLaunchKernelGGL((rollout_fwd_cp_kernel<INTEG_, FORCES_, ZMU_, true>), dim3(grid), dim3(block), 0, st, a); \
                                          else hipLaunchKernelGGL((rollout_fwd_cp_kernel<INTEG_, FORCES_, ZMU_, false>), dim3(grid), dim3(block), 0, st, a); } while (0)
#define MF_CP_F(INTEG_)                                          \
  do {                                                           \
    if (forces) { if (zmu) MF_CP(INTEG_, true, true); else MF_CP(INTEG_, true, false); }    \
    else        { if (zmu) MF_CP(INTEG_, false, true); else MF_CP(INTEG_, false, false); }  \
  } while (0)
  if (integ == MF_INTEG_DYNAMICS) MF_CP_F(MF_INTEG_DYNAMICS); else MF_CP_F(MF_INTEG_ODEINT_EULER);
#undef MF_CP_F
#undef MF_CP
  hipError_t e = hipGetLastError();
  MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("rollout_fwd (component-parallel) launch: ") + hipGetErrorString(e));
  return MF_OK;
}

}  // namespace mf

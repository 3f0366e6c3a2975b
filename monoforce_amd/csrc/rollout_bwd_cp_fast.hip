// Backward rollout, component-parallel lane mapping (rollout_bwd_cp_kernel.h): float32 fast-math instantiation and the
// host-side choice between it and the one-point-per-lane kernels.
#include "rollout_bwd_cp_kernel.h"
#include "rollout_fwd_cp_kernel.h"

namespace mf {

// Measured, backward, N = 4 (tools/ab_cp.py; ms component-parallel vs one point per lane): B = 256 0.38 / 0.88, 1024 0.39 / 0.89,
// 2048 0.45 / 0.90, 4096 0.63 / 0.95, 8192 0.81 / 0.99, 16384 1.63 / 1.62, 32768 3.19 / 2.89 -- up to two waves per SIMD (a lane
// owns one footprint cell here: its gradient accumulator flushes with one atomic per map, which is what bounds the one-point-
// per-lane kernel once the chip is full); MF_CP_BWD_MAX_WAVES overrides (0 disables)
static long long cp_bwd_max_waves() {
  static const long long v = getenv("MF_CP_BWD_MAX_WAVES") ? atoll(getenv("MF_CP_BWD_MAX_WAVES")) : -1;
  return v >= 0 ? v : 2 * device_simds();      // two waves per SIMD (MI355X: 2048)
}

static bool cp_bwd_covers(const MfRolloutDesc* d, bool joints, int scalar_bytes = 4) {
  if (d->math_mode != MF_MATH_FAST || d->N > 4 || joints) return false;
  if (d->points_per_lane != 0 && d->points_per_lane != MF_LANES_COMPONENT) return false;
  const long long waves = ((long long)d->B + 3) / 4;
  if (d->points_per_lane == 0 && waves > cp_bwd_max_waves()) return false;
  // 32-bit byte offsets into the saved rows and the upstream gradients
  const long long row = (long long)d->N * 3 > 9 ? (long long)d->N * 3 : 9;
  if ((long long)d->T * d->B * row * scalar_bytes >= (1ll << 32)) return false;
  return true;
}
bool use_component_parallel_bwd(const MfRolloutDesc* d, const MfRolloutBwdBufs* p, int scalar_bytes) { return cp_bwd_covers(d, p->joint_angles != nullptr, scalar_bytes); }

// The compact per-step record (rollout_fwd_cp_kernel.h REC, layout in rollout_cp_common.h): kept where BOTH directions run
// component-parallel and the launch has at most one wave per SIMD -- 256 B per rollout and step (131 MB at B = 1024, T = 500; round
// 2's record was 1 KiB and stopped paying at B = 2048, where its stores bound the forward): one more store per wave-step forward;
// backward no contact chain to recompute (~65 instructions, 7 transcendentals), and the forward's own values at every clamp and
// kink.  Its backward streams the record through LDS with two more waves per workgroup (default integrator: <= 512 workgroups;
// dynamics(), whose ring slots carry the Rodrigues coefficients as well: <= 256) or, default integrator only, reads it in the
// computing wave itself up to 1024 waves.  dynamics() read by the one wave LOSES against recomputing (B = 1024: 0.499 vs 0.436 ms
// backward), so beyond its streaming range it keeps no record (MF_CP_RECORD_DYNAMICS=1 forces one: A/B runs, parity tests of that
// kernel).  MF_CP_RECORD_MAX_WAVES overrides the size limit (0 disables).
long long cp_record_bytes(const MfRolloutDesc* d, int scalar_bytes) {
  static const long long max_waves_env = getenv("MF_CP_RECORD_MAX_WAVES") ? atoll(getenv("MF_CP_RECORD_MAX_WAVES")) : -1;
  const long long max_waves = max_waves_env >= 0 ? max_waves_env : device_simds();      // one wave per SIMD
  if (!d || d->B <= 0 || d->T <= 0) return 0;
  MfRolloutFwdBufs f{};
  static const bool dyn = getenv("MF_CP_RECORD_DYNAMICS") && atoi(getenv("MF_CP_RECORD_DYNAMICS")) != 0;
  static const bool one_wave = getenv("MF_CP_BWD_MODE") && atoi(getenv("MF_CP_BWD_MODE")) == kCpSaved;
  if (d->has_joints) return 0;
  if (!use_component_parallel(d, &f, scalar_bytes) || !cp_bwd_covers(d, false, scalar_bytes)) return 0;
  const long long waves = ((long long)d->B + 3) / 4;
  if (waves > max_waves) return 0;
  if (d->integrator != MF_INTEG_ODEINT_EULER && !dyn && (one_wave || waves > (long long)cp_stream_max_grid(d->integrator))) return 0;
  const long long bytes = (long long)d->T * d->B * 16 * 4 * scalar_bytes;      // cp::kRecBytesPerLane<S> per lane and step
  if (bytes >= (1ll << 32)) return 0;
  return bytes;
}

// The fused physics loss rides on the STREAMING backward (its fetching waves form dL/dXs -- and, with MF_LOSS_VALUE_IN_BACKWARD, the
// value): a launch that keeps a record and streams it, either integrator (dynamics(): <= 256 workgroups).  The forward half -- the
// LOSS kernels that accumulate the value while they write the rows -- exists for the default integrator only (cp_loss_in_forward);
// dynamics() takes the value from the backward launch or from mf_physics_loss_value_* on the written rows.
bool cp_loss_fusable(const MfRolloutDesc* d) {
  static const bool one_wave = getenv("MF_CP_BWD_MODE") && atoi(getenv("MF_CP_BWD_MODE")) == kCpSaved;
  if (!d || d->layout != MF_LAYOUT_TIME_MAJOR || one_wave) return false;
  if (d->integrator != MF_INTEG_ODEINT_EULER && d->integrator != MF_INTEG_DYNAMICS) return false;
  if (cp_record_bytes(d, 4) <= 0) return false;
  const long long grid = ((long long)d->B * 16 + 63) / 64;
  return grid <= (long long)cp_stream_max_grid(d->integrator);
}
// The record-reading kernel (kCpSaved: beyond the streaming form's grid, up to one wave per SIMD) re-gathers every cell's (z, mu); with
// a shared float32 pair it reads them interleaved.  MF_CP_BWD_ZMU=0: two 4-byte loads per cell (A/B runs, parity of the two).
bool cp_bwd_wants_zmu(const MfRolloutDesc* d, bool has_rec, bool has_mu) {
  static const bool off = getenv("MF_CP_BWD_ZMU") && atoi(getenv("MF_CP_BWD_ZMU")) == 0;
  static const int forced = getenv("MF_CP_BWD_MODE") ? atoi(getenv("MF_CP_BWD_MODE")) : -1;
  if (off || !d || !d->map_shared || !has_rec || !has_mu || d->integrator != MF_INTEG_ODEINT_EULER || d->math_mode != MF_MATH_FAST) return false;
  if ((long long)d->H * d->W * 8 >= (1ll << 31)) return false;      // 32-bit byte offsets into the (z, mu) cells
  const long long grid = ((long long)d->B * 16 + 63) / 64;
  return forced == kCpSaved || grid > (long long)cp_stream_max_grid(d->integrator);
}
bool cp_loss_in_forward(const MfRolloutDesc* d) { return cp_loss_fusable(d) && d->integrator == MF_INTEG_ODEINT_EULER; }

}  // namespace mf
// (round 6: no kernel needs the buffer any more -- the component-parallel kernels compile the control gradient out, the multi-wave ones test
//  for NULL, the one-point-per-lane ones send the rows to a 3-float dump per rollout; kept for callers that ask)
extern "C" int mf_rollout_bwd_wants_gcontrols(const MfRolloutDesc* d) { (void)d; return 0; }
namespace mf { long long mw_record_bytes(const MfRolloutDesc* d, int scalar_bytes); }   // rollout_bwd_mw_fast.hip
extern "C" long long mf_rollout_record_bytes(const MfRolloutDesc* d) {
  const long long cp = mf::cp_record_bytes(d, 4);
  return cp > 0 ? cp : mf::mw_record_bytes(d, 4);
}
// the float64 validation build of the component-parallel kernels (points_per_lane = MF_LANES_COMPONENT): 32-byte quads
extern "C" long long mf_rollout_record_bytes_f64(const MfRolloutDesc* d) {
  if (!d || d->points_per_lane != MF_LANES_COMPONENT) return 0;
  const long long cp = mf::cp_record_bytes(d, 8);
  return cp > 0 ? cp : mf::mw_record_bytes(d, 8);      // (bodies of 5..512 points: rollout_bwd_mw_kernel.h's record, 32 bytes per rollout-step)
}
namespace mf {

int launch_rollout_bwd_cp_f32(const RolloutBwdArgs<float>& a, int integ, bool xs_only, hipStream_t st) {
  if (integ == MF_INTEG_DYNAMICS) return launch_rollout_bwd_cp_dynamics_f32(a, xs_only, st);
  return launch_rollout_bwd_cp_variant<float, MF_INTEG_ODEINT_EULER>(a, xs_only, st);
}

}  // namespace mf

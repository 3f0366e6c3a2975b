// Backward rollout of one rollout over several waves (rollout_bwd_mw_kernel.h): float32 fast-math instantiations and the rules
// that send a launch to them.
#include <cstdlib>
#include "rollout_bwd_mw_kernel.h"

namespace mf {

// The launches these kernels serve: float32 MF_MATH_FAST, either integrator, rigid body, one point per lane --
//   * bodies of 65..512 points spread over 2 / 4 / 8 waves by choose_lane_map (<= 2048 waves per launch), and
//   * bodies of 5..64 points (8 / 16 / 32 / 64 lanes per rollout) up to two waves per SIMD (the positions-only instantiations hold
//     237 registers; beyond, the forward goes out in chunks and the general kernels take over).  N <= 4 has the component-parallel kernels.
// MF_MW_BWD=0 keeps the general kernel (A/B runs, parity tests of the two against each other).
static bool mw_shape(const MfRolloutDesc* d) {
  static const bool off = getenv("MF_MW_BWD") && atoi(getenv("MF_MW_BWD")) == 0;
  if (off || !d || d->B <= 0 || d->T <= 0 || d->N <= 4 || d->N > 512) return false;
  if (d->math_mode != MF_MATH_FAST || d->has_joints) return false;
  if (d->integrator != MF_INTEG_ODEINT_EULER && d->integrator != MF_INTEG_DYNAMICS) return false;
  if (d->points_per_lane == 4) return false;
  const LaneMap m = choose_lane_map(d->B, d->N, d->points_per_lane == MF_LANES_COMPONENT ? 0 : d->points_per_lane);
  if (m.PPL != 1 || m.G < 8) return false;
  return m.G > 64 || (long long)d->B * m.G <= 2048ll * 64;
}
long long mw_record_bytes(const MfRolloutDesc* d) {
  if (!mw_shape(d)) return 0;
  return (long long)d->T * d->B * kMwRecFloats * (long long)sizeof(float);
}
bool use_multiwave_bwd(const MfRolloutDesc* d, const MfRolloutBwdBufs* p) {
  return mw_shape(d) && p->rec != nullptr && !p->joint_angles && !p->loss;
}

int launch_rollout_bwd_mw_f32(const RolloutBwdArgs<float>& a, int G, int integ, bool xs_only, hipStream_t st) {
  // LDS gradient tiles (rollout_bwd_mw_kernel.h) while every workgroup of the launch is resident with its tiles: 160 KB per CU,
  // 256 CUs.  MF_MW_TILE=0 keeps the register accumulators (A/B runs, parity of the two routes).
  static const bool tile_off = getenv("MF_MW_TILE") && atoi(getenv("MF_MW_TILE")) == 0;
  bool launched = false;
#define MF_LAUNCH(G_, XS_, T_, I_) hipLaunchKernelGGL((rollout_bwd_mw_kernel<G_, XS_, T_, I_>), dim3(grid), dim3(blk), 0, st, a)
#define MF_CASE(G_)                                                                                              \
  if (!launched && G == G_) {                                                                                    \
    launched = true;                                                                                             \
    constexpr int blk = G_ > 64 ? G_ : 64;                                                                       \
    constexpr int TE = mw_tile_edge(G_);                                                                         \
    constexpr long long lds = (long long)(G_ > 64 ? 1 : 64 / G_) * 2 * (TE + 1) * TE * 4 + 4096;                       \
    const unsigned grid = (unsigned)(((long long)a.B * G_ + blk - 1) / blk);                                     \
    const bool tile = TE > 0 && !tile_off && (long long)((grid + 255) / 256) * lds <= 160 * 1024 && (long long)a.H * a.W < (1ll << 30); \
    const bool dyn = integ == MF_INTEG_DYNAMICS;                                                                 \
    if (tile) {                                                                                                  \
      if constexpr (TE > 0) {                                                                                    \
        if (dyn) { if (xs_only) MF_LAUNCH(G_, true, TE, MF_INTEG_DYNAMICS); else MF_LAUNCH(G_, false, TE, MF_INTEG_DYNAMICS); }          \
        else     { if (xs_only) MF_LAUNCH(G_, true, TE, MF_INTEG_ODEINT_EULER); else MF_LAUNCH(G_, false, TE, MF_INTEG_ODEINT_EULER); }  \
      }                                                                                                          \
    } else {                                                                                                     \
      if (dyn) { if (xs_only) MF_LAUNCH(G_, true, 0, MF_INTEG_DYNAMICS); else MF_LAUNCH(G_, false, 0, MF_INTEG_DYNAMICS); }              \
      else     { if (xs_only) MF_LAUNCH(G_, true, 0, MF_INTEG_ODEINT_EULER); else MF_LAUNCH(G_, false, 0, MF_INTEG_ODEINT_EULER); }      \
    }                                                                                                            \
  }
  MF_CASE(8) MF_CASE(16) MF_CASE(32) MF_CASE(64) MF_CASE(128) MF_CASE(256) MF_CASE(512)
#undef MF_CASE
#undef MF_LAUNCH
  MF_REQUIRE(launched, MF_ERR_UNSUPPORTED, "rollout_bwd: no multi-wave kernel for this lane mapping");
  hipError_t e = hipGetLastError();
  MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("rollout_bwd (multi-wave) launch: ") + hipGetErrorString(e));
  return MF_OK;
}

}  // namespace mf

// Forward rollout: host side of mf_rollout_fwd_* and the reference-order (exact) kernel instantiations.
// This TU is compiled with -ffp-contract=off; the FMA-contracted float32 kernels live in rollout_fwd_fast.hip.
#include "rollout_fwd_cp_kernel.h"

namespace mf {
long long mw_record_bytes(const MfRolloutDesc* d, int scalar_bytes);   // rollout_bwd_mw_fast.hip
int launch_rollout_fwd_mw_rec_f64(const RolloutArgs<double>& a, LaneMap m, int integ, bool forces, hipStream_t st);   // rollout_mw_f64.hip
}

namespace mf {

template <typename S>
static int fill_args(const MfRolloutDesc* d, const MfRolloutFwdBufs* p, RolloutArgs<S>* a, LaneMap* m, int* block) {
  MF_REQUIRE(d && p, MF_ERR_INVALID, "rollout_fwd: null descriptor");
  MF_REQUIRE(d->B > 0 && d->N > 0 && d->H > 1 && d->W > 0, MF_ERR_INVALID, "rollout_fwd: B, N, H, W must be positive");
  MF_REQUIRE(d->T >= 1, MF_ERR_INVALID, "rollout_fwd: T must be >= 1");
  MF_REQUIRE(d->n_tracks == 2 || d->n_tracks == 4, MF_ERR_INVALID, "n_tracks must be 2 or 4");
  MF_REQUIRE(d->integrator == MF_INTEG_DYNAMICS || d->integrator == MF_INTEG_ODEINT_EULER, MF_ERR_INVALID,
             "rollout_fwd: unknown integrator");
  MF_REQUIRE(d->layout == MF_LAYOUT_BATCH_MAJOR || d->layout == MF_LAYOUT_TIME_MAJOR, MF_ERR_INVALID,
             "rollout_fwd: unknown layout");
  MF_REQUIRE(d->math_mode == MF_MATH_EXACT || d->math_mode == MF_MATH_FAST, MF_ERR_INVALID, "rollout_fwd: unknown math_mode");
  MF_REQUIRE(p->z && p->controls && p->ts && p->points && p->part && p->x0 && p->xd0 && p->R0 && p->w0, MF_ERR_INVALID,
             "rollout_fwd: null input buffer");
  MF_REQUIRE(p->Xs && p->Rs && (p->cost_rows || (p->Xds && p->Omegas)), MF_ERR_INVALID, "rollout_fwd: null output buffer");
  MF_REQUIRE((p->Fs != nullptr) == (p->Ff != nullptr), MF_ERR_INVALID, "rollout_fwd: pass both force buffers or neither");
  MF_REQUIRE((long long)d->H * d->W < (1ll << 30), MF_ERR_UNSUPPORTED, "rollout_fwd: grid too large");
  MF_REQUIRE(d->H < (1 << 23), MF_ERR_UNSUPPORTED, "rollout_fwd: grid too large (H must be below 2^23)");
  MF_REQUIRE(d->H >= 2, MF_ERR_INVALID, "rollout_fwd: the grid needs at least 2 x 2 cells");
  MF_REQUIRE(d->N <= 512, MF_ERR_UNSUPPORTED, "rollout_fwd: more than 512 contact points");
  MF_REQUIRE(d->map_shared || (long long)d->B * d->H * d->W * (long long)sizeof(S) < (1ll << 32), MF_ERR_UNSUPPORTED,
             "rollout_fwd: per-rollout maps of 4 GiB or more in total (use a shared map or split the batch)");
  *block = d->block ? d->block : 64;
  MF_REQUIRE(*block == 64 || *block == 128 || *block == 256, MF_ERR_INVALID, "rollout_fwd: block must be 64, 128 or 256");
  *m = choose_lane_map(d->B, d->N, d->points_per_lane == MF_LANES_COMPONENT ? 0 : d->points_per_lane);
  const int fstride = d->force_stride ? d->force_stride : d->N;
  MF_REQUIRE(!p->Fs || fstride >= m->G * m->PPL, MF_ERR_INVALID,
             "rollout_fwd: force_stride too small -- allocate Fs/Ff with mf_rollout_force_stride(desc) point slots per row");

  a->B = d->B; a->T = d->T; a->N = d->N; a->H = d->H; a->W = d->W;
  a->n_tracks = d->n_tracks; a->layout = d->layout; a->map_shared = d->map_shared; a->skip_snap = d->skip_snap; a->default_state = d->default_state; a->b0 = 0;
  const bool strided = d->controls_stride_b != 0 || d->controls_stride_t != 0;
  a->ctrl_sb = strided ? d->controls_stride_b : d->T * 2; a->ctrl_st = strided ? d->controls_stride_t : 2;
  MF_REQUIRE(a->ctrl_sb >= 0 && a->ctrl_st >= 0, MF_ERR_INVALID, "rollout_fwd: negative controls stride");
  a->fstride = fstride;
  a->mass = (S)d->mass; a->inv_mass = (S)(1.0 / d->mass); a->mg = (S)(d->mass * d->gravity); a->k = (S)d->stiffness;
  a->damp = (S)d->damping; a->omega_max = (S)d->omega_max; a->res = (S)d->grid_res; a->inv_res = (S)(1.0 / (double)(S)d->grid_res);   // RN(1 / res) of the ROUNDED res (Mth::cell_coord)
  a->d_max = (S)d->d_max; a->dt = (S)d->dt;
  a->half_ly = (S)(d->robot_size_y / 2.0);
  a->sink = (S)(d->mass * d->gravity / (d->stiffness + 1e-6));
  for (int i = 0; i < 9; ++i) a->Iinv[i] = (S)d->Iinv[i];
  a->z = (const S*)p->z; a->mu = (const S*)p->mu; a->controls = (const S*)p->controls; a->ts = (const S*)p->ts;
  a->points = (const S*)p->points; a->part = p->part;
  a->x0 = (S*)p->x0; a->xd0 = (const S*)p->xd0; a->R0 = (const S*)p->R0; a->w0 = (const S*)p->w0;
  a->Xs = (S*)p->Xs; a->Xds = (S*)p->Xds; a->Rs = (S*)p->Rs; a->Om = (S*)p->Omegas; a->Fs = (S*)p->Fs; a->Ff = (S*)p->Ff;
  a->Xraw = (S*)p->Xraw;
  a->joint_angles = (const S*)p->joint_angles;
  a->cost_rows = (S*)p->cost_rows; a->pose_stride = d->pose_stride > 0 ? d->pose_stride : 1;
  a->path_cost = (S*)p->path_cost;
  a->zmu = nullptr;
  a->rec = nullptr;
  a->loss_T2 = 0; a->loss_gt = nullptr; a->loss_row_w = nullptr; a->loss_partial = nullptr; a->loss_ticket = nullptr;
  a->loss_out = nullptr; a->loss_inv_count = (S)0; a->loss_poison = nullptr;
  for (int i = 0; i < 12; ++i) a->joint_xyz[i] = (S)d->joint_xyz[i];
  if (p->joint_angles) {
    MF_REQUIRE(d->n_tracks == 4, MF_ERR_UNSUPPORTED, "rollout_fwd: joint angles need 4 driving parts (fl, fr, rl, rr)");
    *m = choose_lane_map(d->B, d->N, 0);     // the articulated kernels exist for the multi-wave mappings (small batches of
    if (m->G <= 64) *m = choose_lane_map(d->B, d->N, 4);   // a large body) and the 4-points-per-lane ones
    MF_REQUIRE(fstride >= m->G * m->PPL, MF_ERR_INVALID, "rollout_fwd: force_stride too small for the articulated kernels");
  }
  return MF_OK;
}

}  // namespace mf

extern "C" int mf_rollout_force_stride(const MfRolloutDesc* d) {
  if (!d || d->B <= 0 || d->N <= 0 || d->N > 512) return -1;
  mf::LaneMap m = mf::choose_lane_map(d->B, d->N, (d->has_joints || d->points_per_lane == MF_LANES_COMPONENT) ? 0 : d->points_per_lane);
  if (d->has_joints && m.G <= 64) m = mf::choose_lane_map(d->B, d->N, 4);
  const int lanes = m.G * m.PPL;
  return lanes > d->N ? lanes : d->N;
}

namespace mf {
// (z, mu) of the shared maps interleaved for the ZMU kernels; without a friction map the second component is never used
template <typename S>
__global__ void __launch_bounds__(256) interleave_maps_kernel(const S* __restrict__ z, const S* __restrict__ mu, int n,
                                                             cp::Pk2<S>* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = cp::Pk2<S>{z[i], mu ? mu[i] : (S)1};
}
// true (and a.zmu set, the interleave pass launched) when this launch can read the interleaved copy
// the launch shapes whose kernels read the shared maps interleaved ...
static bool zmu_shape(const MfRolloutDesc* d, bool joints, const LaneMap& m, int scalar_bytes) {
  if (!d->map_shared || d->math_mode != MF_MATH_FAST || joints || m.PPL != 1 || m.G > 64) return false;
  return (long long)d->H * d->W * (long long)scalar_bytes < (1ll << 31);   // 32-bit byte offsets into the (z, mu) cells
}
// ... and those that run the interleave pass into a scratch they are offered:
// below ~half a wave per SIMD the launch is bound by the instruction stream of its waves; the extra pass (a second launch in
// front of the rollout, ~10 us) then costs what the two saved gathers bring (measured: B = 1024 path costs 0.306 -> 0.323 ms)
// (the float64 validation build takes the pass whenever it is offered: its purpose is to run the ZMU kernels)
static bool zmu_pass(const MfRolloutDesc* d, const LaneMap& m, int scalar_bytes) {
  return scalar_bytes != 4 || (long long)d->B * m.G >= device_simds() / 2 * 64;      // (half a wave per SIMD)
}
template <typename S>
static bool use_interleaved_maps(const MfRolloutDesc* d, const MfRolloutFwdBufs* p, RolloutArgs<S>* a, const LaneMap& m, hipStream_t st) {
  if ((!p->zmu_scratch && !p->zmu) || !zmu_shape(d, p->joint_angles != nullptr, m, (int)sizeof(S))) return false;
  if (p->zmu && p->mu) {   // the caller staged the interleaved pair itself (mf_terrain_stage_fwd_f32): no pass, no batch-size condition
    a->zmu = (const S*)p->zmu;
    return true;
  }
  if (!p->zmu_scratch) return false;
  if (!zmu_pass(d, m, (int)sizeof(S))) return false;
  const int n = d->H * d->W;
  hipLaunchKernelGGL((interleave_maps_kernel<S>), dim3((n + 255) / 256), dim3(256), 0, st, a->z, a->mu, n, (cp::Pk2<S>*)p->zmu_scratch);
  a->zmu = (const S*)p->zmu_scratch;
  return true;
}

inline int launch_rollout_fwd_cp_any(const RolloutArgs<float>& a, int integ, bool forces, bool zmu, hipStream_t st) { return launch_rollout_fwd_cp_f32(a, integ, forces, zmu, st); }
inline int launch_rollout_fwd_cp_any(const RolloutArgs<double>& a, int integ, bool forces, bool zmu, hipStream_t st) { return launch_rollout_fwd_cp_f64(a, integ, forces, zmu, st); }

// The component-parallel launch (a rollout over a 16-lane row, rollout_fwd_cp_kernel.h) with everything that rides on it: the
// interleaved maps, the per-step record for the backward, the fused physics loss.  S = float: the kernels the dispatcher picks for
// few rollouts of a small body; S = double: their validation build, on explicit request (points_per_lane = MF_LANES_COMPONENT).
template <typename S>
static int rollout_fwd_cp(const MfRolloutDesc* d, const MfRolloutFwdBufs* p, RolloutArgs<S>& a, hipStream_t st) {
  const bool forces = p->Fs != nullptr;
  const bool zmu = use_interleaved_maps<S>(d, p, &a, LaneMap{16, 1}, st);
  if (p->rec && cp_record_bytes(d, (int)sizeof(S)) > 0) {      // the per-step record for the backward (MfRolloutFwdBufs.rec)
    MF_REQUIRE(((uintptr_t)p->rec & 15) == 0, MF_ERR_INVALID, "rollout_fwd: rec must be 16-byte aligned");
    a.rec = (S*)p->rec;
  }
  if (p->loss && (p->loss->flags & MF_LOSS_VALUE_IN_BACKWARD)) {      // the backward will form the value: mark it as not yet known
    MF_REQUIRE(cp_loss_fusable(d), MF_ERR_UNSUPPORTED, "rollout_fwd: this launch cannot carry the fused physics loss (mf_rollout_loss_fusable)");
    MF_REQUIRE(p->loss->loss, MF_ERR_INVALID, "rollout_fwd: MF_LOSS_VALUE_IN_BACKWARD needs MfRolloutLoss.loss");
    a.loss_poison = (S*)p->loss->loss;
  } else if (p->loss) {      // physics_loss inside the launch (MfRolloutLoss)
    const MfRolloutLoss* L = p->loss;
    MF_REQUIRE(cp_loss_in_forward(d), MF_ERR_UNSUPPORTED, "rollout_fwd: this launch cannot accumulate the physics loss itself (the LOSS kernels ride on the "
               "default integrator; dynamics(): MF_LOSS_VALUE_IN_BACKWARD, or mf_physics_loss_value_* on the rows)");
    MF_REQUIRE(!forces && d->layout == MF_LAYOUT_TIME_MAJOR, MF_ERR_INVALID, "rollout_fwd: the fused physics loss needs Fs = Ff = NULL and MF_LAYOUT_TIME_MAJOR");
    MF_REQUIRE(L->T2 > 0 && L->gt && L->row_w && L->partial && L->ticket && L->loss, MF_ERR_INVALID, "rollout_fwd: incomplete MfRolloutLoss");
    MF_REQUIRE((long long)d->B * L->T2 * 3 * (long long)sizeof(S) < (1ll << 32), MF_ERR_UNSUPPORTED, "rollout_fwd: ground truth of 4 GiB or more");
    a.loss_T2 = L->T2; a.loss_gt = (const S*)L->gt; a.loss_row_w = (const S*)L->row_w;
    a.loss_partial = (S*)L->partial; a.loss_ticket = L->ticket; a.loss_out = (S*)L->loss;
    a.loss_inv_count = (S)(1.0 / ((double)d->B * L->T2 * 3));
  }
  return launch_rollout_fwd_cp_any(a, d->integrator, forces, zmu, st);
}
}  // namespace mf

// 1 where mf_rollout_fwd_f32 given `zmu_scratch` (and no `zmu`) fills it with the interleaved (z, mu) pair on the component-parallel route:
// the caller may then hand the same buffer to mf_rollout_bwd_f32 as `zmu` (same step, same maps) and spare the backward its own pass
extern "C" int mf_rollout_fwd_stages_zmu(const MfRolloutDesc* d) {
  if (!d || d->B <= 0 || d->T <= 0 || d->has_joints) return 0;
  MfRolloutFwdBufs none{};
  const mf::LaneMap m{16, 1};
  return mf::use_component_parallel(d, &none) && mf::zmu_shape(d, false, m, 4) && mf::zmu_pass(d, m, 4) ? 1 : 0;
}

extern "C" int mf_rollout_fwd_f32(const MfRolloutDesc* d, const MfRolloutFwdBufs* p, void* s) {
  mf::RolloutArgs<float> a;
  mf::LaneMap m;
  int block;
  int rc = mf::fill_args<float>(d, p, &a, &m, &block);
  if (rc != MF_OK) return rc;
  MF_REQUIRE((((uintptr_t)p->zmu_scratch | (uintptr_t)p->zmu) & 7) == 0, MF_ERR_INVALID, "rollout_fwd: zmu_scratch / zmu must be 8-byte aligned");
  if (p->joint_angles) {
    MF_REQUIRE(p->Fs && p->Ff && !p->cost_rows, MF_ERR_UNSUPPORTED, "rollout_fwd: articulated rollouts write all six outputs");
    if (d->math_mode == MF_MATH_FAST) return mf::launch_rollout_fwd_joints_fast_f32(a, m, d->integrator, block, (hipStream_t)s);
    return mf::launch_rollout_fwd<float, false, true>(a, m, d->integrator, block, (hipStream_t)s);
  }
  if (p->cost_rows) {   // path-cost mode: cost rows + decimated poses (see MfRolloutFwdBufs.cost_rows)
    MF_REQUIRE(d->math_mode == MF_MATH_FAST && d->layout == MF_LAYOUT_TIME_MAJOR && d->pose_stride >= 1, MF_ERR_UNSUPPORTED,
               "rollout_fwd: cost rows need float32 MF_MATH_FAST, MF_LAYOUT_TIME_MAJOR and pose_stride >= 1");
    MF_REQUIRE(!p->Xds && !p->Omegas && !p->Fs && !p->Ff && !p->Xraw, MF_ERR_INVALID,
               "rollout_fwd: with cost_rows only Xs and Rs (decimated) are written -- pass NULL for Xds, Omegas, Fs, Ff, Xraw");
    if (m.PPL == 4 && m.G < 64) m = mf::choose_lane_map(d->B, d->N, 1);
    if (mf::use_interleaved_maps<float>(d, p, &a, m, (hipStream_t)s))
      return mf::launch_rollout_fwd_zmu_f32(a, m, d->integrator, block, false, false, d->cost_project != 0 ? 2 : 1, (hipStream_t)s);
    return mf::launch_rollout_fwd_cost_f32(a, m, d->integrator, block, d->cost_project != 0, (hipStream_t)s);
  }
  const bool forces = p->Fs != nullptr;
  if (!forces && (d->math_mode != MF_MATH_FAST || p->joint_angles)) {
    mf::set_error("rollout_fwd: the states-only kernels (Fs = Ff = NULL) exist for float32 MF_MATH_FAST rigid-body rollouts only");
    return MF_ERR_UNSUPPORTED;
  }
  if (d->math_mode == MF_MATH_FAST) {
    if (mf::use_component_parallel(d, p))   // few rollouts of a small body: a rollout over 16 lanes (rollout_fwd_cp_kernel.h)
      return mf::rollout_fwd_cp<float>(d, p, a, (hipStream_t)s);
    MF_REQUIRE(!p->loss, MF_ERR_UNSUPPORTED, "rollout_fwd: this launch cannot carry the fused physics loss (mf_rollout_loss_fusable)");
    if (!forces && m.PPL == 4 && m.G < 64) m = mf::choose_lane_map(d->B, d->N, 1);
    // >= one wave per SIMD and a one-point-per-lane mapping within a wave: the split-store kernels (rollout_fwd_kernel.h)
    const bool split = m.PPL == 1 && m.G <= 64 && (long long)d->B * m.G >= mf::device_simds() * 64;
    if (p->rec && mf::mw_record_bytes(d, 4) > 0) {      // the 16-byte record of rollout_bwd_mw_kernel.h
      MF_REQUIRE(((uintptr_t)p->rec & 15) == 0, MF_ERR_INVALID, "rollout_fwd: rec must be 16-byte aligned");
      a.rec = (float*)p->rec;
    }
    if (mf::use_interleaved_maps<float>(d, p, &a, m, (hipStream_t)s))
      return mf::launch_rollout_fwd_zmu_f32(a, m, d->integrator, block, forces, split, 0, (hipStream_t)s);
    if (split)
      return mf::launch_rollout_fwd_split_fast_f32(a, m, d->integrator, block, forces, (hipStream_t)s);
    return mf::launch_rollout_fwd_fast_f32(a, m, d->integrator, block, forces, (hipStream_t)s);
  }
  MF_REQUIRE(!p->loss, MF_ERR_UNSUPPORTED, "rollout_fwd: the fused physics loss exists for the float32 fast-math kernels only");
  return mf::launch_rollout_fwd<float, false>(a, m, d->integrator, block, (hipStream_t)s);
}

extern "C" int mf_rollout_fwd_f64(const MfRolloutDesc* d, const MfRolloutFwdBufs* p, void* s) {
  mf::RolloutArgs<double> a;
  mf::LaneMap m;
  int block;
  int rc = mf::fill_args<double>(d, p, &a, &m, &block);
  if (rc != MF_OK) return rc;
  // the float64 VALIDATION build of the component-parallel kernels (rollout_fwd_cp_f64.hip): on explicit request only
  if (d->points_per_lane == MF_LANES_COMPONENT && mf::use_component_parallel(d, p, 8)) {
    MF_REQUIRE((((uintptr_t)p->zmu_scratch | (uintptr_t)p->zmu) & 15) == 0, MF_ERR_INVALID, "rollout_fwd: zmu_scratch / zmu must be 16-byte aligned");
    return mf::rollout_fwd_cp<double>(d, p, a, (hipStream_t)s);
  }
  // ... and of the recording one-point-per-lane kernels of 5..512-point bodies (rollout_fwd_kernel.h FAST / REC, whose backward is
  // rollout_bwd_mw_kernel.h): same request, with the record buffer
  if (d->points_per_lane == MF_LANES_COMPONENT && p->rec && !p->joint_angles && !p->cost_rows && !p->loss && mf::mw_record_bytes(d, 8) > 0) {
    MF_REQUIRE(((uintptr_t)p->rec & 31) == 0, MF_ERR_INVALID, "rollout_fwd: rec must be 32-byte aligned");
    a.rec = (double*)p->rec;
    return mf::launch_rollout_fwd_mw_rec_f64(a, m, d->integrator, p->Fs != nullptr, (hipStream_t)s);
  }
  if (!p->Fs) { mf::set_error("rollout_fwd: float64 needs the force buffers (states only: the component-parallel validation build, points_per_lane = MF_LANES_COMPONENT)"); return MF_ERR_UNSUPPORTED; }
  if (p->cost_rows) { mf::set_error("rollout_fwd: cost rows exist for float32 only"); return MF_ERR_UNSUPPORTED; }
  if (p->loss) { mf::set_error("rollout_fwd: the fused physics loss exists for the component-parallel kernels only"); return MF_ERR_UNSUPPORTED; }
  if (p->joint_angles) return mf::launch_rollout_fwd<double, false, true>(a, m, d->integrator, block, (hipStream_t)s);
  return mf::launch_rollout_fwd<double, false>(a, m, d->integrator, block, (hipStream_t)s);   // float64 is always exact
}

namespace mf {
template <typename S>
__global__ void default_state_kernel(int B, int T, const S* __restrict__ controls, S* __restrict__ x0, S* __restrict__ xd0,
                                     S* __restrict__ R0, S* __restrict__ w0) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const S v = controls[(size_t)b * T * 2 + 0], w = controls[(size_t)b * T * 2 + 1];
  const S zero = (S)0, one = (S)1;
  x0[b * 3 + 0] = zero; x0[b * 3 + 1] = zero; x0[b * 3 + 2] = zero;
  xd0[b * 3 + 0] = v; xd0[b * 3 + 1] = zero; xd0[b * 3 + 2] = zero;
  w0[b * 3 + 0] = zero; w0[b * 3 + 1] = zero; w0[b * 3 + 2] = w;
#pragma unroll
  for (int c = 0; c < 9; ++c) R0[b * 9 + c] = (c % 4 == 0) ? one : zero;
}
template <typename S>
static int default_state(int B, int T, const S* controls, S* x0, S* xd0, S* R0, S* w0, void* s) {
  MF_REQUIRE(B > 0 && T > 0 && controls && x0 && xd0 && R0 && w0, MF_ERR_INVALID, "rollout_default_state: bad argument");
  hipLaunchKernelGGL((default_state_kernel<S>), dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)s, B, T, controls, x0, xd0, R0, w0);
  hipError_t e = hipGetLastError();
  MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("rollout_default_state launch: ") + hipGetErrorString(e));
  return MF_OK;
}
}  // namespace mf

extern "C" int mf_rollout_default_state_f32(int32_t B, int32_t T, const float* controls, float* x0, float* xd0, float* R0, float* w0, void* s) {
  return mf::default_state<float>(B, T, controls, x0, xd0, R0, w0, s);
}
extern "C" int mf_rollout_default_state_f64(int32_t B, int32_t T, const double* controls, double* x0, double* xd0, double* R0, double* w0, void* s) {
  return mf::default_state<double>(B, T, controls, x0, xd0, R0, w0, s);
}

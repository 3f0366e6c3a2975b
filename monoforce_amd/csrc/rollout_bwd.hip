// Backward rollout: host side of mf_rollout_bwd_* and the reference-order (exact) kernel instantiations.
// Compiled with -ffp-contract=off; the FMA-contracted float32 kernels live in rollout_bwd_fast.hip.
#include "rollout_bwd_cp_kernel.h"
#include "rollout_bwd_mw_kernel.h"

namespace mf {
// (z, mu) of the shared maps interleaved (as rollout_fwd.hip's pass for the forward's ZMU kernels)
template <typename S>
__global__ void __launch_bounds__(256) interleave_maps_bwd_kernel(const S* __restrict__ z, const S* __restrict__ mu, int n, cp::Pk2<S>* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = cp::Pk2<S>{z[i], mu[i]};
}

long long mw_record_bytes(const MfRolloutDesc* d, int scalar_bytes);   // rollout_bwd_mw_fast.hip

// The launches of a positions-only upstream that go to the XS_ONLY one-point-per-lane kernels (rollout_bwd_kernel.h): float32 fast math, rigid
// body, beyond the component-parallel range (N <= 4: > two waves per SIMD) and the record-reading multi-wave range (5..64 points: > two waves per
// SIMD), one point per lane inside a wave, from half a wave per SIMD up.  These kernels carry the fused physics loss (LOSS) as well.
static bool xs_bwd_off() { static const bool off = getenv("MF_BWD_XS") && atoi(getenv("MF_BWD_XS")) == 0; return off; }
static long long xs_bwd_min_waves() {
  static const long long v = getenv("MF_BWD_XS_MIN_WAVES") ? atoll(getenv("MF_BWD_XS_MIN_WAVES")) : device_simds() / 2;
  return v;
}
static bool xs_bwd_shape(const MfRolloutDesc* d, const LaneMap& m) {
  return !xs_bwd_off() && m.PPL == 1 && m.G <= 64 && (long long)d->B * m.G >= xs_bwd_min_waves() * 64;
}
// the component-parallel backward in its EARLY-RECOMPUTE form (no record, more than one wave per SIMD: 4097 .. 8192 rollouts of a <= 4-point
// body, either integrator): dL/dXs formed where the row is consumed (rollout_bwd_cp_kernel.h ONE1 -- instantiations
// of their own, the kernels without the loss are untouched); the value comes from mf_physics_loss_value_* on the forward's rows
bool cp_loss_one_wave(const MfRolloutDesc* d, int scalar_bytes) {
  static const bool off = getenv("MF_CP_LOSS_ONE_WAVE") && atoi(getenv("MF_CP_LOSS_ONE_WAVE")) == 0;      // A/B: the unfused route
  if (off || !d || d->layout != MF_LAYOUT_TIME_MAJOR || d->has_joints || cp_loss_fusable(d)) return false;
  if (scalar_bytes != 4 && d->points_per_lane != MF_LANES_COMPONENT) return false;
  // the record-reading form and late recompute (up to one wave per SIMD): no gain (profiles/r6_ab_one_wave_loss.txt)
  if (cp_record_bytes(d, scalar_bytes) > 0 || ((long long)d->B * 16 + 63) / 64 <= (long long)device_simds()) return false;
  MfRolloutBwdBufs none{};
  return use_component_parallel_bwd(d, &none, scalar_bytes);
}
bool xs_loss_fusable(const MfRolloutDesc* d) {
  static const bool off = getenv("MF_BWD_XS_LOSS") && atoi(getenv("MF_BWD_XS_LOSS")) == 0;      // A/B: the unfused route (dense dL/dXs rows)
  if (off || !d || d->B <= 0 || d->T <= 0 || d->N <= 0 || d->N > 64 || d->has_joints) return false;
  if (d->math_mode != MF_MATH_FAST || d->layout != MF_LAYOUT_TIME_MAJOR) return false;
  if (d->integrator != MF_INTEG_ODEINT_EULER && d->integrator != MF_INTEG_DYNAMICS) return false;
  if (d->points_per_lane == MF_LANES_COMPONENT || d->points_per_lane == 4) return false;
  MfRolloutBwdBufs none{};
  if (use_component_parallel_bwd(d, &none, 4)) return false;      // the component-parallel kernels' range
  if (mw_record_bytes(d, 4) > 0) return false;                     // the record-reading multi-wave kernels' range
  return xs_bwd_shape(d, choose_lane_map(d->B, d->N, d->points_per_lane));
}

// the positions-only launches whose cell gradients leave through the workgroup's LDS window (rollout_bwd_kernel.h WIN): four lanes per rollout
// on ONE shared map pair with a power-of-two side
static bool xs_win_off() { static const bool off = getenv("MF_BWD_WIN") && atoi(getenv("MF_BWD_WIN")) == 0; return off; }
static bool xs_win_shape(const MfRolloutDesc* d, const LaneMap& m) {
  return !xs_win_off() && d->map_shared && m.G == 4 && d->H == d->W && (d->H & (d->H - 1)) == 0;
}
bool xs_bwd_window(const MfRolloutDesc* d) {
  if (!d || d->B <= 0 || d->N <= 0 || d->N > 4 || d->has_joints || d->math_mode != MF_MATH_FAST) return false;
  if (d->points_per_lane == MF_LANES_COMPONENT || d->points_per_lane == 4) return false;
  MfRolloutBwdBufs none{};
  if (use_component_parallel_bwd(d, &none, 4)) return false;
  const LaneMap m = choose_lane_map(d->B, d->N, d->points_per_lane);
  return xs_bwd_shape(d, m) && xs_win_shape(d, m);
}

template <typename S>
static int rollout_bwd(const MfRolloutDesc* d, const MfRolloutBwdBufs* p, void* stream) {
  MF_REQUIRE(d && p, MF_ERR_INVALID, "rollout_bwd: null descriptor");
  MF_REQUIRE(d->B > 0 && d->N > 0 && d->H > 1 && d->W > 0 && d->T >= 1, MF_ERR_INVALID, "rollout_bwd: B, T, N, H, W must be positive");
  MF_REQUIRE(d->n_tracks == 2 || d->n_tracks == 4, MF_ERR_INVALID, "n_tracks must be 2 or 4");
  MF_REQUIRE(d->integrator == MF_INTEG_DYNAMICS || d->integrator == MF_INTEG_ODEINT_EULER, MF_ERR_INVALID,
             "rollout_bwd: unknown integrator");
  MF_REQUIRE(p->z && p->controls && p->ts && p->points && p->part && p->x_init && p->xd0 && p->R0 && p->w0, MF_ERR_INVALID,
             "rollout_bwd: null input buffer");
  MF_REQUIRE(p->Xraw && p->Xds && p->Rs && p->Omegas, MF_ERR_INVALID, "rollout_bwd: null saved-state buffer");
  MF_REQUIRE(p->gz && p->gxd0 && p->gR0 && p->gw0, MF_ERR_INVALID, "rollout_bwd: null gradient output buffer");
  MF_REQUIRE(d->N <= 512, MF_ERR_UNSUPPORTED, "rollout_bwd: more than 512 contact points");
  MF_REQUIRE(d->H < (1 << 23), MF_ERR_UNSUPPORTED, "rollout_bwd: grid too large (H must be below 2^23)");
  MF_REQUIRE(d->H >= 2, MF_ERR_INVALID, "rollout_bwd: the grid needs at least 2 x 2 cells");
  MF_REQUIRE(d->controls_stride_b == 0 && d->controls_stride_t == 0, MF_ERR_UNSUPPORTED, "rollout_bwd: controls must be contiguous [B][T][2]");
  MF_REQUIRE(d->map_shared || (long long)d->B * d->H * d->W * (long long)sizeof(S) < (1ll << 32), MF_ERR_UNSUPPORTED,
             "rollout_bwd: per-rollout maps of 4 GiB or more in total (use a shared map or split the batch)");
  MF_REQUIRE((long long)(d->grad_copies > 1 ? d->grad_copies : 1) * d->H * d->W * (long long)sizeof(S) < (1ll << 32), MF_ERR_UNSUPPORTED,
             "rollout_bwd: gradient copies of 4 GiB or more in total");
  int block = d->block ? d->block : 64;
  MF_REQUIRE(block == 64 || block == 128 || block == 256, MF_ERR_INVALID, "rollout_bwd: block must be 64, 128 or 256");

  RolloutBwdArgs<S> a;
  a.B = d->B; a.T = d->T; a.N = d->N; a.H = d->H; a.W = d->W;
  a.n_tracks = d->n_tracks; a.layout = d->layout; a.map_shared = d->map_shared; a.skip_snap = d->skip_snap;
  a.grad_copies = d->grad_copies > 1 ? d->grad_copies : 1;
  a.mass = (S)d->mass; a.inv_mass = (S)(1.0 / d->mass); a.inv_res = (S)(1.0 / (double)(S)d->grid_res); a.mg = (S)(d->mass * d->gravity); a.k = (S)d->stiffness; a.damp = (S)d->damping;
  a.omega_max = (S)d->omega_max; a.res = (S)d->grid_res; a.d_max = (S)d->d_max; a.dt = (S)d->dt;
  a.half_ly = (S)(d->robot_size_y / 2.0);
  a.sink = (S)(d->mass * d->gravity / (d->stiffness + 1e-6));
  for (int i = 0; i < 9; ++i) a.Iinv[i] = (S)d->Iinv[i];
  for (int i = 0; i < 12; ++i) a.joint_xyz[i] = (S)d->joint_xyz[i];
  a.joint_angles = (const S*)p->joint_angles;
  a.gjoint = (S*)p->gjoint_angles;
  a.rec = nullptr;
  a.zmu = nullptr;
  a.loss_T2 = 0; a.loss_gt = nullptr; a.loss_row_stamp = nullptr; a.loss_row_w = nullptr; a.loss_gloss = nullptr; a.loss_inv_count = (S)0;
  a.loss_partial = nullptr; a.loss_ticket = nullptr; a.loss_out = nullptr; a.loss_near = nullptr; a.loss_w = nullptr;
  MF_REQUIRE(!p->gjoint_angles || p->joint_angles, MF_ERR_INVALID, "rollout_bwd: gjoint_angles without joint_angles");
  MF_REQUIRE(!d->has_joints == !p->joint_angles, MF_ERR_INVALID, "rollout_bwd: joint_angles must be given exactly when desc->has_joints is set");
  MF_REQUIRE(!p->joint_angles || d->n_tracks == 4, MF_ERR_INVALID, "rollout_bwd: joint angles need the 4 driving parts of robot 'marv'");
  a.z = (const S*)p->z; a.mu = (const S*)p->mu; a.controls = (const S*)p->controls; a.ts = (const S*)p->ts;
  a.points = (const S*)p->points; a.part = p->part;
  a.x_init = (const S*)p->x_init; a.xd0 = (const S*)p->xd0; a.R0 = (const S*)p->R0; a.w0 = (const S*)p->w0;
  a.Xraw = (const S*)p->Xraw; a.Xds = (const S*)p->Xds; a.Rs = (const S*)p->Rs; a.Om = (const S*)p->Omegas;
  if (p->loss) {      // the forward's fused physics loss: dL/dXs is formed inside the kernel from Xs, the ground truth and gloss
    const MfRolloutLoss* L = p->loss;
    const bool loss_cp = (sizeof(S) == 4 || d->points_per_lane == MF_LANES_COMPONENT) && cp_loss_fusable(d) && p->rec && !p->joint_angles;
    const bool loss_xs = !loss_cp && sizeof(S) == 4 && xs_loss_fusable(d) && !p->joint_angles;
    // (3: the one-wave forms of the component-parallel backward -- record read by the computing wave, early / late recompute)
    const bool loss_cp1 = !loss_cp && !loss_xs && cp_loss_one_wave(d, (int)sizeof(S)) && !p->joint_angles && !(L->flags & MF_LOSS_VALUE_IN_BACKWARD);
    MF_REQUIRE(!(loss_cp1 || loss_xs) || (L->near && L->w), MF_ERR_INVALID, "rollout_bwd: this fused loss reads MfRolloutLoss.near and .w");
    MF_REQUIRE(loss_cp || loss_xs || loss_cp1, MF_ERR_UNSUPPORTED,
               "rollout_bwd: this launch cannot carry the fused physics loss (mf_rollout_loss_fusable: 1 = the streaming component-parallel backward, "
               "the forward's record required; 2 = the saturated positions-only kernels)");
    MF_REQUIRE((long long)d->B * L->T2 * 3 * (long long)sizeof(S) < (1ll << 32), MF_ERR_UNSUPPORTED, "rollout_bwd: ground truth of 4 GiB or more");
    MF_REQUIRE(!p->gXs && !p->gXds && !p->gRs && !p->gOmegas && !p->gFs && !p->gFf, MF_ERR_INVALID,
               "rollout_bwd: with a fused loss the six upstream gradients must be NULL");
    MF_REQUIRE(L->T2 > 0 && L->gt && L->row_stamp && L->row_w && L->gloss && L->Xs, MF_ERR_INVALID, "rollout_bwd: incomplete MfRolloutLoss");
    a.loss_T2 = L->T2; a.loss_gt = (const S*)L->gt; a.loss_row_stamp = L->row_stamp; a.loss_row_w = (const S*)L->row_w; a.loss_gloss = (const S*)L->gloss;
    a.loss_inv_count = (S)(1.0 / ((double)d->B * L->T2 * 3));
    a.loss_near = L->near; a.loss_w = (const S*)L->w;
    if (L->flags & MF_LOSS_VALUE_IN_BACKWARD) {      // the fetching waves also form the loss value
      MF_REQUIRE(L->partial && L->ticket && L->loss, MF_ERR_INVALID, "rollout_bwd: MF_LOSS_VALUE_IN_BACKWARD needs MfRolloutLoss.partial / ticket / loss");
      a.loss_partial = (S*)L->partial; a.loss_ticket = L->ticket; a.loss_out = (S*)L->loss;
    }
  }
  const bool any_null = !p->gXs || !p->gXds || !p->gRs || !p->gOmegas || !p->gFs || !p->gFf;
  MF_REQUIRE(!any_null || p->zeros, MF_ERR_INVALID,
             "rollout_bwd: an upstream gradient is NULL but `zeros` (>= max(9, 3) zero scalars) was not provided");
  const S* zr = (const S*)p->zeros;
  a.gXs = p->gXs ? (const S*)p->gXs : zr;       a.sXs = p->gXs ? 3 : 0;
  if (p->loss) { a.gXs = (const S*)p->loss->Xs; a.sXs = 3; }      // the fetching waves read Xs rows where they would read dL/dXs rows
  a.gXds = p->gXds ? (const S*)p->gXds : zr;    a.sXds = p->gXds ? 3 : 0;
  a.gOm = p->gOmegas ? (const S*)p->gOmegas : zr; a.sOm = p->gOmegas ? 3 : 0;
  a.gRs = p->gRs ? (const S*)p->gRs : zr;       a.sRs = p->gRs ? 9 : 0;
  a.gFs = p->gFs ? (const S*)p->gFs : zr;       a.sFs = p->gFs ? 3 : 0;
  a.gFf = p->gFf ? (const S*)p->gFf : zr;       a.sFf = p->gFf ? 3 : 0;
  a.gz = (S*)p->gz; a.gmu = (S*)p->gmu; a.gcontrols = (S*)p->gcontrols;
  a.gc_sb = 2 * d->T; a.gc_st = 2;
  if (!p->gcontrols) { a.gcontrols = (S*)p->gw0; a.gc_sb = 3; a.gc_st = 0; }      // one-point-per-lane kernels (RolloutBwdArgs.gc_sb); the others test for NULL
  a.gx0 = (S*)p->gx0; a.gxd0 = (S*)p->gxd0; a.gR0 = (S*)p->gR0; a.gw0 = (S*)p->gw0;

  hipStream_t st = (hipStream_t)stream;
  if (p->joint_angles) {   // articulated body: exact arithmetic, default lane mappings (the backward recomputes every step, so it
                           // need not mirror the forward's mapping)
    const LaneMap mj = choose_lane_map(d->B, d->N, 0);
    if (sizeof(S) == 4 && d->math_mode == MF_MATH_FAST)
      return launch_rollout_bwd_joints_fast_f32(*reinterpret_cast<const RolloutBwdArgs<float>*>(&a), mj, d->integrator, block, st);
    if (sizeof(S) == 4) return launch_rollout_bwd_joints_f32(*reinterpret_cast<const RolloutBwdArgs<float>*>(&a), mj, d->integrator, block, st);
    return launch_rollout_bwd_joints_f64(*reinterpret_cast<const RolloutBwdArgs<double>*>(&a), mj, d->integrator, block, st);
  }
  // float32: the dispatcher's choice for few rollouts of a small body; float64: the VALIDATION build of the same kernels, on explicit
  // request only (points_per_lane = MF_LANES_COMPONENT; rollout_bwd_cp_f64.hip)
  const bool cp = (sizeof(S) == 4 || d->points_per_lane == MF_LANES_COMPONENT) && use_component_parallel_bwd(d, p, (int)sizeof(S));
  if (cp) {   // few rollouts of a small body: a rollout over 16 lanes
    a.gcontrols = (S*)p->gcontrols;      // (these kernels compile the control gradient out instead: GCTRL)
    if (p->rec && cp_record_bytes(d, (int)sizeof(S)) > 0) {      // the forward kept its per-step record: read it instead of recomputing
      MF_REQUIRE(((uintptr_t)p->rec & 15) == 0, MF_ERR_INVALID, "rollout_bwd: rec must be 16-byte aligned");
      a.rec = (const S*)p->rec;
    }
    const bool xs_only = (p->gXs || p->loss) && !p->gXds && !p->gRs && !p->gOmegas && !p->gFs && !p->gFf;
    if constexpr (sizeof(S) == 4) {
      if ((p->zmu || p->zmu_scratch) && cp_bwd_wants_zmu(d, a.rec != nullptr, p->mu != nullptr)) {      // interleaved (z, mu) for the record-reading kernel
        MF_REQUIRE((((uintptr_t)p->zmu_scratch | (uintptr_t)p->zmu) & 7) == 0, MF_ERR_INVALID, "rollout_bwd: zmu_scratch / zmu must be 8-byte aligned");
        if (p->zmu) a.zmu = (const S*)p->zmu;
        else {
          const int n = d->H * d->W;
          hipLaunchKernelGGL((interleave_maps_bwd_kernel<S>), dim3((n + 255) / 256), dim3(256), 0, st, a.z, a.mu, n, (cp::Pk2<S>*)p->zmu_scratch);
          a.zmu = (const S*)p->zmu_scratch;
        }
      }
      return launch_rollout_bwd_cp_f32(a, d->integrator, xs_only, st);
    }
    else return launch_rollout_bwd_cp_f64(a, d->integrator, xs_only, st);
  }
  const LaneMap m = choose_lane_map(d->B, d->N, d->points_per_lane == MF_LANES_COMPONENT ? 0 : d->points_per_lane);
  // one rollout over several waves, from the forward's 16-byte record (float64: the validation build, on explicit request)
  if ((sizeof(S) == 4 || d->points_per_lane == MF_LANES_COMPONENT) && use_multiwave_bwd(d, p)) {
    MF_REQUIRE(((uintptr_t)p->rec & (4 * sizeof(S) - 1)) == 0, MF_ERR_INVALID, "rollout_bwd: rec must be aligned to its quads");
    a.rec = (const S*)p->rec;
    a.gcontrols = (S*)p->gcontrols;      // (tested for NULL by the kernel)
    const bool xs_only = !p->gXds && !p->gRs && !p->gOmegas && !p->gFs && !p->gFf;
    if constexpr (sizeof(S) == 4) return launch_rollout_bwd_mw_f32(a, m.G, d->integrator, xs_only, st);
    else return launch_rollout_bwd_mw_f64(a, m.G, d->integrator, xs_only, st);
  }
  if (sizeof(S) == 4 && d->math_mode == MF_MATH_FAST) {
    // accumulator carry-over between adjacent cells (rollout_bwd_kernel.h): ~55 more instructions per step, half the atomics --
    // a gain from ~3 waves per 4 CUs upwards (B = 4096 at N = 4: 1.00 -> 0.94 ms; B = 65536: 9.5 -> 5.5 ms), a loss below
    const RolloutBwdArgs<float>& af = *reinterpret_cast<const RolloutBwdArgs<float>*>(&a);
    // positions-only upstream (physics_loss) on a one-point-per-lane mapping inside a wave, from half a wave per SIMD up: the XS_ONLY
    // kernels -- and, for ONE shared map pair with a friction map, the interleaved (z, mu) copy (the caller's staged pair, or the
    // scratch it offers, refilled here: one 65 536-cell pass in front of a launch of >= 1 ms).  MF_BWD_XS=0 / MF_BWD_XS_ZMU=0: A/B.
    static const bool xs_zmu_off = getenv("MF_BWD_XS_ZMU") && atoi(getenv("MF_BWD_XS_ZMU")) == 0;
    const bool xs_only = (p->gXs || p->loss) && !p->gXds && !p->gRs && !p->gOmegas && !p->gFs && !p->gFf;
    const bool fused = p->loss != nullptr;      // LOSS instantiations: dL/dXs formed from the Xs rows, the ground truth and the stamp tables
    // (from half a wave per SIMD: right above the component-parallel kernels' range -- 10 240 / 12 288 / 14 336 rollouts of the 4-point body
    //  1.14 / 1.39 / 1.60 ms on the general kernels, 0.91 / 0.90 / 0.93 here, tools/ab_between.sh; MF_BWD_XS_MIN_WAVES overrides)
    if (xs_only && xs_bwd_shape(d, m)) {
      RolloutBwdArgs<float> ax = af;
      bool zmu = false;
      if (!xs_zmu_off && d->map_shared && p->mu && (p->zmu || p->zmu_scratch) && (long long)d->H * d->W * 8 < (1ll << 31)) {
        MF_REQUIRE((((uintptr_t)p->zmu_scratch | (uintptr_t)p->zmu) & 7) == 0, MF_ERR_INVALID, "rollout_bwd: zmu_scratch / zmu must be 8-byte aligned");
        if (p->zmu) ax.zmu = (const float*)p->zmu;
        else {
          const int n = d->H * d->W;
          hipLaunchKernelGGL((interleave_maps_bwd_kernel<float>), dim3((n + 255) / 256), dim3(256), 0, st, ax.z, ax.mu, n, (cp::Pk2<float>*)p->zmu_scratch);
          ax.zmu = (const float*)p->zmu_scratch;
        }
        zmu = true;
      }
      // ... and, four lanes per rollout on ONE shared map pair: the accumulators' writes go to a 128 x 128-cell LDS window per workgroup
      // (rollout_bwd_kernel.h WIN; 128 KB of LDS = one workgroup per CU: 256 threads at one wave per SIMD, 512 from two up).  MF_BWD_WIN=0: A/B.
      if (xs_win_shape(d, m)) {      // (power-of-two side: cell -> window row / column by shift and mask)
        const long long waves = ((long long)d->B * m.G + 63) / 64;
        const bool two = waves >= 2ll * device_simds();      // (two waves per SIMD: eight-wave workgroups, accumulator carry-over)
        if (fused) return launch_rollout_bwd_xs_win_loss_fast_f32(ax, m, d->integrator, two ? 512 : 256, zmu, two, st);
        return launch_rollout_bwd_xs_win_fast_f32(ax, m, d->integrator, two ? 512 : 256, zmu, two, st);
      }
      if (fused) return launch_rollout_bwd_xs_loss_fast_f32(ax, m, d->integrator, block, zmu, st);
      return launch_rollout_bwd_xs_fast_f32(ax, m, d->integrator, block, zmu, st);
    }
    MF_REQUIRE(!fused, MF_ERR_UNSUPPORTED, "rollout_bwd: this launch cannot carry the fused physics loss (mf_rollout_loss_fusable)");
    // ... and one rollout per wave with several points per lane (65 .. 512 points beyond the multi-wave range), positions-only upstream
    static const bool xs_ppl_off = getenv("MF_BWD_XS_PPL") && atoi(getenv("MF_BWD_XS_PPL")) == 0;      // A/B: the general kernel
    if (!xs_ppl_off && !xs_bwd_off() && xs_only && !fused && m.G == 64 && (m.PPL == 2 || m.PPL == 4 || m.PPL == 8) && (long long)d->B >= xs_bwd_min_waves())
      return launch_rollout_bwd_xs_ppl_fast_f32(af, m, d->integrator, block, st);
    if ((long long)d->B * m.G >= 3ll * device_cus() / 4 * 64) return launch_rollout_bwd_carry_fast_f32(af, m, d->integrator, block, st);
    return launch_rollout_bwd_fast_f32(af, m, d->integrator, block, st);
  }
  MF_REQUIRE(!p->loss, MF_ERR_UNSUPPORTED, "rollout_bwd: this launch cannot carry the fused physics loss (mf_rollout_loss_fusable)");
  return launch_rollout_bwd<S, false>(a, m, d->integrator, block, st);
}

}  // namespace mf

// 0 = no; 1 = both directions on the component-parallel kernels with the streaming backward (value in the forward launch, in the backward
// launch -- MF_LOSS_VALUE_IN_BACKWARD -- or from mf_physics_loss_value_*); 2 = the BACKWARD of a saturated launch (positions-only one-point-
// per-lane kernels): pass MfRolloutBwdBufs.loss with flags = 0, take the value from mf_physics_loss_value_* on the forward's rows
// 3 = the BACKWARD of a component-parallel launch in its early-recompute form (4097 .. 8192 rollouts): as 2, without MF_LOSS_VALUE_IN_BACKWARD
extern "C" int mf_rollout_loss_fusable(const MfRolloutDesc* d) {
  return mf::cp_loss_fusable(d) ? 1 : (mf::xs_loss_fusable(d) ? 2 : (mf::cp_loss_one_wave(d, 4) ? 3 : 0));
}
// 1 where a positions-only backward of this shape (float32) sends its cell gradients through the workgroups' LDS windows: a workgroup then adds
// its window to gradient copy blockIdx % grad_copies ONCE, at its end -- few copies suffice (the caller's reduction over them is what grows)
extern "C" int mf_rollout_bwd_window(const MfRolloutDesc* d) { return mf::xs_bwd_window(d) ? 1 : 0; }
extern "C" int mf_rollout_bwd_f32(const MfRolloutDesc* d, const MfRolloutBwdBufs* p, void* s) {
  return mf::rollout_bwd<float>(d, p, s);
}
extern "C" int mf_rollout_bwd_f64(const MfRolloutDesc* d, const MfRolloutBwdBufs* p, void* s) {
  return mf::rollout_bwd<double>(d, p, s);
}

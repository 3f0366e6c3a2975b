// Fused DPhysics rollout, backward pass (gfx950): reverse-time adjoint of rollout_fwd.hip.
//
// Replaces the autograd graph the reference builds through `forward_kinematics` / `dynamics` / `dynamics_odeint`
// (/root/reference/monoforce/src/monoforce/models/traj_predictor/dphysics.py:172-272, :467-528; T x ~300 nodes) with one
// kernel: walk the steps backwards, reload the state each step started from (they are the forward's own outputs, so
// checkpoints are free), recompute that step's intermediates, and apply the hand-derived vector-Jacobian product.
// Same lane mapping as the forward (G lanes per rollout, PPL points per lane, adjoint state replicated across the group,
// cross-lane sums on DPP).  Gradients w.r.t. the height / friction cells are scattered with hardware float atomics.
//
// Autograd conventions reproduced: clamp passes gradient iff lo <= x <= hi; `.long()` cell indices are
// constants (queries influence samples only through the fractions); |v| has zero gradient at v = 0;
// x / clamp(|x|, eps) differentiates through |x| only when |x| >= eps.
#pragma once
#include "rollout_fwd_kernel.h"   // Mth<>, locate_m<>, LaneMap

namespace mf {

template <typename S>
struct RolloutBwdArgs {
  int B, T, N, H, W, n_tracks, layout, map_shared, skip_snap, grad_copies;
  S mass, inv_mass, mg, k, damp, omega_max, res, inv_res, d_max, dt, half_ly, sink;
  S Iinv[9];
  const S *z, *mu, *controls, *ts, *points;
  const int* part;
  const S *x_init, *xd0, *R0, *w0;
  const S *Xraw, *Xds, *Rs, *Om;
  const S *gXs, *gXds, *gRs, *gOm, *gFs, *gFf;   // never NULL here: the host substitutes `zeros` with a zero stride
  int sXs, sXds, sRs, sOm, sFs, sFf;              // floats per row element group: 3 / 9 (present) or 0 (absent -> zeros)
  S *gz, *gmu, *gcontrols, *gx0, *gxd0, *gR0, *gw0;
  const S* joint_angles;   // [B,T,4] flipper angles, or NULL
  S joint_xyz[12];
  S* gjoint;               // [B,T,4] out: gradient of the flipper angles, or NULL
  const S* rec;            // component-parallel kernels: the forward's per-step record (rollout_fwd_cp_kernel.h), or NULL
  const S* zmu;            // record-reading component-parallel kernel (ZMU): the shared maps interleaved, S[H*W][2] = (z, mu), or NULL
  // fused physics loss (MfRolloutLoss; streaming component-parallel backward): gXs points at the forward's Xs rows then
  int loss_T2;
  const S* loss_gt;        // NULL: no fused loss
  const int* loss_row_stamp;
  const S* loss_row_w;
  const S* loss_gloss;
  S loss_inv_count;
  S* loss_partial;         // MF_LOSS_VALUE_IN_BACKWARD (NULL otherwise): per-workgroup partial sums, the ticket, the mean
  unsigned* loss_ticket;
  S* loss_out;
  // (round 6, appended: the offsets of everything above are those of round 5) the per-STAMP tables of the fused loss, for the one-point-per-
  // lane LOSS kernels: near[j] = output row of stamp j (strictly increasing), w[j] = its weight
  const int* loss_near;
  const S* loss_w;
  // element strides of the control-gradient rows over rollouts / over steps: (2 T, 2) = the caller's [B][T][2]; (3, 0) = nobody wants the
  // control gradient -- gcontrols then points at gw0 and every step's pair lands on the rollout's own gw0 row, which the kernel's final
  // store overwrites (same lanes, program order): the same instruction stream without 8 B per rollout-step of stores (65 MB at 16 384 x 500)
  int gc_sb, gc_st;
};

#ifdef MF_NO_ATOMICS
__device__ __forceinline__ void atomic_add(float* p, float v) { if (v == 1.2345e-30f) *p = v; }
#else
__device__ __forceinline__ void atomic_add(float* p, float v) { unsafeAtomicAdd(p, v); }
#endif
__device__ __forceinline__ void atomic_add(double* p, double v) { unsafeAtomicAdd(p, v); }

// LDS float add without return (ds_add_f32); `p` must point into __shared__ memory
__device__ __forceinline__ void lds_add(float* p, float v) {
  __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false);
}
__device__ __forceinline__ void lds_add(double* p, double v) { unsafeAtomicAdd(p, v); }
template <typename S>
__device__ __forceinline__ bool inside(S v, S lo, S hi) { return v >= lo && v <= hi; }
template <typename S>
__device__ __forceinline__ S* at32(S* base, unsigned elem) {
  return reinterpret_cast<S*>(reinterpret_cast<char*>(base) + (size_t)(elem * (unsigned)sizeof(S)));
}

#define MF_CROSS(o, a, b)                    \
  do {                                       \
    (o)[0] = (a)[1] * (b)[2] - (a)[2] * (b)[1]; \
    (o)[1] = (a)[2] * (b)[0] - (a)[0] * (b)[2]; \
    (o)[2] = (a)[0] * (b)[1] - (a)[1] * (b)[0]; \
  } while (0)

// XS_ONLY (round 5): the only upstream gradient is dL/dXs (physics_loss, losses.py:102-127: positions at the stamped rows) -- the
// other five row loads per step, their adds and the impulse adjoints are compiled out.  ZMU: ONE shared (z, mu) pair read interleaved,
// two 16-byte loads per footprint instead of eight 4-byte ones (gather4x2, rollout_fwd_kernel.h).  Both serve the saturated launches of
// <= 4-point bodies (B > 8192), which PMC shows bound by the CU's L1 address path, not by VALU issue
// (profiles/r5_pmc_backward_B16384.txt): the distinct addresses per wave-step drop from ~1060 to ~650.
// WIN (round 5, saturated launches of a shared map pair): what the register accumulators write when a point changes cell goes to a
// kWinW x kWinW-cell window of both gradient maps in LDS instead of out as device-scope float atomics -- those execute at the memory
// side and sit in the in-order vmcnt queue in front of the next step's row loads (a 64-lane wave flushes somebody's cells on nearly
// every step): with the atomics compiled out the backward of 32 768 rollouts takes 1.33 ms instead of 2.57 (profiles/r5_ab_bwd_win.txt).
// ds_add_f32 costs ~12 cycles per ACTIVE lane (tools/microbench/lds_atomics.hip: 768 cycles for a full wave, whatever the addresses;
// ds_add_u32 / _u64: ~30) -- affordable for the ~50 cell changes per wave-step, not for eight adds per lane and step (measured: no gain),
// so the accumulators and their carry-over stay.  The window is centred on the start of the workgroup's first rollout; a cell outside
// it takes the global atomic as before.  One pass at the end adds the non-zero window cells to gradient copy blockIdx % grad_copies.
constexpr int kWinW = 128;
// (workgroup-uniform) origin of the window: centred on the start of the workgroup's first rollout `b0`, inside the map; zero-fills the
// window; a barrier follows in the caller
template <typename S, bool FAST>
__device__ __forceinline__ void win_open(const RolloutBwdArgs<S>& a, S* win, int b0, int* wx0, int* wy0) {
  b0 = min(b0, a.B - 1);
  const int cx = (int)mf_clamp(Mth<S, FAST>::cell_coord(a.x_init[b0 * 3 + 0], a.d_max, a.res, a.inv_res), (S)-262144.0, (S)262144.0);
  const int cy = (int)mf_clamp(Mth<S, FAST>::cell_coord(a.x_init[b0 * 3 + 1], a.d_max, a.res, a.inv_res), (S)-262144.0, (S)262144.0);
  *wx0 = __builtin_amdgcn_readfirstlane(max(min(cx - kWinW / 2, a.H - kWinW), 0));
  *wy0 = __builtin_amdgcn_readfirstlane(max(min(cy - kWinW / 2, a.H - kWinW), 0));
  typedef float f4 __attribute__((ext_vector_type(4)));
  for (int i = threadIdx.x; i < 2 * kWinW * kWinW / 4; i += blockDim.x) reinterpret_cast<f4*>(win)[i] = f4{0.f, 0.f, 0.f, 0.f};
}
// one cell's pair into the window if the cell (flat map index `idx`, H = 2^shift) lies inside it; false: the caller takes the atomics
template <typename S>
__device__ __forceinline__ bool win_emit(S* win, unsigned win_flat0, unsigned win_shift, unsigned h_mask, unsigned idx, S vz, S vm, bool want_gmu) {
  const unsigned d = idx - win_flat0, rx = d >> win_shift, ry = d & h_mask;
  if ((rx < (unsigned)kWinW) & (ry < (unsigned)kWinW)) {
    S* wz = win + (rx * kWinW + ry);
    lds_add(wz, vz);
    if (want_gmu) lds_add(wz + kWinW * kWinW, vm);
    return true;
  }
  return false;
}
// after the workgroup's last emit and a barrier: the non-zero window cells to gradient copy blockIdx % grad_copies, coalesced
template <typename S>
__device__ __forceinline__ void win_close(const RolloutBwdArgs<S>& a, const S* win, int wx0, int wy0) {
  const unsigned HW = (unsigned)a.H * (unsigned)a.W;
  const unsigned coff = (a.map_shared ? (unsigned)(blockIdx.x % a.grad_copies) : 0u) * HW;
  const bool want_gmu = a.gmu != nullptr && a.mu != nullptr;
  for (int i = threadIdx.x; i < kWinW * kWinW; i += blockDim.x) {
    const int ix = wx0 + i / kWinW, iy = wy0 + i % kWinW;
    if (ix < a.H && iy < a.H) {
      const unsigned cell = (unsigned)iy + (unsigned)a.H * (unsigned)ix;
      const S vz = win[i];
      if (vz != (S)0) atomic_add(at32(a.gz, coff + cell), vz);
      if (want_gmu) {
        const S vm = win[kWinW * kWinW + i];
        if (vm != (S)0) atomic_add(at32(a.gmu, coff + cell), vm);
      }
    }
  }
}
// LOSS (round 6; XS_ONLY launches): `physics_loss` (losses.py:102-127) inside the launch -- a.gXs points at the forward's own Xs rows and the
// kernel forms dL/dXs at the stamped rows itself from Xs, the ground truth and the stamp's weight (the formula of physics_loss.hip,
// same rounding): no [T][B][3] gradient tensor to clear, fill and read (98 MB at 16 384 rollouts, 10 % of its rows non-zero), no loss-
// gradient launch.  The row tables are indexed by the (wave-uniform) output row: scalar loads.
template <typename S>
__device__ __forceinline__ S xs_loss_grad(S scale, S xs, S g, S w) {      // scale = 2 gloss / (B T2 3)
#pragma clang fp contract(off)
  return scale * w * (xs * w - g * w);
}
template <typename S>
__device__ __forceinline__ S xs_loss_term(S xs, S g, S w) {
#pragma clang fp contract(off)
  const S d = xs * w - g * w;
  return d * d;
}
// MF_LOSS_VALUE_IN_BACKWARD on the one-point-per-lane LOSS kernels: every thread of the workgroup arrives with its share (0 for lanes that
// count nothing); one partial sum per workgroup in a fixed order, and the workgroup that takes the last ticket adds the partial sums in index
// order and writes the mean (as physics_loss_value_kernel does): deterministic for a given launch shape.
template <typename S>
__device__ __forceinline__ void xs_loss_value_finish(const RolloutBwdArgs<S>& a, S acc) {
  if (a.loss_out == nullptr) return;      // (workgroup-uniform)
  __shared__ S wave_sum[8];
  __shared__ unsigned last_wg;
  const int nw = (int)((blockDim.x + 63) >> 6), lane = (int)(threadIdx.x & 63), wv = (int)(threadIdx.x >> 6);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if (lane == 0) wave_sum[wv] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    S tot = (S)0;
    for (int k = 0; k < nw; ++k) tot += wave_sum[k];
    __builtin_nontemporal_store(tot, a.loss_partial + blockIdx.x);
    __threadfence();
    last_wg = atomicAdd(a.loss_ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!last_wg) return;
  __threadfence();
  S tot = (S)0;
  for (unsigned k = threadIdx.x; k < gridDim.x; k += blockDim.x) tot += __builtin_nontemporal_load(a.loss_partial + k);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d, 64);
  __syncthreads();
  if (lane == 0) wave_sum[wv] = tot;
  __syncthreads();
  if (threadIdx.x == 0) {
    S sum = (S)0;
    for (int k = 0; k < nw; ++k) sum += wave_sum[k];
    a.loss_out[0] = sum * a.loss_inv_count;
    *a.loss_ticket = 0u;
  }
}
template <typename S, int G, int PPL, int INTEG, bool FAST, bool JOINTS, bool CARRY, bool XS_ONLY, bool ZMU, bool WIN, bool LOSS = false>
__device__ __forceinline__ void rollout_bwd_body(const RolloutBwdArgs<S>& a, S* win, const unsigned win_flat0, const unsigned win_shift, S* l_acc = nullptr) {
  using M = Mth<S, FAST>;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = tid / G;
  const int gl = tid % G;
  if (b >= a.B) return;
  const S one = (S)1, zero = (S)0;
  const int HW = a.H * a.W, last = HW - 1;
  // group reductions; a rollout spread over several waves (G > 64: one workgroup = one rollout) exchanges through LDS
  __shared__ S gs_lds[G > 64 ? 2 * (G / 64) * kGroupSumMaxValues : 1];
  GroupSum<G, S> gs;
  gs.lds = gs_lds;
  // uniform base pointers + 32-bit element offsets (host guarantees < 4 GiB per array): scalar-base loads and atomics
  const unsigned moff = a.map_shared ? 0u : (unsigned)b * (unsigned)HW;
  const S* zmap = a.z;
  const bool has_mu = a.mu != nullptr;  // wave-uniform
  const S* mumap = has_mu ? a.mu : a.z;
  // shared map: rollout b scatters into private copy b % grad_copies (summed by the caller) -- see monoforce_hip.h
  const unsigned goff = a.map_shared ? (unsigned)(b % a.grad_copies) * (unsigned)HW : (unsigned)b * (unsigned)HW;
  S* gzmap = a.gz;
  const bool want_gmu = a.gmu != nullptr && has_mu;  // wave-uniform
  S* gmumap = want_gmu ? a.gmu : a.gz;

  S P[PPL][3];    // contact points used by the step (articulated per step when JOINTS)
  S P0[PPL][3];   // rest configuration (cfg.robot_points): the terrain snap and the articulation start from it
  S Iv[9];        // inverse inertia used by the step
  int part[PPL];
  bool act[PPL];
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    int i = gl * PPL + j;       // blocked, as in the forward
    act[j] = i < a.N;
    int ii = act[j] ? i : 0;
    P[j][0] = P0[j][0] = a.points[ii * 3 + 0];
    P[j][1] = P0[j][1] = a.points[ii * 3 + 1];
    P[j][2] = P0[j][2] = a.points[ii * 3 + 2];
    part[j] = act[j] ? a.part[ii] : -1;
  }
#pragma unroll
  for (int c = 0; c < 9; ++c) Iv[c] = a.Iinv[c];

  const size_t row_stride = (a.layout == MF_LAYOUT_TIME_MAJOR) ? (size_t)a.B : 1;
  const size_t row0 = (a.layout == MF_LAYOUT_TIME_MAJOR) ? (size_t)b : (size_t)b * a.T;
  const S* ctrl = a.controls + (size_t)b * a.T * 2;
  S* gctrl = a.gcontrols + (size_t)b * (size_t)a.gc_sb;

  // adjoint of the state (x, xd, R, w) [+ the impulse accumulators of the ODEINT extended state]
  // UNSUM (round 5; the positions-only kernels of rollouts inside a wave): the adjoint state is kept UN-SUMMED over the lanes of a rollout
  // -- every lane holds the part its own points contributed, the state is the sum of the parts.  Every use of it is linear, and only
  // the two parts met by per-point data -- the adjoints of the linear and angular velocity -- are summed every step, with the control
  // gradient that is stored: 8 + 5 lane sums per step instead of 23 + 5 (the saturated launches are VALU-bound: r5h_pmc_backward_sat_B32768.txt).
  constexpr bool UNSUM = XS_ONLY && G <= 64 && !JOINTS;
  const S up_lane = (!UNSUM || gl == 0) ? one : zero;      // upstream gradients of the body state enter ONE lane's part
  S lx[3] = {zero, zero, zero}, lxd[3] = {zero, zero, zero}, lw[3] = {zero, zero, zero}, lR[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) lR[c] = zero;
  S laFs[PPL][3], laFf[PPL][3];
#pragma unroll
  for (int j = 0; j < PPL; ++j)
#pragma unroll
    for (int c = 0; c < 3; ++c) laFs[j][c] = laFf[j][c] = zero;

  // Memory inputs of a step that do not depend on the adjoint are loaded AHEAD of their use and always BEFORE the step's
  // atomics in program order: vmcnt is an in-order counter over loads, stores and atomics, so a load issued after the
  // atomics could only be waited for together with them (~1 us under contention).
  //   StateIn  the state step n started from (a saved forward output), its controls and step size -- prefetched one
  //            iteration ahead, right after the gathers;
  //   UpIn     the upstream gradients of the output row a step produced -- loaded just before the atomics of the
  //            following (later-time) step and folded into the adjoint at the top of the next iteration.
  struct StateIn {
    S x[3], xd[3], R[9], w[3];
    S cv, cw, t0, t1;   // raw loads only: nothing here is touched before its use, so no wait is pulled forward
  };
  struct UpIn {
    S gXs[3], gXds[3], gRs[9], gOm[3];
    S gFs[PPL][3], gFf[PPL][3];
    S lg[3], lw;        // LOSS: ground truth of the row's stamp, its weight (raw loads; the gradient is formed where the row is consumed)
    bool stamped;       // LOSS: the row carries a stamp (wave-uniform)
  };
  static_assert(!LOSS || XS_ONLY, "the fused physics loss is a positions-only upstream");
  const S loss_scale = LOSS ? (S)2 * a.loss_gloss[0] * a.loss_inv_count : zero;      // as csrc/physics_loss.hip: (2 gloss) / count
  const S* const loss_gt_b = LOSS ? a.loss_gt + (size_t)b * (size_t)a.loss_T2 * 3u : nullptr;
  // The rows are visited from the last one down and the stamps' rows increase with the stamp: ONE current stamp l_j (its row and weight in
  // registers) is compared with the row at hand and stepped down when it is met; its successor's row and weight are loaded right then -- a
  // whole iteration before their first use, from an address that depends on no load of this iteration -- and the ground-truth address
  // depends on l_j alone.  No branch, no wait of its own.  (First attempt: row_stamp[row] -> ground-truth address, a dependent load in front
  // of the step's requests: 0.93 -> 1.13 ms at 16 384 rollouts.  Second: the successor's loads behind a `l_j >= 0 ?` -- the compiler made it
  // a branch with an s_waitcnt vmcnt(0) inside the loop: 1.21 ms.)
  S l_val = zero;      // LOSS: this lane's share of the loss value (sum of the weighted squared errors of its rollout's stamped rows)
  // (the stamp index lives in a VECTOR register on purpose -- an opaque per-lane zero is added to it: left to itself the compiler sees a
  //  uniform address, keeps the loaded row in a scalar register and puts the v_readfirstlane -- and with it an s_waitcnt for nearly the
  //  whole prefetch group -- right behind the load: 0.93 -> 1.21 ms at 16 384 rollouts, profiles/r6_ab_fused_loss_sat.txt)
  int l_opaque_zero = 0;
  if constexpr (LOSS) asm volatile("v_mov_b32 %0, 0" : "=v"(l_opaque_zero));
  int l_j = LOSS ? a.loss_T2 - 1 + l_opaque_zero : 0;
  int l_row = LOSS ? a.loss_near[l_j] : -1;
  S l_w = LOSS ? a.loss_w[l_j] : zero;
  const S* l_gt = LOSS ? loss_gt_b + (size_t)(unsigned)l_j * 3u : nullptr;
  const int* const l_near_tab = LOSS ? a.loss_near : nullptr;
  const S* const l_w_tab = LOSS ? a.loss_w : nullptr;
  const int n_steps = (INTEG == MF_INTEG_ODEINT_EULER) ? a.T - 1 : a.T;
  // Running pointers to the rows of the step being prefetched: stepping them back by wave-uniform deltas replaces a dozen
  // 64-bit row * stride multiplications per iteration.
  struct Ptrs {
    const S *x, *xd, *w, *R, *c, *t, *g1, *g2, *g3, *g4;
    const S* f1[PPL];
    const S* f2[PPL];
    int trow;           // LOSS: time index of the output row g1 points at (wave-uniform)
  };
  auto make_ptrs = [&](int m, Ptrs& p) {       // rows of step m: the state it started from, the upstream of the row it produced
    const size_t in_row = row0 + (size_t)(INTEG == MF_INTEG_ODEINT_EULER ? m : m - 1) * row_stride;   // (m - 1 unused for m = 0)
    // (T = 1 with the default integrator has no step at all: the prologue's prefetch of "step 0" must stay inside the one row there is --
    //  unclamped it read one row past every upstream array, a fault whenever such an array ended on a page boundary)
    const size_t out_row = row0 + (size_t)(INTEG == MF_INTEG_ODEINT_EULER ? min(m + 1, a.T - 1) : m) * row_stride;
    p.x = a.Xraw + in_row * 3; p.xd = a.Xds + in_row * 3; p.w = a.Om + in_row * 3; p.R = a.Rs + in_row * 9;
    p.c = ctrl + (size_t)m * 2; p.t = a.ts + m;
    p.g1 = a.gXs + out_row * a.sXs; p.g2 = a.gXds + out_row * a.sXds; p.g3 = a.gOm + out_row * a.sOm; p.g4 = a.gRs + out_row * a.sRs;
    p.trow = INTEG == MF_INTEG_ODEINT_EULER ? min(m + 1, a.T - 1) : m;
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      const size_t pt = out_row * a.N + min(gl * PPL + j, a.N - 1);   // clamped: inactive slots read a valid row, masked at use
      p.f1[j] = a.gFs + pt * a.sFs; p.f2[j] = a.gFf + pt * a.sFf;
    }
  };
  const size_t d3 = row_stride * 3, d9 = row_stride * 9;
  const size_t dg1 = row_stride * a.sXs, dg2 = row_stride * a.sXds, dg3 = row_stride * a.sOm, dg4 = row_stride * a.sRs;
  const size_t df1 = row_stride * a.N * a.sFs, df2 = row_stride * a.N * a.sFf;
  auto step_back = [&](Ptrs& p, size_t k) {    // k = 1: one step earlier; k = 0: stay (the prefetch of step 0 repeats itself)
    p.x -= k * d3; p.xd -= k * d3; p.w -= k * d3; p.R -= k * d9; p.c -= k * 2; p.t -= k;
    p.g1 -= k * dg1; p.g2 -= k * dg2; p.g3 -= k * dg3; p.g4 -= k * dg4;
    p.trow -= (int)k;
#pragma unroll
    for (int j = 0; j < PPL; ++j) { p.f1[j] -= k * df1; p.f2[j] -= k * df2; }
  };
  auto load_state = [&](const Ptrs& p, int m, StateIn& s) {
    // DYNAMICS step 0 starts from the initial state, every other step from a saved row: pointer selects, no branch
    const bool init = (INTEG == MF_INTEG_DYNAMICS) && m == 0;
    const S* sx = init ? a.x_init + b * 3 : p.x;
    const S* sxd = init ? a.xd0 + b * 3 : p.xd;
    const S* sw = init ? a.w0 + b * 3 : p.w;
    const S* sR = init ? a.R0 + b * 9 : p.R;
#pragma unroll
    for (int c = 0; c < 3; ++c) { s.x[c] = sx[c]; s.xd[c] = sxd[c]; s.w[c] = sw[c]; }
#pragma unroll
    for (int c = 0; c < 9; ++c) s.R[c] = sR[c];
    s.cv = p.c[0]; s.cw = p.c[1];
    s.t0 = p.t[0]; s.t1 = p.t[m + 1 < a.T ? 1 : 0];
  };
  auto load_upstream = [&](const Ptrs& p, UpIn& u) {
    // absent upstream gradients point at a zero row with stride 0 (host side), so these loads are unconditional
#pragma unroll
    for (int c = 0; c < 3; ++c) u.gXs[c] = p.g1[c];
    if constexpr (LOSS) {      // (unconditional loads: an unstamped row reads the current stamp's ground truth -- the same line again -- masked at use)
      u.stamped = (p.trow == l_row) & (l_j >= 0);
      u.lw = l_w;
#pragma unroll
      for (int c = 0; c < 3; ++c) u.lg[c] = l_gt[c];
      l_j -= u.stamped ? 1 : 0;
      l_gt -= (u.stamped & (l_j >= 0)) ? 3 : 0;      // (running pointer to the current stamp's ground truth; it stays on stamp 0 at the end)
      const unsigned jn = (unsigned)max(l_j, 0);
      l_row = l_near_tab[jn];      // (first used by the NEXT call: a whole iteration for the two loads to land)
      l_w = l_w_tab[jn];
    }
    if constexpr (!XS_ONLY) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { u.gXds[c] = p.g2[c]; u.gOm[c] = p.g3[c]; }
#pragma unroll
      for (int c = 0; c < 9; ++c) u.gRs[c] = p.g4[c];
#pragma unroll
      for (int j = 0; j < PPL; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) { u.gFs[j][c] = p.f1[j][c]; u.gFf[j][c] = p.f2[j][c]; }   // masked by act[] where they are consumed
    }
  };
  auto add_upstream_state = [&](const UpIn& u) {
    S g[3] = {u.gXs[0], u.gXs[1], u.gXs[2]};
    if constexpr (LOSS) {      // dL/dXs of this row: masked by the stamp, not by a zero weight (an unstamped row of a diverged rollout may hold inf / NaN)
      // (selects, not a wave-uniform branch around the stamped rows' arithmetic: the branch put a vmcnt(0) in front of every row --
      //  profiles/r6_ab_fused_loss_sat.txt, 1.00 -> 1.21 ms)
#pragma unroll
      for (int c = 0; c < 3; ++c) g[c] = u.stamped ? xs_loss_grad(loss_scale, u.gXs[c], u.lg[c], u.lw) : zero;
      // MF_LOSS_VALUE_IN_BACKWARD: the weighted squared error of the row as well (one lane of the rollout's group counts it)
      const S e2 = xs_loss_term(u.gXs[0], u.lg[0], u.lw) + xs_loss_term(u.gXs[1], u.lg[1], u.lw) + xs_loss_term(u.gXs[2], u.lg[2], u.lw);
      l_val += (u.stamped & (gl == 0)) ? e2 : zero;
    }
    if constexpr (UNSUM) {
      const S ms = up_lane * a.sink;
      lx[0] += up_lane * g[0]; lx[1] += up_lane * g[1]; lx[2] += up_lane * g[2];
      lR[2] += g[0] * ms; lR[5] += g[1] * ms; lR[8] += g[2] * ms;
    } else {
    lx[0] += g[0]; lx[1] += g[1]; lx[2] += g[2];
    lR[2] += g[0] * a.sink; lR[5] += g[1] * a.sink; lR[8] += g[2] * a.sink;   // Xs = x + R[:,2] * sink
    }
    if constexpr (!XS_ONLY) {
      lxd[0] += u.gXds[0]; lxd[1] += u.gXds[1]; lxd[2] += u.gXds[2];
#pragma unroll
      for (int c = 0; c < 9; ++c) lR[c] += u.gRs[c];
      lw[0] += u.gOm[0]; lw[1] += u.gOm[1]; lw[2] += u.gOm[2];
    }
  };

  if (INTEG == MF_INTEG_ODEINT_EULER) {
    // the last control of the grid is never used by the explicit scheme
    if (gl == 0) { gctrl[(a.T - 1) * a.gc_st + 0] = zero; gctrl[(a.T - 1) * a.gc_st + 1] = zero; }
  }

  // Scatter-add of the cell gradients.  Device-scope float atomics execute at the memory side (the per-XCD L2s are not
  // coherent): every one is a fabric transaction (PMC: WRITE_SIZE = 32 B per atomic, 0.5 GB per launch at B = 1024), and
  // same-address ones serialise.  A robot moves <= 0.2 cell per step, so consecutive steps of a point hit the SAME four
  // cells: their contributions are accumulated in registers (acc_*) and only written when the point changes cell
  // (about every 5th step).  A move to an edge-adjacent cell keeps two of the four cells of the footprint: those two
  // accumulators are carried over to their new slots and only the two cells left behind are written (half the atomics;
  // measured 9.5 -> 5.5 ms at B = 65536).  The write is deferred to the next iteration (st_*), after that step's loads.
  unsigned acc_idx[PPL][4], st_idx[PPL][4];
  S acc_z[PPL][4], acc_m[PPL][4], st_z[PPL][4], st_m[PPL][4];
  bool st_pending[PPL], st_all[PPL];      // slots 0, 1 of the stash are pending / all four are (the point jumped)
  // The carry-over costs ~55 instructions per step: a loss while a launch is bound by the instruction stream of its few
  // waves (B = 1024, N = 4: 0.91 -> 0.93 ms), a gain as soon as the atomics matter (B = 4096: 1.00 -> 0.94 ms).
  constexpr bool carry_over = CARRY;      // compile-time: the float32 fast-math kernels exist in both forms, the host picks
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    st_pending[j] = st_all[j] = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) { acc_idx[j][q] = st_idx[j][q] = 0u; acc_z[j][q] = acc_m[j][q] = st_z[j][q] = st_m[j][q] = zero; }
  }
  // one cell's accumulated pair goes out: WIN -- into the workgroup's LDS window when the cell lies inside it (flat index -> window
  // row / column by shift and mask: the host takes this kernel for power-of-two H only), else (and without WIN) device-scope atomics
  auto emit = [&](unsigned idx, S vz, S vm) {
    if constexpr (WIN) {
      if (win_emit(win, win_flat0, win_shift, (unsigned)(a.H - 1), idx, vz, vm, want_gmu)) return;
    }
    atomic_add(at32(gzmap, goff + idx), vz);
    if (want_gmu) atomic_add(at32(gmumap, goff + idx), vm);
  };
  auto flush_stash = [&]() {
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      if (st_pending[j]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) emit(st_idx[j][q], st_z[j][q], st_m[j][q]);
        if (st_all[j]) {
#pragma unroll
          for (int q = 2; q < 4; ++q) emit(st_idx[j][q], st_z[j][q], st_m[j][q]);
        }
      }
      st_pending[j] = st_all[j] = false;
    }
  };

  // control gradient of the previous iteration, stored one iteration late; before the first one it rewrites the last row with
  // zeros (ODEINT: that row IS zero; DYNAMICS: the first iteration's own result overwrites it, same lanes, program order)
  S* gctrl_pending = gctrl + (size_t)(a.T - 1) * (size_t)a.gc_st;
  S gv_pending = zero, gwc_pending = zero;
  StateIn cur;
  UpIn up;
  Ptrs rp;
  make_ptrs(max(n_steps - 1, 0), rp);
  load_state(rp, max(n_steps - 1, 0), cur);
  load_upstream(rp, up);
  __builtin_amdgcn_s_waitcnt(0);   // nothing loaded before the loop is still in flight when the in-loop waits are counted
  for (int n = n_steps - 1; n >= 0; --n) {
    add_upstream_state(up);
    S x[3], xd[3], R[9], w[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { x[c] = cur.x[c]; xd[c] = cur.xd[c]; w[c] = cur.w[c]; }
#pragma unroll
    for (int c = 0; c < 9; ++c) R[c] = cur.R[c];
    const S cv = cur.cv, cw = cur.cw;
    // the articulated body of this step: a function of the joint angles only, constant w.r.t. everything differentiated
    if (JOINTS) articulate_body<S, G, PPL, FAST>(gs, a.joint_angles + ((size_t)b * a.T + n) * 4, a.joint_xyz, a.mass / (S)a.N, P0, part, act, P, Iv);

    // ---------------------------------------------------------------------------------------------------
    // forward recompute (identical arithmetic to rollout_fwd.hip)
    // ---------------------------------------------------------------------------------------------------
    Cell<S> cell[PPL];
    S zc4[PPL][4], mc4[PPL][4];
    S r[PPL][3], vp[PPL][3], nrm[PPL][3], nl[PPL], muq[PPL], cj[PPL], Aj[PPL], F0[PPL][3], pzv[PPL];
    S csum = zero;
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      S px = P[j][0] * R[0] + P[j][1] * R[1] + P[j][2] * R[2] + x[0];
      S py = P[j][0] * R[3] + P[j][1] * R[4] + P[j][2] * R[5] + x[1];
      S pz = P[j][0] * R[6] + P[j][1] * R[7] + P[j][2] * R[8] + x[2];
      pzv[j] = pz;
      r[j][0] = px - x[0]; r[j][1] = py - x[1]; r[j][2] = pz - x[2];
      vp[j][0] = xd[0] + (w[1] * r[j][2] - w[2] * r[j][1]);
      vp[j][1] = xd[1] + (w[2] * r[j][0] - w[0] * r[j][2]);
      vp[j][2] = xd[2] + (w[0] * r[j][1] - w[1] * r[j][0]);
      cell[j] = locate_m<S, FAST>(px, py, a.d_max, a.res, a.inv_res, a.H, last);
      const Cell<S>& c = cell[j];
      if constexpr (ZMU) {      // the footprint in both maps as two 16-byte loads of the interleaved (z, mu) pair
        gather4x2(a.zmu, c, last, zc4[j], mc4[j]);
      } else {
      zc4[j][0] = ld32(zmap, moff + (unsigned)c.ic); zc4[j][1] = ld32(zmap, moff + (unsigned)c.i_f); zc4[j][2] = ld32(zmap, moff + (unsigned)c.il); zc4[j][3] = ld32(zmap, moff + (unsigned)c.ifl);
      // unconditional (mumap aliases z when there is no friction map; the select follows the blend): one basic block, and
      // no consumer of the gathers ahead of the loads and atomics issued below
      mc4[j][0] = ld32(mumap, moff + (unsigned)c.ic); mc4[j][1] = ld32(mumap, moff + (unsigned)c.i_f); mc4[j][2] = ld32(mumap, moff + (unsigned)c.il); mc4[j][3] = ld32(mumap, moff + (unsigned)c.ifl);
      }
    }
    // Issue order of a step's memory operations (vmcnt retires loads, stores and atomics in order, so a wait for a load is
    // a wait for everything issued before it): gathers of this step | map-gradient atomics and control-gradient store
    // deferred from the PREVIOUS step | prefetch of the next step's rows.  Nothing is younger than the prefetch, so the
    // wait for it at the top of the next iteration is not also a wait for a store issued a few instructions earlier.
    flush_stash();
    gctrl_pending[0] = gv_pending; gctrl_pending[1] = gwc_pending;   // every lane of the group, same values
    StateIn nxt;
    step_back(rp, n > 0 ? 1 : 0);
    load_state(rp, max(n - 1, 0), nxt);      // prefetch (step 0 harmlessly reloads itself); younger than the gathers above
    UpIn up_next;
    load_upstream(rp, up_next);
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      const Cell<S>& c = cell[j];
      S zq = blend(c, zc4[j][0], zc4[j][1], zc4[j][2], zc4[j][3]);
      muq[j] = has_mu ? blend(c, mc4[j][0], mc4[j][1], mc4[j][2], mc4[j][3]) : blend_ones(c);
      S gx = M::div(zc4[j][1] - zc4[j][0], a.res), gy = M::div(zc4[j][2] - zc4[j][0], a.res);
      nl[j] = mf_max(M::sqrt(gx * gx + gy * gy + one), (S)1e-6);
      nrm[j][0] = M::div(-gx, nl[j]); nrm[j][1] = M::div(-gy, nl[j]); nrm[j][2] = M::div(one, nl[j]);
      S dh = pzv[j] - zq;
      S cc = M::sigmoid_m10(dh);
      cj[j] = act[j] ? cc : zero;
      csum += cj[j];
      S vn = vp[j][0] * nrm[j][0] + vp[j][1] * nrm[j][1] + vp[j][2] * nrm[j][2];
      Aj[j] = a.k * dh + a.damp * vn;
      F0[j][0] = -(Aj[j] * nrm[j][0]); F0[j][1] = -(Aj[j] * nrm[j][1]); F0[j][2] = -(Aj[j] * nrm[j][2]);
    }
    csum = gs.sum(csum);

    const S coln = M::sqrt(R[0] * R[0] + R[3] * R[3] + R[6] * R[6]);
    const S el = mf_max(coln, (S)1e-6);
    const S e[3] = {M::div(R[0], el), M::div(R[3], el), M::div(R[6], el)};
    const S inv_csum = M::div(one, csum);
    const S tv_lo = cv - cw * a.half_ly, tv_hi = cv + cw * a.half_ly;

    S F1[PPL][3], Fr[PPL][3], Ff[PPL][3], Gf[PPL][3], st[PPL][3], slip[PPL][3], sn[PPL], Nn[PPL], tv[PPL];
    S sTau[3] = {zero, zero, zero};
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        F1[j][c] = FAST ? F0[j][c] * cj[j] * inv_csum : F0[j][c] * cj[j] / csum;
        Fr[j][c] = mf_clamp(F1[j][c], -a.mg, a.mg);
      }
      Nn[j] = M::sqrt(Fr[j][0] * Fr[j][0] + Fr[j][1] * Fr[j][1] + Fr[j][2] * Fr[j][2]);
      tv[j] = (part[j] < 0) ? zero : ((part[j] & 1) ? tv_hi : tv_lo);
#pragma unroll
      for (int c = 0; c < 3; ++c) slip[j][c] = muq[j] * (tv[j] * e[c] - vp[j][c]);
      sn[j] = slip[j][0] * nrm[j][0] + slip[j][1] * nrm[j][1] + slip[j][2] * nrm[j][2];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        st[j][c] = slip[j][c] - sn[j] * nrm[j][c];
        Gf[j][c] = Nn[j] * st[j][c];
        Ff[j][c] = mf_clamp(Gf[j][c], -a.mg, a.mg);
      }
      if (!act[j]) {
#pragma unroll
        for (int c = 0; c < 3; ++c) Fr[j][c] = Ff[j][c] = zero;
      }
      S f[3] = {Fr[j][0] + Ff[j][0], Fr[j][1] + Ff[j][1], Fr[j][2] + Ff[j][2]};
      sTau[0] += r[j][1] * f[2] - r[j][2] * f[1];
      sTau[1] += r[j][2] * f[0] - r[j][0] * f[2];
      sTau[2] += r[j][0] * f[1] - r[j][1] * f[0];
    }
    gs.sum_n(sTau);
    S wraw[3], wd[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      wraw[c] = Iv[c * 3 + 0] * sTau[0] + Iv[c * 3 + 1] * sTau[1] + Iv[c * 3 + 2] * sTau[2];
      wd[c] = mf_clamp(wraw[c], -a.omega_max, a.omega_max);
    }

    // ---------------------------------------------------------------------------------------------------
    // integrator backward: adjoint of the step's outputs -> (g_xdd, g_wd, g_Fs_i, g_Ff_i) + adjoint of its inputs
    // ---------------------------------------------------------------------------------------------------
    S gxdd[3], gwd[3], gFr[PPL][3], gFf[PPL][3];
    if (INTEG == MF_INTEG_ODEINT_EULER) {
      const S h = cur.t1 - cur.t0;
#pragma unroll
      for (int j = 0; j < PPL; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) { if constexpr (!XS_ONLY) { laFs[j][c] += act[j] ? up.gFs[j][c] : zero; laFf[j][c] += act[j] ? up.gFf[j][c] : zero; } }
      if constexpr (UNSUM) {
        S tot6[6] = {lxd[0], lxd[1], lxd[2], lw[0], lw[1], lw[2]};      // the velocity adjoints proper
        gs.sum_n(tot6);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          gxdd[c] = h * tot6[c];
          gwd[c] = h * tot6[3 + c];
          lxd[c] += h * lx[c];
        }
      } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gxdd[c] = h * lxd[c];
        gwd[c] = h * lw[c];
        lxd[c] += h * lx[c];               // x' = x + h xd
      }
      }
#pragma unroll
      for (int j = 0; j < PPL; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) { gFr[j][c] = XS_ONLY ? zero : h * laFs[j][c]; gFf[j][c] = XS_ONLY ? zero : h * laFf[j][c]; }
      // R' = R + h [w]x R : column-wise dR_c = w x R_c
      S lRn[9];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        S gcol[3] = {h * lR[0 * 3 + c], h * lR[1 * 3 + c], h * lR[2 * 3 + c]};
        S rc[3] = {R[0 * 3 + c], R[1 * 3 + c], R[2 * 3 + c]};
        S t1[3], t2[3];
        MF_CROSS(t1, rc, gcol);            // d/dw of (w x R_c) . g  =  R_c x g
        lw[0] += t1[0]; lw[1] += t1[1]; lw[2] += t1[2];
        MF_CROSS(t2, gcol, w);             // d/dR_c                  =  g x w
        lRn[0 * 3 + c] = lR[0 * 3 + c] + t2[0];
        lRn[1 * 3 + c] = lR[1 * 3 + c] + t2[1];
        lRn[2 * 3 + c] = lR[2 * 3 + c] + t2[2];
      }
#pragma unroll
      for (int c = 0; c < 9; ++c) lR[c] = lRn[c];
    } else {
      const S h = a.dt;
      // forces of this step are outputs themselves
#pragma unroll
      for (int j = 0; j < PPL; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) { gFr[j][c] = (!XS_ONLY && act[j]) ? up.gFs[j][c] : zero; gFf[j][c] = (!XS_ONLY && act[j]) ? up.gFf[j][c] : zero; }
      // R' = R M(w'),  w' = w + wd h,  M = I + K sin(th h) + K^2 (1 - cos(th h)),  K = [w']x / max(|w'|, eps)
      S wn[3] = {w[0] + wd[0] * h, w[1] + wd[1] * h, w[2] + wd[2] * h};
      S th = M::sqrt(wn[0] * wn[0] + wn[1] * wn[1] + wn[2] * wn[2]);
      S den = mf_max(th, (S)1e-6);
      S kv[3] = {M::div(wn[0], den), M::div(wn[1], den), M::div(wn[2], den)};
      S sn_, oc;
      M::sincos_small(th * h, &sn_, &oc);
      S cs_ = one - oc;
      S K[9] = {zero, -kv[2], kv[1], kv[2], zero, -kv[0], -kv[1], kv[0], zero};
      S K2[9], M[9];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j2 = 0; j2 < 3; ++j2) {
          K2[i * 3 + j2] = K[i * 3 + 0] * K[0 * 3 + j2] + K[i * 3 + 1] * K[1 * 3 + j2] + K[i * 3 + 2] * K[2 * 3 + j2];
          M[i * 3 + j2] = ((i == j2 ? one : zero) + K[i * 3 + j2] * sn_) + K2[i * 3 + j2] * oc;
        }
      S gM[9], lRn[9];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j2 = 0; j2 < 3; ++j2) {
          gM[i * 3 + j2] = R[0 * 3 + i] * lR[0 * 3 + j2] + R[1 * 3 + i] * lR[1 * 3 + j2] + R[2 * 3 + i] * lR[2 * 3 + j2];   // R^T lR
          lRn[i * 3 + j2] = lR[i * 3 + 0] * M[j2 * 3 + 0] + lR[i * 3 + 1] * M[j2 * 3 + 1] + lR[i * 3 + 2] * M[j2 * 3 + 2];  // lR M^T
        }
      S ga = zero, gb = zero, gK[9];
#pragma unroll
      for (int c = 0; c < 9; ++c) { ga += gM[c] * K[c]; gb += gM[c] * K2[c]; }
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j2 = 0; j2 < 3; ++j2) {
          // d(K K) -> gM K^T + K^T gM
          S t = zero;
#pragma unroll
          for (int m = 0; m < 3; ++m) t += gM[i * 3 + m] * K[j2 * 3 + m] + K[m * 3 + i] * gM[m * 3 + j2];
          gK[i * 3 + j2] = sn_ * gM[i * 3 + j2] + oc * t;
        }
      S gk[3] = {gK[7] - gK[5], gK[2] - gK[6], gK[3] - gK[1]};
      S gth = ga * h * cs_ + gb * h * sn_;
      S gwn[3] = {M::div(gk[0], den), M::div(gk[1], den), M::div(gk[2], den)};
      if (th >= (S)1e-6) gth += M::div(-(gk[0] * wn[0] + gk[1] * wn[1] + gk[2] * wn[2]), den * den);
      if (th > zero) {
#pragma unroll
        for (int c = 0; c < 3; ++c) gwn[c] += M::div(gth * wn[c], th);
      }
      if constexpr (UNSUM) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          lw[c] += gwn[c];
          lxd[c] += h * lx[c];
        }
        S tot6[6] = {lxd[0], lxd[1], lxd[2], lw[0], lw[1], lw[2]};
        gs.sum_n(tot6);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          gwd[c] = h * tot6[3 + c];
          gxdd[c] = h * tot6[c];
        }
      } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        lw[c] += gwn[c];
        gwd[c] = h * lw[c];                // w' = w + wd h
        lxd[c] += h * lx[c];               // x' = x + xd' h
        gxdd[c] = h * lxd[c];              // xd' = xd + xdd h
      }
      }
#pragma unroll
      for (int c = 0; c < 9; ++c) lR[c] = lRn[c];
    }

    // ---------------------------------------------------------------------------------------------------
    // RHS backward
    // ---------------------------------------------------------------------------------------------------
    S gtau[3];
    {
      S m0 = inside(wraw[0], -a.omega_max, a.omega_max) ? gwd[0] : zero;
      S m1 = inside(wraw[1], -a.omega_max, a.omega_max) ? gwd[1] : zero;
      S m2 = inside(wraw[2], -a.omega_max, a.omega_max) ? gwd[2] : zero;
#pragma unroll
      for (int c = 0; c < 3; ++c) gtau[c] = Iv[0 * 3 + c] * m0 + Iv[1 * 3 + c] * m1 + Iv[2 * 3 + c] * m2;   // Iinv^T
    }
    const S gsum[3] = {FAST ? gxdd[0] * a.inv_mass : gxdd[0] / a.mass, FAST ? gxdd[1] * a.inv_mass : gxdd[1] / a.mass,
                       FAST ? gxdd[2] * a.inv_mass : gxdd[2] / a.mass};

    S ge[3] = {zero, zero, zero}, gv = zero, gwc = zero, gS = zero;
    S gdh_p[PPL], gc_p[PPL], gn[PPL][3], gvp[PPL][3], gmuq[PPL], gr[PPL][3];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      S f[3] = {Fr[j][0] + Ff[j][0], Fr[j][1] + Ff[j][1], Fr[j][2] + Ff[j][2]};
      S gf[3];
      MF_CROSS(gf, gtau, r[j]);            // tau += r x f : df = gtau x r
      MF_CROSS(gr[j], f, gtau);            //                dr = f x gtau
      S gFr_[3], gG[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gFr_[c] = XS_ONLY ? gsum[c] + gf[c] : gFr[j][c] + gsum[c] + gf[c];
        S gFf_ = XS_ONLY ? gsum[c] + gf[c] : gFf[j][c] + gsum[c] + gf[c];
        gG[c] = inside(Gf[j][c], -a.mg, a.mg) ? gFf_ : zero;
      }
      S gNn = gG[0] * st[j][0] + gG[1] * st[j][1] + gG[2] * st[j][2];
      S gst[3] = {Nn[j] * gG[0], Nn[j] * gG[1], Nn[j] * gG[2]};
      S gsn = -(gst[0] * nrm[j][0] + gst[1] * nrm[j][1] + gst[2] * nrm[j][2]);
      S gslip[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gn[j][c] = -sn[j] * gst[c] + gsn * slip[j][c];
        gslip[c] = gst[c] + gsn * nrm[j][c];
      }
      S cmdv[3] = {tv[j] * e[0] - vp[j][0], tv[j] * e[1] - vp[j][1], tv[j] * e[2] - vp[j][2]};
      gmuq[j] = gslip[0] * cmdv[0] + gslip[1] * cmdv[1] + gslip[2] * cmdv[2];
      S gcmd[3] = {muq[j] * gslip[0], muq[j] * gslip[1], muq[j] * gslip[2]};
#pragma unroll
      for (int c = 0; c < 3; ++c) gvp[j][c] = -gcmd[c];
      if (part[j] >= 0 && act[j]) {
        S gtv = gcmd[0] * e[0] + gcmd[1] * e[1] + gcmd[2] * e[2];
#pragma unroll
        for (int c = 0; c < 3; ++c) ge[c] += tv[j] * gcmd[c];
        gv += gtv;
        gwc += ((part[j] & 1) ? a.half_ly : -a.half_ly) * gtv;
      }
      if (Nn[j] > zero) {
#pragma unroll
        for (int c = 0; c < 3; ++c) gFr_[c] += M::div(gNn * Fr[j][c], Nn[j]);
      }
      S gF1[3], d = zero, gA = zero;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gF1[c] = inside(F1[j][c], -a.mg, a.mg) ? gFr_[c] : zero;
        d += gF1[c] * F0[j][c];
      }
      gc_p[j] = FAST ? d * inv_csum : d / csum;
      gS += FAST ? -(d * cj[j]) * inv_csum * inv_csum : -(d * cj[j]) / (csum * csum);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        S gF0 = FAST ? gF1[c] * cj[j] * inv_csum : gF1[c] * cj[j] / csum;
        gA += -(gF0 * nrm[j][c]);
        gn[j][c] += -Aj[j] * gF0;
      }
      gdh_p[j] = a.k * gA;
      S gvn = a.damp * gA;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gvp[j][c] += gvn * nrm[j][c];
        gn[j][c] += gvn * vp[j][c];
      }
    }
    gS = gs.sum(gS);

    S gx_[3] = {zero, zero, zero}, gxd_[3] = {zero, zero, zero}, gw_[3] = {zero, zero, zero}, gR_[9];
    S gja_[4] = {zero, zero, zero, zero};
#pragma unroll
    for (int c = 0; c < 9; ++c) gR_[c] = zero;
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      {   // inactive slots carry cj = 0 and zero forces, so all of their contributions below are exactly 0
        const Cell<S>& c = cell[j];
        S gc = gc_p[j] + gS;
        S gdh = gdh_p[j] + gc * ((S)-10) * cj[j] * (one - cj[j]);
        S gzq = -gdh;
        // n = u / |u|, u = (-gx, -gy, 1)
        S dotn = gn[j][0] * nrm[j][0] + gn[j][1] * nrm[j][1] + gn[j][2] * nrm[j][2];
        S gu0 = M::div(gn[j][0] - dotn * nrm[j][0], nl[j]), gu1 = M::div(gn[j][1] - dotn * nrm[j][1], nl[j]);
        S ggx = M::div(-gu0, a.res), ggy = M::div(-gu1, a.res);
        const S w00 = (one - c.fx) * (one - c.fy), w01 = (one - c.fx) * c.fy, w10 = c.fx * (one - c.fy), w11 = c.fx * c.fy;
        {
          const unsigned ni[4] = {(unsigned)c.ic, (unsigned)c.i_f, (unsigned)c.il, (unsigned)c.ifl};
          const S nz[4] = {gzq * w00 - ggx - ggy, gzq * w01 + ggx, gzq * w10 + ggy, gzq * w11};
          const S nm[4] = {gmuq[j] * w00, gmuq[j] * w01, gmuq[j] * w10, gmuq[j] * w11};
          if (carry_over) {   // compile-time
            // footprint slots: 0 = (ix, iy), 1 = (ix+1, iy), 2 = (ix, iy+1), 3 = (ix+1, iy+1).  Which old slots do the new ones
            // coincide with?  (Index equality, so it also holds where the flat-index clamp folds cells onto each other.)
            // (non-short-circuit & and | throughout: everything below must stay straight-line selects, no branches)
            const unsigned o0 = acc_idx[j][0], o1 = acc_idx[j][1], o2 = acc_idx[j][2], o3 = acc_idx[j][3];
            const bool same = !act[j] | ((ni[0] == o0) & (ni[1] == o1) & (ni[2] == o2) & (ni[3] == o3));
            const bool cxp = (ni[0] == o1) & (ni[2] == o3), cxm = (ni[1] == o0) & (ni[3] == o2);
            const bool cyp = (ni[0] == o2) & (ni[1] == o3), cym = (ni[2] == o0) & (ni[3] == o1);
            const bool xp = !same & cxp;                          // moved one cell in +x: new 0,2 = old 1,3
            const bool xm = !same & !cxp & cxm;                   // -x: new 1,3 = old 0,2
            const bool yp = !same & !cxp & !cxm & cyp;            // +y: new 0,1 = old 2,3
            const bool ym = !same & !cxp & !cxm & !cyp & cym;     // -y: new 2,3 = old 0,1
            const bool jump = !same & !cxp & !cxm & !cyp & !cym;
            st_pending[j] = !same;                 // the stash was flushed at the top of this iteration, so it is free
            st_all[j] = jump;
            // the cells left behind go to stash slots 0, 1 (a jump leaves all four: slots 2, 3 as well):
            //   slot 0 <- old 1 (xm), old 2 (ym), else old 0 (xp, yp, jump);  slot 1 <- old 2 (xp), old 3 (xm, ym), else old 1 (yp, jump)
            const bool e13 = xm | ym;
            st_idx[j][0] = xm ? o1 : (ym ? o2 : o0);
            st_z[j][0] = xm ? acc_z[j][1] : (ym ? acc_z[j][2] : acc_z[j][0]);
            st_m[j][0] = xm ? acc_m[j][1] : (ym ? acc_m[j][2] : acc_m[j][0]);
            st_idx[j][1] = xp ? o2 : (e13 ? o3 : o1);
            st_z[j][1] = xp ? acc_z[j][2] : (e13 ? acc_z[j][3] : acc_z[j][1]);
            st_m[j][1] = xp ? acc_m[j][2] : (e13 ? acc_m[j][3] : acc_m[j][1]);
            st_idx[j][2] = o2; st_z[j][2] = acc_z[j][2]; st_m[j][2] = acc_m[j][2];
            st_idx[j][3] = o3; st_z[j][3] = acc_z[j][3]; st_m[j][3] = acc_m[j][3];
            // carried-over accumulators for the new slots
            const S cz[4] = {same ? acc_z[j][0] : xp ? acc_z[j][1] : yp ? acc_z[j][2] : zero,
                             same ? acc_z[j][1] : xm ? acc_z[j][0] : yp ? acc_z[j][3] : zero,
                             same ? acc_z[j][2] : xp ? acc_z[j][3] : ym ? acc_z[j][0] : zero,
                             same ? acc_z[j][3] : xm ? acc_z[j][2] : ym ? acc_z[j][1] : zero};
            const S cm[4] = {same ? acc_m[j][0] : xp ? acc_m[j][1] : yp ? acc_m[j][2] : zero,
                             same ? acc_m[j][1] : xm ? acc_m[j][0] : yp ? acc_m[j][3] : zero,
                             same ? acc_m[j][2] : xp ? acc_m[j][3] : ym ? acc_m[j][0] : zero,
                             same ? acc_m[j][3] : xm ? acc_m[j][2] : ym ? acc_m[j][1] : zero};
  #pragma unroll
            for (int q = 0; q < 4; ++q) {
              acc_idx[j][q] = act[j] ? ni[q] : acc_idx[j][q];
              acc_z[j][q] = cz[q] + nz[q];
              acc_m[j][q] = cm[q] + nm[q];
            }
          } else {            // plain form: any change of cell writes all four accumulators (fewest instructions per step)
            const bool same = !act[j] | ((ni[0] == acc_idx[j][0]) & (ni[1] == acc_idx[j][1]) & (ni[2] == acc_idx[j][2]) & (ni[3] == acc_idx[j][3]));
            st_pending[j] = !same;
            st_all[j] = !same;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              st_idx[j][q] = acc_idx[j][q]; st_z[j][q] = acc_z[j][q]; st_m[j][q] = acc_m[j][q];
              acc_idx[j][q] = same ? acc_idx[j][q] : ni[q];
              acc_z[j][q] = same ? acc_z[j][q] + nz[q] : nz[q];
              acc_m[j][q] = same ? acc_m[j][q] + nm[q] : nm[q];
            }
          }
        }
        S zfx, zfy, mfx, mfy;
        blend_grad(c, zc4[j][0], zc4[j][1], zc4[j][2], zc4[j][3], &zfx, &zfy);
        {
          S ofx, ofy;                     // a map of ones without a friction map (its blend still moves with fx, fy by rounding)
          blend_grad(c, mc4[j][0], mc4[j][1], mc4[j][2], mc4[j][3], &mfx, &mfy);
          blend_grad(c, one, one, one, one, &ofx, &ofy);
          mfx = has_mu ? mfx : ofx; mfy = has_mu ? mfy : ofy;
        }
        S gp[3] = {M::div(gzq * zfx + gmuq[j] * mfx, a.res), M::div(gzq * zfy + gmuq[j] * mfy, a.res), gdh};
        // v_p = xd + w x r
        S t1[3], t2[3];
        S gP[3] = {zero, zero, zero};      // adjoint of the body-frame point (articulated bodies: feeds the joint angles)
        MF_CROSS(t1, gvp[j], w);           // dr += gvp x w
        MF_CROSS(t2, r[j], gvp[j]);        // dw += r x gvp
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          gr[j][q] += t1[q];
          gw_[q] += t2[q];
          gxd_[q] += gvp[j][q];
          gx_[q] += gp[q];
          S qa = gp[q] + gr[j][q];         // p = R P + x,  r = p - x
          gR_[q * 3 + 0] += qa * P[j][0];
          gR_[q * 3 + 1] += qa * P[j][1];
          gR_[q * 3 + 2] += qa * P[j][2];
          if constexpr (JOINTS) { gP[0] += qa * R[q * 3 + 0]; gP[1] += qa * R[q * 3 + 1]; gP[2] += qa * R[q * 3 + 2]; }   // R^T qa
        }
        if constexpr (JOINTS) {
          // ... and through the inertia of the articulated body (dphysics.py:196-197): wd = I^-1 tau, I = sum_j m (|P_j|^2 E -
          // P_j P_j^T).  With a = I^-T g_wd (= gtau) and b = I^-1 tau (= wraw) the gradient of I is -a b^T, and
          // dL/dP_j = m (2 tr(gI) P_j - (gI + gI^T) P_j) = m (-2 (a.b) P_j + a (b.P_j) + b (a.P_j))
          const S mp = act[j] ? a.mass / (S)a.N : zero;
          const S ab = gtau[0] * wraw[0] + gtau[1] * wraw[1] + gtau[2] * wraw[2];
          const S bP = wraw[0] * P[j][0] + wraw[1] * P[j][1] + wraw[2] * P[j][2];
          const S aP = gtau[0] * P[j][0] + gtau[1] * P[j][1] + gtau[2] * P[j][2];
#pragma unroll
          for (int c = 0; c < 3; ++c) gP[c] += mp * (((S)-2 * ab) * P[j][c] + gtau[c] * bP + wraw[c] * aP);
          // P_j = J + Ry(theta) (P0_j - J) for the points of flipper q: dP/dtheta = (P_z - J_z, 0, -(P_x - J_x))
          const int qj = max(part[j], 0);
          const S dth = gP[0] * (P[j][2] - a.joint_xyz[qj * 3 + 2]) - gP[2] * (P[j][0] - a.joint_xyz[qj * 3 + 0]);
#pragma unroll
          for (int q = 0; q < 4; ++q) gja_[q] += (part[j] == q) ? dth : zero;
        }
      }
    }
    if constexpr (UNSUM) {      // the lane's own parts, no sums -- but for the control gradient this step stores
#pragma unroll
      for (int c = 0; c < 3; ++c) { lx[c] += gx_[c]; lxd[c] += gxd_[c]; lw[c] += gw_[c]; }
#pragma unroll
      for (int c = 0; c < 9; ++c) lR[c] += gR_[c];
      S red2[2] = {gv, gwc};
      gs.sum_n(red2);
      gv = red2[0]; gwc = red2[1];
    } else {
    S red[JOINTS ? 27 : 23];
#pragma unroll
      for (int c = 0; c < 3; ++c) { red[c] = gx_[c]; red[3 + c] = gxd_[c]; red[6 + c] = gw_[c]; red[9 + c] = ge[c]; }
#pragma unroll
      for (int c = 0; c < 9; ++c) red[12 + c] = gR_[c];
      red[21] = gv; red[22] = gwc;
      if constexpr (JOINTS) { red[23] = gja_[0]; red[24] = gja_[1]; red[25] = gja_[2]; red[26] = gja_[3]; }
      gs.sum_n(red);       // the step's 23 adjoint sums in one batched reduction (multi-wave groups: one LDS exchange)
      if constexpr (JOINTS) {
        if (a.gjoint != nullptr && gl == 0) {
          S* o = a.gjoint + ((size_t)b * a.T + n) * 4;
          o[0] = red[23]; o[1] = red[24]; o[2] = red[25]; o[3] = red[26];
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) { lx[c] += red[c]; lxd[c] += red[3 + c]; lw[c] += red[6 + c]; ge[c] = red[9 + c]; }
#pragma unroll
      for (int c = 0; c < 9; ++c) lR[c] += red[12 + c];
      gv = red[21]; gwc = red[22];
    }
    {   // e = col0(R) / max(|col0|, eps): through |col0| only when it is >= eps
      const S dote = (coln >= (S)1e-6) ? ge[0] * e[0] + ge[1] * e[1] + ge[2] * e[2] : zero;
      lR[0] += M::div(ge[0] - dote * e[0], el);
      lR[3] += M::div(ge[1] - dote * e[1], el);
      lR[6] += M::div(ge[2] - dote * e[2], el);
    }
    gctrl_pending = gctrl + n * a.gc_st; gv_pending = gv; gwc_pending = gwc;   // stored by the next iteration (or after the loop)
    cur = nxt;
    up = up_next;
  }
  flush_stash();
  gctrl_pending[0] = gv_pending; gctrl_pending[1] = gwc_pending;
#pragma unroll
  for (int j = 0; j < PPL; ++j) {          // what is still accumulated in registers
    if (act[j]) {
#pragma unroll
      for (int q = 0; q < 4; ++q) emit(acc_idx[j][q], acc_z[j][q], acc_m[j][q]);
    }
  }

  if (INTEG == MF_INTEG_ODEINT_EULER) {   // output 0 is the initial state itself (its forces are constant zeros)
    make_ptrs(-1, rp);              // ODEINT: the output row of "step -1" is row 0
    load_upstream(rp, up);
    add_upstream_state(up);
  }

  if constexpr (UNSUM) {      // the adjoint of the initial state proper
    S tot[18];
#pragma unroll
    for (int c = 0; c < 3; ++c) { tot[c] = lx[c]; tot[3 + c] = lxd[c]; tot[6 + c] = lw[c]; }
#pragma unroll
    for (int c = 0; c < 9; ++c) tot[9 + c] = lR[c];
    gs.sum_n(tot);
#pragma unroll
    for (int c = 0; c < 3; ++c) { lx[c] = tot[c]; lxd[c] = tot[3 + c]; lw[c] = tot[6 + c]; }
#pragma unroll
    for (int c = 0; c < 9; ++c) lR[c] = tot[9 + c];
  }
  // terrain snap of the initial height: x.z = mean_i blend(z; cell((R0 P_i + x0).xy))   (dphysics.py:567-571)
  S gx0[3] = {lx[0], lx[1], lx[2]};
  if (!a.skip_snap) {
    S R0[9], x0[2];
#pragma unroll
    for (int c = 0; c < 9; ++c) R0[c] = a.R0[b * 9 + c];
    x0[0] = a.x_init[b * 3 + 0]; x0[1] = a.x_init[b * 3 + 1];
    const S g = lx[2] / (S)a.N;
    S sx = zero, sy = zero, sR[6] = {zero, zero, zero, zero, zero, zero};
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      if (act[j]) {
        S px = P0[j][0] * R0[0] + P0[j][1] * R0[1] + P0[j][2] * R0[2] + x0[0];
        S py = P0[j][0] * R0[3] + P0[j][1] * R0[4] + P0[j][2] * R0[5] + x0[1];
        Cell<S> c = locate_m<S, FAST>(px, py, a.d_max, a.res, a.inv_res, a.H, last);
        S v0 = ld32(zmap, moff + (unsigned)c.ic), v1 = ld32(zmap, moff + (unsigned)c.i_f), v2 = ld32(zmap, moff + (unsigned)c.il), v3 = ld32(zmap, moff + (unsigned)c.ifl);
        atomic_add(at32(gzmap, goff + (unsigned)c.ic), g * (one - c.fx) * (one - c.fy));
        atomic_add(at32(gzmap, goff + (unsigned)c.i_f), g * (one - c.fx) * c.fy);
        atomic_add(at32(gzmap, goff + (unsigned)c.il), g * c.fx * (one - c.fy));
        atomic_add(at32(gzmap, goff + (unsigned)c.ifl), g * c.fx * c.fy);
        S dfx, dfy;
        blend_grad(c, v0, v1, v2, v3, &dfx, &dfy);
        S gpx = M::div(g * dfx, a.res), gpy = M::div(g * dfy, a.res);
        sx += gpx; sy += gpy;
#pragma unroll
        for (int q = 0; q < 3; ++q) { sR[q] += gpx * P0[j][q]; sR[3 + q] += gpy * P0[j][q]; }
      }
    }
    gx0[0] += gs.sum(sx);
    gx0[1] += gs.sum(sy);
    gx0[2] = zero;                          // the caller's x0.z is overwritten, so nothing flows to it
#pragma unroll
    for (int q = 0; q < 6; ++q) lR[q] += gs.sum(sR[q]);
  }
  if constexpr (LOSS) { if (l_acc) *l_acc = l_val; }
  if (gl == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (a.gx0) a.gx0[b * 3 + c] = gx0[c];
      a.gxd0[b * 3 + c] = lxd[c];
      a.gw0[b * 3 + c] = lw[c];
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) a.gR0[b * 9 + c] = lR[c];
  }
}

template <typename S, int G, int PPL, int INTEG, bool FAST, bool JOINTS = false, bool CARRY = true, bool XS_ONLY = false, bool ZMU = false, bool WIN = false, bool LOSS = false>
__global__ void __launch_bounds__(WIN ? 512 : (G > 256 ? G : 256)) rollout_bwd_kernel(const RolloutBwdArgs<S> a) {
  if constexpr (WIN) {
    static_assert(G <= 64 && sizeof(S) == 4, "the LDS gradient window serves float32 rollouts inside a wave");
    __shared__ S win[2 * kWinW * kWinW];
    int wx0, wy0;
    win_open<S, FAST>(a, win, (int)((blockIdx.x * blockDim.x) / G), &wx0, &wy0);
    __syncthreads();
    const unsigned sh = 31u - (unsigned)__builtin_clz((unsigned)a.H);      // H = 2^sh (host-checked)
    S l_acc = (S)0;
    rollout_bwd_body<S, G, PPL, INTEG, FAST, JOINTS, CARRY, XS_ONLY, ZMU, true, LOSS>(a, win, (unsigned)wy0 + ((unsigned)wx0 << sh), sh, &l_acc);
    __syncthreads();
    win_close(a, win, wx0, wy0);
    if constexpr (LOSS) xs_loss_value_finish(a, l_acc);
  } else {
    S l_acc = (S)0;
    rollout_bwd_body<S, G, PPL, INTEG, FAST, JOINTS, CARRY, XS_ONLY, ZMU, false, LOSS>(a, nullptr, 0, 0, &l_acc);
    if constexpr (LOSS) {
      static_assert(G <= 64, "the fused loss serves rollouts inside a wave");
      xs_loss_value_finish(a, l_acc);
    }
  }
}

template <typename S, bool FAST, bool JOINTS = false, bool CARRY = true>
int launch_rollout_bwd(const RolloutBwdArgs<S>& a, LaneMap m, int integ, int block, hipStream_t st) {
  if (m.G > 64) block = m.G;   // a rollout spread over several waves: exactly one rollout per workgroup (LDS + barrier)
  const long long threads = (long long)a.B * m.G;
  const unsigned grid = (unsigned)((threads + block - 1) / block);
  bool launched = false;
#define MF_CASE(G_, P_)                                                                                                           \
  if (!launched && m.G == G_ && m.PPL == P_) {                                                                                     \
    launched = true;                                                                                                               \
    if (integ == MF_INTEG_DYNAMICS)                                                                                                \
      MF_KLAUNCH((rollout_bwd_kernel<S, G_, P_, MF_INTEG_DYNAMICS, FAST, JOINTS, CARRY>), dim3(grid), dim3(block), 0, st, a);      \
    else                                                                                                                           \
      MF_KLAUNCH((rollout_bwd_kernel<S, G_, P_, MF_INTEG_ODEINT_EULER, FAST, JOINTS, CARRY>), dim3(grid), dim3(block), 0, st, a);  \
  }
  MF_CASE(4, 1) MF_CASE(8, 1) MF_CASE(16, 1) MF_CASE(32, 1) MF_CASE(64, 1) MF_CASE(64, 2) MF_CASE(64, 4) MF_CASE(64, 8)
  MF_CASE(128, 1) MF_CASE(256, 1) MF_CASE(512, 1)
  if constexpr (!JOINTS) {   // the 4-points-per-lane mappings are a tuning / test option of the rigid-body kernels
    MF_CASE(1, 4) MF_CASE(2, 4) MF_CASE(4, 4) MF_CASE(8, 4) MF_CASE(16, 4) MF_CASE(32, 4)
  }
#undef MF_CASE
  MF_REQUIRE(launched, MF_ERR_UNSUPPORTED, "rollout_bwd: no kernel for this lane mapping");
  hipError_t e = hipGetLastError();
  MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("rollout_bwd launch: ") + hipGetErrorString(e));
  return MF_OK;
}

// the positions-only (XS_ONLY) instantiations with accumulator carry-over, one point per lane inside a wave (G = 4 .. 64), plain or
// interleaved maps: the saturated launches of small bodies (rollout_bwd_xs_fast.hip)
template <typename S, bool ZMU, bool WIN = false, bool CARRY = true, bool LOSS = false>
int launch_rollout_bwd_xs(const RolloutBwdArgs<S>& a, LaneMap m, int integ, int block, hipStream_t st) {
  const long long threads = (long long)a.B * m.G;
  const unsigned grid = (unsigned)((threads + block - 1) / block);
  bool launched = false;
#define MF_CASE(G_)                                                                                                                              \
  if (!launched && m.G == G_ && m.PPL == 1) {                                                                                                     \
    launched = true;                                                                                                                              \
    if (integ == MF_INTEG_DYNAMICS)                                                                                                               \
      MF_KLAUNCH((rollout_bwd_kernel<S, G_, 1, MF_INTEG_DYNAMICS, true, false, CARRY, true, ZMU, WIN, LOSS>), dim3(grid), dim3(block), 0, st, a);       \
    else                                                                                                                                          \
      MF_KLAUNCH((rollout_bwd_kernel<S, G_, 1, MF_INTEG_ODEINT_EULER, true, false, CARRY, true, ZMU, WIN, LOSS>), dim3(grid), dim3(block), 0, st, a);   \
  }
  MF_CASE(4)
  if constexpr (!WIN) { MF_CASE(8) MF_CASE(16) MF_CASE(32) MF_CASE(64) }
#undef MF_CASE
  MF_REQUIRE(launched, MF_ERR_UNSUPPORTED, "rollout_bwd: no positions-only kernel for this lane mapping");
  hipError_t e = hipGetLastError();
  MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("rollout_bwd (positions only) launch: ") + hipGetErrorString(e));
  return MF_OK;
}
// ... and for one rollout per wave with 2 / 4 / 8 points per lane (bodies of 65 .. 512 points beyond the record-reading multi-wave range):
// a positions-only upstream compiles out 6 PPL row loads and the impulse adjoints per lane and step (rollout_bwd_xs_ppl_fast.hip; round 6)
template <typename S>
int launch_rollout_bwd_xs_ppl(const RolloutBwdArgs<S>& a, LaneMap m, int integ, int block, hipStream_t st) {
  const long long threads = (long long)a.B * m.G;
  const unsigned grid = (unsigned)((threads + block - 1) / block);
  bool launched = false;
#define MF_CASE(P_)                                                                                                                              \
  if (!launched && m.G == 64 && m.PPL == P_) {                                                                                                    \
    launched = true;                                                                                                                              \
    if (integ == MF_INTEG_DYNAMICS)                                                                                                               \
      MF_KLAUNCH((rollout_bwd_kernel<S, 64, P_, MF_INTEG_DYNAMICS, true, false, true, true, false, false>), dim3(grid), dim3(block), 0, st, a);       \
    else                                                                                                                                          \
      MF_KLAUNCH((rollout_bwd_kernel<S, 64, P_, MF_INTEG_ODEINT_EULER, true, false, true, true, false, false>), dim3(grid), dim3(block), 0, st, a);   \
  }
  MF_CASE(2) MF_CASE(4) MF_CASE(8)
#undef MF_CASE
  MF_REQUIRE(launched, MF_ERR_UNSUPPORTED, "rollout_bwd: no positions-only kernel for this lane mapping");
  hipError_t e = hipGetLastError();
  MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("rollout_bwd (positions only, several points per lane) launch: ") + hipGetErrorString(e));
  return MF_OK;
}
int launch_rollout_bwd_xs_ppl_fast_f32(const RolloutBwdArgs<float>& a, LaneMap m, int integ, int block, hipStream_t st);      // rollout_bwd_xs_ppl_fast.hip
int launch_rollout_bwd_xs_fast_f32(const RolloutBwdArgs<float>& a, LaneMap m, int integ, int block, bool zmu, hipStream_t st);      // rollout_bwd_xs_fast.hip
int launch_rollout_bwd_xs_win_fast_f32(const RolloutBwdArgs<float>& a, LaneMap m, int integ, int block, bool zmu, bool carry, hipStream_t st);  // rollout_bwd_xs_win_fast.hip
// ... the same with the fused physics loss (LOSS; a.loss_gt set): rollout_bwd_xs_loss_fast.hip, rollout_bwd_xs_win_loss_fast.hip
int launch_rollout_bwd_xs_loss_fast_f32(const RolloutBwdArgs<float>& a, LaneMap m, int integ, int block, bool zmu, hipStream_t st);
int launch_rollout_bwd_xs_win_loss_fast_f32(const RolloutBwdArgs<float>& a, LaneMap m, int integ, int block, bool zmu, bool carry, hipStream_t st);

// defined in rollout_bwd_fast.hip (plain flush) and rollout_bwd_carry_fast.hip (accumulator carry-over)
int launch_rollout_bwd_fast_f32(const RolloutBwdArgs<float>& a, LaneMap m, int integ, int block, hipStream_t st);
int launch_rollout_bwd_carry_fast_f32(const RolloutBwdArgs<float>& a, LaneMap m, int integ, int block, hipStream_t st);
// defined in rollout_bwd_joints.hip (exact arithmetic, like the articulated forward)
int launch_rollout_bwd_joints_f32(const RolloutBwdArgs<float>& a, LaneMap m, int integ, int block, hipStream_t st);
int launch_rollout_bwd_joints_f64(const RolloutBwdArgs<double>& a, LaneMap m, int integ, int block, hipStream_t st);
// defined in rollout_bwd_joints_fast.hip
int launch_rollout_bwd_joints_fast_f32(const RolloutBwdArgs<float>& a, LaneMap m, int integ, int block, hipStream_t st);

}  // namespace mf

// Fused DPhysics rollout, forward pass (gfx950) -- kernel template, instantiated by rollout_fwd.hip (reference-order
// float32/float64 arithmetic) and rollout_fwd_fast.hip (float32 with hardware reciprocal / rsqrt / exp and FMA contraction).
//
// One kernel runs the whole T-step scan of `DPhysics.dphysics()`
// (/root/reference/monoforce/src/monoforce/models/traj_predictor/dphysics.py:530-594): per step the per-contact-point
// height/friction sample (`interpolate_grid`, :385-455, bug-for-bug), the spring-damper + friction contact forces and the
// rigid-body wrench (`forward_kinematics`, :172-272), one Euler step of either integrator (`dynamics` :467-497 /
// `dynamics_odeint` :499-528), and the six API outputs.
//
// Mapping (wave64): a rollout is owned by G consecutive lanes, each lane owning PPL consecutive contact points; the
// 18-float rigid-body state is replicated in the G lanes, so the only cross-lane traffic per step is two all-reduces (sum
// of contact weights; wrench), on DPP for G <= 16.  The time axis is a dependent chain and stays serial in the lane.
// A single wave issues about one instruction per 4-5 cycles whatever the dependences, so a latency-bound launch (few waves
// per SIMD, e.g. B = 1024 x N = 4) wants the FEWEST instructions per wave and step (PPL = 1), while a launch that fills
// the chip wants the least redundant work (PPL = 4: the replicated state update is amortised over 4 points).
//
// Memory: map cells are gathered straight from global memory (both maps are read-only; 2 x 256 x 256 x 4 B = 512 KiB lives
// in every XCD's L2, the handful of cells under a slowly moving robot in the CU's L1 -- DESIGN.md discusses why an LDS
// tile does not pay).  Outputs are time-major by default: the rows a wave writes in one step are contiguous.  CDNA's vmcnt
// retires loads and stores in order, so a wait for any load is a wait for every store issued before it: each step issues
// ALL its loads first, then the (non-temporal) stores of the previous row, and the waits are vmcnt(k >= number of stores).
// Variants: JOINTS (articulated body), FORCES = false (states only), COST (path-cost rows + decimated poses).
#pragma once
#include <cstdlib>
#include "rollout_common.h"

namespace mf {

template <typename S>
struct RolloutArgs {
  int B, T, N, H, W, n_tracks, layout, map_shared, skip_snap, fstride, default_state, ctrl_sb, ctrl_st, b0;
  S mass, inv_mass, mg, k, damp, omega_max, res, inv_res, d_max, dt, half_ly, sink;
  S Iinv[9];
  const S* z;
  const S* mu;
  const S* controls;
  const S* ts;
  const S* points;
  const int* part;
  S* x0;
  const S* xd0;
  const S* R0;
  const S* w0;
  S* Xs;
  S* Xds;
  S* Rs;
  S* Om;
  S* Fs;
  S* Ff;
  S* Xraw;
  const S* joint_angles;  // S[B][T][4] flipper angles (JOINTS kernels only)
  S joint_xyz[12];        // joint positions of the (up to 4) driving parts
  S* cost_rows;           // COST kernels: S[T][B][4] = (R20, R21, R22, std over the points of |F_spring|) per output row
  int pose_stride;        // COST kernels: Xs / Rs hold every pose_stride-th output row only
  S* path_cost;           // COST kernels, optional: S[B] std over the T output rows of the 4th cost-row component
  const S* zmu;           // ZMU kernels: the shared height and friction maps interleaved, S[H*W][2] = (z, mu) per cell
  S* rec;                 // component-parallel kernels, optional: the compact per-step record for the backward, [T][B*16 lanes] 16-byte quads
  // fused physics loss (MfRolloutLoss; component-parallel LOSS kernels): stamps, weights, ground truth, reduction scratch
  int loss_T2;
  const S* loss_gt;
  const S* loss_row_w;
  S* loss_partial;
  unsigned* loss_ticket;
  S* loss_out;
  S loss_inv_count;
  S* loss_poison;          // MF_LOSS_VALUE_IN_BACKWARD: the launch marks the loss as not yet known (NaN)
};

// Arithmetic policy.  Exact: IEEE divide / sqrt, libm exp and sincos, un-fused mul+add (the TU is built with
// -ffp-contract=off) -- tracks the reference's eager float32 op sequence to the last bit for ~100 steps.
// Fast (float32 only): v_rcp / v_rsq / v_exp (1 ulp), FMA contraction; same tolerances hold (DESIGN.md "Numerics").
template <typename S, bool FAST>
struct Mth {
  static __device__ __forceinline__ S div(S a, S b) { return a / b; }
  static __device__ __forceinline__ S cell_coord(S q, S d_max, S res, S) { return (q + d_max) / res; }
  static __device__ __forceinline__ S sqrt(S v) { return mf_sqrt(v); }
  static __device__ __forceinline__ S sigmoid_m10(S dh) { return (S)1 / ((S)1 + mf_exp((S)10 * dh)); }
  // v / max(|v|, eps) given |v|^2
  static __device__ __forceinline__ S inv_len(S len2) { return (S)1 / mf_max(mf_sqrt(len2), (S)1e-6); }
  static __device__ __forceinline__ S len_of(S len2, S) { return mf_sqrt(len2); }      // |v| given |v|^2 (and inv_len(|v|^2), unused here)
  static constexpr bool kReciprocalNorm = false;
  static __device__ __forceinline__ void sincos_small(S v, S* s, S* omc) { S c; mf_sincos(v, s, &c); *omc = (S)1 - c; }
  static __device__ __forceinline__ S clamp(S v, S lo, S hi) { return mf_clamp(v, lo, hi); }   // torch.clamp, NaN propagates
  static __device__ __forceinline__ void sincos(S v, S* sn, S* cs) { mf_sincos(v, sn, cs); }
};
template <>
struct Mth<float, true> {
  static __device__ __forceinline__ float div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
  // The cell coordinate decides an INDEX (`.long()`, dphysics.py:419-420) and the interpolation is discontinuous across cell edges
  // (SURVEY fact 5), so this one quotient is the reference's IEEE division bit for bit also in fast mode: u = a * (1 / res) is off
  // by an ulp for about one argument in four -- enough to truncate a query on a cell edge into the neighbouring cell -- and one
  // Newton step on the exact remainder, u' = fma(fma(-u, res, a), 1 / res, u), is the correctly rounded a / res when 1 / res is
  // itself correctly rounded (Markstein's theorem; the host passes inv_res = RN(1 / res): tools/check_exact_div.py verifies it
  // against the division for every float32 argument on the map at res = 0.05 / 0.1).  Two instructions on two lanes.
  static __device__ __forceinline__ float cell_coord(float q, float d_max, float res, float inv_res) {
    const float a = q + d_max;
    const float u = a * inv_res;
    return fmaf(fmaf(-u, res, a), inv_res, u);
  }
  static __device__ __forceinline__ float sqrt(float v) { return __builtin_amdgcn_sqrtf(v); }
  static __device__ __forceinline__ float sigmoid_m10(float dh) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(14.426950408889634f * dh));  // exp(10 dh) = 2^(10 log2(e) dh)
  }
  static __device__ __forceinline__ float inv_len(float len2) { return __builtin_amdgcn_rsqf(fmaxf(len2, 1e-12f)); }
  // |v| = |v|^2 / |v| from the reciprocal at hand: a multiply instead of a second transcendental (below the 1e-6 floor of inv_len
  // the product is |v|^2 1e6 < |v|: an angle of < 1e-8 rad per step either way)
  static __device__ __forceinline__ float len_of(float len2, float il) { return len2 * il; }
  static constexpr bool kReciprocalNorm = true;
  static __device__ __forceinline__ float clamp(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }  // 1 instr
  // joint angles (|a| of a few radians at most): hardware sin / cos after the 1 / 2pi scaling, ~1e-6 absolute
  static __device__ __forceinline__ void sincos(float v, float* sn, float* cs) { *sn = __sinf(v); *cs = __cosf(v); }
  // sin(v), 1 - cos(v) of the Rodrigues step, v = |w| dt ~ 1e-2.  Branch-free (a branch here splits the basic block the
  // backward's two instruction streams are interleaved in): |v| < 1 / 4 -- Taylor series to v^7 / v^8 (truncation < 1e-11
  // relative), and 1 - cos without the cancellation; a body spinning faster than 25 rad/s at dt = 0.01 gets the hardware
  // sin / cos (~1e-6 absolute on values of order 1).  (Round 4: the series ran to v^11 / v^12 below |v| < 1 -- four more FMAs
  // in a loop that runs at its issue bound.)
  static __device__ __forceinline__ void sincos_small(float v, float* s, float* omc) {
    const float v2 = v * v;
    const float ts = v * (1.0f + v2 * (-1.0f / 6 + v2 * (1.0f / 120 - v2 * (1.0f / 5040))));
    const float tc = v2 * (0.5f + v2 * (-1.0f / 24 + v2 * (1.0f / 720 - v2 * (1.0f / 40320))));
    const float rev = __builtin_amdgcn_fractf(v * 0.15915494309189535f);
    const bool small = fabsf(v) < 0.25f;
    *s = small ? ts : __builtin_amdgcn_sinf(rev);
    *omc = small ? tc : 1.0f - __builtin_amdgcn_cosf(rev);
  }
};

template <typename S, bool FAST>
__device__ __forceinline__ Cell<S> locate_m(S qx, S qy, S d_max, S res, S inv_res, int H, int last) {
  const S lim = (S)262144.0;
  S ux = Mth<S, FAST>::cell_coord(qx, d_max, res, inv_res);
  S uy = Mth<S, FAST>::cell_coord(qy, d_max, res, inv_res);
  int ix = (int)Mth<S, FAST>::clamp(ux, -lim, lim);  // trunc toward zero, like .long()
  int iy = (int)Mth<S, FAST>::clamp(uy, -lim, lim);
  Cell<S> c;
  c.fx = ux - (S)ix;
  c.fy = uy - (S)iy;
  int base = iy + __mul24(H, ix);   // |ix| <= 2^18 and H < 2^23 (host-checked): the 24-bit multiply is exact, one v_mad_i32_i24
  c.ic = min(max(base, 0), last);
  c.i_f = min(max(base + H, 0), last);
  c.il = min(max(base + 1, 0), last);
  c.ifl = min(max(base + 1 + H, 0), last);
  return c;
}

// Load base[elem] with a wave-uniform base pointer and a 32-bit element offset: the byte offset is formed in 32 bits (the
// host guarantees B*H*W*sizeof(S) < 4 GiB for per-rollout maps) so the access uses the scalar-base + 32-bit-offset form.
template <typename S>
__device__ __forceinline__ S ld32(const S* base, unsigned elem) {
  return *reinterpret_cast<const S*>(reinterpret_cast<const char*>(base) + (size_t)(elem * (unsigned)sizeof(S)));
}

// The four cells of a bilinear footprint with TWO loads instead of four.  c/l and f/fl are neighbours in memory (flat index
// base, base + 1 and base + H, base + H + 1) unless the reference's clamp of the FLAT index (dphysics.py:432-435) folds them
// onto cell 0 or HW - 1, so each pair is one 2-element load at p = min(index, HW - 2) and the clamped cases pick the element
// they name: bit-identical to four single loads for every index, in or out of the map.  The L1 (TCP) looks up one line per
// lane and cycle for these divergent gathers -- at 16 waves per CU that, not VALU issue or HBM, bounds the saturated kernel
// (PMC: TCP_TOTAL_CACHE_ACCESSES = 1 per cycle and CU), so halving the lane accesses is what counts.  v = (c, f, l, fl).
template <typename S>
struct CellPair { S a, b; };
template <typename S>
__device__ __forceinline__ void gather4(const S* map, unsigned moff, const Cell<S>& c, int last, S (&v)[4]) {
  const int p1 = min(c.ic, last - 1), p2 = min(c.i_f, last - 1);   // last >= 1 (host-checked: H >= 2)
  const CellPair<S> q1 = *reinterpret_cast<const CellPair<S>*>(reinterpret_cast<const char*>(map) + (size_t)((moff + (unsigned)p1) * (unsigned)sizeof(S)));
  const CellPair<S> q2 = *reinterpret_cast<const CellPair<S>*>(reinterpret_cast<const char*>(map) + (size_t)((moff + (unsigned)p2) * (unsigned)sizeof(S)));
  v[0] = c.ic != p1 ? q1.b : q1.a;
  v[2] = c.il != p1 ? q1.b : q1.a;
  v[1] = c.i_f != p2 ? q2.b : q2.a;
  v[3] = c.ifl != p2 ? q2.b : q2.a;
}

// ZMU kernels (one map pair shared by all rollouts): height and friction interleaved cell by cell, so the footprint of a
// point in BOTH maps is two 4-element loads: (z_c, mu_c, z_l, mu_l) and (z_f, mu_f, z_fl, mu_fl).  Same values, same
// clamp handling as gather4 -- a quarter of the L1 lookups of eight single loads.
template <typename S>
struct CellQuad { S za, ma, zb, mb; };
template <typename S>
__device__ __forceinline__ void gather4x2(const S* zmu, const Cell<S>& c, int last, S (&z)[4], S (&m)[4]) {
  const int p1 = min(c.ic, last - 1), p2 = min(c.i_f, last - 1);
  const CellQuad<S> q1 = *reinterpret_cast<const CellQuad<S>*>(reinterpret_cast<const char*>(zmu) + (size_t)((unsigned)p1 * (unsigned)(2 * sizeof(S))));
  const CellQuad<S> q2 = *reinterpret_cast<const CellQuad<S>*>(reinterpret_cast<const char*>(zmu) + (size_t)((unsigned)p2 * (unsigned)(2 * sizeof(S))));
  const bool ec = c.ic != p1, el = c.il != p1, ef = c.i_f != p2, efl = c.ifl != p2;
  z[0] = ec ? q1.zb : q1.za;   m[0] = ec ? q1.mb : q1.ma;
  z[2] = el ? q1.zb : q1.za;   m[2] = el ? q1.mb : q1.ma;
  z[1] = ef ? q2.zb : q2.za;   m[1] = ef ? q2.mb : q2.ma;
  z[3] = efl ? q2.zb : q2.za;  m[3] = efl ? q2.mb : q2.ma;
}

// update_joints (dphysics.py:326-358): rotate every driving part about the y-axis through its joint by the step's angle, then
// the inertia of the articulated body about the body origin and its inverse (dphysics.py:196-197, 107-141) -- per step and per
// rollout; `ja` = the 4 joint angles of this (rollout, step), P0 = rest configuration, P / Iv = articulated points / I^-1.
template <typename S, int G, int PPL, bool FAST = false>
__device__ __forceinline__ void articulate_body(GroupSum<G, S>& gs, const S* ja, const S* joint_xyz, S mp, const S (&P0)[PPL][3],
                                                const int (&part)[PPL], const bool (&act)[PPL], S (&P)[PPL][3], S (&Iv)[9]) {
  const S one = (S)1, zero = (S)0;
  S sj[4], cj4[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) Mth<S, FAST>::sincos(ja[q], &sj[q], &cj4[q]);
  S I6[6] = {zero, zero, zero, zero, zero, zero};   // xx, yy, zz, xy, xz, yz
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    const int q = max(part[j], 0);
    const S sn = sj[q], cs = cj4[q];
    const S jx = joint_xyz[q * 3 + 0], jz = joint_xyz[q * 3 + 2];
    const S dx = P0[j][0] - jx, dz = P0[j][2] - jz;
    const S rx = (dx * cs + dz * sn) + jx, rz = (-(dx * sn) + dz * cs) + jz;   // (p - xyz) @ Ry^T + xyz
    P[j][0] = part[j] >= 0 ? rx : P0[j][0];
    P[j][1] = P0[j][1];
    P[j][2] = part[j] >= 0 ? rz : P0[j][2];
    const S px = P[j][0], py = P[j][1], pzz = P[j][2];
    const S wgt = act[j] ? mp : zero;
    I6[0] += wgt * (py * py + pzz * pzz); I6[1] += wgt * (px * px + pzz * pzz); I6[2] += wgt * (px * px + py * py);
    I6[3] -= wgt * px * py; I6[4] -= wgt * px * pzz; I6[5] -= wgt * py * pzz;
  }
  gs.sum_n(I6);
  // inverse of the symmetric 3x3 by cofactors
  const S a00 = I6[0], a11 = I6[1], a22 = I6[2], a01 = I6[3], a02 = I6[4], a12 = I6[5];
  const S c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
  const S c11 = a00 * a22 - a02 * a02, c12 = a01 * a02 - a00 * a12, c22 = a00 * a11 - a01 * a01;
  const S idet = Mth<S, FAST>::div(one, a00 * c00 + a01 * c01 + a02 * c02);
  Iv[0] = c00 * idet; Iv[1] = c01 * idet; Iv[2] = c02 * idet;
  Iv[3] = c01 * idet; Iv[4] = c11 * idet; Iv[5] = c12 * idet;
  Iv[6] = c02 * idet; Iv[7] = c12 * idet; Iv[8] = c22 * idet;
}

template <typename S, int G, int PPL, int INTEG, bool FAST, bool JOINTS = false, bool FORCES = true, int COST = 0, bool SPLIT = false, bool ZMU = false, bool REC = false>
__global__ void __launch_bounds__(G > 256 ? G : 256) rollout_fwd_kernel(const RolloutArgs<S> a) {
  using M = Mth<S, FAST>;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = tid / G + a.b0;   // b0: first rollout of this launch (very large batches go out in several launches)
  const int gl = tid % G;
  if (b >= a.B) return;  // whole groups leave together; live groups never read dead lanes
  const S one = (S)1, zero = (S)0;
  const int HW = a.H * a.W, last = HW - 1;
  const bool has_mu = a.mu != nullptr;  // wave-uniform
  // group reductions; a rollout spread over several waves (G > 64: one workgroup = one rollout) exchanges through LDS
  __shared__ S gs_lds[G > 64 ? 2 * (G / 64) * kGroupSumMaxValues : 1];
  GroupSum<G, S> gs;
  gs.lds = gs_lds;
  // uniform base pointer + 32-bit element offset (the host guarantees B*H*W < 2^31 for per-rollout maps): scalar-base loads
  const unsigned moff = a.map_shared ? 0u : (unsigned)b * (unsigned)HW;
  const S* zmap = a.z;
  const S* mumap = has_mu ? a.mu : a.z;

  // this lane's contact points
  S P[PPL][3];   // contact points used by the step (articulated per step when JOINTS)
  S P0[PPL][3];  // rest configuration (cfg.robot_points)
  S Iv[9];       // inverse inertia used by the step
  int part[PPL];
  bool act[PPL];
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    int i = gl * PPL + j;  // blocked: a lane owns PPL consecutive points (contiguous force rows per lane)
    act[j] = i < a.N;
    int ii = act[j] ? i : 0;
    P[j][0] = a.points[ii * 3 + 0];
    P[j][1] = a.points[ii * 3 + 1];
    P[j][2] = a.points[ii * 3 + 2];
    part[j] = act[j] ? a.part[ii] : -1;
#pragma unroll
    for (int c = 0; c < 3; ++c) P0[j][c] = P[j][c];
  }
#pragma unroll
  for (int c = 0; c < 9; ++c) Iv[c] = a.Iinv[c];

  // state, replicated across the group
  S x[3], xd[3], R[9], w[3];
  if (a.default_state) {   // the reference's default start (dphysics.py:554-559), written back for the caller / the backward
    const S v0 = a.controls[(size_t)b * a.ctrl_sb + 0], w0 = a.controls[(size_t)b * a.ctrl_sb + 1];
#pragma unroll
    for (int c = 0; c < 3; ++c) { x[c] = zero; xd[c] = c == 0 ? v0 : zero; w[c] = c == 2 ? w0 : zero; }
#pragma unroll
    for (int c = 0; c < 9; ++c) R[c] = (c % 4 == 0) ? one : zero;
    if (gl == 0) {
      S* oxd = const_cast<S*>(a.xd0); S* oR = const_cast<S*>(a.R0); S* ow = const_cast<S*>(a.w0);
#pragma unroll
      for (int c = 0; c < 3; ++c) { a.x0[b * 3 + c] = x[c]; oxd[b * 3 + c] = xd[c]; ow[b * 3 + c] = w[c]; }
#pragma unroll
      for (int c = 0; c < 9; ++c) oR[b * 9 + c] = R[c];
    }
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      x[c] = a.x0[b * 3 + c];
      xd[c] = a.xd0[b * 3 + c];
      w[c] = a.w0[b * 3 + c];
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) R[c] = a.R0[b * 9 + c];
  }

  // start at the terrain height: x.z <- mean_i interp(z, (P R^T + x)_i)   (dphysics.py:567-571)
  if (!a.skip_snap) {
    S acc = zero;
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      S px = P[j][0] * R[0] + P[j][1] * R[1] + P[j][2] * R[2] + x[0];
      S py = P[j][0] * R[3] + P[j][1] * R[4] + P[j][2] * R[5] + x[1];
      Cell<S> c = locate_m<S, FAST>(px, py, a.d_max, a.res, a.inv_res, a.H, last);
      S z4[4];
      gather4(zmap, moff, c, last, z4);
      S v = blend(c, z4[0], z4[1], z4[2], z4[3]);
      acc += act[j] ? v : zero;
    }
    acc = gs.sum(acc);
    x[2] = acc / (S)a.N;
    if (gl == 0) a.x0[b * 3 + 2] = x[2];
  }

  // running output pointers (one bump per step instead of 64-bit index arithmetic)
  const size_t row_stride = (a.layout == MF_LAYOUT_TIME_MAJOR) ? (size_t)a.B : 1;  // rows between consecutive t
  const size_t row0 = (a.layout == MF_LAYOUT_TIME_MAJOR) ? (size_t)b : (size_t)b * a.T;
  S* pXs = a.Xs + row0 * 3;
  S* pXds = a.Xds + row0 * 3;
  S* pOm = a.Om + row0 * 3;
  S* pRs = a.Rs + row0 * 9;
  // without a request for the unshifted positions they are written to the Xs row itself and then overwritten by the shifted
  // ones (same lane, same address, program order): no branch in the store sequence, no extra HBM traffic
  S* pXraw = (a.Xraw ? a.Xraw : a.Xs) + row0 * 3;
  const size_t frow = (size_t)a.fstride * 3;  // floats per force row (>= G * PPL points: stores need no predication)
  S* pFs = a.Fs + row0 * frow + (size_t)gl * PPL * 3;
  S* pFf = a.Ff + row0 * frow + (size_t)gl * PPL * 3;

  S oFs[PPL][3], oFf[PPL][3];  // forces of the pending output row (ODEINT: running impulses, dphysics.py:506-509)
#pragma unroll
  for (int j = 0; j < PPL; ++j)
#pragma unroll
    for (int c = 0; c < 3; ++c) oFs[j][c] = oFf[j][c] = zero;

  // Stores of the pending output row (the state registers ARE that row).  Every lane of the group stores the same state
  // values to the same addresses (no exec-mask branch on the issue stream; the coalescer merges them) and its own forces.
  // COST kernels (trajectory shooting): instead of the 84 + 24 N bytes of a full output row they write one
  // 16-byte cost row -- what the reference's path costs read (monoforce_node.py:91: norm(F_springs).std(points);
  // diff_physics.py:263-266: roll / pitch from the last row of R) -- and keep every pose_stride-th pose (the nodes publish
  // poses[::pose_step]).  Pose row r holds output row m = min(r * pose_stride, T - 1).  A pose is stored by a wave-uniform
  // branch at the very END of the step that produced it: the arithmetic of a step stays one basic block, so the FMA
  // contraction -- and with it every bit of the trajectory -- is the one of the full-output kernels.
  S* pC = COST ? a.cost_rows + row0 * 4 : nullptr;
  // SPLIT kernels: the replicated state of a group is written by its lanes in turn -- lane role (gl & 3) = 0..3 stores the
  // vec3 of Xraw / Xs / Xds / Omegas and row min(role, 2) of R: 2 store instructions per step instead of 7 four-times-redundant
  // ones (+2 for the forces).  Once every SIMD has a wave the CU's memory pipeline, which takes ~16 cycles per instruction
  // whatever its payload, bounds the kernel (B = 16384: 0.67 -> 0.53 ms); below that the 15 selects it costs lose
  // (B = 1024: 0.31 -> 0.34 ms), so the host picks by launch size.
  const int role = gl & 3;
  const bool role_b0 = (role & 1) != 0, role_b1 = (role & 2) != 0;
  // roles 0 and 1 both store x + R[:, 2] * s: s = 0 is the unshifted Xraw row, s = sink the Xs row; without an Xraw buffer
  // role 0 writes the Xs row as well (same address, same value as role 1)
  const S sink_l = (role == 0 && a.Xraw) ? zero : a.sink;
  S* pV3 = role == 0 ? pXraw : role == 1 ? pXs : role == 2 ? pXds : pOm;
  S* pRrow = pRs + 3 * min(role, 2);
  int pose_wait = 0;   // output rows still to pass before the next pose is due
  S pc_n = zero, pc_mean = zero, pc_m2 = zero;   // path cost accumulators (COST)
  auto store_pose = [&]() {
    pXs[0] = x[0] + R[2] * a.sink; pXs[1] = x[1] + R[5] * a.sink; pXs[2] = x[2] + R[8] * a.sink;
#pragma unroll
    for (int c = 0; c < 9; ++c) pRs[c] = R[c];
  };
  const S inv_N = one / (S)a.N, inv_Nm1 = one / (S)max(a.N - 1, 1);
  auto emit_row = [&](size_t adv) {
#ifdef MF_DBG_NOSTORE   // A/B hook (tools/ab_rollout.py): time the kernel without its output stream
    if (a.B > 0) return;
#endif
    if (COST) {
      auto stc = [](S* p, S v) { __builtin_nontemporal_store(v, p); };
      S nrm_j[PPL], nsum = zero;
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        nrm_j[j] = M::sqrt(oFs[j][0] * oFs[j][0] + oFs[j][1] * oFs[j][1] + oFs[j][2] * oFs[j][2]);
        nsum += act[j] ? nrm_j[j] : zero;
      }
      const S mean = gs.sum(nsum) * inv_N;
      S dsum = zero;
#pragma unroll
      for (int j = 0; j < PPL; ++j) { const S dv = nrm_j[j] - mean; dsum += act[j] ? dv * dv : zero; }
      const S sdev = M::sqrt(gs.sum(dsum) * inv_Nm1);   // unbiased, like torch.std
      // Third row of the rotation the cost is read from.  The explicit Euler scheme lets R drift off SO(3) (|R R^T - I| up to
      // 0.06 after 500 steps) and the reference takes roll / pitch through scipy's `Rotation.from_matrix`, which first
      // projects onto the nearest rotation (polar factor U V^T): two Newton steps X <- (X + X^-T) / 2 reproduce it to ~1e-7
      // (the second one only for the row that is stored).  dynamics() keeps R orthonormal by construction: stored as is.
      S r20 = R[6], r21 = R[7], r22 = R[8];
      if (INTEG == MF_INTEG_ODEINT_EULER && COST == 2) {   // COST = 2: the caller reads roll / pitch from the rows
        const S half = (S)0.5;
        const S c00 = R[4] * R[8] - R[5] * R[7], c01 = R[5] * R[6] - R[3] * R[8], c02 = R[3] * R[7] - R[4] * R[6];
        const S c10 = R[2] * R[7] - R[1] * R[8], c11 = R[0] * R[8] - R[2] * R[6], c12 = R[1] * R[6] - R[0] * R[7];
        const S c20 = R[1] * R[5] - R[2] * R[4], c21 = R[2] * R[3] - R[0] * R[5], c22 = R[0] * R[4] - R[1] * R[3];
        const S hid = M::div(half, R[0] * c00 + R[1] * c01 + R[2] * c02);          // 1 / (2 det)
        const S Y[9] = {half * R[0] + hid * c00, half * R[1] + hid * c01, half * R[2] + hid * c02,
                        half * R[3] + hid * c10, half * R[4] + hid * c11, half * R[5] + hid * c12,
                        half * R[6] + hid * c20, half * R[7] + hid * c21, half * R[8] + hid * c22};
        const S d20 = Y[1] * Y[5] - Y[2] * Y[4], d21 = Y[2] * Y[3] - Y[0] * Y[5], d22 = Y[0] * Y[4] - Y[1] * Y[3];
        const S hid2 = M::div(half, Y[6] * d20 + Y[7] * d21 + Y[8] * d22);
        r20 = half * Y[6] + hid2 * d20; r21 = half * Y[7] + hid2 * d21; r22 = half * Y[8] + hid2 * d22;
      }
      stc(pC + 0, r20); stc(pC + 1, r21); stc(pC + 2, r22); stc(pC + 3, sdev);
      pC += adv * 4;
      // running mean / sum of squared deviations of sdev over the output rows (Welford); DYNAMICS' placeholder row (adv = 0)
      // does not count
      const S wgt = adv ? one : zero;
      pc_n += wgt;
      const S dlt = sdev - pc_mean;
      pc_mean += wgt * M::div(dlt, mf_max(pc_n, one));
      pc_m2 += wgt * dlt * (sdev - pc_mean);
      return;   // the decimated poses are written at the END of the step that produced them (store_pose below)
    }
    // streaming (non-temporal) stores: the rows are never read again by this kernel and must not evict the map cells the
    // gathers keep hitting in L1 / L2
    auto st = [](S* p, S v) { __builtin_nontemporal_store(v, p); };
    if (SPLIT && G >= 4) {
      S v3[3], rr[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const S t0 = x[c] + R[3 * c + 2] * sink_l;
        const S t1 = role_b0 ? w[c] : xd[c];
        v3[c] = role_b1 ? t1 : t0;
        const S r0 = role_b0 ? R[3 + c] : R[c];
        rr[c] = role_b1 ? R[6 + c] : r0;
      }
      st(pV3 + 0, v3[0]); st(pV3 + 1, v3[1]); st(pV3 + 2, v3[2]);
      st(pRrow + 0, rr[0]); st(pRrow + 1, rr[1]); st(pRrow + 2, rr[2]);
      pV3 += adv * 3; pRrow += adv * 9;
      if (FORCES) {
#pragma unroll
        for (int j = 0; j < PPL; ++j)
#pragma unroll
          for (int c = 0; c < 3; ++c) st(pFs + j * 3 + c, oFs[j][c]);
#pragma unroll
        for (int j = 0; j < PPL; ++j)
#pragma unroll
          for (int c = 0; c < 3; ++c) st(pFf + j * 3 + c, oFf[j][c]);
        pFs += adv * frow; pFf += adv * frow;
      }
      return;
    }
    // A rollout over several waves (fast kernels): the waves hold the same state, so they share its stores --
    // wave 0 the positions, wave 1 the velocities, wave 2 the rotation (two waves: 0 and 1 split them) -- by wave-uniform branches.
    // -DMF_NO_SHARED_STATE_STORES: every wave stores everything (A/B).
#ifdef MF_NO_SHARED_STATE_STORES
    constexpr bool kShare = false;
#else
    constexpr bool kShare = FAST && G > 64 && PPL == 1 && !JOINTS;
#endif
    const int wv = kShare ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    constexpr int kNW = G > 64 ? G / 64 : 1;
    const bool do_x = !kShare || wv == 0, do_v = !kShare || wv == (kNW > 2 ? 1 : 0), do_R = !kShare || wv == (kNW > 2 ? 2 : 1);
    if (do_x) {
      st(pXraw + 0, x[0]); st(pXraw + 1, x[1]); st(pXraw + 2, x[2]);
      st(pXs + 0, x[0] + R[2] * a.sink);  // Xs += Rs[..., :, 2] * m g / (k + 1e-6)   (dphysics.py:587-589)
      st(pXs + 1, x[1] + R[5] * a.sink);
      st(pXs + 2, x[2] + R[8] * a.sink);
    }
    if (do_v) {
      st(pXds + 0, xd[0]); st(pXds + 1, xd[1]); st(pXds + 2, xd[2]);
      st(pOm + 0, w[0]); st(pOm + 1, w[1]); st(pOm + 2, w[2]);
    }
    if (do_R) {
#pragma unroll
      for (int c = 0; c < 9; ++c) st(pRs + c, R[c]);
    }
    pXs += adv * 3; pXds += adv * 3; pOm += adv * 3; pRs += adv * 9; pXraw += adv * 3;
    if (FORCES) {   // compile-time: callers that only consume the states (training) skip 24 N of the 80 + 56 N bytes per step
#pragma unroll
      for (int j = 0; j < PPL; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) st(pFs + j * 3 + c, oFs[j][c]);
#pragma unroll
      for (int j = 0; j < PPL; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) st(pFf + j * 3 + c, oFf[j][c]);
      pFs += adv * frow; pFf += adv * frow;
    }
  };

  const int n_steps = (INTEG == MF_INTEG_ODEINT_EULER) ? a.T - 1 : a.T;

  const S* ctrl = a.controls + (size_t)b * a.ctrl_sb;   // row of this rollout; ctrl_st = 0: one (v, w) for the whole horizon
  S cv = ctrl[0], cw = ctrl[1];
  S h_ode = (INTEG == MF_INTEG_ODEINT_EULER && a.T > 1) ? a.ts[1] - a.ts[0] : zero;   // step size of the current step

  if (COST && INTEG == MF_INTEG_ODEINT_EULER) {   // output row 0 = the initial state (DYNAMICS: row 0 is produced by step 0)
    store_pose();
    pXs += row_stride * 3; pRs += row_stride * 9;
    pose_wait = a.pose_stride - 1;
  }
  // everything loaded so far is complete before the loop: otherwise the waits inside it also have to cover these loads, and a
  // conservative in-loop wait is a wait for the previous step's stores
  __builtin_amdgcn_s_waitcnt(0);
  // PIPE: a rollout spread over several waves (G > 64), default integrator.  Its two group sums cost an LDS round trip and a
  // barrier each, and one wave per SIMD has nothing to put under them -- unless the step is software-pipelined like the
  // component-parallel kernels' (rollout_fwd_cp_kernel.h): the explicit scheme knows pose n + 1 when step n starts, so its
  // body-frame arms go between the contact count's LDS write and its barrier, its footprints and the requests for their cells
  // between the wrench's write and its barrier, and the cells have the rest of the step and the head of the next to arrive.
  // (Rollouts INSIDE a wave, G = 8 .. 64, keep the plain step: measured with -DMF_PIPE_MIN_G=8, forward with forces, pipelined / plain --
  //  1024 x 32 points 0.395 / 0.358 ms, 256 x 64 0.394 / 0.367, 2048 x 20 0.437 / 0.419; 256 x 16 0.304 / 0.308, 512 x 8 0.297 / 0.299:
  //  their group sums are DPP adds with nothing to hide, and the pipelined form holds two footprints in registers.)
#ifdef MF_PIPE_MIN_G      // A/B hook: the smallest group that takes the pipelined step
  constexpr int kPipeMinG = MF_PIPE_MIN_G;
#else
  constexpr int kPipeMinG = 128;
#endif
  constexpr bool PIPE = FAST && G >= kPipeMinG && PPL == 1 && INTEG == MF_INTEG_ODEINT_EULER && !JOINTS && COST == 0;
  if constexpr (PIPE) {
    // ---- the pieces of one step ----
    // geometry of the contact points under the pose (x, R) and the gathers that depend only on it.  PIPE kernels request the cells
    // (`request`) a step before they select the clamped corners out of them (`arrive`); the others do both at once.
    struct Geo {
      S r[PPL][3], pz[PPL];
      Cell<S> cell[PPL];
      S zc[PPL][4], mc[PPL][4];
      CellQuad<S> q1, q2;          // PIPE: the loads as they were requested (ZMU: (z, mu) x 2 cells; else q1 = z pairs, q2 = mu pairs)
    };
    auto locate_points = [&](Geo& g) {
  #pragma unroll
      for (int j = 0; j < PPL; ++j) {
        // p = P R^T + x ; r = p - x   (:200)
        S px = P[j][0] * R[0] + P[j][1] * R[1] + P[j][2] * R[2] + x[0];
        S py = P[j][0] * R[3] + P[j][1] * R[4] + P[j][2] * R[5] + x[1];
        g.pz[j] = P[j][0] * R[6] + P[j][1] * R[7] + P[j][2] * R[8] + x[2];
        g.r[j][0] = px - x[0]; g.r[j][1] = py - x[1]; g.r[j][2] = g.pz[j] - x[2];
        g.cell[j] = locate_m<S, FAST>(px, py, a.d_max, a.res, a.inv_res, a.H, last);
      }
    };
    auto geometry = [&](Geo& g) {
      locate_points(g);
  #pragma unroll
      for (int j = 0; j < PPL; ++j) {
        const Cell<S>& c = g.cell[j];
        if constexpr (ZMU) gather4x2(a.zmu, c, last, g.zc[j], g.mc[j]);
        else gather4(zmap, moff, c, last, g.zc[j]);
      }
  #pragma unroll
      for (int j = 0; j < PPL; ++j) {   // unconditional (mumap aliases z without a friction map; selected after the blend): no branch
        const Cell<S>& c = g.cell[j];
        if constexpr (!ZMU) gather4(mumap, moff, c, last, g.mc[j]);
      }
    };
    // kinematics that need no gathered cell: point velocities, thrust direction, track speeds
    struct Kin { S vp[PPL][3], e0, e1, e2, tv_lo, tv_hi; };
    auto kinematics = [&](const Geo& g, Kin& k) {
  #pragma unroll
      for (int j = 0; j < PPL; ++j) {  // v_p = xd + w x r   (:204)
        k.vp[j][0] = xd[0] + (w[1] * g.r[j][2] - w[2] * g.r[j][1]);
        k.vp[j][1] = xd[1] + (w[2] * g.r[j][0] - w[0] * g.r[j][2]);
        k.vp[j][2] = xd[2] + (w[0] * g.r[j][1] - w[1] * g.r[j][0]);
      }
      // thrust direction = normalized first column of R   (:237)
      const S il = M::inv_len(R[0] * R[0] + R[3] * R[3] + R[6] * R[6]);
      if (M::kReciprocalNorm) { k.e0 = R[0] * il; k.e1 = R[3] * il; k.e2 = R[6] * il; }
      else { const S el = mf_max(M::sqrt(R[0] * R[0] + R[3] * R[3] + R[6] * R[6]), (S)1e-6); k.e0 = R[0] / el; k.e1 = R[3] / el; k.e2 = R[6] / el; }
      k.tv_lo = cv - cw * a.half_ly; k.tv_hi = cv + cw * a.half_ly;  // (:75-104)
    };
    // The explicit scheme moves x with the OLD xd and R with the OLD w: neither depends on this step's forces.
    auto advance_pose = [&](S h) {
      S dR[9];
  #pragma unroll
      for (int j2 = 0; j2 < 3; ++j2) {
        dR[0 * 3 + j2] = w[1] * R[2 * 3 + j2] - w[2] * R[1 * 3 + j2];
        dR[1 * 3 + j2] = w[2] * R[0 * 3 + j2] - w[0] * R[2 * 3 + j2];
        dR[2 * 3 + j2] = w[0] * R[1 * 3 + j2] - w[1] * R[0 * 3 + j2];
      }
  #pragma unroll
      for (int c = 0; c < 3; ++c) x[c] = x[c] + h * xd[c];
  #pragma unroll
      for (int c = 0; c < 9; ++c) R[c] = R[c] + h * dR[c];
    };
    // contact model, up to the (per-lane partial of the) contact count
    struct Con { S nrm[PPL][3], muq[PPL], cw8[PPL], Fr[PPL][3], Ff[PPL][3]; };
    auto contact_normal = [&](const Geo& g, const Kin& k, Con& q) {
      S csum = zero;
  #pragma unroll
      for (int j = 0; j < PPL; ++j) {
        const Cell<S>& c = g.cell[j];
        S zq, mub;                                                 // height, normal, friction under the point (:211-216)
        blend2(c, g.zc[j], g.mc[j], &zq, &mub);
        // no friction map = a map of ones (dphysics.py:562): its blend is the (rounded) sum of the four weights, no loads involved
        q.muq[j] = has_mu ? mub : blend_ones(c);
        S gx = M::div(g.zc[j][1] - g.zc[j][0], a.res), gy = M::div(g.zc[j][2] - g.zc[j][0], a.res);
        if (M::kReciprocalNorm) {
          const S inl = M::inv_len(gx * gx + gy * gy + one);
          q.nrm[j][0] = -gx * inl; q.nrm[j][1] = -gy * inl; q.nrm[j][2] = inl;
        } else {
          const S nl = mf_max(M::sqrt(gx * gx + gy * gy + one), (S)1e-6);
          q.nrm[j][0] = -gx / nl; q.nrm[j][1] = -gy / nl; q.nrm[j][2] = one / nl;
        }
        S dh = g.pz[j] - zq;  // soft contact + spring-damper along the normal   (:220-230)
        S cj = M::sigmoid_m10(dh);
        cj = act[j] ? cj : zero;
        q.cw8[j] = cj;
        csum += cj;
        S vn = k.vp[j][0] * q.nrm[j][0] + k.vp[j][1] * q.nrm[j][1] + k.vp[j][2] * q.nrm[j][2];
        S A = a.k * dh + a.damp * vn;
        q.Fr[j][0] = -(A * q.nrm[j][0]); q.Fr[j][1] = -(A * q.nrm[j][1]); q.Fr[j][2] = -(A * q.nrm[j][2]);
      }
      return csum;
    };
    // ... from the contact count to the (per-lane partials of the) wrench: wr = (sum F [or sum Fs, sum Ff], sum tau)
    constexpr int kWr = FAST ? 6 : 9;
    auto contact_wrench = [&](const Geo& g, const Kin& k, Con& q, S csum, S (&wr)[kWr]) {
      const S inv_csum = FAST ? M::div(one, csum) : one;
      S sFr[3] = {zero, zero, zero}, sFf[3] = {zero, zero, zero}, sTau[3] = {zero, zero, zero};
  #pragma unroll
      for (int j = 0; j < PPL; ++j) {
  #pragma unroll
        for (int c = 0; c < 3; ++c) {  // (:232-233)
          S f = FAST ? q.Fr[j][c] * q.cw8[j] * inv_csum : q.Fr[j][c] * q.cw8[j] / csum;
          q.Fr[j][c] = M::clamp(f, -a.mg, a.mg);
        }
        S Nn = M::sqrt(q.Fr[j][0] * q.Fr[j][0] + q.Fr[j][1] * q.Fr[j][1] + q.Fr[j][2] * q.Fr[j][2]);  // (:238)
        S tv = (part[j] < 0) ? zero : ((part[j] & 1) ? k.tv_hi : k.tv_lo);
        S s0 = q.muq[j] * (tv * k.e0 - k.vp[j][0]);  // slip (:247); cmd = 0 for non-driving points
        S s1 = q.muq[j] * (tv * k.e1 - k.vp[j][1]);
        S s2 = q.muq[j] * (tv * k.e2 - k.vp[j][2]);
        S sn = s0 * q.nrm[j][0] + s1 * q.nrm[j][1] + s2 * q.nrm[j][2];
        q.Ff[j][0] = M::clamp(Nn * (s0 - sn * q.nrm[j][0]), -a.mg, a.mg);  // (:248-251)
        q.Ff[j][1] = M::clamp(Nn * (s1 - sn * q.nrm[j][1]), -a.mg, a.mg);
        q.Ff[j][2] = M::clamp(Nn * (s2 - sn * q.nrm[j][2]), -a.mg, a.mg);
        if (!act[j]) {
  #pragma unroll
          for (int c = 0; c < 3; ++c) q.Fr[j][c] = q.Ff[j][c] = zero;
        }
        S f0 = q.Fr[j][0] + q.Ff[j][0], f1 = q.Fr[j][1] + q.Ff[j][1], f2 = q.Fr[j][2] + q.Ff[j][2];
        sTau[0] += g.r[j][1] * f2 - g.r[j][2] * f1;  // r x (Fs + Ff)   (:255)
        sTau[1] += g.r[j][2] * f0 - g.r[j][0] * f2;
        sTau[2] += g.r[j][0] * f1 - g.r[j][1] * f0;
        if (FAST) { sFr[0] += f0; sFr[1] += f1; sFr[2] += f2; }
        else {
  #pragma unroll
          for (int c = 0; c < 3; ++c) { sFr[c] += q.Fr[j][c]; sFf[c] += q.Ff[j][c]; }
        }
      }
      if constexpr (FAST) {   // one batched reduction of the wrench (multi-wave groups: one LDS exchange)
  #pragma unroll
        for (int c = 0; c < 3; ++c) { wr[c] = sFr[c]; wr[3 + c] = sTau[c]; }
      } else {                // exact mode keeps the reference's two separate force sums
  #pragma unroll
        for (int c = 0; c < 3; ++c) { wr[c] = sFr[c]; wr[3 + c] = sFf[c]; wr[6 + c] = sTau[c]; }
      }
    };
    // ... and from the summed wrench to the accelerations: wd = clamp(I^-1 tau), xdd = (m g ghat + sum F) / m
    auto accelerations = [&](const S (&wr)[kWr], S (&xdd)[3], S (&wd)[3], S (&wraw)[3]) {
      const S* sTau = wr + (FAST ? 3 : 6);
      // omega_d = clamp(I^-1 tau) (body-frame I with world-frame torque, as the reference)   (:256-257)
  #pragma unroll
      for (int c = 0; c < 3; ++c) {
        wraw[c] = Iv[c * 3 + 0] * sTau[0] + Iv[c * 3 + 1] * sTau[1] + Iv[c * 3 + 2] * sTau[2];
        wd[c] = M::clamp(wraw[c], -a.omega_max, a.omega_max);
      }
      // xdd = (m g ghat + sum Fs + sum Ff) / m   (:264-266)
      if (FAST) { xdd[0] = wr[0] * a.inv_mass; xdd[1] = wr[1] * a.inv_mass; xdd[2] = (wr[2] - a.mg) * a.inv_mass; }
      else { xdd[0] = (wr[0] + wr[3]) / a.mass; xdd[1] = (wr[1] + wr[4]) / a.mass; xdd[2] = ((-a.mg + wr[2]) + wr[5]) / a.mass; }
    };
    // the force-dependent half of torchdiffeq's fixed-grid euler: y_{n+1} = y_n + (t_{n+1} - t_n) f(t_n, y_n), f = (xd, xdd, [w]x R, wd, Fs, Ff)
    auto advance_velocities = [&](const Con& q, const S (&xdd)[3], const S (&wd)[3], S h) {
  #pragma unroll
      for (int c = 0; c < 3; ++c) {
        xd[c] = xd[c] + h * xdd[c];
        w[c] = w[c] + h * wd[c];
      }
  #pragma unroll
      for (int j = 0; j < PPL; ++j)
  #pragma unroll
        for (int c = 0; c < 3; ++c) {
          oFs[j][c] = oFs[j][c] + h * q.Fr[j][c];
          oFf[j][c] = oFf[j][c] + h * q.Ff[j][c];
        }
    };

    auto request = [&](Geo& g) {      // the footprint's cells as they lie in memory; no dependent instruction here
      const Cell<S>& c = g.cell[0];
      const int p1 = min(c.ic, last - 1), p2 = min(c.i_f, last - 1);
      if constexpr (ZMU) {
        g.q1 = *reinterpret_cast<const CellQuad<S>*>(reinterpret_cast<const char*>(a.zmu) + (size_t)((unsigned)p1 * (unsigned)(2 * sizeof(S))));
        g.q2 = *reinterpret_cast<const CellQuad<S>*>(reinterpret_cast<const char*>(a.zmu) + (size_t)((unsigned)p2 * (unsigned)(2 * sizeof(S))));
      } else {
        const CellPair<S> z1 = *reinterpret_cast<const CellPair<S>*>(reinterpret_cast<const char*>(zmap) + (size_t)((moff + (unsigned)p1) * (unsigned)sizeof(S)));
        const CellPair<S> z2 = *reinterpret_cast<const CellPair<S>*>(reinterpret_cast<const char*>(zmap) + (size_t)((moff + (unsigned)p2) * (unsigned)sizeof(S)));
        const CellPair<S> m1 = *reinterpret_cast<const CellPair<S>*>(reinterpret_cast<const char*>(mumap) + (size_t)((moff + (unsigned)p1) * (unsigned)sizeof(S)));
        const CellPair<S> m2 = *reinterpret_cast<const CellPair<S>*>(reinterpret_cast<const char*>(mumap) + (size_t)((moff + (unsigned)p2) * (unsigned)sizeof(S)));
        g.q1 = CellQuad<S>{z1.a, z1.b, z2.a, z2.b};      // (za, ma, zb, mb) reused as (z1.a, z1.b, z2.a, z2.b)
        g.q2 = CellQuad<S>{m1.a, m1.b, m2.a, m2.b};
      }
    };
    auto arrive = [&](Geo& g) {       // gather4 / gather4x2's choice among the loaded cells (the reference clamps the FLAT index)
      const Cell<S>& c = g.cell[0];
      const int p1 = min(c.ic, last - 1), p2 = min(c.i_f, last - 1);
      const bool ec = c.ic != p1, el = c.il != p1, ef = c.i_f != p2, efl = c.ifl != p2;
      if constexpr (ZMU) {
        g.zc[0][0] = ec ? g.q1.zb : g.q1.za;   g.mc[0][0] = ec ? g.q1.mb : g.q1.ma;
        g.zc[0][2] = el ? g.q1.zb : g.q1.za;   g.mc[0][2] = el ? g.q1.mb : g.q1.ma;
        g.zc[0][1] = ef ? g.q2.zb : g.q2.za;   g.mc[0][1] = ef ? g.q2.mb : g.q2.ma;
        g.zc[0][3] = efl ? g.q2.zb : g.q2.za;  g.mc[0][3] = efl ? g.q2.mb : g.q2.ma;
      } else {
        g.zc[0][0] = ec ? g.q1.ma : g.q1.za;   g.mc[0][0] = ec ? g.q2.ma : g.q2.za;
        g.zc[0][2] = el ? g.q1.ma : g.q1.za;   g.mc[0][2] = el ? g.q2.ma : g.q2.za;
        g.zc[0][1] = ef ? g.q1.mb : g.q1.zb;   g.mc[0][1] = ef ? g.q2.mb : g.q2.zb;
        g.zc[0][3] = efl ? g.q1.mb : g.q1.zb;  g.mc[0][3] = efl ? g.q2.mb : g.q2.zb;
      }
    };
    const S* ts_pair = a.ts;
    // The height-dependent half of the contact model: terrain sample, normal, friction, penetration and the contact weight of a
    // point under a POSE -- no velocity in it, so it is evaluated a step ahead (pose n + 1 is known when step n starts) and the
    // contact count of step n + 1 rides in the SAME workgroup exchange as the wrench of step n: one LDS round trip and one
    // barrier per step instead of two.
    struct Hgt { S nrm[3], muq, dh, cj; };
    auto height_part = [&](const Geo& g, Hgt& hq) {
      const Cell<S>& c = g.cell[0];
      S zq, mub;                                                   // height, normal, friction under the point (:211-216)
      blend2(c, g.zc[0], g.mc[0], &zq, &mub);
      hq.muq = has_mu ? mub : blend_ones(c);
      const S gx = M::div(g.zc[0][1] - g.zc[0][0], a.res), gy = M::div(g.zc[0][2] - g.zc[0][0], a.res);
      const S inl = M::inv_len(gx * gx + gy * gy + one);
      hq.nrm[0] = -gx * inl; hq.nrm[1] = -gy * inl; hq.nrm[2] = inl;
      hq.dh = g.pz[0] - zq;                                        // soft contact (:220-222)
      const S cj = M::sigmoid_m10(hq.dh);
      hq.cj = act[0] ? cj : zero;
    };
    S csum_cur = zero;      // contact count of the step about to run (total over the workgroup)
    static_assert(kWr == 6, "the pipelined step exchanges the fast-math wrench (6) and one contact count");
    __shared__ __attribute__((aligned(4 * sizeof(S)))) S xch_lds[G > 64 ? TransposedExchange<(G > 64 ? G / 64 : 1), S>::kWords : 4];
    TransposedExchange<(G > 64 ? G / 64 : 1), S> xch;
    xch.lds = xch_lds;
    auto pipe_step = [&](int n, Geo& g, Hgt& hq, Geo& g_next, Hgt& hq_next) {
      // next step's controls and step size: requested before this step's stores (vmcnt retires in order)
      const int nn = min(n + 1, a.T - 1);
      const S cv_next = ctrl[nn * a.ctrl_st + 0], cw_next = ctrl[nn * a.ctrl_st + 1];
      const int tp = max(min(n + 1, a.T - 2), 0);
      const S ts_a = ts_pair[tp], ts_b = ts_pair[tp + 1];
      emit_row(row_stride);                       // row n: the state as it stands
      Kin k;
      kinematics(g, k);
      advance_pose(h_ode);                        // pose n + 1 (the explicit scheme moves x with the OLD xd, R with the OLD w)
      locate_points(g_next);
      request(g_next);                            // its cells: in flight under the contact chain of step n
      Con q;                                      // spring-damper along the normal (:224-230), velocities of step n
      {
        const S vn = k.vp[0][0] * hq.nrm[0] + k.vp[0][1] * hq.nrm[1] + k.vp[0][2] * hq.nrm[2];
        const S A = a.k * hq.dh + a.damp * vn;
        q.nrm[0][0] = hq.nrm[0]; q.nrm[0][1] = hq.nrm[1]; q.nrm[0][2] = hq.nrm[2];
        q.muq[0] = hq.muq; q.cw8[0] = hq.cj;
        q.Fr[0][0] = -(A * hq.nrm[0]); q.Fr[0][1] = -(A * hq.nrm[1]); q.Fr[0][2] = -(A * hq.nrm[2]);
      }
      S wr[kWr];
      contact_wrench(g, k, q, csum_cur, wr);
      arrive(g_next);
      height_part(g_next, hq_next);
      S ex[kWr + 1];
  #pragma unroll
      for (int c = 0; c < kWr; ++c) ex[c] = wr[c];
      ex[kWr] = hq_next.cj;
      // wrench of step n + contact count of step n + 1: one transposed workgroup exchange (mf_common.h)
      if constexpr (G > 64) {
        const S v8[8] = {ex[0], ex[1], ex[2], ex[3], ex[4], ex[5], ex[6], zero};
        xch.post(v8, zero);
        S tot[7];
        xch.template wait<7>(tot);
  #pragma unroll
        for (int c = 0; c < 7; ++c) ex[c] = tot[c];
      } else {      // a rollout inside a wave: seven DPP group sums
        gs.template sum_n<kWr + 1>(ex);
      }
  #pragma unroll
      for (int c = 0; c < kWr; ++c) wr[c] = ex[c];
      S xdd[3], wd[3], wraw[3];
      accelerations(wr, xdd, wd, wraw);
      if constexpr (REC) {
        // the multi-wave backward's record (rollout_bwd_mw_kernel.h): the contact count and the unclamped angular acceleration
        // this step evaluated, 16 bytes per rollout-step; every lane stores the same quad to the same address (no branch: the
        // arithmetic of the step stays one basic block, so the trajectory's bits are those of the kernel without a record)
        typedef S f4v __attribute__((ext_vector_type(4)));
        const f4v rv = {csum_cur, wraw[0], wraw[1], wraw[2]};
        __builtin_nontemporal_store(rv, reinterpret_cast<f4v*>(a.rec + ((size_t)n * a.B + b) * 4));
      }
      advance_velocities(q, xdd, wd, h_ode);
      csum_cur = ex[kWr];
      cv = cv_next; cw = cw_next;
      h_ode = ts_b - ts_a;
    };
    Geo gA, gB;
    Hgt hA, hB;
    if (n_steps > 0) {      // prologue: cells and contact count of step 0
      locate_points(gA); request(gA);
      __builtin_amdgcn_s_waitcnt(0);
      arrive(gA);
      height_part(gA, hA);
      csum_cur = gs.sum(hA.cj);
    }
    __builtin_amdgcn_s_waitcnt(0);
    int n = 0;
    for (; n + 1 < n_steps; n += 2) { pipe_step(n, gA, hA, gB, hB); pipe_step(n + 1, gB, hB, gA, hA); }
    if (n < n_steps) pipe_step(n, gA, hA, gB, hB);
  } else {
  __shared__ __attribute__((aligned(4 * sizeof(S)))) S xg_lds[(FAST && G > 64) ? TransposedExchange<(G > 64 ? G / 64 : 1), S>::kWords : 4];
  TransposedExchange<(G > 64 ? G / 64 : 1), S> xch_g;
  xch_g.lds = xg_lds;
  for (int n = 0; n < n_steps; ++n) {
    if (JOINTS) articulate_body<S, G, PPL, FAST>(gs, a.joint_angles + ((size_t)b * a.T + n) * 4, a.joint_xyz, a.mass / (S)a.N, P0, part, act, P, Iv);
    // ---- geometry of the contact points and the gathers that depend only on it ----
    S r[PPL][3], pz[PPL];
    Cell<S> cell[PPL];
    S zc[PPL][4], mc[PPL][4];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      // p = P R^T + x ; r = p - x   (:200)
      S px = P[j][0] * R[0] + P[j][1] * R[1] + P[j][2] * R[2] + x[0];
      S py = P[j][0] * R[3] + P[j][1] * R[4] + P[j][2] * R[5] + x[1];
      pz[j] = P[j][0] * R[6] + P[j][1] * R[7] + P[j][2] * R[8] + x[2];
      r[j][0] = px - x[0]; r[j][1] = py - x[1]; r[j][2] = pz[j] - x[2];
      cell[j] = locate_m<S, FAST>(px, py, a.d_max, a.res, a.inv_res, a.H, last);
      const Cell<S>& c = cell[j];
      if constexpr (ZMU) gather4x2(a.zmu, c, last, zc[j], mc[j]);
      else gather4(zmap, moff, c, last, zc[j]);
    }
#pragma unroll
    for (int j = 0; j < PPL; ++j) {   // unconditional (mumap aliases z without a friction map; selected after the blend): no branch
      const Cell<S>& c = cell[j];
      if constexpr (!ZMU) gather4(mumap, moff, c, last, mc[j]);
    }
    // next step's controls (the lookup argmin|t - ts| is the step index on the grid, dphysics.py:183)
    const int nn = min(n + 1, a.T - 1);
    const S cv_next = ctrl[nn * a.ctrl_st + 0], cw_next = ctrl[nn * a.ctrl_st + 1];
    // ... and its step size h = ts[n+2] - ts[n+1] (torchdiffeq's fixed grid): loaded HERE, before the stores below -- a load
    // issued after them would make its wait (vmcnt is in-order) a wait for this step's stores as well
    S ts_a = zero, ts_b = zero;
    if (INTEG == MF_INTEG_ODEINT_EULER) {
      // one 8-byte load of the adjacent pair (ts[n+1], ts[n+2]); in the last iteration, whose h is never used, the pair is
      // clamped into the grid
      const int tp = max(min(n + 1, a.T - 2), 0);
      // (a vector load on purpose: as a scalar load through the constant address space it shares lgkmcnt with the LDS traffic
      // of the split-store kernels and every LDS wait becomes a wait for it -- measured slower at 16 k rollouts)
      ts_a = a.ts[tp]; ts_b = a.ts[tp + 1];        // T >= 2 inside the loop, so tp + 1 <= T - 1
    }

    // ---- stores of the previous step's row: younger than the gathers above ----
    // DYNAMICS has nothing pending at n = 0: it writes the initial state into row 0 without advancing, and the real row 0
    // overwrites it one iteration later (same lane, same addresses, program order) -- keeps the loop one basic block
    emit_row((INTEG == MF_INTEG_ODEINT_EULER || n > 0) ? row_stride : 0);

    // ---- work that does not need the gathered cells ----
    S vp[PPL][3];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {  // v_p = xd + w x r   (:204)
      vp[j][0] = xd[0] + (w[1] * r[j][2] - w[2] * r[j][1]);
      vp[j][1] = xd[1] + (w[2] * r[j][0] - w[0] * r[j][2]);
      vp[j][2] = xd[2] + (w[0] * r[j][1] - w[1] * r[j][0]);
    }
    // thrust direction = normalized first column of R   (:237)
    const S il = M::inv_len(R[0] * R[0] + R[3] * R[3] + R[6] * R[6]);
    S e0, e1, e2;
    if (M::kReciprocalNorm) { e0 = R[0] * il; e1 = R[3] * il; e2 = R[6] * il; }
    else { const S el = mf_max(M::sqrt(R[0] * R[0] + R[3] * R[3] + R[6] * R[6]), (S)1e-6); e0 = R[0] / el; e1 = R[3] / el; e2 = R[6] / el; }
    const S tv_lo = cv - cw * a.half_ly, tv_hi = cv + cw * a.half_ly;  // (:75-104)
    // The explicit scheme moves x with the OLD xd and R with the OLD w: neither depends on this step's forces, so that half
    // of the update is done here, under the latency of the gathers (x, R are not read again below; r, pz, e are taken).
    if (INTEG == MF_INTEG_ODEINT_EULER) {
      S dR[9];
#pragma unroll
      for (int j2 = 0; j2 < 3; ++j2) {
        dR[0 * 3 + j2] = w[1] * R[2 * 3 + j2] - w[2] * R[1 * 3 + j2];
        dR[1 * 3 + j2] = w[2] * R[0 * 3 + j2] - w[0] * R[2 * 3 + j2];
        dR[2 * 3 + j2] = w[0] * R[1 * 3 + j2] - w[1] * R[0 * 3 + j2];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) x[c] = x[c] + h_ode * xd[c];
#pragma unroll
      for (int c = 0; c < 9; ++c) R[c] = R[c] + h_ode * dR[c];
    }

    // ---- contact model ----
    S nrm[PPL][3], muq[PPL], cw8[PPL], Fr[PPL][3];
    S csum = zero;
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      const Cell<S>& c = cell[j];
      S zq, mub;                                                 // height, normal, friction under the point (:211-216)
      blend2(c, zc[j], mc[j], &zq, &mub);
      // no friction map = a map of ones (dphysics.py:562): its blend is the (rounded) sum of the four weights, no loads involved
      muq[j] = has_mu ? mub : blend_ones(c);
      S gx = M::div(zc[j][1] - zc[j][0], a.res), gy = M::div(zc[j][2] - zc[j][0], a.res);
      if (M::kReciprocalNorm) {
        const S inl = M::inv_len(gx * gx + gy * gy + one);
        nrm[j][0] = -gx * inl; nrm[j][1] = -gy * inl; nrm[j][2] = inl;
      } else {
        const S nl = mf_max(M::sqrt(gx * gx + gy * gy + one), (S)1e-6);
        nrm[j][0] = -gx / nl; nrm[j][1] = -gy / nl; nrm[j][2] = one / nl;
      }
      S dh = pz[j] - zq;  // soft contact + spring-damper along the normal   (:220-230)
      S cj = M::sigmoid_m10(dh);
      cj = act[j] ? cj : zero;
      cw8[j] = cj;
      csum += cj;
      S vn = vp[j][0] * nrm[j][0] + vp[j][1] * nrm[j][1] + vp[j][2] * nrm[j][2];
      S A = a.k * dh + a.damp * vn;
      Fr[j][0] = -(A * nrm[j][0]); Fr[j][1] = -(A * nrm[j][1]); Fr[j][2] = -(A * nrm[j][2]);
    }
    csum = gs.sum(csum);  // n_contact_pts (:231)
    const S inv_csum = FAST ? M::div(one, csum) : one;

    S sFr[3] = {zero, zero, zero}, sFf[3] = {zero, zero, zero}, sTau[3] = {zero, zero, zero};
    S Ff[PPL][3];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {  // (:232-233)
        S f = FAST ? Fr[j][c] * cw8[j] * inv_csum : Fr[j][c] * cw8[j] / csum;
        Fr[j][c] = M::clamp(f, -a.mg, a.mg);
      }
      S Nn = M::sqrt(Fr[j][0] * Fr[j][0] + Fr[j][1] * Fr[j][1] + Fr[j][2] * Fr[j][2]);  // (:238)
      S tv = (part[j] < 0) ? zero : ((part[j] & 1) ? tv_hi : tv_lo);
      S s0 = muq[j] * (tv * e0 - vp[j][0]);  // slip (:247); cmd = 0 for non-driving points
      S s1 = muq[j] * (tv * e1 - vp[j][1]);
      S s2 = muq[j] * (tv * e2 - vp[j][2]);
      S sn = s0 * nrm[j][0] + s1 * nrm[j][1] + s2 * nrm[j][2];
      Ff[j][0] = M::clamp(Nn * (s0 - sn * nrm[j][0]), -a.mg, a.mg);  // (:248-251)
      Ff[j][1] = M::clamp(Nn * (s1 - sn * nrm[j][1]), -a.mg, a.mg);
      Ff[j][2] = M::clamp(Nn * (s2 - sn * nrm[j][2]), -a.mg, a.mg);
      if (!act[j]) {
#pragma unroll
        for (int c = 0; c < 3; ++c) Fr[j][c] = Ff[j][c] = zero;
      }
      S f0 = Fr[j][0] + Ff[j][0], f1 = Fr[j][1] + Ff[j][1], f2 = Fr[j][2] + Ff[j][2];
      sTau[0] += r[j][1] * f2 - r[j][2] * f1;  // r x (Fs + Ff)   (:255)
      sTau[1] += r[j][2] * f0 - r[j][0] * f2;
      sTau[2] += r[j][0] * f1 - r[j][1] * f0;
      if (FAST) { sFr[0] += f0; sFr[1] += f1; sFr[2] += f2; }
      else {
#pragma unroll
        for (int c = 0; c < 3; ++c) { sFr[c] += Fr[j][c]; sFf[c] += Ff[j][c]; }
      }
    }
    if (FAST) {   // one batched reduction of the wrench (multi-wave groups: one LDS exchange)
      S wr[6] = {sFr[0], sFr[1], sFr[2], sTau[0], sTau[1], sTau[2]};
      if constexpr (FAST && G > 64) {      // ... the transposed one (mf_common.h): about half the instructions of six plain workgroup sums
        const S v8[8] = {wr[0], wr[1], wr[2], wr[3], wr[4], wr[5], zero, zero};
        xch_g.post(v8, zero);
        S tot[6];
        xch_g.template wait<6>(tot);
#pragma unroll
        for (int c = 0; c < 6; ++c) wr[c] = tot[c];
      } else {
        gs.sum_n(wr);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) { sFr[c] = wr[c]; sTau[c] = wr[3 + c]; }
    } else {      // exact mode keeps the reference's two separate force sums
      S wr[9] = {sFr[0], sFr[1], sFr[2], sFf[0], sFf[1], sFf[2], sTau[0], sTau[1], sTau[2]};
      gs.sum_n(wr);
#pragma unroll
      for (int c = 0; c < 3; ++c) { sFr[c] = wr[c]; sFf[c] = wr[3 + c]; sTau[c] = wr[6 + c]; }
    }
    // omega_d = clamp(I^-1 tau) (body-frame I with world-frame torque, as the reference)   (:256-257)
    S wd[3], wraw[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      wraw[c] = Iv[c * 3 + 0] * sTau[0] + Iv[c * 3 + 1] * sTau[1] + Iv[c * 3 + 2] * sTau[2];
      wd[c] = M::clamp(wraw[c], -a.omega_max, a.omega_max);
    }
    if constexpr (REC) {   // the record of rollout_bwd_mw_kernel.h: every lane of the group stores the same quad (no branch)
      typedef S f4v __attribute__((ext_vector_type(4)));
      const f4v rv = {csum, wraw[0], wraw[1], wraw[2]};
      __builtin_nontemporal_store(rv, reinterpret_cast<f4v*>(a.rec + ((size_t)n * a.B + b) * 4));
    }
    // xdd = (m g ghat + sum Fs + sum Ff) / m   (:264-266)
    S xdd[3];
    if (FAST) { xdd[0] = sFr[0] * a.inv_mass; xdd[1] = sFr[1] * a.inv_mass; xdd[2] = (sFr[2] - a.mg) * a.inv_mass; }
    else { xdd[0] = (sFr[0] + sFf[0]) / a.mass; xdd[1] = (sFr[1] + sFf[1]) / a.mass; xdd[2] = ((-a.mg + sFr[2]) + sFf[2]) / a.mass; }

    if (INTEG == MF_INTEG_DYNAMICS) {
      // update_state (:274-288): xd += xdd h ; x += xd_new h ; w += wd h ; R <- R (I + K sin + K^2 (1 - cos))
      const S h = a.dt;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        xd[c] = xd[c] + xdd[c] * h;
        x[c] = x[c] + xd[c] * h;
        w[c] = w[c] + wd[c] * h;
      }
      const S th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
      const S th = M::sqrt(th2);
      S k0, k1, k2;  // K = [w]x / max(|w|, eps)
      if (M::kReciprocalNorm) { const S id = M::inv_len(th2); k0 = w[0] * id; k1 = w[1] * id; k2 = w[2] * id; }
      else { const S den = mf_max(th, (S)1e-6); k0 = w[0] / den; k1 = w[1] / den; k2 = w[2] / den; }
      S sn, oc;
      M::sincos_small(th * h, &sn, &oc);
      // K = [[0,-k2,k1],[k2,0,-k0],[-k1,k0,0]];  M = I + K sin + (K K) (1 - cos)
      S K[9] = {zero, -k2, k1, k2, zero, -k0, -k1, k0, zero};
      S Mx[9];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j2 = 0; j2 < 3; ++j2) {
          S kk = K[i * 3 + 0] * K[0 * 3 + j2] + K[i * 3 + 1] * K[1 * 3 + j2] + K[i * 3 + 2] * K[2 * 3 + j2];
          Mx[i * 3 + j2] = ((i == j2 ? one : zero) + K[i * 3 + j2] * sn) + kk * oc;
        }
      S Rn[9];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j2 = 0; j2 < 3; ++j2)
          Rn[i * 3 + j2] = R[i * 3 + 0] * Mx[0 * 3 + j2] + R[i * 3 + 1] * Mx[1 * 3 + j2] + R[i * 3 + 2] * Mx[2 * 3 + j2];
#pragma unroll
      for (int c = 0; c < 9; ++c) R[c] = Rn[c];
#pragma unroll
      for (int j = 0; j < PPL; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) { oFs[j][c] = Fr[j][c]; oFf[j][c] = Ff[j][c]; }  // true forces of this step
    } else {
      // torchdiffeq fixed-grid euler: y_{n+1} = y_n + (t_{n+1} - t_n) f(t_n, y_n), f = (xd, xdd, [w]x R, wd, Fs, Ff)
      const S h = h_ode;          // x and R were advanced above; the force-dependent half follows
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        xd[c] = xd[c] + h * xdd[c];
        w[c] = w[c] + h * wd[c];
      }
#pragma unroll
      for (int j = 0; j < PPL; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          oFs[j][c] = oFs[j][c] + h * Fr[j][c];
          oFf[j][c] = oFf[j][c] + h * Ff[j][c];
        }
    }
    cv = cv_next; cw = cw_next;
    h_ode = ts_b - ts_a;
    if (COST) {   // the state is now output row m = n + 1 (ODEINT) / n (DYNAMICS)
      const bool due = pose_wait == 0;
      if (due || n == n_steps - 1) store_pose();
      pXs += due ? row_stride * 3 : 0; pRs += due ? row_stride * 9 : 0;
      pose_wait = due ? a.pose_stride - 1 : pose_wait - 1;
    }
  }
  }
  if (n_steps > 0 || INTEG == MF_INTEG_ODEINT_EULER) emit_row(row_stride);
  // the force path cost itself: norm(F_springs).std(points).std(time) (monoforce_node.py:91), unbiased like torch.std
  if (COST && a.path_cost != nullptr && gl == 0) a.path_cost[b] = M::sqrt(pc_m2 / (pc_n - one));
}

// Lane mapping for (B, N): G lanes per rollout x PPL points per lane (see the header comment).
struct LaneMap { int G, PPL; };
static inline LaneMap choose_lane_map(int B, int N, int points_per_lane) {
  int g1 = 4;
  while (g1 < N) g1 <<= 1;  // lanes per rollout at one point per lane
  // One point per lane whenever the body fits a wave: measured faster than 4 points per lane over the whole range
  // B = 256 .. 65536 (N = 4) once the fast-math kernels cut the per-lane instruction count (tools/sweep_mapping.py).
  bool wide = g1 <= 64;
  if (points_per_lane == 1 && g1 <= 64) wide = true;
  if (points_per_lane == 4) wide = false;
  if (N <= 4) return wide ? LaneMap{4, 1} : LaneMap{1, 4};
  if (N <= 8) return wide ? LaneMap{8, 1} : LaneMap{2, 4};
  if (N <= 16) return wide ? LaneMap{16, 1} : LaneMap{4, 4};
  if (N <= 32) return wide ? LaneMap{32, 1} : LaneMap{8, 4};
  if (N <= 64) return wide ? LaneMap{64, 1} : LaneMap{16, 4};
  // Larger bodies: one wave per rollout with 2 / 4 / 8 points per lane -- unless the batch is so small that this leaves
  // most of the chip idle (the reference's own use: 4 .. 64 rollouts of a 175- or 223-point robot).  Then ONE rollout is
  // spread over 2, 4 or 8 waves of a workgroup, one point per lane (GroupSum exchanges through LDS): ~2.3x fewer instructions
  // per wave and step.  Measured at N = 223: forward 0.79 vs 1.85 ms, backward 2.0 vs 5.6 ms for B <= 256; 0.98 / 2.9 vs
  // 1.38 / 4.4 ms at B = 512 (2 waves per SIMD); a tie at B = 1024 -- so up to 2048 waves per launch.
  if (points_per_lane != 4) {
    const int g = N <= 128 ? 128 : (N <= 256 ? 256 : 512);
    if ((long long)B * (g / 64) <= 2 * device_simds()) return LaneMap{g, 1};      // two waves per SIMD (MI355X: 2048)
  }
  if (N <= 128) return points_per_lane != 4 ? LaneMap{64, 2} : LaneMap{32, 4};
  if (N <= 256) return LaneMap{64, 4};
  return LaneMap{64, 8};
}

// Instantiated mappings: one point per lane (G = 4..64) and (64, 2/4/8) always; the 4-points-per-lane mappings with G < 64
// only for the full-output rigid-body kernels (they are a tuning / test option, see choose_lane_map).
// Round 6: the controls of a saturated launch read ONCE in front of it.  A step loads the next step's (v, w) one step ahead (~0.7 us at
// 16 384 rollouts) inside its dependent chain, and vmcnt retires loads in order: while the [B][T][2] array sits in the memory-side cache
// (forward after forward) that is free, but the forward of a fit / train step follows a backward that has streamed ~700 MB through the
// caches, and every 64-byte line of controls then costs an HBM round trip in front of the step's gathers -- forward 0.34 -> 0.43 ms at
// 16 384 rollouts, 0.35 again with this 16 us pass (tools/ab_step_fwd2.py, profiles/r6_ab_step_fwd.txt).  A deeper in-kernel prefetch does
// not help: a load that misses stalls the younger gathers behind it wherever it is issued.  MF_FWD_TOUCH_CONTROLS=0: A/B.
template <int UNUSED = 0>      // (a template: one definition across the translation units that include this header)
__global__ void __launch_bounds__(256) touch_lines_kernel(const float4* __restrict__ p, long long n16, float4* __restrict__ sink) {
  float4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) {
    const float4 v = p[i];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (acc.x == 1.2345e-30f && acc.y == 5.4321e-30f) *sink = acc;      // (never: keeps the loads)
}
template <typename S>
static inline void touch_controls(const RolloutArgs<S>& a, const LaneMap& m, int b0, int nb, hipStream_t st) {
  static const bool off = getenv("MF_FWD_TOUCH_CONTROLS") && atoi(getenv("MF_FWD_TOUCH_CONTROLS")) == 0;
  if (off || a.ctrl_st == 0 || a.ctrl_sb != a.T * 2 || m.G > 64) return;      // (one pair per rollout, or rows that are not adjacent: nothing to stream)
  if ((long long)nb * m.G < device_simds() / 2 * 64) return;                   // (below half a wave per SIMD a step is longer than the round trip)
  const long long bytes = (long long)nb * a.T * 2 * (long long)sizeof(S);
  if (bytes < (8ll << 20) || bytes > (192ll << 20)) return;                    // (a few MB stay resident anyway; more than the cache holds is futile)
  const char* base = reinterpret_cast<const char*>(a.controls + (size_t)b0 * a.ctrl_sb);
  const char* al = reinterpret_cast<const char*>(((uintptr_t)base + 15) & ~(uintptr_t)15);
  const long long n16 = (bytes - (al - base)) / 16;
  hipLaunchKernelGGL((touch_lines_kernel<0>), dim3(2048), dim3(256), 0, st, reinterpret_cast<const float4*>(al), n16, reinterpret_cast<float4*>(const_cast<char*>(al)));
}

template <typename S, bool FAST, bool JOINTS = false, bool FORCES = true, int COST = 0, bool SPLIT = false, bool ZMU = false>
int launch_rollout_fwd(const RolloutArgs<S>& a, LaneMap m, int integ, int block, hipStream_t st) {
  if (m.G > 64) block = m.G;   // a rollout spread over several waves: exactly one rollout per workgroup (LDS + barrier)
  // More than two waves per SIMD do not help these kernels -- their gathers then miss the CU's L1 more often -- so a very
  // large batch goes out as consecutive launches of <= kChunkWaves waves on the same stream (measured: B = 65536 1.81 -> 1.73 ms,
  // B = 131072 4.23 -> 3.90 ms; MF_CHUNK_WAVES=0 disables).  The chunk is a whole number of workgroups; results do not depend on it.
  static const long long kChunkEnv = getenv("MF_CHUNK_WAVES") ? atoll(getenv("MF_CHUNK_WAVES")) : -1;
  const long long kChunkWaves = kChunkEnv >= 0 ? kChunkEnv : 2 * device_simds();      // two waves per SIMD (MI355X: 2048)
  int chunk_B = a.B;
  if (m.G <= 64 && kChunkWaves > 0 && (long long)a.B * m.G > kChunkWaves * 64) chunk_B = (int)(kChunkWaves * 64 / m.G);
  bool launched = false;
  for (int b0 = 0; b0 < a.B; b0 += chunk_B) {
  RolloutArgs<S> ac = a;
  ac.b0 = b0;
  const long long threads = (long long)((a.B - b0 < chunk_B) ? a.B - b0 : chunk_B) * m.G;
  const unsigned grid = (unsigned)((threads + block - 1) / block);
  touch_controls(a, m, b0, (a.B - b0 < chunk_B) ? a.B - b0 : chunk_B, st);
  launched = false;
#define MF_CASE(G_, P_)                                                                                                                   \
  if (!launched && m.G == G_ && m.PPL == P_) {                                                                                             \
    launched = true;                                                                                                                       \
    if (integ == MF_INTEG_DYNAMICS)                                                                                                        \
      MF_KLAUNCH((rollout_fwd_kernel<S, G_, P_, MF_INTEG_DYNAMICS, FAST, JOINTS, FORCES, COST, SPLIT, ZMU>), dim3(grid), dim3(block), 0, st, ac);     \
    else                                                                                                                                   \
      MF_KLAUNCH((rollout_fwd_kernel<S, G_, P_, MF_INTEG_ODEINT_EULER, FAST, JOINTS, FORCES, COST, SPLIT, ZMU>), dim3(grid), dim3(block), 0, st, ac); \
  }
  if (!JOINTS) { MF_CASE(4, 1) MF_CASE(8, 1) MF_CASE(16, 1) MF_CASE(32, 1) MF_CASE(64, 1) }
  if constexpr (!SPLIT && !ZMU) {   // SPLIT / ZMU kernels: the one-point-per-lane mappings up to a wave only
  if (!JOINTS) { MF_CASE(64, 2) }
  MF_CASE(128, 1) MF_CASE(256, 1) MF_CASE(512, 1)
  if (FORCES) { MF_CASE(1, 4) MF_CASE(2, 4) MF_CASE(4, 4) MF_CASE(8, 4) MF_CASE(16, 4) MF_CASE(32, 4) }
  MF_CASE(64, 4) MF_CASE(64, 8)
  }
#undef MF_CASE
  MF_REQUIRE(launched, MF_ERR_UNSUPPORTED, "rollout_fwd: no kernel for this lane mapping");
  }
  hipError_t e = hipGetLastError();
  MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("rollout_fwd launch: ") + hipGetErrorString(e));
  return MF_OK;
}

// The one-point-per-lane mappings (both integrators) with the record for their backward (rollout_bwd_mw_kernel.h):
// a.rec != NULL.  Bodies of 5..64 points (several rollouts per wave; plain or interleaved maps) and of 65..512 (one per workgroup).
template <bool FORCES, bool ZMU = false, bool SPLIT = false, typename S = float>
int launch_rollout_fwd_mw_rec(const RolloutArgs<S>& a, LaneMap m, int integ, hipStream_t st) {
  bool launched = false;
#define MF_CASE(G_)                                                                                                          \
  if (!launched && m.G == G_ && m.PPL == 1) {                                                                                \
    launched = true;                                                                                                         \
    const int blk = G_ > 64 ? G_ : 64;                                                                                       \
    const unsigned grid = (unsigned)(((long long)a.B * G_ + blk - 1) / blk);                                                 \
    if (integ == MF_INTEG_DYNAMICS)                                                                                          \
      MF_KLAUNCH((rollout_fwd_kernel<S, G_, 1, MF_INTEG_DYNAMICS, true, false, FORCES, 0, SPLIT, ZMU, true>),    \
                         dim3(grid), dim3(blk), 0, st, a);                                                                   \
    else                                                                                                                     \
      MF_KLAUNCH((rollout_fwd_kernel<S, G_, 1, MF_INTEG_ODEINT_EULER, true, false, FORCES, 0, SPLIT, ZMU, true>), \
                         dim3(grid), dim3(blk), 0, st, a);                                                                   \
  }
  MF_CASE(8) MF_CASE(16) MF_CASE(32) MF_CASE(64)
  if constexpr (!ZMU && !SPLIT) { MF_CASE(128) MF_CASE(256) MF_CASE(512) }
#undef MF_CASE
  MF_REQUIRE(launched, MF_ERR_UNSUPPORTED, "rollout_fwd: no recording kernel for this lane mapping");
  hipError_t e = hipGetLastError();
  MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("rollout_fwd launch: ") + hipGetErrorString(e));
  return MF_OK;
}

// defined in rollout_fwd_fast.hip
int launch_rollout_fwd_fast_f32(const RolloutArgs<float>& a, LaneMap m, int integ, int block, bool forces, hipStream_t st);
// defined in rollout_fwd_split_fast.hip (state stores split over the lanes of a group; one-point-per-lane mappings up to a wave)
int launch_rollout_fwd_split_fast_f32(const RolloutArgs<float>& a, LaneMap m, int integ, int block, bool forces, hipStream_t st);
// defined in rollout_fwd_cost.hip
int launch_rollout_fwd_cost_f32(const RolloutArgs<float>& a, LaneMap m, int integ, int block, bool project, hipStream_t st);
// defined in rollout_fwd_zmu_fast.hip (shared maps interleaved as (z, mu); one-point-per-lane mappings up to a wave):
// cost = 0 full outputs (split = state stores spread over the lanes of a group), 1 / 2 cost rows
int launch_rollout_fwd_zmu_f32(const RolloutArgs<float>& a, LaneMap m, int integ, int block, bool forces, bool split, int cost, hipStream_t st);
// defined in rollout_fwd_joints_fast.hip
int launch_rollout_fwd_joints_fast_f32(const RolloutArgs<float>& a, LaneMap m, int integ, int block, hipStream_t st);

}  // namespace mf

// Fused DPhysics rollout, backward pass, COMPONENT-PARALLEL lane mapping (rollout_cp_common.h): the reverse-time adjoint of
// rollout_bwd_kernel.h with a rollout spread over a 16-lane row -- quad = contact point, lane c = component c of every vector /
// row c of R and of its adjoint, lane q = cell q of the bilinear footprint.  float32 fast math, rigid bodies of <= 4 contact
// points, the reference's default integrator (torchdiffeq fixed-grid Euler, dphysics.py:499-528); everything else runs on
// the one-point-per-lane kernels.  Same derivation, same autograd conventions (clamp passes gradient iff
// inside, `.long()` indices are constants, |v| has zero gradient at 0); sums run in a different order.
//
// Why: at the BASELINE shape (1024 rollouts x 4 points) the G = 4 kernel is 64 waves issuing ~840 instructions per step --
// the launch is bound by the instruction stream of one wave.  Here a step is ~1/3 of that per wave (3-vector algebra is
// one instruction per lane, the 18-component adjoint update is divided over the lanes instead of repeated by each), and a
// lane owns ONE footprint cell: its gradient accumulator is a single register pair flushed by one atomic per map.
#pragma once
#include "rollout_bwd_kernel.h"
#include "rollout_cp_common.h"

namespace mf {

struct __attribute__((aligned(4))) CpF3 { float a, b, c; };
__device__ __forceinline__ CpF3 ld3(const float* base, unsigned off) {
  return *reinterpret_cast<const CpF3*>(reinterpret_cast<const char*>(base) + (size_t)off);
}
__device__ __forceinline__ float ld1(const float* base, unsigned off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (size_t)off);
}

template <int INTEG>
__global__ void __launch_bounds__(256) rollout_bwd_cp_kernel(const RolloutBwdArgs<float> a) {
  static_assert(INTEG == MF_INTEG_ODEINT_EULER, "the component-parallel backward covers the default integrator");
  using namespace cp;
  using M = Mth<float, true>;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = tid >> 4;
  if (b >= a.B) return;
  const int p = (tid >> 2) & 3, q = tid & 3, cc = q < 3 ? q : 2;
  const float one = 1.0f, zero = 0.0f;
  const int HW = a.H * a.W, last = HW - 1;
  const unsigned moff = a.map_shared ? 0u : (unsigned)b * (unsigned)HW;
  const float* zmap = a.z;
  const bool has_mu = a.mu != nullptr;
  const float* mumap = has_mu ? a.mu : a.z;
  const unsigned goff = a.map_shared ? (unsigned)(b % a.grad_copies) * (unsigned)HW : (unsigned)b * (unsigned)HW;
  float* gzmap = a.gz;
  const bool want_gmu = a.gmu != nullptr && has_mu;
  float* gmumap = want_gmu ? a.gmu : a.gz;

  // ---- per-lane constants (as the forward, rollout_fwd_cp_kernel.h) ----
  const bool act = p < a.N;
  const int pi = act ? p : 0;
  const float P0 = a.points[pi * 3 + 0], P1 = a.points[pi * 3 + 1], P2 = a.points[pi * 3 + 2];
  const int part = act ? a.part[pi] : -1;
  const float tv_v = part < 0 ? zero : one;
  const float tv_w = part < 0 ? zero : ((part & 1) ? a.half_ly : -a.half_ly);
  const float I0 = a.Iinv[cc * 3 + 0], I1 = a.Iinv[cc * 3 + 1], I2 = a.Iinv[cc * 3 + 2];     // row cc of I^-1
  const float J0 = a.Iinv[0 * 3 + cc], J1 = a.Iinv[1 * 3 + cc], J2 = a.Iinv[2 * 3 + cc];     // column cc (I^-T)
  const int cell_off = ((q & 1) ? a.H : 0) + ((q & 2) ? 1 : 0);
  const float wa_s = (q & 2) ? one : -one, wa_o = (q & 2) ? zero : one;
  const float wb_s = (q & 1) ? one : -one, wb_o = (q & 1) ? zero : one;
  const float n_mul = q < 2 ? -a.inv_res : zero, n_add = q < 2 ? zero : one;
  // cell gradient of the normal's finite differences: cells (c, f, l, fl) get (-ggx - ggy, +ggx, +ggy, 0); ggx sits in lane 0,
  // ggy in lane 1, t = quad_perm[1,0,1,1](gg) brings the partner over: nz += sA * gg + sB * t
  const float sA = q == 0 ? -one : zero, sB = q == 0 ? -one : (q == 3 ? zero : one);
  const float sel_xy = q < 2 ? one : zero;       // components that receive d(sample)/d(position) through the fractions
  const float mg = a.mg;

  // ---- adjoint of the state: component cc / row cc ----
  float lx = zero, lxd = zero, lw = zero, lR0 = zero, lR1 = zero, lR2 = zero;
  float laFs = zero, laFf = zero;     // adjoint of this point's impulse accumulators (extended ODE state)

  // rows: per-lane byte offsets (loop-invariant) + wave-uniform row offsets stepped by scalar arithmetic
  const unsigned row_stride = (a.layout == MF_LAYOUT_TIME_MAJOR) ? (unsigned)a.B : 1u;
  const unsigned row0 = (a.layout == MF_LAYOUT_TIME_MAJOR) ? (unsigned)b : (unsigned)b * (unsigned)a.T;
  const unsigned v3 = (row0 * 3u + (unsigned)cc) * 4u, v9 = (row0 * 9u + (unsigned)cc * 3u) * 4u;
  const unsigned pcl = (unsigned)min(p, a.N - 1);      // inactive quads read a valid force row, masked at use
  // upstream rows: element strides 3 / 9 when present, 0 when the host substituted the zero row
  const unsigned u_xs = (row0 * (unsigned)a.sXs + (unsigned)cc) * 4u, u_xds = (row0 * (unsigned)a.sXds + (unsigned)cc) * 4u;
  const unsigned u_om = (row0 * (unsigned)a.sOm + (unsigned)cc) * 4u, u_r = (row0 * (unsigned)a.sRs + (unsigned)cc * 3u) * 4u;
  const unsigned u_fs = ((row0 * (unsigned)a.N + pcl) * (unsigned)a.sFs + (unsigned)cc) * 4u;
  const unsigned u_ff = ((row0 * (unsigned)a.N + pcl) * (unsigned)a.sFf + (unsigned)cc) * 4u;
  const float* ctrl = a.controls + (size_t)b * a.T * 2;
  float* gctrl = a.gcontrols + (size_t)b * a.T * 2;

  struct StateIn { float x, xd, w, R0, R1, R2, cv, cw, t0, t1; };
  struct UpIn { float gXs, gXds, gOm, gR0, gR1, gR2, gFs, gFf; };
  const int n_steps = a.T - 1;
  auto load_state = [&](int m, StateIn& s) {            // the state step m started from = saved output row m
    const size_t ro = (size_t)m * row_stride;            // wave-uniform
    const float* bx = a.Xraw + ro * 3; const float* bxd = a.Xds + ro * 3; const float* bw = a.Om + ro * 3; const float* bR = a.Rs + ro * 9;
    s.x = ld1(bx, v3); s.xd = ld1(bxd, v3); s.w = ld1(bw, v3);
    const CpF3 r = ld3(bR, v9);
    s.R0 = r.a; s.R1 = r.b; s.R2 = r.c;
    s.cv = ctrl[m * 2 + 0]; s.cw = ctrl[m * 2 + 1];
    s.t0 = a.ts[m]; s.t1 = a.ts[m + 1 < a.T ? m + 1 : m];
  };
  auto load_upstream = [&](int orow, UpIn& u) {         // upstream gradients of output row `orow`
    const size_t ro = (size_t)orow * row_stride;
    u.gXs = ld1(a.gXs + ro * a.sXs, u_xs); u.gXds = ld1(a.gXds + ro * a.sXds, u_xds); u.gOm = ld1(a.gOm + ro * a.sOm, u_om);
    const CpF3 r = ld3(a.gRs + ro * a.sRs, u_r);
    u.gR0 = r.a; u.gR1 = r.b; u.gR2 = r.c;
    u.gFs = ld1(a.gFs + ro * a.N * a.sFs, u_fs); u.gFf = ld1(a.gFf + ro * a.N * a.sFf, u_ff);
  };
  auto add_upstream_state = [&](const UpIn& u) {
    lx += u.gXs;
    lR2 = fmaf(u.gXs, a.sink, lR2);                     // Xs = x + R[:, 2] * sink
    lxd += u.gXds;
    lR0 += u.gR0; lR1 += u.gR1; lR2 += u.gR2;
    lw += u.gOm;
  };

  gctrl[(a.T - 1) * 2 + 0] = zero; gctrl[(a.T - 1) * 2 + 1] = zero;   // the last control is never used by the explicit scheme

  // Cell-gradient accumulator of this lane's footprint cell: contributions of consecutive steps to the SAME cell (a robot
  // moves <= 0.2 cell per step) add up in registers; when the lane's cell changes, the old pair goes to a stash that is
  // flushed with one atomic per map in the NEXT iteration, after that step's loads (vmcnt retires in order).
  unsigned acc_idx = 0u, st_idx = 0u;
  float acc_z = zero, acc_m = zero, st_z = zero, st_m = zero;
  bool st_pending = false;
  auto flush_stash = [&]() {
    if (st_pending) {
      atomic_add(at32(gzmap, goff + st_idx), st_z);
      if (want_gmu) atomic_add(at32(gmumap, goff + st_idx), st_m);
    }
    st_pending = false;
  };

  float* gctrl_pending = gctrl + (size_t)(a.T - 1) * 2;
  float gv_pending = zero, gwc_pending = zero;
  StateIn cur;
  UpIn up;
  load_state(max(n_steps - 1, 0), cur);
  load_upstream(max(n_steps - 1, 0) + 1 < a.T ? max(n_steps - 1, 0) + 1 : 0, up);
  __builtin_amdgcn_s_waitcnt(0);
  for (int n = n_steps - 1; n >= 0; --n) {
    add_upstream_state(up);
    const float x = cur.x, xd = cur.xd, w = cur.w, R0 = cur.R0, R1 = cur.R1, R2 = cur.R2;
    const float cv = cur.cv, cw = cur.cw;

    // ------------------------------------------------------------------------------------------------
    // forward recompute (the arithmetic of rollout_fwd_cp_kernel.h)
    // ------------------------------------------------------------------------------------------------
    const float r = P0 * R0 + P1 * R1 + P2 * R2;
    const float pc = r + x;
    const float lim = 262144.0f;
    const float uq = M::cell_coord(pc, a.d_max, a.res, a.inv_res);
    const int ui = (int)M::clamp(uq, -lim, lim);
    const float fr = uq - (float)ui;
    const int base = dppi<kB1>(ui) + __mul24(a.H, dppi<kB0>(ui));
    const int idx = min(max(base + cell_off, 0), last);
    const float zc = ld32(zmap, moff + (unsigned)idx);
    const float mc = ld32(mumap, moff + (unsigned)idx);
    // issue order (vmcnt is one in-order counter): gathers | atomics + control-gradient store deferred from the previous
    // step | prefetch of the next step's rows
    flush_stash();
    gctrl_pending[0] = gv_pending; gctrl_pending[1] = gwc_pending;      // every lane of the row: same address, same value
    StateIn nxt;
    UpIn up_next;
    load_state(max(n - 1, 0), nxt);
    load_upstream(n, up_next);          // output row n = the row step n - 1 produced (row 0 after the loop)

    const float wa = fmaf(wa_s, dpp<kB0>(fr), wa_o), wb = fmaf(wb_s, dpp<kB1>(fr), wb_o);
    const float wq = wa * wb;
    const float r1 = dpp<kRot1>(r), r2 = dpp<kRot2>(r);
    const float w1 = dpp<kRot1>(w), w2 = dpp<kRot2>(w);
    const float vp = xd + (w1 * r2 - w2 * r1);
    const float coln2 = dot3(R0, R0);
    const float il = M::inv_len(coln2);
    const float e = R0 * il;
    const float tv = tv_v * cv + tv_w * cw;

    const float zq = sum4(wq * zc);
    const float mcv = has_mu ? mc : one;
    const float mub = sum4(wq * mcv);
    const float dz = zc - dpp<kB0>(zc);
    const float u = fmaf(dpp<kN12>(dz), n_mul, n_add);
    const float inl = M::inv_len(dot3(u, u));
    const float nrm = u * inl;
    const float dh = dpp<kB2>(pc) - zq;
    float cj = M::sigmoid_m10(dh);
    cj = act ? cj : zero;
    const float csum = sum_points(cj);
    const float inv_csum = M::div(one, csum);
    const float vn = dot3(vp, nrm);
    const float A = a.k * dh + a.damp * vn;
    const float F0 = -(A * nrm);
    const float F1 = F0 * cj * inv_csum;
    const float Fr = M::clamp(F1, -mg, mg);
    const float Nn = M::sqrt(dot3(Fr, Fr));
    const float cmdv = tv * e - vp;
    const float s = mub * cmdv;
    const float sn = dot3(s, nrm);
    const float stv = s - sn * nrm;
    const float Gf = Nn * stv;
    const float Ff = M::clamp(Gf, -mg, mg);
    const float f = Fr + Ff;
    const float f1 = dpp<kRot1>(f), f2 = dpp<kRot2>(f);
    const float tau = r1 * f2 - r2 * f1;
    const float Tsum = sum_points(tau);
    const float wraw = I0 * dpp<kB0>(Tsum) + I1 * dpp<kB1>(Tsum) + I2 * dpp<kB2>(Tsum);

    // ------------------------------------------------------------------------------------------------
    // integrator backward (torchdiffeq fixed-grid Euler): adjoint of the step's outputs -> (g_xdd, g_wd, g_Fs, g_Ff)
    // ------------------------------------------------------------------------------------------------
    const float h = cur.t1 - cur.t0;
    laFs += act ? up.gFs : zero;
    laFf += act ? up.gFf : zero;
    const float gxdd = h * lxd, gwd = h * lw;
    lxd = fmaf(h, lx, lxd);                               // x' = x + h xd
    const float gFr_up = h * laFs, gFf_up = h * laFf;
    {   // R' = R + h [w]x R, column by column: d/dw of (w x R_j) . g_j = R_j x g_j ; d/dR_j = g_j x w   (g_j = h lR[:, j])
      const float g0 = h * lR0, g1 = h * lR1, g2 = h * lR2;
      const float g01 = dpp<kRot1>(g0), g02 = dpp<kRot2>(g0), g11 = dpp<kRot1>(g1), g12 = dpp<kRot2>(g1), g21 = dpp<kRot1>(g2), g22 = dpp<kRot2>(g2);
      lw += (dpp<kRot1>(R0) * g02 - dpp<kRot2>(R0) * g01) + (dpp<kRot1>(R1) * g12 - dpp<kRot2>(R1) * g11) + (dpp<kRot1>(R2) * g22 - dpp<kRot2>(R2) * g21);
      lR0 += g01 * w2 - g02 * w1;
      lR1 += g11 * w2 - g12 * w1;
      lR2 += g21 * w2 - g22 * w1;
    }

    // ------------------------------------------------------------------------------------------------
    // RHS backward
    // ------------------------------------------------------------------------------------------------
    const float mwd = inside(wraw, -a.omega_max, a.omega_max) ? gwd : zero;
    const float gtau = J0 * dpp<kB0>(mwd) + J1 * dpp<kB1>(mwd) + J2 * dpp<kB2>(mwd);      // I^-T m
    const float gsum = gxdd * a.inv_mass;
    const float gt1 = dpp<kRot1>(gtau), gt2 = dpp<kRot2>(gtau);
    const float gf = gt1 * r2 - gt2 * r1;                 // tau += r x f : df = gtau x r
    float gr = f1 * gt2 - f2 * gt1;                       //                dr = f x gtau
    float gFr = gFr_up + gsum + gf;
    const float gFf_ = gFf_up + gsum + gf;
    const float gG = inside(Gf, -mg, mg) ? gFf_ : zero;
    const float gNn = dot3(gG, stv);
    const float gst = Nn * gG;
    const float gsn = -dot3(gst, nrm);
    float gn = gsn * s - sn * gst;
    const float gslip = gst + gsn * nrm;
    const float gmuq = dot3(gslip, cmdv);
    const float gcmd = mub * gslip;
    float gvp = -gcmd;
    const float gtv = dot3(gcmd, e);                       // tv_v = tv_w = 0 for non-driving points
    const float ge_p = tv * gcmd;
    const float gv_p = tv_v * gtv, gwc_p = tv_w * gtv;
    gFr = fmaf(Nn > zero ? gNn * M::div(one, Nn) : zero, Fr, gFr);
    const float gF1 = inside(F1, -mg, mg) ? gFr : zero;
    const float dF = dot3(gF1, F0);
    const float gc_p = dF * inv_csum;
    const float gS = sum_points(-(dF * cj) * inv_csum * inv_csum);
    const float gF0 = gF1 * cj * inv_csum;
    const float gA = -dot3(gF0, nrm);
    gn = fmaf(-A, gF0, gn);
    const float gdh_p = a.k * gA;
    const float gvn = a.damp * gA;
    gvp = fmaf(gvn, nrm, gvp);
    gn = fmaf(gvn, vp, gn);
    const float gcw = gc_p + gS;
    const float gdh = gdh_p + gcw * (-10.0f) * cj * (one - cj);
    const float gzq = -gdh;
    // n = u / |u|, u = (-gx, -gy, 1): components 0, 1 carry the finite differences
    const float dotn = dot3(gn, nrm);
    const float gg = -((gn - dotn * nrm) * inl) * a.inv_res;      // lane 0: ggx, lane 1: ggy
    const float ggp = dpp<0x51>(gg);                                // quad_perm [1,0,1,1]
    const float nz = fmaf(gzq, wq, sA * gg + sB * ggp);
    const float nm = gmuq * wq;
    {   // this lane's cell accumulator
      const unsigned ni = (unsigned)idx;
      const bool same = !act | (ni == acc_idx);          // absent points contribute exact zeros: never flushed
      st_pending = !same;
      st_idx = acc_idx; st_z = acc_z; st_m = acc_m;
      acc_idx = act ? ni : acc_idx;
      acc_z = same ? acc_z + nz : nz;
      acc_m = same ? acc_m + nm : nm;
    }
    // d(sample)/d(position) through the fractions only: d wq / d fx = wa_s * wb, d wq / d fy = wb_s * wa
    const float vq = gzq * zc + gmuq * mcv;
    const float gpx = sum4(vq * (wa_s * wb)), gpy = sum4(vq * (wb_s * wa));
    const float gp = q == 0 ? gpx * a.inv_res : (q == 1 ? gpy * a.inv_res : gdh);
    // v_p = xd + w x r
    const float gvp1 = dpp<kRot1>(gvp), gvp2 = dpp<kRot2>(gvp);
    gr += gvp1 * w2 - gvp2 * w1;                           // dr += gvp x w
    const float gw_p = r1 * gvp2 - r2 * gvp1;              // dw += r x gvp
    const float qa = gp + gr;                              // p = R P + x, r = p - x
    // sums over the contact points
    lx += sum_points(gp);
    lxd += sum_points(gvp);
    lw += sum_points(gw_p);
    lR0 += sum_points(qa * P0); lR1 += sum_points(qa * P1); lR2 += sum_points(qa * P2);
    const float ge = sum_points(ge_p);
    const float gv = sum_points(gv_p), gwc = sum_points(gwc_p);
    {   // e = col0(R) / max(|col0|, eps): through |col0| only when it is >= eps
      const float dote = coln2 >= 1e-12f ? dot3(ge, e) : zero;
      lR0 = fmaf(ge - dote * e, il, lR0);
    }
    gctrl_pending = gctrl + n * 2; gv_pending = gv; gwc_pending = gwc;
    cur = nxt;
    up = up_next;
  }
  flush_stash();
  gctrl_pending[0] = gv_pending; gctrl_pending[1] = gwc_pending;
  if (act) {                               // what is still accumulated in registers
    atomic_add(at32(gzmap, goff + acc_idx), acc_z);
    if (want_gmu) atomic_add(at32(gmumap, goff + acc_idx), acc_m);
  }
  // output 0 is the initial state itself (its forces are constant zeros)
  if (n_steps == 0) load_upstream(0, up);      // T == 1: the loop never prefetched row 0
  add_upstream_state(up);

  // terrain snap of the initial height: x.z = mean_i blend(z; cell((R0 P_i + x0).xy))   (dphysics.py:567-571)
  float gx0 = lx;
  if (!a.skip_snap) {
    const float Ra = a.R0[b * 9 + cc * 3 + 0], Rb = a.R0[b * 9 + cc * 3 + 1], Rc = a.R0[b * 9 + cc * 3 + 2];
    const float x0c = a.x_init[b * 3 + cc];
    const float g = dpp<kB2>(lx) / (float)a.N;
    const float pc = (P0 * Ra + P1 * Rb + P2 * Rc) + x0c;
    const float lim = 262144.0f;
    const float uq = M::cell_coord(pc, a.d_max, a.res, a.inv_res);
    const int ui = (int)M::clamp(uq, -lim, lim);
    const float fr = uq - (float)ui;
    const int base = dppi<kB1>(ui) + __mul24(a.H, dppi<kB0>(ui));
    const int idx = min(max(base + cell_off, 0), last);
    const float wa = fmaf(wa_s, dpp<kB0>(fr), wa_o), wb = fmaf(wb_s, dpp<kB1>(fr), wb_o);
    const float zc = ld32(zmap, moff + (unsigned)idx);
    if (act) atomic_add(at32(gzmap, goff + (unsigned)idx), g * (wa * wb));
    const float gpx = sum4(zc * (wa_s * wb)) * g * a.inv_res, gpy = sum4(zc * (wb_s * wa)) * g * a.inv_res;
    float gpxy = q == 0 ? gpx : (q == 1 ? gpy : zero);
    gpxy = act ? gpxy : zero;
    gx0 = (q < 2 ? lx : zero) + sum_points(gpxy);        // the caller's x0.z is overwritten, so nothing flows to it
    lR0 += sum_points(gpxy * P0); lR1 += sum_points(gpxy * P1); lR2 += sum_points(gpxy * P2);
  }
  if (p == 0) {
    if (a.gx0) a.gx0[b * 3 + cc] = gx0;
    a.gxd0[b * 3 + cc] = lxd;
    a.gw0[b * 3 + cc] = lw;
    a.gR0[b * 9 + cc * 3 + 0] = lR0; a.gR0[b * 9 + cc * 3 + 1] = lR1; a.gR0[b * 9 + cc * 3 + 2] = lR2;
  }
}

bool use_component_parallel_bwd(const MfRolloutDesc* d, const MfRolloutBwdBufs* p);
int launch_rollout_bwd_cp_f32(const RolloutBwdArgs<float>& a, int integ, hipStream_t st);

}  // namespace mf

// Fused DPhysics rollout, forward pass, COMPONENT-PARALLEL lane mapping (see rollout_cp_common.h): float32 fast math, rigid
// bodies of up to 4 contact points, the mapping for launches that leave most SIMDs without a wave (B <= ~4096 at N = 4).
// Same physics, same outputs as rollout_fwd_kernel.h (dphysics.py:172-272, 385-455, 467-594 of the reference); sums run in a
// different order (butterflies over lanes), so results agree with the other mappings to rounding, not bit for bit.
#pragma once
#include "rollout_cp_common.h"
#include "rollout_fwd_kernel.h"
#include <type_traits>

namespace mf {

// LOSS (default integrator, states only): physics_loss (losses.py:102-127) accumulated while the stamped rows are written
// (MfRolloutLoss): per lane the squared, time-weighted error of its position component at the <= T2 rows `near` names -- a
// scalar compare per row, a dozen instructions at a stamped one, its ground truth prefetched one stamp ahead -- then one
// partial sum per workgroup and the mean by the workgroup that takes the last ticket.  No [T][B][3] gradient rows, no loss launches.
template <typename S, int INTEG, bool FORCES, bool ZMU, bool REC, bool LOSS = false>
// (waves_per_eu: these launches hold at most two waves per SIMD -- without the hint the scheduler guards an occupancy of eight and leaves
//  DPP hazards as s_nops rather than cross 64 registers: 335 -> 309 instructions per two steps with the record, 303 -> 301 without)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) rollout_fwd_cp_kernel(const RolloutArgs<S> a) {
  using namespace cp;
  using M = Mth<S, std::is_same<S, float>::value>;      // float: fast math; double (the validation build): exact
  using Msk = typename MaskOf<S>::type;
  constexpr unsigned kS = (unsigned)sizeof(S);
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = (tid >> 4) + a.b0;      // one 16-lane row per rollout
  if (b >= a.B) return;                  // whole rows leave together; DPP never crosses a row
  if (a.loss_poison != nullptr && tid == 0) a.loss_poison[0] = (S)__builtin_nanf("");      // MF_LOSS_VALUE_IN_BACKWARD: not yet known
  const int p = (tid >> 2) & 3;          // quad = contact point
  const int q = tid & 3;                 // lane of the quad: cell role q, component role cc
  const int cc = q < 3 ? q : 2;
  const S one = S(1.0), zero = S(0.0);
  const int HW = a.H * a.W, last = HW - 1;
  const bool has_mu = a.mu != nullptr;   // wave-uniform
  const unsigned moff = a.map_shared ? 0u : (unsigned)b * (unsigned)HW;
  const S* zmap = a.z;
  const S* mumap = has_mu ? a.mu : a.z;

  // ---- per-lane constants ----
  const bool act = p < a.N;
  const int pi = act ? p : 0;
  const S P0 = a.points[pi * 3 + 0], P1 = a.points[pi * 3 + 1], P2 = a.points[pi * 3 + 2];
  const int part = act ? a.part[pi] : -1;
  // track speed of this point = tv_v * v + tv_w * w (dphysics.py:75-104, 242-246); 0 for non-driving points
  const S tv_v = part < 0 ? zero : one;
  const S tv_w = part < 0 ? zero : ((part & 1) ? a.half_ly : -a.half_ly);
  const S I0 = a.Iinv[cc * 3 + 0], I1 = a.Iinv[cc * 3 + 1], I2 = a.Iinv[cc * 3 + 2];   // row cc of I^-1
  const S grav_c = cc == 2 ? a.mg * a.inv_mass : zero;
  const int cell_off = ((q & 1) ? a.H : 0) + ((q & 2) ? 1 : 0);     // c, f (+x neighbour: +H), l (+y: +1), fl
  // weight of cell q = (q & 2 ? fx : 1 - fx) * (q & 1 ? fy : 1 - fy)   (dphysics.py:442-445: fx pairs with the +y neighbour)
  const S wa_s = (q & 2) ? one : -one, wa_o = (q & 2) ? zero : one;
  const S wb_s = (q & 1) ? one : -one, wb_o = (q & 1) ? zero : one;
  const S n_mul = q < 2 ? -a.inv_res : zero, n_add = q < 2 ? zero : one;   // u = (-gx, -gy, 1)

  // ---- state: component cc / row cc in this lane, replicated over the four quads ----
  S x, xd, w, R0, R1, R2;
  if (a.default_state) {   // the reference's default start (dphysics.py:554-559), written back for the caller / the backward
    const S v0 = a.controls[(size_t)b * a.ctrl_sb + 0], w0 = a.controls[(size_t)b * a.ctrl_sb + 1];
    x = zero; xd = cc == 0 ? v0 : zero; w = cc == 2 ? w0 : zero;
    R0 = cc == 0 ? one : zero; R1 = cc == 1 ? one : zero; R2 = cc == 2 ? one : zero;
    if (p == 0) {
      S* oxd = const_cast<S*>(a.xd0); S* oR = const_cast<S*>(a.R0); S* ow = const_cast<S*>(a.w0);
      a.x0[b * 3 + cc] = x; oxd[b * 3 + cc] = xd; ow[b * 3 + cc] = w;
      oR[b * 9 + cc * 3 + 0] = R0; oR[b * 9 + cc * 3 + 1] = R1; oR[b * 9 + cc * 3 + 2] = R2;
    }
  } else {
    x = a.x0[b * 3 + cc]; xd = a.xd0[b * 3 + cc]; w = a.w0[b * 3 + cc];
    R0 = a.R0[b * 9 + cc * 3 + 0]; R1 = a.R0[b * 9 + cc * 3 + 1]; R2 = a.R0[b * 9 + cc * 3 + 2];
  }

  // footprint of the point under position component pc: this lane's cell index and weight
  auto footprint = [&](S pc, int* idx, S* wq, S* ou = nullptr) {
    const S lim = S(262144.0);
    const S u = M::cell_coord(pc, a.d_max, a.res, a.inv_res);      // lanes 0, 1: ux, uy
    const int ui = (int)M::clamp(u, -lim, lim);                         // trunc toward zero, like .long()
    const S fr = u - (S)ui;
    const int base = dppi<kB1>(ui) + __mul24(a.H, dppi<kB0>(ui));       // iy + H * ix
    *idx = min(max(base + cell_off, 0), last);                          // the reference clamps the FLAT index (:432-435)
    const S wa = mf_fma(wa_s, dpp<kB0>(fr), wa_o), wb = mf_fma(wb_s, dpp<kB1>(fr), wb_o);   // exact: 1 - f or f
    *wq = wa * wb;
    if (ou) *ou = u;
  };

  // start at the terrain height: x.z <- mean_i interp(z, (P R^T + x)_i)   (dphysics.py:567-571)
  if (!a.skip_snap) {
    const S pc = (P0 * R0 + P1 * R1 + P2 * R2) + x;
    int idx; S wq;
    footprint(pc, &idx, &wq);
    const S zq = dot4(wq, ld32(zmap, moff + (unsigned)idx));
    const S acc = sum_points(act ? zq : zero);
    const S xz = acc / (S)a.N;
    x = cc == 2 ? xz : x;
    if (p == 0 && q == 2) a.x0[b * 3 + 2] = xz;
  }

  // ---- output rows: base + per-lane byte offset (fixed) + wave-uniform byte offset of the time step (rollout_cp_common.h) ----
  const unsigned row_stride = (a.layout == MF_LAYOUT_TIME_MAJOR) ? (unsigned)a.B : 1u;   // rows between consecutive t
  const unsigned row0 = (a.layout == MF_LAYOUT_TIME_MAJOR) ? (unsigned)b : (unsigned)b * (unsigned)a.T;
  // vec3 rows: quad 0 writes Xs, quad 1 Xds, quad 2 Omegas, quad 3 the unshifted positions (or Xs again when nobody wants them):
  // four different arrays in one store instruction, so this one takes per-lane 64-bit addresses
  S* v3base = p == 0 ? a.Xs : p == 1 ? a.Xds : p == 2 ? a.Om : (a.Xraw ? a.Xraw : a.Xs);
  const S sink_l = (p == 3 && a.Xraw) ? zero : a.sink;
  const Msk m_xd = p == 1 ? ~(Msk)0 : (Msk)0, m_w = p == 2 ? ~(Msk)0 : (Msk)0, m_x = ~(m_xd | m_w);
  char* p3 = reinterpret_cast<char*>(v3base) + (size_t)(row0 * 3u + (unsigned)cc) * kS;
  const Rsrc rCtrl = make_rsrc(a.controls);
  unsigned o9 = (row0 * 9u + (unsigned)cc * 3u) * kS;
  const unsigned frow = (unsigned)a.fstride * 3u;
  unsigned ofs = (row0 * frow + (unsigned)p * 3u + (unsigned)cc) * kS, off = ofs;
  const unsigned d3 = row_stride * 3u * kS, d9 = row_stride * 9u * kS, df = row_stride * frow * kS;
  // wave-uniform running row pointers (scalar registers, scalar adds); the per-lane part is the fixed 32-bit offset
  const char* pRs = reinterpret_cast<const char*>(a.Rs);
  const char* pFs = reinterpret_cast<const char*>(a.Fs);
  const char* pFf = reinterpret_cast<const char*>(a.Ff);

  S oFs = zero, oFf = zero;   // forces of the pending output row (ODEINT: running impulses, dphysics.py:506-509)

  // the pending row from explicit values (the pipelined loop has already advanced x and R when it stores the row)
  auto emit_row = [&](S ex, S exd, S ew, S e0, S e1, S e2, unsigned adv) {
    const S vx = mf_fma(e2, sink_l, ex);          // Xs += Rs[..., :, 2] * m g / (k + 1e-6)   (dphysics.py:587-589)
    const S v3 = mask_or(mask_or(mask_or(zero, vx, m_x), exd, m_xd), ew, m_w);
    __builtin_nontemporal_store(v3, reinterpret_cast<S*>(p3));
    // (scalar base + RUNNING 32-bit per-lane offset: the step's advance is one vector add per array.  A scalar running pointer
    // costs s_add + s_addc, and scalar instructions take a full issue slot when a SIMD holds one wave: 22 of them per step were
    // 11 % of the forward, tools/pmc_groups.sh)
    bstore3(pRs, o9, 0u, e0, e1, e2);               // row cc of R: one 12-byte store (all quads, same address, same value)
    if (FORCES) { bstore1(pFs, ofs, 0u, oFs); bstore1(pFf, off, 0u, oFf); }
    if (adv) { p3 += d3; o9 += d9; ofs += df; off += df; }
  };

  const int n_steps = (INTEG == MF_INTEG_ODEINT_EULER) ? a.T - 1 : a.T;
  unsigned v_ctrl = (unsigned)b * (unsigned)a.ctrl_sb * kS;      // this rollout's control rows (bytes)
  S cv, cw;
  bload2(rCtrl, v_ctrl, 0u, &cv, &cw);
  S h_ode = (INTEG == MF_INTEG_ODEINT_EULER && a.T > 1) ? a.ts[1] - a.ts[0] : zero;
  // Everything of a step that depends on the pose (x, R) only: r = R P (component cc), p = r + x, this lane's footprint cell
  // and weight, the gathers, the thrust direction.  The explicit scheme knows the NEXT pose as soon as a step starts
  // (x' = x + h xd, R' = R + h [w]x R use the old xd, w), so its kernels compute the geometry of step n + 1 -- and issue its
  // gathers -- while the contact chain of step n runs: two independent instruction streams in one basic block fill each
  // other's dependency stalls, and a gather has a whole step to arrive.
  struct Geo { S r, pc, wq, zc, mc, e, u, il, coln2; int idx; };
  auto geometry = [&](S gx, S g0, S g1, S g2) {
    Geo g;
    g.r = cp_body_r(P0, P1, P2, g0, g1, g2);         // (:200)
    g.pc = g.r + gx;
    footprint(g.pc, &g.idx, &g.wq, &g.u);
    const int idx = g.idx;
    if constexpr (ZMU) {
      const Pk2<S> zm = *reinterpret_cast<const Pk2<S>*>(reinterpret_cast<const char*>(a.zmu) + (size_t)((unsigned)idx * 2u * kS));
      g.zc = zm.a; g.mc = zm.b;
    } else {
      g.zc = ld32(zmap, moff + (unsigned)idx);
      g.mc = ld32(mumap, moff + (unsigned)idx);       // aliases z without a friction map; selected after the blend
    }
    g.coln2 = dot3(g0, g0);
    g.il = M::inv_len(g.coln2);
    g.e = g0 * g.il;                                  // thrust direction = normalized first column of R (:237)
    return g;
  };
  // contact model + wrench of one step from its geometry and the state's velocities: (xdd, wd, F_spring, F_friction)
  // REC: the compact per-step record for the backward (layout: rollout_cp_common.h) -- ONE 16-byte store per lane and step of four
  // values the lane holds anyway (no merging by lane role: a select is a VALU issue slot, and at one wave per SIMD the issue slots
  // ARE the time; round 2 wrote four stores, 1 KiB per rollout-step: forward-with-record 4.05x its algorithmic HBM bytes, backward
  // 2.06x).  One scalar base + one running 32-bit per-lane offset: mf_rollout_record_bytes keeps the record < 4 GiB.
  char* const pRec0 = reinterpret_cast<char*>(a.rec);
  unsigned rec_off = (unsigned)tid * kRecBytesPerLane<S>;
  const unsigned rec_step = (unsigned)a.B * 16u * kRecBytesPerLane<S>;
  typedef S f4v __attribute__((ext_vector_type(4)));
  auto rec_store = [&](const f4v& rq, unsigned adv) {
#if defined(MF_REC_NOSTORE)        // A/B builds (tools/build_variant.sh): what the record's store costs / the plain-store form
    asm volatile("" :: "v"(rq));
#elif defined(MF_REC_PLAIN)
    *reinterpret_cast<f4v*>(pRec0 + (size_t)rec_off) = rq;
#else
    __builtin_nontemporal_store(rq, reinterpret_cast<f4v*>(pRec0 + (size_t)rec_off));
#endif
    if (adv) rec_off += rec_step;
  };
  auto contact = [&](const Geo& g, S vxd, S vw, S tv, S* xdd, S* wd, S* oFr, S* oFf, f4v* rq) {
    const S vp = cp_vel(vxd, vw, g.r);                             // v_p = xd + w x r   (:204)
    const S zq = dot4(g.wq, g.zc);                               // height under the point (:211)
    const S mub = dot4(g.wq, has_mu ? g.mc : one);             // friction (:216); no map = a map of ones (:562)
    const S dz = g.zc - dpp<kB0>(g.zc);                           // lane 1: z_f - z_c, lane 2: z_l - z_c
    const S u = mf_fma(dpp<kN12>(dz), n_mul, n_add);                // (-gx, -gy, 1)
    const S inl = M::inv_len(dot3(u, u));
    const S nrm = u * inl;
    const S dh = dpp<kB2>(g.pc) - zq;                             // soft contact + spring-damper along the normal (:220-230)
    S cj = M::sigmoid_m10(dh);
    cj = act ? cj : zero;
    const S csum = sum_points(cj);                                // n_contact_pts (:231)
    const S inv_csum = M::div(one, csum);
    const S vn = dot3(vp, nrm);
    const S A = cp_normal_force(a.k, dh, a.damp, vn);
    const S Fr = M::clamp(cp_spring(A, nrm, cj, inv_csum), -a.mg, a.mg);      // (:232-233)
    const S Nn = M::sqrt(dot3(Fr, Fr));                           // (:238)
    const S s = mub * cp_cmd(tv, g.e, vp);                        // slip (:247)
    const S sn = dot3(s, nrm);
    const S Ff = M::clamp(Nn * cp_tangent(s, sn, nrm), -a.mg, a.mg);   // (:248-251); absent points: cj = 0 -> Fr = Nn = Ff = 0
    const S f = Fr + Ff;
    const S tau = unrot(cross_pre(g.r, f));                        // r x (Fs + Ff)   (:255)
    const S Fsum = sum_points(f), Tsum = sum_points(tau);
    // omega_d = clamp(I^-1 tau)   (:256-257); xdd = (m g ghat + sum F) / m   (:264-266)
    const S wraw = cp_wraw(I0, I1, I2, Tsum);
    *wd = M::clamp(wraw, -a.omega_max, a.omega_max);
    *xdd = Fsum * a.inv_mass - grav_c;
    *oFr = Fr; *oFf = Ff;
    if constexpr (REC) *rq = f4v{g.u, cj, wraw, A};
  };

  // ---- fused physics loss: the next stamped row (wave-uniform: the stamps are the same for every rollout) and its ground truth ----
  // The stamp the rollout is waiting for is re-read EVERY step (one vector load that hits L1, two scalar loads), next to the step's
  // other loads.  (With the loads inside a branch taken at the stamped rows the wait-count pass could no longer tell how many
  // loads are in flight behind the step's gathers and waited for them a step early -- a full L2 round trip per step:
  // 0.165 -> 0.189 ms.)
  S l_acc = zero;
  int l_j = 0;                                                    // stamps met so far = index of the one the rollout is waiting for
  const int l_last = LOSS ? a.loss_T2 - 1 : 0;
  const unsigned l_lane = LOSS ? ((unsigned)b * (unsigned)a.loss_T2 * 3u + (unsigned)cc) : 0u;      // this lane's component of its rollout's first stamp
  struct Stamp { S g, w; };
  auto loss_peek = [&](int row) {                               // weight of output row `row` (0: no stamp) and the awaited stamp's ground truth
    Stamp st = {zero, zero};
    if constexpr (LOSS) {
      st.g = ld32(a.loss_gt, l_lane + (unsigned)(min(l_j, l_last) * 3));
      st.w = a.loss_row_w[row];
    }
    return st;
  };
  auto loss_row = [&](S ex, S e2, const Stamp& st) {      // an output row: ex = position component, e2 = R[:, 2] component
    if constexpr (LOSS) {
      // branch-free: the term is formed at EVERY row, with weight 0 off the stamps (six VALU instructions; a branch at the stamped
      // rows -- even one holding arithmetic only -- ends the basic block, and the tail of one step and the head of the next no
      // longer fill each other's stalls: 0.165 -> 0.20 ms)
      const S term = cp_loss_term(mf_fma(e2, a.sink, ex), st.g, st.w);
      l_acc += st.w != zero ? term : zero;                       // (masked by the stamp: a diverged tail is inf * 0 = NaN otherwise)
      l_j += st.w != zero ? 1 : 0;                               // (weights are 1 / (1 + gamma t) > 0)
    }
  };

  if constexpr (INTEG == MF_INTEG_ODEINT_EULER) {
    // One step of the two-stream pipeline: geometry `g` / controls (cv, cw) / step size h of step n come in, those of step
    // n + 1 go out into the OTHER buffer set -- the loop below alternates two sets, so nothing is moved between iterations.
    S t_cur = a.T > 1 ? a.ts[1] : zero;
    const unsigned ctrl_step = (unsigned)a.ctrl_st * kS, v_ctrl_last = v_ctrl + (unsigned)(a.T - 1) * ctrl_step;
    auto ode_step = [&](int n, const Geo& g, Geo& g_next, S cv_n, S cw_n, S h, S& cv_next, S& cw_next, S& h_next) {
      // next step's controls and step size: loaded before the stores below (vmcnt retires in order)
      // (vector arithmetic on purpose -- a running per-lane offset clamped at the last row, the previous time stamp carried over:
      // the scalar index / address chains they replace were 10 scalar instructions per step, each a full issue slot here)
      v_ctrl = min(v_ctrl + ctrl_step, v_ctrl_last);
      bload2(rCtrl, v_ctrl, 0u, &cv_next, &cw_next);
      const S t_next = a.ts[min(n + 2, a.T - 1)];
      const Stamp stamp = loss_peek(n + 1);
      const S tv = cp_track(tv_v, tv_w, cv_n, cw_n);
      // ---- stream B: pose and geometry of step n + 1 (torchdiffeq fixed-grid euler: y_{n+1} = y_n + h f(t_n, y_n)) ----
      // (column j of R is a 3-vector over the lanes: dR_j = w x R_j)
      const S d0 = h * cross_pre(w, R0), d1 = h * cross_pre(w, R1), d2 = h * cross_pre(w, R2);
      const S xn = mf_fma(h, xd, x);
      const S Rn0 = R0 + unrot(d0), Rn1 = R1 + unrot(d1), Rn2 = R2 + unrot(d2);
      g_next = geometry(xn, Rn0, Rn1, Rn2);           // (after the last step: the final pose -- unused, in range)
      // ---- row n, AFTER the gathers in program order: their wait a step later then covers no store of this step ----
      emit_row(x, xd, w, R0, R1, R2, 1u);
      x = xn; R0 = Rn0; R1 = Rn1; R2 = Rn2;
      // ---- stream A: contact chain of step n ----
      S xdd, wd, Fr, Ff;
      f4v rq;
      contact(g, xd, w, tv, &xdd, &wd, &Fr, &Ff, &rq);
      // (measured and not kept: the quad held back and stored a step later next to the row stores -- its last value, the angular
      //  acceleration, only exists where the next step's chain starts -- 0.1662 -> 0.1698 ms at B = 1024; plain instead of
      //  non-temporal stores 0.1662 -> 0.1689; the store itself is ~4 us of the launch, keeping its four values alive ~3)
      if constexpr (REC) rec_store(rq, 1u);
      xd = mf_fma(h, xdd, xd);
      w = mf_fma(h, wd, w);
      oFs = mf_fma(h, Fr, oFs);
      oFf = mf_fma(h, Ff, oFf);
      h_next = t_next - t_cur;          // (after the last step: 0, unused)
      t_cur = t_next;
      // fused loss: row n + 1 (the pose is already the next one's) -- its wave-uniform branch sits at the very END of the step, so the
      // step's two instruction streams stay in one basic block
      loss_row(x, R2, stamp);
    };
    Geo gA, gB;
    S cvA = cv, cwA = cw, hA = h_ode, cvB = zero, cwB = zero, hB = zero;
    if (n_steps > 0) gA = geometry(x, R0, R1, R2);
    loss_row(x, R2, loss_peek(0));     // (row 0 = the start state)
    __builtin_amdgcn_s_waitcnt(0);
    int n = 0;
    for (; n + 1 < n_steps; n += 2) {
      ode_step(n, gA, gB, cvA, cwA, hA, cvB, cwB, hB);
      ode_step(n + 1, gB, gA, cvB, cwB, hB, cvA, cwA, hA);
    }
    if (n < n_steps) ode_step(n, gA, gB, cvA, cwA, hA, cvB, cwB, hB);
  } else {
    __builtin_amdgcn_s_waitcnt(0);
    for (int n = 0; n < n_steps; ++n) {
      // next step's controls: loaded before the stores below (vmcnt retires in order)
      const int nn = min(n + 1, a.T - 1);
      S cv_next, cw_next;
      bload2(rCtrl, v_ctrl, __builtin_amdgcn_readfirstlane((unsigned)(nn * a.ctrl_st) * kS), &cv_next, &cw_next);
      const S tv = cp_track(tv_v, tv_w, cv, cw);
      S xdd, wd, Fr, Ff;
      // dynamics(): the next pose needs this step's forces (x += xd_new h, R <- R M(w_new)): one stream
      const Geo g = geometry(x, R0, R1, R2);
      emit_row(x, xd, w, R0, R1, R2, n > 0 ? 1u : 0u);      // n = 0: the initial state as a placeholder in row 0, overwritten one iteration later
      f4v rq;
      contact(g, xd, w, tv, &xdd, &wd, &Fr, &Ff, &rq);
      if constexpr (REC) rec_store(rq, 1u);
      // update_state (:274-288): xd += xdd h ; x += xd_new h ; w += wd h ; R <- R (I + K sin + K^2 (1 - cos))
      const S h = a.dt;
      xd = mf_fma(xdd, h, xd);
      x = mf_fma(xd, h, x);
      w = mf_fma(wd, h, w);
      const S th2 = dot3(w, w);
      const S il = M::inv_len(th2);
      const S kc = w * il;                                         // K = [w]x / max(|w|, eps)
      const S k1 = dpp<kRot1>(kc), k2 = dpp<kRot2>(kc);
      S sn_, oc;
      M::sincos_small(M::len_of(th2, il) * h, &sn_, &oc);
      // |k|^2 (1 above the eps floor): fast math from the scalars at hand -- two multiplies, no second cross-lane sum in the step's chain
      const S kk = M::kReciprocalNorm ? th2 * (il * il) : dot3(kc, kc);
      // row c of M = I + K sin + K^2 (1 - cos), stored relative to the diagonal: m0 = M[c][c], m1 = M[c][c+1], m2 = M[c][c+2]
      // (K[c][c+1] = -k_{c+2}, K[c][c+2] = k_{c+1}, K^2 = k k^T - |k|^2 I)
      const S ock = oc * kc;
      const S m0 = one + oc * (kc * kc - kk);
      const S m1 = mf_fma(ock, k1, -(sn_ * k2));
      const S m2 = mf_fma(ock, k2, sn_ * k1);
      // R'[c][j] = sum_m R[c][m] M[m][j]; M[m][j] lives in lane m as m_{(j - m) mod 3}
      const S n0 = R0 * dpp<kB0>(m0) + R1 * dpp<kB1>(m2) + R2 * dpp<kB2>(m1);
      const S n1 = R0 * dpp<kB0>(m1) + R1 * dpp<kB1>(m0) + R2 * dpp<kB2>(m2);
      const S n2 = R0 * dpp<kB0>(m2) + R1 * dpp<kB1>(m1) + R2 * dpp<kB2>(m0);
      R0 = n0; R1 = n1; R2 = n2;
      oFs = Fr; oFf = Ff;                                               // true forces of this step
      cv = cv_next; cw = cw_next;
    }
  }
  if (n_steps > 0 || INTEG == MF_INTEG_ODEINT_EULER) emit_row(x, xd, w, R0, R1, R2, 1u);
  if constexpr (LOSS) {
    static_assert(!LOSS || INTEG == MF_INTEG_ODEINT_EULER, "the fused loss rides on the default integrator's kernels");
    // One partial sum per workgroup (= wave), in a fixed order: the three component lanes of each rollout's first quad leave their
    // sums in LDS, lane 0 adds them row by row (a trailing workgroup may hold fewer than four rollouts: its absent rows stay zero).
    __shared__ S l_sh[80];
    const int lane = threadIdx.x;                              // (one wave per workgroup)
    if (lane < 16) l_sh[lane] = zero;
    if (p == 0 && q < 3) l_sh[(lane >> 4) * 4 + q] = l_acc;    // LDS executes a wave's operations in order
    __syncthreads();
    unsigned last = 0u;
    if (lane == 0) {
      S tot = zero;
#pragma unroll
      for (int r = 0; r < 4; ++r) tot += (l_sh[r * 4 + 0] + l_sh[r * 4 + 1]) + l_sh[r * 4 + 2];
      __builtin_nontemporal_store(tot, a.loss_partial + blockIdx.x);
      __threadfence();
      last = atomicAdd(a.loss_ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (last) {                                                // every workgroup has written its partial sum: the mean, in index order
      __threadfence();
      const int n_act = min(64, (a.B - (int)blockIdx.x * 4) * 16);     // live lanes of this (possibly trailing) workgroup: the first n_act
      S tot = zero;
      for (unsigned k2 = (unsigned)lane; k2 < gridDim.x; k2 += (unsigned)n_act) tot += __builtin_nontemporal_load(a.loss_partial + k2);
      l_sh[16 + lane] = tot;
      __syncthreads();
      if (lane == 0) {
        S sum = zero;
        for (int k2 = 0; k2 < n_act; ++k2) sum += l_sh[16 + k2];
        a.loss_out[0] = sum * a.loss_inv_count;
        *a.loss_ticket = 0u;
      }
    }
  }
}

// true when the component-parallel kernels cover this launch: float32 fast math, a rigid body of <= 4 points, full outputs
// (or states only), and few enough rollouts that the launch is bound by the instruction stream of its waves
bool use_component_parallel(const MfRolloutDesc* d, const MfRolloutFwdBufs* p, int scalar_bytes = 4);      // (8: the float64 validation build)
int launch_rollout_fwd_cp_f32(const RolloutArgs<float>& a, int integ, bool forces, bool zmu, hipStream_t st);   // a.rec: record wanted; a.loss_gt: fused loss
int launch_rollout_fwd_cp_f64(const RolloutArgs<double>& a, int integ, bool forces, bool zmu, hipStream_t st);  // the validation build (rollout_cp_f64.hip)
bool cp_loss_fusable(const MfRolloutDesc* d);           // the backward of this launch can carry the fused physics loss (either integrator)
bool cp_loss_in_forward(const MfRolloutDesc* d);        // ... and its forward can accumulate the value itself (LOSS kernels: default integrator)
long long cp_record_bytes(const MfRolloutDesc* d, int scalar_bytes = 4);      // bytes of the per-step record a launch of this shape writes (0: none)

// one launch of the instantiation the arguments call for (S = float: rollout_fwd_cp_fast.hip; S = double, the validation build:
// rollout_fwd_cp_f64.hip)
template <typename S>
int launch_rollout_fwd_cp_t(const RolloutArgs<S>& a, int integ, bool forces, bool zmu, hipStream_t st) {
  // one wave = 4 rollouts: B = 1024 puts one wave on each of the 256 CUs; workgroups of one wave, or of four where the dispatcher would
  // otherwise stack waves on a SIMD (wave_unit_block; the kernels that carry the fused loss -- <= two waves per CU -- are one-wave workgroups)
  const long long threads = (long long)a.B * 16;
  const int block = a.loss_gt ? 64 : (int)wave_unit_block((unsigned)((threads + 63) / 64));
  const unsigned grid = (unsigned)((threads + block - 1) / block);
  const bool rec = a.rec != nullptr;
  if (a.loss_gt) {      // fused physics loss: default integrator, states only (the host checked)
    constexpr int I = MF_INTEG_ODEINT_EULER;
    if (rec) { if (zmu) MF_KLAUNCH((rollout_fwd_cp_kernel<S, I, false, true, true, true>), dim3(grid), dim3(block), 0, st, a);
               else MF_KLAUNCH((rollout_fwd_cp_kernel<S, I, false, false, true, true>), dim3(grid), dim3(block), 0, st, a); }
    else     { if (zmu) MF_KLAUNCH((rollout_fwd_cp_kernel<S, I, false, true, false, true>), dim3(grid), dim3(block), 0, st, a);
               else MF_KLAUNCH((rollout_fwd_cp_kernel<S, I, false, false, false, true>), dim3(grid), dim3(block), 0, st, a); }
    hipError_t e = hipGetLastError();
    MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("rollout_fwd (component-parallel, fused loss) launch: ") + hipGetErrorString(e));
    return MF_OK;
  }
#define MF_CP(INTEG_, FORCES_, ZMU_) do { if (rec) MF_KLAUNCH((rollout_fwd_cp_kernel<S, INTEG_, FORCES_, ZMU_, true>), dim3(grid), dim3(block), 0, st, a); \
                                          else MF_KLAUNCH((rollout_fwd_cp_kernel<S, INTEG_, FORCES_, ZMU_, false>), dim3(grid), dim3(block), 0, st, a); } while (0)
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

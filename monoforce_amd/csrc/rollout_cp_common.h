// Component-parallel lane mapping of the rollout kernels (gfx950, wave64) -- shared lane algebra.
//
// The G = 4 mapping (rollout_fwd_kernel.h) gives one contact point to a lane: every 3-vector operation of a step costs three
// instructions, and the 18-float body update is repeated by each of the four lanes.  A launch of the BASELINE shape
// (1024 rollouts x 4 points) is then 64 waves on 1024 SIMDs, each issuing ~300 instructions per step at one per ~4.4 cycles:
// the step time IS the instruction count of one wave.  Here a rollout owns a 16-lane DPP row instead: quad p (lanes 4p..4p+3)
// is contact point p, and inside a quad
//   * lane c = 0, 1, 2 holds COMPONENT c of every 3-vector (x, xd, w, r, v_p, n, F, tau, ...) and ROW c of R; lane 3 mirrors
//     lane 2 (same values, same store addresses -- so stores need no exec mask and the loop stays one basic block);
//   * lane q = 0..3 gathers CELL q of the bilinear footprint (c, f, l, fl): one load per lane instead of two or four.
// Dot products are a multiply and two DPP adds (quad rotations), cross products read their rotated operands through DPP
// operands, sums over the contact points are two row rotations (row_ror:8, row_ror:4).  ~2.2x fewer instructions per wave and
// step; 4 rollouts per wave, so B = 1024 is 256 waves -- still at most one per SIMD up to B = 4096.
#pragma once
#include "mf_common.h"

namespace mf {
namespace cp {

// quad_perm encodings (sel0 | sel1 << 2 | sel2 << 4 | sel3 << 6)
constexpr int kRot1 = 0x09;    // [1,2,0,0]: lane c reads component (c+1)%3; lane 3 = lane 2's choice
constexpr int kRot2 = 0x52;    // [2,0,1,1]: lane c reads component (c+2)%3
constexpr int kB0 = 0x00, kB1 = 0x55, kB2 = 0xAA, kB3 = 0xFF;   // broadcast lane 0 / 1 / 2 / 3 of the quad
constexpr int kXor1 = 0xB1, kXor2 = 0x4E;
constexpr int kN12 = 0xA9;     // [1,2,2,2]: lane 0 <- 1, lane 1 <- 2 (finite differences z_f - z_c, z_l - z_c to components 0, 1)
constexpr int kRor4 = 0x124, kRor8 = 0x128;   // rotate within the 16-lane row

template <int CTRL, typename S>
__device__ __forceinline__ S dpp(S v) { return dpp_mov<CTRL>(v); }
template <int CTRL>
__device__ __forceinline__ int dppi(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }

// The kernels of this mapping are templates on the scalar S: float (fast math: the kernels every BASELINE configuration runs) and
// double (the VALIDATION build of the same source: exact arithmetic, no chaos -- tests/ hold it to the float64 oracle over the full
// horizon).  Bit masks of the lane algebra by type (mf_fma: mf_common.h):
template <typename S> struct MaskOf;
template <> struct MaskOf<float> { typedef unsigned type; };
template <> struct MaskOf<double> { typedef unsigned long long type; };

// sum of the three components held by lanes 0..2 of a quad; every lane (3 included, as lane 2's mirror) gets the total
template <typename S>
__device__ __forceinline__ S sum3(S v) { return (v + dpp<kRot1>(v)) + dpp<kRot2>(v); }
// a . b over the components: the product is rounded on its own (contracting it into the first add would need the rotated copy
// in a register of its own: one more instruction), then two DPP adds
template <typename S>
__device__ __forceinline__ S dot3(S a, S b) {
#pragma clang fp contract(off)
  const S v = a * b;
  return (v + dpp<kRot1>(v)) + dpp<kRot2>(v);
}
// bitwise select through precomputed lane masks (all ones / zero): a plain ternary on the lane role turns into branches
__device__ __forceinline__ float mask_or(float acc, float v, unsigned m) {
  return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, acc) | (__builtin_bit_cast(unsigned, v) & m));
}
__device__ __forceinline__ double mask_or(double acc, double v, unsigned long long m) {
  return __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, acc) | (__builtin_bit_cast(unsigned long long, v) & m));
}
// sum over the four lanes of a quad (cell roles)
template <typename S>
__device__ __forceinline__ S sum4(S v) { v += dpp<kXor1>(v); return v + dpp<kXor2>(v); }
// sum_q a_q b_q over the four lanes of a quad, THE SAME BITS in all four: the products are rounded before the butterfly (a
// contracted first add would make lane 0 hold fma(a0, b0, round(a1 b1)) and lane 1 fma(a1, b1, round(a0 b0)) -- one ulp apart;
// on the sampled height that ulp is multiplied by the contact stiffness, 5e4 N/m), and (p0 + p1) + (p2 + p3) commutes
template <typename S>
__device__ __forceinline__ S dot4(S a, S b) {
#pragma clang fp contract(off)
  S v = a * b;
  v += dpp<kXor1>(v);
  return v + dpp<kXor2>(v);
}
// sum over the four quads of a row, lane position by lane position (contact points of one rollout)
template <typename S>
__device__ __forceinline__ S sum_points(S v) { v += dpp<kRor8>(v); return v + dpp<kRor4>(v); }
// Cross products with the components in the lanes.  d = cross_pre(a, b) holds (a x b)_{c+2} in lane c:
//   d_c = a_c b_{c+1} - a_{c+1} b_c  -- two multiplies that take their rotated operand as a DPP operand, one subtract, no
// register spent on rotated copies; unrot() brings component c home to lane c (one more DPP operand of whatever consumes it,
// or one move).  Sums of cross products are un-rotated once: unrot(d1 + d2 + d3).
template <typename S>
__device__ __forceinline__ S cross_pre(S a, S b) {
#pragma clang fp contract(off)
  return a * dpp<kRot1>(b) - dpp<kRot1>(a) * b;
}
template <typename S>
__device__ __forceinline__ S unrot(S d) { return dpp<kRot1>(d); }

// ---- the contact model's formulas, ONE definition for the forward and the backward -------------------------------------------
// The backward rebuilds a step's intermediates from the saved state rows (and, where the forward kept one, from its compact
// per-step record).  Autograd differentiates the function the forward EVALUATED -- its clamp decisions, the sign of the normal
// force at the |F_n| kink -- so the rebuild has to reproduce the forward's values bit for bit, not to rounding: every formula
// both sides evaluate is written once, here, with its fused multiply-adds spelled out (left to the compiler, a * b + c * d may
// contract one way in the forward's loop and the other way in the backward's).  Same inputs, same instructions, same bits.
template <typename S>
__device__ __forceinline__ S cp_body_r(S P0, S P1, S P2, S g0, S g1, S g2) {      // r = R P, this lane's row (dphysics.py:200)
  return mf_fma(P2, g2, mf_fma(P1, g1, P0 * g0));
}
template <typename S>
__device__ __forceinline__ S cp_vel(S xd, S w, S r) { return xd + unrot(cross_pre(w, r)); }      // v_p = xd + w x r (:204)
template <typename S>
__device__ __forceinline__ S cp_track(S tv_v, S tv_w, S cv, S cw) { return mf_fma(tv_v, cv, tv_w * cw); }   // (:75-104)
template <typename S>
__device__ __forceinline__ S cp_normal_force(S k, S dh, S damp, S vn) { return mf_fma(k, dh, damp * vn); }    // A = k dh + d v_n (:230)
template <typename S>
__device__ __forceinline__ S cp_spring(S A, S nrm, S cj, S inv_csum) {                      // F_spring before its clamp (:230-232)
#pragma clang fp contract(off)
  return -(A * nrm) * (cj * inv_csum);
}
template <typename S>
__device__ __forceinline__ S cp_cmd(S tv, S e, S vp) { return mf_fma(tv, e, -vp); }               // cmd - v_p (:247)
template <typename S>
__device__ __forceinline__ S cp_tangent(S s, S sn, S nrm) { return mf_fma(-sn, nrm, s); }         // s - (s . n) n (:248-249)
// omega_d before its clamp: row c of I^-1 times the torque (the total sits replicated in the quad's component lanes)   (:256)
template <typename S>
__device__ __forceinline__ S cp_wraw(S I0, S I1, S I2, S Tsum) {
  return mf_fma(I2, dpp<kB2>(Tsum), mf_fma(I1, dpp<kB1>(Tsum), I0 * dpp<kB0>(Tsum)));
}
// physics_loss (losses.py:122-127) of one position component at a stamped row, and its derivative: the arithmetic of
// csrc/physics_loss.hip (pred w - gt w, squared; products rounded on their own), so the fused and the two-kernel route agree bit for bit
template <typename S>
__device__ __forceinline__ S cp_loss_term(S xs, S g, S w) {
#pragma clang fp contract(off)
  const S d = xs * w - g * w;
  return d * d;
}
template <typename S>
__device__ __forceinline__ S cp_loss_grad(S scale, S xs, S g, S w) {      // scale = 2 gloss / (B T2 3)
#pragma clang fp contract(off)
  return scale * w * (xs * w - g * w);
}
// bitwise merge (a where the mask is set, b elsewhere): one v_bfi_b32
__device__ __forceinline__ float bfi(unsigned m, float a, float b) {
  return __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, a) & m) | (__builtin_bit_cast(unsigned, b) & ~m));
}
__device__ __forceinline__ double bfi(unsigned long long m, double a, double b) {
  return __builtin_bit_cast(double, (__builtin_bit_cast(unsigned long long, a) & m) | (__builtin_bit_cast(unsigned long long, b) & ~m));
}
// an int32 riding in a plane of scalars (the streaming backward's ring): its bits in a float, its value in a double
__device__ __forceinline__ float idx_as(float, int i) { return __builtin_bit_cast(float, i); }
__device__ __forceinline__ double idx_as(double, int i) { return (double)i; }
__device__ __forceinline__ int idx_of(float v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ int idx_of(double v) { return (int)v; }
constexpr int kMir2 = 0xA4;    // quad_perm [0,1,2,2]: lane 3 <- lane 2 (a per-component value stored by lanes 0..2 of a quad)

// ---- the forward's compact per-step record (MfRolloutFwdBufs.rec): ONE 16-byte quad per lane and step, 256 B per rollout-step ----
// What the backward cannot get back from the saved state rows without redoing the contact chain -- and only values the storing
// lane holds anyway (no merge by lane role in the forward: at one wave per SIMD an issue slot is time):
//   x : lanes 0, 1 of a quad: the cell coordinates u = (p + d_max) / res of the point (x, y) -- index and fraction follow exactly:
//       i = trunc(u), f = u - i, and i + f == u in float32; lanes 2, 3: unused
//   y : the contact weight c of the point          z : the unclamped angular acceleration (this lane's component)
//   w : A = k dh + d v_n of the point
// Everything else of round 2's 1 KiB record is rebuilt by the waves that read it, with the forward's own instructions on the same
// inputs (so: the same bits): the gathered cells come back from the L2 (the maps are 512 KiB), normal, blended friction, 1 / sum c,
// |F_n|, s . n, |R[:, 0]| from those and the state rows.  A step's slab is [B * 16 lanes] quads: a wave's store / load is one KiB.
template <typename S>
constexpr unsigned kRecBytesPerLane = 4u * (unsigned)sizeof(S);      // 16 (float); the float64 validation build: 32

// ---- rows of the [T][B][...] arrays -----------------------------------------------------------------------------------------
// A row is addressed as wave-uniform base pointer + wave-uniform byte offset of the time step + a per-lane byte offset that
// never changes (32 bits: the host keeps every array of a component-parallel launch below 4 GiB).
// Measured alternative, NOT kept: the same accesses through buffer descriptors (raw_buffer_load / _store with the step offset
// in the scalar soffset operand) spend no vector instruction on addresses at all, but ran slower at every batch size --
// forward 0.207 vs 0.184 ms, backward 0.473 vs 0.439 ms at B = 1024 (twelve descriptors do not fit the scalar registers next
// to everything else, so words of them are re-assembled with s_mov before most accesses).  tools/microbench/buffer_ops.hip
// keeps the device check of that variant (it found that __builtin_bit_cast applied to an ext_vector ELEMENT reads element 0).
using Rsrc = const char*;
__device__ __forceinline__ Rsrc make_rsrc(const void* p) { return reinterpret_cast<const char*>(p); }
template <typename S> struct __attribute__((aligned(sizeof(S)))) Pk2 { S a, b; };
template <typename S> struct __attribute__((aligned(sizeof(S)))) Pk3 { S a, b, c; };
template <typename S>
__device__ __forceinline__ S bload1(Rsrc r, unsigned voff, unsigned soff) { return *reinterpret_cast<const S*>(r + (size_t)soff + (size_t)voff); }
template <typename S>
__device__ __forceinline__ void bload2(Rsrc r, unsigned voff, unsigned soff, S* a, S* b) {
  const Pk2<S> v = *reinterpret_cast<const Pk2<S>*>(r + (size_t)soff + (size_t)voff); *a = v.a; *b = v.b;
}
template <typename S>
__device__ __forceinline__ void bload3(Rsrc r, unsigned voff, unsigned soff, S* a, S* b, S* c) {
  const Pk3<S> v = *reinterpret_cast<const Pk3<S>*>(r + (size_t)soff + (size_t)voff); *a = v.a; *b = v.b; *c = v.c;
}
// output rows are written once and never re-read by the kernel: streaming (non-temporal) stores, so they do not evict the map
// cells the gathers keep hitting in L1 / L2
template <typename S>
__device__ __forceinline__ void bstore1(Rsrc r, unsigned voff, unsigned soff, S a) {
  __builtin_nontemporal_store(a, reinterpret_cast<S*>(const_cast<char*>(r) + (size_t)soff + (size_t)voff));
}
template <typename S>
__device__ __forceinline__ void bstore2(Rsrc r, unsigned voff, unsigned soff, S a, S b) {
  *reinterpret_cast<Pk2<S>*>(const_cast<char*>(r) + (size_t)soff + (size_t)voff) = Pk2<S>{a, b};
}
template <typename S>
__device__ __forceinline__ void bstore3(Rsrc r, unsigned voff, unsigned soff, S a, S b, S c) {
  if constexpr (sizeof(S) == 4) {      // one global_store_dwordx3
    typedef S f3v __attribute__((ext_vector_type(3)));
    typedef f3v __attribute__((aligned(sizeof(S)))) f3u;
    const f3v v = {a, b, c};
    __builtin_nontemporal_store(v, reinterpret_cast<f3u*>(const_cast<char*>(r) + (size_t)soff + (size_t)voff));
  } else {
    // (a 3-vector of doubles is stored as FOUR -- clang widens vec3 stores to vec4 with an undefined last element, which the backend
    //  narrows back for 12 bytes but not for 24: the fourth double landed on the next row's first element, racing with its owner)
    S* p = reinterpret_cast<S*>(const_cast<char*>(r) + (size_t)soff + (size_t)voff);
    __builtin_nontemporal_store(a, p); __builtin_nontemporal_store(b, p + 1); __builtin_nontemporal_store(c, p + 2);
  }
}

}  // namespace cp
}  // namespace mf

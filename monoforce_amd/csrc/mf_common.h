// Shared device helpers for the monoforce gfx950 kernels: cross-lane reductions on DPP, small vector math,
// error plumbing.  CDNA4 only (wave64); no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include <string>

#include "../../include/monoforce_hip.h"

namespace mf {

void set_error(const std::string& msg);  // capi.hip
// Compute units / SIMDs of the device the launch goes to (hipDeviceProp_t::multiProcessorCount; MI355X: 256 CUs x 4 SIMDs).  The
// dispatch thresholds of the rollout kernels are expressed in these (waves per SIMD, workgroups per CU), not in literals tuned on one
// box.  Where no device answers -- the policy queries of a CPU-only process, e.g. mf_rollout_record_bytes -- an MI355X is assumed.
int device_cus();   // capi.hip
inline long long device_simds() { return 4ll * device_cus(); }
// Workgroup size of the kernels whose unit of work is ONE wave (four component-parallel rollouts) with nothing shared between waves.
// Measured (profiles/r4_ab_block.txt, twice, on different boxes): with three or four waves per CU the same launch is 20-30 % faster
// as 256-thread workgroups (ONE per CU) than as 64-thread ones -- record-reading backward at 4096 rollouts 0.54 -> 0.44 ms, recording
// forward 0.265 -> 0.193 ms, and 0.373 -> 0.248 ms for an A/B build of the backward with every memory operation compiled out
// (profiles/r4_ab_saved_variants.txt).  The cause is NOT identified: the dispatcher spreads the waves evenly over the SIMDs either way
// (tools/microbench/wave_placement.hip, also with 208 registers and LDS), they start within 2 us of each other, and a plain 8 / 32 KB
// FMA loop runs equally fast in both forms (tools/microbench/ifetch_lockstep.hip).  Below two waves per CU 64-thread workgroups reach
// more CUs; from two waves per SIMD up the forms measure the same.  So the rule is the measured one: 256 threads between two and four
// waves per CU, 64 elsewhere.
inline unsigned wave_unit_block(unsigned waves) {
  static const int env = getenv("MF_CP_BLOCK") ? atoi(getenv("MF_CP_BLOCK")) : 0;      // A/B (tools/ab_block.sh): 64 / 128 / 256
  if (env == 64 || env == 128 || env == 256) return (unsigned)env;
  const unsigned cus = (unsigned)device_cus();
  return waves > 2u * cus && waves <= 4u * cus ? 256u : 64u;
}

// Which kernel an entry point launched (mf_last_launch, capi.hip): the rollout dispatchers note the host stub, grid and workgroup
// of every launch in thread-local storage -- two stores, no formatting; the name is looked up (hipKernelNameRefByPtr) and demangled
// only when somebody asks.
void note_launch(const void* kernel_stub, unsigned grid, unsigned block);
#define MF_KLAUNCH(kernel, grid, block, shmem, stream, ...)                                             \
  do {                                                                                                  \
    const dim3 mf_g_ = (grid), mf_b_ = (block);                                                         \
    ::mf::note_launch(reinterpret_cast<const void*>(&kernel), mf_g_.x, mf_b_.x);                         \
    hipLaunchKernelGGL(kernel, mf_g_, mf_b_, shmem, stream, __VA_ARGS__);                               \
  } while (0)

#define MF_REQUIRE(cond, code, msg)        \
  do {                                     \
    if (!(cond)) {                         \
      ::mf::set_error(msg);                \
      return (code);                       \
    }                                      \
  } while (0)

// ---------------------------------------------------------------------------------------------------------
// DPP lane permutes.  dpp_ctrl encodings (gfx9): quad_perm = sel0 | sel1<<2 | sel2<<4 | sel3<<6,
// row_mirror = 0x140 (lane i <-> 15-i of a 16-lane row), row_half_mirror = 0x141 (i <-> 7-i of an 8-lane half row).
// ---------------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  int2 p = __builtin_bit_cast(int2, v);
  p.x = __builtin_amdgcn_update_dpp(0, p.x, CTRL, 0xF, 0xF, true);
  p.y = __builtin_amdgcn_update_dpp(0, p.y, CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, p);
}
// DPP move restricted to the rows of ROW_MASK (the other rows read 0): the row_bcast steps of a wave64 reduction
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov_rows(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov_rows(double v) {
  int2 p = __builtin_bit_cast(int2, v);
  p.x = __builtin_amdgcn_update_dpp(0, p.x, CTRL, ROW_MASK, 0xF, false);
  p.y = __builtin_amdgcn_update_dpp(0, p.y, CTRL, ROW_MASK, 0xF, false);
  return __builtin_bit_cast(double, p);
}
__device__ __forceinline__ float read_lane63(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63)); }
__device__ __forceinline__ double read_lane63(double v) {
  int2 p = __builtin_bit_cast(int2, v);
  p.x = __builtin_amdgcn_readlane(p.x, 63);
  p.y = __builtin_amdgcn_readlane(p.y, 63);
  return __builtin_bit_cast(double, p);
}
// value of lane `l` (wave-uniform index) as a scalar
__device__ __forceinline__ float mf_readlane(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
__device__ __forceinline__ double mf_readlane(double v, int l) {
  int2 p = __builtin_bit_cast(int2, v);
  p.x = __builtin_amdgcn_readlane(p.x, l);
  p.y = __builtin_amdgcn_readlane(p.y, l);
  return __builtin_bit_cast(double, p);
}
__device__ __forceinline__ float lane_xor(float v, int m) { return __shfl_xor(v, m, 64); }
__device__ __forceinline__ double lane_xor(double v, int m) { return __shfl_xor(v, m, 64); }
// v(lane) + v(lane ^ 16) / v(lane) + v(lane ^ 32) on gfx950's row / half-wave swaps: V_PERMLANE16_SWAP exchanges the odd rows of its
// first operand with the even rows of its second, V_PERMLANE32_SWAP the upper half of the first with the lower half of the second --
// given the same value twice they return (rows 0 0 2 2, rows 1 1 3 3) and (lower lower, upper upper), whose sum is the pair sum in
// every lane.  Plain VALU instructions: no LDS-crossbar round trip (ds_bpermute_b32), no address register; the operands are added
// in the order (even row + odd row), so the result is the bits of own + partner.
__device__ __forceinline__ float pair_sum_xor16(float v) {
#ifdef MF_NO_PERMLANE_SWAP      // A/B hook (tools/build_variant.sh): the LDS-crossbar permute
  return v + lane_xor(v, 16);
#endif
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float pair_sum_xor32(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ double pair_sum_xor16(double v) { return v + lane_xor(v, 16); }
__device__ __forceinline__ double pair_sum_xor32(double v) { return v + lane_xor(v, 32); }

// All-reduce (sum) over aligned groups of G consecutive lanes; every lane of the group gets the total.
// G <= 16 stays on DPP (no LDS crossbar round trip): the butterfly xor1, xor2, then the mirror steps, which
// are valid because after each step the already-merged sub-groups hold identical values.
template <int G, typename S>
__device__ __forceinline__ S group_sum(S v) {
  if (G == 1) return v;
  static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16 || G == 32 || G == 64, "G must be a power of two <= 64");
  if (G >= 2) v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  if (G >= 4) v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  if (G >= 8) v += dpp_mov<0x141>(v);  // row_half_mirror
  if (G >= 16) v += dpp_mov<0x140>(v); // row_mirror
  if (G == 32) v = pair_sum_xor16(v);
  if (G >= 64 && sizeof(S) == 8) {   // float64: the permute form (the scalar-register form below miscompiled in the largest
    v += lane_xor(v, 16);            // float64 backward kernel, 64 lanes x 8 points, which spills heavily)
    v += lane_xor(v, 32);
  } else if (G >= 64) {
    // whole wave: every lane of a 16-lane row now holds its row total.  row_bcast:15 adds row r-1's total into rows 1 and 3,
    // row_bcast:31 adds lane 31's (= rows 0+1) into rows 2 and 3: lane 63 holds the wave total, which is wave-uniform and
    // comes back through a scalar register -- no LDS-crossbar permutes (ds_bpermute) at all.
    v += dpp_mov_rows<0x142, 0xA>(v);
    v += dpp_mov_rows<0x143, 0xC>(v);
    v = read_lane63(v);
  }
  return v;
}

// Group reductions of a rollout kernel.  Up to 64 lanes per rollout this is group_sum<G> (DPP / lane permutes, stateless).
// Larger bodies at small batch sizes spread ONE rollout over G / 64 waves of a workgroup (blockDim = G): every wave
// all-reduces its 64 lanes, writes the partial to LDS, one s_barrier, and every lane adds the G / 64 partials in a fixed
// order (deterministic).  Two LDS slots alternate: a wave can run at most one barrier ahead of the slowest one, so when it
// writes slot p again every wave has finished reading it.  All threads of the workgroup must make the same calls.
constexpr int kGroupSumMaxValues = 28;   // most values one sum_n() call reduces (the backward's 23 adjoint components + 4 joint-angle gradients)
template <int G, typename S>
struct GroupSum {
  static constexpr int NW = G > 64 ? G / 64 : 1;
  S* lds = nullptr;      // 2 * NW * kGroupSumMaxValues scalars of shared memory when G > 64
  unsigned par = 0;
  __device__ __forceinline__ S sum(S v) {
    if (G <= 64) return group_sum<(G <= 64 ? G : 64)>(v);
    S a[1] = {v};
    sum_n<1>(a);
    return a[0];
  }
  template <int K>
  __device__ __forceinline__ void sum_n(S (&v)[K]) {
    post<K>(v);
    wait<K>(v);
  }
  // The two halves of sum_n for callers with work to put between them (multi-wave groups: under the LDS write and ahead of the
  // barrier): post = every wave's own total, written to its LDS row; wait = the barrier and the sum over the rows, fixed order.
  template <int K>
  __device__ __forceinline__ void post(S (&v)[K]) {
    static_assert(K <= kGroupSumMaxValues, "raise kGroupSumMaxValues");
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = group_sum<(G <= 64 ? G : 64)>(v[k]);
    if (G <= 64) return;
    S* slot = lds + par * (NW * kGroupSumMaxValues);
    const int wave = threadIdx.x >> 6;
    // every lane of the wave holds the same total and writes it to the same word: no branch (a basic-block boundary here
    // would change which multiply-add pairs around a call get contracted, i.e. the bits of the trajectory)
#pragma unroll
    for (int k = 0; k < K; ++k) slot[wave * kGroupSumMaxValues + k] = v[k];
  }
  template <int K>
  __device__ __forceinline__ void wait(S (&v)[K]) {
    if (G <= 64) return;
    S t[K * NW];
    fetch<K>(t);
    fold<K>(t, v);
  }
  // ... and wait = fetch (the barrier and the reads of every wave's row) + fold (their sum), for callers that also have work to put
  // under the reads
  template <int K>
  __device__ __forceinline__ void fetch(S (&t)[K * NW]) {
    if (G <= 64) return;
    S* slot = lds + par * (NW * kGroupSumMaxValues);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int w = 0; w < NW; ++w) t[k * NW + w] = slot[w * kGroupSumMaxValues + k];
    par ^= 1u;
  }
  template <int K>
  __device__ __forceinline__ void fold(const S (&t)[K * NW], S (&v)[K]) {
    if (G <= 64) return;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      S acc = t[k * NW];
#pragma unroll
      for (int w = 1; w < NW; ++w) acc += t[k * NW + w];
      v[k] = acc;
    }
  }
};

// Workgroup sum of up to nine values (float32; float64 in the validation build) per step for ONE rollout spread over NW = G / 64 waves (the multi-wave rollout kernels).
// GroupSum's plain form -- every value summed over the wave's 64 lanes (6 DPP steps each), one LDS word per wave, NW partials added
// by every lane -- costs ~13 instructions per value.  Here the first EIGHT values are reduced TRANSPOSED within each 16-lane row:
// every DPP step (row mirror, half-row mirror, quad reverse) exchanges HALF of the values a lane still holds with a partner that
// keeps the other half, so the row totals cost 8 -> 4 -> 2 -> 1 adds instead of 8 x 4, and lane l ends with the row total of
// value (l >> 1) & 7.  Row totals go to LDS ([value][wave x 4 rows]); after ONE barrier lanes 0..8 add the 4 NW partials of
// "their" value in a fixed order (deterministic) and the totals come back through scalar registers (v_readlane).  A ninth value
// takes the plain row sum.  Two LDS slots alternate (a wave runs at most one barrier ahead of the slowest).
template <int NW, typename S = float>
struct TransposedExchange {
  static constexpr int RS = 4 * NW;               // row totals per value
  static constexpr int kWords = 2 * 9 * RS;       // scalars of shared memory, aligned to four of them (16 bytes; float64: 32)
  S* lds = nullptr;
  unsigned par = 0;
  __device__ __forceinline__ void post(const S (&v)[8], S v9) {
    const int l = threadIdx.x & 63;
    const bool b3 = (l & 8) != 0, b2 = (l & 4) != 0, b1 = (l & 2) != 0;
    S t[4], u[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {      // row mirror (i <-> 15 - i): the lanes of the upper half-row keep values 4..7
      const S keep = b3 ? v[k + 4] : v[k], send = b3 ? v[k] : v[k + 4];
      t[k] = keep + dpp_mov<0x140>(send);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {      // half-row mirror (i <-> 7 - i)
      const S keep = b2 ? t[k + 2] : t[k], send = b2 ? t[k] : t[k + 2];
      u[k] = keep + dpp_mov<0x141>(send);
    }
    S xv;
    {                                  // quad reverse (i <-> 3 - i)
      const S keep = b1 ? u[1] : u[0], send = b1 ? u[0] : u[1];
      xv = keep + dpp_mov<0x1B>(send);
    }
    xv += dpp_mov<0xB1>(xv);           // the neighbour holds the same value's other half
    S yv = v9;                     // the ninth value: plain row sum
    yv += dpp_mov<0xB1>(yv); yv += dpp_mov<0x4E>(yv); yv += dpp_mov<0x141>(yv); yv += dpp_mov<0x140>(yv);
    const int vi = (b3 ? 4 : 0) + (b2 ? 2 : 0) + (b1 ? 1 : 0);
    S* slot = lds + par * (9 * RS);
    const int col = (int)(threadIdx.x >> 6) * 4 + (l >> 4);
    slot[vi * RS + col] = xv;          // (the two lanes of a pair write the same total to the same word)
    slot[8 * RS + col] = yv;
  }
  template <int K>
  __device__ __forceinline__ void wait(S (&v)[K]) {      // totals of values 0 .. K - 1, workgroup-uniform
    static_assert(K <= 9, "nine values at most");
    typedef S f4v __attribute__((ext_vector_type(4)));
    const S* slot = lds + par * (9 * RS);
    __syncthreads();
    const int rk = min((int)(threadIdx.x & 63), 8);
    const f4v* pr = reinterpret_cast<const f4v*>(slot + rk * RS);
    S tot = (S)0;
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) { const f4v q = pr[wv]; tot += ((q.x + q.y) + (q.z + q.w)); }
    par ^= 1u;
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = mf_readlane(tot, k);
  }
};

// ---------------------------------------------------------------------------------------------------------
// scalar helpers, overloaded on the arithmetic type
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float mf_fma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double mf_fma(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ float mf_sqrt(float v) { return sqrtf(v); }
__device__ __forceinline__ double mf_sqrt(double v) { return sqrt(v); }
__device__ __forceinline__ float mf_exp(float v) { return expf(v); }
__device__ __forceinline__ double mf_exp(double v) { return exp(v); }
__device__ __forceinline__ void mf_sincos(float v, float* s, float* c) { sincosf(v, s, c); }
__device__ __forceinline__ void mf_sincos(double v, double* s, double* c) { sincos(v, s, c); }
template <typename S>
__device__ __forceinline__ S mf_clamp(S v, S lo, S hi) {
  // torch.clamp semantics: min(max(v, lo), hi); NaN propagates
  return v < lo ? lo : (v > hi ? hi : v);
}
template <typename S>
__device__ __forceinline__ S mf_max(S a, S b) { return a > b ? a : b; }

}  // namespace mf

// LSS BEV voxel pooling ("splat") for gfx950: sum the lifted camera features of all frustum points that fall into the
// same BEV voxel.  Replaces `LiftSplatShoot.voxel_pooling`
// (/root/reference/monoforce/src/monoforce/models/terrain_encoder/lss.py:238-280) and its `cumsum_trick` / `QuickCumsum`
// helpers (terrain_encoder/utils.py:144-181: argsort by voxel rank, prefix sum over ALL kept points, first-difference at
// run ends -- an fp32 prefix sum that loses ~1e-3 relative accuracy).
//
// Design (HBM-bound: read x once, write the dense BEV grid once, both fully coalesced):
//   prepare  (geometry only; reusable by forward AND backward, and across steps while the calibration is unchanged)
//     keys      one thread per point: voxel index with the reference's float32 arithmetic, trunc-toward-zero, in-bounds
//               mask -> linear voxel id (or -1); integer histogram of points per voxel
//     scan      exclusive prefix sum of the histogram -> CSR offsets
//     fill      CSR lists of point ids per voxel
//     rank      order each voxel's CSR segment by point id (thread per point: rank inside its short segment)
//   forward   one workgroup per tile of 64 consecutive voxels of one BEV plane.  float32 rows of whole channel quads (the reference's
//             shapes): splat_fwd_quad_kernel -- sixteen lanes per point, the tile's OCCUPIED voxels dealt to the workgroup's sixteen
//             rows by rank, eight feature rows in flight per row and eight waves per SIMD (the comment at the kernel has the
//             measurements that led there); other shapes / float64: splat_fwd_kernel -- 16 voxels per wave, lane = channel, a
//             16-deep branch-free load pipeline.  Every voxel is summed in ascending point order (deterministic, bit-reproducible,
//             the same bits from both kernels); the [voxel][channel] tile is transposed through LDS so every output row segment
//             (64 voxels of one channel plane) leaves as one contiguous 256 B store.  Tiles without points store zeros.
//   backward  the same tiling in reverse: the grad tile is loaded coalesced per channel plane, and each kept point's
//             feature-gradient row is written as one coalesced row (QuickCumsum.backward is exactly this gather);
//             rows of dropped points are zero-filled by the keys pass of the backward.
#include <cstdlib>
#include "mf_common.h"

namespace mf {

constexpr int kTileVox = 64;
// row loads in flight per wave in the fused backward's gather over a pixel's depth bins (unpipelined before: 152 -> 143 us at
// B = 8); the forward kernels: kFwdPipe below
constexpr int kPipeBwd = 8;

struct SplatWs {        // carve-up of the caller's workspace (all int32)
  int* keys;            // [P]      linear voxel id of each point, -1 = dropped
  int* count;           // [V]      points per voxel
  int* cursor;          // [V]      fill cursors
  int* offsets;         // [V + 1]  CSR offsets
  int* list;            // [P]      point ids grouped by voxel, ascending inside each voxel
  int* scratch;         // [P]      the same in arrival order (before the rank pass)
  int* block_sums;      // [ceil(V / 2048) + 1]
};

static inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

static size_t carve(const MfSplatDesc* d, void* base, SplatWs* ws) {
  const size_t P = (size_t)d->B * d->n_per_sample;
  const size_t V = (size_t)d->B * d->nz * d->nx * d->ny;
  const size_t nblk = (V + 2047) / 2048 + 1;
  size_t off = 0;
  char* b = (char*)base;
  auto take = [&](size_t n_int) { int* p = (int*)(b + off); off += align256(n_int * sizeof(int)); return p; };
  int* keys = take(P);
  int* count = take(V);      // count and cursor are adjacent so one memset clears both
  int* cursor = take(V);
  int* offsets = take(V + 1);
  int* list = take(P);
  int* scratch = take(P);
  int* bs = take(nblk);
  if (ws) { ws->keys = keys; ws->count = count; ws->cursor = cursor; ws->offsets = offsets; ws->list = list; ws->scratch = scratch; ws->block_sums = bs; }
  return off;
}

// ---------------------------------------------------------------------------------------------------------
// prepare
// ---------------------------------------------------------------------------------------------------------
// voxel of one point: idx = ((geom - (bx - dx/2)) / dx).long()   (lss.py:246): float32 subtract, IEEE divide, trunc toward zero;
// -1 = dropped (lss.py:253-255).  NaN / +-inf / huge values fail the range tests and are dropped, as in the reference.
__device__ __forceinline__ int voxel_key(float gx, float gy, float gz, int b, int nx, int ny, int nz, float ox, float oy, float oz,
                                         float dx, float dy, float dz) {
  const float vx = (gx - ox) / dx;
  const float vy = (gy - oy) / dy;
  const float vz = (gz - oz) / dz;
  const float lim = 1073741824.0f;
  if (!(vx > -lim && vx < lim && vy > -lim && vy < lim && vz > -lim && vz < lim)) return -1;
  const int ix = (int)vx, iy = (int)vy, iz = (int)vz;
  if (!(ix >= 0 && ix < nx && iy >= 0 && iy < ny && iz >= 0 && iz < nz)) return -1;
  return ((b * nz + iz) * nx + ix) * ny + iy;
}

__global__ void __launch_bounds__(256) splat_keys_kernel(const float* __restrict__ geom, int P, int n_per_sample, int nx, int ny,
                                                        int nz, float ox, float oy, float oz, float dx, float dy, float dz,
                                                        int* __restrict__ keys, int* __restrict__ count) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int key = voxel_key(geom[(size_t)p * 3 + 0], geom[(size_t)p * 3 + 1], geom[(size_t)p * 3 + 2], p / n_per_sample, nx, ny, nz,
                            ox, oy, oz, dx, dy, dz);
  if (key >= 0) atomicAdd(count + key, 1);
  keys[p] = key;
}

// get_geometry (lss.py:204-224) fused into the key pass: the ego-frame position of frustum point (u, v, d) of camera k is
//   q = (u, v, d) - post_trans;  a = inv(post_rots) q;  b = (a.x a.z, a.y a.z, a.z);  g = (rots inv(intrins)) b + trans
// evaluated in the reference's order with every product and sum rounded on its own (no FMA): the voxel index is a
// truncation, so a last-bit difference in g could move a point on a voxel face to the neighbour.  `cams` holds per camera
// post_trans[3], inv(post_rots)[9], (rots inv(intrins))[9], trans[3] (row-major, 24 floats; the two 3x3 products of
// 3x3 matrices stay on the host side of the boundary).  The [B,N,D,fH,fW,3] geometry tensor is never materialised.
__device__ __forceinline__ void mat_apply(const float* __restrict__ m, float p0, float p1, float p2, float* o) {
#pragma clang fp contract(off)
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = (m[i * 3 + 0] * p0 + m[i * 3 + 1] * p1) + m[i * 3 + 2] * p2;
}

__global__ void __launch_bounds__(256) splat_keys_frustum_kernel(const float* __restrict__ frustum, const float* __restrict__ cams,
                                                                int P, int n_per_sample, int pts_per_cam, int nx, int ny, int nz,
                                                                float ox, float oy, float oz, float dx, float dy, float dz,
                                                                int* __restrict__ keys, int* __restrict__ count) {
#pragma clang fp contract(off)
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int f = p % pts_per_cam;
  const float* cam = cams + (size_t)(p / pts_per_cam) * 24;
  const float q0 = frustum[(size_t)f * 3 + 0] - cam[0], q1 = frustum[(size_t)f * 3 + 1] - cam[1], q2 = frustum[(size_t)f * 3 + 2] - cam[2];
  float a[3], g[3];
  mat_apply(cam + 3, q0, q1, q2, a);
  mat_apply(cam + 12, a[0] * a[2], a[1] * a[2], a[2], g);
  const int key = voxel_key(g[0] + cam[21], g[1] + cam[22], g[2] + cam[23], p / n_per_sample, nx, ny, nz, ox, oy, oz, dx, dy, dz);
  if (key >= 0) atomicAdd(count + key, 1);
  keys[p] = key;
}

// exclusive scan, 3 passes: 2048 elements per 256-thread block
__global__ void __launch_bounds__(256) scan_block_kernel(const int* __restrict__ in, int n, int* __restrict__ out, int* __restrict__ block_sums) {
  __shared__ int wave_tot[4];
  const int base = blockIdx.x * 2048 + threadIdx.x * 8;
  int v[8], s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = (base + i < n) ? in[base + i] : 0; s += v[i]; }
  // inclusive scan of s across the wave, then across the 4 waves
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = s;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  int wave_off = 0;
  for (int w = 0; w < wave; ++w) wave_off += wave_tot[w];
  int run = wave_off + inc - s;
#pragma unroll
  for (int i = 0; i < 8; ++i) { if (base + i < n) out[base + i] = run; run += v[i]; }
  if (threadIdx.x == 255) block_sums[blockIdx.x] = wave_off + inc;
}

__global__ void __launch_bounds__(256) scan_sums_kernel(int* __restrict__ block_sums, int nblk, int* __restrict__ total_out) {
  // single workgroup: serial over chunks of 256 block sums (nblk <= a few thousand)
  __shared__ int wave_tot[4];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int c0 = 0; c0 < nblk; c0 += 256) {
    const int i = c0 + threadIdx.x;
    const int s = (i < nblk) ? block_sums[i] : 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    int wave_off = 0;
    for (int w = 0; w < wave; ++w) wave_off += wave_tot[w];
    const int carry = carry_s;
    if (i < nblk) block_sums[i] = carry + wave_off + inc - s;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = carry + wave_off + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_out = carry_s;
}

__global__ void __launch_bounds__(256) scan_add_kernel(int* __restrict__ out, int n, const int* __restrict__ block_sums) {
  const int base = blockIdx.x * 2048 + threadIdx.x * 8;
  const int add = block_sums[blockIdx.x];
#pragma unroll
  for (int i = 0; i < 8; ++i) if (base + i < n) out[base + i] += add;
}

__global__ void __launch_bounds__(256) splat_fill_kernel(const int* __restrict__ keys, int P, const int* __restrict__ offsets,
                                                        int* __restrict__ cursor, int* __restrict__ list) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int key = keys[p];
  if (key >= 0) list[offsets[key] + atomicAdd(cursor + key, 1)] = p;
}

// Order every voxel's CSR segment by point id (the atomic fill leaves it in arrival order): one thread per list slot
// computes the rank of its point inside its voxel's segment (segments are short: <= ~30 points at the reference's grids)
// and writes it to its sorted position in a second list.  Afterwards every voxel is summed in ascending point order =>
// bit-reproducible results with no sorting in the hot kernels.
__global__ void __launch_bounds__(256) splat_rank_kernel(const int* __restrict__ keys, const int* __restrict__ offsets, int V,
                                                        const int* __restrict__ list, int* __restrict__ sorted) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= offsets[V]) return;
  const int p = list[i];
  const int key = keys[p];
  const int s = offsets[key], e = offsets[key + 1];
  int rank = 0;
  for (int j = s; j < e; ++j) rank += (list[j] < p) ? 1 : 0;
  sorted[s + rank] = p;
}

// ---------------------------------------------------------------------------------------------------------
// round 5: the plan pass as FIVE launches (clear | keys + histogram + arrival slot | single-pass scan | fill | rank) instead of
// memset + seven -- at the config-4 shapes the pass is launch latency, not work (121 k points, 65 k voxels: ~25 us of kernels inside
// 180 us per call, profiles/r4j_bench_lift_splat.txt) -- and, for a camera rig, with both 3 x 3 inversions of get_geometry
// (lss.py:212, 218) inside the key kernel: no torch.inverse (two LU launches and a host synchronisation per plan) on the host side.
//   * the histogram atomic RETURNS the point's arrival slot inside its voxel: the CSR fill needs no second atomic pass;
//   * the exclusive scan is one kernel (decoupled look-back: a block publishes its aggregate, then its inclusive prefix; a wave of
//     the next block inspects 64 predecessors at once; block ids come from a ticket, so no dispatch order is assumed).
// Same keys, same CSR lists (ascending point ids inside a voxel) as the round-4 pass: every consumer kernel is unchanged.
// ---------------------------------------------------------------------------------------------------------
// count + scan state back to zero: a kernel, not hipMemsetAsync -- the pass also runs INSIDE captured train steps (a plan per forward under
// per-sample augmentation), and a memset node replayed by hipGraphLaunch did not clear the buffer (ROCm 7.2: the second replay ran on
// the first one's histogram and ticket -- out-of-range CSR slots, a look-back that never ends; found with tools/debug_c4_aug.py)
__global__ void __launch_bounds__(256) plan_clear_kernel(int4* __restrict__ p, int n4) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) p[i] = int4{0, 0, 0, 0};
}

__device__ __forceinline__ void emit_key(int p, int key, int* __restrict__ keys, int* __restrict__ count, int* __restrict__ slot) {
  keys[p] = key;
  if (key >= 0) slot[p] = atomicAdd(count + key, 1);      // arrival order inside the voxel (any order: the rank pass sorts by point id)
}

__global__ void __launch_bounds__(256) plan_keys_geom_kernel(const float* __restrict__ geom, int P, int n_per_sample, int nx, int ny, int nz,
                                                            float ox, float oy, float oz, float dx, float dy, float dz,
                                                            int* __restrict__ keys, int* __restrict__ count, int* __restrict__ slot) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  emit_key(p, voxel_key(geom[(size_t)p * 3 + 0], geom[(size_t)p * 3 + 1], geom[(size_t)p * 3 + 2], p / n_per_sample, nx, ny, nz, ox, oy, oz,
                        dx, dy, dz), keys, count, slot);
}

__device__ __forceinline__ int frustum_point_key(const float* __restrict__ frustum, const float* cam, int p, int f, int n_per_sample, int nx, int ny,
                                                 int nz, float ox, float oy, float oz, float dx, float dy, float dz) {
#pragma clang fp contract(off)
  const float q0 = frustum[(size_t)f * 3 + 0] - cam[0], q1 = frustum[(size_t)f * 3 + 1] - cam[1], q2 = frustum[(size_t)f * 3 + 2] - cam[2];
  float a[3], g[3];
  mat_apply(cam + 3, q0, q1, q2, a);
  mat_apply(cam + 12, a[0] * a[2], a[1] * a[2], a[2], g);
  return voxel_key(g[0] + cam[21], g[1] + cam[22], g[2] + cam[23], p / n_per_sample, nx, ny, nz, ox, oy, oz, dx, dy, dz);
}

__global__ void __launch_bounds__(256) plan_keys_cams_kernel(const float* __restrict__ frustum, const float* __restrict__ cams, int P, int n_per_sample,
                                                            int pts_per_cam, int nx, int ny, int nz, float ox, float oy, float oz, float dx,
                                                            float dy, float dz, int* __restrict__ keys, int* __restrict__ count,
                                                            int* __restrict__ slot) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  emit_key(p, frustum_point_key(frustum, cams + (size_t)(p / pts_per_cam) * 24, p, p % pts_per_cam, n_per_sample, nx, ny, nz, ox, oy, oz, dx, dy, dz),
           keys, count, slot);
}

// inverse of a 3 x 3 float32 matrix, formed in float64 (adjugate over the determinant) and rounded ONCE: within half an ulp of the
// exact inverse's entries.  The reference takes torch.inverse (an LU factorisation in float32 whose last bits differ between its CPU
// (LAPACK) and GPU (MAGMA / rocSOLVER) back ends); for the matrices a rig produces -- diagonal-plus-translation intrinsics, scale /
// flip / crop augmentations -- every such inverse is exact up to one rounding per entry and all of them agree.
__device__ __forceinline__ void inverse3_f32(const float* __restrict__ m, float* __restrict__ o) {
  const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
  const double A = e * i - f * h, B = c * h - b * i, C = b * f - c * e;
  const double D = f * g - d * i, E = a * i - c * g, F = c * d - a * f;
  const double G = d * h - e * g, H = b * g - a * h, I = a * e - b * d;
  const double idet = 1.0 / (a * A + b * D + c * G);
  o[0] = (float)(A * idet); o[1] = (float)(B * idet); o[2] = (float)(C * idet);
  o[3] = (float)(D * idet); o[4] = (float)(E * idet); o[5] = (float)(F * idet);
  o[6] = (float)(G * idet); o[7] = (float)(H * idet); o[8] = (float)(I * idet);
}
// the 24 coefficients of camera k as splat_keys_frustum_kernel reads them: post_trans, inv(post_rots), rots inv(intrins), trans
// (the 3 x 3 product in float32 with every product and sum rounded on its own, like a float32 matmul without FMA)
__device__ __forceinline__ void camera_coefficients(const float* __restrict__ rots, const float* __restrict__ trans, const float* __restrict__ intrins,
                                                    const float* __restrict__ post_rots, const float* __restrict__ post_trans, int k, float* __restrict__ cam) {
#pragma clang fp contract(off)
  float kinv[9];
  inverse3_f32(post_rots + (size_t)k * 9, cam + 3);
  inverse3_f32(intrins + (size_t)k * 9, kinv);
  const float* r = rots + (size_t)k * 9;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) cam[12 + i * 3 + j] = (r[i * 3 + 0] * kinv[0 * 3 + j] + r[i * 3 + 1] * kinv[1 * 3 + j]) + r[i * 3 + 2] * kinv[2 * 3 + j];
#pragma unroll
  for (int c = 0; c < 3; ++c) { cam[c] = post_trans[(size_t)k * 3 + c]; cam[21 + c] = trans[(size_t)k * 3 + c]; }
}

__global__ void __launch_bounds__(256) plan_keys_rig_kernel(const float* __restrict__ frustum, const float* __restrict__ rots, const float* __restrict__ trans,
                                                           const float* __restrict__ intrins, const float* __restrict__ post_rots,
                                                           const float* __restrict__ post_trans, int P, int n_per_sample, int pts_per_cam, int nx, int ny,
                                                           int nz, float ox, float oy, float oz, float dx, float dy, float dz, int* __restrict__ keys,
                                                           int* __restrict__ count, int* __restrict__ slot) {
  __shared__ float cam_s[2][24];
  const int p0 = blockIdx.x * blockDim.x, p1 = min(p0 + (int)blockDim.x - 1, P - 1);
  const int k0 = p0 / pts_per_cam, k1 = p1 / pts_per_cam;
  const int p = p0 + threadIdx.x;
  const bool two = k1 - k0 <= 1;      // the block's points belong to at most two cameras (always, for >= 256 points per camera)
  if (two) {
    if (threadIdx.x == 0) camera_coefficients(rots, trans, intrins, post_rots, post_trans, k0, cam_s[0]);
    if (threadIdx.x == 64) camera_coefficients(rots, trans, intrins, post_rots, post_trans, k1, cam_s[1]);
    __syncthreads();
  }
  if (p >= P) return;
  const int k = p / pts_per_cam;
  if (two) {
    emit_key(p, frustum_point_key(frustum, cam_s[k - k0 > 0 ? 1 : 0], p, p - k * pts_per_cam, n_per_sample, nx, ny, nz, ox, oy, oz, dx, dy, dz), keys, count, slot);
  } else {      // (its own call: one pointer that is LDS or a private array would put the array in scratch)
    float own[24];
    camera_coefficients(rots, trans, intrins, post_rots, post_trans, k, own);
    emit_key(p, frustum_point_key(frustum, own, p, p - k * pts_per_cam, n_per_sample, nx, ny, nz, ox, oy, oz, dx, dy, dz), keys, count, slot);
  }
}

// exclusive scan of `in[0 .. n)` into `out[0 .. n]` (out[n] = the total) in ONE launch.  2048 elements per 256-thread block; `state` =
// zero-filled words: state[0] = the block-id ticket, then one 64-bit word per block: flag (1 = aggregate, 2 = inclusive prefix) << 62 | value.
constexpr int kScanItems = 8;
__global__ void __launch_bounds__(256) plan_scan_kernel(const int* __restrict__ in, int n, int* __restrict__ out, unsigned long long* __restrict__ state) {
  __shared__ int wave_tot[4];
  __shared__ int bid_s, prev_s;
  if (threadIdx.x == 0) bid_s = (int)atomicAdd((unsigned*)state, 1u);
  __syncthreads();
  const int bid = bid_s;
  unsigned long long* st = state + 1;      // (64-bit words from state[1]; the host leaves 8 bytes for the ticket)
  const int base = bid * (256 * kScanItems) + threadIdx.x * kScanItems;
  int v[kScanItems], sum = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) { v[i] = (base + i < n) ? in[base + i] : 0; sum += v[i]; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  int wave_off = 0;
  for (int w = 0; w < wave; ++w) wave_off += wave_tot[w];
  const int block_total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
  if (threadIdx.x < 64) {      // wave 0: publish the aggregate, look back, publish the inclusive prefix
    if (lane == 0) __hip_atomic_store(st + bid, (bid == 0 ? 2ull : 1ull) << 62 | (unsigned long long)(unsigned)block_total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int prev = 0;
    int hi = bid - 1;            // nearest predecessor not yet accounted for
    while (hi >= 0) {
      const int j = hi - lane;
      unsigned long long w = 0;
      int polls = 0;      // (bounded: a predecessor that never publishes -- a corrupted state buffer -- must not hang the GPU; the plan is then garbage, not a strike)
      do {
        w = j >= 0 ? __hip_atomic_load(st + j, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : (2ull << 62);      // (before block 0: prefix 0)
      } while (__any((w >> 62) == 0) && ++polls < (1 << 22));
      const unsigned long long has_prefix = __ballot((w >> 62) == 2);
      const int first = __ffsll((long long)has_prefix) - 1;      // nearest lane holding an inclusive prefix (-1: none in this window)
      const int take = first < 0 ? 64 : first + 1;               // lanes 0 .. take - 1 contribute (aggregates, then the prefix)
      int c = lane < take ? (int)(unsigned)(w & 0xffffffffull) : 0;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
      prev += c;
      if (first >= 0) break;
      hi -= 64;
    }
    if (lane == 0) {
      if (bid > 0) __hip_atomic_store(st + bid, 2ull << 62 | (unsigned long long)(unsigned)(prev + block_total), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      prev_s = prev;
    }
  }
  __syncthreads();
  int run = prev_s + wave_off + inc - sum;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) { if (base + i < n) out[base + i] = run; run += v[i]; }
  if (base <= n - 1 && n - 1 < base + kScanItems) out[n] = run;      // the thread that holds the last element writes the total
}

// CSR fill without atomics: the key pass already gave every kept point its arrival slot inside its voxel
__global__ void __launch_bounds__(256) plan_fill_kernel(const int* __restrict__ keys, int P, const int* __restrict__ offsets, const int* __restrict__ slot,
                                                       int* __restrict__ list) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int key = keys[p];
  if (key >= 0) list[offsets[key] + slot[p]] = p;
}

template <typename S>
__device__ __forceinline__ S mul_rounded(S a, S b) {
#pragma clang fp contract(off)
  return a * b;          // rounded on its own, like the reference's materialised product, whatever the TU's contraction mode
}

// ---------------------------------------------------------------------------------------------------------
// segmented accumulation of a wave's points into its voxels of the LDS tile (both forward kernels)
// ---------------------------------------------------------------------------------------------------------
// The points of a wave's kFwdVox consecutive voxels are ONE contiguous, (voxel, point)-sorted range of `list`.  They are taken 64
// at a time; lane l of a chunk learns the voxel of point l by counting the CSR bounds at or below its index, and the rows are
// then streamed through a kFwdPipe-deep register pipeline in which EVERYTHING is unconditional: the prefetch index is clamped
// instead of guarded, a slot past the end contributes a selected zero to the last voxel, and "voxel finished" is not a branch --
// the running sum is written to the tile after every point (the last write of a voxel is its sum; empty voxels keep the zeros
// the tile was cleared with).  That is what lets the compiler count its loads: with guards and a voxel-closing `while` in the
// loop it waited `vmcnt(0)` before every use (23 of 26 waits), i.e. one L2 round trip per POINT, and the densest waves of the
// config-4 rig hold 256 points (46 -> 27 us fused, 75 -> 41 us plain at B = 1).  Sums run in ascending point order as before.
constexpr int kFwdVox = 16;      // voxels per wave (4 waves per 64-voxel tile)
constexpr int kFwdPipe = 16;     // row loads in flight per wave

template <typename S, bool WEIGHTED, typename RowIndex>
__device__ __forceinline__ void accumulate_tile_rows(const S* __restrict__ xc, int C, const int* __restrict__ list, const S* __restrict__ weight,
                                                     RowIndex row_of, int my_off, int start, int n, int v_lo, int v_hi, int lane,
                                                     S (*tile)[kTileVox + 1]) {
  for (int v = v_lo; v < v_lo + kFwdVox; ++v) tile[lane][v] = (S)0;
  const int bound_l = (lane <= v_hi - v_lo) ? my_off : 0x3fffffff;     // lanes past a ragged wave's last voxel: no bound
  S acc = (S)0;
  int prev_v = -1;
  for (int k0 = 0; k0 < n; k0 += 64) {
    const int m = min(64, n - k0);
    const int pid = (lane < m) ? list[start + k0 + lane] : 0;
    const int row = row_of(pid);
    S wl = (S)1;
    if constexpr (WEIGHTED) wl = (lane < m) ? weight[pid] : (S)0;
    // voxel of this lane's point = v_lo + number of CSR bounds <= its index; lanes past the end take the last point's voxel
    int vox = v_lo;
#pragma unroll
    for (int j = 1; j < kFwdVox; ++j) vox += (k0 + lane >= __builtin_amdgcn_readlane(bound_l, j) - start) ? 1 : 0;
    vox = (lane < m) ? vox : __builtin_amdgcn_readlane(vox, m - 1);
    S buf[kFwdPipe];
#pragma unroll
    for (int i = 0; i < kFwdPipe; ++i) buf[i] = xc[(size_t)__builtin_amdgcn_readlane(row, i) * C];
    for (int k = 0; k < m; k += kFwdPipe) {
#pragma unroll
      for (int i = 0; i < kFwdPipe; ++i) {
        const int p = min(k + i, 63);                       // wave-uniform slot
        S val = buf[i];
        buf[i] = xc[(size_t)__builtin_amdgcn_readlane(row, min(p + kFwdPipe, 63)) * C];
        if constexpr (WEIGHTED) val = mul_rounded(mf_readlane(wl, p), val);
        const int vi = __builtin_amdgcn_readlane(vox, p);
        const bool valid = k + i < m;
        acc = (vi != prev_v ? (S)0 : acc) + (valid ? val : (S)0);
        tile[lane][vi] = acc;
        prev_v = vi;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// forward, 16 lanes per point (round 4; float32, C % 4 == 0, plane % 4 == 0 -- the reference's shapes)
// ---------------------------------------------------------------------------------------------------------
// accumulate_tile_rows() above walks ONE point per wave iteration (lane = channel, a 4-byte load per lane): ~18 wave instructions
// per point, and the densest waves of the config-4 rig serialise 256 of them -- PMC (profiles/r4a_pmc_lift_splat.txt): the kernel
// moves exactly its output bytes (WRITE_SIZE = the BEV tensor, FETCH_SIZE ~ the inputs), no LDS bank conflicts, and sits at 6 %
// (B = 1) / 19 % (B = 8) of the HBM roofline: instruction issue and the dependent chain of a tile, not traffic.  Here a point is
// read by a QUARTER wave -- lane q of a 16-lane row loads channels 4q .. 4q + 3 as one 16-byte load -- so a wave instruction covers
// four points and a tile's points are spread over the workgroup's sixteen rows: row `gid` owns voxels gid, gid + 16, gid + 32, gid +
// 48 of the 64-voxel tile (round robin: the points of neighbouring dense voxels go to different rows).  Per row the points of its
// four voxels form one virtual stream, taken sixteen at a time: lane q looks up stream slot k0 + q (its voxel by comparing with the
// row's prefix counts, then list id, depth weight, row index -- all loads of a chunk are issued before the first is used), and the
// sixteen points are then accumulated in order with their row index / weight / flags broadcast inside the row (DPP row_share).
// Sums still run over ascending point ids per voxel with the product rounded on its own: bit-identical to the kernels above.
// The [voxel][channel] tile in LDS is rotated by 4 (v >> 2) channels per voxel row, so that the 16-byte writes of a row stay
// aligned and the transposing reads of the output phase (lane = 4 voxels of one channel -> one global_store_dwordx4) hit 64
// different banks.  Tiles without points skip everything and store zeros (most of a BEV plane at real camera rigs).
template <int CTRL>
__device__ __forceinline__ int row_share_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + CTRL, 0xF, 0xF, true); }
template <int CTRL>
__device__ __forceinline__ float row_share_f(float v) { return __builtin_bit_cast(float, row_share_i<CTRL>(__builtin_bit_cast(int, v))); }

constexpr int kQPitch = 68;      // words per voxel row of the LDS tile (64 channels + 4: rows stay 16-byte aligned)

// n / d for any 32-bit n by a multiply-high and two shifts (Granlund & Montgomery 1994, fig. 4.1): the fused kernels turn a frustum
// point id into (camera, pixel) with two divisions by launch constants -- ~35 instructions each as a hardware-less u32 division,
// four here.  The host forms (m, sh1, sh2) per launch.
struct FastDiv { unsigned m, sh1, sh2; };
static FastDiv fast_div_make(unsigned d) {
  unsigned s = 0;
  while ((1ull << s) < d) ++s;                                  // ceil(log2 d)
  FastDiv f;
  f.m = (unsigned)((((1ull << s) - d) << 32) / d + 1);
  f.sh1 = s < 1 ? s : 1;
  f.sh2 = s > 0 ? s - 1 : 0;
  return f;
}
__device__ __forceinline__ unsigned fast_div(unsigned n, const FastDiv& f) {
  const unsigned t = __umulhi(f.m, n);
  return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

// PIPE (round 4, second form of the point loop: eight rows in flight, eight waves per SIMD): see the comment at the loop.
template <bool WEIGHTED, bool PIPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PIPE ? 8 : 4, PIPE ? 8 : 4))) splat_fwd_quad_kernel(const float* __restrict__ x, const float* __restrict__ weight,
                                                            const int* __restrict__ offsets, const int* __restrict__ list, int C,
                                                            int plane, int tiles_per_plane, FastDiv div_dhw, FastDiv div_hw, int hw,
                                                            float* __restrict__ out) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) float tile[kTileVox * kQPitch];
  const int bz = blockIdx.x / tiles_per_plane;
  const int vid0 = (blockIdx.x % tiles_per_plane) * kTileVox;
  const int nvox = min(kTileVox, plane - vid0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int key0 = bz * plane + vid0;
  // CSR bounds of the tile's voxels: lane l holds offsets[key0 + l], every lane the tile's end
  const int off_l = offsets[key0 + min(lane, nvox)];
  const int off_end = offsets[key0 + nvox];
  const int off_beg = __builtin_amdgcn_readfirstlane(off_l);
  const int q = lane & 15, g = lane >> 4, gid = wave * 4 + g;
  const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
  if (__builtin_amdgcn_readfirstlane(off_end) == off_beg) {      // no points in this tile: zeros, straight from registers
    for (int c0 = 0; c0 < C; c0 += 64) {
      const int nch = min(64, C - c0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cc = wave * 16 + 4 * i + g;
        if (cc < nch) {
          float* o = out + ((size_t)bz * C + c0 + cc) * plane + vid0 + 4 * q;
          if (nvox == kTileVox) __builtin_nontemporal_store(zero4, reinterpret_cast<f4*>(o));
          else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (4 * q + j < nvox) o[j] = 0.f;
          }
        }
      }
    }
    return;
  }
  // this row's four voxels: prefix counts of its virtual point stream
  // (the bounds go through LDS -- one wave's LDS operations execute in order, no barrier: lane l's off_l, then the tile's end)
  __shared__ int offs_sh[4][kTileVox + 1];
  offs_sh[wave][lane] = off_l;
  offs_sh[wave][kTileVox] = off_end;
  int o0, o1, o2, o3, p1, p2, p3, p4, v0 = gid, v1 = gid + 16, v2 = gid + 32, v3 = gid + 48;
  unsigned long long occ = ~0ull;                                  // voxels whose LDS rows are written by their sums (PIPE)
  if constexpr (PIPE) {
    // The OCCUPIED voxels of the tile are dealt to the sixteen rows by rank (row r: the r-th, (r + 16)-th ... occupied voxel): a
    // camera rig fills ~9 of a tile's 64 voxels, sixteen points each, usually neighbours -- dealt by voxel index (the form below)
    // two of them meet in one row (a second chunk for the whole wave) while other waves idle: 1.77 chunks per tile on the slowest
    // wave at the config-4 rig, ~45 % of the slots in use.  Which voxel a row sums does not change any sum.
    const int cnt_l = lane < nvox ? offs_sh[wave][lane + 1] - off_l : 0;
    occ = __ballot(cnt_l > 0);
    const int n_occ = __popcll(occ);
    __shared__ int vox_sh[4][kTileVox];
    if (cnt_l > 0) vox_sh[wave][__popcll(occ & ((1ull << lane) - 1ull))] = lane;
    const bool h0 = gid < n_occ, h1 = gid + 16 < n_occ, h2 = gid + 32 < n_occ, h3 = gid + 48 < n_occ;
    v0 = h0 ? vox_sh[wave][gid] : 0; v1 = h1 ? vox_sh[wave][gid + 16] : 0; v2 = h2 ? vox_sh[wave][gid + 32] : 0; v3 = h3 ? vox_sh[wave][gid + 48] : 0;
    o0 = offs_sh[wave][v0]; o1 = offs_sh[wave][v1]; o2 = offs_sh[wave][v2]; o3 = offs_sh[wave][v3];
    p1 = h0 ? offs_sh[wave][v0 + 1] - o0 : 0;
    p2 = p1 + (h1 ? offs_sh[wave][v1 + 1] - o1 : 0);
    p3 = p2 + (h2 ? offs_sh[wave][v2 + 1] - o2 : 0);
    p4 = p3 + (h3 ? offs_sh[wave][v3 + 1] - o3 : 0);
  } else {
    o0 = offs_sh[wave][gid]; o1 = offs_sh[wave][gid + 16]; o2 = offs_sh[wave][gid + 32]; o3 = offs_sh[wave][gid + 48];
    p1 = gid < nvox ? offs_sh[wave][gid + 1] - o0 : 0;
    p2 = p1 + (gid + 16 < nvox ? offs_sh[wave][gid + 17] - o1 : 0);
    p3 = p2 + (gid + 32 < nvox ? offs_sh[wave][gid + 33] - o2 : 0);
    p4 = p3 + (gid + 48 < nvox ? offs_sh[wave][gid + 49] - o3 : 0);
  }
  const int n_row = p4;
  // chunks of sixteen stream slots until the busiest row of the wave is through
  int n_max = max(n_row, __shfl_xor(n_row, 16, 64));
  n_max = __builtin_amdgcn_readfirstlane(max(n_max, __shfl_xor(n_max, 32, 64)));
  // (PIPE, registers: the row's prefix counts / list offsets / voxels live in ONE register spread over lanes 0..11 of the row and
  //  come back through row_share operands)
  const int tab = q == 0 ? p1 : q == 1 ? p2 : q == 2 ? p3 : q == 3 ? p4 : q == 4 ? o0 : q == 5 ? o1 : q == 6 ? o2 : q == 7 ? o3
                : q == 8 ? v0 : q == 9 ? v1 : q == 10 ? v2 : v3;
  // fused lift: frustum point id ((cam * D + d) * hw + pixel) -> context row cam * hw + pixel
  auto lift_row = [&](int pid) {
    const unsigned cam = fast_div((unsigned)pid, div_dhw), pq = fast_div((unsigned)pid, div_hw);
    return (int)(cam * (unsigned)hw + ((unsigned)pid - pq * (unsigned)hw));
  };
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int cq = min(c0 + 4 * q, C - 4);                      // channel quad of this lane (chunks beyond C: clamped, not stored)
    // the row's own voxel rows start at zero (LDS executes a wave's operations in order: no barrier between these and the sums)
    // (PIPE: the rows of occupied voxels are written by whichever row sums them -- only the others are cleared here)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (!PIPE || !((occ >> (gid + 16 * j)) & 1ull)) *reinterpret_cast<f4*>(&tile[(gid + 16 * j) * kQPitch + 4 * q]) = zero4;
    f4 acc = zero4;
    // this lane's stream slot of the chunk at K0: its point id (slots past the end read a valid id and are masked) and flags.  Which
    // of the row's four voxels the slot falls into is decided by nested selects on scalars (indexing prefix / offset arrays with a
    // computed j -- or a lambda capturing them -- made the compiler keep them in scratch memory).
#define MF_SPLAT_SLOT(K0, PID, META)                                                                       \
    {                                                                                                      \
      const int sl = (K0) + q;                                                                             \
      const bool valid = sl < n_row;                                                                       \
      const bool c1 = sl >= p1, c2 = sl >= p2, c3 = sl >= p3;                                              \
      const int pj = c3 ? p3 : (c2 ? p2 : (c1 ? p1 : 0));                                                  \
      const int pn = c3 ? p4 : (c2 ? p3 : (c1 ? p2 : p1));                                                 \
      const int oj = c3 ? o3 : (c2 ? o2 : (c1 ? o1 : o0));                                                 \
      const int vj = gid + (c3 ? 48 : (c2 ? 32 : (c1 ? 16 : 0)));                                          \
      META = vj | (valid ? 0x100 : 0) | (sl + 1 == pn ? 0x400 : 0);                                        \
      PID = list[valid ? oj + (sl - pj) : off_beg];                                                        \
    }
    // (measured and not kept: the point ids of chunk k + 1 requested behind the row loads of chunk k -- one dependent round trip per
    //  chunk instead of two, but 138 registers instead of 122, i.e. three workgroups per CU instead of four: 25 -> 29 us at B = 1,
    //  71 -> 88 us at B = 8.  The kernel is bound by its instruction issue, which more resident waves fill better.)
    const char* const xrow = reinterpret_cast<const char*>(x) + (size_t)cq * 4;
    const unsigned xstride = (unsigned)C * 4u;
    if constexpr (PIPE) {
      // Round 4, second form of the point loop.  tools/ab_splat_fill.py: a plain fill of the B = 8 output takes 21 us, this kernel on
      // an EMPTY plan 28 us, on the rig's plan 61 us -- the stores are not the bound; a workgroup lives ~9 us, most of it the three to
      // five dependent round trips of its one or two chunks (CSR bounds -> point ids -> weights / rows), with four workgroups on a CU.
      // (Measured and dropped first: the three loads of a slot as three pipeline stages across chunks -- ids two chunks ahead,
      // every row register re-requested as soon as it is used, `vmcnt(17)` throughout: 60.4 -> 63.1 us at B = 8; rows hold one
      // or two chunks, there is nothing to overlap.)  What the kernel responds to is workgroups in flight (three instead of four per
      // CU: 25 -> 29 us, see above), so this form holds EIGHT rows of a chunk in registers instead of sixteen -- the register of
      // point i is re-requested for point i + 8 as soon as it has been used -- and fits eight waves per SIMD.
      // Instruction diet of the same round (the waves of a tile run at ~45 % slot occupancy, so what a slot costs matters): slots
      // past a row's end need NO masking -- they come after the row's last voxel has been stored, what they add to `acc` is never
      // written -- so they read the tile's first point (a valid row) and nothing is selected per point; the slot's row OFFSET is
      // formed once per chunk and reaches the sixteen lanes as the DPP operand of one add with the lane's own channel offset
      // (before: an id through a DPP move, then a quarter-rate v_mul_lo per point); "last point of its voxel" is the sign bit of
      // the slot's word.
      const unsigned xstride = (unsigned)C * 4u, cq4 = (unsigned)cq * 4u;
      for (int k0 = 0; k0 < n_max; k0 += 16) {
        int pid, meta_l;
        {
          const int sl = k0 + q;
          const int t1 = row_share_i<0>(tab), t2 = row_share_i<1>(tab), t3 = row_share_i<2>(tab), t4 = row_share_i<3>(tab);
          const int u0 = row_share_i<4>(tab), u1 = row_share_i<5>(tab), u2 = row_share_i<6>(tab), u3 = row_share_i<7>(tab);
          const bool valid = sl < t4;
          const bool c1 = sl >= t1, c2 = sl >= t2, c3 = sl >= t3;
          const int pj = c3 ? t3 : (c2 ? t2 : (c1 ? t1 : 0));
          const int pn = c3 ? t4 : (c2 ? t3 : (c1 ? t2 : t1));
          const int oj = c3 ? u3 : (c2 ? u2 : (c1 ? u1 : u0));
          const int w0 = row_share_i<8>(tab), w1 = row_share_i<9>(tab), w2 = row_share_i<10>(tab), w3 = row_share_i<11>(tab);
          const int vj = c3 ? w3 : (c2 ? w2 : (c1 ? w1 : w0));
          meta_l = vj | (sl + 1 == pn ? (int)0x80000000 : 0);      // (sl + 1 == pn implies a slot inside the row's stream)
          pid = list[valid ? oj + (sl - pj) : off_beg];
        }
        float w_l = 1.0f;
        int row_l = pid;
        if constexpr (WEIGHTED) { w_l = weight[pid]; row_l = lift_row(pid); }
        const int roff_l = (int)((unsigned)row_l * xstride);
        f4 val[8];
        // (loads pinned in program order: the scheduler issues independent loads last-first, and the first wait then covers them all)
#define MF_SPLAT_LOAD(R, I) val[R] = *reinterpret_cast<const f4*>(reinterpret_cast<const char*>(x) + (size_t)((unsigned)row_share_i<I>(roff_l) + cq4)); \
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_sched_barrier(0);
        MF_SPLAT_LOAD(0, 0) MF_SPLAT_LOAD(1, 1) MF_SPLAT_LOAD(2, 2) MF_SPLAT_LOAD(3, 3) MF_SPLAT_LOAD(4, 4) MF_SPLAT_LOAD(5, 5) MF_SPLAT_LOAD(6, 6) MF_SPLAT_LOAD(7, 7)
        // point I of the chunk from register R (nothing conditional in front of the store of a finished voxel: a load inside a
        // branch makes the wait-count pass wait for everything), then -- NEXT -- the register goes to point I + 8
#define MF_SPLAT_POINT8(R, I, NEXT)                                                                                \
        {                                                                                                          \
          const int meta = row_share_i<I>(meta_l);                                                                 \
          f4 pr = val[R];                                                                                          \
          if constexpr (WEIGHTED) {                                                                                \
            const float wi = row_share_f<I>(w_l);                                                                  \
            pr.x = mul_rounded(wi, pr.x); pr.y = mul_rounded(wi, pr.y); pr.z = mul_rounded(wi, pr.z); pr.w = mul_rounded(wi, pr.w); \
          }                                                                                                        \
          acc = acc + pr;                                    /* (a voxel's first point: 0 + x, as the kernels above) */ \
          NEXT                                                                                                     \
          if (meta < 0) {                                    /* last point of its voxel: its sum goes to the tile */ \
            const int v = meta & 0xff;                                                                             \
            *reinterpret_cast<f4*>(&tile[v * kQPitch + ((4 * q + 4 * (v >> 2)) & 63)]) = acc;                      \
            acc = zero4;                                                                                           \
          }                                                                                                        \
        }
        MF_SPLAT_POINT8(0, 0, MF_SPLAT_LOAD(0, 8)) MF_SPLAT_POINT8(1, 1, MF_SPLAT_LOAD(1, 9)) MF_SPLAT_POINT8(2, 2, MF_SPLAT_LOAD(2, 10))
        MF_SPLAT_POINT8(3, 3, MF_SPLAT_LOAD(3, 11)) MF_SPLAT_POINT8(4, 4, MF_SPLAT_LOAD(4, 12)) MF_SPLAT_POINT8(5, 5, MF_SPLAT_LOAD(5, 13))
        MF_SPLAT_POINT8(6, 6, MF_SPLAT_LOAD(6, 14)) MF_SPLAT_POINT8(7, 7, MF_SPLAT_LOAD(7, 15))
        MF_SPLAT_POINT8(0, 8, ) MF_SPLAT_POINT8(1, 9, ) MF_SPLAT_POINT8(2, 10, ) MF_SPLAT_POINT8(3, 11, )
        MF_SPLAT_POINT8(4, 12, ) MF_SPLAT_POINT8(5, 13, ) MF_SPLAT_POINT8(6, 14, ) MF_SPLAT_POINT8(7, 15, )
#undef MF_SPLAT_POINT8
#undef MF_SPLAT_LOAD
      }
    } else
    for (int k0 = 0; k0 < n_max; k0 += 16) {
      int pid, meta_l;
      MF_SPLAT_SLOT(k0, pid, meta_l)
      float w_l = 1.0f;
      int row_l = pid;
      if constexpr (WEIGHTED) { w_l = weight[pid]; row_l = lift_row(pid); }
      const char* xb = reinterpret_cast<const char*>(x) + (size_t)cq * 4;
      const unsigned rstride = (unsigned)C * 4u;
      f4 val[16];
#define MF_SPLAT_LOAD(I) val[I] = *reinterpret_cast<const f4*>(xb + (size_t)((unsigned)row_share_i<I>(row_l) * rstride));
      MF_SPLAT_LOAD(0) MF_SPLAT_LOAD(1) MF_SPLAT_LOAD(2) MF_SPLAT_LOAD(3) MF_SPLAT_LOAD(4) MF_SPLAT_LOAD(5) MF_SPLAT_LOAD(6) MF_SPLAT_LOAD(7)
      MF_SPLAT_LOAD(8) MF_SPLAT_LOAD(9) MF_SPLAT_LOAD(10) MF_SPLAT_LOAD(11) MF_SPLAT_LOAD(12) MF_SPLAT_LOAD(13) MF_SPLAT_LOAD(14) MF_SPLAT_LOAD(15)
#undef MF_SPLAT_LOAD
#define MF_SPLAT_POINT(I)                                                                                          \
      {                                                                                                            \
        const int meta = row_share_i<I>(meta_l);                                                                   \
        const float wi = row_share_f<I>(w_l);                                                                      \
        if (meta & 0x100) {                                  /* row-uniform: all sixteen lanes of the row agree */  \
          f4 pr = val[I];                                                                                          \
          if constexpr (WEIGHTED) { pr.x = mul_rounded(wi, pr.x); pr.y = mul_rounded(wi, pr.y); pr.z = mul_rounded(wi, pr.z); pr.w = mul_rounded(wi, pr.w); } \
          acc = acc + pr;                                    /* (a voxel's first point: 0 + x, as the kernels above) */ \
          if (meta & 0x400) {                                /* last point of its voxel: its sum goes to the tile, the next starts at 0 */ \
            const int v = meta & 0xff;                                                                             \
            *reinterpret_cast<f4*>(&tile[v * kQPitch + ((4 * q + 4 * (v >> 2)) & 63)]) = acc;                      \
            acc = zero4;                                                                                           \
          }                                                                                                        \
        }                                                                                                          \
      }
      {   // point 0 without a branch around its arithmetic: its row is then used unconditionally, so its load cannot be sunk into a
          // branch (a load inside a branch makes the wait-count pass wait for ALL sixteen; the other rows have LDS writes between
          // their load and their use and stay where they are)
        const int meta = row_share_i<0>(meta_l);
        const float wi = row_share_f<0>(w_l);
        f4 pr = val[0];
        if constexpr (WEIGHTED) { pr.x = mul_rounded(wi, pr.x); pr.y = mul_rounded(wi, pr.y); pr.z = mul_rounded(wi, pr.z); pr.w = mul_rounded(wi, pr.w); }
        const f4 sum = acc + pr;                             // (a voxel's first point: acc is 0, the sum is 0 + x as in the kernels above)
        acc = (meta & 0x100) ? sum : acc;
        if ((meta & 0x500) == 0x500) {
          const int v = meta & 0xff;
          *reinterpret_cast<f4*>(&tile[v * kQPitch + ((4 * q + 4 * (v >> 2)) & 63)]) = acc;
          acc = zero4;
        }
      }
      MF_SPLAT_POINT(1) MF_SPLAT_POINT(2) MF_SPLAT_POINT(3) MF_SPLAT_POINT(4) MF_SPLAT_POINT(5) MF_SPLAT_POINT(6) MF_SPLAT_POINT(7)
      MF_SPLAT_POINT(8) MF_SPLAT_POINT(9) MF_SPLAT_POINT(10) MF_SPLAT_POINT(11) MF_SPLAT_POINT(12) MF_SPLAT_POINT(13) MF_SPLAT_POINT(14) MF_SPLAT_POINT(15)
#undef MF_SPLAT_POINT
    }
    __syncthreads();
    // out[(bz * C + c) * plane + vid]: lane (g, q) = channel row wave * 16 + 4 i + g, voxels 4 q .. 4 q + 3 -> one 16-byte store
    // (the lane's coordinates pass through an empty asm: the output addresses are invariant in this loop over the channel chunks,
    //  the compiler hoisted all of them above the point loop and -- at 64 registers -- spilled them there)
    int qo = q, go = g;
    if constexpr (PIPE) asm volatile("" : "+v"(qo), "+v"(go));
    const int nch = min(64, C - c0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = qo, g = go;
      const int cc = wave * 16 + 4 * i + g;
      const int col = (cc + 4 * q) & 63;                         // rotation of voxel rows 4 q .. 4 q + 3: (v >> 2) = q
      f4 r;
      r.x = tile[(4 * q + 0) * kQPitch + col]; r.y = tile[(4 * q + 1) * kQPitch + col];
      r.z = tile[(4 * q + 2) * kQPitch + col]; r.w = tile[(4 * q + 3) * kQPitch + col];
      if (cc < nch) {
        float* o = out + ((size_t)bz * C + c0 + cc) * plane + vid0 + 4 * q;
        if (nvox == kTileVox) __builtin_nontemporal_store(r, reinterpret_cast<f4*>(o));
        else {      // a ragged last tile: element by element (no indexed copy of r: that would live in scratch memory)
          if (4 * q + 0 < nvox) o[0] = r.x;
          if (4 * q + 1 < nvox) o[1] = r.y;
          if (4 * q + 2 < nvox) o[2] = r.z;
          if (4 * q + 3 < nvox) o[3] = r.w;
        }
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------
// forward / backward tiles
// ---------------------------------------------------------------------------------------------------------
// A workgroup owns a tile of 64 consecutive voxels of one BEV plane, each of its 4 waves 16 of them.  The CSR segments of
// consecutive voxels are contiguous, so a wave's points are ONE contiguous, (voxel, point)-sorted range of `list`: it is
// fetched 64 ids at a time with one coalesced load, and the feature rows (lane = channel: one 256 B row per point) are
// streamed through the register pipeline of accumulate_tile_rows() above -- no per-voxel latency chain.
template <typename S>
__global__ void __launch_bounds__(256) splat_fwd_kernel(const S* __restrict__ x, const int* __restrict__ offsets,
                                                       const int* __restrict__ list, int C, int plane, int tiles_per_plane,
                                                       S* __restrict__ out) {
  __shared__ S tile[64][kTileVox + 1];      // [channel][voxel], +1 pad: conflict-free column writes and row reads
  const int bz = blockIdx.x / tiles_per_plane;                 // (sample, z-slab) plane
  const int vid0 = (blockIdx.x % tiles_per_plane) * kTileVox;  // first voxel of the tile inside the plane
  const int nvox = min(kTileVox, plane - vid0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int key0 = bz * plane + vid0;
  const int v_lo = wave * 16, v_hi = min(v_lo + 16, nvox);     // this wave's voxels (may be empty at a ragged plane end)
  // offsets of the wave's voxels: lane v holds offsets[first + v], v = 0..16
  const int my_off = (v_lo + lane <= nvox && lane <= 16) ? offsets[key0 + v_lo + lane] : 0;
  const int start = __builtin_amdgcn_readfirstlane(my_off);
  const int n = (v_hi > v_lo) ? __builtin_amdgcn_readlane(my_off, v_hi - v_lo) - start : 0;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = min(c0 + lane, C - 1);                       // lanes beyond C read a valid channel and are not stored
    accumulate_tile_rows<S, false>(x + c, C, list, (const S*)nullptr, [](int pid) { return pid; }, my_off, start, n, v_lo, v_hi, lane, tile);
    __syncthreads();
    // out[(bz*C + c) * plane + vid]: 64 consecutive voxels of one channel plane per store
    const int nch = min(64, C - c0);
    for (int cc = wave; cc < nch; cc += 4)
      if (lane < nvox) out[((size_t)bz * C + c0 + cc) * plane + vid0 + lane] = tile[cc][lane];
    __syncthreads();
  }
}

template <typename S>
__global__ void __launch_bounds__(256) splat_bwd_kernel(const S* __restrict__ gout, const int* __restrict__ offsets,
                                                       const int* __restrict__ list, int C, int plane, int tiles_per_plane,
                                                       S* __restrict__ gx) {
  __shared__ S tile[64][kTileVox + 1];
  const int bz = blockIdx.x / tiles_per_plane;
  const int vid0 = (blockIdx.x % tiles_per_plane) * kTileVox;
  const int nvox = min(kTileVox, plane - vid0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int key0 = bz * plane + vid0;
  // skip tiles without points (most of the BEV plane is outside the camera frusta)
  if (offsets[key0 + nvox] == offsets[key0]) return;
  const int v_lo = wave * 16, v_hi = min(v_lo + 16, nvox);
  const int my_off = (v_lo + lane <= nvox && lane <= 16) ? offsets[key0 + v_lo + lane] : 0;
  const int start = __builtin_amdgcn_readfirstlane(my_off);
  const int n = (v_hi > v_lo) ? __builtin_amdgcn_readlane(my_off, v_hi - v_lo) - start : 0;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int nch = min(64, C - c0);
    for (int cc = wave; cc < nch; cc += 4)
      tile[cc][lane] = (lane < nvox) ? gout[((size_t)bz * C + c0 + cc) * plane + vid0 + lane] : (S)0;
    __syncthreads();
    const int c = c0 + lane;
    int v = v_lo;
    int bound = (v_hi > v_lo) ? __builtin_amdgcn_readlane(my_off, 1) - start : 0;
    S g = tile[lane][v_lo < nvox ? v_lo : 0];
    for (int k0 = 0; k0 < n; k0 += 64) {
      const int m = min(64, n - k0);
      const int pid = (lane < m) ? list[start + k0 + lane] : 0;
      for (int k = 0; k < m; ++k) {
        if (k0 + k >= bound) {                       // wave-uniform: advance to the voxel that owns point k
          do { ++v; bound = __builtin_amdgcn_readlane(my_off, v - v_lo + 1) - start; } while (k0 + k >= bound);
          g = tile[lane][v];
        }
        if (c < C) gx[(size_t)__builtin_amdgcn_readlane(pid, k) * C + c] = g;     // one coalesced row per point
      }
    }
    __syncthreads();
  }
}

// rows of dropped points get zero gradient (x[kept] in the reference: no gradient reaches the others)
// (a wave looks at 64 keys with one coalesced load and writes the rows of the few dropped ones; one wave per POINT -- a quarter of a
//  million workgroups at B = 8, 94 % of which leave at once -- was 67 us there, more than half of the backward kernel itself)
template <typename S>
__global__ void __launch_bounds__(256) splat_bwd_zero_kernel(const int* __restrict__ keys, int P, int C, S* __restrict__ gx) {
  const int lane = threadIdx.x & 63;
  const int p0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
  const int p = p0 + lane;
  unsigned long long dropped = __ballot(p < P && keys[min(p, P - 1)] < 0);
  while (dropped) {
    const int k = __ffsll((long long)dropped) - 1;
    dropped &= dropped - 1;
    S* row = gx + (size_t)(p0 + k) * C;
    for (int c = lane; c < C; c += 64) row[c] = (S)0;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Lift fused into the splat (lss.py:63-71 `depth.unsqueeze(1) * context.unsqueeze(2)` + :238-280).
// A frustum point p = ((cam * D + d) * fHW + pixel) never materialises its C-channel feature row: the forward reads the
// point's depth probability depth[p] and the pixel's context row ctx[cam * fHW + pixel][C] (pixel-major, 0.5 MB per sample,
// cache-resident) and accumulates depth * ctx; 17.8 MB of traffic per sample instead of 49.2 MB.
// ---------------------------------------------------------------------------------------------------------

template <typename S>
__global__ void __launch_bounds__(256) lift_splat_fwd_kernel(const S* __restrict__ depth, const S* __restrict__ ctx,
                                                            const int* __restrict__ offsets, const int* __restrict__ list, int C,
                                                            int plane, int tiles_per_plane, int d_hw, int hw, S* __restrict__ out) {
  __shared__ S tile[64][kTileVox + 1];
  const int bz = blockIdx.x / tiles_per_plane;
  const int vid0 = (blockIdx.x % tiles_per_plane) * kTileVox;
  const int nvox = min(kTileVox, plane - vid0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int key0 = bz * plane + vid0;
  const int v_lo = wave * 16, v_hi = min(v_lo + 16, nvox);
  const int my_off = (v_lo + lane <= nvox && lane <= 16) ? offsets[key0 + v_lo + lane] : 0;
  const int start = __builtin_amdgcn_readfirstlane(my_off);
  const int n = (v_hi > v_lo) ? __builtin_amdgcn_readlane(my_off, v_hi - v_lo) - start : 0;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = min(c0 + lane, C - 1);
    // a point's row is its pixel's context row, its weight its depth probability
    accumulate_tile_rows<S, true>(ctx + c, C, list, depth, [=](int pid) { return (pid / d_hw) * hw + pid % hw; }, my_off, start, n, v_lo,
                                  v_hi, lane, tile);
    __syncthreads();
    const int nch = min(64, C - c0);
    for (int cc = wave; cc < nch; cc += 4)
      if (lane < nvox) out[((size_t)bz * C + c0 + cc) * plane + vid0 + lane] = tile[cc][lane];
    __syncthreads();
  }
}

// Backward, step 1: the BEV gradient [B*nz][C][plane] re-laid voxel-major, gT[key][C], for the tiles that hold points (the
// gather below reads whole 256-byte rows; empty tiles are never read).
template <typename S>
__global__ void __launch_bounds__(256) lift_splat_bwd_rows_kernel(const S* __restrict__ gout, const int* __restrict__ offsets, int C,
                                                                 int plane, int tiles_per_plane, S* __restrict__ gT) {
  __shared__ S tile[64][kTileVox + 1];
  const int bz = blockIdx.x / tiles_per_plane;
  const int vid0 = (blockIdx.x % tiles_per_plane) * kTileVox;
  const int nvox = min(kTileVox, plane - vid0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int key0 = bz * plane + vid0;
  if (offsets[key0 + nvox] == offsets[key0]) return;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int nch = min(64, C - c0);
    for (int cc = wave; cc < nch; cc += 4)
      tile[cc][lane] = (lane < nvox) ? gout[((size_t)bz * C + c0 + cc) * plane + vid0 + lane] : (S)0;
    __syncthreads();
    for (int vv = wave; vv < nvox; vv += 4)
      if (lane < nch) gT[(size_t)(key0 + vv) * C + c0 + lane] = tile[lane][vv];
    __syncthreads();
  }
}

// Backward, step 2: one wave per (camera, pixel), lane = channel.  Along the pixel's D depth bins:
//   g_depth[p] = <ctx[pixel], gT[voxel(p)]>   (wave reduction),   g_ctx[pixel] += depth[p] * gT[voxel(p)]   (registers);
// points the pooling dropped contribute nothing (x[kept] in the reference).  No atomics, every output written once.
template <typename S>
__global__ void __launch_bounds__(256) lift_splat_bwd_gather_kernel(const S* __restrict__ depth, const S* __restrict__ ctx,
                                                                   const int* __restrict__ keys, const S* __restrict__ gT, int C, int D,
                                                                   int hw, int n_pix, S* __restrict__ g_depth, S* __restrict__ g_ctx) {
  const int lane = threadIdx.x & 63;
  const int pix = blockIdx.x * 4 + (threadIdx.x >> 6);          // cam * hw + pixel, over all samples and cameras
  if (pix >= n_pix) return;
  const int cam = pix / hw, px = pix % hw;
  const size_t p0 = (size_t)cam * D * hw + px;                  // point index of depth bin 0
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + lane;
    const bool on = c < C;
    const S f = on ? ctx[(size_t)pix * C + c] : (S)0;
    S acc = (S)0;
    for (int d0 = 0; d0 < D; d0 += 64) {
      const int md = min(64, D - d0);
      const int key_l = (lane < md) ? keys[p0 + (size_t)(d0 + lane) * hw] : -1;
      const S dep_l = (lane < md) ? depth[p0 + (size_t)(d0 + lane) * hw] : (S)0;
      // the gradient rows of the bins' voxels through a kPipeBwd-deep pipeline (one L2 round trip each; unpipelined, the 59
      // dependent round trips of a pixel were the kernel); dropped points read row 0 and are masked
      const int cl = on ? c : 0;
      S buf[kPipeBwd];
#pragma unroll
      for (int i = 0; i < kPipeBwd; ++i) buf[i] = gT[(size_t)max(__builtin_amdgcn_readlane(key_l, i), 0) * C + cl];
      // (everything in the loop is unconditional -- clamped prefetch index, masked contributions -- so that the compiler can count
      // its loads; with guards around them it waited for ALL outstanding loads before every use)
      for (int dd = 0; dd < md; dd += kPipeBwd) {
#pragma unroll
        for (int i = 0; i < kPipeBwd; ++i) {
          const int d = min(dd + i, 63);
          const bool valid = dd + i < md;
          const int key = __builtin_amdgcn_readlane(key_l, d);    // wave-uniform; -1 for dropped points and slots past the end
          const S g = (key >= 0 && on) ? buf[i] : (S)0;
          buf[i] = gT[(size_t)max(__builtin_amdgcn_readlane(key_l, min(d + kPipeBwd, 63)), 0) * C + cl];
          acc += mf_readlane(dep_l, d) * g;
          const S dot = group_sum<64>(f * g);
          if (valid && lane == 0) {
            const size_t p = p0 + (size_t)(d0 + d) * hw;
            g_depth[p] = (c0 == 0) ? dot : g_depth[p] + dot;      // channel chunks beyond the first accumulate (same lane, in order)
          }
        }
      }
    }
    if (on) g_ctx[(size_t)pix * C + c] = acc;
  }
}

// ---------------------------------------------------------------------------------------------------------
// fused backward, 16 lanes per pixel / per voxel row (round 4; float32, C % 4 == 0, plane % 4 == 0)
// ---------------------------------------------------------------------------------------------------------
// Step 1, voxel-major rows of the BEV gradient for the voxels that HOLD points (the gather below reads nothing else): the tile comes in
// as 16-byte loads (lane = four voxels of one channel), is transposed through the rotated LDS tile of splat_fwd_quad_kernel, and every
// row of the workgroup writes the 256-byte rows of its occupied voxels -- 7 k of a plane's 65 k voxels at the config-4 rig, where the
// kernel above writes all 64 rows of every occupied tile (13 -> 1.8 MB of writes per sample).
__global__ void __launch_bounds__(256) lift_splat_bwd_rows_quad_kernel(const float* __restrict__ gout, const int* __restrict__ offsets, int C,
                                                                      int plane, int tiles_per_plane, float* __restrict__ gT) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) float tile[kTileVox * kQPitch];
  const int bz = blockIdx.x / tiles_per_plane;
  const int vid0 = (blockIdx.x % tiles_per_plane) * kTileVox;
  const int nvox = min(kTileVox, plane - vid0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int key0 = bz * plane + vid0;
  const int off_l = offsets[key0 + min(lane, nvox)];
  const int off_end = offsets[key0 + nvox];
  if (__builtin_amdgcn_readfirstlane(off_end) == __builtin_amdgcn_readfirstlane(off_l)) return;      // no points: the gather never reads this tile
  const int q = lane & 15, g = lane >> 4, gid = wave * 4 + g;
  __shared__ int offs_sh[4][kTileVox + 1];      // (one wave's LDS operations execute in order: no barrier)
  offs_sh[wave][lane] = off_l;
  offs_sh[wave][kTileVox] = off_end;
  bool occ[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int v = gid + 16 * j;
    occ[j] = v < nvox && offs_sh[wave][v + 1] > offs_sh[wave][v];
  }
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int nch = min(64, C - c0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cc = wave * 16 + 4 * i + g;
      f4 r = {0.f, 0.f, 0.f, 0.f};
      if (cc < nch) {
        const float* o = gout + ((size_t)bz * C + c0 + cc) * plane + vid0 + 4 * q;
        if (nvox == kTileVox) r = *reinterpret_cast<const f4*>(o);
        else { r.x = 4 * q + 0 < nvox ? o[0] : 0.f; r.y = 4 * q + 1 < nvox ? o[1] : 0.f; r.z = 4 * q + 2 < nvox ? o[2] : 0.f; r.w = 4 * q + 3 < nvox ? o[3] : 0.f; }
      }
      const int col = (cc + 4 * q) & 63;                         // rotation of voxel rows 4 q .. 4 q + 3 (v >> 2 = q)
      tile[(4 * q + 0) * kQPitch + col] = r.x; tile[(4 * q + 1) * kQPitch + col] = r.y;
      tile[(4 * q + 2) * kQPitch + col] = r.z; tile[(4 * q + 3) * kQPitch + col] = r.w;
    }
    __syncthreads();
    const int cq = c0 + 4 * q;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int v = gid + 16 * j;
      if (occ[j] && cq < C) {
        const f4 r = *reinterpret_cast<const f4*>(&tile[v * kQPitch + ((4 * q + 4 * (v >> 2)) & 63)]);
        *reinterpret_cast<f4*>(gT + (size_t)(key0 + v) * C + cq) = r;
      }
    }
    __syncthreads();
  }
}

// The plain splat's backward (QuickCumsum.backward: every kept point receives the gradient row of its voxel), sixteen lanes per point
// (round 4; float32, C % 4 == 0, plane % 4 == 0).  splat_bwd_kernel above walks ONE point per wave iteration with 16 voxels per
// wave -- at the config-4 rig a tile's ~140 points sit in ~9 neighbouring voxels, i.e. in one or two of its four waves (122 us at
// B = 8 for 247 MB of rows).  Here the tile comes in through the rotated [voxel][channel] LDS tile of the kernel above, and the tile's
// point list -- ONE contiguous CSR range -- is dealt to the waves in chunks of 64 whatever the voxels: a wave reads 64 point ids
// and their voxels (the key array of the plan) with two coalesced loads, then four points per instruction: a 16-byte LDS read of
// the voxel's row and a 16-byte store per lane.  A pure gather: no summation order to keep.
__global__ void __launch_bounds__(256) splat_bwd_quad_kernel(const float* __restrict__ gout, const int* __restrict__ offsets,
                                                            const int* __restrict__ list, const int* __restrict__ keys, int C, int plane,
                                                            int tiles_per_plane, float* __restrict__ gx) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) float tile[kTileVox * kQPitch];
  const int bz = blockIdx.x / tiles_per_plane;
  const int vid0 = (blockIdx.x % tiles_per_plane) * kTileVox;
  const int nvox = min(kTileVox, plane - vid0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int key0 = bz * plane + vid0;
  const int off_beg = offsets[key0], off_end = offsets[key0 + nvox];      // (scalar loads: the tile's CSR range)
  if (off_end == off_beg) return;
  const int n = off_end - off_beg;
  const int q = lane & 15, g = lane >> 4;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int nch = min(64, C - c0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cc = wave * 16 + 4 * i + g;
      f4 r = {0.f, 0.f, 0.f, 0.f};
      if (cc < nch) {
        const float* o = gout + ((size_t)bz * C + c0 + cc) * plane + vid0 + 4 * q;
        if (nvox == kTileVox) r = *reinterpret_cast<const f4*>(o);
        else { r.x = 4 * q + 0 < nvox ? o[0] : 0.f; r.y = 4 * q + 1 < nvox ? o[1] : 0.f; r.z = 4 * q + 2 < nvox ? o[2] : 0.f; r.w = 4 * q + 3 < nvox ? o[3] : 0.f; }
      }
      const int col = (cc + 4 * q) & 63;                         // rotation of voxel rows 4 q .. 4 q + 3 (v >> 2 = q)
      tile[(4 * q + 0) * kQPitch + col] = r.x; tile[(4 * q + 1) * kQPitch + col] = r.y;
      tile[(4 * q + 2) * kQPitch + col] = r.z; tile[(4 * q + 3) * kQPitch + col] = r.w;
    }
    __syncthreads();
    const int cq = c0 + 4 * q;
    for (int k0 = wave * 64; k0 < n; k0 += 256) {
      const int m = min(64, n - k0);
      const int pid_l = list[off_beg + k0 + min(lane, m - 1)];
      const int v_l = keys[pid_l] - key0;                        // the point's voxel inside the tile
#pragma unroll 4
      for (int i = 0; i < 16; ++i) {
        const int idx = 4 * i + g;                               // this row's point of the chunk
        const int pid = __shfl(pid_l, idx, 64), v = __shfl(v_l, idx, 64);
        if (idx < m && cq < C) {
          const f4 r = *reinterpret_cast<const f4*>(&tile[v * kQPitch + ((4 * q + 4 * (v >> 2)) & 63)]);
          __builtin_nontemporal_store(r, reinterpret_cast<f4*>(gx + (size_t)pid * C + cq));
        }
        if (4 * i + 4 >= m) break;                               // (wave-uniform)
      }
    }
    __syncthreads();
  }
}

// Step 2, sixteen lanes per (camera, pixel), lane q = channels 4 q .. 4 q + 3: along the pixel's D depth bins
//   g_depth[p] = <ctx[pixel], gT[voxel(p)]>  (four products per lane, a 16-lane DPP sum),  g_ctx[pixel] += depth[p] * gT[voxel(p)];
// a wave instruction covers four pixels (the one-pixel-per-wave kernel above spends ~25 instructions per point, 6 of them the 64-lane
// sum).  Bins are taken sixteen at a time: lane q looks up bin d0 + q (voxel key, depth weight), all sixteen row loads are issued
// before the first is used, and lane q keeps the dot product of "its" bin for one strided store per chunk.
__global__ void __launch_bounds__(256) lift_splat_bwd_gather_quad_kernel(const float* __restrict__ depth, const float* __restrict__ ctx,
                                                                        const int* __restrict__ keys, const float* __restrict__ gT, int C, int D,
                                                                        int hw, int n_pix, float* __restrict__ g_depth, float* __restrict__ g_ctx) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, q = lane & 15;
  const int pix = (blockIdx.x * 256 + (int)threadIdx.x) >> 4;   // cam * hw + pixel, over all samples and cameras
  const bool live = pix < n_pix;                                // (whole rows: DPP never crosses a row)
  const int pc = live ? pix : 0;
  const int cam = pc / hw, px = pc % hw;
  const size_t p0 = (size_t)cam * D * hw + px;                  // point index of depth bin 0
  const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int cq = c0 + 4 * q;
    const bool on = cq < C;
    const int cl = on ? cq : 0;
    const f4 f = on ? *reinterpret_cast<const f4*>(ctx + (size_t)pc * C + cq) : zero4;
    f4 acc = zero4;
    for (int d0 = 0; d0 < D; d0 += 16) {
      const bool has = d0 + q < D;
      const size_t pl = p0 + (size_t)min(d0 + q, D - 1) * hw;
      const int key_l = has ? keys[pl] : -1;                     // -1: dropped by the pooling, or past the last bin
      const float dep_l = has ? depth[pl] : 0.f;
      f4 val[16];
#define MF_GLOAD(I) val[I] = *reinterpret_cast<const f4*>(gT + (size_t)max(row_share_i<I>(key_l), 0) * C + cl);
      MF_GLOAD(0) MF_GLOAD(1) MF_GLOAD(2) MF_GLOAD(3) MF_GLOAD(4) MF_GLOAD(5) MF_GLOAD(6) MF_GLOAD(7)
      MF_GLOAD(8) MF_GLOAD(9) MF_GLOAD(10) MF_GLOAD(11) MF_GLOAD(12) MF_GLOAD(13) MF_GLOAD(14) MF_GLOAD(15)
#undef MF_GLOAD
      float dots = 0.f;
#define MF_GPOINT(I)                                                                                                      \
      {                                                                                                                   \
        const bool kept = row_share_i<I>(key_l) >= 0 && on;                                                               \
        const float di = row_share_f<I>(dep_l);                                                                           \
        const f4 gq = kept ? val[I] : zero4;                                                                              \
        acc.x += di * gq.x; acc.y += di * gq.y; acc.z += di * gq.z; acc.w += di * gq.w;                                   \
        float dsum = (f.x * gq.x + f.y * gq.y) + (f.z * gq.z + f.w * gq.w);                                               \
        dsum += dpp_mov<0xB1>(dsum); dsum += dpp_mov<0x4E>(dsum); dsum += dpp_mov<0x141>(dsum); dsum += dpp_mov<0x140>(dsum); \
        dots = q == I ? dsum : dots;                                                                                      \
      }
      MF_GPOINT(0) MF_GPOINT(1) MF_GPOINT(2) MF_GPOINT(3) MF_GPOINT(4) MF_GPOINT(5) MF_GPOINT(6) MF_GPOINT(7)
      MF_GPOINT(8) MF_GPOINT(9) MF_GPOINT(10) MF_GPOINT(11) MF_GPOINT(12) MF_GPOINT(13) MF_GPOINT(14) MF_GPOINT(15)
#undef MF_GPOINT
      if (live && has) {
        const size_t p = p0 + (size_t)(d0 + q) * hw;
        g_depth[p] = (c0 == 0) ? dots : g_depth[p] + dots;      // channel chunks beyond the first accumulate (same lane, in order)
      }
    }
    if (live && on) *reinterpret_cast<f4*>(g_ctx + (size_t)pc * C + cq) = acc;
  }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
static int check_desc(const MfSplatDesc* d) {
  MF_REQUIRE(d, MF_ERR_INVALID, "bev_splat: null descriptor");
  MF_REQUIRE(d->B > 0 && d->n_per_sample > 0 && d->C > 0 && d->nx > 0 && d->ny > 0 && d->nz > 0, MF_ERR_INVALID,
             "bev_splat: B, n_per_sample, C, nx, ny, nz must be positive");
  MF_REQUIRE((long long)d->B * d->n_per_sample < (1ll << 31) && (long long)d->B * d->nz * d->nx * d->ny < (1ll << 31) - 4096,
             MF_ERR_UNSUPPORTED, "bev_splat: more than 2^31 points or voxels");
  MF_REQUIRE(d->dx[0] != 0.f && d->dx[1] != 0.f && d->dx[2] != 0.f, MF_ERR_INVALID, "bev_splat: zero voxel size");
  return MF_OK;
}

#define MF_LAUNCH_OK(what)                                                                             \
  do {                                                                                                 \
    hipError_t e_ = hipGetLastError();                                                                 \
    MF_REQUIRE(e_ == hipSuccess, MF_ERR_LAUNCH, std::string(what) + ": " + hipGetErrorString(e_));     \
  } while (0)

struct RigPtrs { const float *rots, *trans, *intrins, *post_rots, *post_trans; };

// MF_SPLAT_PLAN=0: the round-4 pass (seven launches; A/B runs, equality tests of the two)
static bool plan_v2() {
  static const bool off = getenv("MF_SPLAT_PLAN") && atoi(getenv("MF_SPLAT_PLAN")) == 0;
  return !off;
}

static int splat_prepare(const MfSplatDesc* d, const float* geom, const float* frustum, const float* cams, int pts_per_cam,
                         void* workspace, hipStream_t st, const RigPtrs* rig = nullptr) {
  SplatWs ws;
  carve(d, workspace, &ws);
  const int P = d->B * d->n_per_sample;
  const int V = d->B * d->nz * d->nx * d->ny;
  const dim3 gp((P + 255) / 256), blk(256);
  if (plan_v2() || rig) {
    {      // count + cursor (= the scan's ticket and block states) to zero; both regions are 256-byte multiples
      const int n4 = (int)(((char*)ws.offsets - (char*)ws.count) / 16);
      hipLaunchKernelGGL(plan_clear_kernel, dim3(std::min((n4 + 255) / 256, 1024)), blk, 0, st, (int4*)ws.count, n4);
      MF_LAUNCH_OK("splat_clear");
    }
    // keys + histogram + arrival slots (slots kept in `list` until the rank pass overwrites it with the sorted ids)
    if (geom)
      hipLaunchKernelGGL(plan_keys_geom_kernel, gp, blk, 0, st, geom, P, d->n_per_sample, d->nx, d->ny, d->nz, d->off[0], d->off[1], d->off[2],
                         d->dx[0], d->dx[1], d->dx[2], ws.keys, ws.count, ws.list);
    else if (rig)
      hipLaunchKernelGGL(plan_keys_rig_kernel, gp, blk, 0, st, frustum, rig->rots, rig->trans, rig->intrins, rig->post_rots, rig->post_trans, P,
                         d->n_per_sample, pts_per_cam, d->nx, d->ny, d->nz, d->off[0], d->off[1], d->off[2], d->dx[0], d->dx[1], d->dx[2], ws.keys,
                         ws.count, ws.list);
    else
      hipLaunchKernelGGL(plan_keys_cams_kernel, gp, blk, 0, st, frustum, cams, P, d->n_per_sample, pts_per_cam, d->nx, d->ny, d->nz, d->off[0],
                         d->off[1], d->off[2], d->dx[0], d->dx[1], d->dx[2], ws.keys, ws.count, ws.list);
    MF_LAUNCH_OK("splat_keys");
    const int nblk = (V + 256 * kScanItems - 1) / (256 * kScanItems);
    // the scan's state lives in the (zeroed) cursor region: 8 bytes of ticket + one 64-bit word per block
    MF_REQUIRE((size_t)(nblk + 1) * 8 <= (size_t)((char*)ws.offsets - (char*)ws.cursor), MF_ERR_UNSUPPORTED, "bev_splat: scan state does not fit the workspace");
    hipLaunchKernelGGL(plan_scan_kernel, dim3(nblk), blk, 0, st, ws.count, V, ws.offsets, (unsigned long long*)ws.cursor);
    MF_LAUNCH_OK("splat_scan");
    hipLaunchKernelGGL(plan_fill_kernel, gp, blk, 0, st, ws.keys, P, ws.offsets, ws.list, ws.scratch);
    MF_LAUNCH_OK("splat_fill");
    hipLaunchKernelGGL(splat_rank_kernel, gp, blk, 0, st, ws.keys, ws.offsets, V, ws.scratch, ws.list);
    MF_LAUNCH_OK("splat_sort");
    return MF_OK;
  }
  {      // count + cursor to zero (a kernel: see plan_clear_kernel)
    const int n4 = (int)(((char*)ws.offsets - (char*)ws.count) / 16);
    hipLaunchKernelGGL(plan_clear_kernel, dim3(std::min((n4 + 255) / 256, 1024)), blk, 0, st, (int4*)ws.count, n4);
    MF_LAUNCH_OK("splat_clear");
  }
  if (geom)
    hipLaunchKernelGGL(splat_keys_kernel, dim3((P + 255) / 256), dim3(256), 0, st, geom, P, d->n_per_sample, d->nx, d->ny, d->nz,
                       d->off[0], d->off[1], d->off[2], d->dx[0], d->dx[1], d->dx[2], ws.keys, ws.count);
  else
    hipLaunchKernelGGL(splat_keys_frustum_kernel, dim3((P + 255) / 256), dim3(256), 0, st, frustum, cams, P, d->n_per_sample,
                       pts_per_cam, d->nx, d->ny, d->nz, d->off[0], d->off[1], d->off[2], d->dx[0], d->dx[1], d->dx[2], ws.keys,
                       ws.count);
  MF_LAUNCH_OK("splat_keys");
  const int nblk = (V + 2047) / 2048;
  hipLaunchKernelGGL(scan_block_kernel, dim3(nblk), dim3(256), 0, st, ws.count, V, ws.offsets, ws.block_sums);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(256), 0, st, ws.block_sums, nblk, ws.offsets + V);
  hipLaunchKernelGGL(scan_add_kernel, dim3(nblk), dim3(256), 0, st, ws.offsets, V, ws.block_sums);
  MF_LAUNCH_OK("splat_scan");
  hipLaunchKernelGGL(splat_fill_kernel, dim3((P + 255) / 256), dim3(256), 0, st, ws.keys, P, ws.offsets, ws.cursor, ws.scratch);
  MF_LAUNCH_OK("splat_fill");
  hipLaunchKernelGGL(splat_rank_kernel, dim3((P + 255) / 256), dim3(256), 0, st, ws.keys, ws.offsets, V, ws.scratch, ws.list);
  MF_LAUNCH_OK("splat_sort");
  return MF_OK;
}

// the 16-lanes-per-point forward: float32 rows of whole channel quads, 16-byte aligned output rows; MF_SPLAT_QUAD=0 keeps the
// one-point-per-iteration kernels (A/B runs, bit-equality tests of the two)
static bool quad_pipe() {      // MF_SPLAT_PIPE=0: the two-round-trips-per-chunk point loop (A/B runs)
  static const bool off = getenv("MF_SPLAT_PIPE") && atoi(getenv("MF_SPLAT_PIPE")) == 0;
  return !off;
}
static bool quad_forward_ok(const MfSplatDesc* d, int plane) {
  static const bool off = getenv("MF_SPLAT_QUAD") && atoi(getenv("MF_SPLAT_QUAD")) == 0;
  return !off && d->C % 4 == 0 && plane % 4 == 0 && (long long)d->C * 4 * ((long long)d->B * d->n_per_sample) < (1ll << 32);
}

template <typename S>
static int splat_fwd(const MfSplatDesc* d, const S* x, const void* workspace, S* out, hipStream_t st) {
  SplatWs ws;
  carve(d, const_cast<void*>(workspace), &ws);
  const int plane = d->nx * d->ny;
  const int tpp = (plane + kTileVox - 1) / kTileVox;
  if constexpr (sizeof(S) == 4) {
    if (quad_forward_ok(d, plane)) {      // sixteen lanes per point (splat_fwd_quad_kernel)
      const FastDiv one = fast_div_make(1);
      if (quad_pipe())
        hipLaunchKernelGGL((splat_fwd_quad_kernel<false, true>), dim3(d->B * d->nz * tpp), dim3(256), 0, st, x, (const float*)nullptr, ws.offsets,
                           ws.list, d->C, plane, tpp, one, one, 1, out);
      else
        hipLaunchKernelGGL((splat_fwd_quad_kernel<false, false>), dim3(d->B * d->nz * tpp), dim3(256), 0, st, x, (const float*)nullptr, ws.offsets,
                           ws.list, d->C, plane, tpp, one, one, 1, out);
      MF_LAUNCH_OK("splat_fwd");
      return MF_OK;
    }
  }
  hipLaunchKernelGGL((splat_fwd_kernel<S>), dim3(d->B * d->nz * tpp), dim3(256), 0, st, x, ws.offsets, ws.list, d->C, plane, tpp, out);
  MF_LAUNCH_OK("splat_fwd");
  return MF_OK;
}

template <typename S>
static int splat_bwd(const MfSplatDesc* d, const S* gout, const void* workspace, S* gx, hipStream_t st) {
  SplatWs ws;
  carve(d, const_cast<void*>(workspace), &ws);
  const int P = d->B * d->n_per_sample;
  const int plane = d->nx * d->ny;
  const int tpp = (plane + kTileVox - 1) / kTileVox;
  hipLaunchKernelGGL((splat_bwd_zero_kernel<S>), dim3((P + 255) / 256), dim3(256), 0, st, ws.keys, P, d->C, gx);
  if constexpr (sizeof(S) == 4) {
    if (quad_forward_ok(d, plane)) {      // sixteen lanes per point (splat_bwd_quad_kernel)
      hipLaunchKernelGGL(splat_bwd_quad_kernel, dim3(d->B * d->nz * tpp), dim3(256), 0, st, gout, ws.offsets, ws.list, ws.keys, d->C, plane, tpp, gx);
      MF_LAUNCH_OK("splat_bwd");
      return MF_OK;
    }
  }
  hipLaunchKernelGGL((splat_bwd_kernel<S>), dim3(d->B * d->nz * tpp), dim3(256), 0, st, gout, ws.offsets, ws.list, d->C, plane, tpp, gx);
  MF_LAUNCH_OK("splat_bwd");
  return MF_OK;
}

template <typename S>
static int lift_check(const MfSplatDesc* d) {
  MF_REQUIRE(d->lift_D > 0 && d->lift_hw > 0 && d->n_per_sample % (d->lift_D * d->lift_hw) == 0, MF_ERR_INVALID,
             "bev_lift_splat: n_per_sample must be cameras * lift_D * lift_hw");
  return MF_OK;
}

template <typename S>
static int lift_splat_fwd(const MfSplatDesc* d, const S* depth, const S* ctx, const void* workspace, S* out, hipStream_t st) {
  int rc = lift_check<S>(d);
  if (rc != MF_OK) return rc;
  SplatWs ws;
  carve(d, const_cast<void*>(workspace), &ws);
  const int plane = d->nx * d->ny;
  const int tpp = (plane + kTileVox - 1) / kTileVox;
  if constexpr (sizeof(S) == 4) {
    if (quad_forward_ok(d, plane)) {
      const FastDiv div_dhw = fast_div_make((unsigned)(d->lift_D * d->lift_hw)), div_hw = fast_div_make((unsigned)d->lift_hw);
      if (quad_pipe())
        hipLaunchKernelGGL((splat_fwd_quad_kernel<true, true>), dim3(d->B * d->nz * tpp), dim3(256), 0, st, ctx, depth, ws.offsets, ws.list, d->C, plane,
                           tpp, div_dhw, div_hw, d->lift_hw, out);
      else
        hipLaunchKernelGGL((splat_fwd_quad_kernel<true, false>), dim3(d->B * d->nz * tpp), dim3(256), 0, st, ctx, depth, ws.offsets, ws.list, d->C, plane,
                           tpp, div_dhw, div_hw, d->lift_hw, out);
      MF_LAUNCH_OK("lift_splat_fwd");
      return MF_OK;
    }
  }
  hipLaunchKernelGGL((lift_splat_fwd_kernel<S>), dim3(d->B * d->nz * tpp), dim3(256), 0, st, depth, ctx, ws.offsets, ws.list, d->C, plane, tpp,
                     d->lift_D * d->lift_hw, d->lift_hw, out);
  MF_LAUNCH_OK("lift_splat_fwd");
  return MF_OK;
}

template <typename S>
static int lift_splat_bwd(const MfSplatDesc* d, const S* depth, const S* ctx, const void* workspace, const S* gout, S* gT, S* g_depth,
                          S* g_ctx, hipStream_t st) {
  int rc = lift_check<S>(d);
  if (rc != MF_OK) return rc;
  SplatWs ws;
  carve(d, const_cast<void*>(workspace), &ws);
  const int plane = d->nx * d->ny;
  const int tpp = (plane + kTileVox - 1) / kTileVox;
  const int n_pix = d->B * (d->n_per_sample / d->lift_D);       // samples * cameras * pixels
  if constexpr (sizeof(S) == 4) {
    if (quad_forward_ok(d, plane)) {      // sixteen lanes per voxel row / per pixel
      hipLaunchKernelGGL(lift_splat_bwd_rows_quad_kernel, dim3(d->B * d->nz * tpp), dim3(256), 0, st, gout, ws.offsets, d->C, plane, tpp, gT);
      hipLaunchKernelGGL(lift_splat_bwd_gather_quad_kernel, dim3((n_pix + 15) / 16), dim3(256), 0, st, depth, ctx, ws.keys, gT, d->C, d->lift_D,
                         d->lift_hw, n_pix, g_depth, g_ctx);
      MF_LAUNCH_OK("lift_splat_bwd");
      return MF_OK;
    }
  }
  hipLaunchKernelGGL((lift_splat_bwd_rows_kernel<S>), dim3(d->B * d->nz * tpp), dim3(256), 0, st, gout, ws.offsets, d->C, plane, tpp, gT);
  hipLaunchKernelGGL((lift_splat_bwd_gather_kernel<S>), dim3((n_pix + 3) / 4), dim3(256), 0, st, depth, ctx, ws.keys, gT, d->C, d->lift_D,
                     d->lift_hw, n_pix, g_depth, g_ctx);
  MF_LAUNCH_OK("lift_splat_bwd");
  return MF_OK;
}

}  // namespace mf

#define MF_LIFT_ENTRIES(sfx, S)                                                                                                    \
  extern "C" int mf_bev_lift_splat_fwd_##sfx(const MfSplatDesc* d, const S* depth, const S* ctx, const void* ws, S* out, void* s) { \
    int rc = mf::check_desc(d);                                                                                                    \
    if (rc != MF_OK) return rc;                                                                                                    \
    MF_REQUIRE(depth && ctx && ws && out, MF_ERR_INVALID, "bev_lift_splat_fwd: null buffer");                                      \
    return mf::lift_splat_fwd<S>(d, depth, ctx, ws, out, (hipStream_t)s);                                                          \
  }                                                                                                                                \
  extern "C" int mf_bev_lift_splat_bwd_##sfx(const MfSplatDesc* d, const S* depth, const S* ctx, const void* ws, const S* gout,    \
                                             S* rows_scratch, S* g_depth, S* g_ctx, void* s) {                                     \
    int rc = mf::check_desc(d);                                                                                                    \
    if (rc != MF_OK) return rc;                                                                                                    \
    MF_REQUIRE(depth && ctx && ws && gout && rows_scratch && g_depth && g_ctx, MF_ERR_INVALID, "bev_lift_splat_bwd: null buffer"); \
    return mf::lift_splat_bwd<S>(d, depth, ctx, ws, gout, rows_scratch, g_depth, g_ctx, (hipStream_t)s);                           \
  }
MF_LIFT_ENTRIES(f32, float)
MF_LIFT_ENTRIES(f64, double)

extern "C" size_t mf_bev_splat_workspace_bytes(const MfSplatDesc* d) {
  if (mf::check_desc(d) != MF_OK) return 0;
  return mf::carve(d, nullptr, nullptr);
}
extern "C" int mf_bev_splat_prepare(const MfSplatDesc* d, const float* geom, void* ws, void* s) {
  int rc = mf::check_desc(d);
  if (rc != MF_OK) return rc;
  MF_REQUIRE(geom && ws, MF_ERR_INVALID, "bev_splat_prepare: null buffer");
  return mf::splat_prepare(d, geom, nullptr, nullptr, 0, ws, (hipStream_t)s);
}
extern "C" int mf_bev_splat_prepare_cameras(const MfSplatDesc* d, const float* frustum, int32_t pts_per_cam, const float* cams,
                                            void* ws, void* s) {
  int rc = mf::check_desc(d);
  if (rc != MF_OK) return rc;
  MF_REQUIRE(frustum && cams && ws, MF_ERR_INVALID, "bev_splat_prepare_cameras: null buffer");
  MF_REQUIRE(pts_per_cam > 0 && d->n_per_sample % pts_per_cam == 0, MF_ERR_INVALID,
             "bev_splat_prepare_cameras: n_per_sample must be cameras * pts_per_cam");
  return mf::splat_prepare(d, nullptr, frustum, cams, pts_per_cam, ws, (hipStream_t)s);
}
extern "C" int mf_bev_splat_prepare_rig(const MfSplatDesc* d, const float* frustum, int32_t pts_per_cam, const float* rots, const float* trans,
                                        const float* intrins, const float* post_rots, const float* post_trans, void* ws, void* s) {
  int rc = mf::check_desc(d);
  if (rc != MF_OK) return rc;
  MF_REQUIRE(frustum && rots && trans && intrins && post_rots && post_trans && ws, MF_ERR_INVALID, "bev_splat_prepare_rig: null buffer");
  MF_REQUIRE(pts_per_cam > 0 && d->n_per_sample % pts_per_cam == 0, MF_ERR_INVALID, "bev_splat_prepare_rig: n_per_sample must be cameras * pts_per_cam");
  const mf::RigPtrs rig{rots, trans, intrins, post_rots, post_trans};
  return mf::splat_prepare(d, nullptr, frustum, nullptr, pts_per_cam, ws, (hipStream_t)s, &rig);
}
#define MF_SPLAT_ENTRY(name, S, impl)                                                           \
  extern "C" int name(const MfSplatDesc* d, const S* in, const void* ws, S* out, void* s) {     \
    int rc = mf::check_desc(d);                                                                 \
    if (rc != MF_OK) return rc;                                                                 \
    MF_REQUIRE(in && ws && out, MF_ERR_INVALID, #name ": null buffer");                         \
    return mf::impl<S>(d, in, ws, out, (hipStream_t)s);                                         \
  }
MF_SPLAT_ENTRY(mf_bev_splat_fwd_f32, float, splat_fwd)
MF_SPLAT_ENTRY(mf_bev_splat_fwd_f64, double, splat_fwd)
MF_SPLAT_ENTRY(mf_bev_splat_bwd_f32, float, splat_bwd)
MF_SPLAT_ENTRY(mf_bev_splat_bwd_f64, double, splat_bwd)

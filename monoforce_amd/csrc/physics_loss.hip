// Fused physics loss: the time-discounted position MSE of
// /root/reference/monoforce/src/monoforce/losses.py:102-127 as one gather-reduce kernel and one scatter kernel, instead of
// ~25 small ATen kernels (gather, broadcasts, pow, mean, and an index_put backward that sorts).  The backward writes
// d loss / d Xs directly in the rollout's own (time-major) layout, so the rollout backward consumes it without a copy.
//   loss = mean_{b,j,c} ( (Xs[b, nearest[b,j], c] - Xgt[b,j,c]) * w[b,j] )^2,   w = 1 / (1 + gamma * gt_ts[b,j])
// mf_physics_loss_value_* finishes the mean inside the same launch (the block that takes the last ticket adds the per-block
// partial sums in index order), and mf_reduce_grad_copies_* sums the rollout backward's private gradient copies and clears them
// for the next step: a train step at the BASELINE shape is two ~0.2-0.4 ms kernels, and every ~4 us ATen launch between them
// costs ~10 us of dependent-launch latency (four of them went here, one more in train.py: 0.62 -> 0.59 ms per step).
#include "mf_common.h"

namespace mf {

__device__ __forceinline__ void atomic_add_s(float* p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add_s(double* p, double v) { unsafeAtomicAdd(p, v); }

// one thread per (rollout, ground-truth stamp); per-block partial sums in a fixed order => deterministic loss
template <typename S>
__global__ void __launch_bounds__(256) physics_loss_fwd_kernel(const S* __restrict__ Xs, long long sb, long long st,
                                                              const S* __restrict__ Xgt, const S* __restrict__ gt_ts,
                                                              const int* __restrict__ nearest, int B, int T2, S gamma,
                                                              S* __restrict__ partial) {
  __shared__ S wave_sum[4];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;        // = b * T2 + j
  S acc = (S)0;
  if (i < B * T2) {
    const int b = i / T2;
    const S w = (S)1 / ((S)1 + gamma * gt_ts[i]);
    const S* x = Xs + b * sb + (long long)nearest[i] * st;
    const S* g = Xgt + (size_t)i * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) { const S d = x[c] * w - g[c] * w; acc += d * d; }   // (pred*w - gt*w)^2, as the reference
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (wave_sum[0] + wave_sum[1]) + (wave_sum[2] + wave_sum[3]);
}

// the same, and the block that finishes last turns the partial sums into the mean (fixed order: deterministic); `ticket` is a
// zero-initialised counter the last block resets, so the buffer is reusable launch after launch on one stream
template <typename S>
__global__ void __launch_bounds__(256) physics_loss_value_kernel(const S* __restrict__ Xs, long long sb, long long st,
                                                                const S* __restrict__ Xgt, const S* __restrict__ gt_ts,
                                                                const int* __restrict__ nearest, int B, int T2, S gamma,
                                                                S* __restrict__ partial, unsigned* __restrict__ ticket, S inv_count,
                                                                S* __restrict__ loss, S* __restrict__ zero_fill, long long zero_count) {
  __shared__ S wave_sum[4];
  __shared__ bool last;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;        // = b * T2 + j
  // the buffer the backward will scatter d loss / d Xs into is cleared here (one launch fewer in front of the rollout backward)
  {   // (16-byte stores: the scalar form filled config 3's 98 MB at 16 384 rollouts at 0.6 TB/s -- 0.16 ms in front of a 1 ms backward)
    const long long head = min(zero_count, (long long)(((16u - (unsigned)((uintptr_t)zero_fill & 15u)) & 15u) / sizeof(S)));
    const long long n16 = (zero_count - head) * (long long)sizeof(S) / 16, tail0 = head + n16 * (16 / (long long)sizeof(S));
    int4* z16 = reinterpret_cast<int4*>(zero_fill + head);
    for (long long k = i; k < n16; k += (long long)gridDim.x * blockDim.x) z16[k] = make_int4(0, 0, 0, 0);
    if (i < head) zero_fill[i] = (S)0;
    if (tail0 + i < zero_count) zero_fill[tail0 + i] = (S)0;
  }
  S acc = (S)0;
  if (i < B * T2) {
    // (round 6: neighbouring threads = neighbouring ROLLOUTS of one stamp, so that a wave reads one stretch of an Xs row, measured SLOWER --
    //  0.085 -> 0.136 ms at 16 384 rollouts x 50 stamps: the ground truth, its stamps and the index table are [B][T2] and became strided)
    const int k = i, b = i / T2;
    const S w = (S)1 / ((S)1 + gamma * gt_ts[k]);
    const S* x = Xs + b * sb + (long long)nearest[k] * st;
    const S* g = Xgt + (size_t)k * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) { const S d = x[c] * w - g[c] * w; acc += d * d; }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = (wave_sum[0] + wave_sum[1]) + (wave_sum[2] + wave_sum[3]);
    __threadfence();
    last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  // 256 threads stride over the partial sums in index order, then the same butterfly
  S tot = (S)0;
  for (unsigned k = threadIdx.x; k < gridDim.x; k += 256) tot += __builtin_nontemporal_load(partial + k);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d, 64);
  if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = tot;
  __syncthreads();
  if (threadIdx.x == 0) {
    loss[0] = ((wave_sum[0] + wave_sum[1]) + (wave_sum[2] + wave_sum[3])) * inv_count;
    *ticket = 0u;
  }
}

// out[m][i] = sum_c pool[m][c][i];  pool <- 0   (the rollout backward scatters into `copies` private copies of each map)
template <typename S>
__global__ void __launch_bounds__(256) reduce_grad_copies_kernel(S* __restrict__ pool, int copies, long long n, S* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  S* p = pool + (long long)blockIdx.y * copies * n + i;
  S acc = (S)0;
  for (int c = 0; c < copies; ++c) { acc += p[(long long)c * n]; p[(long long)c * n] = (S)0; }
  out[(long long)blockIdx.y * n + i] = acc;
}

template <typename S>
__global__ void __launch_bounds__(256) physics_loss_bwd_kernel(const S* __restrict__ Xs, long long sb, long long st,
                                                              const S* __restrict__ Xgt, const S* __restrict__ gt_ts,
                                                              const int* __restrict__ nearest, int B, int T2, S gamma,
                                                              const S* __restrict__ gloss, S inv_count, S* __restrict__ gXs) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * T2) return;
  // time-major rows (sb < st): neighbouring threads take neighbouring ROLLOUTS of one stamp -- their Xs / gXs rows are neighbours in
  // memory when the rollouts share their stamps (the common case); batch-major rows: neighbouring stamps of one rollout
  const bool jm = sb < st;
  const int b = jm ? t % B : t / T2, j = jm ? t / B : t % T2;
  const int i = b * T2 + j;
  const S scale = (S)2 * gloss[0] * inv_count;
  const S w = (S)1 / ((S)1 + gamma * gt_ts[i]);
  const int nr = nearest[i];
  const long long o = b * sb + (long long)nr * st;
  const S* g = Xgt + (size_t)i * 3;
  // stamps of a rollout may share a step (then the contributions add up: atomics); a step this stamp has to itself is stored
  bool shared = false;
  for (int k = 0; k < T2; ++k) shared |= (k != j) & (nearest[b * T2 + k] == nr);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const S v = scale * w * (Xs[o + c] * w - g[c] * w);
    if (shared) atomic_add_s(gXs + o + c, v); else gXs[o + c] = v;
  }
}

template <typename S>
static int loss_fwd(const MfLossDesc* d, const S* Xs, const S* Xgt, const S* gt_ts, const int* nearest, S* partial, hipStream_t st) {
  MF_REQUIRE(d && Xs && Xgt && gt_ts && nearest && partial, MF_ERR_INVALID, "physics_loss_fwd: null argument");
  MF_REQUIRE(d->B > 0 && d->T1 > 0 && d->T2 > 0, MF_ERR_INVALID, "physics_loss_fwd: B, T1, T2 must be positive");
  hipLaunchKernelGGL((physics_loss_fwd_kernel<S>), dim3((d->B * d->T2 + 255) / 256), dim3(256), 0, st, Xs, (long long)d->x_stride_b,
                     (long long)d->x_stride_t, Xgt, gt_ts, nearest, d->B, d->T2, (S)d->gamma, partial);
  hipError_t e = hipGetLastError();
  MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("physics_loss_fwd launch: ") + hipGetErrorString(e));
  return MF_OK;
}

template <typename S>
static int loss_value(const MfLossDesc* d, const S* Xs, const S* Xgt, const S* gt_ts, const int* nearest, S* partial, unsigned* ticket,
                      S* loss, S* zero_fill, long long zero_count, hipStream_t st) {
  MF_REQUIRE(d && Xs && Xgt && gt_ts && nearest && partial && ticket && loss, MF_ERR_INVALID, "physics_loss_value: null argument");
  MF_REQUIRE(d->B > 0 && d->T1 > 0 && d->T2 > 0, MF_ERR_INVALID, "physics_loss_value: B, T1, T2 must be positive");
  MF_REQUIRE(zero_count >= 0 && (zero_fill || zero_count == 0), MF_ERR_INVALID, "physics_loss_value: zero_count without zero_fill");
  const double count = (double)d->B * d->T2 * 3;
  hipLaunchKernelGGL((physics_loss_value_kernel<S>), dim3((d->B * d->T2 + 255) / 256), dim3(256), 0, st, Xs, (long long)d->x_stride_b,
                     (long long)d->x_stride_t, Xgt, gt_ts, nearest, d->B, d->T2, (S)d->gamma, partial, ticket, (S)(1.0 / count), loss, zero_fill, zero_count);
  hipError_t e = hipGetLastError();
  MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("physics_loss_value launch: ") + hipGetErrorString(e));
  return MF_OK;
}

template <typename S>
static int reduce_copies(S* pool, int n_maps, int copies, long long n, S* out, hipStream_t st) {
  MF_REQUIRE(pool && out, MF_ERR_INVALID, "reduce_grad_copies: null argument");
  MF_REQUIRE(n_maps > 0 && n_maps < 65536 && copies > 0 && n > 0, MF_ERR_INVALID, "reduce_grad_copies: n_maps, copies, n must be positive");
  hipLaunchKernelGGL((reduce_grad_copies_kernel<S>), dim3((unsigned)((n + 255) / 256), (unsigned)n_maps), dim3(256), 0, st, pool, copies, n, out);
  hipError_t e = hipGetLastError();
  MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("reduce_grad_copies launch: ") + hipGetErrorString(e));
  return MF_OK;
}

template <typename S>
static int loss_bwd(const MfLossDesc* d, const S* Xs, const S* Xgt, const S* gt_ts, const int* nearest, const S* gloss, S* gXs,
                    hipStream_t st) {
  MF_REQUIRE(d && Xs && Xgt && gt_ts && nearest && gloss && gXs, MF_ERR_INVALID, "physics_loss_bwd: null argument");
  MF_REQUIRE(d->B > 0 && d->T1 > 0 && d->T2 > 0, MF_ERR_INVALID, "physics_loss_bwd: B, T1, T2 must be positive");
  const double count = (double)d->B * d->T2 * 3;
  hipLaunchKernelGGL((physics_loss_bwd_kernel<S>), dim3((d->B * d->T2 + 255) / 256), dim3(256), 0, st, Xs, (long long)d->x_stride_b,
                     (long long)d->x_stride_t, Xgt, gt_ts, nearest, d->B, d->T2, (S)d->gamma, gloss, (S)(1.0 / count), gXs);
  hipError_t e = hipGetLastError();
  MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("physics_loss_bwd launch: ") + hipGetErrorString(e));
  return MF_OK;
}

}  // namespace mf

namespace mf {
// nearest[b][j] = argmin_t |pred_ts[b][t] - gt_ts[b][j]| (losses.py:116: `torch.argmin(torch.abs(pred_ts.unsqueeze(1) - gt_ts.unsqueeze(2)), dim=2)`),
// the FIRST minimum like torch.argmin; NaN differences are skipped unless every one is NaN (then 0).  One thread per (rollout, stamp)
// scanning its rollout's T1 predicted stamps (a 2 KB row the 50 threads of a rollout share in L1): the reference's form materialises
// two [B,T2,T1] tensors -- 25.6 M elements at the BASELINE shape, 76 us of three ATen launches -- for 51 200 indices.
template <typename S>
__global__ void __launch_bounds__(256) nearest_steps_kernel(const S* __restrict__ pred_ts, long long pred_sb, const S* __restrict__ gt_ts, long long gt_sb,
                                                           int B, int T1, int T2, int* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;        // = b * T2 + j
  if (i >= B * T2) return;
  const int b = i / T2, j = i - b * T2;
  const S g = gt_ts[b * gt_sb + j];
  const S* p = pred_ts + b * pred_sb;
  S best = (S)INFINITY;
  int arg = 0;
  bool any = false;
  int t = 0;
  for (; t + 8 <= T1; t += 8) {      // eight stamps per trip: their loads are issued together (the chain of compares is what is serial)
    S v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = p[t + k];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const S d = fabs(v[k] - g);
      if (d < best || (!any && d == d)) { best = d; arg = t + k; any = true; }      // strict <: the first minimum
    }
  }
  for (; t < T1; ++t) {
    const S d = fabs(p[t] - g);
    if (d < best || (!any && d == d)) { best = d; arg = t; any = true; }
  }
  out[i] = arg;
}
template <typename S>
static int nearest_steps(int B, int T1, int T2, const S* pred_ts, long long pred_sb, const S* gt_ts, long long gt_sb, int32_t* out, hipStream_t st) {
  MF_REQUIRE(B > 0 && T1 > 0 && T2 > 0 && pred_ts && gt_ts && out && pred_sb >= 0 && gt_sb >= 0, MF_ERR_INVALID, "nearest_steps: bad argument");
  const long long n = (long long)B * T2;
  MF_REQUIRE(n < (1ll << 31), MF_ERR_UNSUPPORTED, "nearest_steps: B * T2 must be below 2^31");
  hipLaunchKernelGGL((nearest_steps_kernel<S>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pred_ts, pred_sb, gt_ts, gt_sb, B, T1, T2, out);
  hipError_t e = hipGetLastError();
  MF_REQUIRE(e == hipSuccess, MF_ERR_LAUNCH, std::string("nearest_steps launch: ") + hipGetErrorString(e));
  return MF_OK;
}
}  // namespace mf
extern "C" int mf_nearest_steps_f32(int32_t B, int32_t T1, int32_t T2, const float* pred_ts, long long pred_stride_b, const float* gt_ts, long long gt_stride_b,
                                    int32_t* nearest, void* s) { return mf::nearest_steps<float>(B, T1, T2, pred_ts, pred_stride_b, gt_ts, gt_stride_b, nearest, (hipStream_t)s); }
extern "C" int mf_nearest_steps_f64(int32_t B, int32_t T1, int32_t T2, const double* pred_ts, long long pred_stride_b, const double* gt_ts, long long gt_stride_b,
                                    int32_t* nearest, void* s) { return mf::nearest_steps<double>(B, T1, T2, pred_ts, pred_stride_b, gt_ts, gt_stride_b, nearest, (hipStream_t)s); }
extern "C" int mf_physics_loss_fwd_f32(const MfLossDesc* d, const float* Xs, const float* Xgt, const float* gt_ts, const int32_t* nearest,
                                       float* partial, void* s) { return mf::loss_fwd<float>(d, Xs, Xgt, gt_ts, nearest, partial, (hipStream_t)s); }
extern "C" int mf_physics_loss_fwd_f64(const MfLossDesc* d, const double* Xs, const double* Xgt, const double* gt_ts, const int32_t* nearest,
                                       double* partial, void* s) { return mf::loss_fwd<double>(d, Xs, Xgt, gt_ts, nearest, partial, (hipStream_t)s); }
extern "C" int mf_physics_loss_bwd_f32(const MfLossDesc* d, const float* Xs, const float* Xgt, const float* gt_ts, const int32_t* nearest,
                                       const float* gloss, float* gXs, void* s) { return mf::loss_bwd<float>(d, Xs, Xgt, gt_ts, nearest, gloss, gXs, (hipStream_t)s); }
extern "C" int mf_physics_loss_bwd_f64(const MfLossDesc* d, const double* Xs, const double* Xgt, const double* gt_ts, const int32_t* nearest,
                                       const double* gloss, double* gXs, void* s) { return mf::loss_bwd<double>(d, Xs, Xgt, gt_ts, nearest, gloss, gXs, (hipStream_t)s); }
extern "C" int mf_physics_loss_value_f32(const MfLossDesc* d, const float* Xs, const float* Xgt, const float* gt_ts, const int32_t* nearest,
                                         float* partial, uint32_t* ticket, float* loss, float* zero_fill, long long zero_count, void* s) { return mf::loss_value<float>(d, Xs, Xgt, gt_ts, nearest, partial, ticket, loss, zero_fill, zero_count, (hipStream_t)s); }
extern "C" int mf_physics_loss_value_f64(const MfLossDesc* d, const double* Xs, const double* Xgt, const double* gt_ts, const int32_t* nearest,
                                         double* partial, uint32_t* ticket, double* loss, double* zero_fill, long long zero_count, void* s) { return mf::loss_value<double>(d, Xs, Xgt, gt_ts, nearest, partial, ticket, loss, zero_fill, zero_count, (hipStream_t)s); }
extern "C" int mf_reduce_grad_copies_f32(float* pool, int n_maps, int copies, long long n, float* out, void* s) { return mf::reduce_copies<float>(pool, n_maps, copies, n, out, (hipStream_t)s); }
extern "C" int mf_reduce_grad_copies_f64(double* pool, int n_maps, int copies, long long n, double* out, void* s) { return mf::reduce_copies<double>(pool, n_maps, copies, n, out, (hipStream_t)s); }

// Store-pattern microbenchmark for the rollout forward: B rollouts x T steps, G = 4 lanes per rollout, 16 rollouts per
// wave, one workgroup = one wave.  Each pattern writes the same ~180 B per rollout-step, laid out differently.
//   hipcc --offload-arch=gfx950 -O3 store_patterns.hip -o store_patterns && ./store_patterns [B] [T]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Bufs { float *xraw, *xs, *xd, *om, *rs, *fs, *ff, *packed; int B, T; };

// P0: what the kernel does today: 4 vec3 arrays + R (9) redundantly from every lane of the group, forces per lane
__global__ void p_current(Bufs a) {
  int tid = blockIdx.x * 64 + threadIdx.x, b = tid >> 2, gl = tid & 3;
  float v = (float)tid;
  size_t r = b;
  for (int n = 0; n < a.T; ++n, r += a.B) {
    float* p;
    p = a.xraw + r * 3; p[0] = v; p[1] = v; p[2] = v;
    p = a.xs + r * 3; p[0] = v; p[1] = v; p[2] = v;
    p = a.xd + r * 3; p[0] = v; p[1] = v; p[2] = v;
    p = a.om + r * 3; p[0] = v; p[1] = v; p[2] = v;
    p = a.rs + r * 9;
#pragma unroll
    for (int c = 0; c < 9; ++c) p[c] = v;
    p = a.fs + (r * 4 + gl) * 3; p[0] = v; p[1] = v; p[2] = v;
    p = a.ff + (r * 4 + gl) * 3; p[0] = v; p[1] = v; p[2] = v;
    v += 1.0f;
  }
}
// P1: one packed row of 48 floats per rollout-step (time-major [T][B][48]); lane gl stores floats [12 gl, 12 gl + 12)
__global__ void p_packed_lane(Bufs a) {
  int tid = blockIdx.x * 64 + threadIdx.x, b = tid >> 2, gl = tid & 3;
  float v = (float)tid;
  float4* p = reinterpret_cast<float4*>(a.packed + (size_t)b * 48 + gl * 12);
  const size_t adv = (size_t)a.B * 12;   // in float4
  for (int n = 0; n < a.T; ++n, p += adv) {
    p[0] = make_float4(v, v, v, v); p[1] = make_float4(v, v, v, v); p[2] = make_float4(v, v, v, v);
    v += 1.0f;
  }
}
// P2: same packed rows, fully coalesced: the wave's 16 rows = 3072 B written as 3 float4 stores at lane * 16 + k * 1024
__global__ void p_packed_coalesced(Bufs a) {
  int tid = blockIdx.x * 64 + threadIdx.x;
  float v = (float)tid;
  float4* p = reinterpret_cast<float4*>(a.packed + (size_t)blockIdx.x * 16 * 48) + threadIdx.x;
  const size_t adv = (size_t)a.B * 12;
  for (int n = 0; n < a.T; ++n, p += adv) {
    p[0] = make_float4(v, v, v, v); p[64] = make_float4(v, v, v, v); p[128] = make_float4(v, v, v, v);
    v += 1.0f;
  }
}
// P3: rollout-major packed [B][T][48]: each group streams its own contiguous 192 B rows
__global__ void p_rollout_major(Bufs a) {
  int tid = blockIdx.x * 64 + threadIdx.x, b = tid >> 2, gl = tid & 3;
  float v = (float)tid;
  float4* p = reinterpret_cast<float4*>(a.packed + (size_t)b * a.T * 48 + gl * 12);
  for (int n = 0; n < a.T; ++n, p += 12) {
    p[0] = make_float4(v, v, v, v); p[1] = make_float4(v, v, v, v); p[2] = make_float4(v, v, v, v);
    v += 1.0f;
  }
}
// P4: time-tiled: [T/8][B][8][48]... each rollout writes 8 consecutive steps (1536 B) contiguously
__global__ void p_time_tiled(Bufs a) {
  int tid = blockIdx.x * 64 + threadIdx.x, b = tid >> 2, gl = tid & 3;
  float v = (float)tid;
  for (int n = 0; n < a.T; ++n) {
    float4* p = reinterpret_cast<float4*>(a.packed + (((size_t)(n >> 3) * a.B + b) * 8 + (n & 7)) * 48 + gl * 12);
    p[0] = make_float4(v, v, v, v); p[1] = make_float4(v, v, v, v); p[2] = make_float4(v, v, v, v);
    v += 1.0f;
  }
}
// P5: current arrays but the 4 vec3 + R split over the lanes of the group (2 dwordx3 per lane)
__global__ void p_split(Bufs a) {
  int tid = blockIdx.x * 64 + threadIdx.x, b = tid >> 2, gl = tid & 3;
  float v = (float)tid;
  float* p3 = (gl == 0 ? a.xraw : gl == 1 ? a.xs : gl == 2 ? a.xd : a.om) + (size_t)b * 3;
  float* pr = a.rs + (size_t)b * 9 + 3 * (gl < 2 ? gl : 2);
  float* pf = a.fs + ((size_t)b * 4 + gl) * 3;
  float* pg = a.ff + ((size_t)b * 4 + gl) * 3;
  const size_t B = a.B;
  for (int n = 0; n < a.T; ++n) {
    p3[0] = v; p3[1] = v; p3[2] = v; pr[0] = v; pr[1] = v; pr[2] = v;
    pf[0] = v; pf[1] = v; pf[2] = v; pg[0] = v; pg[1] = v; pg[2] = v;
    p3 += B * 3; pr += B * 9; pf += B * 12; pg += B * 12;
    v += 1.0f;
  }
}
// P6: two packed arrays: state rows of 24 floats ([T][B][24], lane stores 6) and force rows of 24 ([T][B][4][6])
__global__ void p_two_rows(Bufs a) {
  int tid = blockIdx.x * 64 + threadIdx.x, b = tid >> 2, gl = tid & 3;
  float v = (float)tid;
  float* ps = a.packed + (size_t)b * 24 + gl * 6;
  float* pf = a.packed + (size_t)a.B * a.T * 24 + (size_t)b * 24 + gl * 6;
  const size_t adv = (size_t)a.B * 24;
  for (int n = 0; n < a.T; ++n, ps += adv, pf += adv) {
    *reinterpret_cast<float4*>(ps) = make_float4(v, v, v, v); *reinterpret_cast<float2*>(ps + 4) = make_float2(v, v);
    *reinterpret_cast<float4*>(pf) = make_float4(v, v, v, v); *reinterpret_cast<float2*>(pf + 4) = make_float2(v, v);
    v += 1.0f;
  }
}

// P7: as P6 but each lane writes its 24 B as two dwordx3 (what runtime-strided pointers give)
__global__ void p_two_rows_x3(Bufs a) {
  int tid = blockIdx.x * 64 + threadIdx.x, b = tid >> 2, gl = tid & 3;
  float v = (float)tid;
  float* ps = a.packed + (size_t)b * 24 + gl * 6;
  float* pf = a.packed + (size_t)a.B * a.T * 24 + (size_t)b * 24 + gl * 6;
  float* ps2 = ps + 3; float* pf2 = pf + 3;
  const size_t adv = (size_t)a.B * 24;
  for (int n = 0; n < a.T; ++n, ps += adv, pf += adv, ps2 += adv, pf2 += adv) {
    ps[0] = v; ps[1] = v; ps[2] = v;
    __builtin_amdgcn_sched_barrier(0);
    ps2[0] = v; ps2[1] = v; ps2[2] = v;
    __builtin_amdgcn_sched_barrier(0);
    pf[0] = v; pf[1] = v; pf[2] = v;
    __builtin_amdgcn_sched_barrier(0);
    pf2[0] = v; pf2[1] = v; pf2[2] = v;
    v += 1.0f;
  }
}
// P9: states only, today's arrays (what the training step writes)   P10: states only, packed 96 B rows
__global__ void p_states_current(Bufs a) {
  int tid = blockIdx.x * 64 + threadIdx.x, b = tid >> 2;
  float v = (float)tid;
  size_t r = b;
  for (int n = 0; n < a.T; ++n, r += a.B) {
    float* p;
    p = a.xraw + r * 3; p[0] = v; p[1] = v; p[2] = v;
    p = a.xs + r * 3; p[0] = v; p[1] = v; p[2] = v;
    p = a.xd + r * 3; p[0] = v; p[1] = v; p[2] = v;
    p = a.om + r * 3; p[0] = v; p[1] = v; p[2] = v;
    p = a.rs + r * 9;
#pragma unroll
    for (int c = 0; c < 9; ++c) p[c] = v;
    v += 1.0f;
  }
}
__global__ void p_states_packed(Bufs a) {
  int tid = blockIdx.x * 64 + threadIdx.x, b = tid >> 2, gl = tid & 3;
  float v = (float)tid;
  float* ps = a.packed + (size_t)b * 24 + gl * 6;
  const size_t adv = (size_t)a.B * 24;
  for (int n = 0; n < a.T; ++n, ps += adv) {
    *reinterpret_cast<float4*>(ps) = make_float4(v, v, v, v); *reinterpret_cast<float2*>(ps + 4) = make_float2(v, v);
    v += 1.0f;
  }
}

int main(int argc, char** argv) {
  int B = argc > 1 ? atoi(argv[1]) : 65536, T = argc > 2 ? atoi(argv[2]) : 500;
  Bufs a; a.B = B; a.T = T;
  size_t rows = (size_t)B * T;
  CK(hipMalloc(&a.xraw, rows * 12)); CK(hipMalloc(&a.xs, rows * 12)); CK(hipMalloc(&a.xd, rows * 12)); CK(hipMalloc(&a.om, rows * 12));
  CK(hipMalloc(&a.rs, rows * 36)); CK(hipMalloc(&a.fs, rows * 48)); CK(hipMalloc(&a.ff, rows * 48)); CK(hipMalloc(&a.packed, (size_t)B * ((T + 7) / 8 * 8) * 192));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = B * 4 / 64;
  struct { const char* name; void (*k)(Bufs); double bytes; } ks[] = {
    {"current (7 state stores x4 redundant + 2 force)", p_current, 180.0}, {"split over lanes (2 + 2 dwordx3)", p_split, 180.0},
    {"packed 192 B rows, lane-contiguous 48 B", p_packed_lane, 192.0}, {"packed rows, wave-coalesced 1 KiB stores", p_packed_coalesced, 192.0},
    {"rollout-major packed", p_rollout_major, 192.0}, {"time-tiled x8 packed", p_time_tiled, 192.0}, {"two packed arrays of 96 B rows", p_two_rows, 192.0}, {"two packed arrays, 2 x dwordx3 per lane", p_two_rows_x3, 192.0},
    {"states only, current arrays", p_states_current, 84.0}, {"states only, packed 96 B rows", p_states_packed, 96.0}};
  for (int rep = 0; rep < 3; ++rep)
  for (auto& k : ks) {
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k.k, dim3(grid), dim3(64), 0, 0, a);
    CK(hipEventRecord(e0));
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(k.k, dim3(grid), dim3(64), 0, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("B=%d T=%d %-52s %8.3f ms  %6.2f TB/s\n", B, T, k.name, ms, k.bytes * rows / ms / 1e9);
  }
  return 0;
}

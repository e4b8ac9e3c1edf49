// Shared host/device helpers for libhumor_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/humor_amd.h"

// occupancy hint for a kernel (register budget = 512 / waves); the host SIMT emulator of the test tier has no such notion
#ifdef HA_SIMT_EMU
#define HA_WAVES_PER_EU(lo, hi)
#else
#define HA_WAVES_PER_EU(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#endif

// scheduling fence: no instruction is moved across it by the compiler's scheduler (software pipelining by hand)
#ifdef HA_SIMT_EMU
#define HA_SCHED_FENCE()
#else
#define HA_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

namespace ha {

void set_error(const char* fmt, ...);

// LDS written by some lanes of a wavefront is read by other lanes of the SAME wavefront: no s_barrier needed, only the ordering
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// returnless hardware fp32 add at the L2 / memory side (global_atomic_add_f32; no compare-and-swap loop)
__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
#ifdef HA_SIMT_EMU
  atomicAdd(p, v);
#else
  unsafeAtomicAdd(p, v);
#endif
}

// Zero-fill as a KERNEL (4-byte words).  Not hipMemsetAsync: inside a captured hipGraph a memset node followed by a kernel node was
// observed (ROCm 7.0 / gfx950, round 4) to let the kernel start on the previous replay's contents -- the persistent roll-out kernel then
// found its team counters already full (error word 0x100) -- while kernel -> kernel edges are ordered exactly as in a stream.
static __global__ void zero_words_kernel(unsigned* __restrict__ p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}
// Every kernel launch of the library goes through HA_LAUNCH: with ha_tune_set("cu_poison") (test tier, debug.hip) a kernel that fills the LDS and
// the vector registers of every CU with a bit pattern runs in front of it -- a kernel that reads state it did not write then fails on every box.
extern unsigned g_cu_poison;
int cu_poison_launch(hipStream_t st);
#define HA_LAUNCH(kernel, grid, block, lds, st, ...)                          \
  do {                                                                        \
    if (ha::g_cu_poison) (void)ha::cu_poison_launch(st);                      \
    hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);            \
  } while (0)

// The kernel's dynamic LDS.  gfx950: the usual extern array.  Host emulator (CPU test tier): the block's own buffer -- the persistent roll-out
// kernels run with the 32 blocks of a team resident at the same time there (tests/simt_emu: resident teams), one LDS each.
#ifdef HA_SIMT_EMU
#define HA_DYN_LDS(name) float* const name = simt_emu::block_lds()
#else
#define HA_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) float name[]
#endif

// two values made opaque to the optimiser (it must not hoist what depends on them out of a loop); the constraint letter differs on the host
#ifdef HA_SIMT_EMU
#define HA_OPAQUE2(a, b) asm volatile("" : "+r"(a), "+r"(b))
#else
#define HA_OPAQUE2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#endif

static inline void zero_async(void* p, size_t bytes, hipStream_t st) {
  const size_t n = bytes / 4;       // (every caller clears whole floats)
  if (n == 0) return;
  HA_LAUNCH(zero_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, reinterpret_cast<unsigned*>(p), n);
}

#define HA_CHECK_HIP(expr)                                                                        \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) {                                                                       \
      ha::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));         \
      return HA_ERR_HIP;                                                                          \
    }                                                                                             \
  } while (0)

#define HA_REQUIRE(cond, ...)                                                                     \
  do {                                                                                            \
    if (!(cond)) {                                                                                \
      ha::set_error(__VA_ARGS__);                                                                 \
      return HA_ERR_INVALID_ARG;                                                                  \
    }                                                                                             \
  } while (0)

#define HA_LAUNCH_CHECK()                                                                         \
  do {                                                                                            \
    hipError_t _e = hipGetLastError();                                                            \
    if (_e != hipSuccess) {                                                                       \
      ha::set_error("%s:%d: kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(_e));     \
      return HA_ERR_HIP;                                                                          \
    }                                                                                             \
  } while (0)

// RAII device switch: every entry point runs on the handle's device and restores the caller's.
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
    if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
  }
  ~DeviceGuard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------------
// small fixed-size rotation math, row-major 3x3 in float[9]
// ---------------------------------------------------------------------------------------------
#define HA_HD __host__ __device__ __forceinline__

// R = I + sin(t) K + (1-cos t) K^2, t = ||r + 1e-8||, K = skew(r / t)   (transforms.py:139-170)
HA_HD void rodrigues(const float r[3], float R[9]) {
  const float ux = r[0] + 1e-8f, uy = r[1] + 1e-8f, uz = r[2] + 1e-8f;
  const float t = sqrtf(ux * ux + uy * uy + uz * uz);
  const float nx = r[0] / t, ny = r[1] / t, nz = r[2] / t;
  const float s = sinf(t), c1 = 1.0f - cosf(t);
  // K = [[0,-nz,ny],[nz,0,-nx],[-ny,nx,0]],  K^2 = n n^T - |n|^2 I
  const float nn = nx * nx + ny * ny + nz * nz;
  R[0] = 1.0f + c1 * (nx * nx - nn);
  R[1] = -s * nz + c1 * (nx * ny);
  R[2] = s * ny + c1 * (nx * nz);
  R[3] = s * nz + c1 * (nx * ny);
  R[4] = 1.0f + c1 * (ny * ny - nn);
  R[5] = -s * nx + c1 * (ny * nz);
  R[6] = -s * ny + c1 * (nx * nz);
  R[7] = s * nx + c1 * (ny * nz);
  R[8] = 1.0f + c1 * (nz * nz - nn);
}

// gradient of rodrigues(): gr = dL/dr given gR = dL/dR
HA_HD void rodrigues_bwd(const float r[3], const float gR[9], float gr[3]) {
  const float ux = r[0] + 1e-8f, uy = r[1] + 1e-8f, uz = r[2] + 1e-8f;
  const float t = sqrtf(ux * ux + uy * uy + uz * uz);
  const float it = 1.0f / t;
  const float nx = r[0] * it, ny = r[1] * it, nz = r[2] * it;
  const float s = sinf(t), c = cosf(t), c1 = 1.0f - c;
  const float nn = nx * nx + ny * ny + nz * nz;
  // <gR, K> and <gR, K^2>
  const float gK_dot = -nz * gR[1] + ny * gR[2] + nz * gR[3] - nx * gR[5] - ny * gR[6] + nx * gR[7];
  const float tr = gR[0] + gR[4] + gR[8];
  const float nGn = nx * (gR[0] * nx + gR[1] * ny + gR[2] * nz) + ny * (gR[3] * nx + gR[4] * ny + gR[5] * nz) +
                    nz * (gR[6] * nx + gR[7] * ny + gR[8] * nz);
  const float gK2_dot = nGn - nn * tr;
  const float gt = c * gK_dot + s * gK2_dot;   // d/dt [ s K + (1-c) K^2 ]
  // gradient w.r.t. n: from s*K (skew part of gR) and (1-c)*(n n^T - nn I)
  // d<gR, n n^T>/dn = (gR + gR^T) n ;  d<gR, nn I>/dn = 2 tr n
  const float sx = (gR[0] + gR[0]) * nx + (gR[1] + gR[3]) * ny + (gR[2] + gR[6]) * nz;
  const float sy = (gR[3] + gR[1]) * nx + (gR[4] + gR[4]) * ny + (gR[5] + gR[7]) * nz;
  const float sz = (gR[6] + gR[2]) * nx + (gR[7] + gR[5]) * ny + (gR[8] + gR[8]) * nz;
  const float gnx = s * (gR[7] - gR[5]) + c1 * (sx - 2.0f * tr * nx);
  const float gny = s * (gR[2] - gR[6]) + c1 * (sy - 2.0f * tr * ny);
  const float gnz = s * (gR[3] - gR[1]) + c1 * (sz - 2.0f * tr * nz);
  // n = r / t, t = ||u||, u = r + eps: dn_i/dr_j = delta_ij / t - r_i u_j / t^3 ; dt/dr_j = u_j / t
  const float gn_r = gnx * r[0] + gny * r[1] + gnz * r[2];
  const float k = (gt - gn_r * it * it) * it;
  gr[0] = gnx * it + k * ux;
  gr[1] = gny * it + k * uy;
  gr[2] = gnz * it + k * uz;
}

// C = A * B (3x3)
HA_HD void mat3_mul(const float A[9], const float B[9], float C[9]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
// C = A^T * B
HA_HD void mat3_tmul(const float A[9], const float B[9], float C[9]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
// C = A * B^T
HA_HD void mat3_mult(const float A[9], const float B[9], float C[9]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      C[i * 3 + j] = A[i * 3] * B[j * 3] + A[i * 3 + 1] * B[j * 3 + 1] + A[i * 3 + 2] * B[j * 3 + 2];
}
HA_HD void mat3_vec(const float A[9], const float v[3], float o[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
}
HA_HD void mat3_tvec(const float A[9], const float v[3], float o[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = A[i] * v[0] + A[3 + i] * v[1] + A[6 + i] * v[2];
}

}  // namespace ha

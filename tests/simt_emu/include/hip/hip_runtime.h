// TEST INFRASTRUCTURE ONLY -- a host-side SIMT emulator that shadows <hip/hip_runtime.h>.
//
// tests/simt_emu/build.py compiles the *unmodified* kernel sources of humor_amd/csrc with the host compiler and
// this directory first on the include path, producing tests/simt_emu/_emu/libhumor_amd_emu.so.  It lets the
// CPU-only test tier (`pytest -m "not gpu"`) execute the kernels' control flow, indexing, LDS hand-offs,
// wave shuffles and MFMA fragment maps against the oracle before any GPU time is spent.  It is never built by
// __graft_entry__.build() as part of the product, never loaded by the humor_amd package, and is not a fallback:
// humor_amd raises if libhumor_amd.so (the gfx950 build) is missing.
//
// Model: one OS thread per work-item of a block (blocks run one after another), std::barrier for
// __syncthreads, a per-wave exchange buffer + barrier for cross-lane ops.  Wave size 64.  Every launch starts on
// NaN-filled LDS.  Resident teams (simt_emu::g_resident_blocks > 0): ALL blocks of the launch run at the same time,
// each with its own context and LDS -- the persistent roll-out kernels hand activations from block to block.
#pragma once
#define HA_SIMT_EMU 1
#include <atomic>
#include <barrier>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__
#define __shared__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef int hipError_t;
typedef void* hipStream_t;
constexpr hipError_t hipSuccess = 0;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
struct hipDeviceProp_t { char gcnArchName[64]; };

namespace simt_emu {
extern thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
struct BlockCtx {
  std::barrier<>* block_bar;
  std::vector<std::unique_ptr<std::barrier<>>> wave_bar;
  std::vector<std::vector<uint64_t>> xch;   // per wave, 2 x 64 exchange slots (a/b for MFMA)
};
extern thread_local BlockCtx* t_ctx;
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
extern int g_resident_blocks;      // > 0: the next launches run all their blocks concurrently (teams of the persistent kernels), own LDS each
float* block_lds();                // the calling work-item's block's dynamic LDS
inline int wave_id() { return (int)(t_threadIdx.x >> 6); }
inline int lane_id() { return (int)(t_threadIdx.x & 63); }
inline void wave_sync() { t_ctx->wave_bar[wave_id()]->arrive_and_wait(); }
template <typename T>
inline T shfl_any(T v, int src) {
  static_assert(sizeof(T) <= 8, "shfl payload");
  uint64_t raw = 0;
  std::memcpy(&raw, &v, sizeof(T));
  auto& x = t_ctx->xch[wave_id()];
  x[lane_id()] = raw;
  wave_sync();
  uint64_t got = x[src & 63];
  wave_sync();
  T out;
  std::memcpy(&out, &got, sizeof(T));
  return out;
}
}  // namespace simt_emu

#define threadIdx (simt_emu::t_threadIdx)
#define blockIdx (simt_emu::t_blockIdx)
#define blockDim (simt_emu::t_blockDim)
#define gridDim (simt_emu::t_gridDim)

static inline void __syncthreads() { simt_emu::t_ctx->block_bar->arrive_and_wait(); }
// wave-scope LDS hand-off: lockstep on hardware, a real per-wave barrier here
static inline void __builtin_amdgcn_wave_barrier() { simt_emu::wave_sync(); }
static inline void __builtin_amdgcn_fence(int, const char*) { std::atomic_thread_fence(std::memory_order_seq_cst); }
template <typename T> static inline T __shfl(T v, int src) { return simt_emu::shfl_any(v, src); }
template <typename T> static inline T __shfl_xor(T v, int mask) { return simt_emu::shfl_any(v, simt_emu::lane_id() ^ mask); }
template <typename T> static inline T __shfl_down(T v, int d) { int s = simt_emu::lane_id() + d; return simt_emu::shfl_any(v, s > 63 ? simt_emu::lane_id() : s); }
// wave vote (every lane of the wavefront must call it, as with the shuffles)
static inline int __any(int pred) {
  int v = pred != 0;
  for (int off = 32; off >= 1; off >>= 1) v |= simt_emu::shfl_any(v, simt_emu::lane_id() ^ off);
  return v;
}
static inline int __all(int pred) { return !__any(!pred); }
// (every lane of the wavefront calls it: lane 0's value)
template <typename T> static inline T __builtin_amdgcn_readfirstlane(T v) { return simt_emu::shfl_any(v, 0); }
static inline float atomicAdd(float* p, float v) {
  std::atomic_ref<float> r(*p);
  float old = r.load();
  while (!r.compare_exchange_weak(old, old + v)) {}
  return old;
}
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / sqrtf(x); }
static inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
// (resident teams: a polling work-item sleeps instead of spinning -- thousands of OS threads share a few cores with the producers they wait for)
static inline void __builtin_amdgcn_s_sleep(int) {
  if (simt_emu::g_resident_blocks > 0) std::this_thread::sleep_for(std::chrono::milliseconds(20));
  else std::this_thread::yield();
}
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return std::atomic_ref<unsigned>(*p).fetch_add(v); }
// buffer descriptor + raw buffer loads (the exchange sweeps of rollout_persist.hip): base pointer + byte offset
struct __amdgpu_buffer_rsrc_t { const unsigned char* base; };
static inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(const void* p, int, int, int) { return __amdgpu_buffer_rsrc_t{static_cast<const unsigned char*>(p)}; }
typedef unsigned emu_uv4 __attribute__((ext_vector_type(4)));
typedef unsigned emu_uv2 __attribute__((ext_vector_type(2)));
static inline void __builtin_amdgcn_raw_buffer_store_b128(emu_uv4 v, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, int) {
  std::memcpy(const_cast<unsigned char*>(rs.base) + voff + soff, &v, 16);
}
static inline void __builtin_amdgcn_raw_buffer_store_b64(emu_uv2 v, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, int) {
  std::memcpy(const_cast<unsigned char*>(rs.base) + voff + soff, &v, 8);
}
static inline emu_uv4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, int) {
  emu_uv4 v;
  std::memcpy(&v, rs.base + voff + soff, 16);
  return v;
}

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D col = l&31, row = (r&3)+8*(r>>2)+4*(l>>5)
typedef float f32x16_emu __attribute__((ext_vector_type(16)));
typedef float f32x4_emu __attribute__((ext_vector_type(4)));
static inline f32x16_emu __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, f32x16_emu c, int, int, int) {
  auto& x = simt_emu::t_ctx->xch[simt_emu::wave_id()];
  const int l = simt_emu::lane_id();
  uint32_t ra, rb;
  std::memcpy(&ra, &a, 4);
  std::memcpy(&rb, &b, 4);
  x[l] = ra;
  x[64 + l] = rb;
  simt_emu::wave_sync();
  f32x16_emu d = c;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    float acc = c[r];
    for (int k = 0; k < 2; ++k) {
      uint32_t ua = (uint32_t)x[row + 32 * k], ub = (uint32_t)x[64 + col + 32 * k];
      float fa, fb;
      std::memcpy(&fa, &ua, 4);
      std::memcpy(&fb, &ub, 4);
      acc = fmaf(fa, fb, acc);
    }
    d[r] = acc;
  }
  simt_emu::wave_sync();
  return d;
}
// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D col = l&15, row = 4*(l>>4)+r
static inline f32x4_emu __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, f32x4_emu c, int, int, int) {
  auto& x = simt_emu::t_ctx->xch[simt_emu::wave_id()];
  const int l = simt_emu::lane_id();
  uint32_t ra, rb;
  std::memcpy(&ra, &a, 4);
  std::memcpy(&rb, &b, 4);
  x[l] = ra;
  x[64 + l] = rb;
  simt_emu::wave_sync();
  f32x4_emu d = c;
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * (l >> 4) + r, col = l & 15;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) {
      uint32_t ua = (uint32_t)x[row + 16 * k], ub = (uint32_t)x[64 + col + 16 * k];
      float fa, fb;
      std::memcpy(&fa, &ua, 4);
      std::memcpy(&fb, &ub, 4);
      acc = fmaf(fa, fb, acc);
    }
    d[r] = acc;
  }
  simt_emu::wave_sync();
  return d;
}

// v_mfma_f32_4x4x1_16b_f32: 16 independent blocks of four lanes; in block b lane 4 b + j gets D[i] = C[i] + A(lane 4 b + i) * B(lane 4 b + j), i = 0..3
static inline f32x4_emu __builtin_amdgcn_mfma_f32_4x4x1f32(float a, float b, f32x4_emu c, int, int, int) {
  const int l = simt_emu::lane_id();
  f32x4_emu d = c;
  for (int i = 0; i < 4; ++i) d[i] = fmaf(simt_emu::shfl_any(a, (l & ~3) | i), b, c[i]);
  return d;
}

// ---- runtime API subset ---------------------------------------------------------------------------
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "simt_emu"; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = std::aligned_alloc(256, (n + 255) / 256 * 256); return *p ? hipSuccess : 1; }
static inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
typedef int hipEvent_t;
enum { hipEventDisableTiming = 2, hipStreamNonBlocking = 1 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = 0; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = hipStream_t(); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { std::snprintf(p->gcnArchName, 64, "simt_emu"); return hipSuccess; }

// dynamic LDS: every kernel declares `extern __shared__ float smem[]` inside namespace ha
namespace ha { extern float smem[]; }

#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
  simt_emu::launch(dim3(grid), dim3(block), [&]() { kernel(__VA_ARGS__); })

// Read+write streaming ceiling on one MI355X for the access geometries the LBS skinning kernel can choose from.
// Standalone: hipcc --offload-arch=gfx950 -O3 -o hbm_stream hbm_stream.hip ; ./hbm_stream [MB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float vf4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// VPT float4 per thread.  STRIDED: thread owns VPT consecutive float4 (the "direct" 48-byte pattern for VPT=3);
// otherwise lane-contiguous (thread t owns float4 t, t+256, ... of the block window).
// SYNC: a __syncthreads between the loads and the stores (what the skinning kernel has).
template <int VPT, bool STRIDED, bool NTS, bool NTL, bool SYNC>
__global__ __launch_bounds__(256) void copy_kernel(const vf4* __restrict__ src, vf4* __restrict__ dst, size_t n4, int nwin) {
  extern __shared__ float smem[];
  for (int win = blockIdx.x; win < nwin; win += gridDim.x) {
    const size_t base = (size_t)win * 256 * VPT;
    vf4 v[VPT];
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      const size_t i = base + (STRIDED ? (size_t)threadIdx.x * VPT + k : (size_t)threadIdx.x + 256 * k);
      if (i < n4) v[k] = NTL ? __builtin_nontemporal_load(src + i) : src[i];
    }
    if (SYNC) {
      if (threadIdx.x == 0) smem[0] = v[0].x;
      __syncthreads();
      if (smem[0] == 12345.678f) v[0].y += 1.f;
    }
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      const size_t i = base + (STRIDED ? (size_t)threadIdx.x * VPT + k : (size_t)threadIdx.x + 256 * k);
      if (i < n4) {
        if (NTS) __builtin_nontemporal_store(v[k], dst + i);
        else dst[i] = v[k];
      }
    }
  }
}

struct Cfg { const char* name; void (*fn)(const vf4*, vf4*, size_t, int); int vpt; };

template <int VPT, bool STRIDED, bool NTS, bool NTL, bool SYNC>
Cfg mk(const char* name) { return Cfg{name, copy_kernel<VPT, STRIDED, NTS, NTL, SYNC>, VPT}; }

int main(int argc, char** argv) {
  const double mb = argc > 1 ? atof(argv[1]) : 158.7456;
  const size_t n4 = (size_t)(mb * 1e6 / 16);
  vf4 *src, *dst;
  CK(hipMalloc(&src, n4 * 16 + 64));
  CK(hipMalloc(&dst, n4 * 16 + 64));
  CK(hipMemset(src, 1, n4 * 16));
  CK(hipMemset(dst, 0, n4 * 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<Cfg> cfgs = {
      mk<1, false, false, false, false>("v1 lane      "), mk<1, false, true, false, false>("v1 lane  nts "),
      mk<3, true, false, false, false>("v3 strided   "), mk<3, true, true, false, false>("v3 strided nts"),
      mk<3, true, true, false, true>("v3 strd nts sync"), mk<3, false, false, false, false>("v3 lane      "),
      mk<3, false, true, false, false>("v3 lane nts  "), mk<3, false, true, false, true>("v3 lane nts sync"),
      mk<3, false, false, false, true>("v3 lane sync "), mk<6, false, true, false, false>("v6 lane nts  "),
      mk<6, false, true, false, true>("v6 lane nts sync"), mk<4, false, true, false, false>("v4 lane nts  "),
      mk<4, false, true, true, false>("v4 lane nts ntl"), mk<12, false, true, false, true>("v12 lane nts sync"),
  };
  const int lds_opts[] = {0, 20 * 1024, 40 * 1024, 80 * 1024};         // blocks/CU cap: 8(max), 8, 4, 2
  const int grid_opts[] = {0, 256 * 8, 256 * 4, 256 * 2};                 // 0 = one window per block, else persistent
  printf("copy of %.1f MB (read) + same (write); GB/s counts both directions\n", n4 * 16 / 1e6);
  for (auto& c : cfgs) {
    for (int lds : lds_opts) {
      for (int grid : grid_opts) {
        const int nwin = (int)((n4 + 256 * (size_t)c.vpt - 1) / (256 * (size_t)c.vpt));
        const int g = grid == 0 ? nwin : (grid < nwin ? grid : nwin);
        if (lds && hipFuncSetAttribute((const void*)c.fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) continue;
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(c.fn, dim3(g), dim3(256), lds, 0, src, dst, n4, nwin);
        CK(hipDeviceSynchronize());
        float best = 1e9f, tot = 0.f;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipEventRecord(e0));
          for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(c.fn, dim3(g), dim3(256), lds, 0, src, dst, n4, nwin);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          ms /= 10;
          tot += ms;
          if (ms < best) best = ms;
        }
        printf("%-18s lds=%5d grid=%6d : best %7.1f us  mean %7.1f us  -> %6.0f GB/s\n", c.name, lds, g, best * 1e3, tot / 3 * 1e3,
               2.0 * n4 * 16 / (best * 1e-3) / 1e9);
      }
    }
  }
  return 0;
}

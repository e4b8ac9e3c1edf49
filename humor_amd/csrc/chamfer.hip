// Chamfer distance (nearest-neighbour search in both directions + gradient scatter) on gfx950 -- replaces the reference's only
// native code on the fitting path, humor/utils/chamfer_distance/chamfer_distance.cu (ChamferDistanceKernel :6-137,
// ChamferDistanceGradKernel :158-187) as consumed by FittingLoss.points3d_loss (fitting_loss.py:378-396).
//
// Forward: one thread per query point; the searched cloud is staged through LDS as x / y / z planes in chunks of 1024 points and
// read with 16-byte broadcast reads (every lane of the wave reads the same address: no bank conflicts, 4 candidates per read).
// The squared distance is (x2*x2 + y2*y2) + z2*z2 in fp32 WITHOUT fused multiply-add (the library is built with
// -ffp-contract=off): bit-identical to the reference's CPU path (chamfer_distance.cpp:59-87) and therefore the same int32 argmin
// on every input, ties included (strict '<' while scanning in index order = the lowest index wins, as in both reference paths).
// VALU-bound: ~10 instructions per (query, candidate) pair.
// Backward: gather for the query's own gradient, float atomics for the scattered half (as the reference).
#include "common.h"

namespace ha {

constexpr int CH_CHUNK = 1024;      // candidates staged per pass (12 KiB of LDS)
typedef float cvf4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void chamfer_nn_kernel(int n, const float* __restrict__ xyz, int m, const float* __restrict__ xyz2,
                                                         float* __restrict__ result, int* __restrict__ result_i) {
  extern __shared__ __attribute__((aligned(16))) float smem[];     // [3][CH_CHUNK]
  float* sx = smem;
  float* sy = smem + CH_CHUNK;
  float* sz = smem + 2 * CH_CHUNK;
  const int i = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const bool live = j < n;
  float x1 = 0.f, y1 = 0.f, z1 = 0.f;
  if (live) {
    const float* q = xyz + ((size_t)i * n + j) * 3;
    x1 = q[0]; y1 = q[1]; z1 = q[2];
  }
  const float* P = xyz2 + (size_t)i * m * 3;
  // candidate 0 seeds the search unconditionally, as in the reference's scan (`k == 0 || d < best`): a NaN distance there stays
  float best;
  int best_i = 0;
  {
    const float x2 = P[0] - x1, y2 = P[1] - y1, z2 = P[2] - z1;
    best = (x2 * x2 + y2 * y2) + z2 * z2;
  }
  for (int k2 = 0; k2 < m; k2 += CH_CHUNK) {
    const int cnt = m - k2 < CH_CHUNK ? m - k2 : CH_CHUNK;
    __syncthreads();
    for (int e = threadIdx.x; e < cnt * 3; e += 256) {          // coalesced read of the [cnt][3] run, de-interleaved into planes
      const float v = P[(size_t)k2 * 3 + e];
      const int p = e / 3, c = e - p * 3;
      (c == 0 ? sx : (c == 1 ? sy : sz))[p] = v;
    }
    __syncthreads();
    // four candidates per trip: their minimum against the running best; the bookkeeping (which of the four, the index) runs only
    // when some lane of the wavefront improves -- after the first few hundred candidates almost never (expected improvements per
    // query ~ ln m), so the steady-state trip is 3 LDS reads + 8 arithmetic instructions per candidate pair of lanes.  Strict `<`
    // and "first of the four that equals the minimum" keep the reference's tie rule (lowest index); fminf drops NaN distances as
    // `d < best` does.  (Dead lanes run along: the vote is wave-wide.)
    const int cnt4 = cnt & ~3;
    const cvf4 x1v = {-x1, -x1, -x1, -x1}, y1v = {-y1, -y1, -y1, -y1}, z1v = {-z1, -z1, -z1, -z1};
    for (int k = 0; k < cnt4; k += 4) {
      const cvf4 X = *reinterpret_cast<const cvf4*>(sx + k), Y = *reinterpret_cast<const cvf4*>(sy + k), Z = *reinterpret_cast<const cvf4*>(sz + k);
      // (4-vectors: the compiler issues the packed fp32 forms, v_pk_add_f32 / v_pk_mul_f32 -- IEEE results identical to the scalar ones)
      const cvf4 dx = X + x1v, dy = Y + y1v, dz = Z + z1v;        // x1v = -x1: a - b == a + (-b) exactly
      const cvf4 d = (dx * dx + dy * dy) + dz * dz;
      const float mn = fminf(fminf(d[0], d[1]), fminf(d[2], d[3]));
      const bool better = mn < best;
      if (__any(better)) {
        const int uu = d[0] == mn ? 0 : (d[1] == mn ? 1 : (d[2] == mn ? 2 : 3));
        if (better) { best = mn; best_i = k2 + k + uu; }
      }
    }
    for (int k = cnt4; k < cnt; ++k) {
      const float x2 = sx[k] - x1, y2 = sy[k] - y1, z2 = sz[k] - z1;
      const float d = (x2 * x2 + y2 * y2) + z2 * z2;
      if (d < best) { best = d; best_i = k2 + k; }
    }
  }
  if (live) {
    result[(size_t)i * n + j] = best;
    result_i[(size_t)i * n + j] = best_i;
  }
}

// gradient of dist[i][j] = |p1_j - p2_{idx_j}|^2 : own point by gather (written, not accumulated), matched point by atomics
__global__ void chamfer_grad_kernel(int n, const float* __restrict__ xyz1, int m, const float* __restrict__ xyz2,
                                    const float* __restrict__ grad_dist, const int* __restrict__ idx, float* __restrict__ g_self,
                                    float* __restrict__ g_other, int accumulate_self) {
  const int i = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const size_t a = ((size_t)i * n + j) * 3;
  const int j2 = idx[(size_t)i * n + j];
  const size_t b = ((size_t)i * m + j2) * 3;
  const float g = grad_dist[(size_t)i * n + j] * 2.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = g * (xyz1[a + c] - xyz2[b + c]);
    if (accumulate_self) atomicAdd(&g_self[a + c], v);
    else g_self[a + c] = v;
    atomicAdd(&g_other[b + c], -v);
  }
}

}  // namespace ha

using namespace ha;

extern "C" int ha_chamfer_forward(int b, int n, const float* xyz1, int m, const float* xyz2, float* dist1, int32_t* idx1, float* dist2,
                                  int32_t* idx2, void* stream) {
  HA_REQUIRE(b >= 0 && n >= 1 && m >= 1, "ha_chamfer_forward: need b >= 0, n >= 1, m >= 1");
  HA_REQUIRE(xyz1 && xyz2 && dist1 && idx1 && dist2 && idx2, "ha_chamfer_forward: null argument");
  HA_REQUIRE(b <= 65535, "ha_chamfer_forward: at most 65535 clouds per call");
  if (b == 0) return HA_OK;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = 3 * CH_CHUNK * sizeof(float);
  HA_LAUNCH(chamfer_nn_kernel, dim3(ceil_div(n, 256), b), dim3(256), lds, st, n, xyz1, m, xyz2, dist1, idx1);
  HA_LAUNCH_CHECK();
  HA_LAUNCH(chamfer_nn_kernel, dim3(ceil_div(m, 256), b), dim3(256), lds, st, m, xyz2, n, xyz1, dist2, idx2);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

extern "C" int ha_chamfer_backward(int b, int n, const float* xyz1, int m, const float* xyz2, const float* grad_dist1, const int32_t* idx1,
                                   const float* grad_dist2, const int32_t* idx2, float* grad_xyz1, float* grad_xyz2, void* stream) {
  HA_REQUIRE(b >= 0 && n >= 1 && m >= 1, "ha_chamfer_backward: need b >= 0, n >= 1, m >= 1");
  HA_REQUIRE(xyz1 && xyz2 && grad_dist1 && idx1 && grad_dist2 && idx2 && grad_xyz1 && grad_xyz2, "ha_chamfer_backward: null argument");
  HA_REQUIRE(b <= 65535, "ha_chamfer_backward: at most 65535 clouds per call");
  if (b == 0) return HA_OK;
  hipStream_t st = (hipStream_t)stream;
  // pass 1 writes grad_xyz1 (own points of direction 1) and needs grad_xyz2 zeroed for its scattered half; pass 2 accumulates both
  zero_async(grad_xyz2, (size_t)b * m * 3 * sizeof(float), st);      // (a kernel: see common.h)
  HA_LAUNCH(chamfer_grad_kernel, dim3(ceil_div(n, 256), b), dim3(256), 0, st, n, xyz1, m, xyz2, grad_dist1, idx1, grad_xyz1, grad_xyz2, 0);
  HA_LAUNCH_CHECK();
  HA_LAUNCH(chamfer_grad_kernel, dim3(ceil_div(m, 256), b), dim3(256), 0, st, m, xyz2, n, xyz1, grad_dist2, idx2, grad_xyz2, grad_xyz1, 1);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

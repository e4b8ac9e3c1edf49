// L-BFGS two-loop recursion in coefficient form.  torch.optim.LBFGS (the optimiser the reference drives,
// humor/fitting/motion_optimizer.py:233-254, 284-310, 461-512) evaluates the recursion as 2 x history dependent dot products and
// 2 x history axpys on the n-vector -- ~400 tiny launches per inner iteration at history 100, which is most of an outer
// iteration of stages 1-2.  Here the direction is a linear combination of {g, s_i, y_i}:
//     q = -g - sum_j al_j y_j,   r = H q + sum_j (al_j - be_j) s_j   =>   d = -H g - sum_j (H al_j) y_j + sum_j (al_j - be_j) s_j
// whose coefficients only need inner products that are kept in a Gram matrix G = [S;Y][S;Y]^T (two matrix-vector products when a
// pair is added) and Mg = [S;Y] g (one per iteration):
//     al_i = ro_i ( -s_i.g - sum_{j newer than i} al_j s_i.y_j )
//     be_i = ro_i ( H ( -y_i.g - sum_j al_j y_i.y_j ) + sum_{j older than i} (al_j - be_j) y_i.s_j ),   ro_i = 1 / y_i.s_i
// This kernel runs the two k-step recurrences on ONE wavefront (k <= 128; lanes split the inner sums) and writes the 2k
// coefficients; the caller finishes with d = [S;Y]^T coef - H g (one GEMV).  Same arithmetic as the two-loop recursion up to the
// order of the fp32 summations.
#include <string.h>

#include "common.h"

namespace ha {

constexpr int LB_MAXH = 128;
struct LbfgsArgs {
  int hist, num_old;
  int order[LB_MAXH];        // physical slot of the i-th oldest pair
  const float* G;            // [2 hist][2 hist]: rows/cols 0..hist-1 = s slots, hist..2 hist-1 = y slots
  const float* Mg;           // [2 hist]
  float h_diag;
  const float* h_diag_dev;   // when non-null: the scale is read from device memory (no host round trip)
  float* coef;               // [2 hist]
};

__device__ __forceinline__ float lb_wsum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__global__ __launch_bounds__(64) void lbfgs_coeffs_kernel(LbfgsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // al[LB_MAXH] | ab[LB_MAXH] (= al - be)
  float* al = smem;
  float* ab = smem + LB_MAXH;
  const int lane = threadIdx.x, h = a.hist, k = a.num_old, W = 2 * h;
  const float h_diag = a.h_diag_dev ? a.h_diag_dev[0] : a.h_diag;
  for (int i = lane; i < 2 * h; i += 64) a.coef[i] = 0.f;
  // first loop: newest -> oldest
  for (int i = k - 1; i >= 0; --i) {
    const int pi = a.order[i];
    float part = 0.f;
    for (int j = i + 1 + lane; j < k; j += 64) part += al[j] * a.G[(size_t)pi * W + h + a.order[j]];      // al_j s_i.y_j
    const float s = lb_wsum(part);
    const float ro = 1.0f / a.G[(size_t)pi * W + h + pi];
    if (lane == 0) al[i] = ro * (-a.Mg[pi] - s);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  // second loop: oldest -> newest
  for (int i = 0; i < k; ++i) {
    const int pi = a.order[i];
    float p1 = 0.f, p2 = 0.f;
    for (int j = lane; j < k; j += 64) p1 += al[j] * a.G[(size_t)(h + pi) * W + h + a.order[j]];          // al_j y_i.y_j
    for (int j = lane; j < i; j += 64) p2 += ab[j] * a.G[(size_t)(h + pi) * W + a.order[j]];              // (al_j - be_j) y_i.s_j
    const float s1 = lb_wsum(p1), s2 = lb_wsum(p2);
    const float ro = 1.0f / a.G[(size_t)pi * W + h + pi];
    if (lane == 0) {
      const float be = ro * (h_diag * (-a.Mg[h + pi] - s1) + s2);
      ab[i] = al[i] - be;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  for (int i = lane; i < k; i += 64) {
    const int pi = a.order[i];
    a.coef[pi] = ab[i];                       // s_i
    a.coef[h + pi] = -h_diag * al[i];       // y_i
  }
}

}  // namespace ha

extern "C" int ha_lbfgs_coeffs(int hist, int num_old, const int32_t* order, const float* G, const float* Mg, float h_diag,
                               const float* h_diag_dev, float* coef, void* stream) {
  using namespace ha;
  HA_REQUIRE(hist >= 1 && hist <= LB_MAXH, "ha_lbfgs_coeffs: history size must be in [1, %d]", LB_MAXH);
  HA_REQUIRE(num_old >= 0 && num_old <= hist, "ha_lbfgs_coeffs: num_old out of range");
  HA_REQUIRE(G && Mg && coef && (order || num_old == 0), "ha_lbfgs_coeffs: null argument");
  LbfgsArgs a;
  memset(&a, 0, sizeof(a));
  a.hist = hist; a.num_old = num_old;
  for (int i = 0; i < num_old; ++i) {
    HA_REQUIRE(order[i] >= 0 && order[i] < hist, "ha_lbfgs_coeffs: slot index out of range");
    a.order[i] = order[i];
  }
  a.G = G; a.Mg = Mg; a.h_diag = h_diag; a.h_diag_dev = h_diag_dev; a.coef = coef;
  hipLaunchKernelGGL(lbfgs_coeffs_kernel, dim3(1), dim3(64), 2 * LB_MAXH * sizeof(float), (hipStream_t)stream, a);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

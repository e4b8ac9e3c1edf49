// Negative log-likelihood of the stage-3 initial state under the init-state Gaussian mixture, value and gradient
// (humor/fitting/fitting_loss.py:416-429 init_motion_prior_loss: -sum_b MixtureSameFamily(Categorical(w),
// MultivariateNormal(mu, cov)).log_prob(x_b), x_b = joints | joints_vel | trans_vel | root_orient_vel of frame 0, 138 floats).
//
//   lp[b][k] = c_k - 0.5 |Linv_k (x_b - mu_k)|^2,   c_k = log w_k - log det L_k - D/2 log 2 pi,   L_k = chol(cov_k)
//   nll_b    = -logsumexp_k lp[b][k],    d nll_b / dx = sum_k softmax_k(lp[b]) Linv_k^T Linv_k (x_b - mu_k)
//
// Two launches where the op-by-op evaluation (broadcast difference, batched mat-vec, squares, logsumexp and their autograd) takes 34:
//   gmm_comp_kernel   one block per (sequence, component): y = Linv (x - mu) and Linv^T y as two passes over the 76 KB factor,
//                     each wave a slice of the contraction index, every load a coalesced row segment;
//   gmm_mix_kernel    one block per sequence: logsumexp over the components, the mixture-weighted gradient.
#include "common.h"

namespace ha {

namespace {

constexpr int GMM_WAVES = 8, GMM_NI = 4;     // D <= 64 * GMM_NI

__device__ __forceinline__ float gmm_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// out[i] = sum_j M[j][i] * v[j] over the block: wave w takes the rows j = w, w + GMM_WAVES, ...; partials meet in `part` [GMM_WAVES][D]
__device__ __forceinline__ void gmm_matvec_t(const float* __restrict__ M, const float* v, float* part, int D, int wave, int lane) {
  float acc[GMM_NI];
#pragma unroll
  for (int q = 0; q < GMM_NI; ++q) acc[q] = 0.f;
  // (latency, not bandwidth, bounds this kernel: eight rows = up to 32 loads in flight per trip; out-of-range lanes / rows read a
  // valid address and contribute zero)
  for (int j0 = wave; j0 < D; j0 += 8 * GMM_WAVES) {
    float m[8][GMM_NI], vj[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + u * GMM_WAVES;
      const bool on = j < D;
      vj[u] = on ? v[j] : 0.f;
      const float* row = M + (size_t)(on ? j : 0) * D;
#pragma unroll
      for (int q = 0; q < GMM_NI; ++q) {
        const int i = lane + 64 * q;
        m[u][q] = (64 * q < D) ? row[i < D ? i : 0] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int q = 0; q < GMM_NI; ++q) acc[q] = fmaf(m[u][q], vj[u], acc[q]);
  }
#pragma unroll
  for (int q = 0; q < GMM_NI; ++q) {
    const int i = lane + 64 * q;
    if (i < D) part[wave * D + i] = acc[q];
  }
}

}  // namespace

__global__ __launch_bounds__(GMM_WAVES * 64) void gmm_comp_kernel(ha_gmm_args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int D = a.D;
  float* diff = smem;                 // [D]
  float* y = diff + D;                // [D]
  float* part = y + D;                // [GMM_WAVES][D]
  float* red = part + GMM_WAVES * D;  // [GMM_WAVES]
  const int b = blockIdx.x / a.K, k = blockIdx.x - b * a.K, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // x_b: the segments side by side
  for (int i = tid; i < D; i += GMM_WAVES * 64) {
    int o = i, sidx = 0;
    while (sidx + 1 < a.nseg && o >= a.seg_width[sidx]) { o -= a.seg_width[sidx]; ++sidx; }
    diff[i] = a.seg[sidx][(size_t)b * a.seg_stride[sidx] + o] - a.means[(size_t)k * D + i];
  }
  __syncthreads();
  // y = Linv (x - mu): contraction over the columns of Linv = the rows of its transpose
  gmm_matvec_t(a.LinvT + (size_t)k * D * D, diff, part, D, wave, lane);
  __syncthreads();
  float q = 0.f;
  for (int i = tid; i < D; i += GMM_WAVES * 64) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < GMM_WAVES; ++w) s += part[w * D + i];
    y[i] = s;
    q = fmaf(s, s, q);
  }
  q = gmm_wave_sum(q);
  if (lane == 0) red[wave] = q;
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < GMM_WAVES; ++w) s += red[w];
    a.lp[(size_t)b * a.K + k] = a.cst[k] - 0.5f * s;
  }
  // d(-lp)/dx = Linv^T y: contraction over the rows of Linv
  gmm_matvec_t(a.Linv + (size_t)k * D * D, y, part, D, wave, lane);
  __syncthreads();
  for (int i = tid; i < D; i += GMM_WAVES * 64) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < GMM_WAVES; ++w) s += part[w * D + i];
    a.gpart[((size_t)b * a.K + k) * D + i] = s;
  }
}

__global__ __launch_bounds__(256) void gmm_mix_kernel(ha_gmm_args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];      // (dynamic LDS only: the host emulator tier has no static LDS)
  float* wk = smem;                                                   // [64] mixture weights
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < 64) {
    const float v = tid < a.K ? a.lp[(size_t)b * a.K + tid] : -INFINITY;
    float m = v;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    const float e = tid < a.K ? expf(v - m) : 0.f;
    const float s = gmm_wave_sum(e);
    wk[tid] = e / s;
    if (tid == 0) a.nll[b] = -(m + logf(s));
  }
  __syncthreads();
  for (int i = tid; i < a.D; i += 256) {
    float g = 0.f;
    const float* gp = a.gpart + (size_t)b * a.K * a.D + i;
    for (int k0 = 0; k0 < a.K; k0 += 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = gp[(size_t)(k0 + u < a.K ? k0 + u : 0) * a.D];
#pragma unroll
      for (int u = 0; u < 8; ++u) g = fmaf(k0 + u < a.K ? wk[k0 + u] : 0.f, t[u], g);
    }
    a.g_x[(size_t)b * a.D + i] = g;
  }
}

}  // namespace ha

using namespace ha;

extern "C" int ha_gmm_nll(const ha_gmm_args* args, void* stream) {
  HA_REQUIRE(args, "ha_gmm_nll: null argument");
  const ha_gmm_args& a = *args;
  HA_REQUIRE(a.B >= 1 && a.K >= 1 && a.K <= 64 && a.D >= 1 && a.D <= 64 * GMM_NI, "ha_gmm_nll: need B >= 1, 1 <= K <= 64, 1 <= D <= %d", 64 * GMM_NI);
  HA_REQUIRE(a.nseg >= 1 && a.nseg <= 4, "ha_gmm_nll: 1..4 input segments");
  int width = 0;
  for (int s = 0; s < a.nseg; ++s) {
    HA_REQUIRE(a.seg[s] && a.seg_width[s] >= 1 && a.seg_stride[s] >= a.seg_width[s], "ha_gmm_nll: segment %d is null or has a stride below its width", s);
    width += a.seg_width[s];
  }
  HA_REQUIRE(width == a.D, "ha_gmm_nll: the segment widths add up to %d, the mixture has %d dimensions", width, a.D);
  HA_REQUIRE(a.means && a.Linv && a.LinvT && a.cst && a.lp && a.gpart && a.nll && a.g_x, "ha_gmm_nll: null tensor");
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = ((size_t)(2 + GMM_WAVES) * a.D + GMM_WAVES) * sizeof(float);
  HA_LAUNCH(gmm_comp_kernel, dim3(a.B * a.K), dim3(GMM_WAVES * 64), lds, st, a);
  HA_LAUNCH_CHECK();
  HA_LAUNCH(gmm_mix_kernel, dim3(a.B), dim3(256), 64 * sizeof(float), st, a);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

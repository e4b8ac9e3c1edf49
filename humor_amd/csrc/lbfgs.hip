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
  float* coef;               // [2 hist] (+1 in pair mode: coef[2 hist] = -h_diag, the coefficient of the gradient row)
  // pair mode (ha_lbfgs_pair_coeffs): P [2 hist][3] = M s | M y | M g of the pair just written to `slot`
  const float* P;
  int slot;
  float* Gw;                 // = G, writable
  float* Mgw;                // [2 hist], written from P
  float* scal;               // scal[0] = y.s, scal[1] = y.y
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
  float h_diag = a.h_diag_dev ? a.h_diag_dev[0] : a.h_diag;
  if (a.P) {
    // the pair (s, y) has just been written to rows slot / hist + slot of M: its Gram rows and columns, M g, and the scale y.s / y.y
    for (int i = lane; i < W; i += 64) {
      const float vs = a.P[i * 3], vy = a.P[i * 3 + 1];
      a.Gw[(size_t)a.slot * W + i] = vs;
      a.Gw[(size_t)i * W + a.slot] = vs;
      a.Gw[(size_t)(h + a.slot) * W + i] = vy;
      a.Gw[(size_t)i * W + h + a.slot] = vy;
      a.Mgw[i] = a.P[i * 3 + 2];
    }
    const float ys = a.P[(h + a.slot) * 3], yy = a.P[(h + a.slot) * 3 + 1];
    h_diag = ys / yy;
    if (lane == 0) { a.scal[0] = ys; a.scal[1] = yy; a.coef[W] = -h_diag; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // the Gram entries written above are read back below by other lanes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
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

// ---- one pass over the history for the three products the update needs ----------------------------------------------------------
// P[r][k] = M[r] . V_k for r < rows, V = (M[i0], M[i1], M[i2]) (the new s, the new y and the current gradient row).  torch issues a
// matrix-vector product per right-hand side (three reads of the 60-76 MB history); here one read, fixed summation order (column
// chunks of 512 in chunk order: the replicated multi-GPU optimiser needs bit-identical directions on every rank -- no atomics).
constexpr int GR_CW = 512;
__global__ __launch_bounds__(256) void lbfgs_gram_partial_kernel(const float* __restrict__ M, int n, int rows, int i0, int i1, int i2,
                                                                 float* __restrict__ part) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c0 = blockIdx.x * GR_CW;
  float v[3][8];
  const int idx[3] = {i0, i1, i2};
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c0 + (e >> 2) * 256 + lane * 4 + (e & 3);
      v[k][e] = c < n ? M[(size_t)idx[k] * n + c] : 0.f;
    }
  for (int r = wave; r < rows; r += 4) {
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c0 + (e >> 2) * 256 + lane * 4 + (e & 3);
      m[e] = c < n ? M[(size_t)r * n + c] : 0.f;
    }
    float p[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) p[k] = fmaf(m[e], v[k][e], p[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = lb_wsum(p[k]);
    if (lane == 0) {
      float* dst = part + ((size_t)blockIdx.x * rows + r) * 3;
      dst[0] = p[0]; dst[1] = p[1]; dst[2] = p[2];
    }
  }
}

__global__ __launch_bounds__(64) void lbfgs_gram_reduce_kernel(const float* __restrict__ part, int nb, int rows, float* __restrict__ P) {
  const int r = blockIdx.x, lane = threadIdx.x;
  float p[3] = {0.f, 0.f, 0.f};
  for (int b = lane; b < nb; b += 64) {
    const float* src = part + ((size_t)b * rows + r) * 3;
    p[0] += src[0]; p[1] += src[1]; p[2] += src[2];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) p[k] = lb_wsum(p[k]);
  if (lane == 0) { P[r * 3] = p[0]; P[r * 3 + 1] = p[1]; P[r * 3 + 2] = p[2]; }
}

// out[0] = a.b, out[1] = max|a|, out[2] = sum|a|, out[3] = *extra (or 0): every scalar the line search reads after a closure
// evaluation (loss, g.d, max|g|) or after a direction update (g.d, max|d|) in ONE launch (torch: dot, abs, max, sum, stack).
__global__ __launch_bounds__(1024) void lbfgs_scalars_kernel(int n, const float* __restrict__ a, const float* __restrict__ b,
                                                             const float* __restrict__ extra, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [3][16] per-wave partials
  float* sd = smem;
  float* sm = smem + 16;
  float* ss = smem + 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float d = 0.f, m = 0.f, su = 0.f;
  for (int i = tid; i < n; i += 1024) {
    const float x = a[i];
    d = fmaf(x, b[i], d);
    m = fmaxf(m, fabsf(x));
    su += fabsf(x);
  }
  d = lb_wsum(d);
  su = lb_wsum(su);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if (lane == 0) { sd[wave] = d; sm[wave] = m; ss[wave] = su; }
  __syncthreads();
  if (tid == 0) {
    float D = 0.f, Mx = 0.f, S = 0.f;
    for (int w = 0; w < 16; ++w) { D += sd[w]; Mx = fmaxf(Mx, sm[w]); S += ss[w]; }
    out[0] = D; out[1] = Mx; out[2] = S; out[3] = extra ? extra[0] : 0.f;
  }
}

}  // namespace ha

extern "C" int ha_lbfgs_gram(int n, int rows, const float* M, int i0, int i1, int i2, float* part, float* P, void* stream) {
  using namespace ha;
  HA_REQUIRE(n >= 1 && rows >= 1 && M && part && P, "ha_lbfgs_gram: bad argument");
  HA_REQUIRE(i0 >= 0 && i1 >= 0 && i2 >= 0, "ha_lbfgs_gram: negative row index");
  const int nb = ceil_div(n, GR_CW);
  hipLaunchKernelGGL(lbfgs_gram_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, M, n, rows, i0, i1, i2, part);
  HA_LAUNCH_CHECK();
  hipLaunchKernelGGL(lbfgs_gram_reduce_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, part, nb, rows, P);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

extern "C" int ha_lbfgs_gram_workspace(int n, int rows, int64_t* part_floats) {
  HA_REQUIRE(n >= 1 && rows >= 1 && part_floats, "ha_lbfgs_gram_workspace: bad argument");
  *part_floats = (int64_t)ha::ceil_div(n, ha::GR_CW) * rows * 3;
  return HA_OK;
}

extern "C" int ha_lbfgs_scalars(int n, const float* a, const float* b, const float* extra, float* out, void* stream) {
  using namespace ha;
  HA_REQUIRE(n >= 1 && a && b && out, "ha_lbfgs_scalars: bad argument");
  hipLaunchKernelGGL(lbfgs_scalars_kernel, dim3(1), dim3(1024), 48 * sizeof(float), (hipStream_t)stream, n, a, b, extra, out);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

extern "C" int ha_lbfgs_pair_coeffs(int hist, int num_old, const int32_t* order, int slot, const float* P, float* G, float* Mg,
                                    float* coef, float* scal, void* stream) {
  using namespace ha;
  HA_REQUIRE(hist >= 1 && hist <= LB_MAXH, "ha_lbfgs_pair_coeffs: history size must be in [1, %d]", LB_MAXH);
  HA_REQUIRE(num_old >= 1 && num_old <= hist && slot >= 0 && slot < hist, "ha_lbfgs_pair_coeffs: num_old / slot out of range");
  HA_REQUIRE(order && P && G && Mg && coef && scal, "ha_lbfgs_pair_coeffs: null argument");
  LbfgsArgs a;
  memset(&a, 0, sizeof(a));
  a.hist = hist; a.num_old = num_old;
  for (int i = 0; i < num_old; ++i) {
    HA_REQUIRE(order[i] >= 0 && order[i] < hist, "ha_lbfgs_pair_coeffs: slot index out of range");
    a.order[i] = order[i];
  }
  a.G = G; a.Mg = Mg; a.coef = coef;
  a.P = P; a.slot = slot; a.Gw = G; a.Mgw = Mg; a.scal = scal;
  hipLaunchKernelGGL(lbfgs_coeffs_kernel, dim3(1), dim3(64), 2 * LB_MAXH * sizeof(float), (hipStream_t)stream, a);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

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

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

// The two k-step recurrences are inherently sequential, so what matters is the cost of ONE step.  Walking the Gram matrix in global
// memory (two dependent round trips + a fenced LDS hand-off per step) cost 149 us at k = 100 -- a quarter of a stage-2 closure
// evaluation.  Here the block first gathers the three k x k blocks the recurrences read (s_i.y_j, y_i.y_j in age order; y_i.s_j is
// the transpose of the first) into LDS, all four waves form q_i = sum_j al_j y_i.y_j in parallel between the loops, and wave 0 walks
// the chains out of LDS: ~0.05 us per step.
// value of `v` in lane `src` (wave-uniform src) for every lane: v_readlane_b32 on hardware (a scalar register, ~10 cycles; the
// cross-lane shuffle goes through the LDS crossbar)
__device__ __forceinline__ float lb_bcast(float v, int src) {
#ifdef HA_SIMT_EMU
  return __shfl(v, src);
#else
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
#endif
}

// packed upper triangle (i <= j) of a k x k matrix
__device__ __forceinline__ int lb_tri(int i, int j, int k) { return i * k - (i * (i - 1)) / 2 + (j - i); }

__global__ __launch_bounds__(256) void lbfgs_coeffs_kernel(LbfgsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = a.hist, k = a.num_old, W = 2 * h;
  float* Gsy = smem;                                   // packed upper triangle of s_i . y_j (i <= j: the only entries the loops read)
  float* al = Gsy + (LB_MAXH * (LB_MAXH + 1)) / 2;     // [LB_MAXH]
  float* ab = al + LB_MAXH;                            // [LB_MAXH]  al - be
  float* qv = ab + LB_MAXH;                            // [LB_MAXH]  sum_j al_j y_i.y_j
  float* mgs = qv + LB_MAXH;                           // [LB_MAXH]  s_i . g
  float* mgy = mgs + LB_MAXH;                          // [LB_MAXH]  y_i . g
  int* s_order = reinterpret_cast<int*>(mgy + LB_MAXH);
  for (int i = tid; i < k; i += 256) s_order[i] = a.order[i];
  float h_diag = a.h_diag_dev ? a.h_diag_dev[0] : a.h_diag;
  const float* P = a.P;
  const int slot = a.slot;
  if (P) {
    // install the pair just written to `slot`: its Gram rows / columns, M g, and the scale H = y.s / y.y
    for (int i = tid; i < W; i += 256) {
      const float vs = P[i * 3], vy = P[i * 3 + 1];
      a.Gw[(size_t)slot * W + i] = vs;
      a.Gw[(size_t)i * W + slot] = vs;
      a.Gw[(size_t)(h + slot) * W + i] = vy;
      a.Gw[(size_t)i * W + h + slot] = vy;
      a.Mgw[i] = P[i * 3 + 2];
    }
    const float ys = P[(h + slot) * 3], yy = P[(h + slot) * 3 + 1];
    h_diag = ys / yy;
    if (tid == 0) { a.scal[0] = ys; a.scal[1] = yy; a.coef[W] = -h_diag; }
  }
  // the entries written above are read back below by other threads of this block (same CU: workgroup scope is the right one)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const float* G = a.Gw ? a.Gw : a.G;
  const float* Mg = a.Mgw ? a.Mgw : a.Mg;
  // s_i . y_j, i <= j, into LDS: thread = (column j, row parity), eight independent loads per round trip
  {
    const int j = tid & 127, ih = tid >> 7;
    const int pj = j < k ? s_order[j] : 0;
    for (int i0 = ih; i0 < k; i0 += 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 2 * u;
        v[u] = (i < k && j < k && i <= j) ? G[(size_t)s_order[i] * W + h + pj] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 2 * u;
        if (i < k && j < k && i <= j) Gsy[lb_tri(i, j, k)] = v[u];
      }
    }
  }
  for (int i = tid; i < k; i += 256) {
    const int pi = s_order[i];
    mgs[i] = Mg[pi];
    mgy[i] = Mg[h + pi];
  }
  for (int i = tid; i < W; i += 256) a.coef[i] = 0.f;
  __syncthreads();
  // first loop: newest -> oldest     al_i = ro_i ( -s_i.g - sum_{j newer than i} al_j s_i.y_j )
  // Column-oriented: lane l owns rows l and l + 64 and keeps their running sums in registers; a step is the owner's two FMAs, one
  // readlane broadcast of al_j and one LDS read per owned row -- no cross-lane reduction on the chain (the row-oriented form paid a
  // six-stage wave reduction + a division per step: 0.45 us x 2 k steps).
  const int r0 = lane, r1 = lane + 64;
  const float ro0 = r0 < k ? 1.0f / Gsy[lb_tri(r0, r0, k)] : 0.f, ro1 = r1 < k ? 1.0f / Gsy[lb_tri(r1, r1, k)] : 0.f;
  if (wave == 0) {
    float acc0 = 0.f, acc1 = 0.f;
    for (int j = k - 1; j >= 0; --j) {
      // al_j by its owner (its sum over the newer pairs is complete), then broadcast
      const float mine = j < 64 ? ro0 * (-(r0 < k ? mgs[r0] : 0.f) - acc0) : ro1 * (-(r1 < k ? mgs[r1] : 0.f) - acc1);
      const float aj = lb_bcast(mine, j & 63);
      if (lane == (j & 63)) al[j] = aj;
      if (r0 < j) acc0 = fmaf(aj, Gsy[lb_tri(r0, j, k)], acc0);
      if (r1 < j) acc1 = fmaf(aj, Gsy[lb_tri(r1, j, k)], acc1);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  __syncthreads();
  // q_i = sum_j al_j y_i.y_j for every i (independent of the second recurrence): all waves, eight rows of the Gram matrix per trip
  {
    const int j0 = lane, j1 = lane + 64;
    const int p0 = j0 < k ? s_order[j0] : 0, p1 = j1 < k ? s_order[j1] : 0;
    const float a0 = j0 < k ? al[j0] : 0.f, a1 = j1 < k ? al[j1] : 0.f;
    for (int i0 = wave; i0 < k; i0 += 32) {
      float g0[8], g1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 4 * u;
        const float* row = G + (size_t)(h + s_order[i < k ? i : 0]) * W + h;
        g0[u] = row[p0];
        g1[u] = row[p1];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 4 * u;
        const float sum = lb_wsum(fmaf(a0, g0[u], a1 * g1[u]));
        if (lane == 0 && i < k) qv[i] = sum;
      }
    }
  }
  __syncthreads();
  // second loop: oldest -> newest    be_i = ro_i ( H ( -y_i.g - q_i ) + sum_{j older than i} (al_j - be_j) y_i.s_j ),  y_i.s_j = s_j.y_i
  if (wave == 0) {
    float acc0 = 0.f, acc1 = 0.f;
    const float c0 = r0 < k ? h_diag * (-mgy[r0] - qv[r0]) : 0.f, c1 = r1 < k ? h_diag * (-mgy[r1] - qv[r1]) : 0.f;
    const float al0 = r0 < k ? al[r0] : 0.f, al1 = r1 < k ? al[r1] : 0.f;
    for (int j = 0; j < k; ++j) {
      const float mine = j < 64 ? al0 - ro0 * (c0 + acc0) : al1 - ro1 * (c1 + acc1);     // al_j - be_j by its owner
      const float bj = lb_bcast(mine, j & 63);
      if (lane == (j & 63)) ab[j] = bj;
      if (r0 > j && r0 < k) acc0 = fmaf(bj, Gsy[lb_tri(j, r0, k)], acc0);
      if (r1 > j && r1 < k) acc1 = fmaf(bj, Gsy[lb_tri(j, r1, k)], acc1);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int i = lane; i < k; i += 64) {
      const int pi = s_order[i];
      a.coef[pi] = ab[i];                       // s_i
      a.coef[h + pi] = -h_diag * al[i];       // y_i
    }
  }
}

}  // namespace ha

namespace ha {
constexpr size_t LB_COEFF_LDS = (size_t)((LB_MAXH * (LB_MAXH + 1)) / 2 + 5 * LB_MAXH) * sizeof(float) + LB_MAXH * sizeof(int);

// ---- one pass over the history for the three products the update needs ----------------------------------------------------------
// P[r][k] = M[r] . V_k for r < rows, V = (M[i0], M[i1], M[i2]) (the new s, the new y and the current gradient row).  torch issues a
// matrix-vector product per right-hand side (three reads of the 60-76 MB history); here one read, fixed summation order (column
// chunks of 512 in chunk order: the replicated multi-GPU optimiser needs bit-identical directions on every rank -- no atomics).
constexpr int GR_CW = 512;
__global__ __launch_bounds__(256) void lbfgs_gram_partial_kernel(const float* __restrict__ M, int n, int rows, int i0, int i1, int i2,
                                                                 float* __restrict__ part) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c0 = blockIdx.x * GR_CW;
  // grid.y splits the rows: one block per (column chunk, row group) -- with all 2 h rows in one block only ~570 waves were in flight
  // for a 60-76 MB read (1.3 TB/s)
  const int rpb = (rows + gridDim.y - 1) / gridDim.y, rbeg = blockIdx.y * rpb, rend = (rbeg + rpb) < rows ? (rbeg + rpb) : rows;
  float v[3][8];
  const int idx[3] = {i0, i1, i2};
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c0 + (e >> 2) * 256 + lane * 4 + (e & 3);
      v[k][e] = c < n ? M[(size_t)idx[k] * n + c] : 0.f;
    }
  // four rows (8 x 16-byte... as 4-byte lane-contiguous loads: 32 loads) in flight per trip: the loop is a latency chain otherwise
  for (int r0 = rbeg + wave; r0 < rend; r0 += 16) {
    float m[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + 4 * u < rend ? r0 + 4 * u : r0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = c0 + (e >> 2) * 256 + lane * 4 + (e & 3);
        m[u][e] = c < n ? M[(size_t)r * n + c] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + 4 * u;
      float p[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) p[k] = fmaf(m[u][e], v[k][e], p[k]);
#pragma unroll
      for (int k = 0; k < 3; ++k) p[k] = lb_wsum(p[k]);
      if (lane == 0 && r < rend) {
        float* dst = part + ((size_t)blockIdx.x * rows + r) * 3;
        dst[0] = p[0]; dst[1] = p[1]; dst[2] = p[2];
      }
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
  // one block streams both vectors (n <= a few 100 k): 16-byte loads, eight pairs in flight per trip, scalar tail
  typedef float v4 __attribute__((ext_vector_type(4)));
  const bool vec = ((((uintptr_t)a) | ((uintptr_t)b)) & 15) == 0;
  const int n4 = vec ? n >> 2 : 0;
  const v4* a4 = reinterpret_cast<const v4*>(a);
  const v4* b4 = reinterpret_cast<const v4*>(b);
  for (int i0 = tid; i0 < n4; i0 += 8 * 1024) {
    v4 xa[8], xb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 1024;
      xa[u] = i < n4 ? a4[i] : v4{0.f, 0.f, 0.f, 0.f};
      xb[u] = i < n4 ? b4[i] : v4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        d = fmaf(xa[u][e], xb[u][e], d);
        m = fmaxf(m, fabsf(xa[u][e]));
        su += fabsf(xa[u][e]);
      }
  }
  for (int i = 4 * n4 + tid; i < n; i += 1024) {
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
  HA_LAUNCH(lbfgs_gram_partial_kernel, dim3(nb, 8), dim3(256), 0, (hipStream_t)stream, M, n, rows, i0, i1, i2, part);
  HA_LAUNCH_CHECK();
  HA_LAUNCH(lbfgs_gram_reduce_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, part, nb, rows, P);
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
  HA_LAUNCH(lbfgs_scalars_kernel, dim3(1), dim3(1024), 48 * sizeof(float), (hipStream_t)stream, n, a, b, extra, out);
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
  HA_LAUNCH(lbfgs_coeffs_kernel, dim3(1), dim3(256), LB_COEFF_LDS, (hipStream_t)stream, a);
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
  HA_LAUNCH(lbfgs_coeffs_kernel, dim3(1), dim3(256), LB_COEFF_LDS, (hipStream_t)stream, a);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

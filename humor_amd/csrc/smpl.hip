// SMPL(+H) body model on gfx950: model packing + kernels + C-ABI entry points.
//
// Replaces the un-vendored smplx==0.1.28 `lbs` / `SMPLH.forward` / `VertexJointSelector` that
// humor/body_model/body_model.py:61-68,78-91 delegates to (algorithm restated in oracle/lbs_restated.py).
//
// Kernels
//   smpl_frame_fwd_kernel   one wavefront per frame: lane j = joint j (Rodrigues, rest joints from the
//                           pre-contracted regressor, kinematic chain by tree level through LDS), then lanes =
//                           vertices of 64-wide chunks of the chosen vertex subset (blend-shapes as one
//                           coefficient-vector x blend-matrix product, 4-sparse skinning with A in LDS).
//                           This is the kernel the fitting closure runs (<= 64 vertices are consumed by the
//                           losses), and the producer of A / the coefficient matrix for the dense path.
//   smpl_frame_bwd_kernel   same decomposition in reverse; recomputes forward intermediates, reverse level
//                           scan with parent-side (deterministic) accumulation over a children CSR.
//   pose_blend_mfma_kernel  dense path: v_posed[N, V*3] = C[N,Kc] x Pd[Kc, V*3] on v_mfma_f32_32x32x2_f32
//                           (exact fp32), 64 frames x 32 vertices (x,y,z as three column tiles) per wave.
//   lbs_skin_kernel         dense path: the HBM-streaming skinning kernel, 16-byte coalesced loads/stores of
//                           the [N,V,3] arrays through an LDS transpose, A of the touched frames in LDS.
#include <stdarg.h>
#include <string.h>

#include <vector>

#include "smpl_model.h"

namespace ha {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace ha

namespace ha {
int g_skin_variant = -1;  // -1 = auto: direct kernel, NT stores when the output fits the 256 MiB Infinity Cache   // 0: LDS-staged window, 1: + hoisted weight loads, 2: direct 48-byte per-thread path, 3: frame-pair windows; +4: non-temporal; +8/+16: 2/4 vertex groups per thread (variant 2)
}
extern "C" int ha_tune_set(const char* key, int value) {
  HA_REQUIRE(key, "ha_tune_set: null key");
  if (strcmp(key, "skin_variant") == 0) { ha::g_skin_variant = value; return HA_OK; }
  ha::set_error("ha_tune_set: unknown key '%s'", key);
  return HA_ERR_INVALID_ARG;
}
extern "C" const char* ha_last_error(void) { return ha::g_err; }
extern "C" int ha_abi_version(void) { return 1; }
extern "C" int ha_device_arch(int device, char* buf, int buflen) {
  HA_REQUIRE(buf && buflen > 0, "ha_device_arch: null buffer");
  hipDeviceProp_t prop;
  HA_CHECK_HIP(hipGetDeviceProperties(&prop, device));
  snprintf(buf, buflen, "%s", prop.gcnArchName);
  return HA_OK;
}

namespace ha {

// ---------------------------------------------------------------------------------------------------
// upload helpers
// ---------------------------------------------------------------------------------------------------
template <typename T>
static int upload(T** dst, const std::vector<T>& src) {
  *dst = nullptr;
  if (src.empty()) return HA_OK;
  HA_CHECK_HIP(hipMalloc((void**)dst, src.size() * sizeof(T)));
  HA_CHECK_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return HA_OK;
}

static void free_set(VertexSet& s) {
  if (s.Pd_v) (void)hipFree(s.Pd_v);
  if (s.w) (void)hipFree(s.w);
  if (s.idx) (void)hipFree(s.idx);
  if (s.ids) (void)hipFree(s.ids);
  s = VertexSet();
}

static int build_set(ha_smpl_model* m, int slot, const int32_t* ids, int n) {
  VertexSet& s = m->sets[slot];
  free_set(s);
  s.n = n;
  s.nchunks = ceil_div(n, kChunk);
  s.npad = s.nchunks * kChunk;
  const int K = m->Kfull, V = m->V, nnz = m->nnz;
  std::vector<float> pd((size_t)s.nchunks * K * 3 * kChunk, 0.0f);
  std::vector<float> w((size_t)s.nchunks * nnz * kChunk, 0.0f);
  std::vector<int32_t> ix((size_t)s.nchunks * nnz * kChunk, 0);
  for (int i = 0; i < n; ++i) {
    const int v = ids ? ids[i] : i;
    const int ch = i / kChunk, ln = i % kChunk;
    for (int k = 0; k < K; ++k)
      for (int c = 0; c < 3; ++c)
        pd[(((size_t)ch * K + k) * 3 + c) * kChunk + ln] = m->h_Pd[((size_t)k * V + v) * 3 + c];
    for (int q = 0; q < nnz; ++q) {
      w[((size_t)ch * nnz + q) * kChunk + ln] = m->h_w[(size_t)v * nnz + q];
      ix[((size_t)ch * nnz + q) * kChunk + ln] = m->h_idx[(size_t)v * nnz + q];
    }
  }
  int rc;
  if ((rc = upload(&s.Pd_v, pd)) != HA_OK) return rc;
  if ((rc = upload(&s.w, w)) != HA_OK) return rc;
  if ((rc = upload(&s.idx, ix)) != HA_OK) return rc;
  if (ids) {
    std::vector<int32_t> idv(ids, ids + n);
    if ((rc = upload(&s.ids, idv)) != HA_OK) return rc;
  }
  return HA_OK;
}

}  // namespace ha

using namespace ha;

extern "C" int ha_smpl_model_create(ha_smpl_model** out, int device, int V, int J, int NB, const float* v_template,
                                    const float* shapedirs, const float* posedirs, const float* J_regressor,
                                    const float* weights, const int32_t* parents) {
  HA_REQUIRE(out && v_template && shapedirs && posedirs && J_regressor && weights && parents,
             "ha_smpl_model_create: null argument");
  HA_REQUIRE(V > 0 && J >= 1 && J <= kMaxJoints && NB >= 0 && NB <= 64,
             "ha_smpl_model_create: unsupported sizes V=%d J=%d NB=%d (need J<=64, NB<=64)", V, J, NB);
  for (int j = 1; j < J; ++j)
    HA_REQUIRE(parents[j] >= 0 && parents[j] < j, "ha_smpl_model_create: parents[%d]=%d must satisfy 0<=p<j", j, parents[j]);
  DeviceGuard guard(device);
  HA_REQUIRE(guard.ok, "ha_smpl_model_create: cannot select device %d", device);

  ha_smpl_model* m = new ha_smpl_model();
  m->device = device;
  m->V = V; m->J = J; m->NB = NB; m->P = (J - 1) * 9;
  m->Kfull = NB + 1 + m->P;
  m->Kfull_pad = m->Kfull + (m->Kfull & 1);
  m->Vpad = ceil_div(V, 64) * 64;
  if (m->Kfull_pad > 512) {
    set_error("ha_smpl_model_create: blend basis %d > 512 unsupported", m->Kfull_pad);
    delete m;
    return HA_ERR_UNSUPPORTED;
  }

  // tree tables
  std::vector<int32_t> par(J), dep(J, 0), cstart(J + 1, 0), cidx(J > 1 ? J - 1 : 0);
  par[0] = -1;
  for (int j = 1; j < J; ++j) { par[j] = parents[j]; dep[j] = dep[par[j]] + 1; m->depth = dep[j] > m->depth ? dep[j] : m->depth; }
  for (int j = 1; j < J; ++j) cstart[par[j] + 1]++;
  for (int j = 0; j < J; ++j) cstart[j + 1] += cstart[j];
  {
    std::vector<int32_t> fill(cstart.begin(), cstart.end() - 1);
    for (int j = 1; j < J; ++j) cidx[fill[par[j]]++] = j;
  }
  if (cidx.empty()) cidx.push_back(0);

  // pre-contracted joint regressor (double accumulation, rounded once)
  std::vector<float> Jt((size_t)J * 3), Js((size_t)J * 3 * (NB > 0 ? NB : 1), 0.0f);
  for (int j = 0; j < J; ++j)
    for (int c = 0; c < 3; ++c) {
      double acc = 0.0;
      for (int v = 0; v < V; ++v) acc += (double)J_regressor[(size_t)j * V + v] * (double)v_template[(size_t)v * 3 + c];
      Jt[j * 3 + c] = (float)acc;
      for (int l = 0; l < NB; ++l) {
        double a2 = 0.0;
        for (int v = 0; v < V; ++v)
          a2 += (double)J_regressor[(size_t)j * V + v] * (double)shapedirs[((size_t)v * 3 + c) * NB + l];
        Js[((size_t)j * 3 + c) * NB + l] = (float)a2;
      }
    }

  // blend matrix rows in coefficient order: shapedirs | template | posedirs
  const int K = m->Kfull, P = m->P;
  m->h_Pd = new float[(size_t)K * V * 3];
  for (int v = 0; v < V; ++v)
    for (int c = 0; c < 3; ++c) {
      for (int l = 0; l < NB; ++l) m->h_Pd[((size_t)l * V + v) * 3 + c] = shapedirs[((size_t)v * 3 + c) * NB + l];
      m->h_Pd[((size_t)NB * V + v) * 3 + c] = v_template[(size_t)v * 3 + c];
      for (int k = 0; k < P; ++k) m->h_Pd[((size_t)(NB + 1 + k) * V + v) * 3 + c] = posedirs[((size_t)v * 3 + c) * P + k];
    }

  // sparse skinning weights
  int nnz = 1;
  for (int v = 0; v < V; ++v) {
    int cnt = 0;
    for (int j = 0; j < J; ++j) cnt += weights[(size_t)v * J + j] != 0.0f;
    nnz = cnt > nnz ? cnt : nnz;
  }
  m->nnz = nnz;
  m->h_w = new float[(size_t)V * nnz]();
  m->h_idx = new int32_t[(size_t)V * nnz]();
  for (int v = 0; v < V; ++v) {
    int q = 0;
    for (int j = 0; j < J; ++j) {
      const float wv = weights[(size_t)v * J + j];
      if (wv != 0.0f) { m->h_w[(size_t)v * nnz + q] = wv; m->h_idx[(size_t)v * nnz + q] = j; ++q; }
    }
  }

  int rc = HA_OK;
  auto fail = [&](int code) { ha_smpl_model_destroy(m); return code; };
  if ((rc = upload(&m->Jt, Jt)) != HA_OK) return fail(rc);
  if ((rc = upload(&m->Js, Js)) != HA_OK) return fail(rc);
  if ((rc = upload(&m->parents, par)) != HA_OK) return fail(rc);
  if ((rc = upload(&m->jdepth, dep)) != HA_OK) return fail(rc);
  if ((rc = upload(&m->child_start, cstart)) != HA_OK) return fail(rc);
  if ((rc = upload(&m->child_idx, cidx)) != HA_OK) return fail(rc);
  if ((rc = build_set(m, 0, nullptr, V)) != HA_OK) return fail(rc);

  // MFMA B-operand layout: [Vpad/32][Kfull_pad/2][3][64], lane l <-> (k = 2*kp + (l>>5), vertex = vt*32 + (l&31))
  {
    const int nvt = m->Vpad / 32, KP = m->Kfull_pad / 2;
    std::vector<float> pm((size_t)nvt * KP * 3 * 64, 0.0f);
    for (int vt = 0; vt < nvt; ++vt)
      for (int kp = 0; kp < KP; ++kp)
        for (int c = 0; c < 3; ++c)
          for (int l = 0; l < 64; ++l) {
            const int k = 2 * kp + (l >> 5), v = vt * 32 + (l & 31);
            if (k < K && v < V) pm[(((size_t)vt * KP + kp) * 3 + c) * 64 + l] = m->h_Pd[((size_t)k * V + v) * 3 + c];
          }
    if ((rc = upload(&m->Pd_m, pm)) != HA_OK) return fail(rc);
  }
  if (nnz <= 4) {
    std::vector<float4> w4(V);
    std::vector<uint32_t> i4(V);
    for (int v = 0; v < V; ++v) {
      float ww[4] = {0, 0, 0, 0};
      uint32_t packed = 0;
      for (int q = 0; q < nnz; ++q) { ww[q] = m->h_w[(size_t)v * nnz + q]; packed |= ((uint32_t)m->h_idx[(size_t)v * nnz + q] & 0xff) << (8 * q); }
      w4[v] = make_float4(ww[0], ww[1], ww[2], ww[3]);
      i4[v] = packed;
    }
    if ((rc = upload(&m->w4, w4)) != HA_OK) return fail(rc);
    if ((rc = upload(&m->idx4, i4)) != HA_OK) return fail(rc);
  }
  *out = m;
  return HA_OK;
}

extern "C" int ha_smpl_model_destroy(ha_smpl_model* m) {
  if (!m) return HA_OK;
  DeviceGuard guard(m->device);
  for (int s = 0; s < kMaxSubsets; ++s) free_set(m->sets[s]);
  void* ptrs[] = {m->Jt, m->Js, m->parents, m->jdepth, m->child_start, m->child_idx, m->Pd_m, m->w4, m->idx4};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  delete[] m->h_Pd;
  delete[] m->h_w;
  delete[] m->h_idx;
  delete m;
  return HA_OK;
}

extern "C" int ha_smpl_model_info(const ha_smpl_model* m, int what, int* value) {
  HA_REQUIRE(m && value, "ha_smpl_model_info: null argument");
  switch (what) {
    case 0: *value = m->V; break;
    case 1: *value = m->J; break;
    case 2: *value = m->NB; break;
    case 3: *value = m->nnz; break;
    case 4: *value = m->depth; break;
    case 5: *value = m->Vpad; break;
    case 6: { int c = 0; for (int s = 0; s < kMaxSubsets; ++s) c += m->sets[s].n > 0; *value = c; break; }
    case 7: *value = m->P; break;
    default: set_error("ha_smpl_model_info: unknown query %d", what); return HA_ERR_INVALID_ARG;
  }
  return HA_OK;
}

extern "C" int ha_smpl_model_define_subset(ha_smpl_model* m, int slot, const int32_t* ids, int n) {
  HA_REQUIRE(m && ids, "ha_smpl_model_define_subset: null argument");
  HA_REQUIRE(slot >= 1 && slot < kMaxSubsets, "ha_smpl_model_define_subset: slot %d out of range 1..%d", slot, kMaxSubsets - 1);
  HA_REQUIRE(n >= 1, "ha_smpl_model_define_subset: empty subset");
  for (int i = 0; i < n; ++i) HA_REQUIRE(ids[i] >= 0 && ids[i] < m->V, "ha_smpl_model_define_subset: id %d out of range", ids[i]);
  DeviceGuard guard(m->device);
  return build_set(m, slot, ids, n);
}

// ===================================================================================================
// wave-per-frame kernels
// ===================================================================================================
namespace ha {

struct FrameParams {
  // model
  const float* Jt; const float* Js; const int32_t* parents; const int32_t* jdepth;
  const int32_t* child_start; const int32_t* child_idx;
  int J, NB, Kfull, Kfull_pad, kf4, depth;   // kf4: Kfull_pad rounded to 4 floats (LDS stride)
  // vertex set
  const float* Pd_v; const float* w; const int32_t* idx;
  int nverts, nchunks, nnz;
  // problem
  int N, n_active, Kc;
  const float* pose; const float* betas; const float* transl;
  // forward outputs
  float* verts; float* joints; float* A_out; float* coeffT; int Npad;
  // backward io
  const float* g_verts; const float* g_joints;
  float* g_pose; float* g_betas; float* g_transl;
};

constexpr int FW = 4;  // waves (= frames) per block

// Per-joint forward state kept in registers by lane j.
struct JointState {
  float R[9];    // local rotation
  float Jr[3];   // rest joint (shape-dependent)
  float t[3];    // translation relative to parent rest joint
  float G[12];   // world transform: R (9) | t (3)
  int parent, depth;
};

// Shared prologue of forward and backward: Rodrigues, rest joints, coefficient vector, chain.
// LDS: coeff[Kfull_pad], Gs[J*12] (world transforms; later overwritten with A by the caller).
__device__ __forceinline__ void joint_forward(const FrameParams& p, int f, int lane, float* coeff, float* Gs, JointState& s) {
  const int j = lane;
  const bool isj = j < p.J;
#pragma unroll
  for (int i = 0; i < 9; ++i) s.R[i] = (i % 4 == 0) ? 1.0f : 0.0f;
  s.Jr[0] = s.Jr[1] = s.Jr[2] = 0.0f;
  s.parent = -1;
  s.depth = -1;
  if (isj) {
    s.parent = p.parents[j];
    s.depth = p.jdepth[j];
    if (j < p.n_active) {
      const float* r = p.pose + ((size_t)f * p.J + j) * 3;
      const float rr[3] = {r[0], r[1], r[2]};
      rodrigues(rr, s.R);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float acc = p.Jt[j * 3 + c];
      for (int l = 0; l < p.NB; ++l) acc = fmaf(p.betas[(size_t)f * p.NB + l], p.Js[(j * 3 + c) * p.NB + l], acc);
      s.Jr[c] = acc;
    }
  }
  // coefficient vector: betas | 1 | pose feature (prefix of length Kc is what the kernels iterate over)
  for (int i = lane; i < p.NB; i += 64) coeff[i] = p.betas[(size_t)f * p.NB + i];
  if (lane == 0) {
    coeff[p.NB] = 1.0f;
    if (p.Kc & 1) coeff[p.Kc] = 0.0f;   // pad entry read by the MFMA path's last k-pair
  }
  if (isj && j >= 1 && j < p.n_active) {
#pragma unroll
    for (int i = 0; i < 9; ++i) coeff[p.NB + 1 + (j - 1) * 9 + i] = s.R[i] - ((i % 4 == 0) ? 1.0f : 0.0f);
  }
  // relative translation
  const int psrc = s.parent < 0 ? 0 : s.parent;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float pj = __shfl(s.Jr[c], psrc);
    s.t[c] = s.parent < 0 ? s.Jr[c] : s.Jr[c] - pj;
  }
  // chain by level
  if (isj && s.parent < 0) {
#pragma unroll
    for (int i = 0; i < 9; ++i) s.G[i] = s.R[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) s.G[9 + c] = s.t[c];
#pragma unroll
    for (int i = 0; i < 12; ++i) Gs[j * 12 + i] = s.G[i];
  }
  __syncthreads();
  for (int lvl = 1; lvl <= p.depth; ++lvl) {
    if (isj && s.depth == lvl) {
      float Gp[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) Gp[i] = Gs[s.parent * 12 + i];
      mat3_mul(Gp, s.R, s.G);
      float tt[3];
      mat3_vec(Gp, s.t, tt);
#pragma unroll
      for (int c = 0; c < 3; ++c) s.G[9 + c] = tt[c] + Gp[9 + c];
#pragma unroll
      for (int i = 0; i < 12; ++i) Gs[j * 12 + i] = s.G[i];
    }
    __syncthreads();
  }
}

// blend-shape accumulation for the lane's vertex of `chunk`: v_posed = sum_k coeff[k] * Pd[k]
__device__ __forceinline__ void blend_vertex(const FrameParams& p, int chunk, int lane, const float* coeff, float vp[3]) {
  const float* pd = p.Pd_v + (size_t)chunk * p.Kfull * 192 + lane;
  float ax = 0.f, ay = 0.f, az = 0.f;
  int k = 0;
  for (; k + 4 <= p.Kc; k += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float c = coeff[k + u];
      const float* q = pd + (size_t)(k + u) * 192;
      ax = fmaf(c, q[0], ax);
      ay = fmaf(c, q[64], ay);
      az = fmaf(c, q[128], az);
    }
  }
  for (; k < p.Kc; ++k) {
    const float c = coeff[k];
    const float* q = pd + (size_t)k * 192;
    ax = fmaf(c, q[0], ax);
    ay = fmaf(c, q[64], ay);
    az = fmaf(c, q[128], az);
  }
  vp[0] = ax; vp[1] = ay; vp[2] = az;
}

__device__ __forceinline__ void blend_transform(const FrameParams& p, int chunk, int lane, const float* As, float T[12]) {
#pragma unroll
  for (int i = 0; i < 12; ++i) T[i] = 0.f;
  for (int q = 0; q < p.nnz; ++q) {
    const float wq = p.w[((size_t)chunk * p.nnz + q) * 64 + lane];
    const int jq = p.idx[((size_t)chunk * p.nnz + q) * 64 + lane];
    const float* a = As + jq * 12;
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = fmaf(wq, a[i], T[i]);
  }
}

__global__ __launch_bounds__(FW * 64) void smpl_frame_fwd_kernel(FrameParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int per_wave = p.kf4 + p.J * 12;
  float* coeff = smem + wave * per_wave;
  float* Gs = coeff + p.kf4;
  int f = blockIdx.x * FW + wave;
  const bool valid = f < p.N;
  if (!valid) f = p.N - 1;

  JointState s;
  joint_forward(p, f, lane, coeff, Gs, s);

  float tl[3] = {0.f, 0.f, 0.f};
  if (p.transl) { tl[0] = p.transl[(size_t)f * 3]; tl[1] = p.transl[(size_t)f * 3 + 1]; tl[2] = p.transl[(size_t)f * 3 + 2]; }

  const bool isj = lane < p.J;
  // A = [G.R | G.t - G.R * Jr]; all chain reads of Gs are complete (barrier at the end of joint_forward)
  float A[12];
  if (isj) {
    float gj[3];
    mat3_vec(s.G, s.Jr, gj);
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i] = s.G[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) A[9 + c] = s.G[9 + c] - gj[c];
#pragma unroll
    for (int i = 0; i < 12; ++i) Gs[lane * 12 + i] = A[i];
    if (valid) {
      if (p.joints) {
#pragma unroll
        for (int c = 0; c < 3; ++c) p.joints[((size_t)f * p.J + lane) * 3 + c] = s.G[9 + c] + tl[c];
      }
      if (p.A_out) {
#pragma unroll
        for (int i = 0; i < 12; ++i) p.A_out[((size_t)f * p.J + lane) * 12 + i] = A[i];
      }
    }
  }
  __syncthreads();
  if (p.coeffT && valid) {
    const int kc_pad = p.Kc + (p.Kc & 1);
    for (int k = lane; k < kc_pad; k += 64) p.coeffT[(size_t)k * p.Npad + f] = coeff[k];
  }
  if (p.verts) {
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
      float vp[3], T[12];
      blend_vertex(p, chunk, lane, coeff, vp);
      blend_transform(p, chunk, lane, Gs, T);
      const int v = chunk * 64 + lane;
      if (valid && v < p.nverts) {
        float o[3];
        mat3_vec(T, vp, o);
        float* dst = p.verts + ((size_t)f * p.nverts + v) * 3;
        dst[0] = o[0] + T[9] + tl[0];
        dst[1] = o[1] + T[10] + tl[1];
        dst[2] = o[2] + T[11] + tl[2];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// backward
// LDS per wave: coeff[Kfull_pad] | Gs[J*12] | As[J*12] | gA[J*12] | msg[J*16] | gvp[192] | gco[Kfull_pad]
// ---------------------------------------------------------------------------------------------------
constexpr int kMaxKM = 8;  // Kfull_pad <= 512 -> at most 8 coefficient gradients per lane

__global__ __launch_bounds__(FW * 64) void smpl_frame_bwd_kernel(FrameParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int J = p.J;
  const int per_wave = 2 * p.kf4 + J * (12 * 3 + 16) + 192;
  float* coeff = smem + wave * per_wave;
  float* Gs = coeff + p.kf4;
  float* As = Gs + J * 12;
  float* gA = As + J * 12;
  float* msg = gA + J * 12;
  float* gvp = msg + J * 16;
  float* gco = gvp + 192;
  int f = blockIdx.x * FW + wave;
  const bool valid = f < p.N;
  if (!valid) f = p.N - 1;
  const bool isj = lane < J;

  JointState s;
  joint_forward(p, f, lane, coeff, Gs, s);
  if (isj) {
    float gj[3];
    mat3_vec(s.G, s.Jr, gj);
#pragma unroll
    for (int i = 0; i < 9; ++i) As[lane * 12 + i] = s.G[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) As[lane * 12 + 9 + c] = s.G[9 + c] - gj[c];
#pragma unroll
    for (int i = 0; i < 12; ++i) gA[lane * 12 + i] = 0.f;
  }
  __syncthreads();

  // ---- vertex phase ------------------------------------------------------------------------------
  float gco_reg[kMaxKM];
#pragma unroll
  for (int m = 0; m < kMaxKM; ++m) gco_reg[m] = 0.f;
  float gtl[3] = {0.f, 0.f, 0.f};   // transl gradient partial (vertices of this lane, then joints)
  if (p.g_verts) {
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
      float vp[3], T[12];
      blend_vertex(p, chunk, lane, coeff, vp);
      blend_transform(p, chunk, lane, As, T);
      const int v = chunk * 64 + lane;
      float g[3] = {0.f, 0.f, 0.f};
      if (v < p.nverts) {
        const float* src = p.g_verts + ((size_t)f * p.nverts + v) * 3;
        g[0] = src[0]; g[1] = src[1]; g[2] = src[2];
      }
      gtl[0] += g[0]; gtl[1] += g[1]; gtl[2] += g[2];
      float gv[3];
      mat3_tvec(T, g, gv);          // dL/dv_posed = T_R^T g
      gvp[lane] = gv[0]; gvp[64 + lane] = gv[1]; gvp[128 + lane] = gv[2];
      // dL/dA_j += w * [g (x) v_posed | g]
      for (int q = 0; q < p.nnz; ++q) {
        const float wq = p.w[((size_t)chunk * p.nnz + q) * 64 + lane];
        const int jq = p.idx[((size_t)chunk * p.nnz + q) * 64 + lane];
        if (wq != 0.f) {
          float* dst = gA + jq * 12;
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const float wg = wq * g[a];
            atomicAdd(dst + a * 3 + 0, wg * vp[0]);
            atomicAdd(dst + a * 3 + 1, wg * vp[1]);
            atomicAdd(dst + a * 3 + 2, wg * vp[2]);
            atomicAdd(dst + 9 + a, wg);
          }
        }
      }
      __syncthreads();
      // dL/dcoeff[k] += sum_{v,c} gvp[c][v] * Pd[k][c][v]   (lane = k, each lane streams its own 768-B row)
#pragma unroll
      for (int m = 0; m < kMaxKM; ++m) {
        const int k = lane + 64 * m;
        if (k < p.Kc) {
          const float4* row = reinterpret_cast<const float4*>(p.Pd_v + ((size_t)chunk * p.Kfull + k) * 192);
          const float4* gq = reinterpret_cast<const float4*>(gvp);
          float acc = 0.f;
#pragma unroll 4
          for (int i = 0; i < 48; ++i) {
            const float4 a = row[i];
            const float4 b = gq[i];
            acc = fmaf(a.x, b.x, acc);
            acc = fmaf(a.y, b.y, acc);
            acc = fmaf(a.z, b.z, acc);
            acc = fmaf(a.w, b.w, acc);
          }
          gco_reg[m] += acc;
        }
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int m = 0; m < kMaxKM; ++m) {
    const int k = lane + 64 * m;
    if (k < p.Kfull_pad) gco[k] = (k < p.Kc) ? gco_reg[m] : 0.f;
  }
  __syncthreads();

  // ---- chain backward ----------------------------------------------------------------------------
  // gG = dL/dG_j (world transform): from A_j = [G.R | G.t - G.R Jr] and posed joint = G.t
  float gG[12], gJr[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 12; ++i) gG[i] = 0.f;
  if (isj) {
    float gAj[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) gAj[i] = gA[lane * 12 + i];
    float gjt[3] = {0.f, 0.f, 0.f};
    if (p.g_joints) {
#pragma unroll
      for (int c = 0; c < 3; ++c) gjt[c] = p.g_joints[((size_t)f * J + lane) * 3 + c];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int b = 0; b < 3; ++b) gG[a * 3 + b] = gAj[a * 3 + b] - gAj[9 + a] * s.Jr[b];
      gG[9 + a] = gAj[9 + a] + gjt[a];
      gtl[a] += gjt[a];
    }
    float tmp[3];
    const float gat[3] = {gAj[9], gAj[10], gAj[11]};
    mat3_tvec(s.G, gat, tmp);
    gJr[0] = -tmp[0]; gJr[1] = -tmp[1]; gJr[2] = -tmp[2];
  }
  float gR[9], gt[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 9; ++i) gR[i] = 0.f;
  for (int lvl = p.depth; lvl >= 1; --lvl) {
    if (isj && s.depth == lvl) {
      // own gG is final here: emit the message to the parent and the local gradients
      float Gp[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) Gp[i] = Gs[s.parent * 12 + i];
      float m9[9];
      mat3_mult(gG, s.R, m9);                       // gG.R * R^T
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) m9[a * 3 + b] += gG[9 + a] * s.t[b];   // + gG.t (x) t
      mat3_tmul(Gp, gG, gR);                        // gR = Gp.R^T gG.R
      const float ggt[3] = {gG[9], gG[10], gG[11]};
      mat3_tvec(Gp, ggt, gt);                       // gt = Gp.R^T gG.t
#pragma unroll
      for (int i = 0; i < 9; ++i) msg[lane * 16 + i] = m9[i];
#pragma unroll
      for (int c = 0; c < 3; ++c) { msg[lane * 16 + 9 + c] = gG[9 + c]; msg[lane * 16 + 12 + c] = gt[c]; }
    }
    __syncthreads();
    if (isj && s.depth == lvl - 1) {
      const int c0 = p.child_start[lane], c1 = p.child_start[lane + 1];
      for (int ci = c0; ci < c1; ++ci) {
        const int ch = p.child_idx[ci];
#pragma unroll
        for (int i = 0; i < 12; ++i) gG[i] += msg[ch * 16 + i];
#pragma unroll
        for (int c = 0; c < 3; ++c) gJr[c] -= msg[ch * 16 + 12 + c];
      }
    }
    __syncthreads();
  }
  if (isj && s.parent < 0) {
#pragma unroll
    for (int i = 0; i < 9; ++i) gR[i] = gG[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) gt[c] = gG[9 + c];
  }
  if (isj) {
#pragma unroll
    for (int c = 0; c < 3; ++c) gJr[c] += gt[c];
  }

  // ---- pose gradient -------------------------------------------------------------------------------
  if (isj && valid && p.g_pose) {
    float gr[3] = {0.f, 0.f, 0.f};
    if (lane < p.n_active) {
      if (lane >= 1) {
#pragma unroll
        for (int i = 0; i < 9; ++i) gR[i] += gco[p.NB + 1 + (lane - 1) * 9 + i];
      }
      const float* r = p.pose + ((size_t)f * J + lane) * 3;
      const float rr[3] = {r[0], r[1], r[2]};
      rodrigues_bwd(rr, gR, gr);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) p.g_pose[((size_t)f * J + lane) * 3 + c] = gr[c];
  }
  // ---- betas gradient: coefficient part + rest-joint part ------------------------------------------
  if (p.g_betas) {
    if (isj) {
#pragma unroll
      for (int c = 0; c < 3; ++c) msg[lane * 16 + c] = gJr[c];   // all message reads are behind the last barrier
    }
    __syncthreads();
    for (int l = lane; l < p.NB; l += 64) {
      float acc = gco[l];
      for (int j = 0; j < J; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc = fmaf(msg[j * 16 + c], p.Js[(j * 3 + c) * p.NB + l], acc);
      if (valid) p.g_betas[(size_t)f * p.NB + l] = acc;
    }
  }
  // ---- transl gradient: wave reduction -------------------------------------------------------------
  if (p.g_transl) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float vsum = gtl[c];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) vsum += __shfl_xor(vsum, off);
      if (lane == 0 && valid) p.g_transl[(size_t)f * 3 + c] = vsum;
    }
  }
}

// ===================================================================================================
// dense path: fp32 MFMA pose-blend GEMM
// ===================================================================================================
typedef float f32x16 __attribute__((ext_vector_type(16)));

// grid: 1-D, block = 4 waves = 4 vertex tiles (32 vertices each) x 64 frames.
// Blocks that share a vertex-tile group (the B panel, ~300 KB) are mapped to the same XCD (b % 8).
__global__ __launch_bounds__(256) void pose_blend_mfma_kernel(const float* __restrict__ coeffT, int Npad, int KP, int KPfull,
                                                              const float* __restrict__ Pd_m, float* __restrict__ v_posed,
                                                              int N, int V, int n_vt, int n_ft) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x;
  const int vtg = (b & 7) + 8 * (b / (8 * n_ft));
  const int ft = (b >> 3) % n_ft;
  const int vt = vtg * 4 + wave;
  if (vt >= n_vt) return;
  const int f0 = ft * 64;
  const float* a_ptr = coeffT + (size_t)(lane >> 5) * Npad + f0 + (lane & 31);
  const float* b_ptr = Pd_m + (size_t)vt * KPfull * 192 + lane;
  f32x16 acc[2][3];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[r][c][i] = 0.f;
#pragma unroll 2
  for (int kp = 0; kp < KP; ++kp) {
    const float a0 = a_ptr[(size_t)2 * kp * Npad];
    const float a1 = a_ptr[(size_t)2 * kp * Npad + 32];
    const float b0 = b_ptr[(size_t)kp * 192];
    const float b1 = b_ptr[(size_t)kp * 192 + 64];
    const float b2 = b_ptr[(size_t)kp * 192 + 128];
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
    acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b2, acc[0][2], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b2, acc[1][2], 0, 0, 0);
  }
  const int v = vt * 32 + (lane & 31);
  if (v >= V) return;
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int f = f0 + r * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
      if (f < N) {
        float* dst = v_posed + ((size_t)f * V + v) * 3;
        dst[0] = acc[r][0][i];
        dst[1] = acc[r][1][i];
        dst[2] = acc[r][2][i];
      }
    }
}

// ===================================================================================================
// dense path: streaming linear-blend skinning (HBM-bound)
// Flat partition of the [N*V] vertex index space: block b owns global vertices [1024 b, 1024 b + 1024), i.e. a
// 12 KiB, 16-byte aligned window of both arrays that spans at most two frames (V >= 1024).
// ===================================================================================================
constexpr int kSkinVerts = 1024;

__global__ __launch_bounds__(256) void lbs_skin_kernel(const float* __restrict__ v_posed, const float* __restrict__ A,
                                                       const float* __restrict__ transl, const float4* __restrict__ w4,
                                                       const uint32_t* __restrict__ idx4, float* __restrict__ verts,
                                                       int N, int V, int J, int hoist) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_v = smem;                      // 3072 floats
  float* s_A = smem + kSkinVerts * 3;     // 2 frames x J x 12 floats
  const int tid = threadIdx.x;
  const long long total = (long long)N * V;
  const long long gv0 = (long long)blockIdx.x * kSkinVerts;
  const int n0 = (int)(gv0 / V);
  const long long total3 = total * 3;
  const long long fbase = gv0 * 3;        // first float of the window (multiple of 4 floats)

  // A of the (at most two) frames this window touches
  {
    const int per = J * 3;                // float4s per frame
    const float4* A4 = reinterpret_cast<const float4*>(A);
    float4* sA4 = reinterpret_cast<float4*>(s_A);
    for (int i = tid; i < 2 * per; i += 256) {
      const int fr = i / per;
      if (n0 + fr < N) sA4[i] = A4[(size_t)(n0 + fr) * per + (i - fr * per)];
    }
  }
  // coalesced 16-byte loads of the window
  {
    const float4* src = reinterpret_cast<const float4*>(v_posed + fbase);
    float4* dst = reinterpret_cast<float4*>(s_v);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int i = tid + 256 * q;
      const long long fl = fbase + (long long)i * 4;
      if (fl + 3 < total3) {
        dst[i] = src[i];
      } else {
        for (int e = 0; e < 4; ++e)
          if (fl + e < total3) s_v[i * 4 + e] = v_posed[fl + e];
      }
    }
  }
  const long long frame1_start = (long long)(n0 + 1) * V;
  float4 hw[4];
  uint32_t hid[4];
  if (hoist) {
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
      const long long gv = gv0 + tid + 256 * sidx;
      const int fr = gv >= frame1_start ? 1 : 0;
      int v = (int)(gv - (long long)(n0 + fr) * V);
      if (gv >= total) v = 0;
      hw[sidx] = w4[v];
      hid[sidx] = idx4[v];
    }
  }
  __syncthreads();
#pragma unroll
  for (int sidx = 0; sidx < 4; ++sidx) {
    const int lv = tid + 256 * sidx;
    const long long gv = gv0 + lv;
    if (gv < total) {
      const int fr = gv >= frame1_start ? 1 : 0;
      const int n = n0 + fr;
      const int v = (int)(gv - (long long)n * V);
      const float4 wv = hoist ? hw[sidx] : w4[v];
      const uint32_t id = hoist ? hid[sidx] : idx4[v];
      const float x = s_v[lv * 3], y = s_v[lv * 3 + 1], z = s_v[lv * 3 + 2];
      const float4* Af = reinterpret_cast<const float4*>(s_A + fr * J * 12);
      const float wq[4] = {wv.x, wv.y, wv.z, wv.w};
      float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int jq = (id >> (8 * q)) & 0xff;
        const float4 a0 = Af[jq * 3], a1 = Af[jq * 3 + 1], a2 = Af[jq * 3 + 2];
        r0.x = fmaf(wq[q], a0.x, r0.x); r0.y = fmaf(wq[q], a0.y, r0.y); r0.z = fmaf(wq[q], a0.z, r0.z); r0.w = fmaf(wq[q], a0.w, r0.w);
        r1.x = fmaf(wq[q], a1.x, r1.x); r1.y = fmaf(wq[q], a1.y, r1.y); r1.z = fmaf(wq[q], a1.z, r1.z); r1.w = fmaf(wq[q], a1.w, r1.w);
        r2.x = fmaf(wq[q], a2.x, r2.x); r2.y = fmaf(wq[q], a2.y, r2.y); r2.z = fmaf(wq[q], a2.z, r2.z); r2.w = fmaf(wq[q], a2.w, r2.w);
      }
      // T = [R00 R01 R02 R10 | R11 R12 R20 R21 | R22 t0 t1 t2]
      float tx = 0.f, ty = 0.f, tz = 0.f;
      if (transl) { tx = transl[(size_t)n * 3]; ty = transl[(size_t)n * 3 + 1]; tz = transl[(size_t)n * 3 + 2]; }
      const float ox = fmaf(r0.x, x, fmaf(r0.y, y, fmaf(r0.z, z, r2.y))) + tx;
      const float oy = fmaf(r0.w, x, fmaf(r1.x, y, fmaf(r1.y, z, r2.z))) + ty;
      const float oz = fmaf(r1.z, x, fmaf(r1.w, y, fmaf(r2.x, z, r2.w))) + tz;
      s_v[lv * 3] = ox; s_v[lv * 3 + 1] = oy; s_v[lv * 3 + 2] = oz;
    }
  }
  __syncthreads();
  {
    float4* dst = reinterpret_cast<float4*>(verts + fbase);
    const float4* src = reinterpret_cast<const float4*>(s_v);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int i = tid + 256 * q;
      const long long fl = fbase + (long long)i * 4;
      if (fl + 3 < total3) {
        dst[i] = src[i];
      } else {
        for (int e = 0; e < 4; ++e)
          if (fl + e < total3) verts[fl + e] = s_v[i * 4 + e];
      }
    }
  }
}

typedef float vf4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ void store4(float* dst, float a, float b, float c, float d) {
  vf4 v = {a, b, c, d};
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<vf4*>(dst));
  else *reinterpret_cast<vf4*>(dst) = v;
}

__device__ __forceinline__ void skin_one(const float* s_A, int J, int fr, float4 wv, uint32_t id, float x, float y, float z,
                                         float tx, float ty, float tz, float& ox, float& oy, float& oz) {
  const float4* Af = reinterpret_cast<const float4*>(s_A + fr * J * 12);
  const float wq[4] = {wv.x, wv.y, wv.z, wv.w};
  float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int jq = (id >> (8 * q)) & 0xff;
    const float4 a0 = Af[jq * 3], a1 = Af[jq * 3 + 1], a2 = Af[jq * 3 + 2];
    r0.x = fmaf(wq[q], a0.x, r0.x); r0.y = fmaf(wq[q], a0.y, r0.y); r0.z = fmaf(wq[q], a0.z, r0.z); r0.w = fmaf(wq[q], a0.w, r0.w);
    r1.x = fmaf(wq[q], a1.x, r1.x); r1.y = fmaf(wq[q], a1.y, r1.y); r1.z = fmaf(wq[q], a1.z, r1.z); r1.w = fmaf(wq[q], a1.w, r1.w);
    r2.x = fmaf(wq[q], a2.x, r2.x); r2.y = fmaf(wq[q], a2.y, r2.y); r2.z = fmaf(wq[q], a2.z, r2.z); r2.w = fmaf(wq[q], a2.w, r2.w);
  }
  // T = [R00 R01 R02 R10 | R11 R12 R20 R21 | R22 t0 t1 t2]
  ox = fmaf(r0.x, x, fmaf(r0.y, y, fmaf(r0.z, z, r2.y))) + tx;
  oy = fmaf(r0.w, x, fmaf(r1.x, y, fmaf(r1.y, z, r2.z))) + ty;
  oz = fmaf(r1.z, x, fmaf(r1.w, y, fmaf(r2.x, z, r2.w))) + tz;
}

// Variant 2: no LDS staging of the vertices.  A block owns a window of 1024*GPT consecutive global vertices (16-byte
// aligned, at most two frames); thread t owns the four-vertex groups t, t+256, ... (GPT of them) = GPT x three 16-byte
// loads / stores, all issued up front: the kernel is latency-bound, so bytes in flight per wave are what matters
// (rocprofv3: ~16 resident waves/CU, LDS bank conflicts 6 % of LDS-active cycles -- profiles/r01_run5_pmc_lbs).
template <bool NT, int GPT>
__global__ __launch_bounds__(256) void lbs_skin_direct_kernel(const float* __restrict__ v_posed, const float* __restrict__ A,
                                                              const float* __restrict__ transl, const float4* __restrict__ w4,
                                                              const uint32_t* __restrict__ idx4, float* __restrict__ verts,
                                                              int N, int V, int J, int nblocks) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_A = smem;   // 2 frames x J x 12
  const int tid = threadIdx.x;
  constexpr int WIN = kSkinVerts * GPT;
  const int win = blockIdx.x;       // (an XCD-contiguous remap of the windows measured neutral-to-worse)
  (void)nblocks;
  const long long total = (long long)N * V;
  const long long gv0 = (long long)win * WIN;
  const int n0 = (int)(gv0 / V);
  const long long total3 = total * 3;
  const long long frame1_start = (long long)(n0 + 1) * V;
  float f[GPT][12];
  float4 wv[GPT][4];
  uint32_t id[GPT][4];
  int frs[GPT][4];
  bool full[GPT];
  // issue every global load up front
#pragma unroll
  for (int g = 0; g < GPT; ++g) {
    const long long gvt = gv0 + 4 * (tid + 256 * g);
    const long long fl = gvt * 3;
    full[g] = fl + 11 < total3;
    if (full[g]) {
      const vf4* src = reinterpret_cast<const vf4*>(v_posed + fl);
      vf4 a, b, c;
      a = src[0]; b = src[1]; c = src[2];     // (non-temporal LOADS measured 25-35 % slower; only the stores are NT)
      f[g][0] = a.x; f[g][1] = a.y; f[g][2] = a.z; f[g][3] = a.w; f[g][4] = b.x; f[g][5] = b.y; f[g][6] = b.z; f[g][7] = b.w;
      f[g][8] = c.x; f[g][9] = c.y; f[g][10] = c.z; f[g][11] = c.w;
    } else {
#pragma unroll
      for (int e = 0; e < 12; ++e) f[g][e] = fl + e < total3 ? v_posed[fl + e] : 0.f;
    }
  }
#pragma unroll
  for (int g = 0; g < GPT; ++g)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long gv = gv0 + 4 * (tid + 256 * g) + k;
      const int fr = gv >= frame1_start ? 1 : 0;
      int v = (int)(gv - (long long)(n0 + fr) * V);
      if (gv >= total) v = 0;
      frs[g][k] = fr;
      wv[g][k] = w4[v];
      id[g][k] = idx4[v];
    }
  float tl[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  if (transl) {
#pragma unroll
    for (int fr = 0; fr < 2; ++fr)
      if (n0 + fr < N) {
#pragma unroll
        for (int c = 0; c < 3; ++c) tl[fr][c] = transl[(size_t)(n0 + fr) * 3 + c];
      }
  }
  {
    const int per = J * 3;
    const float4* A4 = reinterpret_cast<const float4*>(A);
    float4* sA4 = reinterpret_cast<float4*>(s_A);
    for (int i = tid; i < 2 * per; i += 256) {
      const int fr = i / per;
      if (n0 + fr < N) sA4[i] = A4[(size_t)(n0 + fr) * per + (i - fr * per)];
    }
  }
  __syncthreads();
#pragma unroll
  for (int g = 0; g < GPT; ++g) {
    float o[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int fr = frs[g][k];
      skin_one(s_A, J, fr, wv[g][k], id[g][k], f[g][3 * k], f[g][3 * k + 1], f[g][3 * k + 2], tl[fr][0], tl[fr][1], tl[fr][2],
               o[3 * k], o[3 * k + 1], o[3 * k + 2]);
    }
    const long long fl = (gv0 + 4 * (tid + 256 * g)) * 3;
    if (full[g]) {
      float* dst = verts + fl;
      store4<NT>(dst, o[0], o[1], o[2], o[3]);
      store4<NT>(dst + 4, o[4], o[5], o[6], o[7]);
      store4<NT>(dst + 8, o[8], o[9], o[10], o[11]);
    } else {
#pragma unroll
      for (int e = 0; e < 12; ++e)
        if (fl + e < total3) verts[fl + e] = o[e];
    }
  }
}

// Variant 3: frame-PAIR windows.  V is even (6890), so two consecutive frames form a 16-byte aligned unit of V/2 groups of
// four vertices whose (frame, vertex) pattern is identical in every pair.  A block owns one 256-group window and walks
// P consecutive pairs with the per-thread skinning weights held in registers (weight/index traffic / P, one barrier per
// pair), prefetching the next pair's 48 bytes per thread while the current one is skinned.
template <bool NT, int P>
__global__ __launch_bounds__(256) void lbs_skin_pairs_kernel(const float* __restrict__ v_posed, const float* __restrict__ A,
                                                             const float* __restrict__ transl, const float4* __restrict__ w4,
                                                             const uint32_t* __restrict__ idx4, float* __restrict__ verts,
                                                             int N, int V, int J, int nwin) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 buffers x 2 frames x J x 12
  const int tid = threadIdx.x;
  const int win = blockIdx.x % nwin, pg = blockIdx.x / nwin;
  const int ngroups = V / 2;                   // groups of 4 vertices per frame pair
  const int grp = win * 256 + tid;
  const bool lane_ok = grp < ngroups;
  const int q0 = 4 * grp;                      // flat vertex offset inside the pair
  const int per = J * 3;                       // float4s of A per frame
  // per-thread constants: the four vertices' frame-in-pair, weights and joints
  float4 wv[4];
  uint32_t id[4];
  int frs[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int q = q0 + k;
    frs[k] = q >= V ? 1 : 0;
    const int v = lane_ok ? q - frs[k] * V : 0;
    wv[k] = w4[v];
    id[k] = idx4[v];
  }
  const int npairs = (N + 1) / 2;
  const int pair0 = pg * P;
  auto load_pair = [&](int pair, float (&f)[12]) {
    const long long fl = ((long long)2 * pair * V + q0) * 3;
    const bool full = lane_ok && (2 * pair + 1 < N || q0 + 3 < V);
    if (full) {
      const vf4* src = reinterpret_cast<const vf4*>(v_posed + fl);
      vf4 a, b, c;
      if (NT) { a = __builtin_nontemporal_load(src); b = __builtin_nontemporal_load(src + 1); c = __builtin_nontemporal_load(src + 2); }
      else { a = src[0]; b = src[1]; c = src[2]; }
      f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
      f[8] = c.x; f[9] = c.y; f[10] = c.z; f[11] = c.w;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool ok = lane_ok && 2 * pair + frs[k] < N;
#pragma unroll
        for (int c = 0; c < 3; ++c) f[3 * k + c] = ok ? v_posed[fl + 3 * k + c] : 0.f;
      }
    }
  };
  float cur[12], nxt[12];
  if (pair0 < npairs) load_pair(pair0, cur);
#pragma unroll 1
  for (int p = 0; p < P; ++p) {
    const int pair = pair0 + p;
    if (pair >= npairs) break;
    float* s_A = smem + (p & 1) * 2 * J * 12;
    {
      const float4* A4 = reinterpret_cast<const float4*>(A);
      float4* sA4 = reinterpret_cast<float4*>(s_A);
      for (int i = tid; i < 2 * per; i += 256) {
        const int fr = i / per;
        if (2 * pair + fr < N) sA4[i] = A4[(size_t)(2 * pair + fr) * per + (i - fr * per)];
      }
    }
    if (p + 1 < P && pair + 1 < npairs) load_pair(pair + 1, nxt);
    float tl[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    if (transl) {
#pragma unroll
      for (int fr = 0; fr < 2; ++fr)
        if (2 * pair + fr < N) {
#pragma unroll
          for (int c = 0; c < 3; ++c) tl[fr][c] = transl[(size_t)(2 * pair + fr) * 3 + c];
        }
    }
    __syncthreads();      // A of this pair is in LDS (the other buffer may still be read by slower waves of the previous pair)
    float o[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int fr = frs[k];
      skin_one(s_A, J, fr, wv[k], id[k], cur[3 * k], cur[3 * k + 1], cur[3 * k + 2], tl[fr][0], tl[fr][1], tl[fr][2], o[3 * k],
               o[3 * k + 1], o[3 * k + 2]);
    }
    const long long fl = ((long long)2 * pair * V + q0) * 3;
    const bool full = lane_ok && (2 * pair + 1 < N || q0 + 3 < V);
    if (full) {
      float* dst = verts + fl;
      store4<NT>(dst, o[0], o[1], o[2], o[3]);
      store4<NT>(dst + 4, o[4], o[5], o[6], o[7]);
      store4<NT>(dst + 8, o[8], o[9], o[10], o[11]);
    } else if (lane_ok) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (2 * pair + frs[k] < N) {
#pragma unroll
          for (int c = 0; c < 3; ++c) verts[fl + 3 * k + c] = o[3 * k + c];
        }
    }
#pragma unroll
    for (int e = 0; e < 12; ++e) cur[e] = nxt[e];
  }
}

static void fill_model(FrameParams& p, const ha_smpl_model* m, int slot) {
  memset(&p, 0, sizeof(p));
  p.Jt = m->Jt; p.Js = m->Js; p.parents = m->parents; p.jdepth = m->jdepth;
  p.child_start = m->child_start; p.child_idx = m->child_idx;
  p.J = m->J; p.NB = m->NB; p.Kfull = m->Kfull; p.Kfull_pad = m->Kfull_pad; p.kf4 = (m->Kfull_pad + 3) & ~3; p.depth = m->depth;
  const VertexSet& s = m->sets[slot];
  p.Pd_v = s.Pd_v; p.w = s.w; p.idx = s.idx;
  p.nverts = s.n; p.nchunks = s.nchunks; p.nnz = m->nnz;
}

}  // namespace ha

static int check_common(const char* fn, const ha_smpl_model* m, int slot, int N, int n_active) {
  HA_REQUIRE(m, "%s: null model", fn);
  HA_REQUIRE(slot >= 0 && slot < kMaxSubsets && m->sets[slot].n > 0, "%s: vertex subset %d is not defined", fn, slot);
  HA_REQUIRE(N >= 1, "%s: N=%d must be >= 1", fn, N);
  HA_REQUIRE(n_active >= 1 && n_active <= m->J, "%s: n_active_joints=%d out of range 1..%d", fn, n_active, m->J);
  return HA_OK;
}

extern "C" int ha_smpl_workspace(const ha_smpl_model* m, int N, int n_active, int64_t* vposed, int64_t* coeff) {
  int rc = check_common("ha_smpl_workspace", m, 0, N, n_active);
  if (rc != HA_OK) return rc;
  const int Kc = m->NB + 1 + (n_active - 1) * 9;
  const int64_t Npad = (int64_t)ceil_div(N, 64) * 64;
  if (vposed) *vposed = (int64_t)N * m->V * 3 + 4;   // +4: the streaming kernel's last 16-byte vector
  if (coeff) *coeff = (int64_t)(Kc + (Kc & 1)) * Npad;
  return HA_OK;
}

extern "C" int ha_lbs_skin(const ha_smpl_model* m, int N, const float* v_posed, const float* A, const float* transl,
                           float* verts, void* stream) {
  HA_REQUIRE(m && v_posed && A && verts, "ha_lbs_skin: null argument");
  HA_REQUIRE(N >= 1, "ha_lbs_skin: N must be >= 1");
  if (m->nnz > 4 || m->V < kSkinVerts) {
    set_error("ha_lbs_skin: needs <=4 influences per vertex and V>=%d (model has nnz=%d V=%d)", kSkinVerts, m->nnz, m->V);
    return HA_ERR_UNSUPPORTED;
  }
  DeviceGuard guard(m->device);
  const long long total = (long long)N * m->V;
  const int blocks = (int)((total + kSkinVerts - 1) / kSkinVerts);
  const size_t lds = (size_t)(kSkinVerts * 3 + 2 * m->J * 12) * sizeof(float);
  const int sv = g_skin_variant >= 0 ? g_skin_variant : (total * 12 <= (200ll << 20) ? 6 : 2);
  const int variant = sv & 3;
  const bool nt = (sv & 4) != 0;
  if (variant == 3 && m->V % 2 == 0) {
    constexpr int P = 4;
    const int nwin = ceil_div(m->V / 2, 256);
    const int npg = ceil_div(ceil_div(N, 2), P);
    const size_t lds3 = (size_t)(2 * 2 * m->J * 12) * sizeof(float);
    if (nt)
      hipLaunchKernelGGL((lbs_skin_pairs_kernel<true, P>), dim3(nwin * npg), dim3(256), lds3, (hipStream_t)stream, v_posed, A, transl,
                         m->w4, m->idx4, verts, N, m->V, m->J, nwin);
    else
      hipLaunchKernelGGL((lbs_skin_pairs_kernel<false, P>), dim3(nwin * npg), dim3(256), lds3, (hipStream_t)stream, v_posed, A, transl,
                         m->w4, m->idx4, verts, N, m->V, m->J, nwin);
  } else if (variant == 2 || variant == 3) {
    const size_t lds2 = (size_t)(2 * m->J * 12) * sizeof(float);
    const int gpt = (sv >> 3) & 3;      // 0: 1 group/thread, 1: 2, 2: 4
#define HA_SKIN_LAUNCH(NTV, G)                                                                                          \
    do {                                                                                                                 \
      const int nb = (int)((total + (long long)kSkinVerts * G - 1) / ((long long)kSkinVerts * G));                       \
      hipLaunchKernelGGL((lbs_skin_direct_kernel<NTV, G>), dim3(nb), dim3(256), lds2, (hipStream_t)stream, v_posed, A,   \
                         transl, m->w4, m->idx4, verts, N, m->V, m->J, nb);                                              \
    } while (0)
    if (gpt == 2 && m->V >= 4 * kSkinVerts) { if (nt) HA_SKIN_LAUNCH(true, 4); else HA_SKIN_LAUNCH(false, 4); }
    else if (gpt == 1 && m->V >= 2 * kSkinVerts) { if (nt) HA_SKIN_LAUNCH(true, 2); else HA_SKIN_LAUNCH(false, 2); }
    else { if (nt) HA_SKIN_LAUNCH(true, 1); else HA_SKIN_LAUNCH(false, 1); }
#undef HA_SKIN_LAUNCH
  } else {
    hipLaunchKernelGGL(lbs_skin_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, v_posed, A, transl, m->w4, m->idx4,
                       verts, N, m->V, m->J, variant);
  }
  HA_LAUNCH_CHECK();
  return HA_OK;
}

extern "C" int ha_smpl_forward(const ha_smpl_model* m, int slot, int N, int n_active, const float* pose, const float* betas,
                               const float* transl, float* verts, float* joints, float* A_out, float* ws_vposed,
                               float* ws_coeff, int algo, void* stream) {
  int rc = check_common("ha_smpl_forward", m, slot, N, n_active);
  if (rc != HA_OK) return rc;
  HA_REQUIRE(pose && betas, "ha_smpl_forward: pose and betas are required");
  HA_REQUIRE(algo >= 0 && algo <= 2, "ha_smpl_forward: unknown algo %d", algo);
  const bool dense_ok = slot == 0 && m->nnz <= 4 && m->V >= kSkinVerts && ws_vposed && ws_coeff && A_out;
  if (algo == 2 && !dense_ok) {
    set_error("ha_smpl_forward: algo 2 needs slot 0, <=4 skinning influences, V>=%d and the A/vposed/coeff workspaces", kSkinVerts);
    return HA_ERR_INVALID_ARG;
  }
  if (algo == 0) algo = (dense_ok && verts) ? 2 : 1;
  DeviceGuard guard(m->device);
  hipStream_t st = (hipStream_t)stream;

  FrameParams p;
  fill_model(p, m, slot);
  p.N = N; p.n_active = n_active; p.Kc = m->NB + 1 + (n_active - 1) * 9;
  p.pose = pose; p.betas = betas; p.transl = transl;
  p.joints = joints; p.A_out = A_out;
  const size_t lds = (size_t)FW * (((m->Kfull_pad + 3) & ~3) + m->J * 12) * sizeof(float);
  const int blocks = ceil_div(N, FW);
  if (algo == 1) {
    p.verts = verts;
    hipLaunchKernelGGL(smpl_frame_fwd_kernel, dim3(blocks), dim3(FW * 64), lds, st, p);
    HA_LAUNCH_CHECK();
    return HA_OK;
  }
  // algo 2: joints/A/coefficients by the frame kernel, then MFMA blend, then streaming skinning
  p.verts = nullptr;
  p.nchunks = 0;
  p.coeffT = ws_coeff;
  p.Npad = ceil_div(N, 64) * 64;
  hipLaunchKernelGGL(smpl_frame_fwd_kernel, dim3(blocks), dim3(FW * 64), lds, st, p);
  HA_LAUNCH_CHECK();
  if (!verts) return HA_OK;
  {
    const int n_vt = m->Vpad / 32, n_ft = p.Npad / 64;
    const int n_vtg = ceil_div(n_vt, 4);
    const int n_vtg8 = ceil_div(n_vtg, 8) * 8;
    const int KP = (p.Kc + 1) / 2;
    hipLaunchKernelGGL(pose_blend_mfma_kernel, dim3(n_vtg8 * n_ft), dim3(256), 0, st, ws_coeff, p.Npad, KP, m->Kfull_pad / 2,
                       m->Pd_m, ws_vposed, N, m->V, n_vt, n_ft);
    HA_LAUNCH_CHECK();
  }
  return ha_lbs_skin(m, N, ws_vposed, A_out, transl, verts, stream);
}

extern "C" int ha_smpl_backward(const ha_smpl_model* m, int slot, int N, int n_active, const float* pose, const float* betas,
                                const float* g_verts, const float* g_joints, float* g_pose, float* g_betas, float* g_transl,
                                void* stream) {
  int rc = check_common("ha_smpl_backward", m, slot, N, n_active);
  if (rc != HA_OK) return rc;
  HA_REQUIRE(pose && betas, "ha_smpl_backward: pose and betas are required");
  DeviceGuard guard(m->device);
  FrameParams p;
  fill_model(p, m, slot);
  p.N = N; p.n_active = n_active; p.Kc = m->NB + 1 + (n_active - 1) * 9;
  p.pose = pose; p.betas = betas;
  p.g_verts = g_verts; p.g_joints = g_joints;
  p.g_pose = g_pose; p.g_betas = g_betas; p.g_transl = g_transl;
  const size_t lds = (size_t)FW * (2 * ((m->Kfull_pad + 3) & ~3) + m->J * (12 * 3 + 16) + 192) * sizeof(float);
  hipLaunchKernelGGL(smpl_frame_bwd_kernel, dim3(ceil_div(N, FW)), dim3(FW * 64), lds, (hipStream_t)stream, p);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

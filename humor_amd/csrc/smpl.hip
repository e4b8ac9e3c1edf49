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
//   lbs_skin_wave_kernel    dense path: the HBM-streaming skinning kernel, lane-contiguous 16-byte loads / non-temporal
//                           stores of the [N,V,3] arrays through a per-wave LDS transpose, A of the touched frames in LDS.
#include <stdarg.h>
#include <string.h>
#include <algorithm>

#include <vector>

#include "smpl_model.h"
#ifndef HA_SIMT_EMU
#include "lane_reduce.h"
#endif

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
// ha_tune_set("skin_variant"): -1 = auto; else bits 0-1: waves per block 4 << b; +4: non-temporal stores;
// +8 / +16: profiling modes of the streaming kernel (LDS-transposed copy only / every lane gathers bone 0)
int g_skin_variant = -1;
int g_dense_gA_sparse = 2;   // ha_tune_set("dense_gA_sparse"): dL/dA of the dense backward -- 2 (default) = MFMA product over the joints each 64-vertex
                             // chunk touches (205 us at N = 1920), 1 = by joint lists (255 us), 0 = dense 64-column MFMA product (417 us)
int g_dense_bwd_waves = 0;   // ha_tune_set("dense_bwd_waves"): wave-count target of the dense backward's K split (0 = default)
extern int g_layer_spb, g_layer_nw, g_layer_finish, g_gemm_rm, g_layer_hsum, g_layer_acc, g_rollout_groups, g_gemm_ks, g_rollout_persist, g_rollout_persist_bwd, g_rollout_persist_inject, g_rollout_pipe, g_rollout_pipe_bwd;   // rollout.hip
extern unsigned g_cu_poison;   // debug.hip
}
extern "C" int ha_tune_set(const char* key, int value) {
  HA_REQUIRE(key, "ha_tune_set: null key");
  if (strcmp(key, "skin_variant") == 0) { ha::g_skin_variant = value; return HA_OK; }
  if (strcmp(key, "dense_bwd_waves") == 0) { ha::g_dense_bwd_waves = value; return HA_OK; }
  if (strcmp(key, "dense_gA_sparse") == 0) { ha::g_dense_gA_sparse = value; return HA_OK; }
  if (strcmp(key, "layer_spb") == 0) { ha::g_layer_spb = value; return HA_OK; }
  if (strcmp(key, "layer_nw") == 0) { ha::g_layer_nw = value; return HA_OK; }
  if (strcmp(key, "layer_finish") == 0) { ha::g_layer_finish = value; return HA_OK; }
  if (strcmp(key, "gemm_rm") == 0) { ha::g_gemm_rm = value; return HA_OK; }
  if (strcmp(key, "layer_hsum") == 0) { ha::g_layer_hsum = value; return HA_OK; }
  if (strcmp(key, "layer_acc") == 0) { ha::g_layer_acc = value; return HA_OK; }
  if (strcmp(key, "rollout_groups") == 0) { ha::g_rollout_groups = value; return HA_OK; }
  if (strcmp(key, "gemm_ks") == 0) { ha::g_gemm_ks = value; return HA_OK; }
  if (strcmp(key, "rollout_persist") == 0) { ha::g_rollout_persist = value; return HA_OK; }
  if (strcmp(key, "rollout_persist_bwd") == 0) { ha::g_rollout_persist_bwd = value; return HA_OK; }
  if (strcmp(key, "rollout_pipe") == 0) { ha::g_rollout_pipe = value; return HA_OK; }
  if (strcmp(key, "rollout_pipe_bwd") == 0) { ha::g_rollout_pipe_bwd = value; return HA_OK; }
  if (strcmp(key, "rollout_persist_inject") == 0) { ha::g_rollout_persist_inject = value; return HA_OK; }
  if (strcmp(key, "cu_poison") == 0) { ha::g_cu_poison = (unsigned)value; return HA_OK; }
  ha::set_error("ha_tune_set: unknown key '%s'", key);
  return HA_ERR_INVALID_ARG;
}
extern "C" const char* ha_last_error(void) { return ha::g_err; }
extern "C" int ha_abi_version(void) { return 3; }
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
  if (s.Pd_k) (void)hipFree(s.Pd_k);
  if (s.w) (void)hipFree(s.w);
  if (s.idx) (void)hipFree(s.idx);
  if (s.Wc) (void)hipFree(s.Wc);
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
  const int Kp = ceil_div(K, 128) * 128;   // whole 128-coefficient groups (dense_gco_kernel reads them as one)
  std::vector<float> pdk((size_t)s.nchunks * 3 * kChunk * Kp, 0.0f);   // coefficient-major copy for the adjoint
  std::vector<float> w((size_t)s.nchunks * nnz * kChunk, 0.0f);
  std::vector<int32_t> ix((size_t)s.nchunks * nnz * kChunk, 0);
  std::vector<float> wc((size_t)s.nchunks * kChunk * kMaxJoints, 0.0f);   // [chunk][vertex][joint], dense (the adjoint's dL/dA sum)
  for (int i = 0; i < n; ++i) {
    const int v = ids ? ids[i] : i;
    const int ch = i / kChunk, ln = i % kChunk;
    for (int k = 0; k < K; ++k)
      for (int c = 0; c < 3; ++c)
      {
        const float val = m->h_Pd[((size_t)k * V + v) * 3 + c];
        pd[(((size_t)ch * K + k) * 3 + c) * kChunk + ln] = val;
        pdk[((size_t)ch * 3 * kChunk + c * kChunk + ln) * Kp + k] = val;
      }
    for (int q = 0; q < nnz; ++q) {
      w[((size_t)ch * nnz + q) * kChunk + ln] = m->h_w[(size_t)v * nnz + q];
      ix[((size_t)ch * nnz + q) * kChunk + ln] = m->h_idx[(size_t)v * nnz + q];
      wc[((size_t)ch * kChunk + ln) * kMaxJoints + m->h_idx[(size_t)v * nnz + q]] += m->h_w[(size_t)v * nnz + q];
    }
  }
  int rc;
  if ((rc = upload(&s.Pd_v, pd)) != HA_OK) return rc;
  if ((rc = upload(&s.Pd_k, pdk)) != HA_OK) return rc;
  if ((rc = upload(&s.w, w)) != HA_OK) return rc;
  if ((rc = upload(&s.idx, ix)) != HA_OK) return rc;
  if ((rc = upload(&s.Wc, wc)) != HA_OK) return rc;
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
  // ancestors at distance 2^r for the pointer-jumping forward chain
  m->nrounds = 0;
  while ((1 << m->nrounds) <= m->depth) ++m->nrounds;
  std::vector<int32_t> anc((size_t)(m->nrounds > 0 ? m->nrounds : 1) * kMaxJoints, -1);
  for (int j = 0; j < J; ++j) anc[j] = par[j];
  for (int r = 1; r < m->nrounds; ++r)
    for (int j = 0; j < J; ++j) {
      const int a = anc[(size_t)(r - 1) * kMaxJoints + j];
      anc[(size_t)r * kMaxJoints + j] = a >= 0 ? anc[(size_t)(r - 1) * kMaxJoints + a] : -1;
    }

  // pre-contracted joint regressor (double accumulation, rounded once)
  std::vector<float> Jt((size_t)J * 3), Js((size_t)(NB > 0 ? NB : 1) * 3 * kMaxJoints, 0.0f);   // Js: [NB][3][64], lane = joint
  for (int j = 0; j < J; ++j)
    for (int c = 0; c < 3; ++c) {
      double acc = 0.0;
      for (int v = 0; v < V; ++v) acc += (double)J_regressor[(size_t)j * V + v] * (double)v_template[(size_t)v * 3 + c];
      Jt[j * 3 + c] = (float)acc;
      for (int l = 0; l < NB; ++l) {
        double a2 = 0.0;
        for (int v = 0; v < V; ++v)
          a2 += (double)J_regressor[(size_t)j * V + v] * (double)shapedirs[((size_t)v * 3 + c) * NB + l];
        Js[((size_t)l * 3 + c) * kMaxJoints + j] = (float)a2;
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
  if ((rc = upload(&m->anc, anc)) != HA_OK) return fail(rc);
  if ((rc = build_set(m, 0, nullptr, V)) != HA_OK) return fail(rc);

  // MFMA B-operand layout [Vpad/32][KQ][3][64][4]: lane l <-> (vertex = vt*32 + (l&31), k = 2*kp + (l>>5)); the 12 floats a lane
  // needs for four consecutive k-pairs ((kp % 4) * 3 + component) are three 16-byte loads, each contiguous across the wave
  {
    const int nvt = m->Vpad / 32, KQ = ceil_div(m->Kfull_pad / 2, 4);
    std::vector<float> pm((size_t)nvt * KQ * 3 * 64 * 4, 0.0f);
    for (int vt = 0; vt < nvt; ++vt)
      for (int kp = 0; kp < KQ * 4; ++kp)
        for (int c = 0; c < 3; ++c)
          for (int l = 0; l < 64; ++l) {
            const int k = 2 * kp + (l >> 5), v = vt * 32 + (l & 31);
            const int e = (kp & 3) * 3 + c;           // position among the lane's 12 floats of this k-pair quad
            if (k < K && v < V)
              pm[((((size_t)vt * KQ + (kp >> 2)) * 3 + (e >> 2)) * 64 + l) * 4 + (e & 3)] = m->h_Pd[((size_t)k * V + v) * 3 + c];
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
    // dense skinning weights [Vpad][64] (B operand of the dense backward's dL/dA product)
    std::vector<float> wd((size_t)m->Vpad * 64, 0.0f);
    for (int v = 0; v < V; ++v)
      for (int q = 0; q < nnz; ++q) wd[(size_t)v * 64 + m->h_idx[(size_t)v * nnz + q]] += m->h_w[(size_t)v * nnz + q];
    if ((rc = upload(&m->Wd, wd)) != HA_OK) return fail(rc);
  }
  {
    // weights by joint: start offsets, (vertex, weight) entries in vertex order, joints by decreasing list length
    std::vector<int32_t> start(J + 1, 0), order(J);
    for (int v = 0; v < V; ++v)
      for (int q = 0; q < nnz; ++q)
        if (m->h_w[(size_t)v * nnz + q] != 0.0f) start[m->h_idx[(size_t)v * nnz + q] + 1]++;
    for (int j = 0; j < J; ++j) start[j + 1] += start[j];
    std::vector<int32_t> jv(start[J] > 0 ? start[J] : 1, 0), fill(start.begin(), start.end() - 1);
    std::vector<float> jw(start[J] > 0 ? start[J] : 1, 0.0f);
    for (int v = 0; v < V; ++v)
      for (int q = 0; q < nnz; ++q) {
        const float wv = m->h_w[(size_t)v * nnz + q];
        if (wv != 0.0f) { const int e = fill[m->h_idx[(size_t)v * nnz + q]]++; jv[e] = v; jw[e] = wv; }
      }
    for (int j = 0; j < J; ++j) order[j] = j;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return start[a + 1] - start[a] > start[b + 1] - start[b]; });
    if ((rc = upload(&m->ja_start, start)) != HA_OK || (rc = upload(&m->ja_v, jv)) != HA_OK || (rc = upload(&m->ja_w, jw)) != HA_OK ||
        (rc = upload(&m->ja_order, order)) != HA_OK)
      return fail(rc);
    // chunk-local compressed joint dimension (experiment): the joints touched by each 64-vertex chunk, in slots
    const int nch = m->Vpad / 64;
    std::vector<int32_t> gcj((size_t)nch * 32, -1), gng(nch, 1);
    std::vector<float> gcw((size_t)nch * 64 * 32, 0.0f);
    bool fits = true;
    for (int c = 0; c < nch && fits; ++c) {
      int used = 0;
      for (int vl = 0; vl < 64 && fits; ++vl) {
        const int v = c * 64 + vl;
        if (v >= V) break;
        for (int q = 0; q < nnz; ++q) {
          const float wv = m->h_w[(size_t)v * nnz + q];
          if (wv == 0.0f) continue;
          const int j = m->h_idx[(size_t)v * nnz + q];
          int slot = -1;
          for (int u = 0; u < used; ++u)
            if (gcj[(size_t)c * 32 + u] == j) slot = u;
          if (slot < 0) {
            if (used == 32) { fits = false; break; }
            slot = used++;
            gcj[(size_t)c * 32 + slot] = j;
          }
          // packed for the kernel's lanes: [chunk][slot group][k-quarter = vl & 3][slot & 15][e = vl >> 2]
          gcw[((((size_t)c * 2 + (slot >> 4)) * 4 + (vl & 3)) * 16 + (slot & 15)) * 16 + (vl >> 2)] += wv;
        }
      }
      gng[c] = used > 16 ? 2 : 1;
    }
    if (fits && ((rc = upload(&m->gc_joint, gcj)) != HA_OK || (rc = upload(&m->gc_w, gcw)) != HA_OK || (rc = upload(&m->gc_ng, gng)) != HA_OK))
      return fail(rc);
  }
  *out = m;
  return HA_OK;
}

extern "C" int ha_smpl_model_destroy(ha_smpl_model* m) {
  if (!m) return HA_OK;
  DeviceGuard guard(m->device);
  for (int s = 0; s < kMaxSubsets; ++s) free_set(m->sets[s]);
  void* ptrs[] = {m->Jt, m->Js, m->parents, m->jdepth, m->child_start, m->child_idx, m->Pd_m, m->w4, m->idx4, m->Wd,
                  m->ja_start, m->ja_v, m->ja_w, m->ja_order, m->anc, m->gc_joint, m->gc_w, m->gc_ng};
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

#ifdef HA_SMPL_TIMING
// profiling build only (tools/smpl_phase_timing.py): phase timestamps of wave 0 of block 0
__device__ unsigned long long g_smpl_pt[2][16];
#define SPT(k, i)                                                         \
  do {                                                                    \
    if (blockIdx.x == 0 && threadIdx.x == 0) g_smpl_pt[k][i] = clock64(); \
  } while (0)
#else
#define SPT(k, i)
#endif

struct FrameParams {
  // model
  const float* Jt; const float* Js; const int32_t* parents; const int32_t* jdepth;
  const int32_t* child_start; const int32_t* child_idx; const int32_t* anc; int nrounds;
  int J, NB, Kfull, Kfull_pad, kf4, depth;   // kf4: Kfull_pad rounded to 4 floats (LDS stride)
  // vertex set
  const float* Pd_v; const float* Pd_k; int Kp; const float* w; const int32_t* idx; const float* Wc;
  int nverts, nchunks, nnz;
  // problem
  int N, n_active, Kc;
  const float* pose; const float* betas; const float* transl;
  // forward outputs
  float* verts; float* joints; float* A_out; float* coeffT; int Npad;
  // "split" form (ha_smpl_forward_split / _backward_split): the first n_head vertices of the set are rows J .. J + n_head - 1 of the
  // joint tensors (row stride jstride = J + n_head per frame), verts / g_verts hold the remaining nverts - n_head
  int n_head, jstride;
  // backward io
  const float* g_verts; const float* g_joints;
  float* g_pose; float* g_betas; float* g_transl;
  // dense backward (ha_smpl_backward_dense): the vertex phase ran in the streaming / MFMA kernels, this kernel only consumes
  // its results: dL/dA [N][J][12], K-split partials of dL/dcoeff [gco_ks][gco_rows][gco_ld], partial vertex-gradient sums
  const float* gA_in; const float* gco_part; int gco_ks, gco_rows, gco_ld; const float* gtl_part; int gtl_np;
  // "parts" form (ha_smpl_forward_parts / _backward_parts): the pose arrives as root [N,3] (in `pose`) + body [N,(n_active-1)*3] (no
  // cat in front of the kernel), one shape row serves betas_div consecutive frames (no expand copy), only the first gj_rows rows of
  // g_joints exist (row stride gj_stride), the pose gradient leaves as g_pose = root part [N,3] + g_body, and every output gradient
  // may carry an addend (the gradient another consumer of the same input produced: no accumulation launch behind the kernel)
  const float* pose_body; int betas_div; int gj_rows, gj_stride;
  float* g_body;
  const float* add_root; const float* add_body; const float* add_betas; const float* add_transl;
};

// axis-angle of joint j of frame f (J joints per frame in the packed form)
__device__ __forceinline__ const float* pose_of(const FrameParams& p, int f, int j, int J) {
  if (!p.pose_body) return p.pose + ((size_t)f * J + j) * 3;
  return j == 0 ? p.pose + (size_t)f * 3 : p.pose_body + ((size_t)f * (p.n_active - 1) + (j - 1)) * 3;
}
__device__ __forceinline__ const float* betas_of(const FrameParams& p, int f) {
  return p.betas + (size_t)(p.betas_div > 1 ? f / p.betas_div : f) * p.NB;
}

constexpr int FW = 4;  // waves (= frames) per block

// 16 values per lane summed over the wavefront, every lane gets the 16 totals: permlane-swap reduce-scatter + DPP adds on the GPU
// (lane_reduce.h: ~50 instructions against 96 ds_bpermute round trips), a plain butterfly on the host emulator tier
__device__ __forceinline__ void wave_sum16_all(float (&v)[16]) {
#ifdef HA_SIMT_EMU
#pragma unroll
  for (int i = 0; i < 16; ++i) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v[i] += __shfl_xor(v[i], off);
  }
#else
  lr::wave_sum16(v);
#endif
}

// 12 floats at a 16-byte aligned LDS address as three 16-byte accesses
__device__ __forceinline__ void lds_ld12(const float* src, float o[12]) {
  const float4* q = reinterpret_cast<const float4*>(src);
  const float4 a = q[0], b = q[1], c = q[2];
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w; o[8] = c.x; o[9] = c.y; o[10] = c.z; o[11] = c.w;
}
__device__ __forceinline__ void lds_st12(float* dst, const float v[12]) {
  float4* q = reinterpret_cast<float4*>(dst);
  q[0] = make_float4(v[0], v[1], v[2], v[3]);
  q[1] = make_float4(v[4], v[5], v[6], v[7]);
  q[2] = make_float4(v[8], v[9], v[10], v[11]);
}

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
      const float* r = pose_of(p, f, j, p.J);
      const float rr[3] = {r[0], r[1], r[2]};
      rodrigues(rr, s.R);
    }
    // (a block's critical path is a chain of load latencies, not bandwidth: 8 shape coefficients x 3 joint-minor rows of Js in
    // flight per trip)
    const float* js = p.Js + j;
    const float* be = betas_of(p, f);
    float a0 = p.Jt[j * 3], a1 = p.Jt[j * 3 + 1], a2 = p.Jt[j * 3 + 2];
    int l = 0;
    for (; l + 8 <= p.NB; l += 8) {
      float b[8], x[8], y[8], z[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        b[u] = be[l + u];
        x[u] = js[((l + u) * 3 + 0) * kMaxJoints]; y[u] = js[((l + u) * 3 + 1) * kMaxJoints]; z[u] = js[((l + u) * 3 + 2) * kMaxJoints];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { a0 = fmaf(b[u], x[u], a0); a1 = fmaf(b[u], y[u], a1); a2 = fmaf(b[u], z[u], a2); }
    }
    for (; l < p.NB; ++l) {
      const float b = be[l];
      a0 = fmaf(b, js[(l * 3 + 0) * kMaxJoints], a0); a1 = fmaf(b, js[(l * 3 + 1) * kMaxJoints], a1); a2 = fmaf(b, js[(l * 3 + 2) * kMaxJoints], a2);
    }
    s.Jr[0] = a0; s.Jr[1] = a1; s.Jr[2] = a2;
  }
  // coefficient vector: betas | 1 | pose feature (prefix of length Kc is what the kernels iterate over)
  for (int i = lane; i < p.NB; i += 64) coeff[i] = betas_of(p, f)[i];
  if (lane == 0) {
    coeff[p.NB] = 1.0f;
    for (int k = p.Kc; k < ((p.Kc + 7) & ~7) && k < p.kf4; ++k) coeff[k] = 0.0f;   // pad entries read by the MFMA path's last k-pair quad
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
  // kinematic chain by pointer jumping: T_j starts as the joint's local transform [R | t] and absorbs, in round r, the accumulated
  // transform of its 2^r-th ancestor -- ceil(log2(depth + 1)) rounds (4 for SMPL+H) instead of one LDS round trip per tree level (10).
  // Gs is the wave's own: ordering inside the wavefront is all that is needed.
  constexpr int kMaxRounds = 6;          // 2^6 > kMaxJoints - 1
  int ar[kMaxRounds];
#pragma unroll
  for (int r = 0; r < kMaxRounds; ++r) ar[r] = (isj && r < p.nrounds) ? p.anc[r * kMaxJoints + j] : -1;
#pragma unroll
  for (int i = 0; i < 9; ++i) s.G[i] = s.R[i];
#pragma unroll
  for (int c = 0; c < 3; ++c) s.G[9 + c] = s.t[c];
#pragma unroll
  for (int r = 0; r < kMaxRounds; ++r) {
    if (r < p.nrounds) {
      if (isj) lds_st12(Gs + j * 12, s.G);
      wave_sync();
      if (ar[r] >= 0) {
        float Ga[12], Rn[9], tt[3];
        lds_ld12(Gs + ar[r] * 12, Ga);
        mat3_mul(Ga, s.G, Rn);
        mat3_vec(Ga, s.G + 9, tt);
#pragma unroll
        for (int i = 0; i < 9; ++i) s.G[i] = Rn[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) s.G[9 + c] = tt[c] + Ga[9 + c];
      }
      wave_sync();
    }
  }
  if (isj) lds_st12(Gs + j * 12, s.G);
  wave_sync();
}

// blend-shape accumulation for the vertices of `chunk`: v_posed = sum_k coeff[k] * Pd[k], for the FW frames of a block at once:
// every wave streams a quarter of the coefficient range of Pd_v and applies it
// to all FW coefficient vectors (coeff of frame g at smem + g * per_wave), so the block reads the chunk's slice of Pd once instead
// of once per frame (the 1920-frame closure call moved 0.7 GB through the L2s per direction for a 170 KB matrix).  The partial
// sums meet in the block's exchange area [wave][frame][256] and are added per frame in wave order.
constexpr int kXchStride = 256;                          // floats per (wave, frame) slot of the block's exchange area
constexpr int kXchFloats = FW * FW * kXchStride;         // 16 KB behind the per-wave regions
__device__ __forceinline__ void blend_vertex_block(const FrameParams& p, int chunk, int wave, int lane, const float* smem, int per_wave,
                                                   float* xch, float vp[3]) {
  const float* pd = p.Pd_v + (size_t)chunk * p.Kfull * 192 + lane;
  const int kq = (p.Kc + FW - 1) / FW, k0 = wave * kq, k1 = k0 + kq < p.Kc ? k0 + kq : p.Kc;
  float acc[FW][3];
#pragma unroll
  for (int g = 0; g < FW; ++g) acc[g][0] = acc[g][1] = acc[g][2] = 0.f;
#pragma unroll 13
  for (int k = k0; k < k1; ++k) {
    const float* q = pd + (size_t)k * 192;
    const float x = q[0], y = q[64], z = q[128];
#pragma unroll
    for (int g = 0; g < FW; ++g) {
      const float c = smem[g * per_wave + k];
      acc[g][0] = fmaf(c, x, acc[g][0]);
      acc[g][1] = fmaf(c, y, acc[g][1]);
      acc[g][2] = fmaf(c, z, acc[g][2]);
    }
  }
  // every wave posts its partials of all FW frames, one barrier, the frame's wave adds them in wave order (the caller keeps a
  // barrier between this read and the next call's writes)
#pragma unroll
  for (int g = 0; g < FW; ++g) {
    float* dst = xch + (wave * FW + g) * kXchStride + lane;
    dst[0] = acc[g][0]; dst[64] = acc[g][1]; dst[128] = acc[g][2];
  }
  __syncthreads();
  vp[0] = vp[1] = vp[2] = 0.f;
#pragma unroll
  for (int w = 0; w < FW; ++w) {
    const float* src = xch + (w * FW + wave) * kXchStride + lane;
    vp[0] += src[0]; vp[1] += src[64]; vp[2] += src[128];
  }
}

// the lane's skinning entries of `chunk` (up to kSkinReg of them in registers, loaded together)
constexpr int kSkinReg = 4;
struct SkinEntries {
  float w[kSkinReg];
  int j[kSkinReg];
};
__device__ __forceinline__ void load_skin(const FrameParams& p, int chunk, int lane, SkinEntries& e) {
#pragma unroll
  for (int q = 0; q < kSkinReg; ++q) {
    const bool on = q < p.nnz;
    const int qq = on ? q : 0;           // (clamped, not branched: the loads of all entries go out together)
    const float wv = p.w[((size_t)chunk * p.nnz + qq) * 64 + lane];
    const int jv = p.idx[((size_t)chunk * p.nnz + qq) * 64 + lane];
    e.w[q] = on ? wv : 0.f;
    e.j[q] = on ? jv : 0;
  }
}

__device__ __forceinline__ void blend_transform(const FrameParams& p, int chunk, int lane, const SkinEntries& e, const float* As, float T[12]) {
#pragma unroll
  for (int i = 0; i < 12; ++i) T[i] = 0.f;
#pragma unroll
  for (int q = 0; q < kSkinReg; ++q) {
    const float* a = As + e.j[q] * 12;
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = fmaf(e.w[q], a[i], T[i]);
  }
  for (int q = kSkinReg; q < p.nnz; ++q) {
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

  SPT(0, 0);
  JointState s;
  joint_forward(p, f, lane, coeff, Gs, s);
  SPT(0, 1);

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
        for (int c = 0; c < 3; ++c) p.joints[((size_t)f * p.jstride + lane) * 3 + c] = s.G[9 + c] + tl[c];
      }
      if (p.A_out) {
#pragma unroll
        for (int i = 0; i < 12; ++i) p.A_out[((size_t)f * p.J + lane) * 12 + i] = A[i];
      }
    }
  }
  __syncthreads();
  if (p.coeffT && valid) {
    // coefficient matrix for the MFMA blend, [k/4][frame][4]: a lane's operands for two k-pairs are one 16-byte load
    const int nq = (p.Kc + 7) >> 3 << 1;               // quads, rounded to whole k-pair quads (8 coefficients)
    for (int q = lane; q < nq; q += 64) {
      float4 v;
      v.x = 4 * q < p.kf4 ? coeff[4 * q] : 0.f;
      v.y = 4 * q + 1 < p.kf4 ? coeff[4 * q + 1] : 0.f;
      v.z = 4 * q + 2 < p.kf4 ? coeff[4 * q + 2] : 0.f;
      v.w = 4 * q + 3 < p.kf4 ? coeff[4 * q + 3] : 0.f;
      *reinterpret_cast<float4*>(p.coeffT + ((size_t)q * p.Npad + f) * 4) = v;
    }
  }
  if (p.verts) {
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
      float vp[3], T[12];
      SkinEntries sk;
      load_skin(p, chunk, lane, sk);
      SPT(0, 2);
      blend_vertex_block(p, chunk, wave, lane, smem, per_wave, smem + FW * per_wave, vp);
      SPT(0, 3);
      blend_transform(p, chunk, lane, sk, Gs, T);
      const int v = chunk * 64 + lane;
      if (valid && v < p.nverts) {
        float o[3];
        mat3_vec(T, vp, o);
        float* dst = v < p.n_head ? p.joints + ((size_t)f * p.jstride + p.J + v) * 3
                                  : p.verts + ((size_t)f * (p.nverts - p.n_head) + (v - p.n_head)) * 3;
        dst[0] = o[0] + T[9] + tl[0];
        dst[1] = o[1] + T[10] + tl[1];
        dst[2] = o[2] + T[11] + tl[2];
      }
      if (chunk + 1 < p.nchunks) __syncthreads();      // the exchange area is rewritten by the next chunk's blend
    }
  }
  SPT(0, 4);
}

// ---------------------------------------------------------------------------------------------------
// backward
// LDS per wave: coeff[Kfull_pad] | Gs[J*12] | As[J*12] | msg[max(J*16, 64*12)] | gvp[192] | gco[Kfull_pad]; then the block's v_posed
// exchange [FW][192]
// ---------------------------------------------------------------------------------------------------
constexpr int kMaxKM = 8;  // Kfull_pad <= 512 -> at most 8 coefficient gradients per lane

// message area: 16 floats per joint in the reverse chain, 12 per vertex of a chunk in the vertex phase
__host__ __device__ inline int bwd_msg_floats(int J) { return J * 16 > kChunk * 12 ? J * 16 : kChunk * 12; }
__host__ __device__ inline int bwd_wave_floats(int kf4, int J) { return 2 * kf4 + J * 24 + bwd_msg_floats(J) + 192; }

__global__ __launch_bounds__(FW * 64) void smpl_frame_bwd_kernel(FrameParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int J = p.J;
  const int per_wave = bwd_wave_floats(p.kf4, J);
  float* coeff = smem + wave * per_wave;
  float* Gs = coeff + p.kf4;
  float* As = Gs + J * 12;
  float* msg = As + J * 12;
  float* gvp = msg + bwd_msg_floats(J);
  float* gco = gvp + 192;
  int f = blockIdx.x * FW + wave;
  const bool valid = f < p.N;
  if (!valid) f = p.N - 1;
  const bool isj = lane < J;

  // children of the lane's joint for the reverse chain, loaded before anything waits on them
  constexpr int kChildReg = 5;
  int ch_reg[kChildReg], ch_c0 = 0, ch_c1 = 0;
  if (isj) { ch_c0 = p.child_start[lane]; ch_c1 = p.child_start[lane + 1]; }
#pragma unroll
  for (int u = 0; u < kChildReg; ++u) ch_reg[u] = ch_c0 + u < ch_c1 ? p.child_idx[ch_c0 + u] : -1;

  SPT(1, 0);
  JointState s;
  joint_forward(p, f, lane, coeff, Gs, s);
  SPT(1, 1);
  if (isj) {
    float gj[3];
    mat3_vec(s.G, s.Jr, gj);
#pragma unroll
    for (int i = 0; i < 9; ++i) As[lane * 12 + i] = s.G[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) As[lane * 12 + 9 + c] = s.G[9 + c] - gj[c];
  }
  __syncthreads();      // (also: every wave's coefficient vector is complete before the block-wide blend reads it)

  // ---- vertex phase ------------------------------------------------------------------------------
  float gco_reg[kMaxKM];
#pragma unroll
  for (int m = 0; m < kMaxKM; ++m) gco_reg[m] = 0.f;
  const int off_gvp = p.kf4 + J * 24 + bwd_msg_floats(J), off_gco = off_gvp + 192;
  float gco_blk[FW][kMaxKM];      // this wave's rows of the chunk, every frame of the block
#pragma unroll
  for (int g = 0; g < FW; ++g)
#pragma unroll
    for (int m = 0; m < kMaxKM; ++m) gco_blk[g][m] = 0.f;
  float gtl[3] = {0.f, 0.f, 0.f};   // transl gradient partial (vertices of this lane, then joints)
  float gAr[12];                    // dL/dA of the lane's joint
#pragma unroll
  for (int i = 0; i < 12; ++i) gAr[i] = 0.f;
  if (p.g_verts || (p.n_head > 0 && p.g_joints)) {
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
      float vp[3], T[12];
      SkinEntries sk;
      load_skin(p, chunk, lane, sk);
      const int v = chunk * 64 + lane;
      float g[3] = {0.f, 0.f, 0.f};
      if (v < p.nverts) {
        const float* src = v < p.n_head ? ((p.g_joints && (p.gj_rows == 0 || J + v < p.gj_rows)) ? p.g_joints + ((size_t)f * (p.gj_stride ? p.gj_stride : p.jstride) + J + v) * 3 : nullptr)
                                        : (p.g_verts ? p.g_verts + ((size_t)f * (p.nverts - p.n_head) + (v - p.n_head)) * 3 : nullptr);
        if (src) { g[0] = src[0]; g[1] = src[1]; g[2] = src[2]; }
      }
      SPT(1, 2);
      blend_vertex_block(p, chunk, wave, lane, smem, per_wave, smem + FW * per_wave, vp);
      SPT(1, 3);
      blend_transform(p, chunk, lane, sk, As, T);
      gtl[0] += g[0]; gtl[1] += g[1]; gtl[2] += g[2];
      float gv[3];
      mat3_tvec(T, g, gv);          // dL/dv_posed = T_R^T g
      gvp[lane] = gv[0]; gvp[64 + lane] = gv[1]; gvp[128 + lane] = gv[2];
      // dL/dA_j += sum_v w_vj [g (x) v_posed | g]: every lane posts the 12-vector of its vertex, then lane j adds up the chunk's
      // vertices against the dense weight column of its joint (one coalesced 256-byte row of Wc and three broadcast LDS reads per
      // vertex; a fixed order, where LDS float atomics took a third of the kernel)
      {
        float u[12];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          u[a * 3 + 0] = g[a] * vp[0]; u[a * 3 + 1] = g[a] * vp[1]; u[a * 3 + 2] = g[a] * vp[2];
          u[9 + a] = g[a];
        }
        lds_st12(msg + lane * 12, u);      // (the message area is idle until the reverse chain)
        wave_sync();
        const float* wc = p.Wc + (size_t)chunk * kChunk * kMaxJoints + lane;
#pragma unroll 8
        for (int vv = 0; vv < kChunk; ++vv) {
          const float wv = wc[vv * kMaxJoints];
          float uv[12];
          lds_ld12(msg + vv * 12, uv);
#pragma unroll
          for (int i = 0; i < 12; ++i) gAr[i] = fmaf(wv, uv[i], gAr[i]);
        }
      }
      __syncthreads();
      SPT(1, 4);
      // dL/dcoeff[g][k] += sum_{v,c} gvp[g][c][v] * Pd[k][c][v] for the FW frames g of the block: lane = 4 coefficients per 256-range
      // (register m = 4 r + e of lane l holds k = 256 r + 4 l + e: one 16-byte load per row and range of the coefficient-major copy),
      // wave w takes the rows i = 48 w .. 48 w + 47 of the chunk's 192 and applies each to all FW frames, so the block reads the
      // chunk's slice of Pd_k once; the per-wave partials are added per frame after the chunk loop
      {
        const float* base = p.Pd_k + (size_t)chunk * 192 * p.Kp + 4 * lane;
        const int i0 = wave * (192 / FW);
        if (p.Kc <= 256) {
#pragma unroll 12
          for (int i = i0; i < i0 + 192 / FW; ++i) {
            const float4 a = *reinterpret_cast<const float4*>(base + (size_t)i * p.Kp);
#pragma unroll
            for (int g = 0; g < FW; ++g) {
              const float gi = smem[g * per_wave + off_gvp + i];
              gco_blk[g][0] = fmaf(a.x, gi, gco_blk[g][0]); gco_blk[g][1] = fmaf(a.y, gi, gco_blk[g][1]);
              gco_blk[g][2] = fmaf(a.z, gi, gco_blk[g][2]); gco_blk[g][3] = fmaf(a.w, gi, gco_blk[g][3]);
            }
          }
        } else {
#pragma unroll 6
          for (int i = i0; i < i0 + 192 / FW; ++i) {
            const float4 a = *reinterpret_cast<const float4*>(base + (size_t)i * p.Kp);
            const float4 b = *reinterpret_cast<const float4*>(base + (size_t)i * p.Kp + 256);
#pragma unroll
            for (int g = 0; g < FW; ++g) {
              const float gi = smem[g * per_wave + off_gvp + i];
              gco_blk[g][0] = fmaf(a.x, gi, gco_blk[g][0]); gco_blk[g][1] = fmaf(a.y, gi, gco_blk[g][1]);
              gco_blk[g][2] = fmaf(a.z, gi, gco_blk[g][2]); gco_blk[g][3] = fmaf(a.w, gi, gco_blk[g][3]);
              gco_blk[g][4] = fmaf(b.x, gi, gco_blk[g][4]); gco_blk[g][5] = fmaf(b.y, gi, gco_blk[g][5]);
              gco_blk[g][6] = fmaf(b.z, gi, gco_blk[g][6]); gco_blk[g][7] = fmaf(b.w, gi, gco_blk[g][7]);
            }
          }
        }
      }
      __syncthreads();
    }
    SPT(1, 5);
    // per frame: the FW partials in wave order -- through the block's exchange area in one step when the coefficient range fits a
    // slot (Kc <= 256), else in FW rounds (round r: wave w adds its partial of frame (w + r) % FW into that frame's gco)
    if (p.Kc <= kXchStride) {
      float* xch = smem + FW * per_wave;
#pragma unroll
      for (int g = 0; g < FW; ++g)
        *reinterpret_cast<float4*>(xch + (wave * FW + g) * kXchStride + 4 * lane) = make_float4(gco_blk[g][0], gco_blk[g][1], gco_blk[g][2], gco_blk[g][3]);
      __syncthreads();
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int w = 0; w < FW; ++w) {
        const float4 v = *reinterpret_cast<const float4*>(xch + (w * FW + wave) * kXchStride + 4 * lane);
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
      }
      gco_reg[0] = t.x; gco_reg[1] = t.y; gco_reg[2] = t.z; gco_reg[3] = t.w;
    } else {
#pragma unroll
    for (int r = 0; r < FW; ++r) {
      const int gs = (wave + r) % FW;
      float* dst = smem + gs * per_wave + off_gco;
#pragma unroll
      for (int m = 0; m < kMaxKM; ++m) {
        const int k = 256 * (m >> 2) + 4 * lane + (m & 3);
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < FW; ++g)
          if (g == gs) v = gco_blk[g][m];
        if (k < p.Kfull_pad) dst[k] = r == 0 ? v : dst[k] + v;
      }
      __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < kMaxKM; ++m) {
      const int k = 256 * (m >> 2) + 4 * lane + (m & 3);
      if (k < p.Kfull_pad) gco_reg[m] = gco[k];
    }
    }
  }
  if (p.gA_in) {
    if (isj) {
#pragma unroll
      for (int i = 0; i < 12; ++i) gAr[i] = p.gA_in[((size_t)f * J + lane) * 12 + i];
    }
#pragma unroll
    for (int m = 0; m < kMaxKM; ++m) {
      const int k = 256 * (m >> 2) + 4 * lane + (m & 3);
      if (k < p.gco_ld)
        for (int ks = 0; ks < p.gco_ks; ++ks) gco_reg[m] += p.gco_part[((size_t)ks * p.gco_rows + f) * p.gco_ld + k];   // fixed order
    }
    if (lane < p.gtl_np) {
#pragma unroll
      for (int c = 0; c < 3; ++c) gtl[c] += p.gtl_part[((size_t)f * p.gtl_np + lane) * 3 + c];
    }
  }
#pragma unroll
  for (int m = 0; m < kMaxKM; ++m) {
    const int k = 256 * (m >> 2) + 4 * lane + (m & 3);
    if (k < p.Kfull_pad) gco[k] = (k < p.Kc) ? gco_reg[m] : 0.f;
  }
  __syncthreads();

  SPT(1, 6);
  // ---- chain backward ----------------------------------------------------------------------------
  // gG = dL/dG_j (world transform): from A_j = [G.R | G.t - G.R Jr] and posed joint = G.t
  float gG[12], gJr[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 12; ++i) gG[i] = 0.f;
  if (isj) {
    float gAj[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) gAj[i] = gAr[i];
    float gjt[3] = {0.f, 0.f, 0.f};
    if (p.g_joints && (p.gj_rows == 0 || lane < p.gj_rows)) {
#pragma unroll
      for (int c = 0; c < 3; ++c) gjt[c] = p.g_joints[((size_t)f * (p.gj_stride ? p.gj_stride : p.jstride) + lane) * 3 + c];
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
  // With G4 the 4x4 world transform of a joint, the chain G_child = G_parent L_child gives dL/dG_p = sum over the subtree of
  // own_d (G_p^-1 G_d)^T = [sum_d own_d G4_d^T] G4_p^-T: plain 12-vector subtree sums (one LDS hand-off per tree level, no matrix work
  // in the dependent chain), then every joint finishes on its own.
  float gR[9], gt[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 9; ++i) gR[i] = 0.f;
  {
    float S[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) S[i] = 0.f;
    const float own_t[3] = {gG[9], gG[10], gG[11]};
    if (isj) {
      mat3_mult(gG, s.G, S);                        // own.R R_w^T
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int b = 0; b < 3; ++b) S[a * 3 + b] += gG[9 + a] * s.G[9 + b];   // + own.t (x) t_w
        S[9 + a] = gG[9 + a];
      }
    }
    for (int lvl = p.depth; lvl >= 1; --lvl) {
      if (isj && s.depth == lvl) lds_st12(msg + lane * 12, S);
      wave_sync();        // the wave's own sums; a parent's slot is not one its children read, so one ordering point per level
      if (isj && s.depth == lvl - 1) {
        auto take = [&](int ch) {
          float c[12];
          lds_ld12(msg + ch * 12, c);
#pragma unroll
          for (int i = 0; i < 12; ++i) S[i] += c[i];
        };
#pragma unroll
        for (int u = 0; u < kChildReg; ++u)
          if (ch_reg[u] >= 0) take(ch_reg[u]);
        for (int ci = ch_c0 + kChildReg; ci < ch_c1; ++ci) take(p.child_idx[ci]);
      }
    }
    if (isj) {
      // X = S G4^-T: X.R = S.R R_w - S.t (x) (R_w^T t_w), X.t = S.t
      float w[3], XR[9];
      mat3_tvec(s.G, s.G + 9, w);
      mat3_mul(S, s.G, XR);
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) XR[a * 3 + b] -= S[9 + a] * w[b];
      const float Xt[3] = {S[9], S[10], S[11]};
      if (s.parent >= 0) {
        float Gp[12];
        lds_ld12(Gs + s.parent * 12, Gp);
        mat3_tmul(Gp, XR, gR);                      // gR = Gp.R^T X.R
        mat3_tvec(Gp, Xt, gt);                      // gt = Gp.R^T X.t
      } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) gR[i] = XR[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) gt[c] = Xt[c];
      }
      // rest joint: + own translation gradient, - the children's (sum_c gt_c = R_w^T (S.t - own.t))
      const float dt[3] = {S[9] - own_t[0], S[10] - own_t[1], S[11] - own_t[2]};
      float q[3];
      mat3_tvec(s.G, dt, q);
#pragma unroll
      for (int c = 0; c < 3; ++c) gJr[c] += gt[c] - q[c];
    }
  }

  SPT(1, 7);
  // ---- pose gradient -------------------------------------------------------------------------------
  if (isj && valid && p.g_pose) {
    float gr[3] = {0.f, 0.f, 0.f};
    if (lane < p.n_active) {
      if (lane >= 1) {
#pragma unroll
        for (int i = 0; i < 9; ++i) gR[i] += gco[p.NB + 1 + (lane - 1) * 9 + i];
      }
      const float* r = pose_of(p, f, lane, J);
      const float rr[3] = {r[0], r[1], r[2]};
      rodrigues_bwd(rr, gR, gr);
    }
    if (!p.pose_body) {
#pragma unroll
      for (int c = 0; c < 3; ++c) p.g_pose[((size_t)f * J + lane) * 3 + c] = gr[c];
    } else if (lane == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) p.g_pose[(size_t)f * 3 + c] = gr[c] + (p.add_root ? p.add_root[(size_t)f * 3 + c] : 0.f);
    } else if (lane < p.n_active && p.g_body) {
      const size_t o = ((size_t)f * (p.n_active - 1) + (lane - 1)) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) p.g_body[o + c] = gr[c] + (p.add_body ? p.add_body[o + c] : 0.f);
    }
  }
  SPT(1, 8);
  // ---- betas gradient: coefficient part + rest-joint part ------------------------------------------
  if (p.g_betas) {
    // rest-joint part: sum over joints (lanes) of gJr . Js[joint][:, l]; 16 shape coefficients per trip (48 loads in flight, one
    // 16-value wave sum)
    const float* js = p.Js + (isj ? lane : 0);
    for (int l0 = 0; l0 < p.NB; l0 += 16) {
      float part[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int l = l0 + u;
        part[u] = 0.f;
        if (isj && l < p.NB)
          part[u] = fmaf(gJr[0], js[(l * 3 + 0) * kMaxJoints], fmaf(gJr[1], js[(l * 3 + 1) * kMaxJoints], gJr[2] * js[(l * 3 + 2) * kMaxJoints]));
      }
      wave_sum16_all(part);
      float mine = part[0];
#pragma unroll
      for (int u = 1; u < 16; ++u)
        if (lane == u) mine = part[u];
      if (lane < 16 && l0 + lane < p.NB && valid)
        p.g_betas[(size_t)f * p.NB + l0 + lane] = gco[l0 + lane] + mine + (p.add_betas ? p.add_betas[(size_t)f * p.NB + l0 + lane] : 0.f);
    }
  }
  SPT(1, 9);
  // ---- transl gradient: wave reduction -------------------------------------------------------------
  if (p.g_transl) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float vsum = gtl[c];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) vsum += __shfl_xor(vsum, off);
      if (lane == 0 && valid) p.g_transl[(size_t)f * 3 + c] = vsum + (p.add_transl ? p.add_transl[(size_t)f * 3 + c] : 0.f);
    }
  }
  SPT(1, 10);
}

// ===================================================================================================
// dense path: fp32 MFMA pose-blend GEMM
// ===================================================================================================
typedef float f32x16 __attribute__((ext_vector_type(16)));

// grid: 1-D, block = 4 waves = 4 vertex tiles (32 vertices each) x 64 frames.
// Blocks that share a vertex-tile group (the B panel, ~300 KB) are mapped to the same XCD (b % 8).
// Operands arrive as 16-byte loads: per four k-pairs a lane issues 4 loads of coefficients (frames l&31 and 32 + l&31, two
// coefficient quads) and 3 loads of the blend matrix, against 24 MFMAs.
__global__ __launch_bounds__(256) void pose_blend_mfma_kernel(const float* __restrict__ coeffQ, int Npad, int KQ, int KQfull,
                                                              const float* __restrict__ Pd_m, float* __restrict__ v_posed,
                                                              int N, int V, int n_vt, int n_ft) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x;
  const int vtg = (b & 7) + 8 * (b / (8 * n_ft));
  const int ft = (b >> 3) % n_ft;
  const int vt = vtg * 4 + wave;
  if (vt >= n_vt) return;
  const int f0 = ft * 64;
  const bool hi = lane >= 32;
  const float4* a_ptr = reinterpret_cast<const float4*>(coeffQ) + f0 + (lane & 31);
  const float4* b_ptr = reinterpret_cast<const float4*>(Pd_m) + (size_t)vt * KQfull * 192 + lane;
  f32x16 acc[2][3];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[r][c][i] = 0.f;
#pragma unroll 2
  for (int kq = 0; kq < KQ; ++kq) {
    // coefficient quads 2 kq and 2 kq + 1 = k-pairs 4 kq .. 4 kq + 3
    const float4 a0q0 = a_ptr[(size_t)(2 * kq) * Npad], a1q0 = a_ptr[(size_t)(2 * kq) * Npad + 32];
    const float4 a0q1 = a_ptr[(size_t)(2 * kq + 1) * Npad], a1q1 = a_ptr[(size_t)(2 * kq + 1) * Npad + 32];
    const float4 b0 = b_ptr[(size_t)kq * 192], b1 = b_ptr[(size_t)kq * 192 + 64], b2 = b_ptr[(size_t)kq * 192 + 128];
    const float a0[4] = {hi ? a0q0.y : a0q0.x, hi ? a0q0.w : a0q0.z, hi ? a0q1.y : a0q1.x, hi ? a0q1.w : a0q1.z};
    const float a1[4] = {hi ? a1q0.y : a1q0.x, hi ? a1q0.w : a1q0.z, hi ? a1q1.y : a1q1.x, hi ? a1q1.w : a1q1.z};
    const float bb[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], bb[3 * j], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], bb[3 * j + 1], acc[0][1], 0, 0, 0);
      acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], bb[3 * j + 2], acc[0][2], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], bb[3 * j], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], bb[3 * j + 1], acc[1][1], 0, 0, 0);
      acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], bb[3 * j + 2], acc[1][2], 0, 0, 0);
    }
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
// Flat partition of the [N*V] vertex index space: a block of NW waves owns 256*NW consecutive global vertices, i.e. a
// 16-byte aligned window of both arrays that spans at most two frames (V >= 256*NW).
// ===================================================================================================
constexpr int kSkinMinVerts = 1024;     // smallest block window (4 waves)
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef float vf2 __attribute__((ext_vector_type(2)));

// One vertex: T = sum_q w_q A_q (12 floats), out = T [v; 1] + transl.  The blend runs as packed fp32 FMAs (v_pk_fma_f32:
// two IEEE fmaf per lane per instruction, bit-identical to the scalar chain) -- at 7 TB/s the VALU, not only the memory
// system, is on the critical path of this kernel (~110 scalar VALU ops per vertex before packing).
__device__ __forceinline__ void skin_one(const float* s_A, int J, int fr, float4 wv, uint32_t id, float x, float y, float z,
                                         float tx, float ty, float tz, float& ox, float& oy, float& oz) {
  const vf4* Af = reinterpret_cast<const vf4*>(s_A + fr * J * 12);
  const float wq[4] = {wv.x, wv.y, wv.z, wv.w};
  vf2 r[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) r[i] = vf2{0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int jq = (id >> (8 * q)) & 0xff;
    const vf4 a0 = Af[jq * 3], a1 = Af[jq * 3 + 1], a2 = Af[jq * 3 + 2];
    const vf2 w2 = {wq[q], wq[q]};
    r[0] = __builtin_elementwise_fma(w2, a0.lo, r[0]);
    r[1] = __builtin_elementwise_fma(w2, a0.hi, r[1]);
    r[2] = __builtin_elementwise_fma(w2, a1.lo, r[2]);
    r[3] = __builtin_elementwise_fma(w2, a1.hi, r[3]);
    r[4] = __builtin_elementwise_fma(w2, a2.lo, r[4]);
    r[5] = __builtin_elementwise_fma(w2, a2.hi, r[5]);
  }
  // T = [R00 R01 | R02 R10 | R11 R12 | R20 R21 | R22 t0 | t1 t2]
  ox = fmaf(r[0].x, x, fmaf(r[0].y, y, fmaf(r[1].x, z, r[4].y))) + tx;
  oy = fmaf(r[1].y, x, fmaf(r[2].x, y, fmaf(r[2].y, z, r[5].x))) + ty;
  oz = fmaf(r[3].x, x, fmaf(r[3].y, y, fmaf(r[4].x, z, r[5].y))) + tz;
}

// "Wave-sliced" streaming kernel: every global access is lane-contiguous (each 16-byte wave instruction covers 1 KiB), which is
// what the HBM/Infinity-Cache path wants (tools/microbench/hbm_stream.hip: 7.0-7.2 TB/s for this geometry with
// non-temporal stores against 4-5.5 TB/s for per-thread 48-byte runs).  The [vertex][xyz] <-> 16-byte-vector transposition
// goes through a PRIVATE 3 KiB LDS slice per wave (256 vertices), so the only block barrier is the early one that
// publishes the <=2 frames' A matrices; the waves of a block then drift apart and overlap their load / skin / store phases.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// INTERIOR: the whole block window lies inside [0, N*V) (every block but the last) -- all guards fold away and the
// addressing is a block-uniform base pointer plus 32-bit lane offsets (the 64-bit per-lane form costs ~100 VALU ops).
template <bool NT, int DIAG, int NW, bool INTERIOR>
__device__ __forceinline__ void skin_wave_body(const float* __restrict__ v_posed, const float* __restrict__ A,
                                               const float* __restrict__ transl, const float4* __restrict__ w4,
                                               const uint32_t* __restrict__ idx4, float* __restrict__ verts, int N, int V, int J,
                                               float* smem) {
  float* s_A = smem;                                   // 2 frames x J x 12
  const int tid = threadIdx.x, lane = tid & 63, wv_id = tid >> 6;
  float* s_v = smem + 2 * J * 12 + wv_id * 768;        // this wave's 256 vertices
  const long long gv0 = (long long)blockIdx.x * (256 * NW);
  const int n0 = (int)(gv0 / V);
  const int rel0 = (int)(gv0 - (long long)n0 * V);     // window start within frame n0
  const long long rem3 = ((long long)N * V - gv0) * 3; // floats from the window start to the end of the arrays
  const bool two = rel0 + 256 * NW > V && n0 + 1 < N;  // does the window reach into frame n0 + 1?
  // A first (the barrier below waits for these only), then the streaming loads
  if (DIAG != 1) {
    const int per = J * 3;
    const float4* A4 = reinterpret_cast<const float4*>(A) + (size_t)n0 * per;
    float4* sA4 = reinterpret_cast<float4*>(s_A);
    const int cnt = two ? 2 * per : per;               // the two frames' A are contiguous in memory
    for (int i = tid; i < cnt; i += 64 * NW) sA4[i] = A4[i];
  }
  const float* src = v_posed + gv0 * 3;                // block-uniform bases (16-byte aligned: 256 * NW * 3 floats per window)
  float* dst = verts + gv0 * 3;
  const int f0 = (wv_id * 192 + lane) * 4;             // this lane's first float within the window
  vf4 q[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int fl = f0 + 256 * k;
    if (INTERIOR || fl + 3 < rem3) {
      q[k] = *reinterpret_cast<const vf4*>(src + fl);
    } else {
      q[k] = vf4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (fl + e < rem3) q[k][e] = src[fl + e];
    }
  }
  float4 wgt[4];
  uint32_t id[4];
  int frs[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int rel = rel0 + wv_id * 256 + lane + 64 * k;
    const int fr = rel >= V ? 1 : 0;
    int v = rel - (fr ? V : 0);
    if (!INTERIOR && n0 + fr >= N) v = 0;
    frs[k] = fr;
    wgt[k] = w4[v];
    id[k] = DIAG == 2 ? 0u : idx4[v];
  }
  float tl[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  if (transl) {
#pragma unroll
    for (int fr = 0; fr < 2; ++fr)
      if (fr == 0 || two) {
#pragma unroll
        for (int c = 0; c < 3; ++c) tl[fr][c] = transl[(size_t)(n0 + fr) * 3 + c];
      }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 3; ++k) *reinterpret_cast<vf4*>(s_v + (lane + 64 * k) * 4) = q[k];
  wave_lds_sync();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float* pv = s_v + (lane + 64 * k) * 3;
    const int fr = frs[k];
    float ox, oy, oz;
    if (DIAG == 1) { ox = pv[0] + wgt[k].x; oy = pv[1] + wgt[k].y; oz = pv[2] + tl[fr][2]; }
    else skin_one(s_A, J, fr, wgt[k], id[k], pv[0], pv[1], pv[2], tl[fr][0], tl[fr][1], tl[fr][2], ox, oy, oz);
    pv[0] = ox; pv[1] = oy; pv[2] = oz;
  }
  wave_lds_sync();
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int fl = f0 + 256 * k;
    const vf4 o = *reinterpret_cast<const vf4*>(s_v + (lane + 64 * k) * 4);
    if (INTERIOR || fl + 3 < rem3) {
      if (NT) __builtin_nontemporal_store(o, reinterpret_cast<vf4*>(dst + fl));
      else *reinterpret_cast<vf4*>(dst + fl) = o;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (fl + e < rem3) dst[fl + e] = o[e];
    }
  }
}

template <bool NT, int DIAG, int NW>
__global__ __launch_bounds__(64 * NW) void lbs_skin_wave_kernel(const float* __restrict__ v_posed, const float* __restrict__ A,
                                                                const float* __restrict__ transl, const float4* __restrict__ w4,
                                                                const uint32_t* __restrict__ idx4, float* __restrict__ verts,
                                                                int N, int V, int J) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((long long)(blockIdx.x + 1) * (256 * NW) <= (long long)N * V)
    skin_wave_body<NT, DIAG, NW, true>(v_posed, A, transl, w4, idx4, verts, N, V, J, smem);
  else
    skin_wave_body<NT, DIAG, NW, false>(v_posed, A, transl, w4, idx4, verts, N, V, J, smem);
}

// ===================================================================================================
// dense path, forward-only callers: pose-blend GEMM with the skinning in its epilogue (ha_smpl_forward algo 3; round 4).
// v_posed is never materialised: 2 x 159 MB of HBM traffic less per 1920-frame call, and the skinning's LDS gather (192 B of bone
// matrices per vertex) runs under another wave's MFMA chain instead of under an HBM stream.
//   * round 3's attempt staged A for 64 frames x 128 vertices per block (160 KB of A for 98 KB of output) and lost.  Here a block owns
//     ONE tile of 32 frames -- their A matrices (78 KB) are loaded once and stay in LDS -- and walks over 27 vertex tiles (864 vertices
//     per block, 6912 per frame tile and XCD): A traffic is 3 % of the output;
//   * block b runs on XCD b % 8 and owns vertex-tile group b % 8: an XCD's share of the blend matrix (2.1 MB) stays in its 4 MB L2;
//   * two blocks per CU (2 x 80 384 B of LDS): one wave's epilogue (16 skin_one per lane: LDS + VALU) overlaps the other's MFMAs;
//   * accumulator register i of lane l is vertex l & 31, frame slot (i & 3) + 8 (i >> 2) + 4 (l >> 5); x / y / z are the three column
//     tiles, so the epilogue is skin_one on registers (bit-identical to blend + lbs_skin) and lanes 0..31 store 384 contiguous bytes.
// Measured (round 4, tools/dense_fwd_timing.py + rocprofv3): N = 1920: 240 us against 184 (blend) + 54 (lbs_skin) -- a draw, the epilogues of two
// waves per SIMD do not hide completely; N = 30720: 3.32 ms against 3.52 (+6 %).  An 8-waves-per-block form (one tile per wave, 128
// registers) spills.  BodyModel therefore takes this path for forward-only calls of >= 4096 frames.
// ===================================================================================================
constexpr int kFusedTilesPerBlock = 27;      // vertex tiles of 32 per block (8 groups cover Vpad = 6912)

template <int NWB, bool PAIR>
__global__ __launch_bounds__(64 * NWB) HA_WAVES_PER_EU(NWB / 2, NWB / 2) void pose_blend_skin_kernel(
    const float* __restrict__ coeffQ, int Npad, int KQ, int KQfull, const float* __restrict__ Pd_m, int N, int V, int n_vt, int n_groups,
    const float* __restrict__ A, const float* __restrict__ transl, const float4* __restrict__ w4, const uint32_t* __restrict__ idx4,
    float* __restrict__ verts, int J) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // A of the tile's 32 frames [32][J * 12] | transl [32][4]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int grp = blockIdx.x % n_groups, ft = blockIdx.x / n_groups;
  const int f0 = ft * 32;
  const int fstride = J * 12, q_per = J * 3;                       // floats / 16-byte quads per frame
  float* s_T = smem + 32 * fstride;
  // ---- the frame tile's A matrices and translations -> LDS (frames beyond N: the last frame's, never stored) ----
  for (int e = threadIdx.x; e < 32 * q_per; e += 64 * NWB) {
    const int slot = e / q_per, q = e - slot * q_per;
    const int f = f0 + slot < N ? f0 + slot : N - 1;
    reinterpret_cast<vf4*>(smem + slot * fstride)[q] = reinterpret_cast<const vf4*>(A + (size_t)f * fstride)[q];
  }
  if (threadIdx.x < 96) {
    const int slot = threadIdx.x / 3, c = threadIdx.x - 3 * slot;
    const int f = f0 + slot < N ? f0 + slot : N - 1;
    s_T[slot * 4 + c] = transl ? transl[(size_t)f * 3 + c] : 0.f;
  }
  __syncthreads();
  const bool hi = lane >= 32;
  const int hl = lane >> 5;
  const float4* a_ptr = reinterpret_cast<const float4*>(coeffQ) + f0 + (lane & 31);
  // A wave works on TWO vertex tiles at a time (tiles vt and vt + 4: 96 accumulators): per four k-pairs 2 coefficient loads and 6
  // blend-matrix loads feed 24 MFMAs -- the operand-to-MFMA ratio of the unfused kernel (one tile: 5 loads per 12 MFMAs, measured
  // at 45 % of the fp32 MFMA rate).  kFusedTilesPerBlock = 27 = 3 x 8 + 3: three pairs per wave, the last three tiles single.
  const int vt_end = (grp + 1) * kFusedTilesPerBlock < n_vt ? (grp + 1) * kFusedTilesPerBlock : n_vt;
  auto epilogue = [&](int vt, const f32x16 (&acc)[3]) {
    const int v = vt * 32 + (lane & 31);
    const bool vok = v < V;
    const float4 wv = vok ? w4[v] : float4{0.f, 0.f, 0.f, 0.f};
    const uint32_t id = vok ? idx4[v] : 0u;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int slot = (i & 3) + 8 * (i >> 2) + 4 * hl;
      float ox, oy, oz;
      skin_one(smem, J, slot, wv, id, acc[0][i], acc[1][i], acc[2][i], s_T[slot * 4], s_T[slot * 4 + 1], s_T[slot * 4 + 2], ox, oy, oz);
      if (vok && f0 + slot < N) {
        float* dst = verts + ((size_t)(f0 + slot) * V + v) * 3;
        dst[0] = ox; dst[1] = oy; dst[2] = oz;
      }
      if ((i & 1) == 1) HA_SCHED_FENCE();          // (two frames' gathers in flight, not sixteen: the accumulators hold the registers)
    }
  };
  for (int vt = grp * kFusedTilesPerBlock + wave; vt < vt_end; vt += (PAIR ? 2 : 1) * NWB) {
    const bool pair = PAIR && vt + NWB < vt_end;                     // (wave-uniform)
    const float4* b_ptr = reinterpret_cast<const float4*>(Pd_m) + (size_t)vt * KQfull * 192 + lane;
    const size_t b_next = (size_t)NWB * KQfull * 192;                // float4 offset of tile vt + NWB
    f32x16 acc[PAIR ? 2 : 1][3];
#pragma unroll
    for (int t2 = 0; t2 < (PAIR ? 2 : 1); ++t2)
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t2][c][i] = 0.f;
    if constexpr (PAIR) if (pair) {
#pragma unroll 2
      for (int kq = 0; kq < KQ; ++kq) {
        // coefficient quads 2 kq and 2 kq + 1 = k-pairs 4 kq .. 4 kq + 3
        const float4 aq0 = a_ptr[(size_t)(2 * kq) * Npad], aq1 = a_ptr[(size_t)(2 * kq + 1) * Npad];
        const float4 b0 = b_ptr[(size_t)kq * 192], b1 = b_ptr[(size_t)kq * 192 + 64], b2 = b_ptr[(size_t)kq * 192 + 128];
        const float4 c0 = b_ptr[b_next + (size_t)kq * 192], c1 = b_ptr[b_next + (size_t)kq * 192 + 64], c2 = b_ptr[b_next + (size_t)kq * 192 + 128];
        const float a0[4] = {hi ? aq0.y : aq0.x, hi ? aq0.w : aq0.z, hi ? aq1.y : aq1.x, hi ? aq1.w : aq1.z};
        const float bb[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
        const float cc[12] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], bb[3 * j], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], bb[3 * j + 1], acc[0][1], 0, 0, 0);
          acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], bb[3 * j + 2], acc[0][2], 0, 0, 0);
          acc[PAIR ? 1 : 0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], cc[3 * j], acc[PAIR ? 1 : 0][0], 0, 0, 0);
          acc[PAIR ? 1 : 0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], cc[3 * j + 1], acc[PAIR ? 1 : 0][1], 0, 0, 0);
          acc[PAIR ? 1 : 0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], cc[3 * j + 2], acc[PAIR ? 1 : 0][2], 0, 0, 0);
        }
      }
      epilogue(vt, acc[0]);
      epilogue(vt + NWB, acc[PAIR ? 1 : 0]);
      continue;
    }
    {
#pragma unroll 2
      for (int kq = 0; kq < KQ; ++kq) {
        const float4 aq0 = a_ptr[(size_t)(2 * kq) * Npad], aq1 = a_ptr[(size_t)(2 * kq + 1) * Npad];
        const float4 b0 = b_ptr[(size_t)kq * 192], b1 = b_ptr[(size_t)kq * 192 + 64], b2 = b_ptr[(size_t)kq * 192 + 128];
        const float a0[4] = {hi ? aq0.y : aq0.x, hi ? aq0.w : aq0.z, hi ? aq1.y : aq1.x, hi ? aq1.w : aq1.z};
        const float bb[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], bb[3 * j], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], bb[3 * j + 1], acc[0][1], 0, 0, 0);
          acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], bb[3 * j + 2], acc[0][2], 0, 0, 0);
        }
      }
      epilogue(vt, acc[0]);
    }
  }
}

// ===================================================================================================
// dense backward (all V vertices carry a gradient: point-cloud / chamfer terms).  The wave-per-frame adjoint re-blends every
// vertex and walks the blend matrix twice per frame (2 x 18 MB of L2 traffic per frame at V = 6890); here the vertex phase
// is three batched kernels over all frames, and smpl_frame_bwd_kernel only finishes the kinematic chain:
//   (1) dense_gvp_kernel   dL/dv_posed = T_R^T g (streaming, HBM-bound), written K-major for the GEMM below
//   (2) dense_gA_kernel    dL/dA_j = sum_v w_vj [g_v (x) v_posed_v | g_v]: per frame a (12 x V) x (V x 64) product with the dense
//                          skinning-weight matrix, v_mfma_f32_16x16x4_f32, the A operand formed on the fly from g and v_posed
//   (3) dense_gco_kernel   dL/dcoeff [N, Kc] = dL/dv_posed [N, 3 V] x Pd^T: v_mfma_f32_32x32x2_f32, K split over waves
// ===================================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kGvpChunksPerBlock = 27;      // 4 blocks per frame at V = 6890 (108 chunks)

// grid (chunk groups, N), 4 waves; wave w of block y takes chunks y * 27 + w, + 4, ...
__global__ __launch_bounds__(256) void dense_gvp_kernel(const float* __restrict__ g_verts, const float* __restrict__ A,
                                                        const float4* __restrict__ w4, const uint32_t* __restrict__ idx4,
                                                        float* __restrict__ gvpT, float* __restrict__ gtl_part, int V, int J,
                                                        int nchunks, int np) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int f = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < J * 12; i += 256) smem[i] = A[(size_t)f * J * 12 + i];
  __syncthreads();
  float gs[3] = {0.f, 0.f, 0.f};
  const int c0 = blockIdx.x * kGvpChunksPerBlock;
  const int c1 = c0 + kGvpChunksPerBlock < nchunks ? c0 + kGvpChunksPerBlock : nchunks;
  for (int ch = c0 + wave; ch < c1; ch += 4) {
    const int v = ch * 64 + lane;
    float g[3] = {0.f, 0.f, 0.f}, gv[3] = {0.f, 0.f, 0.f};
    if (v < V) {
      const float* src = g_verts + ((size_t)f * V + v) * 3;
      g[0] = src[0]; g[1] = src[1]; g[2] = src[2];
      const float4 wv = w4[v];
      const uint32_t id = idx4[v];
      const float wq[4] = {wv.x, wv.y, wv.z, wv.w};
      float T[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) T[i] = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float* a = smem + ((id >> (8 * q)) & 0xff) * 12;
#pragma unroll
        for (int i = 0; i < 9; ++i) T[i] = fmaf(wq[q], a[i], T[i]);
      }
      mat3_tvec(T, g, gv);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      gs[c] += g[c];
      gvpT[((size_t)f * nchunks + ch) * 192 + c * 64 + lane] = gv[c];
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float vsum = gs[c];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) vsum += __shfl_xor(vsum, off);
    if (lane == 0) gtl_part[((size_t)f * np + blockIdx.x * 4 + wave) * 3 + c] = vsum;
  }
}

// One block per frame; wave w accumulates the 16-vertex groups w, w + 4, ...  Rows of the MFMA tile: (a, b) = (row >> 2, row & 3),
// a < 3 the gradient component, b < 3 the v_posed component and b = 3 the translation column; columns: joint 4 (l & 15) + tile.
// dL/dA_j = sum over the joint's vertex list of w [g (x) v_posed | g]: the weights are 4-sparse (27 560 of 6890 x 52 entries for SMPL+H), the
// dense product above spends 16x the arithmetic and re-reads the 1.7 MB weight operand per frame.  One block per frame; a wave takes
// the joints order[w], order[w + 4], ... (longest lists first, dealt round-robin); its lanes stride over the joint's entries (sorted
// by vertex: neighbouring lanes read neighbouring 12-byte records of the frame's g / v_posed, L2-resident after the first touch), then
// a fixed butterfly over the lanes.
__global__ __launch_bounds__(256) void sparse_gA_kernel(const float* __restrict__ g_verts, const float* __restrict__ v_posed,
                                                        const int32_t* __restrict__ jstart, const int32_t* __restrict__ jv,
                                                        const float* __restrict__ jw, const int32_t* __restrict__ jorder,
                                                        float* __restrict__ gA_out, int V, int J) {
  const int f = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* gf = g_verts + (size_t)f * V * 3;
  const float* vf = v_posed + (size_t)f * V * 3;
  for (int jo = wave; jo < J; jo += 4) {
    const int j = jorder[jo];
    const int e0 = jstart[j], e1 = jstart[j + 1];
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.f;
#pragma unroll 4
    for (int e = e0 + lane; e < e1; e += 64) {
      const int v = jv[e];
      const float w = jw[e];
      const float g0 = gf[(size_t)v * 3], g1 = gf[(size_t)v * 3 + 1], g2 = gf[(size_t)v * 3 + 2];
      const float p0 = vf[(size_t)v * 3], p1 = vf[(size_t)v * 3 + 1], p2 = vf[(size_t)v * 3 + 2];
      const float wg[3] = {w * g0, w * g1, w * g2};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        acc[a * 3 + 0] = fmaf(wg[a], p0, acc[a * 3 + 0]);
        acc[a * 3 + 1] = fmaf(wg[a], p1, acc[a * 3 + 1]);
        acc[a * 3 + 2] = fmaf(wg[a], p2, acc[a * 3 + 2]);
        acc[9 + a] += wg[a];
      }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) acc[i] += __shfl_xor(acc[i], off);
    }
    if (lane < 12) {
      float mine = acc[0];
#pragma unroll
      for (int i = 1; i < 12; ++i)
        if (lane == i) mine = acc[i];
      gA_out[((size_t)f * J + j) * 12 + lane] = mine;
    }
  }
}

// Default since round 5 (ha_tune_set("dense_gA_sparse", 2); measured 205 us against 255 us for the joint lists at N = 1920, 8.5 against
// 9.2 ms for the whole dense backward at N = 30 720, profiles/r05_dense_bwd.txt): the dense product below
// restricted, per 64-vertex chunk, to the joints the chunk touches, 16 slots per MFMA tile (one tile for 105 of the 108 chunks of the
// synthetic model, two for the rest) instead of four tiles over all 64 joint columns, an 8 KB weight operand per chunk instead of 16 KB,
// every vertex record read once.  A wave takes the chunks w, w + 4, ...; after a tile's 16 MFMAs lane (a = l >> 4, slot = l & 15) adds
// its four values into the wave's own [joint][12] table in LDS (two slots of a chunk never share a joint: no conflicts, fixed order).
// Round 5, late: the first form read its operands straight from global memory -- per chunk 32 dword loads of which a wave used 48 bytes
// each (lane (k-quarter, a, b) wants g[v][a] and p[v][b]: 12 distinct floats per instruction) plus 16 strided dword loads of the weights:
// 48 load instructions per 16 MFMAs, the kernel sat on the CU's address path (205 us against a 63 us HBM bound).  Now a chunk's 192 + 192
// floats arrive with four 8-byte loads per lane (issued two chunks ahead, pinned behind the chunk's weight loads), go through the wave's LDS slice, and the weights are packed
// [chunk][group][lane][16] so that a lane's 16 B operands are four 16-byte loads: 8 load instructions per chunk.
constexpr int GA_STAGE = 2 * 192;      // floats of LDS per wave: the chunk's g and p records
template <bool PAIR>      // PAIR: V is even, a frame's records are 8-byte aligned
__global__ __launch_bounds__(256) void compressed_gA_kernel(const float* __restrict__ g_verts, const float* __restrict__ v_posed,
                                                            const int32_t* __restrict__ gcj, const float* __restrict__ gcw,
                                                            const int32_t* __restrict__ gng, float* __restrict__ gA_out, int V, int Vpad, int J) {
  extern __shared__ __attribute__((aligned(16))) float smem[];      // [4 waves][64 joints][12] | [4 waves][GA_STAGE]
  const int f = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int row = lane & 15, kq = lane >> 4, a = row >> 2, b = row & 3, col = lane & 15;
  const float* gf = g_verts + (size_t)f * V * 3;
  const float* vf = v_posed + (size_t)f * V * 3;
  float* tab = smem + wave * 768;
  float* stg = smem + 4 * 768 + wave * GA_STAGE;
  for (int i = lane; i < 768; i += 64) tab[i] = 0.f;
  const int nch = Vpad / 64, nfl = V * 3;
  // a chunk's records: floats [192 c, 192 c + 192) of the frame, two per lane and load (8-byte aligned whenever V is even; odd V: scalar
  // loads).  Branch-free: every lane loads from a clamped address, elements beyond the frame are zeroed by a select.
  // (the loaded value is not touched here -- a select right behind the load would make the wave wait for it on the spot; `keep` zeroes
  // it when the chunk goes to LDS, one iteration later)
  auto ld2 = [&](const float* src, int e) -> vf2 {
    vf2 u;
    if constexpr (PAIR) {
      u = *reinterpret_cast<const vf2*>(src + (e < nfl ? e : nfl - 2));
    } else {
      u[0] = src[e < nfl ? e : nfl - 1];
      u[1] = src[e + 1 < nfl ? e + 1 : nfl - 1];
    }
    return u;
  };
  auto keep = [&](vf2 u, int e) -> vf2 { return vf2{e < nfl ? u[0] : 0.f, e + 1 < nfl ? u[1] : 0.f}; };
  auto fetch = [&](int c, vf2 (&r)[4]) {
    const int e0 = 192 * c + 2 * lane, e1 = 192 * c + 128 + 2 * (lane & 31);
    r[0] = ld2(gf, e0); r[1] = ld2(gf, e1);
    r[2] = ld2(vf, e0); r[3] = ld2(vf, e1);
  };
  // the lane's operand addresses inside the staged chunk (rows a = 3 / b = 3 of the 16 x 4 operand tile are the constants 0 / 1)
  const float ga = a < 3 ? 1.f : 0.f;
  const int ia = (a < 3 ? a : 0) + 3 * kq, ib = 192 + (b < 3 ? b : 0) + 3 * kq;
  // two chunks ahead: an iteration (~1 k cycles) is shorter than the memory latency under load
  vf2 nxt[4], nx2[4];
  const int w0 = __builtin_amdgcn_readfirstlane(wave);      // (wave-uniform: the chunk index and its table loads stay scalar)
  if (w0 < nch) {
    fetch(w0, nxt);
    fetch(w0 + 4 < nch ? w0 + 4 : w0, nx2);
  }
  for (int c = w0; c < nch; c += 4) {
    wave_sync();                                  // (the previous chunk's operand reads are done)
    {
      const int e0 = 192 * c + 2 * lane, e1 = 192 * c + 128 + 2 * (lane & 31);
      *reinterpret_cast<vf2*>(stg + 2 * lane) = keep(nxt[0], e0);
      *reinterpret_cast<vf2*>(stg + 192 + 2 * lane) = keep(nxt[2], e0);
      if (lane < 32) {
        *reinterpret_cast<vf2*>(stg + 128 + 2 * lane) = keep(nxt[1], e1);
        *reinterpret_cast<vf2*>(stg + 192 + 128 + 2 * lane) = keep(nxt[3], e1);
      }
    }
    const int ng = gng[c];
    vf4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const vf4*>(gcw + (((size_t)c * 2 + 0) * 64 + lane) * 16 + 4 * q);
    int joint = gcj[(size_t)c * 32 + col];
    // the next chunk's records are requested BEHIND this chunk's weights and the order is pinned: vmcnt counts in order, so the MFMAs' wait
    // for the weights would otherwise also wait for the (HBM) prefetch the compiler had moved in front of them (measured: 31 % MFMA-busy,
    // every chunk exposed to the memory latency).  Unconditional (the last iteration re-reads its own chunk): straight-line code keeps the
    // waits counted instead of vmcnt(0).
    HA_SCHED_FENCE();
#pragma unroll
    for (int i = 0; i < 4; ++i) nxt[i] = nx2[i];
    fetch(c + 8 < nch ? c + 8 : c, nx2);
    HA_SCHED_FENCE();
    wave_sync();
    float av[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float gval = stg[ia + 12 * e] * ga;            // (vertices beyond V were staged as zeros)
      const float pv = stg[ib + 12 * e];
      av[e] = gval * (b < 3 ? pv : 1.f);
    }
    // slot group 0 in the same basic block as the loads above (a loop over the groups lets the compiler sink the weight loads into it, behind the
    // prefetch); the second group (3 of the synthetic model's 108 chunks) under its own uniform branch
    auto product = [&](const vf4 (&w)[4], int jnt) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 16; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], w[e >> 2][e & 3], acc, 0, 0, 0);
      // (two accumulator chains of 8 instead of one of 16: measured, no change -- 7.5 waves per SIMD cover the dependent accumulate)
      // accumulator register r of lane l: row 4 (l >> 4) + r = (a = l >> 4, b = r), slot l & 15
      if (jnt >= 0 && kq < 3) {
        float* d = tab + jnt * 12;
#pragma unroll
        for (int r = 0; r < 3; ++r) d[kq * 3 + r] += acc[r];
        d[9 + kq] += acc[3];
      }
      wave_sync();
    };
    product(bq, joint);
    if (ng > 1) {
      vf4 b1[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) b1[q] = *reinterpret_cast<const vf4*>(gcw + (((size_t)c * 2 + 1) * 64 + lane) * 16 + 4 * q);
      product(b1, gcj[(size_t)c * 32 + 16 + col]);
    }
  }
  __syncthreads();
  for (int i = tid; i < J * 12; i += 256)
    gA_out[(size_t)f * J * 12 + i] = (smem[i] + smem[768 + i]) + (smem[1536 + i] + smem[2304 + i]);
}

__global__ __launch_bounds__(256) void dense_gA_kernel(const float* __restrict__ g_verts, const float* __restrict__ v_posed,
                                                       const float* __restrict__ Wd, float* __restrict__ gA_out, int V, int Vpad, int J) {
  extern __shared__ __attribute__((aligned(16))) float smem[];      // [4 waves][64 joints][12]
  const int f = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int row = lane & 15, kq = lane >> 4, a = row >> 2, b = row & 3, col = lane & 15;
  const float* gf = g_verts + (size_t)f * V * 3;
  const float* vf = v_posed + (size_t)f * V * 3;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int v0 = wave * 16; v0 < Vpad; v0 += 64) {
    float av[4];
    vf4 bw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int v = v0 + 4 * e + kq;
      const bool live = v < V && a < 3;
      const float gval = live ? gf[(size_t)v * 3 + a] : 0.f;
      const float pval = (live && b < 3) ? vf[(size_t)v * 3 + b] : 1.f;
      av[e] = gval * pval;
      bw[e] = *reinterpret_cast<const vf4*>(Wd + (size_t)v * 64 + 4 * col);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bw[e].x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bw[e].y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bw[e].z, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bw[e].w, acc[3], 0, 0, 0);
    }
  }
  // accumulator register r of lane l, tile t: row 4 (l >> 4) + r = (a = l >> 4, b = r), joint 4 (l & 15) + t
  if (kq < 3) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float* d = smem + (wave * 64 + 4 * col + t) * 12;
#pragma unroll
      for (int r = 0; r < 3; ++r) d[kq * 3 + r] = acc[t][r];
      d[9 + kq] = acc[t][3];
    }
  }
  __syncthreads();
  for (int i = tid; i < J * 12; i += 256)
    gA_out[(size_t)f * J * 12 + i] = (smem[i] + smem[768 + i]) + (smem[1536 + i] + smem[2304 + i]);
}

// dL/dcoeff partials: out[ks][f][n] = sum over the k-split's chunks of gvpT[f][k] * Pd_k[k][n].  A wave owns 64 frames x one 128-
// coefficient group (tile t of the group <-> coefficients 4 c + t: one 16-byte load of a Pd_k row feeds four MFMA column
// tiles and the four accumulators of a lane are four consecutive coefficients -> 16-byte stores).  Blocks of one k-split
// run on one XCD (its Pd_k panel stays in that L2).
// Round 5, late: the A operand (lane = frame, 16 bytes of the frame's own 83 KB row per load: 64 cache lines per instruction, the four waves
// of a block -- same frames, different coefficient groups -- each fetching them again) goes through LDS: per 32-k tile the block loads its
// frames' 128-byte row pieces once, line by line (8 lanes per line), double-buffered, one barrier per tile = per 128 MFMAs of a wave.
constexpr int GCO_KT = 32, GCO_ROW = GCO_KT + 4;             // k-values per tile; floats per staged frame row (padded: 16-byte reads, stride 144 B)
constexpr int GCO_RP_FLOATS = 64 * GCO_ROW;                  // one 64-frame row pair of a tile
constexpr int GCO_THREADS = 320;                             // four MFMA waves + one loader wave
// The staging is the job of a FIFTH wave: vmcnt counts a wave's loads in order, so an MFMA wave that had requested the next tile's rows (HBM,
// 2-4 us under load) could not see its next B operand (L2, requested 2 k cycles ahead) arrive before them -- every tile exposed the HBM
// latency (PMC: 62 % MFMA-busy, the waves waiting 71 % of their cycles).  The loader wave waits for HBM; the MFMA waves only ever wait for L2.
__global__ __launch_bounds__(GCO_THREADS) void dense_gco_kernel(const float* __restrict__ gvpT, const float* __restrict__ Pd_k, float* __restrict__ out,
                                                                int N, int nchunks, int Kp, int ngroups, int n_rp, int KS, int cps, int rows_out,
                                                                int ld_out, int nrp_max) {
  extern __shared__ __attribute__((aligned(16))) float smem[];      // [2 buffers][nrp_max][64 frames][GCO_ROW]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int items_per_ks = n_rp * ngroups;
  const int bpk = (items_per_ks + 3) / 4;                       // blocks per k-split
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int ks = 8 * (jb / bpk) + xcd;
  const int item0 = (jb % bpk) * 4;
  if (ks >= KS || item0 >= items_per_ks) return;                // (block-uniform)
  const int rp0 = item0 / ngroups;                              // first row pair of the block
  const int last = item0 + 3 < items_per_ks ? item0 + 3 : items_per_ks - 1;
  const int nrp = last / ngroups - rp0 + 1;                     // row pairs the block's waves need (<= nrp_max)
  const int c0 = ks * cps, c1 = c0 + cps < nchunks ? c0 + cps : nchunks;
  const size_t Ktot = (size_t)nchunks * 192;
  const int kb0 = c0 * 192, kb1 = c1 * 192;                     // (multiples of GCO_KT: 192 = 6 x 32)
  const int buf_floats = nrp_max * GCO_RP_FLOATS;

  if (wave == 4) {
    // ---- loader: per tile and row pair 64 frames x 128 bytes, a cache line per 8 lanes (lane <-> frame l / 8 + 8 j, piece l % 8) --------------
    const int sf = lane >> 3, sp = lane & 7;
    for (int kt = kb0, ib = 0; kt < kb1; kt += GCO_KT, ib ^= 1) {
      float* buf = smem + ib * buf_floats;
      for (int q = 0; q < nrp; ++q) {
        vf4 r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int f = (rp0 + q) * 64 + 8 * j + sf;
          r[j] = *reinterpret_cast<const vf4*>(gvpT + (size_t)(f < N ? f : N - 1) * Ktot + kt + 4 * sp);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<vf4*>(buf + q * GCO_RP_FLOATS + (8 * j + sf) * GCO_ROW + 4 * sp) = r[j];
      }
      // tile kt is staged: the MFMA waves, which have just finished tile kt - GCO_KT out of the other buffer, meet the loader here
      __syncthreads();
    }
    __syncthreads();                                            // (the MFMA waves' barrier behind their last tile)
    return;
  }

  const int item = item0 + wave;
  const bool live = item < items_per_ks;                        // (an idle wave still meets the barriers)
  const int itc = live ? item : items_per_ks - 1;
  const int rp = itc / ngroups, grp = itc % ngroups;
  const int hi = lane >> 5, ln = lane & 31;
  f32x16 acc[2][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][t][i] = 0.f;
  const float* bp = Pd_k + (size_t)(4 * hi) * Kp + 128 * grp + 4 * ln;
  auto load_b = [&](int kb, vf4 (&bv)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = *reinterpret_cast<const vf4*>(bp + (size_t)(kb + e) * Kp);
  };
  auto mma = [&](const vf4 (&av)[2], const vf4 (&bv)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][e], bv[e].x, acc[m][0], 0, 0, 0);
        acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][e], bv[e].y, acc[m][1], 0, 0, 0);
        acc[m][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][e], bv[e].z, acc[m][2], 0, 0, 0);
        acc[m][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][e], bv[e].w, acc[m][3], 0, 0, 0);
      }
  };
  // 8 k-values per MFMA group (k = kb + 4 hi + e); a tile = 4 groups; the next group's B operands in flight under the MFMAs of the current one
  const float* arow = smem + (rp - rp0) * GCO_RP_FLOATS + ln * GCO_ROW + 4 * hi;      // + buffer, + 32 m rows, + 8 g
  vf4 bvA[4], bvB[4];
  load_b(kb0, bvA);
  __syncthreads();                                              // tile 0 is staged
  int ib = 0;
  for (int kt = kb0; kt < kb1; kt += GCO_KT, ib ^= 1) {
    const float* ab = arow + ib * buf_floats;
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
      vf4 av[2];
      load_b(kt + 8 * (g + 1), bvB);
      av[0] = *reinterpret_cast<const vf4*>(ab + 8 * g);
      av[1] = *reinterpret_cast<const vf4*>(ab + 32 * GCO_ROW + 8 * g);
      HA_SCHED_FENCE();
      mma(av, bvA);
      HA_SCHED_FENCE();
      const int kn = kt + 8 * (g + 2) < kb1 ? kt + 8 * (g + 2) : kt;     // (the group after the last one: a harmless re-read)
      load_b(kn, bvA);
      av[0] = *reinterpret_cast<const vf4*>(ab + 8 * (g + 1));
      av[1] = *reinterpret_cast<const vf4*>(ab + 32 * GCO_ROW + 8 * (g + 1));
      HA_SCHED_FENCE();
      mma(av, bvB);
      HA_SCHED_FENCE();
    }
    __syncthreads();                                            // the next tile is staged / this buffer may be overwritten
  }
  if (!live) return;
  // accumulator register i of lane l: row (i & 3) + 8 (i >> 2) + 4 (l >> 5), column l & 31 <-> coefficients 128 grp + 4 (l & 31) + t
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int f = rp * 64 + 32 * m + (i & 3) + 8 * (i >> 2) + 4 * hi;
      if (f < N)
        *reinterpret_cast<vf4*>(out + ((size_t)ks * rows_out + f) * ld_out + 128 * grp + 4 * ln) =
            vf4{acc[m][0][i], acc[m][1][i], acc[m][2][i], acc[m][3][i]};
    }
}

// launch plan of the dense backward (shared by the workspace query and the launch)
struct DenseBwdPlan {
  int nchunks, np, ngroups, ld, n_rp, KS, cps;
  size_t off_gvp, off_gA, off_gtl, off_gco, total;
};
static DenseBwdPlan dense_bwd_plan(const ha_smpl_model* m, int N, int n_active) {
  DenseBwdPlan P;
  const int Kc = m->NB + 1 + (n_active - 1) * 9;
  P.nchunks = m->sets[0].nchunks;
  P.np = ceil_div(P.nchunks, kGvpChunksPerBlock) * 4;
  P.ngroups = ceil_div(Kc, 128);
  P.ld = P.ngroups * 128;
  P.n_rp = ceil_div(N, 64);
  // K split of the dL/dcoeff kernel.  Its waves are MFMA chains of cps chunks (768 MFMAs of 64 cycles each per chunk) and a k-split's blocks
  // all run on ONE XCD (ks % 8: its Pd_k panel stays in that L2), so the launch takes  max over XCDs of  ceil(waves on the XCD / 128 SIMDs) x cps
  // chunk times: pick the split that minimises it (+ a little per split for the partial slabs the reduction reads).  Round 5: N = 1920 went
  // from 27 splits x 4 chunks (240 waves on three of the XCDs = two rounds, 8 chunk times) to 16 x 7 (120 waves per XCD, 7), 0.488 -> 0.435 ms.
  // ha_tune_set("dense_bwd_waves", w) overrides with the old rule (about w waves).
  const int items = P.n_rp * P.ngroups, bpk = ceil_div(items, 4);
  int cps = 1;
  if (g_dense_bwd_waves > 0) {
    int KS = ceil_div(g_dense_bwd_waves, items);
    KS = KS < 1 ? 1 : (KS > P.nchunks ? P.nchunks : KS);
    cps = ceil_div(P.nchunks, KS);
  } else {
    double best = 1e300;
    for (int c = 1; c <= P.nchunks; ++c) {
      const int KS = ceil_div(P.nchunks, c);
      if ((size_t)KS * N * P.ld > ((size_t)1 << 28) && c < P.nchunks) continue;      // partial slabs: at most 1 GiB
      int worst = 0;
      for (int x = 0; x < 8; ++x) {
        const int nks = KS > x ? (KS - x + 7) / 8 : 0;
        const int rounds = ceil_div(nks * bpk * 4, 128);
        worst = rounds > worst ? rounds : worst;
      }
      const double cost = (double)worst * c + 0.05 * KS;
      if (cost < best) { best = cost; cps = c; }
    }
  }
  P.cps = cps;
  P.KS = ceil_div(P.nchunks, P.cps);
  size_t o = 0;
  auto take = [&](size_t n) { const size_t at = o; o += (n + 3) & ~(size_t)3; return at; };
  P.off_gvp = take((size_t)N * P.nchunks * 192);
  P.off_gA = take((size_t)N * m->J * 12);
  P.off_gtl = take((size_t)N * P.np * 3);
  P.off_gco = take((size_t)P.KS * N * P.ld);
  P.total = o;
  return P;
}

static void fill_model(FrameParams& p, const ha_smpl_model* m, int slot) {
  memset(&p, 0, sizeof(p));
  p.Jt = m->Jt; p.Js = m->Js; p.parents = m->parents; p.jdepth = m->jdepth;
  p.child_start = m->child_start; p.child_idx = m->child_idx; p.anc = m->anc; p.nrounds = m->nrounds;
  p.J = m->J; p.NB = m->NB; p.Kfull = m->Kfull; p.Kfull_pad = m->Kfull_pad; p.kf4 = (m->Kfull_pad + 3) & ~3; p.depth = m->depth;
  const VertexSet& s = m->sets[slot];
  p.Pd_v = s.Pd_v; p.Pd_k = s.Pd_k; p.Kp = ceil_div(m->Kfull, 128) * 128; p.w = s.w; p.idx = s.idx; p.Wc = s.Wc;
  p.nverts = s.n; p.nchunks = s.nchunks; p.nnz = m->nnz;
  p.n_head = 0; p.jstride = m->J;
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
  if (coeff) *coeff = (int64_t)((Kc + 7) / 8 * 8) * Npad;   // [k/4][frame][4], whole k-pair quads
  return HA_OK;
}

extern "C" int ha_lbs_skin(const ha_smpl_model* m, int N, const float* v_posed, const float* A, const float* transl,
                           float* verts, void* stream) {
  HA_REQUIRE(m && v_posed && A && verts, "ha_lbs_skin: null argument");
  HA_REQUIRE(N >= 1, "ha_lbs_skin: N must be >= 1");
  if (m->nnz > 4 || m->V < kSkinMinVerts) {
    set_error("ha_lbs_skin: needs <=4 influences per vertex and V>=%d (model has nnz=%d V=%d)", kSkinMinVerts, m->nnz, m->V);
    return HA_ERR_UNSUPPORTED;
  }
  DeviceGuard guard(m->device);
  const long long total = (long long)N * m->V;
  // auto: 8 waves per block while the arrays fit the 256 MiB Infinity Cache, 16 beyond (measured +2 % at N=30720);
  // non-temporal stores always (the output is consumed by another kernel, never re-read by this one)
  const int sv = g_skin_variant >= 0 ? g_skin_variant : ((total * 12 <= (400ll << 20) ? 1 : 2) | 4);
  int nw = 4 << (sv & 3);
  while (nw > 4 && m->V < 256 * nw) nw >>= 1;
  const bool nt = (sv & 4) != 0;
  const int diag = (sv >> 3) & 3;
  const int nb = (int)((total + 256 * nw - 1) / (256 * nw));
  const size_t lds = (size_t)(2 * m->J * 12 + 256 * nw * 3) * sizeof(float);
#define HA_SKIN(NTV, D, NWV) HA_LAUNCH((lbs_skin_wave_kernel<NTV, D, NWV>), dim3(nb), dim3(64 * NWV), lds, (hipStream_t)stream, \
                                               v_posed, A, transl, m->w4, m->idx4, verts, N, m->V, m->J)
  if (diag == 1) HA_SKIN(true, 1, 8);
  else if (diag == 2) HA_SKIN(true, 2, 8);
  else if (nw == 4) { if (nt) HA_SKIN(true, 0, 4); else HA_SKIN(false, 0, 4); }
  else if (nw == 8) { if (nt) HA_SKIN(true, 0, 8); else HA_SKIN(false, 0, 8); }
  else { if (nt) HA_SKIN(true, 0, 16); else HA_SKIN(false, 0, 16); }
#undef HA_SKIN
  HA_LAUNCH_CHECK();
  return HA_OK;
}

extern "C" int ha_smpl_forward(const ha_smpl_model* m, int slot, int N, int n_active, const float* pose, const float* betas,
                               const float* transl, float* verts, float* joints, float* A_out, float* ws_vposed,
                               float* ws_coeff, int algo, void* stream) {
  int rc = check_common("ha_smpl_forward", m, slot, N, n_active);
  if (rc != HA_OK) return rc;
  HA_REQUIRE(pose && betas, "ha_smpl_forward: pose and betas are required");
  HA_REQUIRE(algo >= 0 && algo <= 3, "ha_smpl_forward: unknown algo %d", algo);
  const bool fused_ok = slot == 0 && m->nnz <= 4 && m->V >= kSkinMinVerts && ws_coeff && A_out;
  const bool dense_ok = fused_ok && ws_vposed;
  if (algo == 2 && !dense_ok) {
    set_error("ha_smpl_forward: algo 2 needs slot 0, <=4 skinning influences, V>=%d and the A/vposed/coeff workspaces", kSkinMinVerts);
    return HA_ERR_INVALID_ARG;
  }
  if (algo == 3 && !(fused_ok && verts)) {
    set_error("ha_smpl_forward: algo 3 needs slot 0, <=4 skinning influences, V>=%d, verts and the A/coeff workspaces", kSkinMinVerts);
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
  const size_t lds = ((size_t)FW * (((m->Kfull_pad + 3) & ~3) + m->J * 12) + kXchFloats) * sizeof(float);   // + the block's exchange area
  const int blocks = ceil_div(N, FW);
  if (algo == 1) {
    p.verts = verts;
    HA_LAUNCH(smpl_frame_fwd_kernel, dim3(blocks), dim3(FW * 64), lds, st, p);
    HA_LAUNCH_CHECK();
    return HA_OK;
  }
  // algo 2 / 3: joints / A / coefficients by the frame kernel, then MFMA blend, then streaming skinning (2) or the skinning in the blend's epilogue (3)
  p.verts = nullptr;
  p.nchunks = 0;
  p.coeffT = ws_coeff;
  p.Npad = ceil_div(N, 64) * 64;
  HA_LAUNCH(smpl_frame_fwd_kernel, dim3(blocks), dim3(FW * 64), lds, st, p);
  HA_LAUNCH_CHECK();
  if (!verts) return HA_OK;
  {
    const int n_vt = m->Vpad / 32, n_ft = p.Npad / 64;
    const int n_vtg = ceil_div(n_vt, 4);
    const int n_vtg8 = ceil_div(n_vtg, 8) * 8;
    const int KQ = (p.Kc + 7) / 8;      // quads of k-pairs
    if (algo == 3) {
      // blend + skin in one kernel (forward-only callers: v_posed, the adjoint's input, is never written)
      const int n_groups = ceil_div(n_vt, kFusedTilesPerBlock);
      const size_t lds_f = (size_t)(32 * m->J * 12 + 128) * sizeof(float);
#ifndef HA_SIMT_EMU
      // more than the default 64 KB of dynamic LDS: a per-DEVICE function attribute, remembered in the model handle (one handle per device)
      if (!m->fused_lds_attr_set) {
        HA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pose_blend_skin_kernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        m->fused_lds_attr_set = true;
      }
#endif
      HA_REQUIRE(lds_f <= 80 * 1024, "ha_smpl_forward: algo 3 keeps the A matrices of 32 frames in LDS (J <= 53)");
      HA_LAUNCH((pose_blend_skin_kernel<4, true>), dim3(n_groups * (p.Npad / 32)), dim3(256), lds_f, st, ws_coeff, p.Npad, KQ,
                           ceil_div(m->Kfull_pad / 2, 4), m->Pd_m, N, m->V, n_vt, n_groups, A_out, transl, m->w4, m->idx4, verts, m->J);
      HA_LAUNCH_CHECK();
      return HA_OK;
    }
    HA_LAUNCH(pose_blend_mfma_kernel, dim3(n_vtg8 * n_ft), dim3(256), 0, st, ws_coeff, p.Npad, KQ, ceil_div(m->Kfull_pad / 2, 4),
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
  const size_t lds = ((size_t)FW * bwd_wave_floats((m->Kfull_pad + 3) & ~3, m->J) + kXchFloats) * sizeof(float);
  HA_LAUNCH(smpl_frame_bwd_kernel, dim3(ceil_div(N, FW)), dim3(FW * 64), lds, (hipStream_t)stream, p);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

extern "C" int ha_smpl_forward_split(const ha_smpl_model* m, int slot, int N, int n_active, const float* pose, const float* betas,
                                     const float* transl, int n_head, float* joints_ext, float* verts_tail, void* stream) {
  int rc = check_common("ha_smpl_forward_split", m, slot, N, n_active);
  if (rc != HA_OK) return rc;
  HA_REQUIRE(pose && betas && joints_ext, "ha_smpl_forward_split: pose, betas and joints_ext are required");
  HA_REQUIRE(n_head >= 0 && n_head <= m->sets[slot].n, "ha_smpl_forward_split: n_head=%d out of range 0..%d", n_head, m->sets[slot].n);
  HA_REQUIRE(verts_tail || n_head == m->sets[slot].n, "ha_smpl_forward_split: verts_tail is required for the vertices behind the head");
  DeviceGuard guard(m->device);
  FrameParams p;
  fill_model(p, m, slot);
  p.N = N; p.n_active = n_active; p.Kc = m->NB + 1 + (n_active - 1) * 9;
  p.pose = pose; p.betas = betas; p.transl = transl;
  p.joints = joints_ext; p.verts = verts_tail ? verts_tail : joints_ext;   // (non-null = "evaluate the vertex set")
  p.n_head = n_head; p.jstride = m->J + n_head;
  const size_t lds = ((size_t)FW * (((m->Kfull_pad + 3) & ~3) + m->J * 12) + kXchFloats) * sizeof(float);
  HA_LAUNCH(smpl_frame_fwd_kernel, dim3(ceil_div(N, FW)), dim3(FW * 64), lds, (hipStream_t)stream, p);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

extern "C" int ha_smpl_backward_split(const ha_smpl_model* m, int slot, int N, int n_active, const float* pose, const float* betas,
                                      int n_head, const float* g_joints_ext, const float* g_verts_tail, float* g_pose, float* g_betas,
                                      float* g_transl, void* stream) {
  int rc = check_common("ha_smpl_backward_split", m, slot, N, n_active);
  if (rc != HA_OK) return rc;
  HA_REQUIRE(pose && betas, "ha_smpl_backward_split: pose and betas are required");
  HA_REQUIRE(n_head >= 0 && n_head <= m->sets[slot].n, "ha_smpl_backward_split: n_head=%d out of range 0..%d", n_head, m->sets[slot].n);
  DeviceGuard guard(m->device);
  FrameParams p;
  fill_model(p, m, slot);
  p.N = N; p.n_active = n_active; p.Kc = m->NB + 1 + (n_active - 1) * 9;
  p.pose = pose; p.betas = betas;
  p.g_verts = g_verts_tail; p.g_joints = g_joints_ext;
  p.g_pose = g_pose; p.g_betas = g_betas; p.g_transl = g_transl;
  p.n_head = n_head; p.jstride = m->J + n_head;
  const size_t lds = ((size_t)FW * bwd_wave_floats((m->Kfull_pad + 3) & ~3, m->J) + kXchFloats) * sizeof(float);
  HA_LAUNCH(smpl_frame_bwd_kernel, dim3(ceil_div(N, FW)), dim3(FW * 64), lds, (hipStream_t)stream, p);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

// ---- "parts" form: split pose, shared shape rows, addends (the stage-3 composites of humor_amd/stage3.py) -------------------------
extern "C" int ha_smpl_forward_parts(const ha_smpl_model* m, int slot, int N, int n_active, const float* root, const float* body,
                                     const float* betas, int betas_div, const float* transl, int n_head, float* joints_ext,
                                     float* verts_tail, void* stream) {
  int rc = check_common("ha_smpl_forward_parts", m, slot, N, n_active);
  if (rc != HA_OK) return rc;
  HA_REQUIRE(root && betas && joints_ext && (body || n_active == 1), "ha_smpl_forward_parts: root, body, betas and joints_ext are required");
  HA_REQUIRE(betas_div >= 1 && N % betas_div == 0, "ha_smpl_forward_parts: betas_div=%d must divide N=%d", betas_div, N);
  HA_REQUIRE(n_head >= 0 && n_head <= m->sets[slot].n, "ha_smpl_forward_parts: n_head=%d out of range 0..%d", n_head, m->sets[slot].n);
  const bool joints_only = !verts_tail && n_head == 0;      // no vertex of the subset is wanted: the J joints alone (no blend, no skinning)
  HA_REQUIRE(verts_tail || n_head == m->sets[slot].n || joints_only, "ha_smpl_forward_parts: verts_tail is required for the vertices behind the head");
  DeviceGuard guard(m->device);
  FrameParams p;
  fill_model(p, m, slot);
  if (joints_only) { p.nverts = 0; p.nchunks = 0; }
  p.N = N; p.n_active = n_active; p.Kc = m->NB + 1 + (n_active - 1) * 9;
  p.pose = root; p.pose_body = body ? body : root; p.betas = betas; p.betas_div = betas_div; p.transl = transl;
  p.joints = joints_ext; p.verts = verts_tail ? verts_tail : joints_ext;
  p.n_head = n_head; p.jstride = m->J + n_head;
  const size_t lds = ((size_t)FW * (((m->Kfull_pad + 3) & ~3) + m->J * 12) + kXchFloats) * sizeof(float);
  HA_LAUNCH(smpl_frame_fwd_kernel, dim3(ceil_div(N, FW)), dim3(FW * 64), lds, (hipStream_t)stream, p);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

extern "C" int ha_smpl_backward_parts(const ha_smpl_model* m, int slot, int N, int n_active, const float* root, const float* body,
                                      const float* betas, int betas_div, int n_head, const float* g_joints_ext, int gj_rows,
                                      int gj_stride, const float* g_verts_tail, const float* add_root, const float* add_body,
                                      const float* add_betas, const float* add_transl, float* g_root, float* g_body, float* g_betas,
                                      float* g_transl, void* stream) {
  int rc = check_common("ha_smpl_backward_parts", m, slot, N, n_active);
  if (rc != HA_OK) return rc;
  HA_REQUIRE(root && betas && g_root && (body || n_active == 1), "ha_smpl_backward_parts: root, body, betas and g_root are required");
  HA_REQUIRE(betas_div >= 1 && N % betas_div == 0, "ha_smpl_backward_parts: betas_div=%d must divide N=%d", betas_div, N);
  HA_REQUIRE(n_head >= 0 && n_head <= m->sets[slot].n, "ha_smpl_backward_parts: n_head=%d out of range 0..%d", n_head, m->sets[slot].n);
  HA_REQUIRE(gj_rows >= 0 && gj_rows <= m->J + n_head && gj_stride >= 0, "ha_smpl_backward_parts: gj_rows=%d out of range", gj_rows);
  DeviceGuard guard(m->device);
  FrameParams p;
  fill_model(p, m, slot);
  p.N = N; p.n_active = n_active; p.Kc = m->NB + 1 + (n_active - 1) * 9;
  p.pose = root; p.pose_body = body ? body : root; p.betas = betas; p.betas_div = betas_div;
  if (!g_verts_tail && (n_head == 0 || !g_joints_ext || (gj_rows != 0 && gj_rows <= m->J))) { p.nverts = 0; p.nchunks = 0; }   // no vertex carries a gradient
  p.g_verts = g_verts_tail; p.g_joints = g_joints_ext; p.gj_rows = gj_rows; p.gj_stride = gj_stride;
  p.g_pose = g_root; p.g_body = g_body; p.g_betas = g_betas; p.g_transl = g_transl;
  p.add_root = add_root; p.add_body = add_body; p.add_betas = add_betas; p.add_transl = add_transl;
  p.n_head = n_head; p.jstride = m->J + n_head;
  const size_t lds = ((size_t)FW * bwd_wave_floats((m->Kfull_pad + 3) & ~3, m->J) + kXchFloats) * sizeof(float);
  HA_LAUNCH(smpl_frame_bwd_kernel, dim3(ceil_div(N, FW)), dim3(FW * 64), lds, (hipStream_t)stream, p);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

namespace ha {
// out[b][w] = sum_t src[b][t][w] + add1[b][w] + add2[b][w]  (the per-frame shape gradients of a [B, T] batch back to one row per
// sequence, together with the gradients other consumers of the same shape rows produced: one launch for sum + add + add)
__global__ __launch_bounds__(256) void seq_sum_add_kernel(const float* __restrict__ src, const float* __restrict__ add1,
                                                          const float* __restrict__ add2, float* __restrict__ out, int T, int W) {
  extern __shared__ __attribute__((aligned(16))) float smem[];     // [4][64] partial sums of the four frame slices
  const int b = blockIdx.x, w = threadIdx.x & 63, sl = threadIdx.x >> 6;
  float acc = 0.f;
  if (w < W)
    for (int t = sl; t < T; t += 4) acc += src[((size_t)b * T + t) * W + w];
  smem[sl * 64 + w] = acc;
  __syncthreads();
  if (sl != 0 || w >= W) return;
  acc = (smem[w] + smem[64 + w]) + (smem[128 + w] + smem[192 + w]);
  if (add1) acc += add1[(size_t)b * W + w];
  if (add2) acc += add2[(size_t)b * W + w];
  out[(size_t)b * W + w] = acc;
}
}  // namespace ha

extern "C" int ha_seq_sum_add(int B, int T, int W, const float* src, const float* add1, const float* add2, float* out, void* stream) {
  HA_REQUIRE(src && out && B >= 1 && T >= 1 && W >= 1 && W <= 64, "ha_seq_sum_add: need src, out, B, T >= 1 and 1 <= W <= 64");
  HA_LAUNCH(seq_sum_add_kernel, dim3(B), dim3(256), 256 * sizeof(float), (hipStream_t)stream, src, add1, add2, out, T, W);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

extern "C" int ha_smpl_backward_dense_workspace(const ha_smpl_model* m, int N, int n_active, int64_t* ws_floats) {
  int rc = check_common("ha_smpl_backward_dense_workspace", m, 0, N, n_active);
  if (rc != HA_OK) return rc;
  HA_REQUIRE(ws_floats, "ha_smpl_backward_dense_workspace: null argument");
  *ws_floats = (int64_t)dense_bwd_plan(m, N, n_active).total;
  return HA_OK;
}

extern "C" int ha_smpl_backward_dense(const ha_smpl_model* m, int N, int n_active, const float* pose, const float* betas,
                                      const float* g_verts, const float* g_joints, const float* v_posed, const float* A, float* ws,
                                      float* g_pose, float* g_betas, float* g_transl, void* stream) {
  int rc = check_common("ha_smpl_backward_dense", m, 0, N, n_active);
  if (rc != HA_OK) return rc;
  HA_REQUIRE(pose && betas && g_verts && v_posed && A && ws, "ha_smpl_backward_dense: null argument");
  if (m->nnz > 4 || !m->Wd) {
    set_error("ha_smpl_backward_dense: needs <=4 skinning influences per vertex (model has nnz=%d)", m->nnz);
    return HA_ERR_UNSUPPORTED;
  }
  DeviceGuard guard(m->device);
  hipStream_t st = (hipStream_t)stream;
  const DenseBwdPlan P = dense_bwd_plan(m, N, n_active);
  HA_REQUIRE(P.np <= 64, "ha_smpl_backward_dense: V=%d too large for the partial-sum layout", m->V);
  const VertexSet& s0 = m->sets[0];
  const int Kp = ceil_div(m->Kfull, 128) * 128;
  HA_LAUNCH(dense_gvp_kernel, dim3(P.np / 4, N), dim3(256), (size_t)m->J * 12 * sizeof(float), st, g_verts, A, m->w4, m->idx4,
                     ws + P.off_gvp, ws + P.off_gtl, m->V, m->J, P.nchunks, P.np);
  HA_LAUNCH_CHECK();
  if (g_dense_gA_sparse == 2 && m->gc_joint)
  {
    if ((m->V & 1) == 0 && (reinterpret_cast<uintptr_t>(g_verts) & 7) == 0 && (reinterpret_cast<uintptr_t>(v_posed) & 7) == 0)
      HA_LAUNCH(compressed_gA_kernel<true>, dim3(N), dim3(256), (size_t)(4 * 64 * 12 + 4 * GA_STAGE) * sizeof(float), st, g_verts, v_posed,
                         m->gc_joint, m->gc_w, m->gc_ng, ws + P.off_gA, m->V, m->Vpad, m->J);
    else
      HA_LAUNCH(compressed_gA_kernel<false>, dim3(N), dim3(256), (size_t)(4 * 64 * 12 + 4 * GA_STAGE) * sizeof(float), st, g_verts, v_posed,
                         m->gc_joint, m->gc_w, m->gc_ng, ws + P.off_gA, m->V, m->Vpad, m->J);
  }
  else if (g_dense_gA_sparse)
    HA_LAUNCH(sparse_gA_kernel, dim3(N), dim3(256), 0, st, g_verts, v_posed, m->ja_start, m->ja_v, m->ja_w, m->ja_order, ws + P.off_gA,
                       m->V, m->J);
  else
    HA_LAUNCH(dense_gA_kernel, dim3(N), dim3(256), (size_t)4 * 64 * 12 * sizeof(float), st, g_verts, v_posed, m->Wd, ws + P.off_gA, m->V,
                       m->Vpad, m->J);
  HA_LAUNCH_CHECK();
  {
    const int bpk = ceil_div(P.n_rp * P.ngroups, 4);
    const int nblk = 8 * ceil_div(P.KS, 8) * bpk;
    // row pairs of 64 frames a block's four waves can span (their items are consecutive (row pair, coefficient group) pairs)
    const int nrp_max = P.ngroups == 1 ? 4 : (P.ngroups % 4 == 0 ? 1 : 2);
    const size_t lds_gco = (size_t)2 * nrp_max * GCO_RP_FLOATS * sizeof(float);
#ifndef HA_SIMT_EMU
    if (lds_gco > 64 * 1024 && !m->gco_lds_attr_set) {      // (one 128-coefficient group only: four row pairs per block, 74 KB)
      HA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dense_gco_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
      m->gco_lds_attr_set = true;
    }
#endif
    HA_LAUNCH(dense_gco_kernel, dim3(nblk), dim3(GCO_THREADS), lds_gco, st, ws + P.off_gvp, s0.Pd_k, ws + P.off_gco, N, P.nchunks, Kp, P.ngroups, P.n_rp,
                       P.KS, P.cps, N, P.ld, nrp_max);
    HA_LAUNCH_CHECK();
  }
  FrameParams p;
  fill_model(p, m, 0);
  p.N = N; p.n_active = n_active; p.Kc = m->NB + 1 + (n_active - 1) * 9;
  p.pose = pose; p.betas = betas;
  p.g_verts = nullptr; p.g_joints = g_joints;
  p.g_pose = g_pose; p.g_betas = g_betas; p.g_transl = g_transl;
  p.gA_in = ws + P.off_gA; p.gco_part = ws + P.off_gco; p.gco_ks = P.KS; p.gco_rows = N; p.gco_ld = P.ld;
  p.gtl_part = ws + P.off_gtl; p.gtl_np = P.np;
  const size_t lds = ((size_t)FW * bwd_wave_floats((m->Kfull_pad + 3) & ~3, m->J) + kXchFloats) * sizeof(float);
  HA_LAUNCH(smpl_frame_bwd_kernel, dim3(ceil_div(N, FW)), dim3(FW * 64), lds, st, p);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

#ifdef HA_SMPL_TIMING
extern "C" int ha_debug_smpl_timing(unsigned long long* out /* [2][16] */) {
  HA_CHECK_HIP(hipDeviceSynchronize());
  HA_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(ha::g_smpl_pt), sizeof(unsigned long long) * 2 * 16));
  return HA_OK;
}
#endif

// TEST INFRASTRUCTURE ONLY (CPU tier, host SIMT emulator build -- never part of libhumor_amd.so): the ha_emu_* hooks that run the building
// blocks of the persistent / pipelined roll-out kernels (exchange consumers, GroupNorm and its adjoint, weight packing + MFMA chains of every
// layer role, publish -> gather round trips, dL/dz partials) on the emulator against PyTorch: tests/test_rollout_emu.py.
// The product source is included, not edited: its file-local templates are what the hooks instantiate.  tests/simt_emu/build.py compiles this
// file in place of humor_amd/csrc/rollout_persist.hip.
#include "../../humor_amd/csrc/rollout_persist.hip"

#ifndef HA_SIMT_EMU
#error "emulator build only"
#endif

namespace ha {
// ---- CPU test tier only (host SIMT emulator build; not part of libhumor_amd.so): the exchange consumers on a prepared exchange region --------
// tests/test_rollout_emu.py fills a region the way publish() does -- the {value, tag} granules of channel c, row pair p at byte
// ha_emu_xslot(c) * 32 + 16 p -- and checks gather_norm (GroupNorm + ReLU, statistics) and gather_norm_bwd (its adjoint) against PyTorch.
template <int NQ, int GROUP>
__global__ void emu_gather_norm_kernel(const unsigned char* xch, unsigned tag, const float* gamma, const float* beta, float* xs_out, float* stats, int row0) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int C = NQ * 256;
  float* gb = smem;
  float* xs = smem + 4 * C;
  const int tid = threadIdx.x;
  gb_fill<true>(gb, gamma, beta, C, tid);
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xch, 0, C * 32, 0x00020000);
  const bool ok = gather_norm<NQ, GROUP>(rs, 0u, tag, gb, xs, tid, stats, row0);
  __syncthreads();
  for (int i = tid; i < 4 * C; i += 256) xs_out[i] = ok ? xs[i] : as_f(0x7fc00000u);
}
template <int NQ, int GROUP>
__global__ void emu_gather_norm_bwd_kernel(const unsigned char* xch, unsigned tag, const float* gamma, const float* beta, const float* ht, const float* stats,
                                           float* ds_out, int row0) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int C = NQ * 256;
  float* gb = smem;
  float* ds = smem + 2 * C;
  const int tid = threadIdx.x;
  gb_fill<false>(gb, gamma, beta, C, tid);
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xch, 0, C * 32, 0x00020000);
  GnbRegs<NQ> gr;
  gnb_issue<NQ, GROUP>(ht, stats, row0, tid, gr);
  const bool ok = gather_norm_bwd<NQ, GROUP>(rs, 0u, tag, gb, gr, ds, tid);
  __syncthreads();
  for (int i = tid; i < 4 * C; i += 256) ds_out[i] = ok ? ds[i] : as_f(0x7fc00000u);
}
// lane_reduce.h composites on per-lane inputs (what tools/microbench/persist_probe.hip does on the GPU): in [n][64 lanes] -> out [n_out][64 lanes]
__global__ void emu_lane_reduce_kernel(int which, const float* in, float* out) {
  const int l = threadIdx.x & 63;
  if (which == 0) {             // wave_sum16
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = in[i * 64 + l];
    lr::wave_sum16(v);
    for (int i = 0; i < 16; ++i) out[i * 64 + l] = v[i];
  } else if (which == 1) {      // half_sum8
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = in[i * 64 + l];
    lr::half_sum8(v);
    for (int i = 0; i < 8; ++i) out[i * 64 + l] = v[i];
  } else if (which == 2) {      // block_sum8
    float v[8], o[2];
    for (int i = 0; i < 8; ++i) v[i] = in[i * 64 + l];
    lr::block_sum8(v, o);
    out[l] = o[0]; out[64 + l] = o[1];
  } else if (which == 3) {      // block_sum4
    float v[4];
    for (int i = 0; i < 4; ++i) v[i] = in[i * 64 + l];
    out[l] = lr::block_sum4(v);
  } else {                      // block_sum8_head + kblock_sum2 (the pipelined kernels' split form)
    float v[8], o[2];
    for (int i = 0; i < 8; ++i) v[i] = in[i * 64 + l];
    lr::block_sum8_head(v, o);
    lr::kblock_sum2(o);
    out[l] = o[0]; out[64 + l] = o[1];
  }
}
// publish() of every wave of a team for one layer output: wave g (block g / 4) holds the MFMA partials `sums` of its 4 NCG columns x 4 rows
// (in [g][4 NCG values][64 lanes]); writes the exchange region, the launch-chain slab and the team-layout copy
template <int NCG, int GW>
__global__ void emu_publish_kernel(const float* sums_in, const float* bias, unsigned char* xch, unsigned tag, float* slab, float* ht, int row0) {
  const int lane = threadIdx.x & 63, g = blockIdx.x * 4 + (threadIdx.x >> 6);
  float sums[4 * NCG];
  for (int i = 0; i < 4 * NCG; ++i) sums[i] = sums_in[((size_t)g * 4 * NCG + i) * 64 + lane];
  const int j4 = lane & 3, h4 = 4 * (lane >> 5);
  const float b = NCG == 2 ? bias[8 * g + h4 + j4] : bias[4 * g + j4];
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xch, 0, 1 << 20, 0x00020000);
  publish<NCG, true, GW>(sums, b, 4 * NCG * g, xch, rs, 0u, tag, slab, row0, lane, ht);
}
// ONE forward layer of a team as the persistent kernel computes it: every wave loads its share of the packed weights into its register
// arrays, multiplies the A operand [channel][4 rows] in LDS with v_mfma_f32_4x4x1 chains (mma_layer) and publishes the result
template <int L>
__global__ void emu_layer_kernel(const float* Wreg, const float* bias, const float* x_main, const float* z, unsigned char* xch, unsigned tag, float* slab,
                                 float* ht, int row0) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int CM = L == 0 ? P_XPAD : (L == 1 ? P_H0 : (L == 2 ? P_H1 : P_H2));
  float* xs = smem;
  float* zs = smem + CM * 4;
  const int tid = threadIdx.x, lane = tid & 63, g = blockIdx.x * 4 + (tid >> 6);
  for (int i = tid; i < CM * 4; i += 256) xs[i] = x_main[i];
  for (int i = tid; i < P_ZD * 4; i += 256) zs[i] = z[i];
  __syncthreads();
  float wa[NWA], wv[NREG - NWA > 0 ? NREG - NWA : 1];
  const float* wp = Wreg + (size_t)g * NREG * 64 + lane;
  for (int r = 0; r < NWA; ++r) wa[r] = wp[(size_t)r * 64];
  for (int r = NWA; r < NREG; ++r) wv[r - NWA] = wp[(size_t)r * 64];
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xch, 0, 1 << 20, 0x00020000);
  const int j4 = lane & 3, h4 = 4 * (lane >> 5);
  if constexpr (L <= 1) {
    float acc[8];
    if constexpr (L == 0) mma_layer<NC0, NCZ, 2, R0>(xs, zs, wa, wv, lane, acc);
    else mma_layer<NC1, NCZ, 2, R1>(xs, zs, wa, wv, lane, acc);
    publish<2, true, 64>(acc, bias[8 * g + h4 + j4], 8 * g, xch, rs, 0u, tag, slab, row0, lane, ht);
  } else if constexpr (L == 2) {
    float acc[4];
    mma_layer<NC2, NCZ, 1, R2>(xs, zs, wa, wv, lane, acc);
    publish<1, true, 32>(acc, bias[4 * g + j4], 4 * g, xch, rs, 0u, tag, slab, row0, lane, ht);
  } else {
    if (g < L3_WAVES) {
      float acc[4];
      mma_layer<NC3, NCZ, 1, R3>(xs, zs, wa, wv, lane, acc);
      publish<1, true, 0>(acc, bias[4 * g + j4], 4 * g, xch, rs, 0u, tag, slab, row0, lane);
    }
  }
}
// ONE transposed layer of the adjoint (L = 3, 2, 1): dL/d(input activation of layer L) [4 rows] = dh_L W_L with the adjoint's packing (pack_backward)
template <int L>
__global__ void emu_layer_t_kernel(const float* Wreg_b, const float* dh, unsigned char* xch, unsigned tag, int row0) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int CM = L == 3 ? P_RAWPAD : (L == 2 ? P_H2 : (L == 1 ? P_H1 : P_H0));
  float* sD = smem;
  const int tid = threadIdx.x, lane = tid & 63, g = blockIdx.x * 4 + (tid >> 6);
  for (int i = tid; i < CM * 4; i += 256) sD[i] = dh[i];
  float* sWz = smem + CM * 4 + (tid >> 6) * NLW * 64;           // (layer 0: the LDS-resident tail of K)
  if constexpr (L == 0) {
    const float* wz = Wreg_b + (size_t)g * (NREG_B_ALL + BC0_LDS) * 64 + lane;
    for (int r = 0; r < NLW; ++r) sWz[r * 64 + lane] = wz[(size_t)(NREG_B + r) * 64];
  }
  __syncthreads();
  float wa[NWA_B < NREG_B ? NWA_B : NREG_B], wv[NREG_B - NWA_B > 0 ? NREG_B - NWA_B : 1];
  const float* wp = Wreg_b + (size_t)g * (NREG_B_ALL + BC0_LDS) * 64 + lane;
  for (int r = 0; r < NWA_B && r < NREG_B; ++r) wa[r] = wp[(size_t)r * 64];
  for (int r = NWA_B; r < NREG_B; ++r) wv[r - NWA_B] = wp[(size_t)r * 64];
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xch, 0, 1 << 20, 0x00020000);
  if constexpr (L == 0) {
    float acc[4];
    if (g < L0T_WAVES) {
      float tail[4];
      mma_layer<BC0_REG, 0, 1, BR0>(sD, sD, wa, wv, lane, acc);
      dz_mma<BC0_LDS>(sD + 64 * BC0_REG, sWz + NDZ * 64, lane, tail);
      for (int i = 0; i < 4; ++i) acc[i] += tail[i];
    } else { acc[0] = acc[1] = acc[2] = acc[3] = 0.f; }
    if (g < P_XPAD / 4) publish<1, true, 0>(acc, 0.f, 4 * g, xch, rs, 0u, tag, nullptr, row0, lane);
  } else if constexpr (L == 3) {
    float acc[4];
    mma_layer<BC3, 0, 1, BR3>(sD, sD, wa, wv, lane, acc);
    publish<1, true, 32>(acc, 0.f, 4 * g, xch, rs, 0u, tag, nullptr, row0, lane);
  } else {
    float acc[8];
    if constexpr (L == 2) mma_layer<BC2, 0, 2, BR2>(sD, sD, wa, wv, lane, acc);
    else mma_layer<BC1, 0, 2, BR1>(sD, sD, wa, wv, lane, acc);
    publish<2, true, 64>(acc, 0.f, 8 * g, xch, rs, 0u, tag, nullptr, row0, lane);
  }
}
// dL/dz of one step: every wave's K-split partial products of the four layers (LDS-resident weights of pack_backward, dz_mma, dz_store), then
// dz_reduce_kernel.  dh3 [224][4], dh2 [512][4], dh1 / dh0 [1024][4] = the activation adjoints in the MFMA operand layout
__global__ void emu_dz_kernel(const float* Wreg_b, const float* dh3, const float* dh2, const float* dh1, const float* dh0, float* dz_part, int row0) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sD3 = smem;
  float* sD2 = sD3 + P_RAWPAD * 4;
  float* sD1 = sD2 + P_H2 * 4;
  float* sD0 = sD1 + P_H1 * 4;
  float* sWzAll = sD0 + P_H0 * 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = blockIdx.x * 4 + wave;
  for (int i = tid; i < P_RAWPAD * 4; i += 256) sD3[i] = dh3[i];
  for (int i = tid; i < P_H2 * 4; i += 256) sD2[i] = dh2[i];
  for (int i = tid; i < P_H1 * 4; i += 256) sD1[i] = dh1[i];
  for (int i = tid; i < P_H0 * 4; i += 256) sD0[i] = dh0[i];
  float* sWz = sWzAll + wave * NLW * 64;
  const float* wp = Wreg_b + (size_t)g * (NREG_B_ALL + BC0_LDS) * 64 + lane;
  for (int r = 0; r < NLW; ++r) sWz[r * 64 + lane] = wp[(size_t)(NREG_B + r) * 64];
  __syncthreads();
  float accz[4];
  if (g < DZ3_WAVES) {
    dz_mma<DZ3_CH>(sD3 + 64 * DZ3_CH * (g % (BC3 / DZ3_CH)), sWz + (BRZ3 - BRZ0) * 64, lane, accz);
    dz_store(accz, dz_part, 0, DZ_S3 + g % (BC3 / DZ3_CH), g / (BC3 / DZ3_CH), row0, lane);
  }
  if (g < DZ2_WAVES) {
    dz_mma<DZ2_CH>(sD2 + 64 * DZ2_CH * (g % (BC2 / DZ2_CH)), sWz + (BRZ2 - BRZ0) * 64, lane, accz);
    dz_store(accz, dz_part, 0, DZ_S2 + g % (BC2 / DZ2_CH), g / (BC2 / DZ2_CH), row0, lane);
  }
  if (g < DZ1_WAVES) {
    dz_mma<DZ1_CH>(sD1 + 64 * DZ1_CH * (g % (BC1 / DZ1_CH)), sWz + (BRZ1 - BRZ0) * 64, lane, accz);
    dz_store(accz, dz_part, 0, DZ_S1 + g % (BC1 / DZ1_CH), g / (BC1 / DZ1_CH), row0, lane);
  }
  if (g < DZ0_WAVES) {
    dz_mma<DZ0_CH>(sD0 + 64 * DZ0_CH * (g % (BC0 / DZ0_CH)), sWz, lane, accz);
    dz_store(accz, dz_part, 0, DZ_S0 + g % (BC0 / DZ0_CH), g / (BC0 / DZ0_CH), row0, lane);
  }
}
// ONE forward layer as its ROLE of the pipelined kernels computes it (rollout_pipe.inc): the role's CUs of a team, every wave with its PFx_CG column
// groups -- role-ordered packing (pack_pipe_forward), pipe_mma, pipe_publish_all with the bias from LDS and the column bound of the last wave
template <int L>
__global__ void emu_pipe_layer_kernel(const float* Wreg_pf, const float* bias, const float* x_main, const float* z, unsigned char* xch, unsigned tag,
                                      float* slab, float* ht, int trow) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int CM = L == 0 ? P_XPAD : (L == 1 ? P_H0 : (L == 2 ? P_H1 : P_H2));
  constexpr int NOUT = L == 0 ? P_H0 : (L == 1 ? P_H1 : (L == 2 ? P_H2 : P_RAW));
  constexpr int M0 = L == 0 ? PR0_M0 : (L == 1 ? PR1_M0 : (L == 2 ? PR2_M0 : PR3_M0));
  constexpr int M1 = L == 0 ? PRG_M0 : (L == 1 ? PR2_M0 : (L == 2 ? PR3_M0 : TEAM_CUS));
  float* xs = smem;
  float* zs = xs + CM * 4;
  float* sBias = zs + P_ZD * 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = blockIdx.x;
  for (int i = tid; i < CM * 4; i += 256) xs[i] = x_main[i];
  for (int i = tid; i < P_ZD * 4; i += 256) zs[i] = z[i];
  for (int c = tid; c < 1024; c += 256) sBias[c] = bias[c];
  __syncthreads();
  if (m < M0 || m >= M1) return;
  const int gw = (m - M0) * 4 + wave;
  const float* wp = Wreg_pf + (size_t)(m * 4 + wave) * PF_NREG * 64 + lane;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xch, 0, 1 << 20, 0x00020000);
  if constexpr (L == 0) {
    float wa[PNWA], wv[PF0_NREG - PNWA];
    pipe_load_weights<PF0_NREG>(wp, wa, wv);
    float sums[PF0_CG][4];
    pipe_mma<NC0, NCZ, PF0_CG, 1>(xs, zs, wa, wv, lane, sums);
    pipe_publish_all<PF0_CG, true, 64>(sums, sBias, 4 * PF0_CG * gw, P_H0, xch, rs, 0u, tag, slab, trow, lane, ht);
  } else if constexpr (L == 1 || L == 2) {
    float wa[PNWA], wv[PF1_NREG - PNWA];
    pipe_load_weights<PF1_NREG>(wp, wa, wv);
    float sums[PF1_CG][4];
    pipe_mma<NC1, NCZ, PF1_CG, 2>(xs, zs, wa, wv, lane, sums);
    if constexpr (L == 1) pipe_publish_all<PF1_CG, true, 64>(sums, sBias, 4 * PF1_CG * gw, P_H1, xch, rs, 0u, tag, slab, trow, lane, ht);
    else pipe_publish_all<PF2_CG, true, 32>(sums, sBias, 4 * PF2_CG * gw, P_H2, xch, rs, 0u, tag, slab, trow, lane, ht);
  } else {
    float wa[PF3_NREG < PNWA ? PF3_NREG : PNWA], wv[PF3_NREG - PNWA > 0 ? PF3_NREG - PNWA : 1];
    pipe_load_weights<PF3_NREG>(wp, wa, wv);
    float sums[PF3_CG][4];
    pipe_mma<NC3, NCZ, PF3_CG, 1>(xs, zs, wa, wv, lane, sums);
    pipe_publish_all<PF3_CG, true, 0>(sums, sBias, 4 * PF3_CG * gw, P_RAW, xch, rs, 0u, tag, slab, trow, lane, nullptr);
  }
  (void)NOUT;
}
// One (step, group) of the pipelined ADJOINT's four layer roles (rollout_pipe.inc): every role CU takes its dh (already through the GroupNorm adjoint:
// tested separately) as the LDS operand, runs its transposed product (pack_pipe_backward, pipe_mma, pipe_publish_all into the adjoint's exchange
// layout QX_*) and its dL/dz partial products (LDS-resident vectors, pipe_dz_mma, pipe_dz_store)
__global__ void emu_pipe_layers_t_kernel(const float* Wreg_pb, const float* dh3, const float* dh2, const float* dh1, const float* dh0, unsigned char* xch,
                                         unsigned tag, float* dz_part, int trow) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = blockIdx.x;
  float* sWz = smem + P_H0 * 4 + wave * QB_NLW * 64;
  if (m >= PRG_M0 && m < PR1_M0) return;                       // (the glue adjoint's CU)
  const float* src = m >= PR3_M0 ? dh3 : (m >= PR2_M0 ? dh2 : (m >= PR1_M0 ? dh1 : dh0));
  const int nsrc = m >= PR3_M0 ? P_RAWPAD : (m >= PR2_M0 ? P_H2 : P_H0);
  for (int i = tid; i < P_H0 * 4; i += 256) xs[i] = i < nsrc * 4 ? src[i] : 0.f;
  const float* wp = Wreg_pb + (size_t)(m * 4 + wave) * (QB_NREG + QB_NLW) * 64 + lane;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xch, 0, 1 << 20, 0x00020000);
  const int nrows = 32;
  if (m >= PR3_M0) {
    const int gw = (m - PR3_M0) * 4 + wave;
    float wa[QB3_NREG], wv[1];
    pipe_load_weights<QB3_NREG>(wp, wa, wv);
    for (int r = 0; r < QZ3_N; ++r) sWz[r * 64 + lane] = wp[(size_t)(QB_NREG + r) * 64];
    __syncthreads();
    float sums[QB3_CG][4];
    pipe_mma<BC3, 0, QB3_CG, 1>(xs, xs, wa, wv, lane, sums);
    pipe_publish_all<QB3_CG, true, 32>(sums, nullptr, 4 * QB3_CG * gw, P_H2, xch, rs, QX_GA3, tag, nullptr, 0, lane, nullptr);
    if (gw < 6)
      for (int k = 0; k < 2; ++k) {
        float accz[4];
        pipe_dz_mma<BC3>(xs, sWz + k * BC3 * 64, lane, accz);
        pipe_dz_store(accz, dz_part, 0, QS_3, 2 * gw + k, nrows, trow, lane);
      }
  } else if (m >= PR1_M0) {
    const bool l2 = m >= PR2_M0;
    const int gw = (m - (l2 ? PR2_M0 : PR1_M0)) * 4 + wave;
    float wa[PNWA], wv[1];
    pipe_load_weights<PNWA>(wp, wa, wv);
    for (int r = 0; r < QZ2_N; ++r) sWz[r * 64 + lane] = wp[(size_t)(QB_NREG + r) * 64];
    __syncthreads();
    const int ndz = l2 ? 24 : 48, nsplit = l2 ? 2 : 4;
    if (l2) {
      float sums[QB2_CG][4];
      pipe_mma<BC2, 0, QB2_CG, 1>(xs, xs, wa, wv, lane, sums);
      pipe_publish_all<QB2_CG, true, 64>(sums, nullptr, 4 * QB2_CG * gw, P_H1, xch, rs, QX_GA2, tag, nullptr, 0, lane, nullptr);
    } else {
      float sums[QB1_CG][4];
      pipe_mma<BC1, 0, QB1_CG, 2>(xs, xs, wa, wv, lane, sums);
      pipe_publish_all<QB1_CG, true, 64>(sums, nullptr, 4 * QB1_CG * gw, P_H0, xch, rs, QX_GA1, tag, nullptr, 0, lane, nullptr);
    }
    if (gw < ndz) {
      float accz[4];
      pipe_dz_mma<16>(xs + 64 * 16 * (gw % nsplit), sWz, lane, accz);
      pipe_dz_store(accz, dz_part, 0, (l2 ? QS_2 : QS_1) + gw % nsplit, gw / nsplit, nrows, trow, lane);
    }
  } else {
    const int gw = (m - PR0_M0) * 4 + wave;
    float wa[PNWA], wv[QB0_NREG - PNWA];
    pipe_load_weights<QB0_NREG>(wp, wa, wv);
    for (int r = 0; r < QZ0_N; ++r) sWz[r * 64 + lane] = wp[(size_t)(QB_NREG + r) * 64];
    __syncthreads();
    const int zks = gw >> 2, zc0 = 3 * (gw & 3);
    float sums[QB0_CG][4];
    pipe_mma<BC0, 0, QB0_CG, 2>(xs, xs, wa, wv, lane, sums);
    pipe_publish_all<QB0_CG, true, 0>(sums, nullptr, 4 * QB0_CG * gw, P_XPAD, xch, rs, QX_GX0, tag, nullptr, 0, lane, nullptr);
    for (int k = 0; k < 3; ++k) {
      float accz[4];
      pipe_dz_mma<QZ0_CH>(xs + 64 * QZ0_CH * zks, sWz + k * QZ0_CH * 64, lane, accz);
      pipe_dz_store(accz, dz_part, 0, QS_0 + zks, zc0 + k, nrows, trow, lane);
    }
  }
}
}  // namespace ha

extern "C" int ha_emu_xslot(int group, int col) { return group == 64 ? ha::xslot<64>(col) : (group == 32 ? ha::xslot<32>(col) : ha::xslot<0>(col)); }
// group 64: a 1024-channel activation (layers 1 / 2), group 32: the 512-channel one (layer 3)
extern "C" int ha_emu_gather_norm(int group, const void* xch, unsigned tag, const float* gamma, const float* beta, float* xs_out, float* stats, int row0) {
  if (group == 64)
    hipLaunchKernelGGL((ha::emu_gather_norm_kernel<4, 64>), dim3(1), dim3(256), 0, nullptr, static_cast<const unsigned char*>(xch), tag, gamma, beta, xs_out, stats, row0);
  else if (group == 32)
    hipLaunchKernelGGL((ha::emu_gather_norm_kernel<2, 32>), dim3(1), dim3(256), 0, nullptr, static_cast<const unsigned char*>(xch), tag, gamma, beta, xs_out, stats, row0);
  else
    return HA_ERR_INVALID_ARG;
  return HA_OK;
}
extern "C" int ha_emu_gather_norm_bwd(int group, const void* xch, unsigned tag, const float* gamma, const float* beta, const float* ht, const float* stats,
                                      float* ds_out, int row0) {
  if (group == 64)
    hipLaunchKernelGGL((ha::emu_gather_norm_bwd_kernel<4, 64>), dim3(1), dim3(256), 0, nullptr, static_cast<const unsigned char*>(xch), tag, gamma, beta, ht, stats, ds_out, row0);
  else if (group == 32)
    hipLaunchKernelGGL((ha::emu_gather_norm_bwd_kernel<2, 32>), dim3(1), dim3(256), 0, nullptr, static_cast<const unsigned char*>(xch), tag, gamma, beta, ht, stats, ds_out, row0);
  else
    return HA_ERR_INVALID_ARG;
  return HA_OK;
}
extern "C" int ha_emu_lane_reduce(int which, const float* in, float* out) {
  hipLaunchKernelGGL(ha::emu_lane_reduce_kernel, dim3(1), dim3(64), 0, nullptr, which, in, out);
  return HA_OK;
}
// ncg 2 / group 64: 128 waves x 8 columns = a 1024-channel activation; ncg 1 / group 32: 128 x 4 = 512; ncg 1 / group 0: `waves` x 4 columns, identity slots
extern "C" int ha_emu_publish(int ncg, int group, int waves, const float* sums, const float* bias, void* xch, unsigned tag, float* slab, float* ht, int row0) {
  unsigned char* x = static_cast<unsigned char*>(xch);
  if (waves % 4 != 0) return HA_ERR_INVALID_ARG;
  if (ncg == 2 && group == 64) hipLaunchKernelGGL((ha::emu_publish_kernel<2, 64>), dim3(waves / 4), dim3(256), 0, nullptr, sums, bias, x, tag, slab, ht, row0);
  else if (ncg == 1 && group == 32) hipLaunchKernelGGL((ha::emu_publish_kernel<1, 32>), dim3(waves / 4), dim3(256), 0, nullptr, sums, bias, x, tag, slab, ht, row0);
  else if (ncg == 1 && group == 0) hipLaunchKernelGGL((ha::emu_publish_kernel<1, 0>), dim3(waves / 4), dim3(256), 0, nullptr, sums, bias, x, tag, slab, ht, row0);
  else return HA_ERR_INVALID_ARG;
  return HA_OK;
}
// layer 0..3 of the default decoder: W [Nout][Cmain + 48] row-major, bias [Nout (padded to the waves' columns)], x_main [Cmain padded to 16][4 rows],
// z [48][4 rows]; outputs as ha_emu_publish
extern "C" int ha_emu_persist_layer(int layer, const float* W, const float* bias, const float* x_main, const float* z, void* xch, unsigned tag, float* slab,
                                    float* ht, int row0) {
  if (layer < 0 || layer > 3) return HA_ERR_INVALID_ARG;
  std::vector<float> wr((size_t)ha::NWAVES_TEAM * ha::NREG * 64, 0.f);
  ha::pack_forward_layer(layer, W, wr);
  unsigned char* x = static_cast<unsigned char*>(xch);
  const float* wr_p = wr.data();
  if (layer == 0) hipLaunchKernelGGL(ha::emu_layer_kernel<0>, dim3(32), dim3(256), 0, nullptr, wr_p, bias, x_main, z, x, tag, slab, ht, row0);
  else if (layer == 1) hipLaunchKernelGGL(ha::emu_layer_kernel<1>, dim3(32), dim3(256), 0, nullptr, wr_p, bias, x_main, z, x, tag, slab, ht, row0);
  else if (layer == 2) hipLaunchKernelGGL(ha::emu_layer_kernel<2>, dim3(32), dim3(256), 0, nullptr, wr_p, bias, x_main, z, x, tag, slab, ht, row0);
  else hipLaunchKernelGGL(ha::emu_layer_kernel<3>, dim3(32), dim3(256), 0, nullptr, wr_p, bias, x_main, z, x, tag, slab, ht, row0);
  return HA_OK;
}
// transposed layer 3 / 2 / 1 / 0 of the adjoint (0: identity slots, 352 channels of which 339 live): w = the four forward weight matrices, dh [forward output channels of the layer, padded to 16][4 rows];
// the result arrives in the exchange region (slot map of the consumer's GroupNorm width: 32 for layer 3, 64 for layers 2 and 1)
extern "C" int ha_emu_persist_layer_t(int layer, const float* w0, const float* w1, const float* w2, const float* w3, const float* dh, void* xch, unsigned tag,
                                      int row0) {
  if (layer < 0 || layer > 3) return HA_ERR_INVALID_ARG;
  const float* w[4] = {w0, w1, w2, w3};
  std::vector<float> wb((size_t)ha::NWAVES_TEAM * (ha::NREG_B_ALL + ha::BC0_LDS) * 64, 0.f);
  ha::pack_backward(w, wb);
  unsigned char* x = static_cast<unsigned char*>(xch);
  const float* wb_p = wb.data();
  if (layer == 3) hipLaunchKernelGGL(ha::emu_layer_t_kernel<3>, dim3(32), dim3(256), 0, nullptr, wb_p, dh, x, tag, row0);
  else if (layer == 2) hipLaunchKernelGGL(ha::emu_layer_t_kernel<2>, dim3(32), dim3(256), 0, nullptr, wb_p, dh, x, tag, row0);
  else if (layer == 1) hipLaunchKernelGGL(ha::emu_layer_t_kernel<1>, dim3(32), dim3(256), 0, nullptr, wb_p, dh, x, tag, row0);
  else hipLaunchKernelGGL(ha::emu_layer_t_kernel<0>, dim3(32), dim3(256), 0, nullptr, wb_p, dh, x, tag, row0);
  return HA_OK;
}
// dL/dz of ONE step for the team's four rows (sequences row0 .. row0 + 3 of B = 32): g_z [32][1][48]
extern "C" int ha_emu_persist_dz(const float* w0, const float* w1, const float* w2, const float* w3, const float* dh3, const float* dh2, const float* dh1,
                                 const float* dh0, float* g_z, int row0) {
  const float* w[4] = {w0, w1, w2, w3};
  std::vector<float> wb((size_t)ha::NWAVES_TEAM * (ha::NREG_B_ALL + ha::BC0_LDS) * 64, 0.f);
  ha::pack_backward(w, wb);
  std::vector<float> part((size_t)ha::DZ_SLOTS * 32 * ha::P_ZD, 0.f);
  const float* wb_p = wb.data();
  float* part_p = part.data();
  hipLaunchKernelGGL(ha::emu_dz_kernel, dim3(32), dim3(256), 0, nullptr, wb_p, dh3, dh2, dh1, dh0, part_p, row0);
  const int n = 32 * ha::P_ZD;
  hipLaunchKernelGGL(ha::dz_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, (const float*)part_p, g_z, (const float*)nullptr, 32, 1);
  return HA_OK;
}
// the same layer test for the ROLES of the pipelined kernels (33 .. 256 sequences): w = the four forward weight matrices, bias [1024] of the layer
extern "C" int ha_emu_pipe_layer(int layer, const float* w0, const float* w1, const float* w2, const float* w3, const float* bias, const float* x_main,
                                 const float* z, void* xch, unsigned tag, float* slab, float* ht, int trow) {
  if (layer < 0 || layer > 3) return HA_ERR_INVALID_ARG;
  const float* w[4] = {w0, w1, w2, w3};
  std::vector<float> wp((size_t)ha::NWAVES_TEAM * ha::PF_NREG * 64, 0.f);
  ha::pack_pipe_forward(w, wp);
  unsigned char* x = static_cast<unsigned char*>(xch);
  const float* wp_p = wp.data();
  if (layer == 0) hipLaunchKernelGGL(ha::emu_pipe_layer_kernel<0>, dim3(32), dim3(256), 0, nullptr, wp_p, bias, x_main, z, x, tag, slab, ht, trow);
  else if (layer == 1) hipLaunchKernelGGL(ha::emu_pipe_layer_kernel<1>, dim3(32), dim3(256), 0, nullptr, wp_p, bias, x_main, z, x, tag, slab, ht, trow);
  else if (layer == 2) hipLaunchKernelGGL(ha::emu_pipe_layer_kernel<2>, dim3(32), dim3(256), 0, nullptr, wp_p, bias, x_main, z, x, tag, slab, ht, trow);
  else hipLaunchKernelGGL(ha::emu_pipe_layer_kernel<3>, dim3(32), dim3(256), 0, nullptr, wp_p, bias, x_main, z, x, tag, slab, ht, trow);
  return HA_OK;
}
// the four transposed layer roles of the pipelined adjoint for one (step, group): exchange region in the adjoint's layout (returned offsets: bytes of
// dL/da3, dL/da2, dL/da1, dL/dx), g_z [32][48] for the team's rows trow .. trow + 3
extern "C" int ha_emu_pipe_layers_t(const float* w0, const float* w1, const float* w2, const float* w3, const float* dh3, const float* dh2, const float* dh1,
                                    const float* dh0, void* xch, unsigned tag, float* g_z, int trow, unsigned* offsets) {
  const float* w[4] = {w0, w1, w2, w3};
  std::vector<float> wq((size_t)ha::NWAVES_TEAM * (ha::QB_NREG + ha::QB_NLW) * 64, 0.f);
  ha::pack_pipe_backward(w, wq);
  std::vector<float> part((size_t)ha::QS_SLOTS * 32 * ha::P_ZD, 0.f);
  const float* wq_p = wq.data();
  float* part_p = part.data();
  unsigned char* x = static_cast<unsigned char*>(xch);
  hipLaunchKernelGGL(ha::emu_pipe_layers_t_kernel, dim3(32), dim3(256), 0, nullptr, wq_p, dh3, dh2, dh1, dh0, x, tag, part_p, trow);
  const int n = 32 * ha::P_ZD;
  hipLaunchKernelGGL(ha::pipe_dz_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, (const float*)part_p, g_z, (const float*)nullptr, 32, 1, 32);
  offsets[0] = ha::QX_GA3; offsets[1] = ha::QX_GA2; offsets[2] = ha::QX_GA1; offsets[3] = ha::QX_GX0;
  return HA_OK;
}

// test knobs of the whole-kernel hooks: the failure injection of ha_tune_set("rollout_persist_inject") (member 3 of team 0 leaves at once) and the
// bound of the kernels' waits in polls (a poll sleeps 20 ms on the emulator: 1500 polls = 30 s, far more than a healthy hand-off takes at these sizes)
static int g_emu_inject = 0;
extern "C" int ha_emu_persist_knobs(int inject, int spin_limit) {
  g_emu_inject = inject;
  ha::SPIN_LIMIT = spin_limit > 0 ? spin_limit : 40000;
  return HA_OK;
}

static int emu_persist_team(int B, int S, const float* const* w, const float* const* bs, const float* const* gam, const float* const* bet,
                            const float* past_in0, const float* z_seq, float* world, float* xT, float* raw, unsigned* err,
                            const float* g_world, float* g_past, float* g_z, unsigned* err_bwd) {
  using namespace ha;
  if (B < 1 || B > ROWS * NTEAMS || S < 1) return HA_ERR_INVALID_ARG;
  const int nteams = (B + ROWS - 1) / ROWS;          // one resident team per four sequences (team = block index / 32 on the emulator)
  std::vector<float> wr((size_t)NWAVES_TEAM * NREG * 64, 0.f);
  for (int l = 0; l < 4; ++l) pack_forward_layer(l, w[l], wr);
  const int bpad[4] = {P_H0, P_H1, P_H2, P_RAWPAD}, nout[4] = {P_H0, P_H1, P_H2, P_RAW};
  std::vector<float> bias[4];
  for (int l = 0; l < 4; ++l) {
    bias[l].assign(bpad[l], 0.f);
    for (int i = 0; i < nout[l]; ++i) bias[l][i] = bs[l][i];
  }
  // a private stash: per step [G 32 x 12 | slabs 1024, 1024, 512, 224 x 32 | statistics 3 x 16 x 32 x 2 | glue record 32 x 32 | team layout 32 x (1024, 1024, 512)]
  size_t o = 0;
  auto take = [&](size_t n) { const size_t r = o; o += (n + 63) / 64 * 64; return r; };
  PersistArgs a;
  memset(&a, 0, sizeof(a));
  a.off_G = take(32 * 12);
  for (int l = 0; l < 4; ++l) a.off_dec[l] = take((size_t)bpad[l] * 32);
  for (int l = 0; l < 3; ++l) a.off_gn[l] = take(16 * 32 * 2);
  a.off_gl = take(32 * 32);
  for (int l = 0; l < 3; ++l) a.off_ht[l] = take((size_t)32 * nout[l]);
  a.per_step = o;
  const unsigned nanbits = 0x7fc00000u;
  float nanv;
  memcpy(&nanv, &nanbits, 4);
  // (every buffer the kernels write starts NaN-filled, like the GPU tier's poisoned allocations)
  std::vector<float> steps((size_t)(S + 1) * a.per_step, nanv), xTv((size_t)(S + 1) * P_DINP * 32, nanv), t2j(32 * 3, nanv);
  std::vector<unsigned char> xch(XCH_BYTES, 0);
  std::vector<float> worldv((size_t)B * S * P_STATE, nanv);
  unsigned errw = 0;
  a.B = B; a.S = S;
  a.Wreg = wr.data();
  for (int l = 0; l < 4; ++l) a.bias[l] = bias[l].data();
  for (int l = 0; l < 3; ++l) { a.gamma[l] = gam[l]; a.beta[l] = bet[l]; }
  a.past_in0 = past_in0; a.z_seq = z_seq; a.world = worldv.data(); a.xT = xTv.data(); a.steps = steps.data();
  a.t2j = t2j.data();
  a.xch = xch.data();
  a.err = &errw;
  a.hidden_slabs = 1;
  a.inject = g_emu_inject;
  simt_emu::g_resident_blocks = TEAM_CUS * nteams;
  hipLaunchKernelGGL(rollout_persist_fwd_kernel<false>, dim3(TEAM_CUS * nteams), dim3(256), 0, nullptr, a);
  simt_emu::g_resident_blocks = 0;
  memcpy(world, worldv.data(), worldv.size() * sizeof(float));
  memcpy(xT, xTv.data(), xTv.size() * sizeof(float));
  for (int t = 0; t < S; ++t) memcpy(raw + (size_t)t * P_RAWPAD * 32, steps.data() + (size_t)t * a.per_step + a.off_dec[3], (size_t)P_RAWPAD * 32 * sizeof(float));
  *err = errw;
  if (!g_world) return HA_OK;
  // ---- the adjoint over the forward's stash (no prior part: gx_pri = null) -----------------------------------------------------------------
  std::vector<float> wb((size_t)NWAVES_TEAM * (NREG_B_ALL + BC0_LDS) * 64, 0.f);
  pack_backward(w, wb);
  PersistBwdArgs q;
  memset(&q, 0, sizeof(q));
  q.B = B; q.S = S;
  q.Wreg = wb.data();
  for (int l = 0; l < 3; ++l) { q.gamma[l] = gam[l]; q.beta[l] = bet[l]; }
  q.g_world = g_world; q.gx_pri = nullptr; q.gxp_pad = 0;
  q.xT = xTv.data(); q.steps = steps.data(); q.per_step = a.per_step; q.off_G = a.off_G;
  for (int l = 0; l < 4; ++l) q.off_dec[l] = a.off_dec[l];
  for (int l = 0; l < 3; ++l) { q.off_gn[l] = a.off_gn[l]; q.off_ht[l] = a.off_ht[l]; }
  q.off_gl = a.off_gl;
  q.t2j = t2j.data();
  std::vector<float> gpast((size_t)B * P_DIN, nanv), dzp((size_t)S * DZ_SLOTS * 32 * P_ZD, nanv), gz((size_t)B * S * P_ZD, nanv);
  q.g_past0 = gpast.data(); q.dz_part = dzp.data();
  std::fill(xch.begin(), xch.end(), (unsigned char)0);
  q.xch = xch.data();
  unsigned errb = 0;
  q.err = &errb;
  simt_emu::g_resident_blocks = TEAM_CUS * nteams;
  hipLaunchKernelGGL(rollout_persist_bwd_kernel<false>, dim3(TEAM_CUS * nteams), dim3(256), 0, nullptr, q);
  simt_emu::g_resident_blocks = 0;
  const int n = B * S * P_ZD;
  hipLaunchKernelGGL(dz_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, (const float*)dzp.data(), gz.data(), (const float*)nullptr, B, S);
  memcpy(g_past, gpast.data(), gpast.size() * sizeof(float));
  memcpy(g_z, gz.data(), gz.size() * sizeof(float));
  *err_bwd = errb;
  return HA_OK;
}

// ---- the WHOLE persistent kernels (one team per four sequences) with every team's 32 blocks resident at the same time ---------------------------------------
// Forward: team formation from the block index, the step loop, every exchange hand-off through tagged granules, GroupNorm on the consumer side, the
// glue chains, copy_out -- every block on its own NaN-filled LDS, every output buffer NaN-filled.  w / b: the decoder's four Linear layers
// ([out][in] row-major, biases), g / be: the GroupNorm affines of layers 1 .. 3.  Outputs: world [B][S][348], xT [(S + 1)][340][32] (the state
// slabs the prior network reads: quad layout), raw [S][224][32] (decoder outputs, quad layout), err (the kernel's error word).
// With g_world [B][S][348] != null the one-launch ADJOINT runs over the forward's stash: g_past [B][339], g_z [B][S][48], err_bwd.
extern "C" int ha_emu_persist_team(int B, int S, const float* w0, const float* w1, const float* w2, const float* w3, const float* b0, const float* b1,
                                   const float* b2, const float* b3, const float* g1, const float* be1, const float* g2, const float* be2,
                                   const float* g3, const float* be3, const float* past_in0, const float* z_seq, float* world, float* xT, float* raw,
                                   unsigned* err, const float* g_world, float* g_past, float* g_z, unsigned* err_bwd) {
  const float* w[4] = {w0, w1, w2, w3};
  const float* bs[4] = {b0, b1, b2, b3};
  const float* gam[3] = {g1, g2, g3};
  const float* bet[3] = {be1, be2, be3};
  return emu_persist_team(B, S, w, bs, gam, bet, past_in0, z_seq, world, xT, raw, err, g_world, g_past, g_z, err_bwd);
}

// ---- the WHOLE pipelined kernels (32 < B <= 256) for ONE team: rows 4 team .. 4 team + 3 of every 32-row tile, i.e. with team 0 resident the
// sequences 32 g + 0 .. 3 of the NG = ceil(B / 32) tiles (the other teams' rows stay NaN in the outputs and are not compared).  Layer roles 5 / 16 /
// 8 / 2 CUs + the glue CU, groups flowing through them, forward and one-launch adjoint.  Buffers: world [B][S][348], xT [(S + 1)][NG][340][32],
// g_past [B][339], g_z [B][S][48].
extern "C" int ha_emu_pipe_team(int B, int S, const float* w0, const float* w1, const float* w2, const float* w3, const float* b0, const float* b1,
                                const float* b2, const float* b3, const float* g1, const float* be1, const float* g2, const float* be2,
                                const float* g3, const float* be3, const float* past_in0, const float* z_seq, float* world, float* xT,
                                unsigned* err, const float* g_world, float* g_past, float* g_z, unsigned* err_bwd) {
  using namespace ha;
  if (B <= NTEAMS * ROWS || B > 32 * PG_MAX || S < 1) return HA_ERR_INVALID_ARG;
  const float* w[4] = {w0, w1, w2, w3};
  const float* bs[4] = {b0, b1, b2, b3};
  const float* gam[3] = {g1, g2, g3};
  const float* bet[3] = {be1, be2, be3};
  const int NG = (B + 31) / 32;
  std::vector<float> wp((size_t)NWAVES_TEAM * PF_NREG * 64, 0.f);
  pack_pipe_forward(w, wp);
  const int bpad[4] = {P_H0, P_H1, P_H2, P_RAWPAD}, nout[4] = {P_H0, P_H1, P_H2, P_RAW};
  std::vector<float> bias[4];
  for (int l = 0; l < 4; ++l) {
    bias[l].assign(bpad[l], 0.f);
    for (int i = 0; i < nout[l]; ++i) bias[l][i] = bs[l][i];
  }
  size_t o = 0;
  auto take = [&](size_t n) { const size_t r = o; o += (n + 63) / 64 * 64; return r; };
  PipeArgs a;
  memset(&a, 0, sizeof(a));
  const size_t RT = NG;
  a.off_G = take(RT * 32 * 12);
  for (int l = 0; l < 4; ++l) { a.off_dec[l] = take(RT * bpad[l] * 32); a.dec_pad[l] = bpad[l]; }
  for (int l = 0; l < 3; ++l) a.off_gn[l] = take(RT * 16 * 32 * 2);
  a.off_gl = take(RT * 32 * 32);
  for (int l = 0; l < 3; ++l) a.off_ht[l] = take(RT * 32 * nout[l]);
  a.per_step = o;
  const unsigned nanbits = 0x7fc00000u;
  float nanv;
  memcpy(&nanv, &nanbits, 4);
  std::vector<float> steps((size_t)(S + 1) * a.per_step, nanv), xTv((size_t)(S + 1) * NG * P_DINP * 32, nanv), t2j((size_t)NG * 32 * 3, nanv);
  std::vector<unsigned char> xch(PX_BYTES, 0);
  std::vector<float> worldv((size_t)B * S * P_STATE, nanv);
  unsigned errw = 0;
  a.B = B; a.S = S; a.NG = NG;
  a.Wreg = wp.data();
  for (int l = 0; l < 4; ++l) a.bias[l] = bias[l].data();
  for (int l = 0; l < 3; ++l) { a.gamma[l] = gam[l]; a.beta[l] = bet[l]; }
  a.past_in0 = past_in0; a.z_seq = z_seq; a.world = worldv.data(); a.xT = xTv.data(); a.steps = steps.data();
  a.t2j = t2j.data();
  a.hidden_slabs = 1;
  a.inject = g_emu_inject;
  a.xch = xch.data();
  a.err = &errw;
  simt_emu::g_resident_blocks = TEAM_CUS;
  hipLaunchKernelGGL(rollout_pipe_fwd_kernel<false>, dim3(TEAM_CUS), dim3(256), 0, nullptr, a);
  simt_emu::g_resident_blocks = 0;
  memcpy(world, worldv.data(), worldv.size() * sizeof(float));
  memcpy(xT, xTv.data(), xTv.size() * sizeof(float));
  *err = errw;
  if (!g_world) return HA_OK;
  std::vector<float> wq((size_t)NWAVES_TEAM * (QB_NREG + QB_NLW) * 64, 0.f);
  pack_pipe_backward(w, wq);
  PipeBwdArgs q;
  memset(&q, 0, sizeof(q));
  q.B = B; q.S = S; q.NG = NG;
  q.Wreg = wq.data();
  for (int l = 0; l < 3; ++l) { q.gamma[l] = gam[l]; q.beta[l] = bet[l]; }
  q.g_world = g_world; q.gx_pri = nullptr; q.gxp_pad = 0;
  q.xT = xTv.data(); q.steps = steps.data(); q.per_step = a.per_step; q.off_G = a.off_G; q.off_gl = a.off_gl;
  for (int l = 0; l < 4; ++l) { q.off_dec[l] = a.off_dec[l]; q.dec_pad[l] = a.dec_pad[l]; }
  for (int l = 0; l < 3; ++l) { q.off_gn[l] = a.off_gn[l]; q.off_ht[l] = a.off_ht[l]; }
  q.t2j = t2j.data();
  std::vector<float> gpast((size_t)B * P_DIN, nanv), dzp((size_t)S * QS_SLOTS * NG * 32 * P_ZD, nanv), gz((size_t)B * S * P_ZD, nanv);
  q.g_past0 = gpast.data(); q.dz_part = dzp.data();
  std::fill(xch.begin(), xch.end(), (unsigned char)0);
  q.xch = xch.data();
  unsigned errb = 0;
  q.err = &errb;
  simt_emu::g_resident_blocks = TEAM_CUS;
  hipLaunchKernelGGL(rollout_pipe_bwd_kernel<false>, dim3(TEAM_CUS), dim3(256), 0, nullptr, q);
  simt_emu::g_resident_blocks = 0;
  const int n = B * S * P_ZD;
  hipLaunchKernelGGL(pipe_dz_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, (const float*)dzp.data(), gz.data(), (const float*)nullptr, B, S, NG * 32);
  memcpy(g_past, gpast.data(), gpast.size() * sizeof(float));
  memcpy(g_z, gz.data(), gz.size() * sizeof(float));
  *err_bwd = errb;
  return HA_OK;
}


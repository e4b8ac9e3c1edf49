// HuMoR CVAE roll-out on gfx950: packed-weight fp32-MFMA MLP layers with fused GroupNorm+ReLU prologues (forward)
// and fused GroupNorm/ReLU adjoint prologues (backward), per-step "glue" kernels (residual composition with Rodrigues,
// heading alignment, both frame changes, world-transform accumulation) forward and backward, and the rotation
// conversion kernels.  Replaces the per-step chain of ~120 ATen launches of HumorModel.roll_out
// (humor/models/humor_model.py:785-1017; prior :407-418, decode :445-498, MLP :1206-1241,
// apply_world2local_trans :696-772; compute_world2aligned_mat humor/utils/transforms.py:17-42).
//
// Data layout: every activation lives transposed in 32-row tiles with the channels interleaved in quads,
// X^T[tile][channel/4][32 rows][4], so that the A operand of v_mfma_f32_32x32x2_f32 (lane l <-> row l&31, k = l>>5)
// is read with eight 16-byte loads per lane and 64-channel slice (the layer is a pure latency chain: a wave can keep
// only 63 vector-memory instructions in flight, so 4-byte loads -- 32 per slab and operand -- serialised into several
// memory round trips; in-kernel timestamps: tools/layer_timing.py).  Lane (row, hi) of a slice owns its channels
// cbase + 32*hi + kp, kp = 0..31; weights are re-packed once into the matching B-operand order.  A wave owns a
// 64-channel K-slice (one or two GroupNorm groups: statistics are lane-local, plus one cross-half shuffle for 64-channel
// groups); NW waves of a block split K further and reduce through LDS; blocks split K across the chip and the consumer
// layer's prologue sums those partial slabs (the launch-boundary reduce).
#include <string.h>

#include <algorithm>
#include <unordered_map>
#include <vector>

#include "rollout_persist.h"
#include "rot_math.h"

namespace ha {

constexpr int D_IN = 339, D_STATE = 348, ZD = 48, NJT = 22, NBODY = 21;
// Decoder output ("raw") layout for an output rotation representation of ROTW floats per joint (3: axis-angle, 6: 6-D, 9: 9-D;
// humor_model.py:100-140 output_dim_list): trans 3 | trans_vel 3 | root_orient ROTW | root_orient_vel 3 | pose_body 21 x ROTW |
// joints 66 | joints_vel 66 | contacts 9
template <int ROTW>
struct RawLayout {
  static constexpr int ROOT = 6, RVEL = 6 + ROTW, BODY = 9 + ROTW, JNT = 9 + 22 * ROTW, JVEL = JNT + 66, CONT = JVEL + 66, D = CONT + 9;
};
static_assert(RawLayout<3>::D == 216 && RawLayout<6>::D == 282 && RawLayout<9>::D == 348, "decoder output widths of the three representations");
constexpr int D_INP = 340;       // D_IN rounded up to a channel quad (the pad channel of a state slab is kept at zero)
constexpr int SLICE = 64;        // channels per wave K-slice
constexpr int MAXL = 8;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float vf4 __attribute__((ext_vector_type(4)));

// offset of channel c inside one row's view of a quad-interleaved tile: element (c, row) of tile rt of a C-channel slab
// (C % 4 == 0) is slab[rt * C * 32 + row * 4 + qoff(c)]
__device__ __forceinline__ size_t qoff(int c) { return (size_t)(c >> 2) * 128 + (c & 3); }

// layer-kernel launch policy picked up by ha_humor_net_create (ha_tune_set "layer_spb" / "layer_nw"; 0 = default)
int g_layer_spb = 0, g_layer_nw = 0, g_layer_finish = 1;
int g_layer_hsum = 1;  // summed pre-activation write-back for the adjoint (ha_tune_set "layer_hsum")
// ha_tune_set "layer_acc" (experiment, default off): one-row-tile policy only.  1: the K-split blocks of a decoder layer ADD their
// partial tiles into ONE pre-zeroed slab with hardware fp32 atomics (global_atomic_add_f32) instead of writing nsplit partial
// slabs, so that every consumer block reads one slab instead of up to five.  Measured (profiles/experiments/README.md, round 2):
// the consumers' load phase shrinks by 2-3 k cycles as predicted, but 1024 atomics per block drain in 2.5-5 k cycles and delay the
// next launch: 32x59 fwd+bwd 4.55 -> 4.43 ms only, and the sum order of the partials (hence the last bit) varies run to run.
int g_layer_acc = 0;
int g_gemm_rm = 0;    // row tiles per wave of the batched prior GEMM (ha_tune_set "gemm_rm"; 0 = by size)
// ha_tune_set "gemm_ks" (default 2): two waves share a tile pair, each walks half of K, partial tiles summed through LDS before the
// epilogue -- for GEMMs that leave most SIMDs idle (VPoser: 1920 x 512 x 512 is 480 waves on 1024 SIMDs, a frame-0 decode 16 waves:
// a launch is exactly one wave's serial MFMA chain).  0 = off, 3 = also for large GEMMs (measured slower: tools/gemm_ks_ab.py).
int g_gemm_ks = 2;
// ha_tune_set "rollout_groups": 0 = auto, n >= 1 = split the batch into (at most) n row groups that run the chain side by side on
// their own HIP streams (fork / join on the caller's stream with events; capturable), each with its own stash region and its steps
// issued round-robin.  Sequences are independent, but side-by-side chains only pay where they change the launch policy: dispatch is
// the bound (host ~3.5 us per launch eager; hipGraph replay of 2 / 4 groups at 256 x 119 is no faster either -- profiles/experiments/
// README.md), so auto uses two groups only for exactly two row tiles (each then runs the one-row-tile policy: 64 x 59 fwd+bwd
// 5.97 -> 5.15 ms) and one group otherwise.  Must not change between a forward call and its backward.
int g_rollout_groups = 0;
// ha_tune_set "rollout_persist": the forward of a roll-out of <= 32 sequences as ONE persistent launch with the decoder resident in
// the register files of the XCD teams (rollout_persist.hip) instead of 5 dependent launches per step.  0 = launch chain, 1 = on
// (default; granules published with plain stores: team = one XCD = one L2), 3 = on, granules published write-through (sc1).
// Measured at 32 x 59 (profiles/r03_run1): decoder chain 30.1 -> 15.5 us per step (variant 3: 21.0), forward 1.96 -> 1.09 ms.
int g_rollout_persist = 1;
// ha_tune_set "rollout_persist_bwd": 1 (default) = behind a persistent forward the adjoint is ONE persistent launch too (reverse scan,
// transposed weights resident, rollout_persist.hip); 0 = the launch-chain adjoint reads the persistent forward's stash.
int g_rollout_persist_bwd = 1;
// ha_tune_set "rollout_pipe": 1 (default) = 32 < B <= 256 sequences run the layer-parallel pipelined persistent kernels (rollout_pipe.inc);
// 0 = the launch chain (A/B runs, tests)
int g_rollout_pipe = 1;
// ha_tune_set "rollout_pipe_bwd": 1 = behind a pipelined forward the adjoint is the pipelined persistent launch too; 0 = the launch chain's adjoint
int g_rollout_pipe_bwd = 1;
// ha_tune_set "rollout_persist_inject": test hook -- 1 = the persistent forward drops one CU of team 0, so that the team's bounded waits run
// out and the failure path (NaN results, host-mapped error word, fall-back to the launch chain) can be exercised on a healthy GPU
int g_rollout_persist_inject = 0;

struct PackedLayer {
  int Cin = 0, skip = 0, Nout = 0;
  int nslices_f = 0, main_slices = 0, ntiles_f = 0, Nout_pad = 0;
  int nslices_b = 0, ntiles_b = 0, Nin_pad = 0;
  int group = 0;
  float* Wf = nullptr;   // [ntiles_f][nslices_f][32][64]
  float* Wb = nullptr;   // [ntiles_b][nslices_b][32][64]
  float* bias = nullptr;
  float* gamma = nullptr;  // GroupNorm affine of the activation feeding this layer (Cin channels)
  float* beta = nullptr;
};

}  // namespace ha

struct ha_humor_net {
  int device = 0;
  int n_dec = 0, n_pri = 0;
  int rotw = 3;                            // floats per joint of the decoder's output rotation representation (3 aa, 6 6-D, 9 9-D)
  bool delta = true;                       // the decoder emits residuals (HumorModel(output_delta=True)); ha_humor_net_set_option("output_delta")
  ha::PackedLayer dec[ha::MAXL], pri[ha::MAXL];
  ha::PersistNet* persist = nullptr;      // register-stationary decoder for the persistent forward (null: shape / device not eligible)
  // The stash layout of a call depends on whether the persistent kernels serve it, which in turn depends on mutable state (the tune
  // knob, the asynchronous error word).  A forward call decides ONCE and records the decision for the stash it fills; every later
  // phase of that call and the backward over the same stash use the recorded mode (one host thread per device: no lock).
  struct StashRec { int mode, B, S, knobs; };
  mutable std::unordered_map<const void*, StashRec> stash_mode;
};

namespace ha {

// One (layer, direction) unit of work inside a launch.
struct LayerTask {
  const float* Wp; const float* bias;
  int ntiles, nslices, main_slices, Nout, Nout_pad;
  const float* src; int nsplit_src; int Csrc;       // A operand main part: partial slabs [nsplit][RT][Csrc][32]
  const float* skip; int skip_dim;                  // raw tail part [RT][skip_dim][32]
  int mode;                                         // 0 raw, 1 GN+ReLU (fwd), 3 GN+ReLU adjoint (bwd)
  const float* gamma; const float* beta; int group; float inv_group;
  const float* hsrc; int nsplit_h; int Ch;          // mode 3: forward pre-activation slabs of the same channels ([..][Ch][32])
  float* dst;                                       // [nsplit_dst][RT][Nout_pad][32]
  float* hsum_dst;                                  // mode 1: the summed pre-activations of the source go back as ONE slab [RT][Csrc][32] (or null)
  int spb, nsplit_dst;                              // K-slices per block, ceil(nslices / spb)
  int acc;                                          // 1: all K-splits add into slab 0 of dst (pre-zeroed) with fp32 atomics
  int nblocks;                                      // ntiles * nsplit_dst * RT
};

// One launch = one (layer, direction).  (A single task: the kernel's start-up is a chain of dependent scalar loads, and
// indexing an array of tasks by a block-dependent index cost ~15 of them -- 1.3 us per launch on a 4-7 us launch.)
struct LayerLaunch {
  LayerTask t[1];
  int ntasks;
  int RT;
};

constexpr int MAXSPLIT = 5;   // K-slices / NW never exceeds this for the supported layer widths (K <= 1280)

// sums `nsplit` (<= MAXSPLIT) partial slabs of element (channel c, row) of tile rt.  All loads are unconditional (clamped
// slab index, zero weight beyond nsplit) so they issue back-to-back instead of one round trip per slab.
__device__ __forceinline__ float slab_sum(const float* base, int nsplit, int RT, int C, int rt, int c, int row) {
  const size_t stride = (size_t)RT * C * 32;
  const float* p = base + (size_t)rt * C * 32 + (size_t)row * 4 + qoff(c);
  float t[MAXSPLIT];
#pragma unroll
  for (int s = 0; s < MAXSPLIT; ++s) t[s] = p[(size_t)(s < nsplit ? s : 0) * stride];
  float v = t[0];
#pragma unroll
  for (int s = 1; s < MAXSPLIT; ++s) v += s < nsplit ? t[s] : 0.f;
  return v;
}

// A wave's 64-channel x 32-row fragment of a slab stack: lane (row, hi) reads channels cbase + 32 hi + kp as eight quads.
// All NS x 8 16-byte loads are issued back-to-back (independent) and summed in split order.  nq = number of quads that
// exist (the tail slice of a 339- or 48-channel operand); missing quads read quad 0 and are zeroed.
template <int NS>
__device__ __forceinline__ void load_frag(const float* p, size_t stride, int nq, float (&a)[32]) {
  vf4 part[NS][8];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) part[s][j] = *reinterpret_cast<const vf4*>(p + (size_t)s * stride + (size_t)(j < nq ? j : 0) * 128);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    vf4 v = part[0][j];
#pragma unroll
    for (int s = 1; s < NS; ++s) v += part[s][j];
    if (j >= nq) v = vf4{0.f, 0.f, 0.f, 0.f};
    a[4 * j] = v.x; a[4 * j + 1] = v.y; a[4 * j + 2] = v.z; a[4 * j + 3] = v.w;
  }
}

template <bool SINGLE = false>
__device__ __forceinline__ void load_frag_n(const float* base, int nsplit, int RT, int C, int rt, int cbase, int lane, float (&a)[32]) {
  const size_t stride = (size_t)RT * C * 32;
  const int row = lane & 31, c0 = cbase + 32 * (lane >> 5);
  int nq = (C - c0) >> 2;
  nq = nq < 0 ? 0 : (nq > 8 ? 8 : nq);
  const int q0 = nq > 0 ? (c0 >> 2) : 0;
  const float* p = base + (size_t)rt * C * 32 + (size_t)q0 * 128 + (size_t)row * 4;
  if (SINGLE) { load_frag<1>(p, stride, nq, a); return; }
  switch (nsplit) {
    case 1: load_frag<1>(p, stride, nq, a); return;
    case 2: load_frag<2>(p, stride, nq, a); return;
    case 3: load_frag<3>(p, stride, nq, a); return;
    case 4: load_frag<4>(p, stride, nq, a); return;
    default: load_frag<MAXSPLIT>(p, stride, nq, a); return;
  }
}

// per-row GroupNorm statistics of the lane's 32 channels.  NG = 1: the 64-channel slice is one group (the other half lives
// in lane^32); NG = 2: each half-slice is its own 32-channel group, entirely lane-local.
template <int NG>
__device__ __forceinline__ void gn_stats(const float (&h)[32], float inv_n, float& mean, float& rstd) {
  // four interleaved partial sums: a single 32-long dependent add chain is ~32 x the VALU latency on the launch's critical path
  float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
  for (int kp = 0; kp < 32; kp += 4) { p0 += h[kp]; p1 += h[kp + 1]; p2 += h[kp + 2]; p3 += h[kp + 3]; }
  float s1 = (p0 + p1) + (p2 + p3);
  if (NG == 1) s1 += __shfl_xor(s1, 32);
  const float mu = s1 * inv_n;
  p0 = p1 = p2 = p3 = 0.f;
#pragma unroll
  for (int kp = 0; kp < 32; kp += 4) {
    const float d0 = h[kp] - mu, d1 = h[kp + 1] - mu, d2 = h[kp + 2] - mu, d3 = h[kp + 3] - mu;
    p0 = fmaf(d0, d0, p0); p1 = fmaf(d1, d1, p1); p2 = fmaf(d2, d2, p2); p3 = fmaf(d3, d3, p3);
  }
  float s2 = (p0 + p1) + (p2 + p3);
  if (NG == 1) s2 += __shfl_xor(s2, 32);
  mean = mu;
  rstd = rsqrtf(s2 * inv_n + 1e-5f);
}

template <int NG>
__device__ __forceinline__ void gn_apply(int mode, const float (&gam)[32], const float (&bet)[32], float inv_n,
                                         const float (&h)[32], float (&a)[32]) {
  float mean, rstd;
  gn_stats<NG>(h, inv_n, mean, rstd);
  if (mode == 1) {
#pragma unroll
    for (int kp = 0; kp < 32; ++kp) a[kp] = fmaxf((h[kp] - mean) * rstd * gam[kp] + bet[kp], 0.f);
  } else {
    // adjoint: a holds da; through ReLU and GroupNorm -> dh
    float dxh[32], xh[32];
    float m1p[4] = {0.f, 0.f, 0.f, 0.f}, m2p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kp = 0; kp < 32; ++kp) {
      xh[kp] = (h[kp] - mean) * rstd;
      const float y = xh[kp] * gam[kp] + bet[kp];
      dxh[kp] = (y > 0.f ? a[kp] : 0.f) * gam[kp];
      m1p[kp & 3] += dxh[kp];
      m2p[kp & 3] = fmaf(dxh[kp], xh[kp], m2p[kp & 3]);
    }
    float m1 = (m1p[0] + m1p[1]) + (m1p[2] + m1p[3]), m2 = (m2p[0] + m2p[1]) + (m2p[2] + m2p[3]);
    if (NG == 1) { m1 += __shfl_xor(m1, 32); m2 += __shfl_xor(m2, 32); }
    m1 *= inv_n;
    m2 *= inv_n;
#pragma unroll
    for (int kp = 0; kp < 32; ++kp) a[kp] = rstd * (dxh[kp] - m1 - xh[kp] * m2);
  }
}

#ifdef HA_LAYER_TIMING
// profiling build only (tools/layer_timing.py): phase timestamps of (block 0, wave 0) of the most recent launches
__device__ unsigned long long g_layer_ts[64][10];
__device__ unsigned int g_layer_launches;
#define HA_TS(i, drain)                                                              \
  do {                                                                               \
    if (ts_on) {                                                                     \
      if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       \
      g_layer_ts[ts_slot][i] = clock64();                                            \
    }                                                                                \
  } while (0)
#else
#define HA_TS(i, drain)
#endif

// One block = (task, row tile rt, output tile, K-split ks): its NWT waves walk the block's `spb` consecutive 64-channel
// K-slices (wave w takes slices w, w + NWT, ...), accumulate in registers, reduce across waves through LDS and write one
// partial slab.  spb is the per-network policy (ha_tune_set "layer_spb"): few slices per block spread the fp32 MFMA work
// (256 FLOP/clk/CU) over more CUs but make every consumer block re-read nsplit partial slabs; spb >= nslices is full-K.
// LEAN: every task is a plain GEMM on finished single slabs (behind gn_finish_kernel): no GroupNorm state and no partial-slab
// staging -- 56 VGPRs instead of 338, a quarter of the operand bytes per block.
template <int NWT, bool LEAN>
__global__ __launch_bounds__(NWT * 64) void mlp_layer_kernel(LayerLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // NWT * 1024 floats
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#ifdef HA_LAYER_TIMING
  const bool ts_on = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0;
  unsigned ts_slot = 0;
  if (ts_on) {
    ts_slot = atomicAdd(&g_layer_launches, 1u) & 63;
    g_layer_ts[ts_slot][8] = wall_clock64();
    g_layer_ts[ts_slot][9] = ((unsigned long long)L.t[0].nblocks << 32);
  }
  HA_TS(0, false);
#endif
  // grid = (output tile, K-split, row tile): no integer divisions on the start-up path
  const LayerTask& T = L.t[0];
  const int tile = blockIdx.x, ks = blockIdx.y, rt = blockIdx.z;
  const int hi = lane >> 5;
  const int s_end = (ks + 1) * T.spb < T.nslices ? (ks + 1) * T.spb : T.nslices;

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int slice = ks * T.spb + wave; slice < s_end; slice += NWT) {
    // B operand: this slice's 64 x 32 weight panel
    float bw[32];
    {
      const float* wp = T.Wp + ((size_t)tile * T.nslices + slice) * 32 * 64 + lane * 4;   // [kp / 4][lane][4]
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const vf4 w = *reinterpret_cast<const vf4*>(wp + j * 256);
        bw[4 * j] = w.x; bw[4 * j + 1] = w.y; bw[4 * j + 2] = w.z; bw[4 * j + 3] = w.w;
      }
    }
    const bool is_main = slice < T.main_slices;
    const int cbase = is_main ? slice * SLICE : (slice - T.main_slices) * SLICE;
    // GroupNorm affine of the lane's channels, fetched with the same batch of loads (not behind the activation wait)
    float gam[32], bet[32];
    const bool has_gn = !LEAN && is_main && T.mode != 0;
    if (has_gn) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const vf4 g4 = *reinterpret_cast<const vf4*>(T.gamma + cbase + 32 * hi + 4 * j);
        const vf4 b4 = *reinterpret_cast<const vf4*>(T.beta + cbase + 32 * hi + 4 * j);
        gam[4 * j] = g4.x; gam[4 * j + 1] = g4.y; gam[4 * j + 2] = g4.z; gam[4 * j + 3] = g4.w;
        bet[4 * j] = b4.x; bet[4 * j + 1] = b4.y; bet[4 * j + 2] = b4.z; bet[4 * j + 3] = b4.w;
      }
    }
    float a[32], h[32];
    if (!is_main) load_frag_n<true>(T.skip, 1, L.RT, T.skip_dim, rt, cbase, lane, a);
    else {
      load_frag_n<LEAN>(T.src, T.nsplit_src, L.RT, T.Csrc, rt, cbase, lane, a);
      if (!LEAN && T.mode == 3) load_frag_n(T.hsrc, T.nsplit_h, L.RT, T.Ch, rt, cbase, lane, h);
    }
    HA_TS(1, false);
    HA_TS(2, true);
    if (!LEAN && is_main && T.mode == 1 && T.hsum_dst && tile == 0 && T.nsplit_src > 1) {
      // The producer left split-K partial slabs; this block has just summed its slices of them.  The blocks of output tile 0 (they
      // cover every slice once) write the sums back as one slab: the adjoint pass then reads ONE slab of pre-activations
      // instead of nsplit (its launches are bound by the ~1 KB-per-instruction issue rate of those loads).
      const int c0 = cbase + 32 * hi;
      float* hd = T.hsum_dst + (size_t)rt * T.Csrc * 32 + (size_t)(c0 >> 2) * 128 + (size_t)(lane & 31) * 4;
#pragma unroll
      for (int jq = 0; jq < 8; ++jq)
        if (c0 + 4 * jq < T.Csrc) *reinterpret_cast<vf4*>(hd + (size_t)jq * 128) = vf4{a[4 * jq], a[4 * jq + 1], a[4 * jq + 2], a[4 * jq + 3]};
    }
    if (has_gn) {
      // GroupNorm over groups of T.group (64 or 32) channels: the lane holds one half-slice of its row
      const float inv_n = T.inv_group;
      if (T.mode == 1) {
        if (T.group == SLICE) gn_apply<1>(1, gam, bet, inv_n, a, a);
        else gn_apply<2>(1, gam, bet, inv_n, a, a);
      } else {
        if (T.group == SLICE) gn_apply<1>(3, gam, bet, inv_n, h, a);
        else gn_apply<2>(3, gam, bet, inv_n, h, a);
      }
    }
    HA_TS(3, false);
#pragma unroll
    for (int kp = 0; kp < 32; ++kp) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kp], bw[kp], acc, 0, 0, 0);
  }
  // cross-wave K reduction through LDS, then one 16-byte store per (row, column quad)
#pragma unroll
  for (int i = 0; i < 16; ++i) smem[(wave * 16 + i) * 64 + lane] = acc[i];
  HA_TS(4, true);
  __syncthreads();
  HA_TS(5, false);
  if (threadIdx.x < 256) {
    // thread <-> (row, quad of output columns): accumulator register i of lane l is row (i&3) + 8 (i>>2) + 4 (l>>5),
    // column l&31, so a column quad of one row is 16 contiguous bytes of the staged accumulators
    const int row = threadIdx.x & 31, cq = threadIdx.x >> 5;            // 32 rows x 8 column quads
    const int reg = (row & 3) + 4 * (row >> 3), half = (row >> 2) & 1;
    vf4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NWT; ++w) o += *reinterpret_cast<const vf4*>(smem + (w * 16 + reg) * 64 + half * 32 + cq * 4);
    const int n = tile * 32 + cq * 4;
    if (ks == 0 && T.bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < T.Nout) o[e] += T.bias[n + e];
    }
    if (!LEAN && T.acc) {
      // accumulate policy: the slab was zeroed before the roll-out; returnless global_atomic_add_f32, complete at the kernel boundary
      float* dst = T.dst + (size_t)rt * T.Nout_pad * 32 + (size_t)(n >> 2) * 128 + row * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) atomic_add_f32(dst + e, o[e]);
    } else {
      float* dst = T.dst + ((size_t)ks * L.RT + rt) * T.Nout_pad * 32 + (size_t)(n >> 2) * 128 + row * 4;
      *reinterpret_cast<vf4*>(dst) = o;
    }
  }
  HA_TS(6, true);
  HA_TS(7, false);
}

// Two or more row tiles: the split-K partial slabs are summed and the GroupNorm(+ReLU) prologue -- or its adjoint -- is
// applied ONCE per activation by this kernel into a finished slab, and the layer kernel then runs as a plain GEMM on it
// (mode 0, one slab per operand).  Inside the layer kernel that work is repeated by each of the 32 output-tile blocks (every
// block re-reads up to five partial slabs; 3-4 k cycles of GroupNorm per slice against 2.3 k cycles of MFMA), which held
// the 256-row launches at ~25 us; at 32 rows the extra launch costs more than it saves.
struct FinishLaunch {
  LayerTask t[1];
  float* dst[1];
  int nblk[1];          // blocks of the task = RT * main_slices
  int ntasks, RT;
};

// 256 threads per (task, row tile, 64-channel slice): thread <-> (row = t & 31, channel octet = t >> 5), i.e. two 16-byte loads
// per slab and thread, so the whole fragment is one short batch of loads instead of a 60-100-load chain in one wave; the
// per-row GroupNorm sums cross the waves through LDS (two-pass statistics, as in the layer kernel's prologue).
__global__ __launch_bounds__(256) void gn_finish_kernel(FinishLaunch F) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [pass 4][row 32][octet 8]
  float (*s_red)[32][8] = reinterpret_cast<float (*)[32][8]>(smem);
  const int b = blockIdx.x, ti = 0;
  const LayerTask& T = F.t[0];
  const int row = threadIdx.x & 31, oct = threadIdx.x >> 5;
  const int slice = b % T.main_slices, rt = b / T.main_slices;
  const int c0 = slice * SLICE + oct * 8;            // this thread's 8 channels
  const vf4 g0 = *reinterpret_cast<const vf4*>(T.gamma + c0), g1 = *reinterpret_cast<const vf4*>(T.gamma + c0 + 4);
  const vf4 e0 = *reinterpret_cast<const vf4*>(T.beta + c0), e1 = *reinterpret_cast<const vf4*>(T.beta + c0 + 4);
  // sums `ns` partial slabs of the 8 channels (all loads first, summed in split order)
  auto load8 = [&](const float* base, int ns, int C, float (&v)[8]) {
    const size_t stride = (size_t)F.RT * C * 32;
    const float* p = base + (size_t)rt * C * 32 + (size_t)(c0 >> 2) * 128 + (size_t)row * 4;
    vf4 t0[MAXSPLIT], t1[MAXSPLIT];
#pragma unroll
    for (int k = 0; k < MAXSPLIT; ++k) {
      t0[k] = vf4{0.f, 0.f, 0.f, 0.f};
      t1[k] = t0[k];
      if (k < ns) { t0[k] = *reinterpret_cast<const vf4*>(p + k * stride); t1[k] = *reinterpret_cast<const vf4*>(p + k * stride + 128); }
    }
    vf4 a0 = t0[0], a1 = t1[0];
#pragma unroll
    for (int k = 1; k < MAXSPLIT; ++k) { a0 += t0[k]; a1 += t1[k]; }
    v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
  };
  float x[8], da[8];
  if (T.mode == 1) load8(T.src, T.nsplit_src, T.Csrc, x);
  else { load8(T.src, T.nsplit_src, T.Csrc, da); load8(T.hsrc, T.nsplit_h, T.Ch, x); }
  const float gam[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
  const float bet[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
  // a group is 64 channels (all 8 octets of the row) or 32 (octets 0-3 / 4-7)
  const int per = T.group == SLICE ? 8 : 4, o0 = T.group == SLICE ? 0 : (oct & 4);
  const float inv_n = 1.0f / (float)T.group;
  auto row_sum = [&](int pass, float v) {
    s_red[pass][row][oct] = v;
    __syncthreads();
    float t = 0.f;
    for (int k = 0; k < per; ++k) t += s_red[pass][row][o0 + k];
    return t;
  };
  float p1 = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) p1 += x[i];
  const float mu = row_sum(0, p1) * inv_n;
  float p2 = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { const float d = x[i] - mu; p2 = fmaf(d, d, p2); }
  const float rstd = rsqrtf(row_sum(1, p2) * inv_n + 1e-5f);
  float out[8];
  if (T.mode == 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = fmaxf((x[i] - mu) * rstd * gam[i] + bet[i], 0.f);
  } else {
    float xh[8], dxh[8], m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      xh[i] = (x[i] - mu) * rstd;
      const float y = xh[i] * gam[i] + bet[i];
      dxh[i] = (y > 0.f ? da[i] : 0.f) * gam[i];
      m1 += dxh[i];
      m2 = fmaf(dxh[i], xh[i], m2);
    }
    m1 = row_sum(2, m1) * inv_n;
    m2 = row_sum(3, m2) * inv_n;
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = rstd * (dxh[i] - m1 - xh[i] * m2);
  }
  // finished slab: same quad layout, width = the channels this task consumes (main_slices * 64)
  float* d = F.dst[ti] + (size_t)rt * (T.main_slices * SLICE) * 32 + (size_t)(c0 >> 2) * 128 + (size_t)row * 4;
  *reinterpret_cast<vf4*>(d) = vf4{out[0], out[1], out[2], out[3]};
  *reinterpret_cast<vf4*>(d + 128) = vf4{out[4], out[5], out[6], out[7]};
}


// ---------------------------------------------------------------------------------------------------
// Batched layer GEMM of the PRIOR network.  The prior is off the recurrence in both directions: its input x_t is known once the
// decoder chain has produced the states, and its output adjoints (g_mu, g_var) are given up front.  So instead of riding in
// the per-step launches at M = B rows, each of its layers runs ONCE over all S * RT row tiles (M = 32 S RT rows; 1888 at
// 32 x 59, 30464 at 256 x 119) as an MFMA-shaped GEMM, forward (humor_model.py:407-418 evaluated for all steps at once) and
// adjoint.  A wave owns RM x 2 output tiles of 32 x 32 (RM row tiles x one 64-column pair = one 64-channel or two 32-channel
// GroupNorm groups) and walks the whole K: no split-K partial slabs, no finishing pass -- the GroupNorm(+ReLU) of the
// produced activation (forward) or its adjoint (backward) is applied in the epilogue, where each row's group is complete
// inside the wave (lane-local sums plus one lane^32 shuffle, as in the layer kernel's prologue).  The four waves of a block
// take four adjacent column pairs of the same rows (the A operand is shared through the CU's L1); operands of the next
// K-slice are fetched into a second register set while the MFMAs of the current one issue.  Blocks are dealt to the XCDs in
// column-block-major order so that one XCD's L2 holds one 256-column weight panel (<= 1 MB) plus its rows.
struct GemmTask {
  const float* Wp; const float* bias;     // packed weights [ntiles][nslices][8][64][4]; bias or null
  int ntiles, nslices, Nout;              // 32-column output tiles, 64-channel K slices, valid output columns
  const float* src; int Csrc;             // A operand: finished slab [row tiles][Csrc][32]
  int nrt;                                // row tiles
  int epi;                                // 0 raw -> dst_h | 1 raw -> dst_h, ReLU(GroupNorm(.)) -> dst_a | 3 GroupNorm-ReLU adjoint (with hsrc) -> dst_a
                                          // 4 LeakyReLU(raw) -> dst_a | 5 LeakyReLU adjoint (hsrc = the forward's activation) -> dst_a
  float slope;                            // epi 4 / 5: negative slope
  const float* gamma; const float* beta; int group;
  const float* hsrc; int Ch;              // epi 3: forward pre-activations of the same channels [row tiles][Ch][32]
  float* dst_h; float* dst_a; int Cdst;   // output slabs [row tiles][Cdst][32]
  int nrg, nwork, per_xcd;                // row groups of RM tiles, work items = column blocks x row groups, items per XCD
};

constexpr int GEMM_LDS_WAVE = 2 * 8 * 132;    // floats of epilogue staging per wave: [column tile][quad][32 rows x 4 + pad]

// KS = 2: waves (2 cw, 2 cw + 1) of a block share tile pair cw of the block's two; wave kpart walks half of the K slices, the partial
// tiles meet in LDS and the even wave runs the epilogue.
template <int RM, int KS = 1>
__global__ __launch_bounds__(256) void prior_gemm_kernel(GemmTask T) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int w = (int)(blockIdx.x & 7) * T.per_xcd + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= T.per_xcd || w >= T.nwork) return;      // (block-uniform)
  const int cb = w / T.nrg, rg = w % T.nrg;
  const int kpart = KS == 1 ? 0 : wave % KS;
  const int ct0 = KS == 1 ? (cb * 4 + wave) * 2 : (cb * (4 / KS) + wave / KS) * 2;
  const bool tile_ok = ct0 < T.ntiles;
  if (KS == 1 && !tile_ok) return;                           // (with a K split the idle waves stay for the block barriers)
  const int ct1 = ct0 + 1 < T.ntiles ? ct0 + 1 : ct0;      // an odd last tile: the second accumulator is computed and dropped
  int rt[RM];
#pragma unroll
  for (int m = 0; m < RM; ++m) rt[m] = rg * RM + m < T.nrt ? rg * RM + m : T.nrt - 1;

  f32x16 acc[RM][2];
#pragma unroll
  for (int m = 0; m < RM; ++m)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][c][i] = 0.f;

  const float* wp0 = T.Wp + (size_t)ct0 * T.nslices * 2048 + lane * 4;
  const float* wp1 = T.Wp + (size_t)ct1 * T.nslices * 2048 + lane * 4;
  auto load_b = [&](int slice, float (&b0)[32], float (&b1)[32]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const vf4 u = *reinterpret_cast<const vf4*>(wp0 + (size_t)slice * 2048 + j * 256);
      const vf4 v = *reinterpret_cast<const vf4*>(wp1 + (size_t)slice * 2048 + j * 256);
      b0[4 * j] = u.x; b0[4 * j + 1] = u.y; b0[4 * j + 2] = u.z; b0[4 * j + 3] = u.w;
      b1[4 * j] = v.x; b1[4 * j + 1] = v.y; b1[4 * j + 2] = v.z; b1[4 * j + 3] = v.w;
    }
  };
  // full 64-channel slices: unconditional 16-byte loads (nothing depends on the loaded values before the MFMAs read them)
  auto load_a = [&](int slice, float (&a)[RM][32]) {
#pragma unroll
    for (int m = 0; m < RM; ++m)
      load_frag<1>(T.src + (size_t)rt[m] * T.Csrc * 32 + (size_t)(slice * 16 + 8 * (lane >> 5)) * 128 + (size_t)(lane & 31) * 4, 0, 8, a[m]);
  };
  auto mma = [&](const float (&a)[RM][32], const float (&b0)[32], const float (&b1)[32]) {
#pragma unroll
    for (int kp = 0; kp < 32; ++kp)
#pragma unroll
      for (int m = 0; m < RM; ++m) {
        acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][kp], b0[kp], acc[m][0], 0, 0, 0);
        acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][kp], b1[kp], acc[m][1], 0, 0, 0);
      }
  };
  const int nfull_all = T.Csrc / SLICE < T.nslices ? T.Csrc / SLICE : T.nslices;      // slices whose 64 channels all exist in the slab
  // this wave's share of the full slices: [sbeg, nfull)
  const int per_part = (nfull_all + KS - 1) / KS;
  const int sbeg = KS == 1 ? 0 : (kpart * per_part < nfull_all ? kpart * per_part : nfull_all);
  const int nfull = KS == 1 ? nfull_all : ((sbeg + per_part) < nfull_all ? (sbeg + per_part) : nfull_all);
  if (nfull > sbeg && (KS == 1 || tile_ok)) {
    float aA[RM][32], bA0[32], bA1[32], aB[RM][32], bB0[32], bB1[32];
    load_b(sbeg, bA0, bA1);
    load_a(sbeg, aA);
    // HA_SCHED_FENCE keeps the compiler from sinking the next slice's loads down to their first use (which serialises every
    // memory round trip behind the MFMAs instead of hiding it under them)
    for (int slice = sbeg; slice < nfull; slice += 2) {
      const int s1 = slice + 1 < nfull ? slice + 1 : slice;
      load_b(s1, bB0, bB1);
      load_a(s1, aB);
      HA_SCHED_FENCE();
      mma(aA, bA0, bA1);
      HA_SCHED_FENCE();
      if (slice + 1 < nfull) {
        const int s2 = slice + 2 < nfull ? slice + 2 : slice + 1;
        load_b(s2, bA0, bA1);
        load_a(s2, aA);
        HA_SCHED_FENCE();
        mma(aB, bB0, bB1);
        HA_SCHED_FENCE();
      }
    }
  }
  if (nfull_all < T.nslices && kpart == KS - 1 && (KS == 1 || tile_ok)) {
    // ragged K tail (339 = 5 x 64 + 19 state channels, 96 = 64 + 32 prior outputs): quads beyond the slab width read as zero
    float a[RM][32], b0[32], b1[32];
    load_b(nfull_all, b0, b1);
#pragma unroll
    for (int m = 0; m < RM; ++m) load_frag_n<true>(T.src, 1, 0, T.Csrc, rt[m], nfull_all * SLICE, lane, a[m]);
    mma(a, b0, b1);
  }

  // ---- epilogue: accumulators -> (row, 32 channels of one column tile) per lane through the wave's LDS slice -----------------
  float* sl = smem + wave * GEMM_LDS_WAVE;
  const int row = lane & 31, hh = lane >> 5;
  const int ct = hh ? ct0 + 1 : ct0;                  // this lane's column tile
  const bool ct_ok = ct < T.ntiles;
  float gam[32], bet[32];
  if ((T.epi == 1 || T.epi == 3) && ct_ok) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const vf4 g4 = *reinterpret_cast<const vf4*>(T.gamma + ct * 32 + 4 * j);
      const vf4 b4 = *reinterpret_cast<const vf4*>(T.beta + ct * 32 + 4 * j);
      gam[4 * j] = g4.x; gam[4 * j + 1] = g4.y; gam[4 * j + 2] = g4.z; gam[4 * j + 3] = g4.w;
      bet[4 * j] = b4.x; bet[4 * j + 1] = b4.y; bet[4 * j + 2] = b4.z; bet[4 * j + 3] = b4.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 32; ++k) { gam[k] = 1.f; bet[k] = 0.f; }
  }
#pragma unroll
  for (int m = 0; m < RM; ++m) {
    // accumulator register i of lane l: row (i&3) + 8 (i>>2) + 4 (l>>5), column l&31
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = (i & 3) + 8 * (i >> 2) + 4 * hh, col = lane & 31;
        sl[(c * 8 + (col >> 2)) * 132 + r * 4 + (col & 3)] = acc[m][c][i];
      }
    if (KS == 1) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
      __syncthreads();                                  // the K partners' partial tiles are staged
    }
    float v[32];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      vf4 q = *reinterpret_cast<const vf4*>(sl + (hh * 8 + j) * 132 + row * 4);
      if (KS > 1) {
#pragma unroll
        for (int pk = 1; pk < KS; ++pk) q += *reinterpret_cast<const vf4*>(sl + pk * GEMM_LDS_WAVE + (hh * 8 + j) * 132 + row * 4);
      }
      v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
    }
    if (KS == 1) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();                  // the slice is rewritten by the next row tile
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
      __syncthreads();
      if (kpart != 0) continue;                         // only the even wave of the pair finishes the tile (no block barrier below)
    }
    const bool live = ct_ok && rg * RM + m < T.nrt;
    if (T.bias && ct_ok) {       // the packed bias is zero-padded to whole 32-column tiles
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const vf4 b4 = *reinterpret_cast<const vf4*>(T.bias + ct * 32 + 4 * j);
        v[4 * j] += b4.x; v[4 * j + 1] += b4.y; v[4 * j + 2] += b4.z; v[4 * j + 3] += b4.w;
      }
    }
    const size_t o = (size_t)rt[m] * T.Cdst * 32 + (size_t)(ct * 8) * 128 + (size_t)row * 4;
    auto store32 = [&](float* base, const float (&x)[32]) {
#pragma unroll
      for (int j = 0; j < 8; ++j) *reinterpret_cast<vf4*>(base + o + (size_t)j * 128) = vf4{x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3]};
    };
    if (T.epi <= 1 && live) store32(T.dst_h, v);
    if (T.epi == 4) {
#pragma unroll
      for (int k = 0; k < 32; ++k) v[k] = v[k] > 0.f ? v[k] : v[k] * T.slope;
      if (live) store32(T.dst_a, v);
    } else if (T.epi == 5) {       // sign(LeakyReLU(h)) = sign(h): the stored activation stands in for the pre-activation
      const float* hp = T.hsrc + (size_t)rt[m] * T.Ch * 32 + (size_t)((ct_ok ? ct : ct0) * 8) * 128 + (size_t)row * 4;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const vf4 q = *reinterpret_cast<const vf4*>(hp + (size_t)j * 128);
        v[4 * j] *= q.x > 0.f ? 1.f : T.slope; v[4 * j + 1] *= q.y > 0.f ? 1.f : T.slope;
        v[4 * j + 2] *= q.z > 0.f ? 1.f : T.slope; v[4 * j + 3] *= q.w > 0.f ? 1.f : T.slope;
      }
      if (live) store32(T.dst_a, v);
    } else if (T.epi == 1) {
      const float inv_n = 1.0f / (float)T.group;
      if (T.group == SLICE) gn_apply<1>(1, gam, bet, inv_n, v, v);       // (all lanes: the cross-half shuffle is wave-wide)
      else gn_apply<2>(1, gam, bet, inv_n, v, v);
      if (live) store32(T.dst_a, v);
    } else if (T.epi == 3) {
      float h[32];
      const float* hp = T.hsrc + (size_t)rt[m] * T.Ch * 32 + (size_t)((ct_ok ? ct : ct0) * 8) * 128 + (size_t)row * 4;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const vf4 q = *reinterpret_cast<const vf4*>(hp + (size_t)j * 128);
        h[4 * j] = q.x; h[4 * j + 1] = q.y; h[4 * j + 2] = q.z; h[4 * j + 3] = q.w;
      }
      const float inv_n = 1.0f / (float)T.group;
      if (T.group == SLICE) gn_apply<1>(3, gam, bet, inv_n, h, v);
      else gn_apply<2>(3, gam, bet, inv_n, h, v);
      if (live) store32(T.dst_a, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// glue kernels: one wave per sequence (row); lane j < 22 owns joint j, lanes 1..21 also body rotation j-1,
// lane 0 the root quantities.
// ---------------------------------------------------------------------------------------------------
struct GlueParams {
  int B, S, t, RT;
  // forward stash
  const float* xT;          // [RT][D_IN][32]   input state of step t (transposed)
  float* xT_next;           // [RT][D_IN][32]
  const float* dec_out; int dec_nsplit; int dec_pad;   // decoder raw output slabs [nsplit][RT][dec_pad][32]
  const float* pri_out; int pri_nsplit; int pri_pad;   // prior output slabs
  const float* Gs;          // [RT*32][12] accumulated world transform at step t: G (9) | gt (3)
  float* Gs_next;
  const float* t2j;         // [RT*32][3]
  float* world;             // [B][S][348]
  float* prior_mu; float* prior_var;   // [B][S][48] or null
  // backward
  const float* g_world;     // [B][S][348] or null
  const float* g_prior_mu; const float* g_prior_var;
  float* g_dec_out;         // [RT][dec_pad][32] adjoint of the decoder raw output
  float* g_pri_out;         // [RT][pri_pad][32]
  const float* gx_dir_in;   // [RT][D_IN][32] direct part of dL/dx_{t+1} (from glue_bwd(t+1)); null at t = S-1
  float* gx_dir_out;        // [RT][D_IN][32] direct part of dL/dx_t
  const float* gxp_pri; int gxp_pri_nsplit; int gxp_pri_pad;   // step t+1 layer-0 input-gradient slabs (prior)
  const float* gxp_dec; int gxp_dec_nsplit; int gxp_dec_pad;   // (decoder; columns >= D_IN are dz)
  float* carry;             // [RT*32][16]: gG (9) | ggt (3) | g_t2j (3) | pad
  // dz collection of step t+1: decoder layer input-gradient slabs, z columns at `zoff`
  const float* dz_src[MAXL]; int dz_nsplit[MAXL]; int dz_pad[MAXL]; int dz_off[MAXL]; int dz_n;
  float* g_z;               // [B][S][48]
  float* g_past0;           // [B][D_IN] (final collect only)
};

__device__ __forceinline__ float tsum(const float* base, int nsplit, int RT, int C, int rt, int c, int row) {
  return base ? slab_sum(base, nsplit, RT, C, rt, c, row) : 0.f;
}

struct PredState {
  // lane-local pieces of the decoder prediction (local frame) and what is needed to differentiate it
  float pj[3], jv[3];       // joint position / velocity (lane j < 22)
  float dB[9], Bin[9], pB[9];   // body rotation j-1 (lanes 1..21): delta, input, product
  float ptrans[3], ptvel[3], prvel[3], dR[9], Rin[9], pR[9];   // lane 0
  float raw_aa_b[9], raw_aa_r[9];   // raw rotation outputs (ROTW of them used)
};

// residual rotation from the decoder's raw output (humor_model.py:476-484 convert_to_rotmat(out_val, rep=out_rot_rep)) and its adjoint
template <int ROTW>
__device__ __forceinline__ void delta_rot(const float* raw, float R[9]) {
  if constexpr (ROTW == 3) rodrigues(raw, R);
  else if constexpr (ROTW == 6) rot6d_to_rotmat(raw, R);
  else rot9d_to_rotmat(raw, R);
}
template <int ROTW>
__device__ __forceinline__ void delta_rot_bwd(const float* raw, const float gR[9], float* graw) {
  if constexpr (ROTW == 3) rodrigues_bwd(raw, gR, graw);
  else if constexpr (ROTW == 6) rot6d_to_rotmat_bwd(raw, gR, graw);
  else rot9d_to_rotmat_bwd(raw, gR, graw);
}

// LDS staging of one row's vectors: every lane issues its (independent) loads back-to-back, one barrier, then the
// per-joint math reads LDS.  (The kernels are pure latency: a runtime slab loop per element would serialise ~100 L2/HBM
// round trips per lane.)
constexpr int S_X = 0, S_RAW = 352, S_GXN = 704, S_GW = 1056, S_TOTAL = 1408;      // (the raw area holds up to RawLayout<9>::D = 348 channels)
constexpr int S_SH = 1408, S_RED = 1424, S_TOTAL_BWD = 1456;   // + W(9) ptr(3) of the root | the joint wave's sums (27)

template <int NQ, int NT = 64>   // NQ = ceil(channel quads / NT): thread l stages quads l, l + NT, ... (16-byte loads)
__device__ __forceinline__ void stage_slabs(float* dst, const float* base, int nsplit, int RT, int C, int nch, int rt, int rr,
                                            int lane, bool accumulate) {
  vf4 t[MAXSPLIT][NQ];
#pragma unroll
  for (int sidx = 0; sidx < MAXSPLIT; ++sidx)
#pragma unroll
    for (int i = 0; i < NQ; ++i) t[sidx][i] = vf4{0.f, 0.f, 0.f, 0.f};
  if (base) {
    const size_t stride = (size_t)RT * C * 32;
    const float* p0 = base + (size_t)rt * C * 32 + (size_t)rr * 4;
    // only the slabs that exist are read (nsplit is launch-uniform); every load is issued before the first use
#pragma unroll
    for (int sidx = 0; sidx < MAXSPLIT; ++sidx)
      if (sidx < nsplit) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          const int q = lane + NT * i;
          if (4 * q < nch) t[sidx][i] = *reinterpret_cast<const vf4*>(p0 + (size_t)sidx * stride + (size_t)q * 128);
        }
      }
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    vf4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sidx = 0; sidx < MAXSPLIT; ++sidx) v += t[sidx][i];
    const int q = lane + NT * i;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = 4 * q + e;
      if (c < nch) dst[c] = accumulate ? dst[c] + v[e] : v[e];
    }
  }
}

// DELTA: the decoder emits residuals (vectors add to the input state, rotations left-multiply it: output_delta=True, humor_model.py:460-494);
// !DELTA: it emits the state itself (humor_model.py:331-347 converts only the rotations)
template <int ROTW, bool DELTA>
__device__ __forceinline__ void predict_joints(const float* sX, const float* sRAW, int j, PredState& s) {
  using RL = RawLayout<ROTW>;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    s.pj[c] = sRAW[RL::JNT + 3 * j + c] + (DELTA ? sX[207 + 3 * j + c] : 0.f);
    s.jv[c] = sRAW[RL::JVEL + 3 * j + c] + (DELTA ? sX[273 + 3 * j + c] : 0.f);
  }
}
template <int ROTW, bool DELTA>
__device__ __forceinline__ void predict_body(const float* sX, const float* sRAW, int bidx, PredState& s) {
  using RL = RawLayout<ROTW>;
#pragma unroll
  for (int c = 0; c < ROTW; ++c) s.raw_aa_b[c] = sRAW[RL::BODY + ROTW * bidx + c];
  delta_rot<ROTW>(s.raw_aa_b, s.dB);
  if constexpr (DELTA) {
#pragma unroll
    for (int i = 0; i < 9; ++i) s.Bin[i] = sX[18 + 9 * bidx + i];
    mat3_mul(s.dB, s.Bin, s.pB);
  } else {
#pragma unroll
    for (int i = 0; i < 9; ++i) { s.Bin[i] = 0.f; s.pB[i] = s.dB[i]; }
  }
}
template <int ROTW, bool DELTA>
__device__ __forceinline__ void predict_root(const float* sX, const float* sRAW, PredState& s) {
  using RL = RawLayout<ROTW>;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    s.ptrans[c] = sRAW[c] + (DELTA ? sX[c] : 0.f);
    s.ptvel[c] = sRAW[3 + c] + (DELTA ? sX[3 + c] : 0.f);
    s.prvel[c] = sRAW[RL::RVEL + c] + (DELTA ? sX[15 + c] : 0.f);
  }
#pragma unroll
  for (int c = 0; c < ROTW; ++c) s.raw_aa_r[c] = sRAW[RL::ROOT + c];
  delta_rot<ROTW>(s.raw_aa_r, s.dR);
  if constexpr (DELTA) {
#pragma unroll
    for (int i = 0; i < 9; ++i) s.Rin[i] = sX[6 + i];
    mat3_mul(s.dR, s.Rin, s.pR);
  } else {
#pragma unroll
    for (int i = 0; i < 9; ++i) { s.Rin[i] = 0.f; s.pR[i] = s.dR[i]; }
  }
}

// Forward glue of step t.  Three waves per sequence (the same split as glue_bwd_kernel): wave 0 the root (lane 0), wave 1
// the joints, wave 2 the body rotations, contact logits and the prior outputs.
template <int ROTW, bool DELTA>
__global__ __launch_bounds__(192) void glue_fwd_kernel(GlueParams p) {
  using RL = RawLayout<ROTW>;
  const int r = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int rt = r >> 5, rr = r & 31;
  const bool valid = r < p.B;
  float* XN = p.xT_next + (size_t)rt * D_INP * 32 + (size_t)rr * 4;
  if (!valid) {
    for (int c = tid; c < D_INP; c += 192) XN[qoff(c)] = 0.f;
    return;
  }
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sX = smem + S_X;
  float* sRAW = smem + S_RAW;
  float* sSH = smem + S_SH;
  stage_slabs<1, 192>(sX, p.xT, 1, p.RT, D_INP, D_IN, rt, rr, tid, false);
  stage_slabs<1, 192>(sRAW, p.dec_out, p.dec_nsplit, p.RT, p.dec_pad, RL::D, rt, rr, tid, false);
  if (tid == 0) XN[qoff(D_IN)] = 0.f;      // pad channel of the next state slab
  float G[9], gt[3], t2j[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) G[i] = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) { gt[c] = 0.f; t2j[c] = 0.f; }
  if (wave < 2) {
#pragma unroll
    for (int i = 0; i < 9; ++i) G[i] = p.Gs[(size_t)r * 12 + i];
#pragma unroll
    for (int c = 0; c < 3; ++c) { gt[c] = p.Gs[(size_t)r * 12 + 9 + c]; t2j[c] = p.t2j[(size_t)r * 3 + c]; }
  }
  __syncthreads();
  float* WO = p.world + ((size_t)r * p.S + p.t) * D_STATE;
  PredState s;
  if (wave == 0) {
    // heading alignment from the predicted root orientation
    if (lane == 0) {
      predict_root<ROTW, DELTA>(sX, sRAW, s);
      W2A wa;
      w2a_fwd(s.pR, wa);
#pragma unroll
      for (int i = 0; i < 9; ++i) sSH[i] = wa.W[i];
#pragma unroll
      for (int c = 0; c < 3; ++c) sSH[9 + c] = s.ptrans[c];
    }
  } else if (wave == 1) {
    if (lane < NJT) predict_joints<ROTW, DELTA>(sX, sRAW, lane, s);
  } else {
    if (lane >= 1 && lane < NJT) {
      predict_body<ROTW, DELTA>(sX, sRAW, lane - 1, s);
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        XN[qoff(18 + 9 * (lane - 1) + i)] = s.pB[i];
        WO[18 + 9 * (lane - 1) + i] = s.pB[i];
      }
    }
    if (lane >= 32 && lane < 32 + 9) {
      const int c = lane - 32;
      WO[339 + c] = sRAW[RL::CONT + c];
    }
    if (p.prior_mu && lane < ZD) {
      const float mu = slab_sum(p.pri_out, p.pri_nsplit, p.RT, p.pri_pad, rt, lane, rr);
      const float lv = slab_sum(p.pri_out, p.pri_nsplit, p.RT, p.pri_pad, rt, ZD + lane, rr);
      p.prior_mu[((size_t)r * p.S + p.t) * ZD + lane] = mu;
      p.prior_var[((size_t)r * p.S + p.t) * ZD + lane] = expf(lv);
    }
  }
  __syncthreads();
  float W[9], ptr[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) W[i] = sSH[i];
#pragma unroll
  for (int c = 0; c < 3; ++c) ptr[c] = sSH[9 + c];
  const float wt[3] = {-ptr[0], -ptr[1], 0.f};

  if (wave == 1 && lane < NJT) {
    const int j = lane;
    float q[3], o[3];
    // next input: W (pj + wt + t2j) - t2j ; W jv
#pragma unroll
    for (int c = 0; c < 3; ++c) q[c] = s.pj[c] + wt[c] + t2j[c];
    mat3_vec(W, q, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) XN[qoff(207 + 3 * j + c)] = o[c] - t2j[c];
    mat3_vec(W, s.jv, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) XN[qoff(273 + 3 * j + c)] = o[c];
    // world: G^T (pj + t2j) - t2j - gt ; G^T jv
#pragma unroll
    for (int c = 0; c < 3; ++c) q[c] = s.pj[c] + t2j[c];
    mat3_tvec(G, q, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) WO[207 + 3 * j + c] = o[c] - t2j[c] - gt[c];
    mat3_tvec(G, s.jv, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) WO[273 + 3 * j + c] = o[c];
  }
  if (wave == 0 && lane == 0) {
    float q[3], o[3], M[9];
#pragma unroll
    for (int c = 0; c < 3; ++c) q[c] = s.ptrans[c] + wt[c];
    mat3_vec(W, q, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) XN[qoff(c)] = o[c];
    mat3_vec(W, s.ptvel, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) XN[qoff(3 + c)] = o[c];
    mat3_mul(W, s.pR, M);
#pragma unroll
    for (int i = 0; i < 9; ++i) XN[qoff(6 + i)] = M[i];
    mat3_vec(W, s.prvel, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) XN[qoff(15 + c)] = o[c];
    // world
    float wtr[3];
    mat3_tvec(G, s.ptrans, wtr);
#pragma unroll
    for (int c = 0; c < 3; ++c) { wtr[c] -= gt[c]; WO[c] = wtr[c]; }
    mat3_tvec(G, s.ptvel, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) WO[3 + c] = o[c];
    mat3_tmul(G, s.pR, M);
#pragma unroll
    for (int i = 0; i < 9; ++i) WO[6 + i] = M[i];
    mat3_tvec(G, s.prvel, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) WO[15 + c] = o[c];
    // accumulate the world transform
    mat3_mul(G, W, M);
#pragma unroll
    for (int i = 0; i < 9; ++i) p.Gs_next[(size_t)r * 12 + i] = M[i];
    p.Gs_next[(size_t)r * 12 + 9] = -wtr[0];
    p.Gs_next[(size_t)r * 12 + 10] = -wtr[1];
    p.Gs_next[(size_t)r * 12 + 11] = 0.f;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// a (x) b accumulated into M (3x3): M[i][k] += a_i b_k
__device__ __forceinline__ void outer_acc(float M[9], const float a[3], const float b[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) M[i * 3 + k] = fmaf(a[i], b[k], M[i * 3 + k]);
}

// Backward of step t.  Also collects dL/dz of step t+1 (its decoder backward has completed) and, when
// p.t == -1 ("final collect"), only assembles dL/dpast_in0 and dL/dz_0.
// One block of four waves per sequence: the kernel is a single dependent chain of ~6000 instructions when one wave does
// everything, so the independent pieces run side by side -- wave 0 the root (lane 0), wave 1 the 22 joints, wave 2 the 21
// body rotations, wave 3 the latent / prior / contact adjoints -- and meet at two barriers (W from the root's forward
// recomputation; the joint wave's reduced shared adjoints back to the root).

template <int ROTW, bool DELTA>
__global__ __launch_bounds__(256) void glue_bwd_kernel(GlueParams p) {
  using RL = RawLayout<ROTW>;
  const int r = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int rt = r >> 5, rr = r & 31;
  if (r >= p.B) return;
  const bool last = p.t == p.S - 1;      // no step t+1 behind this one
  const bool final_collect = p.t < 0;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sX = smem + S_X;
  float* sRAW = smem + S_RAW;
  float* sGXN = smem + S_GXN;
  float* sGW = smem + S_GW;
  float* sSH = smem + S_SH;
  float* sRED = smem + S_RED;
  // ---- total adjoint of x_{t+1}: direct part + layer-0 input-gradient slabs of step t+1 (all 256 threads stage) -------
  if (last) {
    for (int c = tid; c < D_IN; c += 256) sGXN[c] = 0.f;
  } else {
    stage_slabs<1, 256>(sGXN, p.gx_dir_in, 1, p.RT, D_INP, D_IN, rt, rr, tid, false);
    stage_slabs<1, 256>(sGXN, p.gxp_pri, p.gxp_pri_nsplit, p.RT, p.gxp_pri_pad, D_IN, rt, rr, tid, true);
    stage_slabs<1, 256>(sGXN, p.gxp_dec, p.gxp_dec_nsplit, p.RT, p.gxp_dec_pad, D_IN, rt, rr, tid, true);
  }
  if (!final_collect) {
    stage_slabs<1, 256>(sX, p.xT, 1, p.RT, D_INP, D_IN, rt, rr, tid, false);
    stage_slabs<1, 256>(sRAW, p.dec_out, p.dec_nsplit, p.RT, p.dec_pad, RL::D, rt, rr, tid, false);
    const float* GWp = p.g_world ? p.g_world + ((size_t)r * p.S + p.t) * D_STATE : nullptr;
    for (int c = tid; c < D_STATE; c += 256) sGW[c] = GWp ? GWp[c] : 0.f;
  }
  // per-sequence state: issued before the barrier so that it overlaps the staging round trip
  float* carry = p.carry + (size_t)r * 16;
  float G[9], gt[3], t2j[3], gGn[9], ggtn[3], g_t2j_acc[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) { G[i] = 0.f; gGn[i] = 0.f; }
#pragma unroll
  for (int c = 0; c < 3; ++c) { gt[c] = 0.f; t2j[c] = 0.f; ggtn[c] = 0.f; g_t2j_acc[c] = 0.f; }
  if (!final_collect && wave < 2) {
#pragma unroll
    for (int i = 0; i < 9; ++i) G[i] = p.Gs[(size_t)r * 12 + i];
#pragma unroll
    for (int c = 0; c < 3; ++c) { gt[c] = p.Gs[(size_t)r * 12 + 9 + c]; t2j[c] = p.t2j[(size_t)r * 3 + c]; }
    if (!last && wave == 0) {
      // incoming carried adjoints of (G', gt') = state after this step
#pragma unroll
      for (int i = 0; i < 9; ++i) gGn[i] = carry[i];
#pragma unroll
      for (int c = 0; c < 3; ++c) { ggtn[c] = carry[9 + c]; g_t2j_acc[c] = carry[12 + c]; }
    }
  }
  __syncthreads();
  auto GXN = [&](int c) -> float { return sGXN[c]; };
  auto gw = [&](int c) { return sGW[c]; };
  if (final_collect) {
    // dz of step 0 ; dL/dpast_in0 = adjoint of x_0 ; t2j = -(x0[207], x0[208], 0)
    if (p.g_z && tid < ZD) {
      float v = 0.f;
      for (int i = 0; i < p.dz_n; ++i) v += slab_sum(p.dz_src[i], p.dz_nsplit[i], p.RT, p.dz_pad[i], rt, p.dz_off[i] + tid, rr);
      p.g_z[((size_t)r * p.S + (p.t + 1)) * ZD + tid] = v;
    }
    for (int c = tid; c < D_IN; c += 256) {
      float v = GXN(c);
      if (c == 207) v -= carry[12];
      if (c == 208) v -= carry[13];
      p.g_past0[(size_t)r * D_IN + c] = v;
    }
    return;
  }

  float* GD = p.g_dec_out + (size_t)rt * p.dec_pad * 32 + (size_t)rr * 4;   // adjoint of the decoder raw output (quad layout)
  float* GX = p.gx_dir_out + (size_t)rt * D_INP * 32 + (size_t)rr * 4;
  PredState s;
  W2A wa;
  // lane-local partial sums of the adjoints shared by the whole sequence (W, G, gt, wt, t2j)
  float gW[9], gG[9], ggt[3] = {0.f, 0.f, 0.f}, gwt[3] = {0.f, 0.f, 0.f}, gt2[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 9; ++i) { gW[i] = 0.f; gG[i] = 0.f; }

  // ---- phase 1 (before W is known) ---------------------------------------------------------------------
  if (wave == 0) {
    if (lane == 0) {
      predict_root<ROTW, DELTA>(sX, sRAW, s);
      w2a_fwd(s.pR, wa);
#pragma unroll
      for (int i = 0; i < 9; ++i) sSH[i] = wa.W[i];
#pragma unroll
      for (int c = 0; c < 3; ++c) sSH[9 + c] = s.ptrans[c];
    }
  } else if (wave == 1) {
    if (lane < NJT) predict_joints<ROTW, DELTA>(sX, sRAW, lane, s);
  } else if (wave == 2) {
    // body rotation: pB = dB * Bin goes unchanged to both outputs (no dependence on W or G)
    if (lane >= 1 && lane < NJT) {
      const int bidx = lane - 1;
      predict_body<ROTW, DELTA>(sX, sRAW, bidx, s);
      float gpB[9], gdB[9], gBin[9], gaa[ROTW];
#pragma unroll
      for (int i = 0; i < 9; ++i) gpB[i] = gw(18 + 9 * bidx + i) + GXN(18 + 9 * bidx + i);
      if constexpr (DELTA) {
        mat3_mult(gpB, s.Bin, gdB);      // gdB = gpB * Bin^T
        mat3_tmul(s.dB, gpB, gBin);      // gBin = dB^T * gpB
      } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) { gdB[i] = gpB[i]; gBin[i] = 0.f; }
      }
      delta_rot_bwd<ROTW>(s.raw_aa_b, gdB, gaa);
#pragma unroll
      for (int c = 0; c < ROTW; ++c) GD[qoff(RL::BODY + ROTW * bidx + c)] = gaa[c];
#pragma unroll
      for (int i = 0; i < 9; ++i) GX[qoff(18 + 9 * bidx + i)] = gBin[i];
    }
  } else {
    // dz of step t+1, contacts, padded decoder channels, prior output adjoint
    if (!last && p.g_z && lane < ZD) {
      float v = 0.f;
      for (int i = 0; i < p.dz_n; ++i) v += slab_sum(p.dz_src[i], p.dz_nsplit[i], p.RT, p.dz_pad[i], rt, p.dz_off[i] + lane, rr);
      p.g_z[((size_t)r * p.S + (p.t + 1)) * ZD + lane] = v;
    }
    if (lane >= 32 && lane < 32 + 9) GD[qoff(RL::CONT + lane - 32)] = gw(339 + lane - 32);
    for (int c = RL::D + lane; c < p.dec_pad; c += 64) GD[qoff(c)] = 0.f;
    if (p.g_pri_out) {
      float* GP = p.g_pri_out + (size_t)rt * p.pri_pad * 32 + (size_t)rr * 4;
      if (lane < ZD) {
        const size_t o = ((size_t)r * p.S + p.t) * ZD + lane;
        GP[qoff(lane)] = p.g_prior_mu ? p.g_prior_mu[o] : 0.f;
        // var = exp(logvar): d/dlogvar = g_var * var (recomputed from the stashed prior output slabs)
        const float var = expf(slab_sum(p.pri_out, p.pri_nsplit, p.RT, p.pri_pad, rt, ZD + lane, rr));
        GP[qoff(ZD + lane)] = p.g_prior_var ? p.g_prior_var[o] * var : 0.f;
      }
      for (int c = 2 * ZD + lane; c < p.pri_pad; c += 64) GP[qoff(c)] = 0.f;
    }
  }
  __syncthreads();
  float W[9], ptr[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) W[i] = sSH[i];
#pragma unroll
  for (int c = 0; c < 3; ++c) ptr[c] = sSH[9 + c];
  const float wt[3] = {-ptr[0], -ptr[1], 0.f};

  // ---- phase 2: joints (wave 1) next to the root (wave 0, lane 0) ----------------------------------------
  float gptrans[3] = {0.f, 0.f, 0.f}, gptvel[3] = {0.f, 0.f, 0.f}, gprvel[3] = {0.f, 0.f, 0.f}, gpR[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) gpR[i] = 0.f;
  if (wave == 1) {
    if (lane < NJT) {
      const int j = lane;
    float g[3], q[3], o[3], gpj[3] = {0.f, 0.f, 0.f}, gjv[3] = {0.f, 0.f, 0.f};
    // world joints: wj = G^T (pj + t2j) - t2j - gt
#pragma unroll
    for (int c = 0; c < 3; ++c) { g[c] = gw(207 + 3 * j + c); q[c] = s.pj[c] + t2j[c]; }
    mat3_vec(G, g, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) { gpj[c] += o[c]; gt2[c] += o[c] - g[c]; ggt[c] -= g[c]; }
    outer_acc(gG, q, g);
    // world joint velocities: G^T jv
#pragma unroll
    for (int c = 0; c < 3; ++c) g[c] = gw(273 + 3 * j + c);
    mat3_vec(G, g, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) gjv[c] += o[c];
    outer_acc(gG, s.jv, g);
    // next-input joints: W (pj + wt + t2j) - t2j
#pragma unroll
    for (int c = 0; c < 3; ++c) { g[c] = GXN(207 + 3 * j + c); q[c] = s.pj[c] + wt[c] + t2j[c]; }
    mat3_tvec(W, g, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) { gpj[c] += o[c]; gwt[c] += o[c]; gt2[c] += o[c] - g[c]; }
    outer_acc(gW, g, q);
    // next-input joint velocities: W jv
#pragma unroll
    for (int c = 0; c < 3; ++c) g[c] = GXN(273 + 3 * j + c);
    mat3_tvec(W, g, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) gjv[c] += o[c];
    outer_acc(gW, g, s.jv);
    // residual composition: pj = raw + x
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      GD[qoff(RL::JNT + 3 * j + c)] = gpj[c];
      GX[qoff(207 + 3 * j + c)] = DELTA ? gpj[c] : 0.f;
      GD[qoff(RL::JVEL + 3 * j + c)] = gjv[c];
      GX[qoff(273 + 3 * j + c)] = DELTA ? gjv[c] : 0.f;
    }
    }
    // reduce the joint lanes' partial sums and hand them to the root
#pragma unroll
    for (int i = 0; i < 9; ++i) { gW[i] = wave_sum(gW[i]); gG[i] = wave_sum(gG[i]); }
#pragma unroll
    for (int c = 0; c < 3; ++c) { ggt[c] = wave_sum(ggt[c]); gwt[c] = wave_sum(gwt[c]); gt2[c] = wave_sum(gt2[c]); }
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 9; ++i) { sRED[i] = gW[i]; sRED[9 + i] = gG[i]; }
#pragma unroll
      for (int c = 0; c < 3; ++c) { sRED[18 + c] = ggt[c]; sRED[21 + c] = gwt[c]; sRED[24 + c] = gt2[c]; }
    }
  } else if (wave == 0 && lane == 0) {
    float g[3], o[3], q[3];
    // carried: gt' = (-wtrans.x, -wtrans.y, 0)
    float gwtr[3] = {gw(0) - ggtn[0], gw(1) - ggtn[1], gw(2)};
    // wtrans = G^T ptrans - gt
    mat3_vec(G, gwtr, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) { gptrans[c] += o[c]; ggt[c] -= gwtr[c]; }
    outer_acc(gG, s.ptrans, gwtr);
    // wtvel = G^T ptvel
#pragma unroll
    for (int c = 0; c < 3; ++c) g[c] = gw(3 + c);
    mat3_vec(G, g, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) gptvel[c] += o[c];
    outer_acc(gG, s.ptvel, g);
    // wR = G^T pR : gpR += G gwR ; gG += pR gwR^T
    float gwR[9], M[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) gwR[i] = gw(6 + i);
    mat3_mul(G, gwR, M);
#pragma unroll
    for (int i = 0; i < 9; ++i) gpR[i] += M[i];
    mat3_mult(s.pR, gwR, M);
#pragma unroll
    for (int i = 0; i < 9; ++i) gG[i] += M[i];
    // wrvel = G^T prvel
#pragma unroll
    for (int c = 0; c < 3; ++c) g[c] = gw(15 + c);
    mat3_vec(G, g, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) gprvel[c] += o[c];
    outer_acc(gG, s.prvel, g);
    // G' = G W : gG += gG' W^T ; gW += G^T gG'
    mat3_mult(gGn, W, M);
#pragma unroll
    for (int i = 0; i < 9; ++i) gG[i] += M[i];
    mat3_tmul(G, gGn, M);
#pragma unroll
    for (int i = 0; i < 9; ++i) gW[i] += M[i];
    // next input: trans' = W (ptrans + wt)
#pragma unroll
    for (int c = 0; c < 3; ++c) { g[c] = GXN(c); q[c] = s.ptrans[c] + wt[c]; }
    mat3_tvec(W, g, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) { gptrans[c] += o[c]; gwt[c] += o[c]; }
    outer_acc(gW, g, q);
    // tvel' = W ptvel
#pragma unroll
    for (int c = 0; c < 3; ++c) g[c] = GXN(3 + c);
    mat3_tvec(W, g, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) gptvel[c] += o[c];
    outer_acc(gW, g, s.ptvel);
    // R' = W pR : gpR += W^T gR' ; gW += gR' pR^T
    float gRn[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) gRn[i] = GXN(6 + i);
    mat3_tmul(W, gRn, M);
#pragma unroll
    for (int i = 0; i < 9; ++i) gpR[i] += M[i];
    mat3_mult(gRn, s.pR, M);
#pragma unroll
    for (int i = 0; i < 9; ++i) gW[i] += M[i];
    // rvel' = W prvel
#pragma unroll
    for (int c = 0; c < 3; ++c) g[c] = GXN(15 + c);
    mat3_tvec(W, g, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) gprvel[c] += o[c];
    outer_acc(gW, g, s.prvel);
  }
  __syncthreads();

  // ---- phase 3: the root finishes (heading alignment and root rotation adjoints, carried state) -------------
  if (wave == 0 && lane == 0) {
#pragma unroll
    for (int i = 0; i < 9; ++i) { gW[i] += sRED[i]; gG[i] += sRED[9 + i]; }
#pragma unroll
    for (int c = 0; c < 3; ++c) { ggt[c] += sRED[18 + c]; gwt[c] += sRED[21 + c]; gt2[c] += sRED[24 + c]; }
    // wt = (-ptrans.x, -ptrans.y, 0)
    gptrans[0] -= gwt[0];
    gptrans[1] -= gwt[1];
    // W = world2aligned(pR)
    float g0, g3;
    w2a_bwd(wa, gW, g0, g3);
    gpR[0] += g0;
    gpR[3] += g3;
    // pR = dR * Rin
    float gdR[9], gRin[9], gaa[ROTW];
    if constexpr (DELTA) {
      mat3_mult(gpR, s.Rin, gdR);
      mat3_tmul(s.dR, gpR, gRin);
    } else {
#pragma unroll
      for (int i = 0; i < 9; ++i) { gdR[i] = gpR[i]; gRin[i] = 0.f; }
    }
    delta_rot_bwd<ROTW>(s.raw_aa_r, gdR, gaa);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      GD[qoff(c)] = gptrans[c];        GX[qoff(c)] = DELTA ? gptrans[c] : 0.f;
      GD[qoff(3 + c)] = gptvel[c];   GX[qoff(3 + c)] = DELTA ? gptvel[c] : 0.f;
      GD[qoff(RL::RVEL + c)] = gprvel[c];   GX[qoff(15 + c)] = DELTA ? gprvel[c] : 0.f;
    }
#pragma unroll
    for (int c = 0; c < ROTW; ++c) GD[qoff(RL::ROOT + c)] = gaa[c];
#pragma unroll
    for (int i = 0; i < 9; ++i) GX[qoff(6 + i)] = gRin[i];
    // carry to step t-1
#pragma unroll
    for (int i = 0; i < 9; ++i) carry[i] = gG[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) { carry[9 + c] = ggt[c]; carry[12 + c] = g_t2j_acc[c] + gt2[c]; }
  }
}

static inline void launch_glue_fwd(int rotw, bool delta, int rows, hipStream_t st, const GlueParams& g) {
  const size_t lds = S_TOTAL_BWD * sizeof(float);
#define HA_GLUE_FWD(R, D) HA_LAUNCH((glue_fwd_kernel<R, D>), dim3(rows), dim3(192), lds, st, g)
  if (delta) { if (rotw == 3) HA_GLUE_FWD(3, true); else if (rotw == 6) HA_GLUE_FWD(6, true); else HA_GLUE_FWD(9, true); }
  else { if (rotw == 3) HA_GLUE_FWD(3, false); else if (rotw == 6) HA_GLUE_FWD(6, false); else HA_GLUE_FWD(9, false); }
#undef HA_GLUE_FWD
}
static inline void launch_glue_bwd(int rotw, bool delta, int rows, hipStream_t st, const GlueParams& g) {
  const size_t lds = S_TOTAL_BWD * sizeof(float);
#define HA_GLUE_BWD(R, D) HA_LAUNCH((glue_bwd_kernel<R, D>), dim3(rows), dim3(256), lds, st, g)
  if (delta) { if (rotw == 3) HA_GLUE_BWD(3, true); else if (rotw == 6) HA_GLUE_BWD(6, true); else HA_GLUE_BWD(9, true); }
  else { if (rotw == 3) HA_GLUE_BWD(3, false); else if (rotw == 6) HA_GLUE_BWD(6, false); else HA_GLUE_BWD(9, false); }
#undef HA_GLUE_BWD
}

// z_t = mu + eps * sqrt(var) (or mu when eps is null) from the prior output slabs of step t; writes the transposed
// latent of step t (the decoder's skip operand) and the row-major z / prior outputs (humor_model.py:1029-1047).
struct SampleParams {
  int B, S, t, RT;
  const float* pri_out; int pri_nsplit; int pri_pad;
  const float* eps;         // [B][S][48] or null
  float* zT_t;              // [RT][48][32]
  float* z_out;             // [B][S][48]
};

__global__ __launch_bounds__(64) void sample_z_kernel(SampleParams p) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const int rt = r >> 5, rr = r & 31;
  if (lane >= ZD) return;
  float* dst = p.zT_t + (size_t)rt * ZD * 32 + (size_t)rr * 4 + qoff(lane);
  if (r >= p.B) { *dst = 0.f; return; }
  const float mu = slab_sum(p.pri_out, p.pri_nsplit, p.RT, p.pri_pad, rt, lane, rr);
  const float lv = slab_sum(p.pri_out, p.pri_nsplit, p.RT, p.pri_pad, rt, ZD + lane, rr);
  const size_t o = ((size_t)r * p.S + p.t) * ZD + lane;
  const float z = p.eps ? mu + p.eps[o] * sqrtf(expf(lv)) : mu;
  *dst = z;
  p.z_out[o] = z;
}

// Prior outputs of every step, off the recurrence: mu / var = exp(logvar) from the stashed prior output slabs (forward), and
// the adjoint of those slabs (g_mu, g_var * var, zero padding) for the backward pass.  One block per (sequence, step).
struct PriorIOParams {
  int B, S, RT;
  const float* pri_out0; size_t step_stride; int pri_nsplit; int pri_pad;   // slabs of step t at pri_out0 + t * step_stride
  float* prior_mu; float* prior_var;                 // forward: [B][S][48]
  const float* g_prior_mu; const float* g_prior_var; // backward inputs (either may be null)
  float* g_pri_all;                                  // backward: [S][RT][pri_pad][32]
};

__global__ __launch_bounds__(64) void prior_io_kernel(PriorIOParams p) {
  const int r = blockIdx.x, t = blockIdx.y, lane = threadIdx.x;
  const int rt = r >> 5, rr = r & 31;
  const float* po = p.pri_out0 + (size_t)t * p.step_stride;
  float mu = 0.f, var = 0.f;
  if (r < p.B && lane < ZD) {
    mu = slab_sum(po, p.pri_nsplit, p.RT, p.pri_pad, rt, lane, rr);
    var = expf(slab_sum(po, p.pri_nsplit, p.RT, p.pri_pad, rt, ZD + lane, rr));
  }
  const size_t o = ((size_t)r * p.S + t) * ZD + lane;
  if (p.prior_mu && r < p.B && lane < ZD) { p.prior_mu[o] = mu; p.prior_var[o] = var; }
  if (p.g_pri_all) {
    float* GP = p.g_pri_all + ((size_t)t * p.RT + rt) * p.pri_pad * 32 + (size_t)rr * 4;
    if (lane < ZD) {
      const bool live = r < p.B;
      GP[qoff(lane)] = (live && p.g_prior_mu) ? p.g_prior_mu[o] : 0.f;
      GP[qoff(ZD + lane)] = (live && p.g_prior_var) ? p.g_prior_var[o] * var : 0.f;
    }
    for (int c = 2 * ZD + lane; c < p.pri_pad; c += 64) GP[qoff(c)] = 0.f;
  }
}

// [B][S][C] -> [S][RT][Cp/4][32][4] (quad-interleaved tiles, zero-padded rows and channels); Cp = C rounded up to 4
__global__ void transpose_in_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int S, int C, int Cp, int RT) {
  const size_t total = (size_t)S * RT * Cp * 32;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(i & 3), rr = (int)((i >> 2) & 31);
    const size_t q = i >> 7;
    const int c = (int)(q % (Cp >> 2)) * 4 + e;
    const size_t q2 = q / (Cp >> 2);
    const int rt = (int)(q2 % RT);
    const int s = (int)(q2 / RT);
    const int r = rt * 32 + rr;
    dst[i] = (r < B && c < C) ? src[((size_t)r * S + s) * C + c] : 0.f;
  }
}

__global__ void init_state_kernel(const float* __restrict__ past0, float* __restrict__ Gs, float* __restrict__ t2j, int B, int rows) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float I[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
  for (int i = 0; i < 12; ++i) Gs[(size_t)r * 12 + i] = I[i];
  t2j[(size_t)r * 3 + 0] = r < B ? -past0[(size_t)r * D_IN + 207] : 0.f;
  t2j[(size_t)r * 3 + 1] = r < B ? -past0[(size_t)r * D_IN + 208] : 0.f;
  t2j[(size_t)r * 3 + 2] = 0.f;
}

// ---------------------------------------------------------------------------------------------------
// rotation conversion kernels
// ---------------------------------------------------------------------------------------------------
__global__ void rodrigues_fwd_kernel(int n, const float* __restrict__ aa, float* __restrict__ R) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float r[3] = {aa[(size_t)i * 3], aa[(size_t)i * 3 + 1], aa[(size_t)i * 3 + 2]};
  float M[9];
  rodrigues(r, M);
#pragma unroll
  for (int k = 0; k < 9; ++k) R[(size_t)i * 9 + k] = M[k];
}
__global__ void rodrigues_bwd_kernel(int n, const float* __restrict__ aa, const float* __restrict__ gR, float* __restrict__ g_aa) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float r[3] = {aa[(size_t)i * 3], aa[(size_t)i * 3 + 1], aa[(size_t)i * 3 + 2]};
  float g[9], o[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) g[k] = gR[(size_t)i * 9 + k];
  rodrigues_bwd(r, g, o);
#pragma unroll
  for (int k = 0; k < 3; ++k) g_aa[(size_t)i * 3 + k] = o[k];
}
__global__ void rotmat_to_aa_fwd_kernel(int n, const float* __restrict__ R, float* __restrict__ aa) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float M[9], o[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) M[k] = R[(size_t)i * 9 + k];
  rotmat_to_aa(M, o);
#pragma unroll
  for (int k = 0; k < 3; ++k) aa[(size_t)i * 3 + k] = o[k];
}
__global__ void rotmat_to_aa_bwd_kernel(int n, const float* __restrict__ R, const float* __restrict__ g_aa, float* __restrict__ gR) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float M[9], g[3], o[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) M[k] = R[(size_t)i * 9 + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) g[k] = g_aa[(size_t)i * 3 + k];
  rotmat_to_aa_bwd(M, g, o);
#pragma unroll
  for (int k = 0; k < 9; ++k) gR[(size_t)i * 9 + k] = o[k];
}

__global__ void rot6d_fwd_kernel(int n, const float* __restrict__ x, float* __restrict__ R) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v[6], M[9];
#pragma unroll
  for (int k = 0; k < 6; ++k) v[k] = x[(size_t)i * 6 + k];
  rot6d_to_rotmat(v, M);
#pragma unroll
  for (int k = 0; k < 9; ++k) R[(size_t)i * 9 + k] = M[k];
}
__global__ void rot6d_bwd_kernel(int n, const float* __restrict__ x, const float* __restrict__ gR, float* __restrict__ gx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v[6], g[9], o[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) v[k] = x[(size_t)i * 6 + k];
#pragma unroll
  for (int k = 0; k < 9; ++k) g[k] = gR[(size_t)i * 9 + k];
  rot6d_to_rotmat_bwd(v, g, o);
#pragma unroll
  for (int k = 0; k < 6; ++k) gx[(size_t)i * 6 + k] = o[k];
}
__global__ void rot9d_fwd_kernel(int n, const float* __restrict__ x, float* __restrict__ R) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v[9], M[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) v[k] = x[(size_t)i * 9 + k];
  rot9d_to_rotmat(v, M);
#pragma unroll
  for (int k = 0; k < 9; ++k) R[(size_t)i * 9 + k] = M[k];
}
__global__ void rot9d_bwd_kernel(int n, const float* __restrict__ x, const float* __restrict__ gR, float* __restrict__ gx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v[9], g[9], o[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) { v[k] = x[(size_t)i * 9 + k]; g[k] = gR[(size_t)i * 9 + k]; }
  rot9d_to_rotmat_bwd(v, g, o);
#pragma unroll
  for (int k = 0; k < 9; ++k) gx[(size_t)i * 9 + k] = o[k];
}

// ---------------------------------------------------------------------------------------------------
// host side: weight packing, stash layout, orchestration
// ---------------------------------------------------------------------------------------------------
template <typename T>
static int upload_vec(T** dst, const std::vector<T>& src) {
  *dst = nullptr;
  if (src.empty()) return HA_OK;
  HA_CHECK_HIP(hipMalloc((void**)dst, src.size() * sizeof(T)));
  HA_CHECK_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return HA_OK;
}

static int pack_layer(PackedLayer& L, const float* W, const float* bias, const float* gamma, const float* beta, int Cin,
                      int skip, int Nout, bool has_gn) {
  L.Cin = Cin; L.skip = skip; L.Nout = Nout;
  L.main_slices = ceil_div(Cin, SLICE);
  L.nslices_f = L.main_slices + ceil_div(skip, SLICE);
  L.ntiles_f = ceil_div(Nout, 32);
  L.Nout_pad = L.ntiles_f * 32;

  L.nslices_b = ceil_div(Nout, SLICE);
  L.ntiles_b = ceil_div(Cin + skip, 32);
  L.Nin_pad = L.ntiles_b * 32;

  L.group = has_gn ? Cin / 16 : 0;
  HA_REQUIRE(ceil_div(L.nslices_f, 4) <= MAXSPLIT && ceil_div(L.nslices_b, 4) <= MAXSPLIT, "layer %dx%d needs more than %d K-splits", Nout,
             Cin + skip, MAXSPLIT);
  const int Kin = Cin + skip;
  // forward pack [tile][slice][kp/4][lane][kp%4]: lane l of (tile, slice, kp) <-> W[n = tile*32 + (l&31)][k]
  std::vector<float> wf((size_t)L.ntiles_f * L.nslices_f * 32 * 64, 0.f);
  for (int tile = 0; tile < L.ntiles_f; ++tile)
    for (int sl = 0; sl < L.nslices_f; ++sl)
      for (int kp = 0; kp < 32; ++kp)
        for (int l = 0; l < 64; ++l) {
          const int n = tile * 32 + (l & 31);
          const int cc = 32 * (l >> 5) + kp;        // lane (n, hi) owns channels 32 hi + kp of its slice
          int k;
          if (sl < L.main_slices) { const int c = sl * SLICE + cc; k = c < Cin ? c : -1; }
          else { const int c = (sl - L.main_slices) * SLICE + cc; k = c < skip ? Cin + c : -1; }
          if (n < Nout && k >= 0) wf[(((size_t)tile * L.nslices_f + sl) * 8 + (kp >> 2)) * 256 + l * 4 + (kp & 3)] = W[(size_t)n * Kin + k];
        }
  // backward pack: A = dh [rows, Nout channels], output columns = input channels (main then skip)
  std::vector<float> wb((size_t)L.ntiles_b * L.nslices_b * 32 * 64, 0.f);
  for (int tile = 0; tile < L.ntiles_b; ++tile)
    for (int sl = 0; sl < L.nslices_b; ++sl)
      for (int kp = 0; kp < 32; ++kp)
        for (int l = 0; l < 64; ++l) {
          const int n = tile * 32 + (l & 31);                 // input channel
          const int k = sl * SLICE + 32 * (l >> 5) + kp;      // output channel
          if (n < Kin && k < Nout) wb[(((size_t)tile * L.nslices_b + sl) * 8 + (kp >> 2)) * 256 + l * 4 + (kp & 3)] = W[(size_t)k * Kin + n];
        }
  int rc;
  if ((rc = upload_vec(&L.Wf, wf)) != HA_OK) return rc;
  if ((rc = upload_vec(&L.Wb, wb)) != HA_OK) return rc;
  std::vector<float> bv(L.Nout_pad, 0.f);      // zero-padded to whole output tiles (vector loads in the GEMM epilogue)
  for (int i = 0; i < Nout; ++i) bv[i] = bias[i];
  if ((rc = upload_vec(&L.bias, bv)) != HA_OK) return rc;
  if (has_gn) {
    std::vector<float> gv(gamma, gamma + Cin), bev(beta, beta + Cin);
    if ((rc = upload_vec(&L.gamma, gv)) != HA_OK) return rc;
    if ((rc = upload_vec(&L.beta, bev)) != HA_OK) return rc;
  }
  return HA_OK;
}

static int pack_mlp(PackedLayer* out, const ha_mlp_desc* d, const char* name) {
  HA_REQUIRE(d->n_linear >= 1 && d->n_linear <= MAXL, "%s: n_linear=%d out of range", name, d->n_linear);
  int cin = d->in_dim - d->skip_dim;
  for (int i = 0; i < d->n_linear; ++i) {
    HA_REQUIRE(d->w[i] && d->b[i], "%s: layer %d weights missing", name, i);
    const bool gn = i > 0;
    if (gn) {
      HA_REQUIRE(d->gn_gamma[i] && d->gn_beta[i], "%s: layer %d GroupNorm affine missing", name, i);
      HA_REQUIRE(cin % 16 == 0 && (cin / 16 == 64 || cin / 16 == 32), "%s: hidden width %d needs GroupNorm groups of 32 or 64 channels", name, cin);
    }
    int rc = pack_layer(out[i], d->w[i], d->b[i], d->gn_gamma[i], d->gn_beta[i], cin, d->skip_dim, d->out_dims[i], gn);
    if (rc != HA_OK) return rc;
    cin = d->out_dims[i];
  }
  return HA_OK;
}

// Offsets (in floats) into the caller-owned stash.
struct StashLayout {
  int RT = 0;
  // launch policy of the layer kernel for this batch size: K-slices per block and the resulting partial-slab counts
  int spb = 4, nw = 4;
  int nsf_pri[MAXL], nsf_dec[MAXL], nsb_dec[MAXL];
  size_t zT = 0, t2j = 0, per_step = 0, steps = 0;
  size_t xT = 0;                  // input states of all steps, contiguous: [(S+1)][RT][D_INP][32] (the batched prior's A operand)
  size_t off_G = 0;
  size_t off_dec[MAXL];
  size_t off_hsum[MAXL];          // per step: summed pre-activations of decoder layer l (one slab), written by layer l+1's forward launch
  // batched prior network: pre-activations of every layer for all steps, [S*RT][Nout_pad][32] each (kept for the adjoint)
  size_t pri_h[MAXL];
  size_t pri_act[2];              // finished activations / adjoints of consecutive layers (ping-pong), [S*RT][1024][32]
  size_t gx_pri = 0;              // dL/dx_t through the prior for all steps, [S*RT][Nin_pad(0)][32]
  size_t smp_pri[MAXL];           // sampling roll-out only: the per-step prior's partial slabs (one step's worth, reused)
  // backward scratch
  size_t gx_dir[2], carry = 0, g_dec_out = 0, g_pri_out = 0;
  size_t bwd_dec[MAXL];
  size_t fin[1] = {0};         // finished-activation scratch of a launch's task (finishing-pass policy only)
  bool finish = false;
  bool hsum = false;             // forward launches write the summed pre-activations back (one-row-tile policy: no finishing pass)
  // accumulate policy (g_layer_acc, one-row-tile policy): every decoder activation / adjoint is ONE slab that the K-split blocks add
  // into.  The slabs must be zero before the chain starts, so the adjoint's scratch is kept per step too and both regions are
  // cleared by one memset per pass ((S+1) x 0.36 MB forward, S x 0.4 MB backward at one row tile).
  bool acc = false;
  // persistent forward (rollout_persist.hip): every decoder activation is ONE complete slab (bias included), written by the
  // persistent kernel; the adjoint reads it like a one-split partial slab
  bool single = false;
  bool pipe = false;             // single, 32 < B <= 256: the layer-parallel pipelined kernels (rollout_pipe.inc), one set of records per row tile
  size_t persist_ws = 0;         // exchange space of the persistent kernel
  size_t off_gn[3] = {0, 0, 0}, off_gl = 0;   // single: per step GroupNorm statistics [16][32][2] x 3, glue record [32][32]
  size_t off_ht[3] = {0, 0, 0};               // single: per step the hidden pre-activations again, team layout [8][channel][4 rows]
  size_t dz_part = 0;            // single: partial dL/dz products of the persistent adjoint [S][31][32][48]
  size_t bwd_set = 0;            // acc: floats per step of the adjoint scratch (bwd_dec[] are offsets of step 0's set); else 0
  size_t bwd_begin = 0, bwd_floats = 0;
  int rd_f(int i) const { return (acc || single) ? 1 : nsf_dec[i]; }     // partial slabs a consumer of decoder layer i's output reads
  int rd_b(int i) const { return acc ? 1 : nsb_dec[i]; }
  size_t total = 0;
};

// row groups of the call being served on this host thread (for_each_group): side-by-side groups run on their own streams, and two
// persistent launches cannot share the chip (each needs one block on every CU), so the one-launch path serves single-group calls only
static thread_local int tl_groups = 1;
// -1: make_layout decides from the live state (workspace queries); 0 / 1: the mode decided at the entry point of the call being served
static thread_local int tl_single_mode = -1;
static thread_local const float* tl_gz_add = nullptr;   // ha_humor_rollout_backward_ex: addend of dL/dz (rows of the group being run)

static void make_layout(const ha_humor_net* net, int B, int S, StashLayout& L, bool allow_acc = true) {
  L.RT = ceil_div(B, 32);
  const size_t RT = L.RT;
  // K is split over blocks of 4 slices (the fp32 MFMA rate is 256 FLOP/clk/CU: the work has to be spread over the chip).  At
  // one row tile the consumer's prologue sums the partial slabs and applies GroupNorm itself (one launch per level, the
  // launch is a latency chain); from two row tiles on a finishing pass does that once per activation (gn_finish_kernel) and
  // the layer kernel runs as a lean GEMM.  Measured with tools/rollout_ab.py (fwd+bwd ms, in-kernel vs finishing pass):
  // 32 rows 5.7 / 6.4, 64 rows 7.9 / 7.2, 96 rows 10.2 / 7.7, 128 rows 10.6 / 8.3, 256 rows x 119 steps 29.6 / 21.2.
  L.spb = g_layer_spb >= 4 ? g_layer_spb : 4;      // MAXSPLIT partial slabs at most (checked against the layer widths at pack time)
  L.nw = 4;
  for (int i = 0; i < net->n_pri; ++i) L.nsf_pri[i] = ceil_div(net->pri[i].nslices_f, L.spb);
  for (int i = 0; i < net->n_dec; ++i) { L.nsf_dec[i] = ceil_div(net->dec[i].nslices_f, L.spb); L.nsb_dec[i] = ceil_div(net->dec[i].nslices_b, L.spb); }
  // GroupNorm prologues once per activation in gn_finish_kernel (see there and the policy note above)
  L.finish = (L.RT >= 2 && g_layer_finish != 0) || g_layer_finish == 2;     // 0: never, 2: always (A/B runs)
  const bool want_single = tl_single_mode >= 0 ? tl_single_mode >= 1 : (g_rollout_persist != 0 && persist_usable(net->persist));
  const bool may_single = allow_acc && !L.finish && B <= 32 && tl_groups == 1;
  // 32 < B <= 256: the pipelined persistent kernels; the launch-chain adjoint behind them keeps its finishing-pass policy (it reads the
  // forward's pre-activations as ONE complete slab per activation and row tile, like behind the B <= 32 kernel)
  L.pipe = allow_acc && B > 32 && B <= 256 && tl_groups == 1 && g_rollout_pipe != 0 && g_layer_finish != 2 && want_single;
  L.single = (may_single && want_single) || L.pipe;
  L.acc = allow_acc && !L.finish && !L.single && g_layer_acc != 0;
  L.hsum = !L.finish && !L.acc && !L.single && g_layer_hsum != 0;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += (n + 63) / 64 * 64; return r; };
  L.zT = take((size_t)S * RT * ZD * 32);
  L.t2j = take(RT * 32 * 3);
  L.xT = take((size_t)(S + 1) * RT * D_INP * 32);
  // per-step region: accumulated world transform and the decoder's partial slabs
  size_t p = 0;
  auto ptake = [&](size_t n) { size_t r = p; p += (n + 63) / 64 * 64; return r; };
  L.off_G = ptake(RT * 32 * 12);
  for (int i = 0; i < net->n_dec; ++i) L.off_dec[i] = ptake((size_t)L.rd_f(i) * RT * net->dec[i].Nout_pad * 32);
  for (int i = 0; i + 1 < net->n_dec; ++i) L.off_hsum[i] = L.hsum ? ptake(RT * net->dec[i].Nout_pad * 32) : 0;
  if (L.single) {
    for (int i = 0; i < 3; ++i) L.off_gn[i] = ptake(RT * 16 * 32 * 2);
    L.off_gl = ptake(RT * 32 * 32);
    for (int i = 0; i < 3; ++i) L.off_ht[i] = ptake(RT * 32 * (size_t)net->dec[i].Nout);
  }
  L.per_step = p;
  L.steps = take(p * (size_t)(S + 1));
  size_t widest = 0;
  for (int i = 0; i < net->n_pri; ++i) {
    L.pri_h[i] = take((size_t)S * RT * net->pri[i].Nout_pad * 32);
    if ((size_t)net->pri[i].Nout_pad > widest) widest = net->pri[i].Nout_pad;
    if ((size_t)net->pri[i].Nin_pad > widest) widest = net->pri[i].Nin_pad;
    L.smp_pri[i] = take((size_t)L.nsf_pri[i] * RT * net->pri[i].Nout_pad * 32);
  }
  L.pri_act[0] = take((size_t)S * RT * widest * 32);
  L.pri_act[1] = take((size_t)S * RT * widest * 32);
  L.gx_pri = take((size_t)S * RT * net->pri[0].Nin_pad * 32);
  L.gx_dir[0] = take(RT * D_INP * 32);
  L.gx_dir[1] = take(RT * D_INP * 32);
  L.carry = take(RT * 32 * 16);
  L.g_dec_out = take(RT * net->dec[net->n_dec - 1].Nout_pad * 32);
  L.g_pri_out = take((size_t)S * RT * net->pri[net->n_pri - 1].Nout_pad * 32);   // prior-output adjoints of all steps
  L.bwd_begin = o;
  for (int i = 0; i < net->n_dec; ++i) L.bwd_dec[i] = take((size_t)L.rd_b(i) * RT * net->dec[i].Nin_pad * 32);
  if (L.acc) {
    L.bwd_set = o - L.bwd_begin;
    take(L.bwd_set * (size_t)(S - 1));          // sets of steps 1 .. S-1 behind step 0's
    L.bwd_floats = o - L.bwd_begin;
  }
  if (L.finish)
    L.fin[0] = take(RT * 1280 * 32);     // widest operand: K <= 1280 (checked at pack time)
  if (L.single) {
    L.persist_ws = take(L.pipe ? pipe_ws_floats() : persist_ws_floats());
    L.dz_part = take((size_t)S * (L.pipe ? pipe_dz_slots() : persist_dz_slots()) * RT * 32 * ZD);
  }
  L.total = o;
}

static void fwd_task(LayerTask& T, const PackedLayer& L, const float* src, int nsplit_src, const float* skip, float* dst, int spb,
                     int nsplit_dst) {
  memset(&T, 0, sizeof(T));
  T.Wp = L.Wf; T.bias = L.bias;
  T.ntiles = L.ntiles_f; T.nslices = L.nslices_f; T.main_slices = L.main_slices; T.Nout = L.Nout; T.Nout_pad = L.Nout_pad;
  T.src = src; T.nsplit_src = nsplit_src; T.Csrc = (L.Cin + 3) & ~3;   // slab width (D_INP for the state input)
  T.skip = skip; T.skip_dim = L.skip;
  T.mode = L.group ? 1 : 0;
  T.gamma = L.gamma; T.beta = L.beta; T.group = L.group ? L.group : 64;
  T.inv_group = 1.0f / (float)T.group;
  T.dst = dst;
  T.spb = spb; T.nsplit_dst = nsplit_dst;
  T.nblocks = L.ntiles_f * nsplit_dst;          // x RT in launch_layers
}

// accumulate policy: more than one K-split block per output tile -> they add into the (pre-zeroed) single slab
static void set_acc(LayerTask& T, bool acc) { T.acc = acc && T.nsplit_dst > 1 ? 1 : 0; }

// backward through layer L: A = dh (adjoint of L's raw output), output = adjoint of L's input activation slabs.
// `Lnext_gn` describes the GroupNorm that follows L (i.e. the consumer layer's gamma/beta/group) when dh has to be
// derived from the consumer's input-gradient slabs (mode 3); null when dh is given directly (mode 0).
static void bwd_task(LayerTask& T, const PackedLayer& L, const float* dsrc, int nsplit_d, int dC, const PackedLayer* Lnext_gn,
                     const float* hsrc, int nsplit_h, float* dst, int spb, int nsplit_dst) {
  memset(&T, 0, sizeof(T));
  T.Wp = L.Wb; T.bias = nullptr;
  T.ntiles = L.ntiles_b; T.nslices = L.nslices_b; T.main_slices = L.nslices_b; T.Nout = L.Cin + L.skip; T.Nout_pad = L.Nin_pad;
  T.src = dsrc; T.nsplit_src = nsplit_d; T.Csrc = dC;
  T.mode = Lnext_gn ? 3 : 0;
  if (Lnext_gn) { T.gamma = Lnext_gn->gamma; T.beta = Lnext_gn->beta; T.group = Lnext_gn->group; }
  else T.group = 64;
  T.inv_group = 1.0f / (float)T.group;
  T.hsrc = hsrc; T.nsplit_h = nsplit_h; T.Ch = L.Nout_pad;
  T.dst = dst;
  T.spb = spb; T.nsplit_dst = nsplit_dst;
  T.nblocks = L.ntiles_b * nsplit_dst;
}

static int launch_layers(LayerLaunch& LL, const StashLayout& L, float* stash, hipStream_t st) {
  if (L.finish) {
    FinishLaunch F;
    memset(&F, 0, sizeof(F));
    F.RT = LL.RT;
    int fb = 0;
    for (int i = 0; i < LL.ntasks; ++i) {
      LayerTask& T = LL.t[i];
      if (T.mode == 0) continue;
      const int k = F.ntasks++;
      F.t[k] = T;
      F.dst[k] = stash + L.fin[i];
      F.nblk[k] = LL.RT * T.main_slices;
      fb += F.nblk[k];
      // the layer kernel now consumes the finished slab as a raw operand
      T.src = F.dst[k]; T.nsplit_src = 1; T.Csrc = T.main_slices * SLICE; T.mode = 0;
      T.hsrc = nullptr; T.nsplit_h = 0;
    }
    if (fb) {
      HA_LAUNCH(gn_finish_kernel, dim3(fb), dim3(256), 4 * 32 * 8 * sizeof(float), st, F);
      HA_LAUNCH_CHECK();
    }
  }
  LayerTask& T0 = LL.t[0];
  const dim3 grid(T0.ntiles, T0.nsplit_dst, LL.RT);
  T0.nblocks *= LL.RT;
  // (8-wave blocks need a register diet first: 256-VGPR cap at 2 waves/SIMD -> scratch spills, measured 4x slower)
  const bool lean = L.finish && T0.mode == 0 && T0.nsplit_src <= 1;
  if (lean) HA_LAUNCH((mlp_layer_kernel<4, true>), grid, dim3(256), 4 * 1024 * sizeof(float), st, LL);
  else HA_LAUNCH((mlp_layer_kernel<4, false>), grid, dim3(256), 4 * 1024 * sizeof(float), st, LL);
  HA_LAUNCH_CHECK();
  return HA_OK;
}


// one batched prior layer (forward l >= 0 / adjoint) over nrt = S * RT row tiles
static int launch_prior_gemm(GemmTask& T, hipStream_t st) {
  HA_REQUIRE((T.epi != 1 && T.epi != 3) || (T.ntiles % 2 == 0 && (T.group == 32 || T.group == 64)), "prior GEMM: GroupNorm epilogue needs whole 64-column pairs");
  // RM = 2 (64 x 64 per wave) once there are enough row tiles to keep every SIMD busy with the larger tile
  int ncb = ceil_div(T.ntiles, 8);
  const int rm = g_gemm_rm == 1 || g_gemm_rm == 2 ? g_gemm_rm : ((T.nrt / 2) * ncb >= 2 * 256 ? 2 : 1);
  // K split (experiment knob): only for the one-row-tile-per-wave form and GEMMs that leave most SIMDs idle
  const int ks = ((g_gemm_ks == 2 || g_gemm_ks == 3) && rm == 1 && T.nslices >= 4 && (g_gemm_ks == 3 || ncb * T.nrt * 4 <= 512)) ? 2 : 1;
  if (ks == 2) ncb = ceil_div(T.ntiles, 4);
  T.nrg = ceil_div(T.nrt, rm);
  T.nwork = ncb * T.nrg;
  T.per_xcd = ceil_div(T.nwork, 8);
  const dim3 grid(T.per_xcd * 8), block(256);
  const size_t lds = 4 * GEMM_LDS_WAVE * sizeof(float);
  if (rm == 2) HA_LAUNCH(prior_gemm_kernel<2>, grid, block, lds, st, T);
  else if (ks == 2) HA_LAUNCH((prior_gemm_kernel<1, 2>), grid, block, lds, st, T);
  else HA_LAUNCH(prior_gemm_kernel<1>, grid, block, lds, st, T);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

// forward of the whole prior network for all S steps (inputs: the state slabs x_0 .. x_{S-1})
static int prior_forward_batched(const ha_humor_net* net, const StashLayout& L, float* stash, int S, hipStream_t st) {
  const int np = net->n_pri;
  for (int l = 0; l < np; ++l) {
    const PackedLayer& P = net->pri[l];
    GemmTask T;
    memset(&T, 0, sizeof(T));
    T.Wp = P.Wf; T.bias = P.bias;
    T.ntiles = P.ntiles_f; T.nslices = P.nslices_f; T.Nout = P.Nout;
    T.src = l == 0 ? stash + L.xT : stash + L.pri_act[(l - 1) & 1];
    T.Csrc = l == 0 ? D_INP : net->pri[l - 1].Nout_pad;
    T.nrt = S * L.RT;
    T.dst_h = stash + L.pri_h[l]; T.Cdst = P.Nout_pad;
    if (l + 1 < np) {
      const PackedLayer& N = net->pri[l + 1];       // the GroupNorm between l and l+1 is stored with its consumer
      T.epi = 1; T.gamma = N.gamma; T.beta = N.beta; T.group = N.group;
      T.dst_a = stash + L.pri_act[l & 1];
    }
    int rc = launch_prior_gemm(T, st);
    if (rc != HA_OK) return rc;
  }
  return HA_OK;
}

// adjoint of the whole prior network for all steps: g_pri_out [S*RT][pad][32] -> gx_pri [S*RT][Nin_pad(0)][32]
static int prior_backward_batched(const ha_humor_net* net, const StashLayout& L, float* stash, int S, hipStream_t st) {
  const int np = net->n_pri;
  for (int l = np - 1; l >= 0; --l) {
    const PackedLayer& P = net->pri[l];
    GemmTask T;
    memset(&T, 0, sizeof(T));
    T.Wp = P.Wb; T.bias = nullptr;
    T.ntiles = P.ntiles_b; T.nslices = P.nslices_b; T.Nout = P.Cin + P.skip;
    T.src = l == np - 1 ? stash + L.g_pri_out : stash + L.pri_act[(l + 1) & 1];
    T.Csrc = P.Nout_pad;
    T.nrt = S * L.RT;
    T.Cdst = P.Nin_pad;
    if (l > 0) {
      T.epi = 3; T.gamma = P.gamma; T.beta = P.beta; T.group = P.group;
      T.hsrc = stash + L.pri_h[l - 1]; T.Ch = net->pri[l - 1].Nout_pad;
      T.dst_a = stash + L.pri_act[l & 1];
    } else {
      T.epi = 0; T.dst_h = stash + L.gx_pri;
    }
    int rc = launch_prior_gemm(T, st);
    if (rc != HA_OK) return rc;
  }
  return HA_OK;
}

constexpr int MAX_GROUPS = 8;

// rows per group (whole 32-row tiles) and the group count for a batch of B sequences
// pipelined: the call runs on the pipelined persistent kernels -- chunks of up to 256 sequences one after another on the caller's stream
// (a persistent launch owns every CU: nothing runs beside it)
static void group_plan(int B, int& ngroups, int& rows_per_group, bool pipelined = false) {
  if (pipelined) {
    rows_per_group = 256;
    ngroups = ceil_div(B, 256);
    return;
  }
  const int RT = ceil_div(B, 32);
  int n = g_rollout_groups > 0 ? g_rollout_groups : (RT == 2 ? 2 : 1);
  if (n > MAX_GROUPS) n = MAX_GROUPS;
  if (n > RT) n = RT;
  rows_per_group = ceil_div(RT, n) * 32;
  ngroups = ceil_div(B, rows_per_group);
}

static bool pipelined_call(int B, int mode) { return mode >= 1 && B > 32 && g_rollout_pipe != 0 && g_layer_finish != 2; }

// the process-wide knobs the stash layout depends on, packed: a backward call checks that they are the ones its forward ran with
static int layout_knobs() {
  return (g_layer_spb & 0xff) | ((g_layer_finish & 3) << 8) | ((g_layer_acc & 1) << 10) | ((g_layer_hsum & 1) << 11) | ((g_rollout_pipe & 1) << 12) |
         ((g_rollout_groups & 0xff) << 13);
}
// The adjoint of a stash filled by a one-launch forward: mode 2 (no launch-chain slabs) can only be read by the one-launch adjoint -- the
// stash decides, whatever the adjoint knobs say by now; mode 1 (slabs present) follows the live knobs.
static bool adjoint_persistent(const StashLayout& L, int mode) {
  if (!L.single) return false;
  if (mode == 2) return true;
  return g_rollout_persist_bwd != 0 && (!L.pipe || g_rollout_pipe_bwd != 0);
}

static size_t group_stash_floats(const ha_humor_net* net, int rows, int S) {
  // the largest of: persistent mode, launch-chain mode, sampling roll-out (the mode may change between the query and the call)
  const int saved = tl_single_mode;
  size_t best = 0;
  for (int mode = 0; mode < 2; ++mode) {
    tl_single_mode = mode;
    StashLayout L, Ls;
    make_layout(net, rows, S, L);
    make_layout(net, rows, S, Ls, false);
    best = std::max(best, std::max(L.total, Ls.total));
  }
  tl_single_mode = saved;
  return best;
}

}  // namespace ha

using namespace ha;

extern "C" int ha_humor_net_create(ha_humor_net** out, int device, const ha_mlp_desc* decoder, const ha_mlp_desc* prior) {
  HA_REQUIRE(out && decoder && prior, "ha_humor_net_create: null argument");
  const int raw = decoder->out_dims[decoder->n_linear - 1];
  const int rotw = raw == RawLayout<3>::D ? 3 : raw == RawLayout<6>::D ? 6 : raw == RawLayout<9>::D ? 9 : 0;
  HA_REQUIRE(decoder->in_dim == D_IN + ZD && decoder->skip_dim == ZD && rotw != 0,
             "ha_humor_net_create: decoder must map [339+48] -> 216 / 282 / 348 (out_rot_rep aa / 6d / 9d) with a 48-d latent skip "
             "(got in=%d skip=%d out=%d)", decoder->in_dim, decoder->skip_dim, raw);
  HA_REQUIRE(prior->in_dim == D_IN && prior->skip_dim == 0 && prior->out_dims[prior->n_linear - 1] == 2 * ZD,
             "ha_humor_net_create: prior must map 339 -> 96");
  HA_REQUIRE(prior->n_linear <= 2 * decoder->n_linear, "ha_humor_net_create: the prior may be at most twice as deep as the decoder");
  DeviceGuard guard(device);
  HA_REQUIRE(guard.ok, "ha_humor_net_create: cannot select device %d", device);
  ha_humor_net* net = new ha_humor_net();
  net->device = device;
  net->n_dec = decoder->n_linear;
  net->n_pri = prior->n_linear;
  net->rotw = rotw;
  int rc;
  if ((rc = pack_mlp(net->dec, decoder, "decoder")) != HA_OK || (rc = pack_mlp(net->pri, prior, "prior")) != HA_OK ||
      (rc = persist_create(&net->persist, device, decoder)) != HA_OK) {
    ha_humor_net_destroy(net);
    return rc;
  }
  *out = net;
  return HA_OK;
}

extern "C" int ha_humor_net_destroy(ha_humor_net* net) {
  if (!net) return HA_OK;
  DeviceGuard guard(net->device);
  persist_destroy(net->persist);
  net->persist = nullptr;
  for (PackedLayer* arr : {net->dec, net->pri})
    for (int i = 0; i < MAXL; ++i) {
      void* ptrs[] = {arr[i].Wf, arr[i].Wb, arr[i].bias, arr[i].gamma, arr[i].beta};
      for (void* p : ptrs)
        if (p) (void)hipFree(p);
    }
  delete net;
  return HA_OK;
}

extern "C" int ha_humor_persist_status(const ha_humor_net* net, int* available, unsigned int* error_word, int64_t* launches) {
  HA_REQUIRE(net && available && error_word && launches, "ha_humor_persist_status: null argument");
  *error_word = persist_error_word(net->persist);
  *launches = persist_launches(net->persist) + (persist_launches_bwd(net->persist) << 32);
  *available = persist_usable(net->persist) ? 1 : 0;
  return HA_OK;
}

extern "C" int ha_humor_persist_ack(const ha_humor_net* net) {
  HA_REQUIRE(net, "ha_humor_persist_ack: null argument");
  persist_ack_failure(net->persist);
  return HA_OK;
}

extern "C" int ha_humor_net_set_option(ha_humor_net* net, const char* key, int value) {
  HA_REQUIRE(net && key, "ha_humor_net_set_option: null argument");
  if (strcmp(key, "output_delta") == 0) {
    net->delta = value != 0;
    if (!net->delta && net->persist) {       // the persistent kernels compose residuals: absolute-output networks take the launch chain
      DeviceGuard guard(net->device);
      persist_destroy(net->persist);
      net->persist = nullptr;
    }
    return HA_OK;
  }
  set_error("ha_humor_net_set_option: unknown option '%s'", key);
  return HA_ERR_INVALID_ARG;
}

#ifdef HA_PERSIST_DEBUG
// debugging build only (tools/build_variant.sh pdebug -DHA_PERSIST_DEBUG): stash offsets (floats) of the regions the persistent
// kernels exchange, so that a script can read them back: [xT, steps, per_step, off_G, off_dec0..3, off_gn0..2, off_gl, dz_part, single]
extern "C" int ha_debug_persist_layout(const ha_humor_net* net, int B, int S, int64_t* out) {
  StashLayout L;
  make_layout(net, B, S, L);
  int64_t v[] = {(int64_t)L.xT, (int64_t)L.steps, (int64_t)L.per_step, (int64_t)L.off_G, (int64_t)L.off_dec[0], (int64_t)L.off_dec[1], (int64_t)L.off_dec[2],
                 (int64_t)L.off_dec[3], (int64_t)L.off_gn[0], (int64_t)L.off_gn[1], (int64_t)L.off_gn[2], (int64_t)L.off_gl, (int64_t)L.dz_part, L.single ? 1 : 0,
                 (int64_t)L.persist_ws};
  for (int i = 0; i < 15; ++i) out[i] = v[i];
  return HA_OK;
}
#endif

#ifdef HA_LAYER_TIMING
extern "C" int ha_debug_layer_timing(unsigned long long* out /* [64][10] */, unsigned int* launches) {
  HA_CHECK_HIP(hipDeviceSynchronize());
  HA_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(ha::g_layer_ts), sizeof(unsigned long long) * 640));
  HA_CHECK_HIP(hipMemcpyFromSymbol(launches, HIP_SYMBOL(ha::g_layer_launches), sizeof(unsigned int)));
  return HA_OK;
}
#endif

extern "C" int ha_humor_rollout_workspace(const ha_humor_net* net, int B, int S, int64_t* stash_floats) {
  HA_REQUIRE(net && stash_floats, "ha_humor_rollout_workspace: null argument");
  HA_REQUIRE(B >= 1 && S >= 1, "ha_humor_rollout_workspace: B and S must be >= 1");
  // whole batch (sampling roll-out: one chain; its slabs never accumulate) or the row groups of forward / backward side by side
  int ng, rpg;
  group_plan(B, ng, rpg);
  const size_t whole = group_stash_floats(net, B, S);
  tl_groups = ng;
  size_t grouped = ng > 1 ? (size_t)ng * group_stash_floats(net, rpg, S) : 0;
  tl_groups = 1;
  if (B > 256) {       // chunks of the pipelined kernels, one after another
    group_plan(B, ng, rpg, true);
    grouped = std::max(grouped, (size_t)ng * group_stash_floats(net, rpg, S));
  }
  *stash_floats = (int64_t)(whole > grouped ? whole : grouped);
  return HA_OK;
}

// One phase of the forward pass of one row group: PH_BEGIN (input transposes, initial state), PH_STEP (decoder layers + glue of
// step t), PH_END (the prior network for all steps).  The caller walks the phases; with several row groups it interleaves their
// steps so that the groups' chains advance side by side on their streams.
enum { PH_BEGIN = 0, PH_STEP = 1, PH_END = 2 };

static int rollout_forward_impl(const ha_humor_net* net, int B, int S, const float* past_in0, const float* z_seq,
                                float* world, float* prior_mu, float* prior_var, float* stash, hipStream_t st, int phase, int t) {
  StashLayout L;
  make_layout(net, B, S, L);
  const int RT = L.RT, rows = RT * 32;
  auto step_ptr = [&](int t) { return stash + L.steps + (size_t)t * L.per_step; };
  auto x_ptr = [&](int t) { return stash + L.xT + (size_t)t * RT * D_INP * 32; };

  if (L.single) {
    // the whole decoder chain (all S steps: layers, glue, world outputs, the stash the adjoint and the prior read) is ONE launch
    if (phase == PH_BEGIN) {
      PersistFwd f;
      f.B = B; f.S = S;
      f.past_in0 = past_in0; f.z_seq = z_seq; f.world = world;
      f.xT = stash + L.xT; f.steps = stash + L.steps; f.per_step = L.per_step; f.off_G = L.off_G;
      for (int l = 0; l < 4; ++l) f.off_dec[l] = L.off_dec[l];
      for (int l = 0; l < 3; ++l) { f.off_gn[l] = L.off_gn[l]; f.off_ht[l] = L.off_ht[l]; }
      f.off_gl = L.off_gl;
      for (int l = 0; l < 4; ++l) f.dec_pad[l] = net->dec[l].Nout_pad;
      f.hidden_slabs = tl_single_mode != 2;
      f.t2j = stash + L.t2j;
      f.ws = stash + L.persist_ws;
      return persist_forward(net->persist, f, ((g_rollout_persist >> 1) & 1) | (g_rollout_persist_inject ? 2 : 0), st);
    }
    if (phase == PH_STEP) return HA_OK;
  }
  if (phase == PH_BEGIN) {
    if (L.acc) zero_async(stash + L.steps, (size_t)(S + 1) * L.per_step * sizeof(float), st);
    HA_LAUNCH(transpose_in_kernel, dim3(256), dim3(256), 0, st, z_seq, stash + L.zT, B, S, ZD, ZD, RT);
    HA_LAUNCH_CHECK();
    HA_LAUNCH(transpose_in_kernel, dim3(64), dim3(256), 0, st, past_in0, x_ptr(0), B, 1, D_IN, D_INP, RT);
    HA_LAUNCH_CHECK();
    HA_LAUNCH(init_state_kernel, dim3(ceil_div(rows, 64)), dim3(64), 0, st, past_in0, step_ptr(0) + L.off_G, stash + L.t2j, B, rows);
    HA_LAUNCH_CHECK();
    return HA_OK;
  }

  const bool with_prior = prior_mu != nullptr;
  const int nd = net->n_dec, np = net->n_pri;
  // the recurrence: decoder layers + glue per step (the prior only consumes the states: it runs afterwards for all steps at once)
  if (phase == PH_STEP) {
    float* sp = step_ptr(t);
    const float* zT = stash + L.zT + (size_t)t * RT * ZD * 32;
    for (int l = 0; l < nd; ++l) {
      LayerLaunch LL;
      memset(&LL, 0, sizeof(LL));
      LL.RT = RT;
      const PackedLayer& P = net->dec[l];
      const float* src = l == 0 ? x_ptr(t) : sp + L.off_dec[l - 1];
      fwd_task(LL.t[LL.ntasks++], P, src, l == 0 ? 1 : L.rd_f(l - 1), zT, sp + L.off_dec[l], L.spb, L.nsf_dec[l]);
      set_acc(LL.t[0], L.acc);
      if (l > 0 && L.hsum) LL.t[0].hsum_dst = sp + L.off_hsum[l - 1];
      int rc = launch_layers(LL, L, stash, st);
      if (rc != HA_OK) return rc;
    }
    GlueParams g;
    memset(&g, 0, sizeof(g));
    g.B = B; g.S = S; g.t = t; g.RT = RT;
    g.xT = x_ptr(t);
    g.xT_next = x_ptr(t + 1);
    const PackedLayer& DL = net->dec[net->n_dec - 1];
    g.dec_out = sp + L.off_dec[net->n_dec - 1]; g.dec_nsplit = L.rd_f(net->n_dec - 1); g.dec_pad = DL.Nout_pad;
    g.Gs = sp + L.off_G;
    g.Gs_next = step_ptr(t + 1) + L.off_G;
    g.t2j = stash + L.t2j;
    g.world = world;
    launch_glue_fwd(net->rotw, net->delta, rows, st, g);
    HA_LAUNCH_CHECK();
    return HA_OK;
  }
  if (with_prior) {
    int rc = prior_forward_batched(net, L, stash, S, st);
    if (rc != HA_OK) return rc;
    PriorIOParams q;
    memset(&q, 0, sizeof(q));
    q.B = B; q.S = S; q.RT = RT;
    q.pri_pad = net->pri[np - 1].Nout_pad;
    q.pri_out0 = stash + L.pri_h[np - 1]; q.step_stride = (size_t)RT * q.pri_pad * 32; q.pri_nsplit = 1;
    q.prior_mu = prior_mu; q.prior_var = prior_var;
    HA_LAUNCH(prior_io_kernel, dim3(rows, S), dim3(64), 0, st, q);
    HA_LAUNCH_CHECK();
  }
  return HA_OK;
}

extern "C" int ha_humor_rollout_sample(const ha_humor_net* net, int B, int S, const float* past_in0, const float* eps_seq,
                                       float* world, float* prior_mu, float* prior_var, float* z_out, float* stash, void* stream) {
  HA_REQUIRE(net && past_in0 && world && prior_mu && prior_var && z_out && stash, "ha_humor_rollout_sample: null argument");
  HA_REQUIRE(B >= 1 && S >= 1, "ha_humor_rollout_sample: B and S must be >= 1");
  DeviceGuard guard(net->device);
  hipStream_t st = (hipStream_t)stream;
  StashLayout L;
  make_layout(net, B, S, L, false);
  const int RT = L.RT, rows = RT * 32;
  auto step_ptr = [&](int t) { return stash + L.steps + (size_t)t * L.per_step; };
  auto x_ptr = [&](int t) { return stash + L.xT + (size_t)t * RT * D_INP * 32; };
  HA_LAUNCH(transpose_in_kernel, dim3(64), dim3(256), 0, st, past_in0, x_ptr(0), B, 1, D_IN, D_INP, RT);
  HA_LAUNCH_CHECK();
  HA_LAUNCH(init_state_kernel, dim3(ceil_div(rows, 64)), dim3(64), 0, st, past_in0, step_ptr(0) + L.off_G, stash + L.t2j, B, rows);
  HA_LAUNCH_CHECK();
  const PackedLayer& DL = net->dec[net->n_dec - 1];
  const PackedLayer& PL = net->pri[net->n_pri - 1];
  for (int t = 0; t < S; ++t) {
    float* sp = step_ptr(t);
    float* zT = stash + L.zT + (size_t)t * RT * ZD * 32;
    // the latent of this step depends on the prior of this step: prior network first (on the recurrence here), then sample,
    // then the decoder
    for (int l = 0; l < net->n_pri; ++l) {
      LayerLaunch LL;
      memset(&LL, 0, sizeof(LL));
      LL.RT = RT;
      const float* src = l == 0 ? x_ptr(t) : stash + L.smp_pri[l - 1];
      fwd_task(LL.t[LL.ntasks++], net->pri[l], src, l == 0 ? 1 : L.nsf_pri[l - 1], nullptr, stash + L.smp_pri[l], L.spb, L.nsf_pri[l]);
      int rc = launch_layers(LL, L, stash, st);
      if (rc != HA_OK) return rc;
    }
    SampleParams sp_;
    memset(&sp_, 0, sizeof(sp_));
    sp_.B = B; sp_.S = S; sp_.t = t; sp_.RT = RT;
    sp_.pri_out = stash + L.smp_pri[net->n_pri - 1]; sp_.pri_nsplit = L.nsf_pri[net->n_pri - 1]; sp_.pri_pad = PL.Nout_pad;
    sp_.eps = eps_seq; sp_.zT_t = zT; sp_.z_out = z_out;
    HA_LAUNCH(sample_z_kernel, dim3(rows), dim3(64), 0, st, sp_);
    HA_LAUNCH_CHECK();
    for (int l = 0; l < net->n_dec; ++l) {
      LayerLaunch LL;
      memset(&LL, 0, sizeof(LL));
      LL.RT = RT;
      const float* src = l == 0 ? x_ptr(t) : sp + L.off_dec[l - 1];
      fwd_task(LL.t[LL.ntasks++], net->dec[l], src, l == 0 ? 1 : L.nsf_dec[l - 1], zT, sp + L.off_dec[l], L.spb, L.nsf_dec[l]);
      int rc = launch_layers(LL, L, stash, st);
      if (rc != HA_OK) return rc;
    }
    GlueParams g;
    memset(&g, 0, sizeof(g));
    g.B = B; g.S = S; g.t = t; g.RT = RT;
    g.xT = x_ptr(t);
    g.xT_next = x_ptr(t + 1);
    g.dec_out = sp + L.off_dec[net->n_dec - 1]; g.dec_nsplit = L.nsf_dec[net->n_dec - 1]; g.dec_pad = DL.Nout_pad;
    g.pri_out = stash + L.smp_pri[net->n_pri - 1]; g.pri_nsplit = L.nsf_pri[net->n_pri - 1]; g.pri_pad = PL.Nout_pad;
    g.Gs = sp + L.off_G;
    g.Gs_next = step_ptr(t + 1) + L.off_G;
    g.t2j = stash + L.t2j;
    g.world = world;
    g.prior_mu = prior_mu; g.prior_var = prior_var;
    launch_glue_fwd(net->rotw, net->delta, rows, st, g);
    HA_LAUNCH_CHECK();
  }
  return HA_OK;
}

// dL/d(prior_mu, prior_var) of all steps -> gx_pri (dL/dx_t through the prior network)
static int prior_adjoint_all(const ha_humor_net* net, const StashLayout& L, int B, int S, const float* g_prior_mu, const float* g_prior_var,
                             float* stash, hipStream_t st) {
  const int np = net->n_pri, RT = L.RT;
  PriorIOParams q;
  memset(&q, 0, sizeof(q));
  q.B = B; q.S = S; q.RT = RT;
  q.pri_pad = net->pri[np - 1].Nout_pad;
  q.pri_out0 = stash + L.pri_h[np - 1]; q.step_stride = (size_t)RT * q.pri_pad * 32; q.pri_nsplit = 1;
  q.g_prior_mu = g_prior_mu; q.g_prior_var = g_prior_var;
  q.g_pri_all = stash + L.g_pri_out;
  HA_LAUNCH(prior_io_kernel, dim3(RT * 32, S), dim3(64), 0, st, q);
  HA_LAUNCH_CHECK();
  return prior_backward_batched(net, L, stash, S, st);
}

static int rollout_backward_impl(const ha_humor_net* net, int B, int S, const float* g_world,
                                 const float* g_prior_mu, const float* g_prior_var, float* stash, float* g_past_in0,
                                 float* g_z_seq, hipStream_t st, int phase, int t) {
  StashLayout L;
  make_layout(net, B, S, L);
  const int RT = L.RT, rows = RT * 32;
  auto step_ptr = [&](int t) { return stash + L.steps + (size_t)t * L.per_step; };
  auto x_ptr = [&](int t) { return stash + L.xT + (size_t)t * RT * D_INP * 32; };
  const bool with_prior = g_prior_mu != nullptr || g_prior_var != nullptr;
  const PackedLayer& DL = net->dec[net->n_dec - 1];
  const int nd = net->n_dec;
  const int gxp_pad = net->pri[0].Nin_pad;
  // adjoint scratch of decoder layer l at step t (accumulate policy: one pre-zeroed set per step; else one set reused by every step)
  auto bd = [&](int l, int t) { return stash + L.bwd_dec[l] + (size_t)(t < 0 ? 0 : (t >= S ? S - 1 : t)) * L.bwd_set; };
  if (L.acc && phase == PH_BEGIN) zero_async(stash + L.bwd_begin, L.bwd_floats * sizeof(float), st);

  // the prior's contribution to dL/dx_t for every step, before the reverse scan (it does not depend on the scan)
  if (with_prior && phase == PH_BEGIN) {
    int rc = prior_adjoint_all(net, L, B, S, g_prior_mu, g_prior_var, stash, st);
    if (rc != HA_OK) return rc;
  }

  auto fill_glue = [&](GlueParams& g, int t) {
    memset(&g, 0, sizeof(g));
    g.B = B; g.S = S; g.t = t; g.RT = RT;
    g.t2j = stash + L.t2j;
    g.carry = stash + L.carry;
    g.g_z = g_z_seq;
    g.g_past0 = g_past_in0;
    // step t+1 products (consumed when t < S-1)
    g.gx_dir_in = stash + L.gx_dir[(t + 1) & 1];
    g.gx_dir_out = stash + L.gx_dir[t & 1];
    g.gxp_dec = bd(0, t + 1); g.gxp_dec_nsplit = L.rd_b(0); g.gxp_dec_pad = net->dec[0].Nin_pad;
    if (with_prior && t + 1 < S) {
      g.gxp_pri = stash + L.gx_pri + (size_t)(t + 1) * RT * gxp_pad * 32; g.gxp_pri_nsplit = 1; g.gxp_pri_pad = gxp_pad;
    }
    g.dz_n = net->n_dec;
    for (int i = 0; i < net->n_dec; ++i) {
      g.dz_src[i] = bd(i, t + 1); g.dz_nsplit[i] = L.rd_b(i); g.dz_pad[i] = net->dec[i].Nin_pad;
      g.dz_off[i] = net->dec[i].Cin;
    }
    if (t >= 0) {
      float* sp = step_ptr(t);
      g.xT = x_ptr(t);
      g.dec_out = sp + L.off_dec[net->n_dec - 1]; g.dec_nsplit = L.rd_f(net->n_dec - 1); g.dec_pad = DL.Nout_pad;
      g.Gs = sp + L.off_G;
      g.g_world = g_world;
      g.g_dec_out = stash + L.g_dec_out;
      g.g_pri_out = nullptr;      // prior-output adjoints of all steps come from prior_io_kernel
    }
  };

  const bool persist_bwd = adjoint_persistent(L, tl_single_mode);
  HA_REQUIRE(!(tl_single_mode == 2 && !(L.single && persist_usable(net->persist))),
             "ha_humor_rollout_backward: this stash was filled by a persistent / pipelined forward without launch-chain slabs (only the one-launch "
             "adjoint can read it) and the persistent path has been disabled since (a launch reported a failure, error word 0x%x): repeat the "
             "forward call -- it will run on the launch chain", persist_error_word(net->persist));
  if (persist_bwd) {
    if (phase == PH_BEGIN) {
      PersistBwd f;
      f.B = B; f.S = S;
      f.g_world = g_world;
      f.gx_pri = with_prior ? stash + L.gx_pri : nullptr; f.gxp_pad = gxp_pad;
      f.xT = stash + L.xT; f.steps = stash + L.steps; f.per_step = L.per_step; f.off_G = L.off_G;
      for (int l = 0; l < 4; ++l) f.off_dec[l] = L.off_dec[l];
      for (int l = 0; l < 3; ++l) { f.off_gn[l] = L.off_gn[l]; f.off_ht[l] = L.off_ht[l]; }
      f.off_gl = L.off_gl;
      for (int l = 0; l < 4; ++l) f.dec_pad[l] = net->dec[l].Nout_pad;
      f.t2j = stash + L.t2j;
      f.g_past0 = g_past_in0; f.g_z = g_z_seq; f.g_z_add = tl_gz_add;
      f.dz_part = stash + L.dz_part;
      f.ws = stash + L.persist_ws;
      return persist_backward(net->persist, f, (g_rollout_persist >> 1) & 1, st);
    }
    return HA_OK;
  }
  if (phase == PH_BEGIN) return HA_OK;
  if (phase == PH_STEP) {
    GlueParams g;
    fill_glue(g, t);
    launch_glue_bwd(net->rotw, net->delta, rows, st, g);
    HA_LAUNCH_CHECK();
    float* sp = step_ptr(t);
    for (int l = nd - 1; l >= 0; --l) {
      LayerLaunch LL;
      memset(&LL, 0, sizeof(LL));
      LL.RT = RT;
      const PackedLayer& P = net->dec[l];
      if (l == nd - 1)
        bwd_task(LL.t[LL.ntasks++], P, stash + L.g_dec_out, 1, P.Nout_pad, nullptr, nullptr, 0, bd(l, t), L.spb, L.nsb_dec[l]);
      else if (L.hsum && L.nsf_dec[l] > 1)      // the forward pass left the summed pre-activations of layer l as one slab
        bwd_task(LL.t[LL.ntasks++], P, bd(l + 1, t), L.rd_b(l + 1), net->dec[l + 1].Nin_pad, &net->dec[l + 1],
                 sp + L.off_hsum[l], 1, bd(l, t), L.spb, L.nsb_dec[l]);
      else
        bwd_task(LL.t[LL.ntasks++], P, bd(l + 1, t), L.rd_b(l + 1), net->dec[l + 1].Nin_pad, &net->dec[l + 1],
                 sp + L.off_dec[l], L.rd_f(l), bd(l, t), L.spb, L.nsb_dec[l]);
      set_acc(LL.t[0], L.acc);
      int rc = launch_layers(LL, L, stash, st);
      if (rc != HA_OK) return rc;
    }
    return HA_OK;
  }
  GlueParams g;
  fill_glue(g, -1);
  launch_glue_bwd(net->rotw, net->delta, rows, st, g);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

// ---- row groups on side streams ---------------------------------------------------------------------
namespace ha {

// side streams + fork/join events, created once per device
struct SidePool {
  bool ready = false;
  hipStream_t stream[MAX_GROUPS];
  hipEvent_t fork, join[MAX_GROUPS];
};
static SidePool g_side[16];

static int side_pool(int device, SidePool** out) {
  HA_REQUIRE(device >= 0 && device < 16, "roll-out: device index %d out of range", device);
  SidePool& P = g_side[device];
  if (!P.ready) {
    HA_CHECK_HIP(hipEventCreateWithFlags(&P.fork, hipEventDisableTiming));
    for (int i = 0; i < MAX_GROUPS; ++i) {
      HA_CHECK_HIP(hipStreamCreateWithFlags(&P.stream[i], hipStreamNonBlocking));
      HA_CHECK_HIP(hipEventCreateWithFlags(&P.join[i], hipEventDisableTiming));
    }
    P.ready = true;
  }
  *out = &P;
  return HA_OK;
}

// runs fn(group, first row, rows, stream, phase, t) over every group and phase: group 0 on the caller's stream, the others on side
// streams that wait for everything already queued on the caller's stream and are joined back into it.  The steps are issued
// round-robin over the groups (the host issues ~3 us per launch, a chain advances one launch per ~6 us: issuing one group's
// whole chain first would leave the others waiting for the host).
template <typename F>
static int for_each_group(int device, int B, int S, bool reverse, hipStream_t st, bool pipelined, F&& fn) {
  int ng, rpg;
  group_plan(B, ng, rpg, pipelined);
  if (pipelined) {
    // one persistent launch (per direction) per chunk of <= 256 sequences, in stream order
    for (int g = 0; g < ng; ++g) {
      const int r0 = g * rpg, rows = (B - r0) < rpg ? (B - r0) : rpg;
      int rc = fn(g, r0, rows, st, PH_BEGIN, 0);
      // (the step phases are no-ops behind a persistent launch; the launch-chain adjoint behind a pipelined forward walks them)
      for (int i = 0; i < S && rc == HA_OK; ++i) rc = fn(g, r0, rows, st, PH_STEP, reverse ? S - 1 - i : i);
      if (rc == HA_OK) rc = fn(g, r0, rows, st, PH_END, 0);
      if (rc != HA_OK) return rc;
    }
    return HA_OK;
  }
  struct GroupScope { int prev; explicit GroupScope(int n) : prev(tl_groups) { tl_groups = n; } ~GroupScope() { tl_groups = prev; } } scope(ng);
  SidePool* P = nullptr;
  if (ng > 1) {
    int rc = side_pool(device, &P);
    if (rc != HA_OK) return rc;
    HA_CHECK_HIP(hipEventRecord(P->fork, st));
    for (int g = 1; g < ng; ++g) HA_CHECK_HIP(hipStreamWaitEvent(P->stream[g], P->fork, 0));
  }
  auto run = [&](int phase, int t) {
    for (int g = 0; g < ng; ++g) {
      const int r0 = g * rpg, rows = ng == 1 ? B : ((B - r0) < rpg ? (B - r0) : rpg);
      const int rc = fn(g, r0, rows, g == 0 ? st : P->stream[g], phase, t);
      if (rc != HA_OK) return rc;
    }
    return (int)HA_OK;
  };
  int rc = run(PH_BEGIN, 0);
  for (int i = 0; i < S && rc == HA_OK; ++i) rc = run(PH_STEP, reverse ? S - 1 - i : i);
  if (rc == HA_OK) rc = run(PH_END, 0);
  if (rc != HA_OK) return rc;
  for (int g = 1; g < ng; ++g) {
    HA_CHECK_HIP(hipEventRecord(P->join[g], P->stream[g]));
    HA_CHECK_HIP(hipStreamWaitEvent(st, P->join[g], 0));
  }
  return HA_OK;
}

}  // namespace ha

extern "C" int ha_humor_rollout_forward(const ha_humor_net* net, int B, int S, const float* past_in0, const float* z_seq,
                                        float* world, float* prior_mu, float* prior_var, float* stash, void* stream) {
  HA_REQUIRE(net && past_in0 && z_seq && world && stash, "ha_humor_rollout_forward: null argument");
  HA_REQUIRE(B >= 1 && S >= 1, "ha_humor_rollout_forward: B and S must be >= 1");
  HA_REQUIRE((prior_mu == nullptr) == (prior_var == nullptr), "ha_humor_rollout_forward: prior_mu and prior_var go together");
  if (persist_take_failure(net->persist)) {
    set_error("ha_humor_rollout_forward: an earlier persistent roll-out launch on this network failed (error word 0x%x: its bounded waits ran out -- was "
              "another kernel holding part of the GPU?); the results of that call are invalid.  This network now uses the launch chain.",
              persist_error_word(net->persist));
    return HA_ERR_HIP;
  }
  DeviceGuard guard(net->device);
  // the roll-out mode of this call, decided once (see ha_humor_net::stash_mode)
  int mode = (g_rollout_persist != 0 && persist_usable(net->persist)) ? 1 : 0;
  const bool piped = pipelined_call(B, mode);
  // mode 2: a pipelined forward whose adjoint will be the pipelined launch too writes the hidden pre-activations in the teams' layout only
  // (no launch-chain slabs: a third of its stores); only the pipelined adjoint can read such a stash, and the backward entry checks that
  // (the same for the B <= 32 kernels)
  if (piped && g_rollout_persist_bwd != 0 && g_rollout_pipe_bwd != 0) mode = 2;
  if (!piped && mode == 1 && B <= 32 && g_layer_finish != 2 && g_rollout_persist_bwd != 0) mode = 2;
  int ng, rpg;
  group_plan(B, ng, rpg, piped);
  tl_groups = piped ? 1 : ng;
  const size_t gs = ng > 1 ? group_stash_floats(net, rpg, S) : 0;
  tl_groups = 1;
  // (one entry per distinct stash address ever seen -- the allocator recycles them, a forward overwrites its entry -- so the map stays
  // small; the wholesale clear is a backstop that a process would need 64 k live stashes to reach)
  if (net->stash_mode.size() > 65536) net->stash_mode.clear();
  net->stash_mode[stash] = ha_humor_net::StashRec{mode, B, S, layout_knobs()};
  tl_single_mode = mode;
  int rc;
  rc = for_each_group(net->device, B, S, false, (hipStream_t)stream, piped, [&](int g, int r0, int rows, hipStream_t st, int phase, int t) {
    const size_t r = (size_t)r0;
    return rollout_forward_impl(net, rows, S, past_in0 + r * D_IN, z_seq + r * S * ZD, world + r * S * D_STATE,
                                prior_mu ? prior_mu + r * S * ZD : nullptr, prior_var ? prior_var + r * S * ZD : nullptr,
                                stash + (size_t)g * gs, st, phase, t);
  });
  tl_single_mode = -1;
  return rc;
}

// dst[i] += src[i] (the launch-chain path of ha_humor_rollout_backward_ex; the persistent path adds in its final reduction)
__global__ void add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

extern "C" int ha_humor_rollout_backward(const ha_humor_net* net, int B, int S, const float* z_seq, const float* g_world,
                                         const float* g_prior_mu, const float* g_prior_var, float* stash, float* g_past_in0,
                                         float* g_z_seq, void* stream) {
  return ha_humor_rollout_backward_ex(net, B, S, z_seq, g_world, g_prior_mu, g_prior_var, stash, g_past_in0, g_z_seq, nullptr, stream);
}

extern "C" int ha_humor_rollout_backward_ex(const ha_humor_net* net, int B, int S, const float* z_seq, const float* g_world,
                                            const float* g_prior_mu, const float* g_prior_var, float* stash, float* g_past_in0,
                                            float* g_z_seq, const float* g_z_add, void* stream) {
  HA_REQUIRE(net && stash && g_past_in0 && g_z_seq, "ha_humor_rollout_backward: null argument");
  HA_REQUIRE(B >= 1 && S >= 1, "ha_humor_rollout_backward: B and S must be >= 1");
  (void)z_seq;
  if (persist_take_failure(net->persist)) {
    set_error("ha_humor_rollout_backward: an earlier persistent roll-out launch on this network failed (error word 0x%x); its results are invalid.  "
              "This network now uses the launch chain.", persist_error_word(net->persist));
    return HA_ERR_HIP;
  }
  DeviceGuard guard(net->device);
  // the mode the forward over this stash recorded (a stash this library has not seen: decided from the live state, as before)
  const auto it = net->stash_mode.find(stash);
  HA_REQUIRE(it != net->stash_mode.end(), "ha_humor_rollout_backward: no forward call of this network has filled this stash");
  HA_REQUIRE(it->second.B == B && it->second.S == S, "ha_humor_rollout_backward: the stash was filled by a forward call of %d x %d, not %d x %d",
             it->second.B, it->second.S, B, S);
  HA_REQUIRE(it->second.knobs == layout_knobs(), "ha_humor_rollout_backward: a layout knob (layer_spb / layer_finish / layer_acc / layer_hsum / "
             "rollout_pipe / rollout_groups) changed between the forward call that filled this stash and its backward");
  tl_single_mode = it->second.mode;
  const bool piped = pipelined_call(B, tl_single_mode >= 0 ? tl_single_mode : ((g_rollout_persist != 0 && persist_usable(net->persist)) ? 1 : 0));
  int ng, rpg;
  group_plan(B, ng, rpg, piped);
  tl_groups = piped ? 1 : ng;
  const size_t gs = ng > 1 ? group_stash_floats(net, rpg, S) : 0;
  tl_groups = 1;
  int rc;
  bool add_in_kernel = false;
  if (g_z_add && tl_single_mode >= 1 && ng == 1) {
    StashLayout L;
    make_layout(net, B, S, L);
    add_in_kernel = adjoint_persistent(L, tl_single_mode);
  }
  tl_gz_add = add_in_kernel ? g_z_add : nullptr;
  rc = for_each_group(net->device, B, S, true, (hipStream_t)stream, piped, [&](int g, int r0, int rows, hipStream_t st, int phase, int t) {
    const size_t r = (size_t)r0;
    return rollout_backward_impl(net, rows, S, g_world ? g_world + r * S * D_STATE : nullptr,
                                 g_prior_mu ? g_prior_mu + r * S * ZD : nullptr, g_prior_var ? g_prior_var + r * S * ZD : nullptr,
                                 stash + (size_t)g * gs, g_past_in0 + r * D_IN, g_z_seq + r * S * ZD, st, phase, t);
  });
  tl_single_mode = -1;
  tl_gz_add = nullptr;
  if (rc == HA_OK && g_z_add && !add_in_kernel) {
    const int n = B * S * ZD;
    HA_LAUNCH(add_inplace_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, g_z_seq, g_z_add, n);
    HA_LAUNCH_CHECK();
  }
  return rc;
}

// ---- rotation conversions -------------------------------------------------------------------------
#define HA_ROT_ENTRY(NAME, KERNEL, ...)                                                   \
  do {                                                                                    \
    HA_REQUIRE(n >= 0, NAME ": n must be >= 0");                                          \
    if (n == 0) return HA_OK;                                                             \
    HA_LAUNCH(KERNEL, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, n, __VA_ARGS__); \
    HA_LAUNCH_CHECK();                                                                    \
    return HA_OK;                                                                         \
  } while (0)

extern "C" int ha_rodrigues_fwd(int n, const float* aa, float* R, void* stream) {
  HA_REQUIRE(aa && R, "ha_rodrigues_fwd: null argument");
  HA_ROT_ENTRY("ha_rodrigues_fwd", rodrigues_fwd_kernel, aa, R);
}
extern "C" int ha_rodrigues_bwd(int n, const float* aa, const float* gR, float* g_aa, void* stream) {
  HA_REQUIRE(aa && gR && g_aa, "ha_rodrigues_bwd: null argument");
  HA_ROT_ENTRY("ha_rodrigues_bwd", rodrigues_bwd_kernel, aa, gR, g_aa);
}
extern "C" int ha_rotmat_to_aa_fwd(int n, const float* R, float* aa, void* stream) {
  HA_REQUIRE(R && aa, "ha_rotmat_to_aa_fwd: null argument");
  HA_ROT_ENTRY("ha_rotmat_to_aa_fwd", rotmat_to_aa_fwd_kernel, R, aa);
}
extern "C" int ha_rotmat_to_aa_bwd(int n, const float* R, const float* g_aa, float* gR, void* stream) {
  HA_REQUIRE(R && g_aa && gR, "ha_rotmat_to_aa_bwd: null argument");
  HA_ROT_ENTRY("ha_rotmat_to_aa_bwd", rotmat_to_aa_bwd_kernel, R, g_aa, gR);
}
extern "C" int ha_rot6d_to_rotmat_fwd(int n, const float* x, float* R, void* stream) {
  HA_REQUIRE(x && R, "ha_rot6d_to_rotmat_fwd: null argument");
  HA_ROT_ENTRY("ha_rot6d_to_rotmat_fwd", rot6d_fwd_kernel, x, R);
}
extern "C" int ha_rot6d_to_rotmat_bwd(int n, const float* x, const float* gR, float* gx, void* stream) {
  HA_REQUIRE(x && gR && gx, "ha_rot6d_to_rotmat_bwd: null argument");
  HA_ROT_ENTRY("ha_rot6d_to_rotmat_bwd", rot6d_bwd_kernel, x, gR, gx);
}
extern "C" int ha_rot9d_to_rotmat_fwd(int n, const float* x, float* R, void* stream) {
  HA_REQUIRE(x && R, "ha_rot9d_to_rotmat_fwd: null argument");
  HA_ROT_ENTRY("ha_rot9d_to_rotmat_fwd", rot9d_fwd_kernel, x, R);
}
extern "C" int ha_rot9d_to_rotmat_bwd(int n, const float* x, const float* gR, float* gx, void* stream) {
  HA_REQUIRE(x && gR && gx, "ha_rot9d_to_rotmat_bwd: null argument");
  HA_ROT_ENTRY("ha_rot9d_to_rotmat_bwd", rot9d_bwd_kernel, x, gR, gx);
}

// ---------------------------------------------------------------------------------------------------
// Frozen MLPs on N independent rows: the VPoser decoder / encoder that bracket every stage-3 evaluation (motion_optimizer.py:
// 1041-1063: Linear + LeakyReLU(0.2), the decoder followed by 6-D -> rotation matrix -> axis-angle) and HuMoR's posterior encoder
// (humor_model.py:180-190: Linear + GroupNorm(16) + ReLU, evaluated once per fit by infer_global_seq).  Same packed layers and the
// same batched GEMM kernel as the prior network: one launch per layer over all ceil(N/32) row tiles, the activation function (or its
// adjoint) in the epilogue.
// ---------------------------------------------------------------------------------------------------
struct ha_mlp {
  int device = 0, n = 0, act = 0, in_dim = 0, out_dim = 0, in_pad = 0, wmax = 0;
  float slope = 0.f;
  ha::PackedLayer L[ha::MAXL];
};

namespace ha {

struct MlpLayout {
  int nrt = 0;
  size_t xT = 0, keep[MAXL], pp[2], out = 0, gx = 0, total = 0;
};

static MlpLayout mlp_layout(const ha_mlp* m, int N) {
  MlpLayout W;
  W.nrt = ceil_div(N, 32);
  size_t o = 0;
  auto take = [&](size_t ch) { const size_t at = o; o += (size_t)W.nrt * ch * 32; return at; };
  W.xT = take(m->in_pad);
  for (int l = 0; l + 1 < m->n; ++l) W.keep[l] = take(m->L[l].Nout_pad);      // act 0: pre-activations, act 1: LeakyReLU outputs
  W.pp[0] = take(m->wmax);
  W.pp[1] = take(m->wmax);
  W.out = take(m->L[m->n - 1].Nout_pad);
  W.gx = take(m->L[0].Nin_pad);
  W.total = o;
  return W;
}

// output slab [nrt][Cpad][32] -> y: tail 0 row-major [N][C]; tail 1 every 6 channels are a 6-D rotation -> axis-angle, [N][C/6][3]
__global__ void mlp_out_kernel(const float* __restrict__ slab, float* __restrict__ y, int N, int C, int Cpad, int tail) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tail == 0) {
    if (i >= (size_t)N * C) return;
    const int r = (int)(i / C), c = (int)(i % C);
    y[i] = slab[(size_t)(r >> 5) * Cpad * 32 + (size_t)(r & 31) * 4 + qoff(c)];
  } else {
    const int J = C / 6;
    if (i >= (size_t)N * J) return;
    const int r = (int)(i / J), j = (int)(i % J);
    const float* s = slab + (size_t)(r >> 5) * Cpad * 32 + (size_t)(r & 31) * 4;
    float v[6], R[9], aa[3];
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = s[qoff(6 * j + k)];
    rot6d_to_rotmat(v, R);
    rotmat_to_aa(R, aa);
#pragma unroll
    for (int k = 0; k < 3; ++k) y[i * 3 + k] = aa[k];
  }
}

// g_y -> adjoint slab [nrt][Cpad][32] of the network output (zero in padded rows / channels); tail 1 goes back through
// axis-angle <- rotation matrix <- 6-D with the forward's raw output (`slab`)
__global__ void mlp_gout_kernel(const float* __restrict__ g_y, const float* __restrict__ slab, float* __restrict__ g, int N, int nrt, int C,
                                int Cpad, int tail) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tail == 0) {
    if (i >= (size_t)nrt * 32 * Cpad) return;
    const int r = (int)(i / Cpad), c = (int)(i % Cpad);
    g[(size_t)(r >> 5) * Cpad * 32 + (size_t)(r & 31) * 4 + qoff(c)] = (r < N && c < C) ? g_y[(size_t)r * C + c] : 0.f;
  } else {
    const int J = C / 6, Jp = (Cpad + 5) / 6;
    if (i >= (size_t)nrt * 32 * Jp) return;
    const int r = (int)(i / Jp), j = (int)(i % Jp);
    const size_t base = (size_t)(r >> 5) * Cpad * 32 + (size_t)(r & 31) * 4;
    float o[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < N && j < J) {
      float v[6], R[9], gR[9];
#pragma unroll
      for (int k = 0; k < 6; ++k) v[k] = slab[base + qoff(6 * j + k)];
      rot6d_to_rotmat(v, R);
      const float ga[3] = {g_y[((size_t)r * J + j) * 3], g_y[((size_t)r * J + j) * 3 + 1], g_y[((size_t)r * J + j) * 3 + 2]};
      rotmat_to_aa_bwd(R, ga, gR);
      rot6d_to_rotmat_bwd(v, gR, o);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k)
      if (6 * j + k < Cpad) g[base + qoff(6 * j + k)] = o[k];
  }
}

}  // namespace ha

extern "C" int ha_mlp_create(ha_mlp** out, int device, const ha_mlp_desc* desc, int act, float slope) {
  HA_REQUIRE(out && desc, "ha_mlp_create: null argument");
  HA_REQUIRE(act == HA_MLP_GN_RELU || act == HA_MLP_LEAKY_RELU, "ha_mlp_create: act must be HA_MLP_GN_RELU or HA_MLP_LEAKY_RELU (got %d)", act);
  HA_REQUIRE(desc->skip_dim == 0, "ha_mlp_create: skip connections are only supported by the roll-out network");
  HA_REQUIRE(desc->n_linear >= 1 && desc->n_linear <= MAXL, "ha_mlp_create: n_linear=%d out of range", desc->n_linear);
  HA_REQUIRE(act != HA_MLP_LEAKY_RELU || slope > 0.f, "ha_mlp_create: the LeakyReLU slope must be > 0 (the adjoint reads the sign from the activation)");
  DeviceGuard guard(device);
  HA_REQUIRE(guard.ok, "ha_mlp_create: cannot select device %d", device);
  ha_mlp* m = new ha_mlp();
  m->device = device; m->n = desc->n_linear; m->act = act; m->slope = slope;
  m->in_dim = desc->in_dim; m->out_dim = desc->out_dims[desc->n_linear - 1];
  m->in_pad = ceil_div(desc->in_dim, 4) * 4;
  int rc = HA_OK;
  if (act == HA_MLP_GN_RELU) rc = pack_mlp(m->L, desc, "ha_mlp_create");
  else {
    int cin = desc->in_dim;
    for (int i = 0; i < desc->n_linear && rc == HA_OK; ++i) {
      if (!(desc->w[i] && desc->b[i])) { ha::set_error("ha_mlp_create: layer %d weights missing", i); rc = HA_ERR_INVALID_ARG; break; }
      rc = pack_layer(m->L[i], desc->w[i], desc->b[i], nullptr, nullptr, cin, 0, desc->out_dims[i], false);
      cin = desc->out_dims[i];
    }
  }
  if (rc != HA_OK) { ha_mlp_destroy(m); return rc; }
  for (int i = 0; i < m->n; ++i) {
    if (m->L[i].Nout_pad > m->wmax) m->wmax = m->L[i].Nout_pad;
    if (m->L[i].Nin_pad > m->wmax) m->wmax = m->L[i].Nin_pad;
  }
  *out = m;
  return HA_OK;
}

extern "C" int ha_mlp_destroy(ha_mlp* m) {
  if (!m) return HA_OK;
  DeviceGuard guard(m->device);
  for (int i = 0; i < MAXL; ++i) {
    PackedLayer& P = m->L[i];
    float* ptrs[5] = {P.Wf, P.Wb, P.bias, P.gamma, P.beta};
    for (float* p : ptrs)
      if (p) (void)hipFree(p);
  }
  delete m;
  return HA_OK;
}

extern "C" int ha_mlp_workspace(const ha_mlp* m, int N, int64_t* ws_floats) {
  HA_REQUIRE(m && ws_floats && N >= 1, "ha_mlp_workspace: bad argument");
  *ws_floats = (int64_t)mlp_layout(m, N).total;
  return HA_OK;
}

extern "C" int ha_mlp_forward(const ha_mlp* m, int N, const float* x, int tail, float* y, float* ws, void* stream) {
  HA_REQUIRE(m && x && y && ws && N >= 1, "ha_mlp_forward: bad argument");
  HA_REQUIRE(tail == HA_MLP_TAIL_NONE || (tail == HA_MLP_TAIL_ROT6D_AA && m->out_dim % 6 == 0), "ha_mlp_forward: tail %d needs an output width that is a multiple of 6", tail);
  DeviceGuard guard(m->device);
  hipStream_t st = (hipStream_t)stream;
  const MlpLayout W = mlp_layout(m, N);
  HA_LAUNCH(transpose_in_kernel, dim3(ceil_div((int)((size_t)W.nrt * m->in_pad * 32), 256)), dim3(256), 0, st, x, ws + W.xT, N, 1, m->in_dim,
                     m->in_pad, W.nrt);
  HA_LAUNCH_CHECK();
  for (int l = 0; l < m->n; ++l) {
    const PackedLayer& P = m->L[l];
    const bool last = l + 1 == m->n;
    GemmTask T;
    memset(&T, 0, sizeof(T));
    T.Wp = P.Wf; T.bias = P.bias;
    T.ntiles = P.ntiles_f; T.nslices = P.nslices_f; T.Nout = P.Nout;
    T.nrt = W.nrt; T.Cdst = P.Nout_pad;
    T.Csrc = l == 0 ? m->in_pad : m->L[l - 1].Nout_pad;
    if (m->act == HA_MLP_LEAKY_RELU) {
      T.src = l == 0 ? ws + W.xT : ws + W.keep[l - 1];
      if (last) { T.epi = 0; T.dst_h = ws + W.out; }
      else { T.epi = 4; T.slope = m->slope; T.dst_a = ws + W.keep[l]; }
    } else {
      T.src = l == 0 ? ws + W.xT : ws + W.pp[(l - 1) & 1];
      if (last) { T.epi = 0; T.dst_h = ws + W.out; }
      else {
        const PackedLayer& Nx = m->L[l + 1];
        T.epi = 1; T.gamma = Nx.gamma; T.beta = Nx.beta; T.group = Nx.group;
        T.dst_h = ws + W.keep[l]; T.dst_a = ws + W.pp[l & 1];
      }
    }
    int rc = launch_prior_gemm(T, st);
    if (rc != HA_OK) return rc;
  }
  const int C = m->out_dim, Cpad = m->L[m->n - 1].Nout_pad;
  const size_t items = tail == HA_MLP_TAIL_NONE ? (size_t)N * C : (size_t)N * (C / 6);
  HA_LAUNCH(mlp_out_kernel, dim3((unsigned)ceil_div((int)items, 256)), dim3(256), 0, st, ws + W.out, y, N, C, Cpad, tail);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

extern "C" int ha_mlp_backward(const ha_mlp* m, int N, const float* g_y, int tail, float* ws, float* g_x, void* stream) {
  HA_REQUIRE(m && g_y && g_x && ws && N >= 1, "ha_mlp_backward: bad argument");
  HA_REQUIRE(tail == HA_MLP_TAIL_NONE || (tail == HA_MLP_TAIL_ROT6D_AA && m->out_dim % 6 == 0), "ha_mlp_backward: tail %d needs an output width that is a multiple of 6", tail);
  DeviceGuard guard(m->device);
  hipStream_t st = (hipStream_t)stream;
  const MlpLayout W = mlp_layout(m, N);
  const int C = m->out_dim, Cpad = m->L[m->n - 1].Nout_pad;
  // the adjoint of the output goes to pp[(n-1)&1]; layer l's task reads pp[l&1] and writes pp[(l-1)&1]
  float* g_out = ws + W.pp[(m->n - 1) & 1];
  const size_t items = tail == HA_MLP_TAIL_NONE ? (size_t)W.nrt * 32 * Cpad : (size_t)W.nrt * 32 * ceil_div(Cpad, 6);
  HA_LAUNCH(mlp_gout_kernel, dim3((unsigned)ceil_div((int)items, 256)), dim3(256), 0, st, g_y, ws + W.out, g_out, N, W.nrt, C, Cpad, tail);
  HA_LAUNCH_CHECK();
  for (int l = m->n - 1; l >= 0; --l) {
    const PackedLayer& P = m->L[l];
    GemmTask T;
    memset(&T, 0, sizeof(T));
    T.Wp = P.Wb; T.bias = nullptr;
    T.ntiles = P.ntiles_b; T.nslices = P.nslices_b; T.Nout = P.Cin;
    T.src = ws + W.pp[l & 1]; T.Csrc = P.Nout_pad;
    T.nrt = W.nrt; T.Cdst = P.Nin_pad;
    if (l == 0) { T.epi = 0; T.dst_h = ws + W.gx; }
    else {
      T.hsrc = ws + W.keep[l - 1]; T.Ch = m->L[l - 1].Nout_pad;
      T.dst_a = ws + W.pp[(l - 1) & 1];
      if (m->act == HA_MLP_LEAKY_RELU) { T.epi = 5; T.slope = m->slope; }
      else { T.epi = 3; T.gamma = P.gamma; T.beta = P.beta; T.group = P.group; }
    }
    int rc = launch_prior_gemm(T, st);
    if (rc != HA_OK) return rc;
  }
  HA_LAUNCH(mlp_out_kernel, dim3((unsigned)ceil_div((int)((size_t)N * m->in_dim), 256)), dim3(256), 0, st, ws + W.gx, g_x, N, m->in_dim,
                     m->L[0].Nin_pad, 0);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

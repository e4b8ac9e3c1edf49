// Persistent, weight-stationary forward of the HuMoR decoder roll-out on gfx950 (humor/models/humor_model.py:785-1017 for
// B <= 32 sequences): ONE launch for all S steps instead of 5 dependent launches per step.
//
// Why: at 32 rows the launch chain is a latency chain (5 dispatches x ~6.5 us per step; the 138 MFLOP of a step are ~1 us of the
// chip's fp32 matrix rate).  The structure that removes the dispatches without a whole-chip barrier per layer:
//   * the sequences are independent, so the batch is cut into 8 TEAMS of 4 sequences, one team per XCD (32 CUs, one L2):
//     a team never talks to another team -- no cross-XCD traffic, no grid barrier;
//   * the decoder's 2.16 M fp32 weights (8.7 MB) do not fit one CU, but they fit the REGISTER FILES of one XCD: every wave of a
//     team (32 CUs x 4 SIMDs, one wave per SIMD, 512 VGPRs) keeps its 286-register share of the four layers for the whole launch
//     (8 x 8.7 MB = 70 MB of the chip's 128 MB of VGPRs) -- per step only activations move;
//   * 4 rows are the native M of v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products per instruction = 16 k's of one
//     4-row x 4-column tile): a wave owns 8 (4) output columns of a layer and the WHOLE K, so there is no cross-wave reduction,
//     only a 16-lane-group reduction of the accumulators;
//   * a layer's output (4 rows x <=1024 channels) goes to the other 31 CUs of the team as 8-byte {value, tag} granules through
//     the team's L2: the data is its own flag (tag = 4 step + layer + 1), consumers sweep it with L1-bypassing (sc1) loads until
//     every tag matches -- one memory round trip per layer, no counters, no fences; GroupNorm + ReLU run on the consumer side
//     (every CU needs the whole activation as its A operand anyway), the residual composition / heading alignment / frame changes
//     of the step ("glue") run redundantly on every CU, one wave per sequence.
// Teams are formed at run time from HW_REG_XCC_ID (a block joins the team of the XCD it landed on), so results never depend on
// the dispatcher's placement; what the path needs is 32 resident blocks per XCD -- every wait is bounded, a team that does not
// complete reports through a host-mapped error word and the caller falls back to the launch chain.
//
// The kernel fills the same stash as the launch-chain forward (states of all steps, accumulated world transforms, one
// pre-activation slab per decoder layer and step), so the batched prior and the existing adjoint run unchanged behind it.
#include "rollout_persist.h"

#include <string.h>

#include <utility>
#include <vector>

#include "rot_math.h"
#ifndef HA_SIMT_EMU
#include "lane_reduce.h"
#endif

namespace ha {

namespace {
constexpr int P_DIN = 339, P_DINP = 340, P_XPAD = 352, P_ZD = 48, P_RAW = 216, P_RAWPAD = 224, P_STATE = 348;
constexpr int P_H0 = 1024, P_H1 = 1024, P_H2 = 512;
constexpr int TEAM_CUS = 32, NTEAMS = 8, ROWS = 4, NWAVES_TEAM = TEAM_CUS * 4;
// K chunks (16 k's each) per layer: main part + 3 chunks of latent skip
constexpr int NC0 = P_XPAD / 16, NC1 = P_H0 / 16, NC2 = P_H1 / 16, NC3 = P_H2 / 16, NCZ = P_ZD / 16;
// weight registers of a wave: [chunk][column group] per layer
constexpr int R0 = 0, R1 = R0 + (NC0 + NCZ) * 2, R2 = R1 + (NC1 + NCZ) * 2, R3 = R2 + (NC2 + NCZ), NREG = R3 + (NC3 + NCZ);
constexpr int NWA = 200;                // weight registers kept in AGPRs (the MFMAs' B operand reads them there); the rest in VGPRs
constexpr int L3_WAVES = P_RAW / 4;      // 54 waves own the 216 output columns of the last layer
// exchange space (bytes): 64-byte header, then per team the four activations as granules [channel][4 rows] x 8 B
constexpr unsigned XCH_HDR = 256;
constexpr unsigned ACT_OFF0 = 0, ACT_OFF1 = ACT_OFF0 + P_H0 * 32, ACT_OFF2 = ACT_OFF1 + P_H1 * 32, ACT_OFF3 = ACT_OFF2 + P_H2 * 32;
constexpr unsigned TEAM_BYTES = ACT_OFF3 + P_RAWPAD * 32;
constexpr unsigned XCH_BYTES = XCH_HDR + NTEAMS * TEAM_BYTES;
// LDS (floats)
constexpr int L_XS0 = 0, L_XS1 = L_XS0 + P_XPAD * 4, L_XS2 = L_XS1 + P_H0 * 4, L_XS3 = L_XS2 + P_H1 * 4, L_ZS = L_XS3 + P_H2 * 4;
constexpr int L_SX = L_ZS + P_ZD * 4, L_SRAW = L_SX + ROWS * P_XPAD, L_SW = L_SRAW + ROWS * P_RAWPAD, L_SG = L_SW + ROWS * P_XPAD;
constexpr int L_DUMMY = L_SG + ROWS * 12, L_MISC = L_DUMMY + P_XPAD * 4, L_TOTAL = L_MISC + 64;      // (L_DUMMY: sink of group 1's second store)
// per-step results of a team (world states, next state slab, accumulated transforms) leave through all 32 CUs, COPY_PER_CU floats each
constexpr int COPY_WORLD = ROWS * P_STATE, COPY_XT = ROWS * P_DINP, COPY_G = ROWS * 12, COPY_TOTAL = COPY_WORLD + COPY_XT + COPY_G;
constexpr int COPY_PER_CU = (COPY_TOTAL + TEAM_CUS - 1) / TEAM_CUS;
constexpr int SPIN_LIMIT = 40000;        // bounded waits (~1 us per spin)
}  // namespace

size_t persist_ws_floats() { return (XCH_BYTES + 3) / 4; }

struct PersistArgs {
  int B, S;
  const float* Wreg;        // [128 waves][NREG][64 lanes]
  const float* bias[4];     // [1024] [1024] [512] [224]
  const float* gamma[3];    // GroupNorm affine of the inputs of layers 1..3
  const float* beta[3];
  const float* past_in0;
  const float* z_seq;
  float* world;
  float* xT;
  float* steps;
  size_t per_step, off_G, off_dec[4];
  float* t2j;
  unsigned char* xch;
  unsigned* err;            // host-mapped error word
};

struct PersistNet {
  int device = 0;
  float* Wreg = nullptr;
  float* bias[4] = {nullptr, nullptr, nullptr, nullptr};
  float* gamma[3] = {nullptr, nullptr, nullptr};
  float* beta[3] = {nullptr, nullptr, nullptr};
  unsigned* err_host = nullptr;    // hipHostMalloc'ed, mapped
  unsigned* err_dev = nullptr;
  bool disabled = false;
  long long launches = 0;
};

#ifndef HA_SIMT_EMU

typedef float pvf4 __attribute__((ext_vector_type(4)));
typedef unsigned puv4 __attribute__((ext_vector_type(4)));

#ifdef HA_PERSIST_TIMING
// profiling build only (tools/persist_phase_timing.py): phase timestamps (s_memtime) of one wave for steps PT_T0 .. PT_T0 + 7
constexpr int PT_T0 = 8, PT_N = 24;
__device__ unsigned long long g_pts[2][8][PT_N];      // [0]: team 0 member 5 (an ordinary CU), [1]: team 0 member 0 (the writer)
#define PT(i)                                                                        \
  do {                                                                               \
    if (pt_on && t >= PT_T0 && t < PT_T0 + 8) g_pts[pt_slot][t - PT_T0][i] = clock64(); \
  } while (0)
#define PT_ARGS , bool pt_on, int pt_slot, int t
#define PT_PASS , pt_on, pt_slot, t
#else
#define PT(i)
#define PT_ARGS
#define PT_PASS
#endif

__device__ __forceinline__ size_t pq(int c) { return (size_t)(c >> 2) * 128 + (c & 3); }

__device__ __forceinline__ float as_f(unsigned u) { return __uint_as_float(u); }
__device__ __forceinline__ unsigned as_u(float f) { return __float_as_uint(f); }

// Sweeps this thread's NQ channels (channel tid + 256 q: 4 granules = 32 bytes each) until every tag matches; the loads bypass
// the CU's L1 (sc1).  Returns false when the wait ran out (a team member never published).
template <int NQ>
__device__ __forceinline__ bool sweep(__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned tag, bool active, int tid, float (&x)[NQ][4]) {
  for (int spins = 0;; ++spins) {
    unsigned diff = 0;          // (no short-circuit: one straight-line batch of loads and compares)
    if (active) {
      puv4 lo[NQ], hi[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        lo[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + (unsigned)(tid + 256 * q) * 32u, 0, 16);
        hi[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + (unsigned)(tid + 256 * q) * 32u + 16u, 0, 16);
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        diff |= (lo[q].y ^ tag) | (lo[q].w ^ tag) | (hi[q].y ^ tag) | (hi[q].w ^ tag);
        x[q][0] = as_f(lo[q].x); x[q][1] = as_f(lo[q].z); x[q][2] = as_f(hi[q].x); x[q][3] = as_f(hi[q].z);
      }
    }
    if (__all(diff == 0)) return true;
    if (spins > SPIN_LIMIT) return false;
    __builtin_amdgcn_s_sleep(1);
  }
}

// consumer side of a hidden activation: sweep, GroupNorm (two-pass statistics per row and group of GROUP channels) + ReLU, the
// finished A operand to LDS as [channel][4 rows]
template <int NQ, int GROUP, int PTI = 0>
__device__ __forceinline__ bool gather_norm(__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned tag, const float (&gam)[NQ], const float (&bet)[NQ],
                                            float* xs, int tid PT_ARGS) {
  float x[NQ][4];
  if (!sweep<NQ>(rs, off, tag, true, tid, x)) return false;
  PT(PTI);
  // two-pass statistics of the lane's NQ x 4 (channel quarter, row) values over their groups: all sums of a pass in one
  // reduce-scatter / all-gather (lane_reduce.h)
  const float inv_n = 1.0f / (float)GROUP;
  float mu[NQ * 4], var[NQ * 4];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int i = 0; i < 4; ++i) mu[4 * q + i] = x[q][i];
  if constexpr (GROUP == 64) lr::wave_sum16(mu);
  else lr::half_sum8(mu);
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      mu[4 * q + i] *= inv_n;
      const float d = x[q][i] - mu[4 * q + i];
      var[4 * q + i] = d * d;
    }
  if constexpr (GROUP == 64) lr::wave_sum16(var);
  else lr::half_sum8(var);
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    pvf4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // (v_rsq_f32 itself: the argument is >= 1e-5, so rsqrtf's denormal pre-scaling never applies and the result is the same)
      const float rstd = __builtin_amdgcn_rsqf(var[4 * q + i] * inv_n + 1e-5f);
      o[i] = fmaxf((x[q][i] - mu[4 * q + i]) * rstd * gam[q] + bet[q], 0.f);
    }
    *reinterpret_cast<pvf4*>(xs + (size_t)(tid + 256 * q) * 4) = o;
  }
  return true;
}

// one layer's share of a wave: all K chunks against the resident weight registers.  The A operands come from LDS in batches of
// MB chunks, the next batch's reads issued ahead of the current batch's MFMAs (the compiler otherwise waits for every read right
// before its first use: one LDS round trip per 4 MFMAs); two accumulators per column group break the 2-pass dependent chain.
constexpr int MB = 16;
template <int NC_MAIN, int FIRST>
__device__ __forceinline__ void load_a(const float* xs, const float* zs, int lane, float (&av)[MB]) {
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    const int c = FIRST + i;
    if (c < NC_MAIN) av[i] = xs[64 * c + lane];
    else if (c < NC_MAIN + NCZ) av[i] = zs[64 * (c - NC_MAIN) + lane];
  }
}
// The MFMAs are written as inline asm so that the weight operand can stay in the AGPR half of the wave's 512 registers: the B
// operand of an MFMA may be an AGPR, but the compiler only reads weights it has first copied to a VGPR (v_accvgpr_read + s_nop + MFMA
// through ONE temporary: 32 cycles per MFMA measured, against 8 of issue).  NWA weight registers live in AGPRs ("a"), the rest in
// VGPRs ("v").  8 accumulators (AGPR quads) per wave are used round-robin: a dependent accumulate has ~64 cycles of latency, and
// the asm statements are volatile so the rotation (distance 8 between two MFMAs on one accumulator) is kept as written.  The
// compiler's hazard recogniser does not look inside asm: the first MFMA of a chain takes the literal 0 as its C operand (no
// compiler-written accumulator is read), and mma_layer ends with s_nop 7 before the VALU reads the results.
template <class F, int... I>
__device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int R, bool FIRST>
__device__ __forceinline__ void mfma_w(pvf4& acc, float av, const float (&wa)[NWA], const float (&wv)[NREG - NWA]) {
  if constexpr (R < NWA) {
    if constexpr (FIRST) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, 0" : "=&a"(acc) : "v"(av), "a"(wa[R]));
    else asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(av), "a"(wa[R]));
  } else {
    if constexpr (FIRST) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, 0" : "=&a"(acc) : "v"(av), "v"(wv[R - NWA]));
    else asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(av), "v"(wv[R - NWA]));
  }
}
template <int NC_MAIN, int NCG, int ROFF, int FIRST>
__device__ __forceinline__ void mma_batch(const float (&av)[MB], const float (&wa)[NWA], const float (&wv)[NREG - NWA], pvf4 (&acc)[NCG][8 / NCG]) {
  constexpr int NACC = 8 / NCG;
  // (compile-time chunk index: the weight register and the accumulator are template arguments of mfma_w)
  auto body = [&](auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr int c = FIRST + i;
    if constexpr (c < NC_MAIN + NCZ) {
      if constexpr (NCG == 2) {
        mfma_w<ROFF + c * 2 + 0, (c < NACC)>(acc[0][c % NACC], av[i], wa, wv);
        mfma_w<ROFF + c * 2 + 1, (c < NACC)>(acc[1][c % NACC], av[i], wa, wv);
      } else {
        mfma_w<ROFF + c, (c < NACC)>(acc[0][c % NACC], av[i], wa, wv);
      }
    }
  };
  static_for(body, std::make_integer_sequence<int, MB>{});
}
template <int NC_MAIN, int NCG, int ROFF, int FIRST>
struct MmaSteps {
  static __device__ __forceinline__ void run(const float* xs, const float* zs, const float (&wa)[NWA], const float (&wv)[NREG - NWA], int lane,
                                             float (&cur)[MB], pvf4 (&acc)[NCG][8 / NCG]) {
    if constexpr (FIRST < NC_MAIN + NCZ) {
      float nxt[MB];
      if constexpr (FIRST + MB < NC_MAIN + NCZ) load_a<NC_MAIN, FIRST + MB>(xs, zs, lane, nxt);
      HA_SCHED_FENCE();
      mma_batch<NC_MAIN, NCG, ROFF, FIRST>(cur, wa, wv, acc);
      HA_SCHED_FENCE();
      MmaSteps<NC_MAIN, NCG, ROFF, FIRST + MB>::run(xs, zs, wa, wv, lane, nxt, acc);
    }
  }
};
// result: sums[4 g + i] = the lane's partial (its k-block) of column group g, row i
template <int NC_MAIN, int NCG, int ROFF>
__device__ __forceinline__ void mma_layer(const float* xs, const float* zs, const float (&wa)[NWA], const float (&wv)[NREG - NWA], int lane,
                                          float (&sums)[4 * NCG]) {
  constexpr int NACC = 8 / NCG;
  static_assert(NC_MAIN + NCZ >= NACC, "every accumulator chain starts with a literal-0 MFMA");
  pvf4 acc[NCG][NACC];
  float first[MB];
  load_a<NC_MAIN, 0>(xs, zs, lane, first);
  MmaSteps<NC_MAIN, NCG, ROFF, 0>::run(xs, zs, wa, wv, lane, first, acc);
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");      // MFMA results -> VALU reads (hazard the compiler cannot see)
#pragma unroll
  for (int g = 0; g < NCG; ++g) {
#pragma unroll
    for (int st = NACC / 2; st >= 1; st /= 2)
#pragma unroll
      for (int k = 0; k < st; ++k) acc[g][k] += acc[g][k + st];
#pragma unroll
    for (int i = 0; i < 4; ++i) sums[4 * g + i] = acc[g][0][i];
  }
}

// publishes the wave's NCG column groups (columns col0 + 4 g + j) of layer-output `off`.  The k-block partials are summed with the
// reduce-scatter of lane_reduce.h, which leaves lane (half h, row parity p, column j, k-block 0) with
//   NCG = 2: column group h, rows 2 p and 2 p + 1  -> ONE 16-byte store of two {value, tag} granules,
//   NCG = 1: row 2 h + p                            -> ONE 8-byte granule;
// the same lanes write the plain pre-activation slab [channel quads][32 rows][4] for the adjoint / prior.
template <int NCG, bool SC1>
__device__ __forceinline__ void publish(const float (&sums)[4 * NCG], float bias_lane, int col0, unsigned char* team_xch, __amdgpu_buffer_rsrc_t rs,
                                        unsigned off, unsigned tag, float* slab, int row0, int lane) {
  const int h = lane >> 5, p = (lane >> 4) & 1, j = lane & 3;
  const bool storer = (lane & 12) == 0;
  if constexpr (NCG == 2) {
    float o[2];
    lr::block_sum8(sums, o);
    if (storer) {
      const int col = col0 + 4 * h + j;
      const float v0 = o[0] + bias_lane, v1 = o[1] + bias_lane;
      const puv4 gr = {as_u(v0), tag, as_u(v1), tag};
      const unsigned goff = off + (unsigned)col * 32u + (unsigned)p * 16u;
      if (SC1) __builtin_amdgcn_raw_buffer_store_b128(gr, rs, goff, 0, 16);
      else *reinterpret_cast<puv4*>(team_xch + goff) = gr;
      float* sp = slab + pq(col) + (size_t)(row0 + 2 * p) * 4;
      sp[0] = v0;
      sp[4] = v1;
    }
  } else {
    const float t = lr::block_sum4(sums);
    if (storer) {
      const int col = col0 + j, row = 2 * h + p;
      const float v0 = t + bias_lane;
      typedef unsigned puv2 __attribute__((ext_vector_type(2)));
      const puv2 gr = {as_u(v0), tag};
      const unsigned goff = off + (unsigned)col * 32u + (unsigned)row * 8u;
      if (SC1) __builtin_amdgcn_raw_buffer_store_b64(gr, rs, goff, 0, 16);
      else *reinterpret_cast<puv2*>(team_xch + goff) = gr;
      slab[pq(col) + (size_t)(row0 + row) * 4] = v0;
    }
  }
}

// rodrigues() of common.h with ONE argument reduction for sine and cosine (the glue is a dependent chain: every instruction counts)
__device__ __forceinline__ void rodrigues_sc(const float r[3], float R[9]) {
  const float ux = r[0] + 1e-8f, uy = r[1] + 1e-8f, uz = r[2] + 1e-8f;
  const float t = sqrtf(ux * ux + uy * uy + uz * uz);
  const float nx = r[0] / t, ny = r[1] / t, nz = r[2] / t;
  float s, c;
  sincosf(t, &s, &c);
  const float c1 = 1.0f - c;
  const float nn = nx * nx + ny * ny + nz * nz;
  R[0] = 1.0f + c1 * (nx * nx - nn);
  R[1] = -s * nz + c1 * (nx * ny);
  R[2] = s * ny + c1 * (nx * nz);
  R[3] = s * nz + c1 * (nx * ny);
  R[4] = 1.0f + c1 * (ny * ny - nn);
  R[5] = -s * nx + c1 * (ny * nz);
  R[6] = -s * ny + c1 * (nx * nz);
  R[7] = s * nx + c1 * (ny * nz);
  R[8] = 1.0f + c1 * (nz * nz - nn);
}
// w2a_fwd() of rot_math.h (heading alignment, transforms.py:17-42) through rodrigues_sc
__device__ __forceinline__ void w2a_sc(const float pR[9], float W[9]) {
  const float rx = -pR[0], ry = -pR[3];
  const float nrm = sqrtf(rx * rx + ry * ry);
  const float u = rx / (nrm + 1e-6f);
  const float angle = acosf(fminf(fmaxf(u, -1.0f), 1.0f));
  const float sg = -ry / (fabsf(ry) + 1e-6f);
  const float aa[3] = {0.f, 0.f, sg * angle};
  rodrigues_sc(aa, W);
}

template <bool SC1>
__global__ __launch_bounds__(256) HA_WAVES_PER_EU(1, 1) void rollout_persist_fwd_kernel(PersistArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs0 = smem + L_XS0;
  float* xs1 = smem + L_XS1;
  float* xs2 = smem + L_XS2;
  float* xs3 = smem + L_XS3;
  float* zs = smem + L_ZS;
  float* sX = smem + L_SX;
  float* sRAW = smem + L_SRAW;
  float* sW = smem + L_SW;
  float* sG = smem + L_SG;
  float* sDummy = smem + L_DUMMY;
  volatile int* misc = reinterpret_cast<volatile int*>(smem + L_MISC);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;

  // ---- team formation: a block belongs to the XCD it runs on ------------------------------------------------------------
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;       // HW_REG_XCC_ID
    unsigned* cnt = reinterpret_cast<unsigned*>(a.xch);
    misc[0] = (int)xcc;
    misc[1] = (int)atomicAdd(cnt + xcc, 1u);
    misc[2] = 0;
  }
  __syncthreads();
  // (block-uniform values out of LDS: readfirstlane keeps them -- and the buffer descriptor built from them -- in SGPRs)
  const int team = __builtin_amdgcn_readfirstlane(misc[0]), m = __builtin_amdgcn_readfirstlane(misc[1]);
  if (m >= TEAM_CUS) {        // more than 32 blocks on this XCD: another XCD is short of one, its team will time out
    if (tid == 0) __hip_atomic_store(a.err, 0x100u | (unsigned)team, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  if (a.S == 0) return;       // residency / placement probe
  const int g = m * 4 + wave;                 // wave of the team
  const int row0 = team * ROWS;               // the team's sequences
  unsigned char* team_xch = a.xch + XCH_HDR + (size_t)team * TEAM_BYTES;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(team_xch, 0, TEAM_BYTES, 0x00020000);
  const bool writer = m == 0;                 // member 0 writes the per-sequence results (world states, state slabs)

  // ---- resident weights -------------------------------------------------------------------------------------------------
  float wa[NWA], wv[NREG - NWA];
  {
    const float* wp = a.Wreg + (size_t)g * NREG * 64 + lane;
#pragma unroll
    for (int r = 0; r < NWA; ++r) wa[r] = wp[(size_t)r * 64];
#pragma unroll
    for (int r = NWA; r < NREG; ++r) wv[r - NWA] = wp[(size_t)r * 64];
  }
  // bias of the column this lane publishes (publish(): two-group layers lane (h, j) -> column 4 h + j of the wave's eight)
  const int j4 = lane & 3, h4 = 4 * (lane >> 5);
  const float b0 = a.bias[0][8 * g + h4 + j4], b1 = a.bias[1][8 * g + h4 + j4], b2 = a.bias[2][4 * g + j4];
  const float b3 = g < L3_WAVES ? a.bias[3][4 * g + j4] : 0.f;
  float gam1[4], bet1[4], gam2[4], bet2[4], gam3[2], bet3[2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    gam1[q] = a.gamma[0][tid + 256 * q]; bet1[q] = a.beta[0][tid + 256 * q];
    gam2[q] = a.gamma[1][tid + 256 * q]; bet2[q] = a.beta[1][tid + 256 * q];
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) { gam3[q] = a.gamma[2][tid + 256 * q]; bet3[q] = a.beta[2][tid + 256 * q]; }

  // ---- initial state: x_0 (row-major copy for the glue, [channel][row] copy as the A operand), z_0, world transform ---------
  for (int e = tid; e < ROWS * P_XPAD; e += 256) {
    const int i = e / P_XPAD, c = e - i * P_XPAD, r = row0 + i;
    const float v = (c < P_DIN && r < a.B) ? a.past_in0[(size_t)r * P_DIN + c] : 0.f;
    sX[e] = v;
    xs0[c * 4 + i] = v;
    if (writer && c < P_DINP) a.xT[pq(c) + (size_t)r * 4] = v;
  }
  const int zi = tid / P_ZD, zc = tid - zi * P_ZD;          // thread <-> (row, latent channel) for tid < 192
  const bool zlive = tid < ROWS * P_ZD && row0 + zi < a.B;
  if (tid < ROWS * P_ZD) zs[zc * 4 + zi] = zlive ? a.z_seq[((size_t)(row0 + zi) * a.S) * P_ZD + zc] : 0.f;
  // per-sequence state of wave `wave`'s row (identical in every lane)
  const int myrow = row0 + wave;
  float G[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}, gt[3] = {0.f, 0.f, 0.f}, t2j[3] = {0.f, 0.f, 0.f};
  if (myrow < a.B) {
    t2j[0] = -a.past_in0[(size_t)myrow * P_DIN + 207];
    t2j[1] = -a.past_in0[(size_t)myrow * P_DIN + 208];
  }
  if (writer && lane == 0) {
    float* Gs = a.steps + a.off_G + (size_t)myrow * 12;
#pragma unroll
    for (int i = 0; i < 9; ++i) Gs[i] = G[i];
    Gs[9] = Gs[10] = Gs[11] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) a.t2j[(size_t)myrow * 3 + c] = t2j[c];
  }

  // The results of step t (world states of the team's sequences, the next state slab, the accumulated transforms) sit in every CU's
  // LDS after the glue (every CU computes them): CU m writes elements [m, m + 1) x COPY_PER_CU of the team's record, one coalesced
  // store per thread.  Element order: world [row][348] | state slab quads [quad][row][4] | transforms [row][12].
  auto copy_out = [&](int t) {
    if (tid < COPY_PER_CU) {
      const int e = m * COPY_PER_CU + tid;
      if (e < COPY_WORLD) {
        const int i = e / P_STATE, c = e - i * P_STATE;
        if (row0 + i < a.B) a.world[((size_t)(row0 + i) * a.S + t) * P_STATE + c] = sW[i * P_XPAD + c];
      } else if (e < COPY_WORLD + COPY_XT) {
        const int e2 = e - COPY_WORLD, q = e2 >> 4, i = (e2 >> 2) & 3, k = e2 & 3;
        a.xT[(size_t)(t + 1) * P_DINP * 32 + (size_t)q * 128 + (size_t)(row0 + i) * 4 + k] = sX[i * P_XPAD + 4 * q + k];
      } else if (e < COPY_TOTAL) {
        const int e3 = e - COPY_WORLD - COPY_XT;
        a.steps[(size_t)(t + 1) * a.per_step + a.off_G + (size_t)row0 * 12 + e3] = sG[e3];
      }
    }
  };
#ifdef HA_PERSIST_TIMING
  const bool pt_on = team == 0 && (m == 5 || m == 0) && tid == 0;
  const int pt_slot = m == 0 ? 1 : 0;
#endif
  bool fail = false;
  for (int t = 0; t < a.S; ++t) {
    const unsigned tag = 4u * (unsigned)t;
    float* sp = a.steps + (size_t)t * a.per_step;
    // latent of the next step: issued now, written to LDS in the glue phase
    float z_next = 0.f;
    if (zlive && t + 1 < a.S) z_next = a.z_seq[((size_t)(row0 + zi) * a.S + (t + 1)) * P_ZD + zc];
    __syncthreads();                                   // xs0 / zs of this step are complete
    if (t > 0) copy_out(t - 1);                        // (stores only: nothing waits for them)
    PT(0);
    // ---- layer 0: [x_t | z_t] (raw) -> 1024 ---------------------------------------------------------------------------
    {
      float acc[8];
      mma_layer<NC0, 2, R0>(xs0, zs, wa, wv, lane, acc);
      PT(1);
      publish<2, SC1>(acc, b0, 8 * g, team_xch, rs, ACT_OFF0, tag + 1, sp + a.off_dec[0], row0, lane);
      PT(2);
    }
    // ---- layer 1 ----------------------------------------------------------------------------------------------------------
    if (!gather_norm<4, 64, 3>(rs, ACT_OFF0, tag + 1, gam1, bet1, xs1, tid PT_PASS)) fail = true;
    PT(4);
    if (fail) misc[2] = 1;
    __syncthreads();
    if (misc[2]) break;
    PT(5);
    {
      float acc[8];
      mma_layer<NC1, 2, R1>(xs1, zs, wa, wv, lane, acc);
      PT(6);
      publish<2, SC1>(acc, b1, 8 * g, team_xch, rs, ACT_OFF1, tag + 2, sp + a.off_dec[1], row0, lane);
      PT(7);
    }
    // ---- layer 2 ----------------------------------------------------------------------------------------------------------
    if (!gather_norm<4, 64, 8>(rs, ACT_OFF1, tag + 2, gam2, bet2, xs2, tid PT_PASS)) fail = true;
    PT(9);
    if (fail) misc[2] = 1;
    __syncthreads();
    if (misc[2]) break;
    PT(10);
    {
      float acc[4];
      mma_layer<NC2, 1, R2>(xs2, zs, wa, wv, lane, acc);
      PT(11);
      publish<1, SC1>(acc, b2, 4 * g, team_xch, rs, ACT_OFF2, tag + 3, sp + a.off_dec[2], row0, lane);
      PT(12);
    }
    // ---- layer 3 (GroupNorm groups of 32) -------------------------------------------------------------------------------
    if (!gather_norm<2, 32, 13>(rs, ACT_OFF2, tag + 3, gam3, bet3, xs3, tid PT_PASS)) fail = true;
    PT(14);
    if (fail) misc[2] = 1;
    __syncthreads();
    if (misc[2]) break;
    PT(15);
    if (g < L3_WAVES) {       // (wave-uniform)
      float acc[4];
      mma_layer<NC3, 1, R3>(xs3, zs, wa, wv, lane, acc);
      PT(16);
      publish<1, SC1>(acc, b3, 4 * g, team_xch, rs, ACT_OFF3, tag + 4, sp + a.off_dec[3], row0, lane);
      PT(17);
    }
    // ---- glue: decoder output of the 4 rows -> every CU --------------------------------------------------------------------
    {
      float x[1][4];
      if (!sweep<1>(rs, ACT_OFF3, tag + 4, tid < P_RAW, tid, x)) fail = true;
      if (tid < P_RAW) {
#pragma unroll
        for (int i = 0; i < 4; ++i) sRAW[i * P_RAWPAD + tid] = x[0][i];
      }
    }
    PT(18);
    if (fail) misc[2] = 1;
    __syncthreads();                                   // also: every wave of this CU is past its layer-3 reads of zs / xs3
    if (misc[2]) break;
    PT(19);
    if (tid < ROWS * P_ZD) zs[zc * 4 + zi] = z_next;
    {
      // One wave per sequence, two lane groups running the same instruction stream: group 0 (lanes 0..31) produces the next input
      // state (frame change by the heading alignment W), group 1 (lanes 32..63) the world-frame output (frame change by the
      // accumulated G^T).  In a group: lanes 0..21 = joints, lanes 0..20 also body rotation b, lane 21 the root.  Results go to LDS
      // (next state: row-major for the next glue + [channel][row] as the next A operand; world state: staging); the global copies
      // are written afterwards by all CUs of the team, one coalesced slice each (copy_out).
      const int jj = lane & 31, grp = lane >> 5;
      const float* X = sX + wave * P_XPAD;
      const float* RW = sRAW + wave * P_RAWPAD;
      const bool jl = jj < 22, root = jj == 21;
      const int aoff = jj < 21 ? 12 + 3 * jj : 6, roff = jj < 21 ? 18 + 9 * jj : 6;
      float pj[3] = {0.f, 0.f, 0.f}, jv[3] = {0.f, 0.f, 0.f}, pR[9], ptrans[3] = {0.f, 0.f, 0.f}, ptvel[3] = {0.f, 0.f, 0.f}, prvel[3] = {0.f, 0.f, 0.f};
      float Wm[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) { pR[i] = 0.f; Wm[i] = 0.f; }
      if (jl) {
        float aa[3], dR[9], Rin[9];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          pj[c] = RW[75 + 3 * jj + c] + X[207 + 3 * jj + c];
          jv[c] = RW[141 + 3 * jj + c] + X[273 + 3 * jj + c];
          aa[c] = RW[aoff + c];
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) Rin[i] = X[roff + i];
        rodrigues_sc(aa, dR);
        mat3_mul(dR, Rin, pR);
      }
      if (root) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          ptrans[c] = RW[c] + X[c];
          ptvel[c] = RW[3 + c] + X[3 + c];
          prvel[c] = RW[9 + c] + X[15 + c];
        }
      }
      if (lane == 21) w2a_sc(pR, Wm);
      const float craw = (lane >= 32 && lane < 41) ? RW[207 + lane - 32] : 0.f;      // contact logits
      // the root's heading alignment and translation to every lane
      float W[9], ptr[3];
#pragma unroll
      for (int i = 0; i < 9; ++i) W[i] = as_f(__builtin_amdgcn_readlane(as_u(Wm[i]), 21));
#pragma unroll
      for (int c = 0; c < 3; ++c) ptr[c] = as_f(__builtin_amdgcn_readlane(as_u(ptrans[c]), 21));
      // group 0: y = W (p + wt + t2j) - t2j ; W v        group 1: y = G^T (p + t2j) - t2j - gt ; G^T v
      float M[9], add[3], sub[3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) M[3 * i + k] = grp ? G[3 * k + i] : W[3 * i + k];
      add[0] = grp ? 0.f : -ptr[0];
      add[1] = grp ? 0.f : -ptr[1];
      add[2] = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) sub[c] = grp ? gt[c] : 0.f;
      float* Wn = sW + wave * P_XPAD;
      // group 0 -> next state (row-major + [channel][row]), group 1 -> world staging (+ a sink): two unconditional stores per value
      float* dst1 = grp ? Wn : sX + wave * P_XPAD;
      float* dst2 = (grp ? sDummy : xs0) + wave;
      auto put = [&](int c, float v) {
        dst1[c] = v;
        dst2[c * 4] = v;
      };
      if (jl) {
        float q[3], o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) q[c] = pj[c] + add[c] + t2j[c];
        mat3_vec(M, q, o);
#pragma unroll
        for (int c = 0; c < 3; ++c) put(207 + 3 * jj + c, o[c] - t2j[c] - sub[c]);
        mat3_vec(M, jv, o);
#pragma unroll
        for (int c = 0; c < 3; ++c) put(273 + 3 * jj + c, o[c]);
        if (!root) {
#pragma unroll
          for (int i = 0; i < 9; ++i) put(18 + 9 * jj + i, pR[i]);
        } else {
          // root: trans' = M (ptrans + add) - sub ; tvel' = M ptvel ; R' = M pR ; rvel' = M prvel
          float Rm[9];
#pragma unroll
          for (int c = 0; c < 3; ++c) q[c] = ptrans[c] + add[c];
          mat3_vec(M, q, o);
#pragma unroll
          for (int c = 0; c < 3; ++c) put(c, o[c] - sub[c]);
          mat3_vec(M, ptvel, o);
#pragma unroll
          for (int c = 0; c < 3; ++c) put(3 + c, o[c]);
          mat3_mul(M, pR, Rm);
#pragma unroll
          for (int i = 0; i < 9; ++i) put(6 + i, Rm[i]);
          mat3_vec(M, prvel, o);
#pragma unroll
          for (int c = 0; c < 3; ++c) put(15 + c, o[c]);
        }
      }
      if (lane >= 32 && lane < 41) Wn[339 + lane - 32] = craw;
      // accumulate the world transform (every lane, identical): wtrans = G^T ptrans - gt ; G' = G W ; gt' = (-wtrans.x, -wtrans.y, 0)
      float wtr[3], GW[9];
      mat3_tvec(G, ptr, wtr);
      mat3_mul(G, W, GW);
#pragma unroll
      for (int i = 0; i < 9; ++i) G[i] = GW[i];
      gt[0] = -(wtr[0] - gt[0]);
      gt[1] = -(wtr[1] - gt[1]);
      gt[2] = 0.f;
      if (lane == 0) {
        pvf4* gs = reinterpret_cast<pvf4*>(sG + wave * 12);
        gs[0] = pvf4{GW[0], GW[1], GW[2], GW[3]};
        gs[1] = pvf4{GW[4], GW[5], GW[6], GW[7]};
        gs[2] = pvf4{GW[8], gt[0], gt[1], gt[2]};
      }
    }
    PT(20);
  }
  __syncthreads();
  if (!misc[2]) copy_out(a.S - 1);
  if (misc[2] && tid == 0) __hip_atomic_store(a.err, 0x200u | (unsigned)team, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

#endif  // !HA_SIMT_EMU

#ifdef HA_PERSIST_TIMING
}  // namespace ha
extern "C" int ha_debug_persist_timing(unsigned long long* out /* [2][8][24] */) {
  HA_CHECK_HIP(hipDeviceSynchronize());
  HA_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(ha::g_pts), sizeof(unsigned long long) * 2 * 8 * 24));
  return HA_OK;
}
namespace ha {
#endif

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
template <typename T>
static int p_upload(T** dst, const std::vector<T>& src) {
  *dst = nullptr;
  HA_CHECK_HIP(hipMalloc((void**)dst, src.size() * sizeof(T)));
  HA_CHECK_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return HA_OK;
}

void persist_destroy(PersistNet* p) {
  if (!p) return;
#ifndef HA_SIMT_EMU
  DeviceGuard guard(p->device);
  if (p->Wreg) (void)hipFree(p->Wreg);
  for (float* q : p->bias)
    if (q) (void)hipFree(q);
  for (float* q : p->gamma)
    if (q) (void)hipFree(q);
  for (float* q : p->beta)
    if (q) (void)hipFree(q);
  if (p->err_host) (void)hipHostFree(p->err_host);
#endif
  delete p;
}

bool persist_usable(PersistNet* p) {
  if (!p || p->disabled) return false;
  if (p->err_host && *reinterpret_cast<volatile unsigned*>(p->err_host) != 0) p->disabled = true;   // sticky: reported by an earlier launch
  return !p->disabled;
}

long long persist_launches(PersistNet* p) { return p ? p->launches : 0; }

unsigned persist_error_word(PersistNet* p) { return (p && p->err_host) ? *reinterpret_cast<volatile unsigned*>(p->err_host) : 0u; }

int persist_create(PersistNet** out, int device, const ha_mlp_desc* d) {
  *out = nullptr;
#ifdef HA_SIMT_EMU
  (void)device; (void)d;
  return HA_OK;        // the host emulator runs blocks one after another: no persistent teams there
#else
  // the shape this path is built for: the default HuMoR decoder [339 + 48] -> 1024 -> 1024 -> 512 -> 216 with the latent skip
  if (!(d->n_linear == 4 && d->in_dim == P_DIN + P_ZD && d->skip_dim == P_ZD && d->out_dims[0] == P_H0 && d->out_dims[1] == P_H1 &&
        d->out_dims[2] == P_H2 && d->out_dims[3] == P_RAW))
    return HA_OK;
  hipDeviceProp_t prop;
  HA_CHECK_HIP(hipGetDeviceProperties(&prop, device));
  if (prop.multiProcessorCount != NTEAMS * TEAM_CUS) return HA_OK;
  PersistNet* p = new PersistNet();
  p->device = device;
  const int Kin[4] = {P_DIN + P_ZD, P_H0 + P_ZD, P_H1 + P_ZD, P_H2 + P_ZD}, Cmain[4] = {P_DIN, P_H0, P_H1, P_H2};
  const int NCm[4] = {NC0, NC1, NC2, NC3}, NCGv[4] = {2, 2, 1, 1}, RO[4] = {R0, R1, R2, R3}, Nout[4] = {P_H0, P_H1, P_H2, P_RAW};
  std::vector<float> wr((size_t)NWAVES_TEAM * NREG * 64, 0.f);
  for (int g = 0; g < NWAVES_TEAM; ++g)
    for (int l = 0; l < 4; ++l)
      for (int c = 0; c < NCm[l] + NCZ; ++c)
        for (int cg = 0; cg < NCGv[l]; ++cg)
          for (int ln = 0; ln < 64; ++ln) {
            const int b = ln >> 2, j = ln & 3;
            const int col = (NCGv[l] == 2 ? 8 * g : 4 * g) + 4 * cg + j;
            int k;
            if (c < NCm[l]) { k = 16 * c + b; if (k >= Cmain[l]) k = -1; }
            else k = Cmain[l] + 16 * (c - NCm[l]) + b;
            float v = 0.f;
            if (col < Nout[l] && k >= 0) v = d->w[l][(size_t)col * Kin[l] + k];
            wr[((size_t)g * NREG + RO[l] + c * NCGv[l] + cg) * 64 + ln] = v;
          }
  int rc = p_upload(&p->Wreg, wr);
  const int bpad[4] = {P_H0, P_H1, P_H2, P_RAWPAD};
  for (int l = 0; l < 4 && rc == HA_OK; ++l) {
    std::vector<float> bv(bpad[l], 0.f);
    for (int i = 0; i < Nout[l]; ++i) bv[i] = d->b[l][i];
    rc = p_upload(&p->bias[l], bv);
  }
  for (int l = 1; l < 4 && rc == HA_OK; ++l) {
    if (!d->gn_gamma[l] || !d->gn_beta[l]) { rc = HA_ERR_INVALID_ARG; set_error("persist_create: GroupNorm affine of layer %d missing", l); break; }
    std::vector<float> gv(d->gn_gamma[l], d->gn_gamma[l] + Cmain[l]), bv(d->gn_beta[l], d->gn_beta[l] + Cmain[l]);
    rc = p_upload(&p->gamma[l - 1], gv);
    if (rc == HA_OK) rc = p_upload(&p->beta[l - 1], bv);
  }
  if (rc == HA_OK) {
    hipError_t e = hipHostMalloc((void**)&p->err_host, 64, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&p->err_dev, p->err_host, 0);
    if (e != hipSuccess) { set_error("persist_create: host-mapped error word: %s", hipGetErrorString(e)); rc = HA_ERR_HIP; }
    else *p->err_host = 0;
  }
  if (rc == HA_OK) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rollout_persist_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, L_TOTAL * 4);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(rollout_persist_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, L_TOTAL * 4);
    if (e != hipSuccess) { set_error("persist_create: LDS attribute: %s", hipGetErrorString(e)); rc = HA_ERR_HIP; }
  }
  if (rc == HA_OK) {
    // placement / residency probe: one launch of the real kernel with zero steps; every XCD must have received exactly 32 blocks
    unsigned char* xch = nullptr;
    hipError_t e = hipMalloc((void**)&xch, XCH_HDR);
    if (e == hipSuccess) e = hipMemset(xch, 0, XCH_HDR);
    if (e == hipSuccess) {
      PersistArgs a;
      memset(&a, 0, sizeof(a));
      a.xch = xch;
      a.err = p->err_dev;
      hipLaunchKernelGGL(rollout_persist_fwd_kernel<false>, dim3(NTEAMS * TEAM_CUS), dim3(256), L_TOTAL * 4, 0, a);
      e = hipDeviceSynchronize();
    }
    unsigned cnt[NTEAMS] = {0};
    if (e == hipSuccess) e = hipMemcpy(cnt, xch, sizeof(cnt), hipMemcpyDeviceToHost);
    if (xch) (void)hipFree(xch);
    bool even = e == hipSuccess && *p->err_host == 0;
    for (int i = 0; i < NTEAMS; ++i) even = even && cnt[i] == (unsigned)TEAM_CUS;
    if (!even) { persist_destroy(p); return HA_OK; }      // not an error: the launch chain serves this device
  }
  if (rc != HA_OK) { persist_destroy(p); return rc; }
  *out = p;
  return HA_OK;
#endif
}

int persist_forward(PersistNet* p, const PersistFwd& f, int variant, hipStream_t st) {
#ifdef HA_SIMT_EMU
  (void)p; (void)f; (void)variant; (void)st;
  set_error("persistent roll-out: not available on the host emulator");
  return HA_ERR_INVALID_ARG;
#else
  HA_REQUIRE(p && f.B >= 1 && f.B <= NTEAMS * ROWS && f.S >= 1, "persistent roll-out: needs 1 <= B <= 32 sequences");
  HA_REQUIRE(f.S < (1 << 28), "persistent roll-out: too many steps");
  HA_CHECK_HIP(hipMemsetAsync(f.ws, 0, XCH_BYTES, st));        // tags, team counters (tag 0 never matches)
  PersistArgs a;
  memset(&a, 0, sizeof(a));
  a.B = f.B; a.S = f.S;
  a.Wreg = p->Wreg;
  for (int l = 0; l < 4; ++l) a.bias[l] = p->bias[l];
  for (int l = 0; l < 3; ++l) { a.gamma[l] = p->gamma[l]; a.beta[l] = p->beta[l]; }
  a.past_in0 = f.past_in0; a.z_seq = f.z_seq; a.world = f.world; a.xT = f.xT; a.steps = f.steps;
  a.per_step = f.per_step; a.off_G = f.off_G;
  for (int l = 0; l < 4; ++l) a.off_dec[l] = f.off_dec[l];
  a.t2j = f.t2j;
  a.xch = reinterpret_cast<unsigned char*>(f.ws);
  a.err = p->err_dev;
  if (variant & 1) hipLaunchKernelGGL(rollout_persist_fwd_kernel<true>, dim3(NTEAMS * TEAM_CUS), dim3(256), L_TOTAL * 4, st, a);
  else hipLaunchKernelGGL(rollout_persist_fwd_kernel<false>, dim3(NTEAMS * TEAM_CUS), dim3(256), L_TOTAL * 4, st, a);
  HA_LAUNCH_CHECK();
  ++p->launches;
  return HA_OK;
#endif
}

}  // namespace ha

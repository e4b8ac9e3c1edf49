// Cross-lane sums for the persistent roll-out kernel (gfx950): N independent values per lane, reduced over the wave (or each
// half-wave) with a reduce-scatter / all-gather over the rows instead of N separate butterflies.
//
// Building blocks:
//   * v_permlane32_swap a, b : a' = {a.lo, b.lo}, b' = {a.hi, b.hi}   (halves of 32 lanes)
//     v_permlane16_swap a, b : a' = {a.r0, b.r0, a.r2, b.r2}, b' = {a.r1, b.r1, a.r3, b.r3}   (rows of 16 lanes)
//     so a' + b' holds "a summed over the pair of halves (rows)" in the lower half (even rows) and the same for b in the upper half
//     (odd rows): ONE swap + ONE add reduces TWO values and halves the number of live values;
//   * v_add_f32 with a DPP row_ror operand (one instruction per value and stage) inside the 16-lane rows.
// tools/microbench/persist_probe.hip runs exactly these functions on the GPU against a host sum.
#pragma once
#include <hip/hip_runtime.h>

namespace ha {
namespace lr {

#ifdef HA_SIMT_EMU
// Host SIMT emulator (CPU test tier): the PRIMITIVES of this file as shuffles with the semantics stated above (the gfx950 forms below are
// builtins / inline asm); the composite sums further down are shared source.  tests/test_rollout_emu.py runs them through the ha_emu_* hooks of
// rollout_persist.hip; tools/microbench/persist_probe.hip runs the real instructions on the GPU against the same host sums.
__device__ __forceinline__ int emu_lane() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ void swap32(float& a, float& b) {      // a' = {a.lo, b.lo}, b' = {a.hi, b.hi}
  const int l = emu_lane();
  const float pa = __shfl(a, l ^ 32), pb = __shfl(b, l ^ 32);
  if (l < 32) b = pa; else a = pb;
}
__device__ __forceinline__ void swap16(float& a, float& b) {      // a' = {a.r0, b.r0, a.r2, b.r2}, b' = {a.r1, b.r1, a.r3, b.r3}
  const int l = emu_lane();
  const float pa = __shfl(a, l ^ 16), pb = __shfl(b, l ^ 16);
  if (((l >> 4) & 1) == 0) b = pa; else a = pb;
}
__device__ __forceinline__ float emu_add_ror(float v, int ror) {
  const int l = emu_lane();
  return v + __shfl(v, (l & ~15) | ((l + ror) & 15));
}
__device__ __forceinline__ float add_ror8(float v) { return emu_add_ror(v, 8); }
__device__ __forceinline__ float add_ror4(float v) { return emu_add_ror(v, 4); }
__device__ __forceinline__ float add_ror2(float v) { return emu_add_ror(v, 2); }
__device__ __forceinline__ float add_ror1(float v) { return emu_add_ror(v, 1); }
template <int N, int R0, int R1>
__device__ __forceinline__ void emu_stages(float (&t)[N]) {       // (stage by stage over the N values, as the asm blocks order them)
  for (int ror = R0; ror >= R1; ror >>= 1)
    for (int i = 0; i < N; ++i) t[i] = emu_add_ror(t[i], ror);
}
__device__ __forceinline__ void row_sum4(float (&t)[4]) { emu_stages<4, 8, 1>(t); }
__device__ __forceinline__ void row_sum8(float (&t)[8]) { emu_stages<8, 8, 1>(t); }
__device__ __forceinline__ void parity_sum2(float (&t)[2]) { emu_stages<2, 8, 2>(t); }
__device__ __forceinline__ void parity_sum4(float (&t)[4]) { emu_stages<4, 8, 2>(t); }
__device__ __forceinline__ void kblock_sum4(float (&t)[4]) {
  for (int ror = 4; ror <= 8; ror <<= 1)
    for (int i = 0; i < 4; ++i) t[i] = emu_add_ror(t[i], ror);
}
__device__ __forceinline__ void kblock_sum2(float (&t)[2]) {
  for (int ror = 4; ror <= 8; ror <<= 1)
    for (int i = 0; i < 2; ++i) t[i] = emu_add_ror(t[i], ror);
}
#else

typedef unsigned u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void swap32(float& a, float& b) {
  const u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r.x);
  b = __uint_as_float(r.y);
}
__device__ __forceinline__ void swap16(float& a, float& b) {
  const u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r.x);
  b = __uint_as_float(r.y);
}
// v + (v rotated by ROR inside its 16-lane row) as one VALU instruction, in place.  (Inline asm: the compiler expands the builtin
// form into v_mov + v_mov_dpp + v_add.)  A DPP source must not have been written by the VALU in the previous two wait states and
// the compiler does not see inside the asm, so every instance carries its own `s_nop 1`.
#define HA_LR_DPP_ADD(NAME, ROR)                                                                                   \
  __device__ __forceinline__ float NAME(float v) {                                                               \
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:" #ROR " row_mask:0xf bank_mask:0xf" : "+v"(v)); \
    return v;                                                                                                    \
  }
HA_LR_DPP_ADD(add_ror8, 8)
HA_LR_DPP_ADD(add_ror4, 4)
HA_LR_DPP_ADD(add_ror2, 2)
HA_LR_DPP_ADD(add_ror1, 1)
#undef HA_LR_DPP_ADD

// every lane of a 16-lane row gets the row's sum, N (>= 4) values side by side
template <int N>
__device__ __forceinline__ void row_sum(float (&t)[N]);

// Four (eight) independent values, each summed over its 16-lane row, every lane of the row ending up with the (bitwise identical)
// total: ONE asm block, the adds of the values interleaved so that a result is read by DPP only four (eight) instructions after it
// was written -- one `s_nop 1` for the whole block instead of one per add.  This is the GroupNorm reduction of the persistent
// kernels (rollout_persist.hip: a 16-lane row holds exactly one normalisation group).
__device__ __forceinline__ void row_sum4(float (&t)[4]) {
#define HA_LR_STAGE4(ROR)                                                             \
  "v_add_f32_dpp %0, %0, %0 row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t"         \
  "v_add_f32_dpp %1, %1, %1 row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t"         \
  "v_add_f32_dpp %2, %2, %2 row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t"         \
  "v_add_f32_dpp %3, %3, %3 row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t"
  asm volatile("s_nop 1\n\t" HA_LR_STAGE4(8) HA_LR_STAGE4(4) HA_LR_STAGE4(2) HA_LR_STAGE4(1)
               : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
#undef HA_LR_STAGE4
}
__device__ __forceinline__ void row_sum8(float (&t)[8]) {
#define HA_LR_STAGE8(ROR)                                                             \
  "v_add_f32_dpp %0, %0, %0 row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t"         \
  "v_add_f32_dpp %1, %1, %1 row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t"         \
  "v_add_f32_dpp %2, %2, %2 row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t"         \
  "v_add_f32_dpp %3, %3, %3 row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t"         \
  "v_add_f32_dpp %4, %4, %4 row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t"         \
  "v_add_f32_dpp %5, %5, %5 row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t"         \
  "v_add_f32_dpp %6, %6, %6 row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t"         \
  "v_add_f32_dpp %7, %7, %7 row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t"
  asm volatile("s_nop 1\n\t" HA_LR_STAGE8(8) HA_LR_STAGE8(4) HA_LR_STAGE8(2) HA_LR_STAGE8(1)
               : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
#undef HA_LR_STAGE8
}

// Two (four) values, each summed over the EIGHT lanes of equal parity of its 16-lane row (rotations by 8, 4, 2 keep the parity): the
// GroupNorm reduction of the half-granule sweeps (rollout_persist.hip: lane parity = row pair of the exchange granule).  Two values:
// a result is read by DPP two instructions after it was written, hence the s_nop 0 between the stages.
__device__ __forceinline__ void parity_sum2(float (&t)[2]) {
#define HA_LR_PS2(ROR)                                                                \
  "v_add_f32_dpp %0, %0, %0 row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t"         \
  "v_add_f32_dpp %1, %1, %1 row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t"
  asm volatile("s_nop 1\n\t" HA_LR_PS2(8) "s_nop 0\n\t" HA_LR_PS2(4) "s_nop 0\n\t" HA_LR_PS2(2) : "+v"(t[0]), "+v"(t[1]));
#undef HA_LR_PS2
}
__device__ __forceinline__ void parity_sum4(float (&t)[4]) {
#define HA_LR_PS4(ROR)                                                                \
  "v_add_f32_dpp %0, %0, %0 row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t"         \
  "v_add_f32_dpp %1, %1, %1 row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t"         \
  "v_add_f32_dpp %2, %2, %2 row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t"         \
  "v_add_f32_dpp %3, %3, %3 row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t"
  asm volatile("s_nop 1\n\t" HA_LR_PS4(8) HA_LR_PS4(4) HA_LR_PS4(2) : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
#undef HA_LR_PS4
}

#endif  // primitives, part 1 (HA_SIMT_EMU)

template <int N>
__device__ __forceinline__ void row_sum(float (&t)[N]) {
  if constexpr (N == 4) row_sum4(t);
  else if constexpr (N == 8) row_sum8(t);
  else {
#pragma unroll
    for (int n = 0; n < N; ++n) t[n] = add_ror8(t[n]);
#pragma unroll
    for (int n = 0; n < N; ++n) t[n] = add_ror4(t[n]);
#pragma unroll
    for (int n = 0; n < N; ++n) t[n] = add_ror2(t[n]);
#pragma unroll
    for (int n = 0; n < N; ++n) t[n] = add_ror1(t[n]);
  }
}

// 16 values, each summed over all 64 lanes; every lane ends up with all 16 sums (in place)
__device__ __forceinline__ void wave_sum16(float (&v)[16]) {
  float u[8], t[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) {           // lower half: v[i] over {l, l^32}; upper half: v[i+8]
    float a = v[i], b = v[i + 8];
    swap32(a, b);
    u[i] = a + b;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {           // even rows: u[i] over {l, l^16}; odd rows: u[i+4]
    float a = u[i], b = u[i + 4];
    swap16(a, b);
    t[i] = a + b;
  }
  row_sum(t);                               // lane (half h, row parity p): t[i] = total of v[8 h + 4 p + i]
  float g[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {           // even-row values to g[i], odd-row values to g[4 + i], in every row of the half
    float a = t[i], b = t[i];
    swap16(a, b);
    g[i] = a;
    g[4 + i] = b;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {           // lower-half values to v[j], upper-half values to v[8 + j], in both halves
    float a = g[j], b = g[j];
    swap32(a, b);
    v[j] = a;
    v[8 + j] = b;
  }
}

// 8 values, each summed over the 32 lanes of its half-wave; every lane ends up with the 8 sums of its half (in place)
__device__ __forceinline__ void half_sum8(float (&v)[8]) {
  float t[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a = v[i], b = v[i + 4];
    swap16(a, b);
    t[i] = a + b;
  }
  row_sum(t);                               // lane (row parity p): t[i] = half-wave total of v[4 p + i]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a = t[i], b = t[i];
    swap16(a, b);
    v[i] = a;
    v[4 + i] = b;
  }
}

// The last two stages of block_sum8 / block_sum4 (sum over the four k-blocks b' of a 16-lane row: row_ror 4, then row_ror 8) for FOUR (TWO)
// independent values in one asm block: a value's second add reads its first add's result three (two, with the s_nop) instructions
// later, so the block needs one leading s_nop instead of one per add -- and the four dependent chains overlap instead of running one
// after the other (a wave that owns a SIMD alone has nobody else to hide the DPP latency behind).
#ifndef HA_SIMT_EMU
__device__ __forceinline__ void kblock_sum4(float (&t)[4]) {
#define HA_LR_KB4(ROR)                                                                \
  "v_add_f32_dpp %0, %0, %0 row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t"         \
  "v_add_f32_dpp %1, %1, %1 row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t"         \
  "v_add_f32_dpp %2, %2, %2 row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t"         \
  "v_add_f32_dpp %3, %3, %3 row_ror:" #ROR " row_mask:0xf bank_mask:0xf\n\t"
  asm volatile("s_nop 1\n\t" HA_LR_KB4(4) HA_LR_KB4(8) : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
#undef HA_LR_KB4
}
__device__ __forceinline__ void kblock_sum2(float (&t)[2]) {
  asm volatile("s_nop 1\n\t"
               "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 0\n\t"
               "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
               : "+v"(t[0]), "+v"(t[1]));
}
#endif  // !HA_SIMT_EMU
// block_sum8 without its last two stages: afterwards out[0], out[1] still have to be summed over the row's four k-blocks (kblock_sum*)
__device__ __forceinline__ void block_sum8_head(const float (&v)[8], float (&out)[2]) {
  float u[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a = v[i], b = v[i + 4];
    swap32(a, b);
    u[i] = a + b;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float a = u[i], b = u[i + 2];
    swap16(a, b);
    out[i] = a + b;
  }
}

// MFMA 4x4x1 accumulators: lane = 16 r + 4 b' + j holds the partial of k-block b = 4 r + b' for column j.  Sums over the 16
// k-blocks, reduce-scatter form: 8 values in (two column groups x 4 rows), afterwards lane (half h, row parity p, column j, any b')
// holds in out[0], out[1] the totals of values 4 h + 2 p and 4 h + 2 p + 1 (column group h, rows 2 p and 2 p + 1).
__device__ __forceinline__ void block_sum8(const float (&v)[8], float (&out)[2]) {
  float u[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a = v[i], b = v[i + 4];
    swap32(a, b);
    u[i] = a + b;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float a = u[i], b = u[i + 2];
    swap16(a, b);
    out[i] = add_ror8(add_ror4(a + b));
  }
}
// 4 values in (one column group x 4 rows): afterwards lane (half h, row parity p, column j) holds the total of value 2 h + p
__device__ __forceinline__ float block_sum4(const float (&v)[4]) {
  float u[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float a = v[i], b = v[i + 2];
    swap32(a, b);
    u[i] = a + b;
  }
  float a = u[0], b = u[1];
  swap16(a, b);
  return add_ror8(add_ror4(a + b));
}

}  // namespace lr
}  // namespace ha

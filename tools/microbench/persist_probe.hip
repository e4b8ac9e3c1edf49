// Hardware assumptions of the persistent roll-out (humor_amd/csrc/rollout_persist.hip), checked on the GPU box:
//   1. lane layout of v_mfma_f32_4x4x1_16b_f32: D_v[lane 4b+j] += A[lane 4b+v] * B[lane 4b+j]
//   2. v_permlane16_swap / v_permlane32_swap of two copies + add = all-reduce over lane^16 / lane^32
//   3. DPP row_ror adds = all-reduce inside a 16-lane row
//   4. XCC_ID census of a 256-block, one-block-per-CU launch: 32 blocks per XCD
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/microbench/persist_probe tools/microbench/persist_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../../humor_amd/csrc/lane_reduce.h"
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef unsigned uv2 __attribute__((ext_vector_type(2)));

__global__ void mfma_probe(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  vf4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
  for (int i = 0; i < 4; ++i) d[i * 64 + l] = acc[i];
}
__global__ void lane_probe(float* out) {
  const int l = threadIdx.x;
  float v = (float)(1 << (l & 15)) + 65536.f * (float)(l >> 4);      // distinct contributions
  const uv2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  out[l] = __uint_as_float(r.x) + __uint_as_float(r.y);
  const uv2 q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  out[64 + l] = __uint_as_float(q.x) + __uint_as_float(q.y);
  float w = (float)(1 << (l & 15));
  w += __builtin_amdgcn_update_dpp(w, w, 0x128, 0xf, 0xf, false);
  w += __builtin_amdgcn_update_dpp(w, w, 0x124, 0xf, 0xf, false);
  w += __builtin_amdgcn_update_dpp(w, w, 0x122, 0xf, 0xf, false);
  w += __builtin_amdgcn_update_dpp(w, w, 0x121, 0xf, 0xf, false);
  out[128 + l] = w;
  float u = (float)(1 << (l >> 2));        // lane (b, j): sum over b for fixed j via row_ror 4, 8 + the two swaps
  u += __builtin_amdgcn_update_dpp(u, u, 0x124, 0xf, 0xf, false);
  u += __builtin_amdgcn_update_dpp(u, u, 0x128, 0xf, 0xf, false);
  { const uv2 s = __builtin_amdgcn_permlane16_swap(__float_as_uint(u), __float_as_uint(u), false, false); u = __uint_as_float(s.x) + __uint_as_float(s.y); }
  { const uv2 s = __builtin_amdgcn_permlane32_swap(__float_as_uint(u), __float_as_uint(u), false, false); u = __uint_as_float(s.x) + __uint_as_float(s.y); }
  out[192 + l] = u;
}
// the reduce-scatter / all-gather sums of lane_reduce.h exactly as the kernel uses them; inputs are small integers (exact sums)
__global__ void lr_probe(const float* in /* [16][64] */, float* out) {
  const int l = threadIdx.x;
  float v[16];
  for (int n = 0; n < 16; ++n) v[n] = in[n * 64 + l];
  float a[16];
  for (int n = 0; n < 16; ++n) a[n] = v[n];
  ha::lr::wave_sum16(a);
  for (int n = 0; n < 16; ++n) out[n * 64 + l] = a[n];
  float b[8];
  for (int n = 0; n < 8; ++n) b[n] = v[n];
  ha::lr::half_sum8(b);
  for (int n = 0; n < 8; ++n) out[1024 + n * 64 + l] = b[n];
  float c[8], o2[2];
  for (int n = 0; n < 8; ++n) c[n] = v[n];
  ha::lr::block_sum8(c, o2);
  out[1536 + l] = o2[0];
  out[1600 + l] = o2[1];
  float d[4];
  for (int n = 0; n < 4; ++n) d[n] = v[n];
  out[1664 + l] = ha::lr::block_sum4(d);
}

__global__ void xcc_census(unsigned* xcc_of_block, unsigned long long* t) {
  extern __shared__ float smem[];
  if (threadIdx.x == 0) {
    xcc_of_block[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    t[blockIdx.x] = wall_clock64();
    smem[0] = 1.f;
  }
}

int main() {
  int fails = 0;
  // ---- 1. MFMA layout ----
  float *a, *b, *d;
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
  std::vector<float> ha(64), hb(64), hd(256);
  for (int l = 0; l < 64; ++l) ha[l] = (float)(l + 1);
  hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice);
  int bad = 0;
  for (int lb = 0; lb < 64; ++lb) {
    for (int l = 0; l < 64; ++l) hb[l] = l == lb ? 1.f : 0.f;
    hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
    mfma_probe<<<1, 64>>>(a, b, d);
    hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
    for (int v = 0; v < 4; ++v)
      for (int l = 0; l < 64; ++l) {
        const float expect = l == lb ? ha[(lb & ~3) + v] : 0.f;
        if (hd[v * 64 + l] != expect) {
          if (bad < 12) printf("mfma4x4x1: B one-hot lane %d: D reg %d lane %d = %g, expected %g\n", lb, v, l, hd[v * 64 + l], expect);
          ++bad;
        }
      }
  }
  printf("1. mfma_f32_4x4x1 layout D_v[4b+j] = A[4b+v] B[4b+j]: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
  fails += bad != 0;
  // ---- 2/3. lane reductions ----
  float* out;
  hipMalloc(&out, 1024);
  lane_probe<<<1, 64>>>(out);
  std::vector<float> ho(256);
  hipMemcpy(ho.data(), out, 1024, hipMemcpyDeviceToHost);
  auto val = [](int l) { return (float)(1 << (l & 15)) + 65536.f * (float)(l >> 4); };
  int b16 = 0, b32 = 0, brow = 0, bblk = 0;
  for (int l = 0; l < 64; ++l) {
    if (ho[l] != val(l) + val(l ^ 16)) { if (b16 < 4) printf("permlane16_swap: lane %d got %g expected %g\n", l, ho[l], val(l) + val(l ^ 16)); ++b16; }
    if (ho[64 + l] != val(l) + val(l ^ 32)) { if (b32 < 4) printf("permlane32_swap: lane %d got %g expected %g\n", l, ho[64 + l], val(l) + val(l ^ 32)); ++b32; }
    if (ho[128 + l] != 65535.f) { if (brow < 4) printf("row_ror sum: lane %d got %g expected 65535\n", l, ho[128 + l]); ++brow; }
    if (ho[192 + l] != 65535.f) { if (bblk < 4) printf("block sum: lane %d got %g expected 65535\n", l, ho[192 + l]); ++bblk; }
  }
  printf("2. permlane16_swap all-reduce: %s   permlane32_swap all-reduce: %s\n", b16 ? "FAIL" : "PASS", b32 ? "FAIL" : "PASS");
  printf("3. row_ror 8/4/2/1 row sum: %s   (b, j) block sum: %s\n", brow ? "FAIL" : "PASS", bblk ? "FAIL" : "PASS");
  fails += (b16 != 0) + (b32 != 0) + (brow != 0) + (bblk != 0);
  // ---- 3b. lane_reduce.h ----
  {
    std::vector<float> hin(16 * 64), hout(1728);
    unsigned rng = 12345u;
    for (auto& x : hin) { rng = rng * 1664525u + 1013904223u; x = (float)((int)((rng >> 16) % 41) - 20); }
    float *din, *dout;
    hipMalloc(&din, hin.size() * 4); hipMalloc(&dout, hout.size() * 4);
    hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice);
    lr_probe<<<1, 64>>>(din, dout);
    hipMemcpy(hout.data(), dout, hout.size() * 4, hipMemcpyDeviceToHost);
    int e16 = 0, e8 = 0, eb8 = 0, eb4 = 0;
    for (int n = 0; n < 16; ++n) {
      float tot = 0.f;
      for (int l = 0; l < 64; ++l) tot += hin[n * 64 + l];
      for (int l = 0; l < 64; ++l) if (hout[n * 64 + l] != tot) { if (e16 < 4) printf("wave_sum16: value %d lane %d got %g expected %g\n", n, l, hout[n * 64 + l], tot); ++e16; }
    }
    for (int n = 0; n < 8; ++n)
      for (int hh = 0; hh < 2; ++hh) {
        float tot = 0.f;
        for (int l = 0; l < 32; ++l) tot += hin[n * 64 + 32 * hh + l];
        for (int l = 0; l < 32; ++l) if (hout[1024 + n * 64 + 32 * hh + l] != tot) { if (e8 < 4) printf("half_sum8: value %d lane %d got %g expected %g\n", n, 32 * hh + l, hout[1024 + n * 64 + 32 * hh + l], tot); ++e8; }
      }
    for (int l = 0; l < 64; ++l) {
      const int hh = l >> 5, p = (l >> 4) & 1, j = l & 3;
      for (int e = 0; e < 2; ++e) {        // block_sum8: value 4 h + 2 p + e summed over the lanes with the same j
        const int n = 4 * hh + 2 * p + e;
        float tot = 0.f;
        for (int k = 0; k < 16; ++k) tot += hin[n * 64 + 4 * k + j];
        const float got = hout[1536 + 64 * e + l];
        if (got != tot) { if (eb8 < 4) printf("block_sum8: lane %d out[%d] got %g expected %g\n", l, e, got, tot); ++eb8; }
      }
      const int n = 2 * hh + p;
      float tot = 0.f;
      for (int k = 0; k < 16; ++k) tot += hin[n * 64 + 4 * k + j];
      if (hout[1664 + l] != tot) { if (eb4 < 4) printf("block_sum4: lane %d got %g expected %g\n", l, hout[1664 + l], tot); ++eb4; }
    }
    printf("3b. lane_reduce.h: wave_sum16 %s  half_sum8 %s  block_sum8 %s  block_sum4 %s\n", e16 ? "FAIL" : "PASS", e8 ? "FAIL" : "PASS", eb8 ? "FAIL" : "PASS", eb4 ? "FAIL" : "PASS");
    fails += (e16 != 0) + (e8 != 0) + (eb8 != 0) + (eb4 != 0);
  }
  // ---- 4. XCC census ----
  unsigned* xb;
  unsigned long long* tb;
  hipMalloc(&xb, 1024); hipMalloc(&tb, 2048);
  hipFuncSetAttribute((const void*)xcc_census, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  for (int rep = 0; rep < 3; ++rep) {
    xcc_census<<<256, 256, 100 * 1024>>>(xb, tb);
    std::vector<unsigned> hx(256);
    hipMemcpy(hx.data(), xb, 1024, hipMemcpyDeviceToHost);
    int cnt[8] = {0}, modok = 1;
    for (int i = 0; i < 256; ++i) { cnt[hx[i] & 7]++; modok &= (int)(hx[i] & 7) == (i & 7); }
    int even = 1;
    for (int i = 0; i < 8; ++i) even &= cnt[i] == 32;
    printf("4. XCC census run %d: [%d %d %d %d %d %d %d %d]  block b on XCD b%%8: %s  32 per XCD: %s\n", rep, cnt[0], cnt[1], cnt[2], cnt[3], cnt[4], cnt[5],
           cnt[6], cnt[7], modok ? "yes" : "no", even ? "PASS" : "FAIL");
    if (rep == 2) fails += !even;
  }
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  printf("device: %s, %d CUs, clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
  printf(fails ? "PROBE: %d FAILED\n" : "PROBE: all assumptions hold\n", fails);
  return fails;
}

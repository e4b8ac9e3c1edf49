// The SMPL body of a frame under a second root pose, as the rigid image of the first one.
//
// MotionOptimizer evaluates the body model twice per stage-3 closure on the SAME body pose and shape: once with the prior-frame
// root trajectory and once with the camera-frame one (humor/fitting/motion_optimizer.py:573 and :584).  With root rotation R, root
// translation t and rest root joint J0, every SMPL output point is  X = R (x - J0) + J0 + t  for a root-independent x
// (body_model.py:146-153 -> lbs: the root is the first link of the kinematic chain and `trans` is added last), so under a second
// root (R', t')
//     X' = Q (X - p) + p - t + t',     Q = R' R^T,     p = J0 + t = the root joint of the first evaluation.
// One block per frame maps the joints and vertices of the first evaluation; the adjoint returns the gradients of the points and of
// both root poses (axis-angle, through rodrigues).  Replaces the second SMPL forward + backward of the closure.
#include "common.h"

namespace ha {

namespace {

constexpr int RG_NRED = 15;   // per-frame sums of the adjoint: g d^T (9) | sum g (3) | sum Q^T g (3)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

struct RigidFrame {
  float R[9], R2[9], Q[9], p[3], sh[3];   // sh = p - t + t'
};

__device__ __forceinline__ void rigid_frame(const ha_rigid_image_args& a, int f, RigidFrame& r) {
  const float ro[3] = {a.root[(size_t)f * 3], a.root[(size_t)f * 3 + 1], a.root[(size_t)f * 3 + 2]};
  const float ro2[3] = {a.root2[(size_t)f * 3], a.root2[(size_t)f * 3 + 1], a.root2[(size_t)f * 3 + 2]};
  rodrigues(ro, r.R);
  rodrigues(ro2, r.R2);
  mat3_mult(r.R2, r.R, r.Q);                // R' R^T
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    r.p[c] = a.joints[(size_t)f * a.J * 3 + c];
    r.sh[c] = r.p[c] - a.trans[(size_t)f * 3 + c] + a.trans2[(size_t)f * 3 + c];
  }
}

}  // namespace

__global__ void rigid_image_fwd_kernel(ha_rigid_image_args a) {
  const int f = blockIdx.x;
  RigidFrame r;
  rigid_frame(a, f, r);
  const int P = a.J + a.V;
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    const bool jt = i < a.J;
    const size_t o = jt ? ((size_t)f * a.J + i) * 3 : ((size_t)f * a.V + (i - a.J)) * 3;
    const float* src = (jt ? a.joints : a.verts) + o;
    float* dst = (jt ? a.joints2 : a.verts2) + o;
    const float d[3] = {src[0] - r.p[0], src[1] - r.p[1], src[2] - r.p[2]};
    float q[3];
    mat3_vec(r.Q, d, q);
    dst[0] = q[0] + r.sh[0]; dst[1] = q[1] + r.sh[1]; dst[2] = q[2] + r.sh[2];
  }
}

__global__ void rigid_image_bwd_kernel(ha_rigid_image_args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];      // [4][RG_NRED] (dynamic LDS only: the host emulator tier has no static LDS)
  float (*red)[RG_NRED] = reinterpret_cast<float (*)[RG_NRED]>(smem);
  const int f = blockIdx.x;
  RigidFrame r;
  rigid_frame(a, f, r);
  const int P = a.J + a.V;
  float acc[RG_NRED];
#pragma unroll
  for (int k = 0; k < RG_NRED; ++k) acc[k] = 0.f;
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    const bool jt = i < a.J;
    const size_t o = jt ? ((size_t)f * a.J + i) * 3 : ((size_t)f * a.V + (i - a.J)) * 3;
    const float* gsrc = jt ? a.g_joints2 : a.g_verts2;
    float g[3] = {0.f, 0.f, 0.f};
    if (gsrc) { g[0] = gsrc[o]; g[1] = gsrc[o + 1]; g[2] = gsrc[o + 2]; }
    const float* src = (jt ? a.joints : a.verts) + o;
    const float d[3] = {src[0] - r.p[0], src[1] - r.p[1], src[2] - r.p[2]};
    float q[3];
    mat3_tvec(r.Q, g, q);                   // dL/dX = Q^T g
    float* gdst = (jt ? a.g_joints : a.g_verts) + o;
    const float* gadd = jt ? a.g_joints_add : a.g_verts_add;      // the gradient another reader of the same tensor produced (or null)
    if (gadd) { gdst[0] = q[0] + gadd[o]; gdst[1] = q[1] + gadd[o + 1]; gdst[2] = q[2] + gadd[o + 2]; }
    else { gdst[0] = q[0]; gdst[1] = q[1]; gdst[2] = q[2]; }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      acc[3 * c + 0] = fmaf(g[c], d[0], acc[3 * c + 0]);
      acc[3 * c + 1] = fmaf(g[c], d[1], acc[3 * c + 1]);
      acc[3 * c + 2] = fmaf(g[c], d[2], acc[3 * c + 2]);
      acc[9 + c] += g[c];
      acc[12 + c] += q[c];
    }
  }
  // fixed-order sums: butterfly inside the wavefront, then the wavefronts in order
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
#pragma unroll
  for (int k = 0; k < RG_NRED; ++k) acc[k] = wave_sum(acc[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < RG_NRED; ++k) red[wave][k] = acc[k];
  }
  __syncthreads();      // also orders thread 0's own dL/dX store of the root joint before its update below
  if (threadIdx.x != 0) return;
  float s[RG_NRED];
#pragma unroll
  for (int k = 0; k < RG_NRED; ++k) {
    s[k] = red[0][k];
    for (int w = 1; w < nw; ++w) s[k] += red[w][k];
  }
  // p is the root joint of the first evaluation: dL/dp = sum (g - Q^T g)
  float* g0 = a.g_joints + (size_t)f * a.J * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    g0[c] += s[9 + c] - s[12 + c];
    a.g_trans[(size_t)f * 3 + c] = -s[9 + c];
    a.g_trans2[(size_t)f * 3 + c] = s[9 + c];
  }
  // Q = R' R^T: dL/dR' = gQ R, dL/dR = gQ^T R'
  float gR2[9], gR[9], gaa[3];
  mat3_mul(s, r.R, gR2);
  mat3_tmul(s, r.R2, gR);
  const float ro[3] = {a.root[(size_t)f * 3], a.root[(size_t)f * 3 + 1], a.root[(size_t)f * 3 + 2]};
  const float ro2[3] = {a.root2[(size_t)f * 3], a.root2[(size_t)f * 3 + 1], a.root2[(size_t)f * 3 + 2]};
  rodrigues_bwd(ro, gR, gaa);
  a.g_root[(size_t)f * 3] = gaa[0]; a.g_root[(size_t)f * 3 + 1] = gaa[1]; a.g_root[(size_t)f * 3 + 2] = gaa[2];
  rodrigues_bwd(ro2, gR2, gaa);
  a.g_root2[(size_t)f * 3] = gaa[0]; a.g_root2[(size_t)f * 3 + 1] = gaa[1]; a.g_root2[(size_t)f * 3 + 2] = gaa[2];
}

}  // namespace ha

using namespace ha;

static int rigid_threads(const ha_rigid_image_args& a) { return a.J + a.V <= 128 ? 64 : 256; }

extern "C" int ha_rigid_image_forward(const ha_rigid_image_args* args, void* stream) {
  HA_REQUIRE(args, "ha_rigid_image_forward: null argument");
  const ha_rigid_image_args& a = *args;
  HA_REQUIRE(a.N >= 1 && a.J >= 1 && a.V >= 0, "ha_rigid_image_forward: N, J must be >= 1 and V >= 0");
  HA_REQUIRE(a.joints && a.root && a.trans && a.root2 && a.trans2 && a.joints2, "ha_rigid_image_forward: null tensor");
  HA_REQUIRE(a.V == 0 || (a.verts && a.verts2), "ha_rigid_image_forward: vertex tensors missing");
  HA_LAUNCH(rigid_image_fwd_kernel, dim3(a.N), dim3(rigid_threads(a)), 0, (hipStream_t)stream, a);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

extern "C" int ha_rigid_image_backward(const ha_rigid_image_args* args, void* stream) {
  HA_REQUIRE(args, "ha_rigid_image_backward: null argument");
  const ha_rigid_image_args& a = *args;
  HA_REQUIRE(a.N >= 1 && a.J >= 1 && a.V >= 0, "ha_rigid_image_backward: N, J must be >= 1 and V >= 0");
  HA_REQUIRE(a.joints && a.root && a.trans && a.root2 && a.trans2, "ha_rigid_image_backward: forward tensors missing");
  HA_REQUIRE(a.g_joints && a.g_root && a.g_trans && a.g_root2 && a.g_trans2, "ha_rigid_image_backward: null gradient output");
  HA_REQUIRE(a.V == 0 || (a.verts && a.g_verts), "ha_rigid_image_backward: vertex tensors missing");
  HA_LAUNCH(rigid_image_bwd_kernel, dim3(a.N), dim3(rigid_threads(a)), 4 * RG_NRED * sizeof(float), (hipStream_t)stream, a);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

// What MotionOptimizer.rollout_latent_motion does with the roll-out's output, in one launch per direction
// (humor/fitting/motion_optimizer.py:950-1019): rotation matrices -> axis-angle for the root and the 21 body joints
// (rotation_matrix_to_angle_axis, transforms.py:243-389), frame 0 (the optimised initial state) prepended, contact logits ->
// sigmoid confidences / 0.5-thresholded labels scattered to the 22 SMPL joints through CONTACT_INDS (amass_utils.py:21-23) with
// frame 0 copied from frame 1, and -- when a floor is optimised -- the inverse cam2prior map of the root trajectory back into the
// camera frame (apply_cam2prior(..., inverse=True), motion_optimizer.py:678-742: R^T rodrigues(root) -> axis-angle,
// R^T (trans - trans_0) - t).  Replaces ~100 element-wise / gather / cat launches per closure (forward + autograd backward).
// One wavefront per frame (b, t); lane j < 22 owns SMPL joint j (lane 0 also the root trajectory and the camera-frame map).
// The adjoint's per-sequence sums (d/dR, d/dt of cam2prior, d/dtrans_0) are written as per-frame partials and added in frame
// order by a second small launch: no atomics.
#include <string.h>

#include "rot_math.h"

namespace ha {

__device__ const int CONTACT_INDS_DEV[9] = {0, 4, 5, 7, 8, 10, 11, 20, 21};
constexpr int PW = 348, NPART = 15;       // world state width; per-frame partial sums: g_R (9) | g_t (3) | g_trans0 (3)

__device__ __forceinline__ int contact_slot(int joint) {
#pragma unroll
  for (int k = 0; k < 9; ++k)
    if (CONTACT_INDS_DEV[k] == joint) return k;
  return -1;
}

__global__ __launch_bounds__(64) void rollout_post_fwd_kernel(ha_rollout_post_args a) {
  const int T = a.S + 1, f = blockIdx.x, b = f / T, t = f - b * T, j = threadIdx.x;
  const float* w = t > 0 ? a.world + ((size_t)b * a.S + (t - 1)) * PW : nullptr;
  float aa_root[3] = {0.f, 0.f, 0.f}, tr[3] = {0.f, 0.f, 0.f};
  if (j < 22) {
    // rotation of joint j (root or body j-1) as axis-angle
    float aa[3];
    if (t == 0) {
      const float* src = j == 0 ? a.root0 + (size_t)b * 3 : a.pose0 + (size_t)b * 63 + (j - 1) * 3;
      aa[0] = src[0]; aa[1] = src[1]; aa[2] = src[2];
    } else {
      float R[9];
      const float* src = j == 0 ? w + 6 : w + 18 + 9 * (j - 1);
#pragma unroll
      for (int i = 0; i < 9; ++i) R[i] = src[i];
      rotmat_to_aa(R, aa);
    }
    float* dst = j == 0 ? a.root_orient + (size_t)f * 3 : a.pose_body + (size_t)f * 63 + (j - 1) * 3;
    dst[0] = aa[0]; dst[1] = aa[1]; dst[2] = aa[2];
    if (j == 0) { aa_root[0] = aa[0]; aa_root[1] = aa[1]; aa_root[2] = aa[2]; }
    // joint position
    const float* js = t == 0 ? a.joints0 + ((size_t)b * 22 + j) * 3 : w + 207 + 3 * j;
    float* jd = a.joints + ((size_t)f * 22 + j) * 3;
    jd[0] = js[0]; jd[1] = js[1]; jd[2] = js[2];
    // contacts: frame 0 repeats frame 1 (motion_optimizer.py:990-996)
    const int k = contact_slot(j);
    float conf = 0.f, lab = 0.f;
    if (k >= 0) {
      const float* wc = a.world + ((size_t)b * a.S + (t > 0 ? t - 1 : 0)) * PW + 339;
      conf = 1.0f / (1.0f + expf(-wc[k]));
      lab = conf > 0.5f ? 1.f : 0.f;
    }
    a.contacts_conf[(size_t)f * 22 + j] = conf;
    a.contacts[(size_t)f * 22 + j] = lab;
  }
  if (j == 0) {
    const float* ts = t == 0 ? a.trans0 + (size_t)b * 3 : w;
    tr[0] = ts[0]; tr[1] = ts[1]; tr[2] = ts[2];
    float* td = a.trans + (size_t)f * 3;
    td[0] = tr[0]; td[1] = tr[1]; td[2] = tr[2];
    if (a.c2p_R) {
      const float* R = a.c2p_R + (size_t)b * 9;
      const float* t0 = a.trans0 + (size_t)b * 3;
      float Rm[9], N[9], aa[3];
      rodrigues(aa_root, Rm);
      mat3_tmul(R, Rm, N);                       // R^T Rm
      rotmat_to_aa(N, aa);
      float* cr = a.cam_root_orient + (size_t)f * 3;
      cr[0] = aa[0]; cr[1] = aa[1]; cr[2] = aa[2];
      const float d[3] = {tr[0] - t0[0], tr[1] - t0[1], tr[2] - t0[2]};
      float o[3];
      mat3_tvec(R, d, o);                        // R^T d
      float* ct = a.cam_trans + (size_t)f * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) ct[c] = o[c] - a.c2p_t[(size_t)b * 3 + c];
    }
  }
}

__global__ __launch_bounds__(64) void rollout_post_bwd_kernel(ha_rollout_post_args a) {
  const int T = a.S + 1, f = blockIdx.x, b = f / T, t = f - b * T, j = threadIdx.x;
  const float* w = t > 0 ? a.world + ((size_t)b * a.S + (t - 1)) * PW : nullptr;
  float* gw = t > 0 ? a.g_world + ((size_t)b * a.S + (t - 1)) * PW : nullptr;
  auto in3 = [&](const float* p, size_t off, float (&o)[3]) {
    if (p) { o[0] = p[off]; o[1] = p[off + 1]; o[2] = p[off + 2]; } else { o[0] = o[1] = o[2] = 0.f; }
  };
  float part[NPART];
#pragma unroll
  for (int i = 0; i < NPART; ++i) part[i] = 0.f;
  if (gw) {                                        // channels no output reads: translational / angular / joint velocities
    for (int c = 3 + j; c < 6; c += 64) gw[c] = 0.f;
    for (int c = 15 + j; c < 18; c += 64) gw[c] = 0.f;
    for (int c = 273 + j; c < 339; c += 64) gw[c] = 0.f;
  }
  if (j < 22) {
    // ---- joint rotation: adjoint of R -> axis-angle (root: plus the camera-frame branch) ---------------------------------------
    float g_aa[3];
    if (j == 0) in3(a.g_root_orient, (size_t)f * 3, g_aa);
    else in3(a.g_pose_body, (size_t)f * 63 + (j - 1) * 3, g_aa);
    if (j == 0) {
      float g_tr[3];
      in3(a.g_trans, (size_t)f * 3, g_tr);
      if (a.c2p_R) {
        const float* R = a.c2p_R + (size_t)b * 9;
        const float* t0 = a.trans0 + (size_t)b * 3;
        const float* aa_root = a.root_orient + (size_t)f * 3;       // forward output (frame 0: root0)
        float g_cr[3], g_ct[3];
        in3(a.g_cam_root_orient, (size_t)f * 3, g_cr);
        in3(a.g_cam_trans, (size_t)f * 3, g_ct);
        // cam_root = aa(N), N = R^T Rm, Rm = rodrigues(aa_root)
        float Rm[9], N[9], gN[9], gRm[9], gaa[3];
        const float ar[3] = {aa_root[0], aa_root[1], aa_root[2]};
        rodrigues(ar, Rm);
        mat3_tmul(R, Rm, N);
        rotmat_to_aa_bwd(N, g_cr, gN);
        mat3_mul(R, gN, gRm);                    // gRm = R gN
        // gR[j][i] += sum_k Rm[j][k] gN[i][k]  ->  gR += Rm gN^T
        float gR[9];
        mat3_mult(Rm, gN, gR);
        rodrigues_bwd(ar, gRm, gaa);
#pragma unroll
        for (int c = 0; c < 3; ++c) g_aa[c] += gaa[c];
        // cam_trans = R^T (x_t - x_0) - t
        const float* xt = t == 0 ? t0 : w;
        const float d[3] = {xt[0] - t0[0], xt[1] - t0[1], xt[2] - t0[2]};
        float Rg[3];
        mat3_vec(R, g_ct, Rg);                   // R g_ct
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          g_tr[c] += Rg[c];
          part[12 + c] -= Rg[c];
          part[9 + c] -= g_ct[c];
        }
        // gR[j][i] += d_j g_ct_i
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) gR[r * 3 + c] += d[r] * g_ct[c];
#pragma unroll
        for (int i = 0; i < 9; ++i) part[i] = gR[i];
      }
      if (t == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) part[12 + c] += g_tr[c];
      } else {
        gw[0] = g_tr[0]; gw[1] = g_tr[1]; gw[2] = g_tr[2];
      }
    }
    if (t == 0) {
      float* dst = j == 0 ? a.g_root0 + (size_t)b * 3 : a.g_pose0 + (size_t)b * 63 + (j - 1) * 3;
      dst[0] = g_aa[0]; dst[1] = g_aa[1]; dst[2] = g_aa[2];
    } else {
      float R[9], gR[9];
      const float* src = j == 0 ? w + 6 : w + 18 + 9 * (j - 1);
#pragma unroll
      for (int i = 0; i < 9; ++i) R[i] = src[i];
      rotmat_to_aa_bwd(R, g_aa, gR);
      float* dst = j == 0 ? gw + 6 : gw + 18 + 9 * (j - 1);
#pragma unroll
      for (int i = 0; i < 9; ++i) dst[i] = gR[i];
    }
    // ---- joint positions ------------------------------------------------------------------------------------------------------------
    float gj[3];
    in3(a.g_joints, ((size_t)f * 22 + j) * 3, gj);
    float* jd = t == 0 ? a.g_joints0 + ((size_t)b * 22 + j) * 3 : gw + 207 + 3 * j;
    jd[0] = gj[0]; jd[1] = gj[1]; jd[2] = gj[2];
    // ---- contacts: confidence of frame t comes from the logits of step max(t-1, 0); frame 1's logits also feed frame 0 -----------------
    const int k = contact_slot(j);
    if (k >= 0 && t > 0) {
      const float conf = a.contacts_conf[(size_t)f * 22 + j];
      float g = a.g_contacts_conf ? a.g_contacts_conf[(size_t)f * 22 + j] : 0.f;
      if (t == 1 && a.g_contacts_conf) g += a.g_contacts_conf[(size_t)(f - 1) * 22 + j];
      gw[339 + k] = g * conf * (1.0f - conf);
    }
  }
  if (j == 0) {
    float* P = a.partial + (size_t)f * NPART;
#pragma unroll
    for (int i = 0; i < NPART; ++i) P[i] = part[i];
  }
}

// per sequence: sums of the per-frame partials in frame order
__global__ __launch_bounds__(64) void rollout_post_reduce_kernel(ha_rollout_post_args a) {
  const int T = a.S + 1, b = blockIdx.x, i = threadIdx.x;
  if (i >= NPART) return;
  float s = 0.f;
  for (int t = 0; t < T; ++t) s += a.partial[((size_t)b * T + t) * NPART + i];
  if (i < 9) { if (a.g_c2p_R) a.g_c2p_R[(size_t)b * 9 + i] = s; }
  else if (i < 12) { if (a.g_c2p_t) a.g_c2p_t[(size_t)b * 3 + (i - 9)] = s; }
  else a.g_trans0[(size_t)b * 3 + (i - 12)] = s;
}

}  // namespace ha

using namespace ha;

extern "C" int ha_rollout_post_forward(const ha_rollout_post_args* args, void* stream) {
  HA_REQUIRE(args, "ha_rollout_post_forward: null argument");
  const ha_rollout_post_args& a = *args;
  HA_REQUIRE(a.B >= 1 && a.S >= 1, "ha_rollout_post_forward: B and S must be >= 1");
  HA_REQUIRE(a.world && a.trans0 && a.root0 && a.pose0 && a.joints0, "ha_rollout_post_forward: null input");
  HA_REQUIRE(a.trans && a.root_orient && a.pose_body && a.joints && a.contacts_conf && a.contacts, "ha_rollout_post_forward: null output");
  HA_REQUIRE((a.c2p_R == nullptr) == (a.c2p_t == nullptr), "ha_rollout_post_forward: c2p_R and c2p_t go together");
  HA_REQUIRE(!a.c2p_R || (a.cam_trans && a.cam_root_orient), "ha_rollout_post_forward: camera-frame outputs missing");
  HA_LAUNCH(rollout_post_fwd_kernel, dim3(a.B * (a.S + 1)), dim3(64), 0, (hipStream_t)stream, a);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

extern "C" int ha_rollout_post_backward(const ha_rollout_post_args* args, void* stream) {
  HA_REQUIRE(args, "ha_rollout_post_backward: null argument");
  const ha_rollout_post_args& a = *args;
  HA_REQUIRE(a.B >= 1 && a.S >= 1, "ha_rollout_post_backward: B and S must be >= 1");
  HA_REQUIRE(a.world && a.trans0 && a.root_orient && a.contacts_conf, "ha_rollout_post_backward: forward tensors missing");
  HA_REQUIRE(a.g_world && a.g_trans0 && a.g_root0 && a.g_pose0 && a.g_joints0 && a.partial, "ha_rollout_post_backward: null output");
  HA_REQUIRE(!a.c2p_R || (a.g_c2p_R && a.g_c2p_t), "ha_rollout_post_backward: cam2prior gradient outputs missing");
  hipStream_t st = (hipStream_t)stream;
  HA_LAUNCH(rollout_post_bwd_kernel, dim3(a.B * (a.S + 1)), dim3(64), 0, st, a);
  HA_LAUNCH_CHECK();
  HA_LAUNCH(rollout_post_reduce_kernel, dim3(a.B), dim3(64), 0, st, a);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

// Everything between the frame-0 SMPL evaluation and the roll-out, in one launch per direction (stage 3 with an optimised floor):
//   compute_cam2prior          humor/fitting/fitting_utils.py:149-190 (parse_floor_plane :87-104, compute_plane_intersection :61-76)
//   apply_cam2prior (forward)  humor/fitting/motion_optimizer.py:678-742 for the single key frame
//   initial roll-out state     humor/fitting/motion_optimizer.py:905-942 (batch_rodrigues of root / body pose, concatenation)
// The reference evaluates SMPL three times here (camera frame for cam2prior, prior frame inside apply_cam2prior for the root height,
// prior frame again for the initial joints).  Rotating the root about the rest root joint j0 is a rigid motion of the whole posed
// body, J(R Rc, t') = R (J(Rc, t) - j0 - t) + j0 + t', so the prior-frame joints follow from the camera-frame ones and ONE SMPL
// evaluation remains; the translated root lands on (0, 0, root_height - j0.z) exactly as in the reference (its R (trans + t)
// with t = -trans is identically zero).  ~230 element-wise launches (forward + autograd backward) and two SMPL calls per
// closure become two launches.  One wavefront per sequence; lane j < 22 owns SMPL joint j (and body rotation j - 1).
#include <string.h>

#include "rot_math.h"

namespace ha {

__device__ __forceinline__ float pre_wsum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void cross3(const float a[3], const float b[3], float o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

// the cam2prior frame of one sequence and the intermediates its adjoint needs
struct C2P {
  float L, sgn, nh[3];           // |floor|, orientation sign, floor / |floor|
  float n[3], off;               // plane (normal, offset)
  float Rc[9];                   // rodrigues(root0)
  float a1, q, s1;               // floor_trans = trans - s1 n, s1 = a1 / q, a1 = off - n.trans, q = -n.n
  float br[3], den, s2;          // body right, n.br, s2 = a1 / den
  float rs[3], lr, sg;           // signed right axis before normalisation, its length, sign
  float right[3], u[3], lu, fwd[3];
  float R[9];
};

__device__ __forceinline__ void c2p_forward(const float floor[3], const float trans[3], const float root[3], C2P& c) {
  c.L = sqrtf(dot3(floor, floor));
#pragma unroll
  for (int i = 0; i < 3; ++i) c.nh[i] = floor[i] / c.L;
  c.sgn = c.nh[1] > 0.f ? -1.f : 1.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) c.n[i] = c.sgn * c.nh[i];
  c.off = c.sgn * c.L;
  rodrigues(root, c.Rc);
  c.a1 = c.off - dot3(c.n, trans);
  c.q = -dot3(c.n, c.n);
  c.s1 = c.a1 / c.q;
  c.br[0] = -c.Rc[0]; c.br[1] = -c.Rc[3]; c.br[2] = -c.Rc[6];
  c.den = dot3(c.n, c.br);
  c.s2 = c.a1 / c.den;
  // right = (trans + s2 br) - (trans - s1 n)
  c.sg = c.s2 < 0.f ? -1.f : 1.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) c.rs[i] = c.sg * (c.s2 * c.br[i] + c.s1 * c.n[i]);
  c.lr = sqrtf(dot3(c.rs, c.rs));
#pragma unroll
  for (int i = 0; i < 3; ++i) c.right[i] = c.rs[i] / c.lr;
  cross3(c.n, c.right, c.u);
  c.lu = sqrtf(dot3(c.u, c.u));
#pragma unroll
  for (int i = 0; i < 3; ++i) { c.fwd[i] = c.u[i] / c.lu; c.R[i] = c.right[i]; c.R[3 + i] = c.fwd[i]; c.R[6 + i] = c.n[i]; }
}

__global__ __launch_bounds__(64) void fit_pre_fwd_kernel(ha_fit_pre_args a) {
  const int b = blockIdx.x, j = threadIdx.x;
  const float* floor = a.floor + (size_t)b * 3;
  const float* trans = a.trans0 + (size_t)b * 3;
  const float* root = a.root0 + (size_t)b * 3;
  const float* jc = a.jcam + (size_t)b * (a.jcam_stride ? a.jcam_stride : 66);
  float f3[3] = {floor[0], floor[1], floor[2]}, t3[3] = {trans[0], trans[1], trans[2]}, r3[3] = {root[0], root[1], root[2]};
  C2P c;
  c2p_forward(f3, t3, r3, c);                      // (every lane: uniform, no hand-off needed)
  const float jc0[3] = {jc[0], jc[1], jc[2]};
  const float j0[3] = {jc0[0] - t3[0], jc0[1] - t3[1], jc0[2] - t3[2]};
  const float hroot = (c.off - dot3(c.n, jc0)) / c.q;           // root height above the floor
  const float tp[3] = {0.f, 0.f, hroot - j0[2]};
  float* P = a.past_in + (size_t)b * 339;
  if (j < 22) {
    const float d[3] = {jc[j * 3] - jc0[0], jc[j * 3 + 1] - jc0[1], jc[j * 3 + 2] - jc0[2]};
    float o[3];
    mat3_vec(c.R, d, o);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float v = o[i] + j0[i] + tp[i];
      a.joints_p[((size_t)b * 22 + j) * 3 + i] = v;
      P[207 + 3 * j + i] = v;
      P[273 + 3 * j + i] = a.joints_vel[((size_t)b * 22 + j) * 3 + i];
    }
    if (j >= 1) {
      const float* pa = a.pose0 + (size_t)b * 63 + (j - 1) * 3;
      const float aa[3] = {pa[0], pa[1], pa[2]};
      float Rb[9];
      rodrigues(aa, Rb);
#pragma unroll
      for (int i = 0; i < 9; ++i) P[18 + 9 * (j - 1) + i] = Rb[i];
    } else {
      float Rp[9], aa[3], Rr[9];
      mat3_mul(c.R, c.Rc, Rp);
      rotmat_to_aa(Rp, aa);
      rodrigues(aa, Rr);                          // the reference converts to axis-angle and back (motion_optimizer.py:693-700, 917)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        a.root_p[(size_t)b * 3 + i] = aa[i];
        a.trans_p[(size_t)b * 3 + i] = tp[i];
        a.c2p_t[(size_t)b * 3 + i] = -t3[i];
        P[i] = tp[i];
        P[3 + i] = a.trans_vel[(size_t)b * 3 + i];
        P[15 + i] = a.root_orient_vel[(size_t)b * 3 + i];
      }
#pragma unroll
      for (int i = 0; i < 9; ++i) { P[6 + i] = Rr[i]; a.c2p_R[(size_t)b * 9 + i] = c.R[i]; }
      a.root_height[b] = hroot;
    }
  }
}

__global__ __launch_bounds__(64) void fit_pre_bwd_kernel(ha_fit_pre_args a) {
  const int b = blockIdx.x, j = threadIdx.x;
  const float* floor = a.floor + (size_t)b * 3;
  const float* trans = a.trans0 + (size_t)b * 3;
  const float* root = a.root0 + (size_t)b * 3;
  const float* jc = a.jcam + (size_t)b * (a.jcam_stride ? a.jcam_stride : 66);
  float f3[3] = {floor[0], floor[1], floor[2]}, t3[3] = {trans[0], trans[1], trans[2]}, r3[3] = {root[0], root[1], root[2]};
  C2P c;
  c2p_forward(f3, t3, r3, c);
  const float jc0[3] = {jc[0], jc[1], jc[2]};
  const float* GP = a.g_past_in ? a.g_past_in + (size_t)b * 339 : nullptr;
  auto gp = [&](int i) { return GP ? GP[i] : 0.f; };
  auto opt3 = [&](const float* p, size_t off, int i) { return p ? p[off + i] : 0.f; };

  // ---- per-joint part: joints_p = R (jc_j - jc_0) + j0 + trans_p ; body rotations ; velocities pass through ------------------------
  float gjp[3] = {0.f, 0.f, 0.f}, gd[3] = {0.f, 0.f, 0.f}, gRj[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) gRj[i] = 0.f;
  if (j < 22) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      gjp[i] = gp(207 + 3 * j + i) + opt3(a.g_joints_p, ((size_t)b * 22 + j) * 3, i);
      a.g_joints_vel[((size_t)b * 22 + j) * 3 + i] = gp(273 + 3 * j + i) + opt3(a.add_joints_vel, ((size_t)b * 22 + j) * 3, i);
    }
    const float d[3] = {jc[j * 3] - jc0[0], jc[j * 3 + 1] - jc0[1], jc[j * 3 + 2] - jc0[2]};
    mat3_tvec(c.R, gjp, gd);                      // R^T g
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) gRj[r * 3 + cc] = gjp[r] * d[cc];
    if (j >= 1) {
      const float* pa = a.pose0 + (size_t)b * 63 + (j - 1) * 3;
      const float aa[3] = {pa[0], pa[1], pa[2]};
      float gRb[9], gaa[3];
#pragma unroll
      for (int i = 0; i < 9; ++i) gRb[i] = gp(18 + 9 * (j - 1) + i);
      rodrigues_bwd(aa, gRb, gaa);
#pragma unroll
      for (int i = 0; i < 3; ++i) a.g_pose0[(size_t)b * 63 + (j - 1) * 3 + i] = gaa[i] + opt3(a.add_pose0, (size_t)b * 63 + (j - 1) * 3, i);
    }
  }
  // sums over the joints
  float gR[9], s_gjp[3], s_gd[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) gR[i] = pre_wsum(gRj[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) { s_gjp[i] = pre_wsum(gjp[i]); s_gd[i] = pre_wsum(gd[i]); }
  // d/d jcam: joint j gets R^T g_j, joint 0 additionally -(sum_j R^T g_j) (written below together with the j0 / height terms)
  if (j >= 1 && j < 22) {
#pragma unroll
    for (int i = 0; i < 3; ++i) a.g_jcam[((size_t)b * 22 + j) * 3 + i] = gd[i];
  }
  if (j != 0) return;

  // ---- the sequence-level chain (lane 0) ----------------------------------------------------------------------------------------
  float g_tp[3], g_rp[3], g_t[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    g_tp[i] = gp(i) + opt3(a.g_trans_p, (size_t)b * 3, i) + s_gjp[i];
    g_rp[i] = opt3(a.g_root_p, (size_t)b * 3, i);
    g_t[i] = opt3(a.g_c2p_t, (size_t)b * 3, i);
    a.g_trans_vel[(size_t)b * 3 + i] = gp(3 + i) + opt3(a.add_trans_vel, (size_t)b * 3, i);
    a.g_root_orient_vel[(size_t)b * 3 + i] = gp(15 + i) + opt3(a.add_root_orient_vel, (size_t)b * 3, i);
  }
  if (a.g_c2p_R) {
#pragma unroll
    for (int i = 0; i < 9; ++i) gR[i] += a.g_c2p_R[(size_t)b * 9 + i];
  }
  float g_j0[3] = {s_gjp[0], s_gjp[1], s_gjp[2]};
  // trans_p = (0, 0, hroot - j0.z)
  float g_h = g_tp[2] + (a.g_root_height ? a.g_root_height[b] : 0.f);
  g_j0[2] -= g_tp[2];
  float g_jc0[3], g_tr[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    g_jc0[i] = gd[i] - s_gd[i] + g_j0[i];        // own R^T g_0, minus everybody's (d_j = jc_j - jc_0), plus j0 = jc_0 - trans
    g_tr[i] = -g_j0[i] - g_t[i];                 // j0 = jc_0 - trans ; c2p_t = -trans
  }
  // root orientation: root_p = aa(Rp), Rp = R Rc, P[6:15] = rodrigues(root_p)
  float Rp[9], aa[3], gRr[9], gaa[3], gRp[9], gRc[9], M[9];
  mat3_mul(c.R, c.Rc, Rp);
  rotmat_to_aa(Rp, aa);
#pragma unroll
  for (int i = 0; i < 9; ++i) gRr[i] = gp(6 + i);
  rodrigues_bwd(aa, gRr, gaa);
#pragma unroll
  for (int i = 0; i < 3; ++i) g_rp[i] += gaa[i];
  rotmat_to_aa_bwd(Rp, g_rp, gRp);
  mat3_mult(gRp, c.Rc, M);                        // gR += gRp Rc^T
#pragma unroll
  for (int i = 0; i < 9; ++i) gR[i] += M[i];
  mat3_tmul(c.R, gRp, gRc);                       // gRc = R^T gRp
  // root height h = (off - n.jc0) / q
  float g_n[3] = {gR[6], gR[7], gR[8]}, g_off = 0.f;
  {
    const float ah = c.off - dot3(c.n, jc0);
    const float g_a = g_h / c.q, g_q = -g_h * ah / (c.q * c.q);
    g_off += g_a;
#pragma unroll
    for (int i = 0; i < 3; ++i) { g_n[i] += -g_a * jc0[i] - 2.f * g_q * c.n[i]; g_jc0[i] -= g_a * c.n[i]; }
  }
  // fwd = u / |u|, u = n x right ; right = rs / |rs|
  float g_right[3] = {gR[0], gR[1], gR[2]};
  {
    const float g_f[3] = {gR[3], gR[4], gR[5]};
    const float pf = dot3(g_f, c.fwd);
    float g_u[3], t1[3], t2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) g_u[i] = (g_f[i] - pf * c.fwd[i]) / c.lu;
    cross3(c.right, g_u, t1);                     // d/dn  (n x r).g = r x g
    cross3(g_u, c.n, t2);                         // d/dr  = g x n
#pragma unroll
    for (int i = 0; i < 3; ++i) { g_n[i] += t1[i]; g_right[i] += t2[i]; }
  }
  float g_s1, g_s2, g_br[3];
  {
    const float pr = dot3(g_right, c.right);
    float g_rs[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) g_rs[i] = c.sg * (g_right[i] - pr * c.right[i]) / c.lr;      // w.r.t. (s2 br + s1 n)
    g_s2 = dot3(g_rs, c.br);
    g_s1 = dot3(g_rs, c.n);
#pragma unroll
    for (int i = 0; i < 3; ++i) { g_br[i] = c.s2 * g_rs[i]; g_n[i] += c.s1 * g_rs[i]; }
  }
  // s1 = a1 / q, s2 = a1 / den ; a1 = off - n.trans, q = -n.n, den = n.br
  {
    const float g_a1 = g_s1 / c.q + g_s2 / c.den;
    const float g_q = -g_s1 * c.a1 / (c.q * c.q), g_den = -g_s2 * c.a1 / (c.den * c.den);
    g_off += g_a1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      g_n[i] += -g_a1 * t3[i] - 2.f * g_q * c.n[i] + g_den * c.br[i];
      g_tr[i] += -g_a1 * c.n[i];
      g_br[i] += g_den * c.n[i];
    }
  }
  gRc[0] -= g_br[0]; gRc[3] -= g_br[1]; gRc[6] -= g_br[2];       // br = -Rc[:, 0]
  float g_root[3];
  rodrigues_bwd(r3, gRc, g_root);
  // plane: n = sgn floor / L, off = sgn L
  {
    float g_nh[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) g_nh[i] = c.sgn * g_n[i];
    const float pn = dot3(g_nh, c.nh);
#pragma unroll
    for (int i = 0; i < 3; ++i)
      a.g_floor[(size_t)b * 3 + i] = (g_nh[i] - pn * c.nh[i]) / c.L + c.sgn * g_off * c.nh[i] + opt3(a.add_floor, (size_t)b * 3, i);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    a.g_trans0[(size_t)b * 3 + i] = g_tr[i];
    a.g_root0[(size_t)b * 3 + i] = g_root[i];
    a.g_jcam[(size_t)b * 22 * 3 + i] = g_jc0[i];
  }
}

}  // namespace ha

using namespace ha;

extern "C" int ha_fit_pre_forward(const ha_fit_pre_args* args, void* stream) {
  HA_REQUIRE(args, "ha_fit_pre_forward: null argument");
  const ha_fit_pre_args& a = *args;
  HA_REQUIRE(a.B >= 1, "ha_fit_pre_forward: B must be >= 1");
  HA_REQUIRE(a.floor && a.trans0 && a.root0 && a.pose0 && a.jcam && a.trans_vel && a.joints_vel && a.root_orient_vel, "ha_fit_pre_forward: null input");
  HA_REQUIRE(a.past_in && a.trans_p && a.root_p && a.joints_p && a.c2p_R && a.c2p_t && a.root_height, "ha_fit_pre_forward: null output");
  HA_LAUNCH(fit_pre_fwd_kernel, dim3(a.B), dim3(64), 0, (hipStream_t)stream, a);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

extern "C" int ha_fit_pre_backward(const ha_fit_pre_args* args, void* stream) {
  HA_REQUIRE(args, "ha_fit_pre_backward: null argument");
  const ha_fit_pre_args& a = *args;
  HA_REQUIRE(a.B >= 1, "ha_fit_pre_backward: B must be >= 1");
  HA_REQUIRE(a.floor && a.trans0 && a.root0 && a.pose0 && a.jcam, "ha_fit_pre_backward: null input");
  HA_REQUIRE(a.g_floor && a.g_trans0 && a.g_root0 && a.g_pose0 && a.g_jcam && a.g_trans_vel && a.g_joints_vel && a.g_root_orient_vel,
             "ha_fit_pre_backward: null gradient output");
  HA_LAUNCH(fit_pre_bwd_kernel, dim3(a.B), dim3(64), 0, (hipStream_t)stream, a);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

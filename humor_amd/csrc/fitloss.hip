// Fused evaluation of the fitting objective's data / regularisation terms and of their gradients: one launch replaces the
// ~250 element-wise / reduction launches (forward + autograd backward) of FittingLoss.root_fit / smpl_fit / motion_fit
// (humor/fitting/fitting_loss.py:94-309; terms :317-484, gmof fitting_utils.py:250-258, perspective_projection :647-676 with
// identity extrinsics).  One wavefront per frame (b, t); every term is a sum over frames, so the wave accumulates its share of
// each term in registers, writes one row of partial sums, and writes the gradient of the WEIGHTED loss with respect to every
// input it owns -- temporal terms (smoothness, bone-length change, contact velocity, overlap consistency) in gather form: the
// wave of frame t recomputes the residuals of the pairs (t-1, t) and (t, t+1) it takes part in, so no gradient is ever
// accumulated across waves (no atomics: bit-reproducible, which the replicated multi-GPU L-BFGS relies on).  A second
// single-block launch adds the per-frame partial sums in a fixed order.
#include "common.h"

namespace ha {

constexpr int NT = HA_FIT_NTERMS;
constexpr float CONTACT_HEIGHT_THRESH = 0.08f;   // fitting_loss.py:18
// humor/body_model/utils.py:9 (bone-length loss only)
__device__ const int SMPL_PARENTS_DEV[22] = {-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 12, 12, 13, 14, 16, 17, 18, 19};

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ bool visible(float o) { return !(o == INFINITY || o == -INFINITY); }   // get_visible_mask: ~isinf

__global__ __launch_bounds__(64) void fit_loss_kernel(ha_fit_args a) {
  const int f = blockIdx.x, lane = threadIdx.x;
  const int T = a.T, b = f / T, t = f - b * T;
  extern __shared__ __attribute__((aligned(16))) float smem[];     // [128] joint -> OpenPose index | [66] bone-length adjoints
  int* s_inv = reinterpret_cast<int*>(smem);
  float* s_cu = smem + 128;
  float tv[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) tv[k] = 0.f;
  const float* w = a.w;

  // ---- camera-frame joints: re-projection (GMoF), 3D joint observations, smoothness ---------------------------------------
  if (a.cam_jtr) {
    const int nj = a.nj;
    for (int j = lane; j < nj; j += 64) s_inv[j] = -1;
    __syncthreads();
    const bool use2d = a.obs_j2d != nullptr && w[HA_FIT_J2D] != 0.f;
    if (use2d && lane < 25) s_inv[a.smpl2op[lane]] = lane;
    __syncthreads();
    const float* P = a.cam_jtr + (size_t)f * nj * 3;
    for (int j = lane; j < nj; j += 64) {
      const float p[3] = {P[j * 3], P[j * 3 + 1], P[j * 3 + 2]};
      float g[3] = {0.f, 0.f, 0.f};
      const int k = s_inv[j];
      if (use2d && k >= 0) {
        const float* o = a.obs_j2d + ((size_t)f * 25 + k) * 3;
        const float conf = o[2] * a.op_mask[k], c2 = conf * conf;
        const float fx = a.cam_f[b * 2], fy = a.cam_f[b * 2 + 1], cx = a.cam_c[b * 2], cy = a.cam_c[b * 2 + 1];
        const float rx = (p[0] / p[2]) * fx + cx - o[0], ry = (p[1] / p[2]) * fy + cy - o[1];
        const float s2 = a.sigma * a.sigma;
        const float x2 = rx * rx, y2 = ry * ry;
        tv[HA_FIT_J2D] += c2 * ((s2 * x2) / (s2 + x2) + (s2 * y2) / (s2 + y2));
        // d/dr [s2 r^2 / (s2 + r^2)] = 2 r s2^2 / (s2 + r^2)^2
        const float dx = 2.f * rx * s2 * s2 / ((s2 + x2) * (s2 + x2)), dy = 2.f * ry * s2 * s2 / ((s2 + y2) * (s2 + y2));
        const float W = w[HA_FIT_J2D] * c2, iz = 1.f / p[2];
        g[0] += W * dx * fx * iz;
        g[1] += W * dy * fy * iz;
        g[2] -= W * (dx * fx * p[0] + dy * fy * p[1]) * iz * iz;
      }
      if (j < 22) {
        if (a.obs_j3d && w[HA_FIT_J3D] != 0.f) {
          const float* o = a.obs_j3d + ((size_t)f * 22 + j) * 3;
#pragma unroll
          for (int c = 0; c < 3; ++c)
            if (visible(o[c])) { const float d = o[c] - p[c]; tv[HA_FIT_J3D] += 0.5f * d * d; g[c] -= w[HA_FIT_J3D] * d; }
        }
        if (w[HA_FIT_SMOOTH] != 0.f) {
          if (t > 0) {
            const float* Q = P - (size_t)nj * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) { const float d = p[c] - Q[j * 3 + c]; tv[HA_FIT_SMOOTH] += 0.5f * d * d; g[c] += w[HA_FIT_SMOOTH] * d; }
          }
          if (t < T - 1) {
            const float* Q = P + (size_t)nj * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) g[c] -= w[HA_FIT_SMOOTH] * (Q[j * 3 + c] - p[c]);
          }
        }
      }
      if (a.g_cam_jtr) {
        float* G = a.g_cam_jtr + ((size_t)f * nj + j) * 3;
        G[0] = g[0]; G[1] = g[1]; G[2] = g[2];
      }
    }
  }

  // ---- key vertices: 3D vertex observations and the overlap consistency of consecutive sub-sequences --------------------------
  if (a.cam_verts) {
    const int nv = a.nv;
    const float* V = a.cam_verts;
    const float wov = w[HA_FIT_OV_VPOS];
    // role "cur": this frame is position k = t of the pair (b-1, b); role "prev": position k = t - (T - ov') of the pair (b, b+1)
    const int ov_c = (a.overlap && wov != 0.f && (b > 0 || a.prev_tail)) ? a.overlap[b] : 0;
    const int ov_p = (a.overlap && wov != 0.f && b + 1 < a.B) ? a.overlap[b + 1] : 0;
    const bool cur_role = t < ov_c, prev_role = ov_p > 0 && t >= T - ov_p;
    // the halo tail (the predecessor of local sequence 0 lives on another rank): its gradient rows are produced by the blocks of b = 0
    const bool halo_role = b == 0 && a.prev_tail && a.g_prev_tail && ov_c > 0;
    for (int v = lane; v < nv; v += 64) {
      const size_t e = ((size_t)f * nv + v) * 3;
      float g[3] = {0.f, 0.f, 0.f}, gh[3] = {0.f, 0.f, 0.f};
      if (a.obs_v3d && w[HA_FIT_V3D] != 0.f) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float o = a.obs_v3d[e + c];
          if (visible(o)) { const float d = o - V[e + c]; tv[HA_FIT_V3D] += 0.5f * d * d; g[c] -= w[HA_FIT_V3D] * d; }
        }
      }
      // d_k = prev[k] - cur[k]; e_k = dL/dd_k = d_k + (d_k - d_{k-1}) [k >= 1] - (d_{k+1} - d_k) [k <= ov-2]
      auto pair_e = [&](const float* prevseq, const float* curseq, int ov, int k, int c, float& dk, float& ddk) {
        // prevseq / curseq point at frame 0 of the two sequences (element (v, c) of frame tt at [tt * nv * 3])
        auto d = [&](int kk) { return prevseq[(size_t)(T - ov + kk) * nv * 3] - curseq[(size_t)kk * nv * 3]; };
        dk = d(k);
        float ek = dk;
        ddk = 0.f;
        if (k >= 1) { ddk = dk - d(k - 1); ek += ddk; }
        if (k <= ov - 2) ek -= d(k + 1) - dk;
        return ek;
      };
      if (cur_role) {
        const float* cs = V + ((size_t)b * T * nv + v) * 3;
        const float* ps = b > 0 ? V + ((size_t)(b - 1) * T * nv + v) * 3 : a.prev_tail + (size_t)v * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float dk, ddk;
          const float ek = pair_e(ps + c, cs + c, ov_c, t, c, dk, ddk);
          tv[HA_FIT_OV_VPOS] += 0.5f * dk * dk;
          tv[HA_FIT_OV_VVEL] += 0.5f * ddk * ddk;        // velocity residual (k-1, k), counted at k
          g[c] -= wov * ek;
        }
      }
      if (prev_role) {
        const float* ps = V + ((size_t)b * T * nv + v) * 3;
        const float* cs = V + ((size_t)(b + 1) * T * nv + v) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float dk, ddk;
          g[c] += wov * pair_e(ps + c, cs + c, ov_p, t - (T - ov_p), c, dk, ddk);
        }
      }
      if (halo_role && t >= T - ov_c) {
        const float* ps = a.prev_tail + (size_t)v * 3;
        const float* cs = V + (size_t)v * 3;          // b == 0
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float dk, ddk;
          gh[c] = wov * pair_e(ps + c, cs + c, ov_c, t - (T - ov_c), c, dk, ddk);
        }
      }
      if (a.g_cam_verts) { a.g_cam_verts[e] = g[0]; a.g_cam_verts[e + 1] = g[1]; a.g_cam_verts[e + 2] = g[2]; }
      if (b == 0 && a.g_prev_tail) {
        float* G = a.g_prev_tail + ((size_t)t * nv + v) * 3;
        G[0] = gh[0]; G[1] = gh[1]; G[2] = gh[2];
      }
    }
  }

  // ---- prior-frame joints / roll-out joints / contact confidences -----------------------------------------------------------
  if (a.pri_joints || a.ro_joints) {
    const int j = lane;
    float gp[3] = {0.f, 0.f, 0.f}, gr[3] = {0.f, 0.f, 0.f}, gc = 0.f;
    float cu[3] = {0.f, 0.f, 0.f};
    const size_t e = ((size_t)f * 22 + j) * 3;
    const int pnj = a.pri_nj;                              // joints per frame of the prior-frame tensor (>= 22)
    const size_t ep = ((size_t)f * pnj + j) * 3;
    if (j < 22) {
      float pj[3] = {0.f, 0.f, 0.f}, rj[3] = {0.f, 0.f, 0.f};
      if (a.pri_joints) { pj[0] = a.pri_joints[ep]; pj[1] = a.pri_joints[ep + 1]; pj[2] = a.pri_joints[ep + 2]; }
      if (a.ro_joints) { rj[0] = a.ro_joints[e]; rj[1] = a.ro_joints[e + 1]; rj[2] = a.ro_joints[e + 2]; }
      if (a.pri_joints && a.ro_joints && w[HA_FIT_JOINT_CONSIST] != 0.f) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float d = pj[c] - rj[c];
          tv[HA_FIT_JOINT_CONSIST] += 0.5f * d * d;
          gp[c] += w[HA_FIT_JOINT_CONSIST] * d;
          gr[c] -= w[HA_FIT_JOINT_CONSIST] * d;
        }
      }
      if (a.ro_joints && a.obs_j3d && w[HA_FIT_J3D_RO] != 0.f) {
        const float* o = a.obs_j3d + e;
#pragma unroll
        for (int c = 0; c < 3; ++c)
          if (visible(o[c])) { const float d = o[c] - rj[c]; tv[HA_FIT_J3D_RO] += 0.5f * d * d; gr[c] -= w[HA_FIT_J3D_RO] * d; }
      }
      if (a.ro_joints && w[HA_FIT_BONE_LEN] != 0.f && j >= 1) {
        const int par = SMPL_PARENTS_DEV[j];
        auto blen = [&](int tt, float (&u)[3]) {
          const float* q = a.ro_joints + ((size_t)(b * T + tt) * 22) * 3;
#pragma unroll
          for (int c = 0; c < 3; ++c) u[c] = q[j * 3 + c] - q[par * 3 + c];
          return sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        };
        float u[3], tmp[3];
        const float L0 = blen(t, u);
        float coef = 0.f;
        if (t > 0) { const float d = L0 - blen(t - 1, tmp); tv[HA_FIT_BONE_LEN] += 0.5f * d * d; coef += d; }
        if (t < T - 1) coef -= blen(t + 1, tmp) - L0;
        coef *= w[HA_FIT_BONE_LEN];
        const float il = L0 > 0.f ? 1.f / L0 : 0.f;      // torch.norm's subgradient at 0 is 0
#pragma unroll
        for (int c = 0; c < 3; ++c) cu[c] = coef * u[c] * il;
      }
      if (a.pri_joints && a.contacts_conf) {
        const float conf = a.contacts_conf[(size_t)f * 22 + j];
        if (w[HA_FIT_CONTACT_VEL] != 0.f) {
          if (t > 0) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const float d = pj[c] - a.pri_joints[ep - (size_t)pnj * 3 + c];
              s += d * d;
              gp[c] += w[HA_FIT_CONTACT_VEL] * conf * d;
            }
            tv[HA_FIT_CONTACT_VEL] += 0.5f * s * conf;
            gc += w[HA_FIT_CONTACT_VEL] * 0.5f * s;
          }
          if (t < T - 1) {
            const float cn = a.contacts_conf[(size_t)(f + 1) * 22 + j];
#pragma unroll
            for (int c = 0; c < 3; ++c) gp[c] -= w[HA_FIT_CONTACT_VEL] * cn * (a.pri_joints[ep + (size_t)pnj * 3 + c] - pj[c]);
          }
        }
        if (w[HA_FIT_CONTACT_H] != 0.f) {
          const float az = fabsf(pj[2]), fd = az > CONTACT_HEIGHT_THRESH ? az - CONTACT_HEIGHT_THRESH : 0.f;
          tv[HA_FIT_CONTACT_H] += fd * conf;
          gc += w[HA_FIT_CONTACT_H] * fd;
          if (az > CONTACT_HEIGHT_THRESH) gp[2] += w[HA_FIT_CONTACT_H] * conf * (pj[2] > 0.f ? 1.f : (pj[2] < 0.f ? -1.f : 0.f));
        }
      }
    }
    // bone-length adjoint: joint j receives +cu[j] as the child end of its bone and -cu[child] from every bone it parents
    __syncthreads();
    if (j < 22) { s_cu[j * 3] = cu[0]; s_cu[j * 3 + 1] = cu[1]; s_cu[j * 3 + 2] = cu[2]; }
    __syncthreads();
    if (j < 22) {
#pragma unroll
      for (int c = 0; c < 3; ++c) gr[c] += cu[c];
      for (int ch = 1; ch < 22; ++ch)
        if (SMPL_PARENTS_DEV[ch] == j) { gr[0] -= s_cu[ch * 3]; gr[1] -= s_cu[ch * 3 + 1]; gr[2] -= s_cu[ch * 3 + 2]; }
      if (a.gmm_gx && t == 0 && a.gmm_g[0]) {     // folded init-state prior: its gradient of frame 0's joints (segment 0 of x_b)
        const float* gx = a.gmm_gx + (size_t)b * a.gmm_D + j * 3;
        gp[0] = fmaf(a.gmm_w, gx[0], gp[0]); gp[1] = fmaf(a.gmm_w, gx[1], gp[1]); gp[2] = fmaf(a.gmm_w, gx[2], gp[2]);
      }
      if (a.g_pri_joints) { a.g_pri_joints[ep] = gp[0]; a.g_pri_joints[ep + 1] = gp[1]; a.g_pri_joints[ep + 2] = gp[2]; }
      if (a.g_ro_joints) { a.g_ro_joints[e] = gr[0]; a.g_ro_joints[e + 1] = gr[1]; a.g_ro_joints[e + 2] = gr[2]; }
      if (a.g_contacts_conf) a.g_contacts_conf[(size_t)f * 22 + j] = gc;
    }
    if (a.g_pri_joints)          // joints beyond the 22 body joints (hands, selected vertices) carry no prior-frame term
      for (int i = 22 * 3 + lane; i < pnj * 3; i += 64) a.g_pri_joints[(size_t)f * pnj * 3 + i] = 0.f;
  }
  // folded init-state prior: segments 1.. of its gradient (joint / root velocities of the initial state) to their own slots
  if (a.gmm_gx && t == 0) {
    int c0 = a.gmm_seg_width[0];
    for (int sgm = 1; sgm < a.gmm_nseg; ++sgm) {
      float* dst = a.gmm_g[sgm];
      const int wd = a.gmm_seg_width[sgm];
      if (dst)
        for (int i = lane; i < wd; i += 64) dst[(size_t)b * a.gmm_g_stride[sgm] + i] = a.gmm_w * a.gmm_gx[(size_t)b * a.gmm_D + c0 + i];
      c0 += wd;
    }
  }

  // ---- pose prior (VPoser latent), motion prior (conditional Gaussian or standard normal) -----------------------------------
  if (a.latent_pose) {
    for (int c = lane; c < a.dlp; c += 64) {
      const float x = a.latent_pose[(size_t)f * a.dlp + c];
      if (w[HA_FIT_POSE_PRIOR] != 0.f) tv[HA_FIT_POSE_PRIOR] += x * x;
      if (a.g_latent_pose) a.g_latent_pose[(size_t)f * a.dlp + c] = 2.f * w[HA_FIT_POSE_PRIOR] * x;
    }
  }
  if (a.latent_motion && t < a.S) {
    const float W = w[HA_FIT_MOTION_PRIOR];
    for (int c = lane; c < a.dz; c += 64) {
      const size_t i = ((size_t)b * a.S + t) * a.dz + c;
      const float x = a.latent_motion[i];
      float gx = 0.f, gm = 0.f, gvv = 0.f;
      if (W != 0.f) {
        if (a.prior_mu) {
          // -log N(x; m, v) = log sqrt(v) + log sqrt(2 pi) + (x - m)^2 / (2 v)      (fitting_loss.py:504-516)
          const float m = a.prior_mu[i], v = a.prior_var[i], d = x - m;
          tv[HA_FIT_MOTION_PRIOR] += logf(sqrtf(v)) + 0.918938533204672742f + d * d / (2.f * v);
          gx = W * d / v;
          gm = -gx;
          gvv = W * (0.5f / v - d * d / (2.f * v * v));
        } else {
          tv[HA_FIT_MOTION_PRIOR] += x * x;
          gx = 2.f * W * x;
        }
      }
      if (a.g_latent_motion) a.g_latent_motion[i] = gx;
      if (a.g_prior_mu) a.g_prior_mu[i] = gm;
      if (a.g_prior_var) a.g_prior_var[i] = gvv;
    }
  }

  // ---- per-sequence terms (frame 0 of each sequence): shape prior, floor regulariser, betas / floor overlap consistency --------
  if (t == 0) {
    const bool has_prev = b > 0, has_next = b + 1 < a.B;
    const float wovb = a.overlap ? w[HA_FIT_OV_BETAS] : 0.f, wovf = a.overlap ? w[HA_FIT_OV_FLOOR] : 0.f;
    if (a.betas) {
      for (int c = lane; c < a.nb; c += 64) {
        const float x = a.betas[(size_t)b * a.nb + c];
        float g = 0.f, gh = 0.f;
        if (w[HA_FIT_SHAPE_PRIOR] != 0.f) { tv[HA_FIT_SHAPE_PRIOR] += x * x; g += 2.f * w[HA_FIT_SHAPE_PRIOR] * a.nsteps * x; }
        if (wovb != 0.f) {
          if (has_prev || a.prev_betas) {
            const float d = (has_prev ? a.betas[(size_t)(b - 1) * a.nb + c] : a.prev_betas[c]) - x;
            tv[HA_FIT_OV_BETAS] += 0.5f * d * d;
            g -= wovb * d;
            gh = wovb * d;
          }
          if (has_next) g += wovb * (x - a.betas[(size_t)(b + 1) * a.nb + c]);
        }
        if (a.g_betas) a.g_betas[(size_t)b * a.nb + c] = g;
        if (b == 0 && a.g_prev_betas) a.g_prev_betas[c] = gh;
      }
    }
    if (a.floor && lane < 3) {
      const int c = lane;
      const float x = a.floor[b * 3 + c];
      float g = 0.f, gh = 0.f;
      if (a.obs_floor && w[HA_FIT_FLOOR_REG] != 0.f) {
        const float d = x - a.obs_floor[b * 4 + c] * a.obs_floor[b * 4 + 3];
        tv[HA_FIT_FLOOR_REG] += 0.5f * d * d;
        g += w[HA_FIT_FLOOR_REG] * a.nsteps * d;
      }
      if (wovf != 0.f) {
        if (has_prev || a.prev_floor) {
          const float d = (has_prev ? a.floor[(b - 1) * 3 + c] : a.prev_floor[c]) - x;
          tv[HA_FIT_OV_FLOOR] += 0.5f * d * d;
          g -= wovf * d;
          gh = wovf * d;
        }
        if (has_next) g += wovf * (x - a.floor[(b + 1) * 3 + c]);
      }
      if (a.g_floor) a.g_floor[b * 3 + c] = g;
      if (b == 0 && a.g_prev_floor) a.g_prev_floor[c] = gh;
    }
  }

  // ---- this frame's share of every term ------------------------------------------------------------------------------------------
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const float s = wsum(tv[k]);
    if (lane == 0) a.partial[(size_t)f * NT + k] = s;
  }
}

// terms[k] = sum over frames (fixed order), loss = sum_k weff_k terms[k].  One wave per term (lane-strided partial sums, then
// an xor-shuffle tree: a fixed association, so the value is bit-reproducible), no block barriers in the summation.
__global__ __launch_bounds__(1024) void fit_reduce_kernel(ha_fit_args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];     // [NT] term values | init-state prior total
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, F = a.B * a.T;
  for (int k = wave; k < NT; k += 16) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int f = lane;
    for (; f + 192 < F; f += 256) {
      s0 += a.partial[(size_t)f * NT + k];
      s1 += a.partial[(size_t)(f + 64) * NT + k];
      s2 += a.partial[(size_t)(f + 128) * NT + k];
      s3 += a.partial[(size_t)(f + 192) * NT + k];
    }
    for (; f < F; f += 64) s0 += a.partial[(size_t)f * NT + k];
    const float s = wsum((s0 + s1) + (s2 + s3));
    if (lane == 0) smem[k] = s;
  }
  // init-state prior total: one wave, lanes stride over the sequences (independent loads), fixed-order butterfly
  if (a.gmm_nll && wave == 15) {
    float v = 0.f;
    for (int b = lane; b < a.B; b += 64) v += a.gmm_nll[b];
    v = wsum(v);
    if (lane == 0) smem[NT] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float loss = 0.f;
    for (int k = 0; k < NT; ++k) {
      a.terms[k] = smem[k];
      const float scale = (k == HA_FIT_SHAPE_PRIOR || k == HA_FIT_FLOOR_REG) ? a.nsteps : 1.f;
      loss += a.w[k] * scale * smem[k];
    }
    if (a.gmm_nll) {
      const float tot = smem[NT];
      if (a.gmm_total) a.gmm_total[0] = tot;
      loss += a.gmm_w * tot;
    }
    a.loss[0] = loss;
  }
}

}  // namespace ha

extern "C" int ha_fit_loss(const ha_fit_args* args, void* stream) {
  using namespace ha;
  HA_REQUIRE(args, "ha_fit_loss: null argument");
  const ha_fit_args& a = *args;
  HA_REQUIRE(a.B >= 1 && a.T >= 1, "ha_fit_loss: B and T must be >= 1");
  HA_REQUIRE(a.partial && a.terms && a.loss, "ha_fit_loss: partial / terms / loss buffers are required");
  HA_REQUIRE(!a.cam_jtr || (a.nj >= 22 && a.nj <= 128), "ha_fit_loss: nj must be in [22, 128]");
  HA_REQUIRE(!a.pri_joints || a.pri_nj >= 22, "ha_fit_loss: pri_nj must be >= 22");
  HA_REQUIRE(!a.obs_j2d || (a.cam_jtr && a.smpl2op && a.op_mask && a.cam_f && a.cam_c), "ha_fit_loss: joints2d needs cam_jtr, smpl2op, op_mask and intrinsics");
  HA_REQUIRE(!a.latent_motion || (a.S >= 1 && a.S <= a.T && a.dz >= 1), "ha_fit_loss: latent steps S must be in [1, T]");
  HA_REQUIRE((a.prior_mu == nullptr) == (a.prior_var == nullptr), "ha_fit_loss: prior_mu and prior_var go together");
  HA_REQUIRE(!a.prev_tail || (a.cam_verts && a.overlap), "ha_fit_loss: a halo tail needs cam_verts and the overlap table");
  HA_REQUIRE(!a.gmm_gx || (a.gmm_nll && a.gmm_D >= 1 && a.gmm_nseg >= 1 && a.gmm_nseg <= 4), "ha_fit_loss: the folded init-state prior needs gmm_nll, gmm_D and 1..4 segments");
  hipStream_t st = (hipStream_t)stream;
  HA_LAUNCH(fit_loss_kernel, dim3(a.B * a.T), dim3(64), (128 + 66) * sizeof(float), st, a);
  HA_LAUNCH_CHECK();
  HA_LAUNCH(fit_reduce_kernel, dim3(1), dim3(1024), (NT + 1) * sizeof(float), st, a);
  HA_LAUNCH_CHECK();
  return HA_OK;
}

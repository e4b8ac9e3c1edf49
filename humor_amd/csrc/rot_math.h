// Rotation conversions used on the roll-out path (device + host), fp32.
#pragma once
#include "common.h"

namespace ha {

// ---------------------------------------------------------------------------------------------------
// rotation matrix -> axis-angle, restating rotation_matrix_to_angle_axis (humor/utils/transforms.py:243-389):
// transpose, 4-way masked quaternion extraction (masks on m22 < 1e-6, m00 > m11, m00 < -m11), q *= 0.5/sqrt(t),
// two_theta = 2 atan2(+-s, +-c), k = two_theta / s (2 when s^2 == 0), NaN -> 0.
// `br` returns the selected branch, `q` the quaternion (w,x,y,z) for the backward pass.
// ---------------------------------------------------------------------------------------------------
struct QuatBranch {
  int br;
  float n[4];   // un-normalised numerators
  float t;      // branch trace term
};

HA_HD void rot_to_quat_branch(const float R[9], QuatBranch& o) {
  // m = R^T : m[i][j] = R[j][i]
  const float m00 = R[0], m01 = R[3], m02 = R[6];
  const float m10 = R[1], m11 = R[4], m12 = R[7];
  const float m20 = R[2], m21 = R[5], m22 = R[8];
  const bool d2 = m22 < 1e-6f, d01 = m00 > m11, d0n1 = m00 < -m11;
  if (d2 && d01) {
    o.br = 0; o.t = 1.f + m00 - m11 - m22;
    o.n[0] = m12 - m21; o.n[1] = o.t; o.n[2] = m01 + m10; o.n[3] = m20 + m02;
  } else if (d2) {
    o.br = 1; o.t = 1.f - m00 + m11 - m22;
    o.n[0] = m20 - m02; o.n[1] = m01 + m10; o.n[2] = o.t; o.n[3] = m12 + m21;
  } else if (d0n1) {
    o.br = 2; o.t = 1.f - m00 - m11 + m22;
    o.n[0] = m01 - m10; o.n[1] = m20 + m02; o.n[2] = m12 + m21; o.n[3] = o.t;
  } else {
    o.br = 3; o.t = 1.f + m00 + m11 + m22;
    o.n[0] = o.t; o.n[1] = m12 - m21; o.n[2] = m20 - m02; o.n[3] = m01 - m10;
  }
}

HA_HD void rotmat_to_aa(const float R[9], float aa[3]) {
  QuatBranch b;
  rot_to_quat_branch(R, b);
  const float sc = 0.5f / sqrtf(b.t);
  const float q0 = b.n[0] * sc, q1 = b.n[1] * sc, q2 = b.n[2] * sc, q3 = b.n[3] * sc;
  const float s2 = q1 * q1 + q2 * q2 + q3 * q3;
  const float s = sqrtf(s2);
  const float tt = 2.0f * (q0 < 0.0f ? atan2f(-s, -q0) : atan2f(s, q0));
  const float k = s2 > 0.0f ? tt / s : 2.0f;
  aa[0] = q1 * k; aa[1] = q2 * k; aa[2] = q3 * k;
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (aa[i] != aa[i]) aa[i] = 0.f;
}

HA_HD void rotmat_to_aa_bwd(const float R[9], const float g_aa_in[3], float gR[9]) {
  QuatBranch b;
  rot_to_quat_branch(R, b);
  const float ist = 1.0f / sqrtf(b.t);
  const float sc = 0.5f * ist;
  const float q[4] = {b.n[0] * sc, b.n[1] * sc, b.n[2] * sc, b.n[3] * sc};
  const float s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const float s = sqrtf(s2);
  const float tt = 2.0f * (q[0] < 0.0f ? atan2f(-s, -q[0]) : atan2f(s, q[0]));
  const float k = s2 > 0.0f ? tt / s : 2.0f;
  float g_aa[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float a = q[i + 1] * k;
    g_aa[i] = (a != a) ? 0.f : g_aa_in[i];      // aa[isnan] = 0 cuts the gradient
  }
  float gq[4] = {0.f, g_aa[0] * k, g_aa[1] * k, g_aa[2] * k};
  if (s2 > 0.0f) {
    const float gk = g_aa[0] * q[1] + g_aa[1] * q[2] + g_aa[2] * q[3];
    const float gtt = gk / s;
    float gs = -gk * tt / s2;
    const float den = s2 + q[0] * q[0];
    gs += gtt * 2.0f * q[0] / den;
    gq[0] = -gtt * 2.0f * s / den;
    const float gs2 = gs / (2.0f * s);
    gq[1] += 2.0f * q[1] * gs2; gq[2] += 2.0f * q[2] * gs2; gq[3] += 2.0f * q[3] * gs2;
  }
  // q_i = 0.5 n_i t^-1/2
  float gn[4];
  float gt = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    gn[i] = gq[i] * sc;
    gt += gq[i] * b.n[i];
  }
  gt *= -0.25f * ist * ist * ist;
  // gradient w.r.t. m (= R^T) entries
  float gm[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) gm[i] = 0.f;
  // indices into gm: m_ij -> i*3+j
  if (b.br == 0) {
    gt += gn[1];
    gm[5] += gn[0]; gm[7] -= gn[0];           // m12 - m21
    gm[1] += gn[2]; gm[3] += gn[2];           // m01 + m10
    gm[6] += gn[3]; gm[2] += gn[3];           // m20 + m02
    gm[0] += gt; gm[4] -= gt; gm[8] -= gt;
  } else if (b.br == 1) {
    gt += gn[2];
    gm[6] += gn[0]; gm[2] -= gn[0];           // m20 - m02
    gm[1] += gn[1]; gm[3] += gn[1];           // m01 + m10
    gm[5] += gn[3]; gm[7] += gn[3];           // m12 + m21
    gm[0] -= gt; gm[4] += gt; gm[8] -= gt;
  } else if (b.br == 2) {
    gt += gn[3];
    gm[1] += gn[0]; gm[3] -= gn[0];           // m01 - m10
    gm[6] += gn[1]; gm[2] += gn[1];           // m20 + m02
    gm[5] += gn[2]; gm[7] += gn[2];           // m12 + m21
    gm[0] -= gt; gm[4] -= gt; gm[8] += gt;
  } else {
    gt += gn[0];
    gm[5] += gn[1]; gm[7] -= gn[1];           // m12 - m21
    gm[6] += gn[2]; gm[2] -= gn[2];           // m20 - m02
    gm[1] += gn[3]; gm[3] -= gn[3];           // m01 - m10
    gm[0] += gt; gm[4] += gt; gm[8] += gt;
  }
  // m = R^T
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) gR[j * 3 + i] = gm[i * 3 + j];
}

// ---------------------------------------------------------------------------------------------------
// heading alignment, compute_world2aligned_mat (humor/utils/transforms.py:17-42)
// ---------------------------------------------------------------------------------------------------
struct W2A {
  float W[9];
  float rx, ry, nrm, u, xp, angle, s, az;
};

HA_HD void w2a_fwd(const float pR[9], W2A& o) {
  o.rx = -pR[0];
  o.ry = -pR[3];
  o.nrm = sqrtf(o.rx * o.rx + o.ry * o.ry);
  o.u = o.rx / (o.nrm + 1e-6f);
  o.xp = fminf(fmaxf(o.u, -1.0f), 1.0f);
  o.angle = acosf(o.xp);
  // axis = cross((rx, ry, 0), (1, 0, 0)) = (0, 0, -ry)
  o.s = -o.ry / (fabsf(o.ry) + 1e-6f);
  o.az = o.s * o.angle;
  const float aa[3] = {0.f, 0.f, o.az};
  rodrigues(aa, o.W);
}

// returns gradient w.r.t. pR[0] and pR[3] (the only entries W depends on)
HA_HD void w2a_bwd(const W2A& o, const float gW[9], float& g_p0, float& g_p3) {
  const float aa[3] = {0.f, 0.f, o.az};
  float gaa[3];
  rodrigues_bwd(aa, gW, gaa);
  const float g_az = gaa[2];
  const float g_s = o.angle * g_az;
  const float g_angle = o.s * g_az;
  // d acos(u) / du = -1 / sqrt(1 - u^2).  With u = rx / d, d = nrm + eps:  1 - u^2 = (ry^2 + eps (2 nrm + eps)) / d^2 -- a sum of
  // non-negative terms.  The literal 1 - u u cancels: within 1e-2 rad of the heading singularity (1 - u^2 ~ 1e-4) one rounding of u u is
  // a 6e-4 relative error of the gradient (round 6: the rows every fp32 path missed the tight bars on; the fp32 reference has that error too)
  const float d = o.nrm + 1e-6f;
  const float g_xp = -g_angle * d / sqrtf(o.ry * o.ry + 1e-6f * (2.0f * o.nrm + 1e-6f));
  const float g_u = (o.u >= -1.0f && o.u <= 1.0f) ? g_xp : 0.f;
  float g_rx = g_u / d;
  float g_ry = 0.f;
  const float g_nrm = -g_u * o.rx / (d * d);
  if (o.nrm > 0.f) {
    g_rx += g_nrm * o.rx / o.nrm;
    g_ry += g_nrm * o.ry / o.nrm;
  }
  const float ar = fabsf(o.ry), da = ar + 1e-6f;
  const float sgn = o.ry > 0.f ? 1.f : (o.ry < 0.f ? -1.f : 0.f);
  // s = -ry / (|ry| + 1e-6)
  const float ds = -(da - o.ry * sgn) / (da * da);
  g_ry += g_s * ds;
  g_p0 = -g_rx;
  g_p3 = -g_ry;
}


// ---------------------------------------------------------------------------------------------------
// 6-D rotation representation -> rotation matrix (Zhou et al.), restating rot6d_to_rotmat (humor/utils/transforms.py:201-220):
// x viewed as [3][2]: a1 = column 0 = (x0, x2, x4), a2 = column 1 = (x1, x3, x5); b1 = a1 / max(|a1|, 1e-12) (F.normalize),
// b2 = normalize(a2 - (b1 . a2) b1), b3 = b1 x b2; R = [b1 | b2 | b3] (columns).
// ---------------------------------------------------------------------------------------------------
HA_HD void rot6d_to_rotmat(const float x[6], float R[9]) {
  const float a1[3] = {x[0], x[2], x[4]}, a2[3] = {x[1], x[3], x[5]};
  const float l1 = sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]);
  const float n1 = l1 > 1e-12f ? l1 : 1e-12f;
  const float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
  const float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
  const float v[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
  const float l2 = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  const float n2 = l2 > 1e-12f ? l2 : 1e-12f;
  const float b2[3] = {v[0] / n2, v[1] / n2, v[2] / n2};
  const float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
#pragma unroll
  for (int i = 0; i < 3; ++i) { R[i * 3] = b1[i]; R[i * 3 + 1] = b2[i]; R[i * 3 + 2] = b3[i]; }
}

HA_HD void rot6d_to_rotmat_bwd(const float x[6], const float gR[9], float gx[6]) {
  const float a1[3] = {x[0], x[2], x[4]}, a2[3] = {x[1], x[3], x[5]};
  const float l1 = sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]);
  const bool c1 = l1 > 1e-12f;
  const float n1 = c1 ? l1 : 1e-12f;
  const float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
  const float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
  const float v[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
  const float l2 = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  const bool c2 = l2 > 1e-12f;
  const float n2 = c2 ? l2 : 1e-12f;
  const float b2[3] = {v[0] / n2, v[1] / n2, v[2] / n2};
  float gb1[3] = {gR[0], gR[3], gR[6]}, gb2[3] = {gR[1], gR[4], gR[7]};
  const float gb3[3] = {gR[2], gR[5], gR[8]};
  // b3 = b1 x b2
  gb1[0] += b2[1] * gb3[2] - b2[2] * gb3[1]; gb1[1] += b2[2] * gb3[0] - b2[0] * gb3[2]; gb1[2] += b2[0] * gb3[1] - b2[1] * gb3[0];
  gb2[0] += gb3[1] * b1[2] - gb3[2] * b1[1]; gb2[1] += gb3[2] * b1[0] - gb3[0] * b1[2]; gb2[2] += gb3[0] * b1[1] - gb3[1] * b1[0];
  // b2 = v / max(|v|, eps): the clamped branch is a plain scaling
  const float p2 = c2 ? gb2[0] * b2[0] + gb2[1] * b2[1] + gb2[2] * b2[2] : 0.f;
  const float gv[3] = {(gb2[0] - p2 * b2[0]) / n2, (gb2[1] - p2 * b2[1]) / n2, (gb2[2] - p2 * b2[2]) / n2};
  // v = a2 - d b1, d = b1 . a2
  const float gd = -(gv[0] * b1[0] + gv[1] * b1[1] + gv[2] * b1[2]);
  float ga2[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    ga2[i] = gv[i] + gd * b1[i];
    gb1[i] += -d * gv[i] + gd * a2[i];
  }
  const float p1 = c1 ? gb1[0] * b1[0] + gb1[1] * b1[1] + gb1[2] * b1[2] : 0.f;
  const float ga1[3] = {(gb1[0] - p1 * b1[0]) / n1, (gb1[1] - p1 * b1[1]) / n1, (gb1[2] - p1 * b1[2]) / n1};
  gx[0] = ga1[0]; gx[2] = ga1[1]; gx[4] = ga1[2];
  gx[1] = ga2[0]; gx[3] = ga2[1]; gx[5] = ga2[2];
}

// ---------------------------------------------------------------------------------------------------
// 9-D rotation representation -> rotation matrix (Levinson et al.), restating rot9d_to_rotmat (humor/utils/transforms.py:222-241):
// M = U S V^T (singular values in descending order), R = U diag(1, 1, det(U V^T)) V^T -- the rotation closest to M.
// SVD of the 3x3 by one-sided (Hestenes) Jacobi rotations applied to the columns of M: accurate to fp32 rounding relative to
// every column pair (the eigen-decomposition of M^T M would square the condition number).
// ---------------------------------------------------------------------------------------------------
struct Svd3 {
  float U[9], V[9], s[3], d;     // M = U diag(s) V^T, s descending; d = det(U V^T) = +-1
};

HA_HD void svd3(const float M[9], Svd3& o) {
  float b[3][3], v[3][3];          // b[k] = column k of M V, v[k] = column k of V
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 3; ++i) { b[k][i] = M[i * 3 + k]; v[k][i] = i == k ? 1.f : 0.f; }
  for (int sweep = 0; sweep < 6; ++sweep) {
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
      const float al = b[p][0] * b[p][0] + b[p][1] * b[p][1] + b[p][2] * b[p][2];
      const float be = b[q][0] * b[q][0] + b[q][1] * b[q][1] + b[q][2] * b[q][2];
      const float ga = b[p][0] * b[q][0] + b[p][1] * b[q][1] + b[p][2] * b[q][2];
      if (ga * ga > 1e-30f && ga * ga > 1e-16f * al * be) {
        const float zeta = (be - al) / (2.0f * ga);
        const float t = (zeta >= 0.f ? 1.0f : -1.0f) / (fabsf(zeta) + sqrtf(1.0f + zeta * zeta));
        const float c = 1.0f / sqrtf(1.0f + t * t), sn = c * t;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const float bp = b[p][i], bq = b[q][i], vp = v[p][i], vq = v[q][i];
          b[p][i] = c * bp - sn * bq; b[q][i] = sn * bp + c * bq;
          v[p][i] = c * vp - sn * vq; v[q][i] = sn * vp + c * vq;
        }
      }
    }
  }
  float n[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) n[k] = sqrtf(b[k][0] * b[k][0] + b[k][1] * b[k][1] + b[k][2] * b[k][2]);
  // descending order (three compare-exchanges)
  auto cswap = [&](int x, int y) {
    if (n[x] < n[y]) {
      const float tn = n[x]; n[x] = n[y]; n[y] = tn;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float tb = b[x][i]; b[x][i] = b[y][i]; b[y][i] = tb;
        const float tv = v[x][i]; v[x][i] = v[y][i]; v[y][i] = tv;
      }
    }
  };
  cswap(0, 1); cswap(1, 2); cswap(0, 1);
  float u[3][3];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float inv = 1.0f / (n[k] > 1e-30f ? n[k] : 1e-30f);
#pragma unroll
    for (int i = 0; i < 3; ++i) u[k][i] = b[k][i] * inv;
  }
  const float cx[3] = {u[0][1] * u[1][2] - u[0][2] * u[1][1], u[0][2] * u[1][0] - u[0][0] * u[1][2], u[0][0] * u[1][1] - u[0][1] * u[1][0]};
  if (n[2] > 1e-6f * n[0] && n[2] > 1e-30f) {
#pragma unroll
    for (int i = 0; i < 3; ++i) u[2][i] = b[2][i] / n[2];
  } else {            // rank-deficient: any unit vector completing the basis (R does not depend on its sign)
#pragma unroll
    for (int i = 0; i < 3; ++i) u[2][i] = cx[i];
  }
  const float detU = cx[0] * u[2][0] + cx[1] * u[2][1] + cx[2] * u[2][2];
  const float vx[3] = {v[0][1] * v[1][2] - v[0][2] * v[1][1], v[0][2] * v[1][0] - v[0][0] * v[1][2], v[0][0] * v[1][1] - v[0][1] * v[1][0]};
  const float detV = vx[0] * v[2][0] + vx[1] * v[2][1] + vx[2] * v[2][2];
  o.d = detU * detV >= 0.f ? 1.0f : -1.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    o.s[k] = n[k];
#pragma unroll
    for (int i = 0; i < 3; ++i) { o.U[i * 3 + k] = u[k][i]; o.V[i * 3 + k] = v[k][i]; }
  }
}

HA_HD void rot9d_to_rotmat(const float x[9], float R[9]) {
  Svd3 f;
  svd3(x, f);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      R[i * 3 + j] = f.U[i * 3] * f.V[j * 3] + f.U[i * 3 + 1] * f.V[j * 3 + 1] + f.d * f.U[i * 3 + 2] * f.V[j * 3 + 2];
}

// dL/dM given dL/dR.  With A = U^T dM V the differential is dR = U Y V^T, Y_kk = 0 and for k < l, e = d_k d_l:
//   Y_kl = (A_kl - e A_lk) / (s_k + e s_l),   Y_lk = -e Y_kl
// (e = +1: the smooth polar-factor derivative, finite also for equal singular values -- where the autograd of torch.svd, which the
// reference differentiates through, divides by s_k^2 - s_l^2; e = -1 only for pairs with the flipped third direction).
HA_HD void rot9d_to_rotmat_bwd(const float x[9], const float gR[9], float gx[9]) {
  Svd3 f;
  svd3(x, f);
  float T[9], Gt[9];
  mat3_tmul(f.U, gR, T);           // U^T gR
  mat3_mul(T, f.V, Gt);            // U^T gR V
  const float dd[3] = {1.0f, 1.0f, f.d};
  float gA[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) gA[i] = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int l = k + 1; l < 3; ++l) {
      const float e = dd[k] * dd[l];
      float den = f.s[k] + e * f.s[l];
      den = fabsf(den) > 1e-20f ? den : 1e-20f;
      const float h = (Gt[k * 3 + l] - e * Gt[l * 3 + k]) / den;
      gA[k * 3 + l] = h;
      gA[l * 3 + k] = -e * h;
    }
  mat3_mul(f.U, gA, T);            // U gA
  mat3_mult(T, f.V, gx);           // U gA V^T
}

}  // namespace ha

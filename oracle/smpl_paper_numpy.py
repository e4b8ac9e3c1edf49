"""ORACLE (test infrastructure only).  A SECOND, independent derivation of the SMPL forward pass: fp64 numpy written from the
equations of the SMPL paper (Loper et al., "SMPL: A Skinned Multi-Person Linear Model", SIGGRAPH Asia 2015, eqs. 2-10), NOT from
the smplx source that oracle/lbs_restated.py restates.  Different structure on purpose -- one frame at a time, an explicit
recursion over the kinematic tree, Rodrigues from the closed form with the angle of the plain vector, world transforms
"un-posed" by the rest joints via G_k . [I | -j_k] -- so that an error common to the restatement and the kernels (which were
both written from the smplx op sequence) cannot hide.  tests/test_oracle.py compares the two on the synthetic model.

    T_P(beta, theta) = T_bar + B_S(beta) + B_P(theta)                      (eq. 6)
    J(beta)          = J_reg (T_bar + B_S(beta))                           (eq. 10)
    B_P(theta)       = sum_n (R_n(theta) - R_n(theta*))  P_n               (eq. 9; theta* = rest pose, R = I)
    G_k              = prod_{j in ancestors(k)} [ R_j | j_j - j_parent(j) ]    (eq. 4, world transform of joint k)
    G'_k             = G_k . [ I | -j_k ]                                   (remove the rest pose, eq. 4)
    v_i'             = sum_k w_{k,i} G'_k [ T_P,i ; 1 ]  (+ translation)    (eq. 2 / 7)
"""
import numpy as np


def rodrigues(r):
    """exp(r^) for one axis-angle vector (eq. 1), fp64."""
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    K = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    return np.eye(3) + np.sin(th) * K + (1.0 - np.cos(th)) * (K @ K)


def smpl_frame(model, pose, betas, transl):
    """model: dict with v_template [V,3], shapedirs [V,3,NB], posedirs [V,3,P], J_regressor [J,V], weights [V,J], parents [J].
    pose [J*3] axis-angle, betas [NB], transl [3] -> (vertices [V,3], posed joints [J,3])."""
    vt = np.asarray(model['v_template'], np.float64)
    S = np.asarray(model['shapedirs'], np.float64)[:, :, :len(betas)]
    P = np.asarray(model['posedirs'], np.float64)
    Jr = np.asarray(model['J_regressor'], np.float64)
    W = np.asarray(model['weights'], np.float64)
    parents = [int(p) for p in model['parents']]
    J = Jr.shape[0]
    v_shaped = vt + S @ np.asarray(betas, np.float64)                         # T_bar + B_S(beta)
    j_rest = Jr @ v_shaped                                                     # J(beta)
    R = [rodrigues(np.asarray(pose[3 * k:3 * k + 3], np.float64)) for k in range(J)]
    feat = np.concatenate([(R[k] - np.eye(3)).reshape(9) for k in range(1, J)])   # vec(R_n - I), n >= 1 (root excluded)
    t_posed = v_shaped + P @ feat                                             # + B_P(theta)
    G = [None] * J
    for k in range(J):                                                        # parents precede children in the SMPL tree
        local = np.eye(4)
        local[:3, :3] = R[k]
        local[:3, 3] = j_rest[k] - (j_rest[parents[k]] if k > 0 else 0.0)
        G[k] = local if k == 0 else G[parents[k]] @ local
    joints = np.stack([G[k][:3, 3] for k in range(J)])
    Gp = []
    for k in range(J):
        unpose = np.eye(4)
        unpose[:3, 3] = -j_rest[k]
        Gp.append(G[k] @ unpose)
    Gp = np.stack(Gp)                                                         # [J,4,4]
    T = np.einsum('vk,kab->vab', W, Gp)                                       # per-vertex blended transform
    vh = np.concatenate([t_posed, np.ones((t_posed.shape[0], 1))], axis=1)
    verts = np.einsum('vab,vb->va', T, vh)[:, :3]
    return verts + np.asarray(transl, np.float64), joints + np.asarray(transl, np.float64)

"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement, in plain PyTorch, of the linear-blend-skinning arithmetic the reference delegates to
the third-party package ``smplx`` (pinned ``smplx==0.1.28`` in /root/reference/requirements.txt:11;
NOT vendored in the reference tree and NOT installed in this image).  The reference reaches it through
humor/body_model/body_model.py:7-9 (imports), :61-68 (constructor), :78-91 (forward kwargs).

Restated from the published upstream algorithm (smplx/lbs.py, smplx/body_models.py,
smplx/vertex_joint_selector.py, smplx/vertex_ids.py of release 0.1.28):

* ``batch_rodrigues``       upstream lbs.batch_rodrigues -- identical, line for line, to the copy the
                            reference keeps in humor/utils/transforms.py:139-170 (which pins this piece).
* ``blend_shapes``          einsum('bl,mkl->bmk', betas, shapedirs)
* ``vertices2joints``       einsum('bik,ji->bjk', vertices, J_regressor)
* ``batch_rigid_transform`` 4x4 chain G_i = G_parent(i) @ [R_i | J_i - J_parent(i)], rel = G - [0 | G @ J_i]
* ``lbs``                   v_shaped -> J -> R -> pose_feature @ posedirs -> chain -> dense W @ A -> apply
* ``VertexJointSelector``   21 extra vertex-picked joints for 'smplh' (face 5, feet 6, hands 10)
* ``SMPLHLayer.forward``    concat global|body|lhand|rhand, (+pose_mean = 0), lbs, selector, + transl

PARITY STATUS: "parity unpinned" w.r.t. the smplx package itself (it cannot be imported here and the
reference ships no golden vectors -- SURVEY.md F2/F9).  Pinned instead by (i) batch_rodrigues vs the
reference's own copy, (ii) analytic identities (zero pose/shape => template; rigid root rotation),
(iii) fp64 vs fp32 agreement, (iv) torch.autograd.gradcheck -- see tests/test_oracle.py.
Works in fp32 or fp64 (dtype follows the inputs).
"""
import numpy as np
import torch
import torch.nn.functional as F

# smplx/vertex_ids.py, release 0.1.28, entry 'smplh'
VERTEX_IDS_SMPLH = {
    'nose': 332, 'reye': 6260, 'leye': 2800, 'rear': 4071, 'lear': 583,
    'rthumb': 6191, 'rindex': 5782, 'rmiddle': 5905, 'rring': 6016, 'rpinky': 6133,
    'lthumb': 2746, 'lindex': 2319, 'lmiddle': 2445, 'lring': 2556, 'lpinky': 2673,
    'LBigToe': 3216, 'LSmallToe': 3226, 'LHeel': 3387, 'RBigToe': 6617, 'RSmallToe': 6624, 'RHeel': 6787,
}
# smplx/vertex_ids.py entry 'smplx' (needed only so the reference's `vertex_ids[model_type]` lookup works)
VERTEX_IDS_SMPLX = {
    'nose': 9120, 'reye': 9929, 'leye': 9448, 'rear': 616, 'lear': 6,
    'rthumb': 8079, 'rindex': 7669, 'rmiddle': 7794, 'rring': 7905, 'rpinky': 8022,
    'lthumb': 5361, 'lindex': 4933, 'lmiddle': 5058, 'lring': 5169, 'lpinky': 5286,
    'LBigToe': 5770, 'LSmallToe': 5780, 'LHeel': 8846, 'RBigToe': 8463, 'RSmallToe': 8474, 'RHeel': 8635,
}
VERTEX_IDS = {'smplh': VERTEX_IDS_SMPLH, 'smplx': VERTEX_IDS_SMPLX}


def selector_indices(vertex_ids, use_hands=True, use_feet_keypoints=True):
    """smplx.VertexJointSelector.__init__: face (nose, reye, leye, rear, lear), feet (L big/small/heel,
    R big/small/heel), then finger tips left hand then right hand (thumb, index, middle, ring, pinky)."""
    idxs = [vertex_ids[k] for k in ('nose', 'reye', 'leye', 'rear', 'lear')]
    if use_feet_keypoints:
        idxs += [vertex_ids[k] for k in ('LBigToe', 'LSmallToe', 'LHeel', 'RBigToe', 'RSmallToe', 'RHeel')]
    if use_hands:
        for hand in ('l', 'r'):
            idxs += [vertex_ids[hand + tip] for tip in ('thumb', 'index', 'middle', 'ring', 'pinky')]
    return np.array(idxs, dtype=np.int64)


def batch_rodrigues(rot_vecs):
    """[N,3] -> [N,3,3]; theta = ||r + 1e-8|| (epsilon on every component inside the norm), no Taylor branch."""
    n = rot_vecs.shape[0]
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos = torch.cos(angle).unsqueeze(1)
    sin = torch.sin(angle).unsqueeze(1)
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros((n, 1), dtype=rot_vecs.dtype, device=rot_vecs.device)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view(n, 3, 3)
    ident = torch.eye(3, dtype=rot_vecs.dtype, device=rot_vecs.device).unsqueeze(0)
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


def blend_shapes(betas, shape_disps):
    return torch.einsum('bl,mkl->bmk', betas, shape_disps)


def vertices2joints(J_regressor, vertices):
    return torch.einsum('bik,ji->bjk', vertices, J_regressor)


def transform_mat(R, t):
    return torch.cat([F.pad(R, [0, 0, 0, 1]), F.pad(t, [0, 0, 0, 1], value=1)], dim=2)


def batch_rigid_transform(rot_mats, joints, parents):
    joints = joints.unsqueeze(-1)
    rel_joints = joints.clone()
    rel_joints[:, 1:] = rel_joints[:, 1:] - joints[:, parents[1:]]
    transforms_mat = transform_mat(rot_mats.reshape(-1, 3, 3), rel_joints.reshape(-1, 3, 1)).reshape(
        -1, joints.shape[1], 4, 4)
    chain = [transforms_mat[:, 0]]
    for i in range(1, parents.shape[0]):
        chain.append(torch.matmul(chain[int(parents[i])], transforms_mat[:, i]))
    transforms = torch.stack(chain, dim=1)
    posed_joints = transforms[:, :, :3, 3]
    joints_homogen = F.pad(joints, [0, 0, 0, 1])
    rel_transforms = transforms - F.pad(torch.matmul(transforms, joints_homogen), [3, 0, 0, 0, 0, 0, 0, 0])
    return posed_joints, rel_transforms


def lbs(betas, pose, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights, return_aux=False):
    """posedirs is [P, V*3] (upstream buffer layout); pose is axis-angle [N, J*3]."""
    n = max(betas.shape[0], pose.shape[0])
    dtype, device = betas.dtype, betas.device
    v_shaped = v_template + blend_shapes(betas, shapedirs)
    J = vertices2joints(J_regressor, v_shaped)
    ident = torch.eye(3, dtype=dtype, device=device)
    rot_mats = batch_rodrigues(pose.reshape(-1, 3)).view(n, -1, 3, 3)
    pose_feature = (rot_mats[:, 1:, :, :] - ident).view(n, -1)
    pose_offsets = torch.matmul(pose_feature, posedirs).view(n, -1, 3)
    v_posed = pose_offsets + v_shaped
    J_transformed, A = batch_rigid_transform(rot_mats, J, parents)
    W = lbs_weights.unsqueeze(0).expand(n, -1, -1)
    num_joints = J_regressor.shape[0]
    T = torch.matmul(W, A.view(n, num_joints, 16)).view(n, -1, 4, 4)
    homogen = torch.ones(n, v_posed.shape[1], 1, dtype=dtype, device=device)
    v_homo = torch.matmul(T, torch.cat([v_posed, homogen], dim=2).unsqueeze(-1))
    verts = v_homo[:, :, :3, 0]
    if return_aux:
        return verts, J_transformed, dict(v_shaped=v_shaped, J=J, rot_mats=rot_mats, pose_feature=pose_feature,
                                          v_posed=v_posed, A=A)
    return verts, J_transformed


class ModelOutput:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class SMPLHLayer(torch.nn.Module):
    """Restatement of smplx.SMPLH(model_path, data_struct=..., num_betas, batch_size, vertex_ids,
    use_pca=False, flat_hand_mean=True) as the reference constructs it (body_model.py:49-64)."""
    NUM_JOINTS = 51          # smplx.SMPLH.NUM_JOINTS (body 21 + 2*15 hand), root excluded
    NUM_BODY_JOINTS = 21
    NUM_HAND_JOINTS = 15
    SHAPE_SPACE_DIM = 300

    def __init__(self, model_path=None, data_struct=None, num_betas=10, batch_size=1, vertex_ids=None,
                 dtype=torch.float32, **unused):
        super().__init__()
        ds = data_struct
        self.batch_size = batch_size
        self.dtype = dtype
        shapedirs = np.asarray(ds.shapedirs)[:, :, :num_betas]
        self.num_betas = num_betas
        t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), dtype=dtype)
        self.register_buffer('shapedirs', t(shapedirs))
        self.register_buffer('v_template', t(ds.v_template))
        self.register_buffer('J_regressor', t(ds.J_regressor))
        posedirs = np.asarray(ds.posedirs)
        self.register_buffer('posedirs', t(np.reshape(posedirs, [-1, posedirs.shape[-1]]).T))
        parents = torch.tensor(np.asarray(ds.kintree_table)[0].astype(np.int64))
        parents[0] = -1
        self.register_buffer('parents', parents)
        self.register_buffer('lbs_weights', t(ds.weights))
        self.register_buffer('faces_tensor', torch.tensor(np.asarray(ds.f).astype(np.int64)))
        self.register_buffer('extra_joints_idxs',
                             torch.tensor(selector_indices(vertex_ids)) if vertex_ids is not None else None)
        # default (zero) parameters baked to batch_size -- the reason the reference needs N == batch_size
        z = lambda d: torch.nn.Parameter(torch.zeros(batch_size, d, dtype=dtype))
        self.betas, self.global_orient, self.body_pose = z(num_betas), z(3), z(63)
        self.left_hand_pose, self.right_hand_pose, self.transl = z(45), z(45), z(3)

    def forward(self, betas=None, global_orient=None, body_pose=None, left_hand_pose=None, right_hand_pose=None,
                transl=None, return_full_pose=False, **unused):
        global_orient = self.global_orient if global_orient is None else global_orient
        body_pose = self.body_pose if body_pose is None else body_pose
        betas = self.betas if betas is None else betas
        left_hand_pose = self.left_hand_pose if left_hand_pose is None else left_hand_pose
        right_hand_pose = self.right_hand_pose if right_hand_pose is None else right_hand_pose
        transl = self.transl if transl is None else transl
        full_pose = torch.cat([global_orient, body_pose, left_hand_pose, right_hand_pose], dim=1)
        verts, joints = lbs(betas, full_pose, self.v_template, self.shapedirs, self.posedirs, self.J_regressor,
                            self.parents, self.lbs_weights)
        if self.extra_joints_idxs is not None:
            joints = torch.cat([joints, torch.index_select(verts, 1, self.extra_joints_idxs)], dim=1)
        joints = joints + transl.unsqueeze(1)
        verts = verts + transl.unsqueeze(1)
        return ModelOutput(vertices=verts, joints=joints, betas=betas, global_orient=global_orient,
                           body_pose=body_pose, left_hand_pose=left_hand_pose, right_hand_pose=right_hand_pose,
                           full_pose=full_pose if return_full_pose else None)

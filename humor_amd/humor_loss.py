"""HumorLoss -- the training-time loss of the HuMoR CVAE with its SMPL terms on the MI355X body-model kernels.

Interface of humor/losses/humor_loss.py:17-392 (constructor keywords, ``forward(pred_dict, gt_dict, cur_epoch, gender, betas)``
-> ``(loss, stats_dict)`` with the same ``stats_dict`` keys), SURVEY.md 8(f4): "training path (HumorLoss SMPL terms reuse the
LBS kernels)".  The KL / regression / contact terms are a handful of element-wise reductions and stay PyTorch; what costs time
in a training step is the SMPL reconstruction (`humor_loss.py:228-345`):

* rotation matrices -> axis-angle for root + 21 joints of prediction AND ground truth  (`ha_rotmat_to_aa_fwd/bwd`);
* per gender, the reference evaluates the body model twice (prediction, ground truth), each zero-padded to `smpl_batch_size`
  rows (`humor_loss.py:263-286`).  Here each gender is ONE dense call on `[prediction rows ; ground-truth rows]` of exactly the
  rows that exist (the kernels take any N; padded rows would be computed and thrown away), i.e. the pose-blend MFMA GEMM and the
  HBM-bound skinning kernel see 2n rows at once;
* the mesh term differentiates through all 6890 vertices: its gradient takes the dense SMPL adjoint
  (`ha_smpl_backward_dense`: streaming dL/dv_posed + two MFMA kernels), the joint / key-vertex terms ride the same call.

GPU tensors only for the SMPL terms (BodyModel has no CPU fallback).  `smpl_batch_size` keeps the reference's meaning as an
upper bound: more rows of one gender than that raise the reference's exception.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .body_model import BodyModel
from .ops import rotation_matrix_to_angle_axis
from .tables import CONTACT_INDS, KEYPT_VERTS, NUM_BODY_JOINTS, SMPL_JOINTS

BETA_SIZE = 16
CONTACT_THRESH = 0.5
# humor/body_model/utils.py:3 (SMPLH_PATH = './body_models/smplh'): where <gender>/model.npz is looked up unless smplh_path is given
SMPLH_PATH = './body_models/smplh'


class HumorLoss(nn.Module):

    def __init__(self,
                 kl_loss=1.0,
                 kl_loss_anneal_start=0,
                 kl_loss_anneal_end=0,
                 kl_loss_cycle_len=-1,
                 regr_trans_loss=1.0,
                 regr_trans_vel_loss=1.0,
                 regr_root_orient_loss=1.0,
                 regr_root_orient_vel_loss=1.0,
                 regr_pose_loss=1.0,
                 regr_pose_vel_loss=1.0,
                 regr_joint_loss=1.0,
                 regr_joint_vel_loss=1.0,
                 regr_joint_orient_vel_loss=1.0,
                 regr_vert_loss=1.0,
                 regr_vert_vel_loss=1.0,
                 contacts_loss=0.0,
                 contacts_vel_loss=0.0,
                 smpl_joint_loss=0.0,
                 smpl_mesh_loss=0.0,
                 smpl_joint_consistency_loss=0.0,
                 smpl_vert_consistency_loss=0.0,
                 smpl_batch_size=480,
                 smplh_path=None, _lib_override=None):
        super(HumorLoss, self).__init__()
        self.kl_loss_weight = kl_loss
        self.kl_loss_anneal_start = kl_loss_anneal_start
        self.kl_loss_anneal_end = kl_loss_anneal_end
        self.use_kl_anneal = self.kl_loss_anneal_end > self.kl_loss_anneal_start
        self.kl_loss_cycle_len = kl_loss_cycle_len
        self.use_kl_cycle = False
        if self.kl_loss_cycle_len > 0:
            self.use_kl_cycle = True
            self.use_kl_anneal = False

        self.contacts_loss_weight = contacts_loss
        self.contacts_vel_loss_weight = contacts_vel_loss
        self.bce_loss = nn.BCEWithLogitsLoss(reduction='none')

        # keys are the ones of the pred / gt dictionaries (humor_loss.py:72-84)
        self.regr_loss_weight_dict = {
            'trans': regr_trans_loss,
            'trans_vel': regr_trans_vel_loss,
            'root_orient': regr_root_orient_loss,
            'root_orient_vel': regr_root_orient_vel_loss,
            'pose_body': regr_pose_loss,
            'pose_body_vel': regr_pose_vel_loss,
            'joints': regr_joint_loss,
            'joints_vel': regr_joint_vel_loss,
            'joints_orient_vel': regr_joint_orient_vel_loss,
            'verts': regr_vert_loss,
            'verts_vel': regr_vert_vel_loss
        }

        self.smpl_joint_loss_weight = smpl_joint_loss
        self.smpl_mesh_loss_weight = smpl_mesh_loss
        self.smpl_joint_consistency_loss_weight = smpl_joint_consistency_loss
        self.smpl_vert_consistency_loss_weight = smpl_vert_consistency_loss

        self.l2_loss = nn.MSELoss(reduction='none')
        self.regr_loss = nn.MSELoss(reduction='none')

        smpl_losses = [self.smpl_joint_loss_weight, self.smpl_mesh_loss_weight, self.smpl_joint_consistency_loss_weight,
                       self.smpl_vert_consistency_loss_weight]
        self.smpl_batch_size = smpl_batch_size
        self.use_smpl_losses = False
        self._lib = _lib_override
        if sum(smpl_losses) > 0.0:
            self.use_smpl_losses = True
            root = SMPLH_PATH if smplh_path is None else smplh_path
            # only the mesh term reads all 6890 vertices; without it the body model runs on the 43 key vertices + joints (the
            # wave-per-frame subset kernels) and returns exactly the key vertices as `v`
            self._dense = self.smpl_mesh_loss_weight > 0.0
            subset = None if self._dense else list(KEYPT_VERTS)
            self.male_bm = BodyModel(bm_path=os.path.join(root, 'male/model.npz'), num_betas=BETA_SIZE,
                                     batch_size=self.smpl_batch_size, vertex_subset=subset, _lib_override=_lib_override)
            self.female_bm = BodyModel(bm_path=os.path.join(root, 'female/model.npz'), num_betas=BETA_SIZE,
                                       batch_size=self.smpl_batch_size, vertex_subset=subset, _lib_override=_lib_override)

    # ------------------------------------------------------------------------------------------------
    def forward(self, pred_dict, gt_dict, cur_epoch, gender=None, betas=None):
        '''
        All data in the dictionaries is B x D (humor_loss.py:106-116).
        '''
        loss = 0.0
        stats_dict = dict()

        # KL divergence between posterior and (learned) prior, with linear / cyclic annealing (humor_loss.py:119-149)
        if self.kl_loss_weight > 0.0:
            qm, qv = pred_dict['posterior_distrib']
            pm, pv = pred_dict['prior_distrib']
            kl_loss = self.kl_normal(qm, qv, pm, pv).mean()
            stats_dict['kl_loss'] = kl_loss
            anneal_weight = 1.0
            if self.use_kl_anneal or self.use_kl_cycle:
                anneal_epoch = cur_epoch
                anneal_start = self.kl_loss_anneal_start
                anneal_end = self.kl_loss_anneal_end
                if self.use_kl_cycle:
                    anneal_epoch = cur_epoch % self.kl_loss_cycle_len
                    anneal_start = 0
                    anneal_end = self.kl_loss_cycle_len // 2
                if anneal_epoch >= anneal_start:
                    anneal_weight = (anneal_epoch - anneal_start) / (anneal_end - anneal_start)
                else:
                    anneal_weight = 0.0
                anneal_weight = 1.0 if anneal_weight > 1.0 else anneal_weight
            loss = loss + anneal_weight * self.kl_loss_weight * kl_loss
            stats_dict['kl_anneal_weight'] = anneal_weight
            stats_dict['kl_weighted_loss'] = loss

        # regression terms (humor_loss.py:155-175)
        for cur_key in gt_dict.keys():
            if cur_key not in self.regr_loss_weight_dict:
                continue
            cur_regr_weight = self.regr_loss_weight_dict[cur_key]
            if cur_regr_weight > 0.0:
                cur = self.regr_loss(pred_dict[cur_key], gt_dict[cur_key]).mean()
                stats_dict[cur_key + '_loss'] = cur
                loss = loss + cur_regr_weight * cur

        # contact classification + its accuracy statistics (humor_loss.py:177-211)
        if self.contacts_loss_weight > 0.0:
            if 'contacts' in gt_dict.keys() and 'contacts' in pred_dict.keys():
                gt_contacts = gt_dict['contacts']
                pred_contacts = pred_dict['contacts']
                cur = self.bce_loss(pred_contacts, gt_contacts).mean()
                stats_dict['contacts_loss'] = cur
                loss = loss + self.contacts_loss_weight * cur

                pred_c = (torch.sigmoid(pred_contacts) > CONTACT_THRESH).to(torch.bool)
                gt_c = gt_contacts.to(torch.bool)
                true_pos_cnt = torch.sum(pred_c & gt_c).to(torch.float)
                false_pos_cnt = torch.sum(pred_c & ~gt_c).to(torch.float)
                false_neg_cnt = torch.sum(~pred_c & gt_c).to(torch.float)
                true_neg_cnt = torch.sum(~pred_c & ~gt_c).to(torch.float)
                stats_dict['contacts_acc'] = (true_pos_cnt + true_neg_cnt) / (true_pos_cnt + false_pos_cnt + false_neg_cnt + true_neg_cnt)
                stats_dict['contacts_pos_acc'] = true_pos_cnt / (true_pos_cnt + false_neg_cnt)
                stats_dict['contacts_neg_acc'] = true_neg_cnt / (true_neg_cnt + false_pos_cnt)
            else:
                print('Cannot compute contact loss without contact pred/gt! Skipping...')

        # joint velocity near zero where contact is predicted (humor_loss.py:214-226)
        if self.contacts_vel_loss_weight > 0.0:
            if 'contacts' in pred_dict.keys() and 'joints_vel' in pred_dict.keys():
                pred_contacts = torch.sigmoid(pred_dict['contacts'])
                pred_joints_vel = pred_dict['joints_vel'].reshape((-1, len(SMPL_JOINTS), 3))
                vel_mag = torch.norm(pred_joints_vel[:, CONTACT_INDS, :], dim=-1)
                cur = (pred_contacts * (vel_mag ** 2)).mean()
                stats_dict['contacts_vel_loss'] = cur
                loss = loss + self.contacts_vel_loss_weight * cur
            else:
                print('Cannot compute contact vel loss without contact and joints_vel pred! Skipping...')

        # terms requiring SMPL reconstruction (humor_loss.py:228-345)
        if self.use_smpl_losses:
            if gender is None or betas is None:
                raise Exception('Must pass gender and betas to MotionVAE loss to use SMPL losses!')
            try:
                pred_trans, pred_orient, pred_pose = pred_dict['trans'], pred_dict['root_orient'], pred_dict['pose_body']
                gt_trans, gt_orient, gt_pose = gt_dict['trans'], gt_dict['root_orient'], gt_dict['pose_body']
            except KeyError:
                raise KeyError('In order to use SMPL losses must have trans, root_orient, and pose_body in pred and gt dicts!')

            B = pred_trans.size(0)
            nj = len(SMPL_JOINTS)
            # rotation matrices -> axis-angle, prediction and ground truth in ONE kernel launch each way
            R_all = torch.cat([pred_orient.reshape(B, 1, 9), pred_pose.reshape(B, NUM_BODY_JOINTS, 9),
                               gt_orient.reshape(B, 1, 9), gt_pose.reshape(B, NUM_BODY_JOINTS, 9)], dim=1)
            aa_all = rotation_matrix_to_angle_axis(R_all.reshape(-1, 3, 3), self._lib).reshape(B, 2 * nj * 3)
            pred_aa, gt_aa = aa_all[:, :nj * 3], aa_all[:, nj * 3:]

            # split by gender (the two body models differ); the reference's concatenation order is male rows, then female rows
            gender = np.asarray(gender)
            order, pieces = [], []
            for gender_name, bm in (('male', self.male_bm), ('female', self.female_bm)):
                idx = np.nonzero(gender[:, 0] == gender_name)[0]
                if idx.size == 0:
                    continue
                if idx.size > self.smpl_batch_size:
                    raise Exception('SMPL model batch size not large enough to accomodate!')
                it = torch.from_numpy(idx).to(pred_trans.device)
                order.append(it)
                n = idx.size
                aa = torch.cat([pred_aa.index_select(0, it), gt_aa.index_select(0, it)], dim=0)
                tr = torch.cat([pred_trans.index_select(0, it), gt_trans.index_select(0, it)], dim=0)
                be = betas.index_select(0, it)
                body = bm(pose_body=aa[:, 3:], betas=torch.cat([be, be], dim=0), root_orient=aa[:, :3], trans=tr)
                pieces.append((body.Jtr[:n, :nj], body.Jtr[n:, :nj], body.v[:n], body.v[n:]))
            order = torch.cat(order, dim=0)
            pred_joints = torch.cat([p[0] for p in pieces], dim=0)
            gt_joints = torch.cat([p[1] for p in pieces], dim=0)
            pred_mesh = torch.cat([p[2] for p in pieces], dim=0)
            gt_mesh = torch.cat([p[3] for p in pieces], dim=0)
            if self._dense:
                pred_verts, gt_verts = pred_mesh[:, KEYPT_VERTS, :], gt_mesh[:, KEYPT_VERTS, :]
            else:
                pred_verts, gt_verts = pred_mesh, gt_mesh        # the subset body returns exactly the key vertices

            if self.smpl_joint_loss_weight > 0.0:
                cur = self.regr_loss(pred_joints, gt_joints).mean()
                stats_dict['smpl_joint_loss'] = cur
                loss = loss + self.smpl_joint_loss_weight * cur
            if self.smpl_mesh_loss_weight > 0.0:
                cur = self.regr_loss(pred_mesh, gt_mesh).mean()
                stats_dict['smpl_mesh_loss'] = cur
                loss = loss + self.smpl_mesh_loss_weight * cur
            if self.smpl_joint_consistency_loss_weight > 0.0:
                if 'joints' not in pred_dict.keys():
                    raise KeyError('Must regress joints in order to use smpl joint consistency loss!')
                regressed_joints = pred_dict['joints'].reshape((B, nj, -1)).index_select(0, order)
                cur = self.regr_loss(pred_joints, regressed_joints).mean()
                stats_dict['smpl_joint_consistency_loss'] = cur
                loss = loss + self.smpl_joint_consistency_loss_weight * cur
            if self.smpl_vert_consistency_loss_weight > 0.0:
                if 'verts' not in pred_dict.keys():
                    raise KeyError('Must regress verts in order to use smpl vert consistency loss!')
                regressed_verts = pred_dict['verts'].reshape((B, len(KEYPT_VERTS), -1)).index_select(0, order)
                cur = self.regr_loss(pred_verts, regressed_verts).mean()
                stats_dict['smpl_vert_consistency_loss'] = cur
                loss = loss + self.smpl_vert_consistency_loss_weight * cur

        if self.kl_loss_weight > 0.0:
            stats_dict['reconstr_weighted_loss'] = loss - stats_dict['kl_weighted_loss']

        return loss, stats_dict

    def zero_pad_tensors(self, pad_list, pad_size):
        '''B x D tensors padded with zero rows (humor_loss.py:352-361); kept for callers, unused by forward.'''
        return [torch.cat([t, torch.zeros((pad_size, t.size(1))).to(t)], dim=0) for t in pad_list]

    def kl_normal(self, qm, qv, pm, pv):
        """KL(q || p) of diagonal normals given means / variances, summed over the last dim (humor_loss.py:364-378)."""
        element_wise = 0.5 * (torch.log(pv) - torch.log(qv) + qv / pv + (qm - pm).pow(2) / pv - 1)
        return element_wise.sum(-1)

    def log_normal(self, x, m, v):
        """log N(x; m, v) summed over the last dim (humor_loss.py:380-392)."""
        log_prob = -torch.log(torch.sqrt(v)) - np.log(np.sqrt(2 * np.pi)) - ((x - m) ** 2 / (2 * v))
        return torch.sum(log_prob, dim=-1)

"""HumorLoss -- the training-time loss of the HuMoR CVAE with its SMPL terms on the MI355X body-model kernels.

Interface of humor/losses/humor_loss.py:17-392: the same constructor keywords (the reference builds it as
``HumorLoss(**loss_dict, smpl_batch_size=...)``, train/train_humor.py:66), ``forward(pred_dict, gt_dict, cur_epoch, gender, betas)`` ->
``(loss, stats_dict)`` with the same ``stats_dict`` keys and the same arithmetic per term ("training path: HumorLoss
SMPL terms reuse the LBS kernels").  The KL / regression / contact terms are a handful of element-wise reductions and stay PyTorch
(they are pinned bit-for-bit to the live reference in the CPU test tier); what costs time in a training step is the SMPL
reconstruction (`humor_loss.py:228-345`), restructured here:

* rotation matrices -> axis-angle for root + 21 joints of prediction AND ground truth in one launch each way (`ha_rotmat_to_aa_*`);
* per gender the reference evaluates the body model twice (prediction, ground truth), each zero-padded to `smpl_batch_size` rows
  (`humor_loss.py:263-286`).  Here each gender is ONE call on `[prediction rows ; ground-truth rows]` of exactly the rows that exist
  (the kernels take any N; padded rows would be computed and thrown away): the pose-blend MFMA GEMM and the HBM-bound skinning kernel
  see 2n rows at once;
* the mesh term differentiates through all 6890 vertices: its gradient takes the dense SMPL adjoint (`ha_smpl_backward_dense`:
  streaming dL/dv_posed + two MFMA kernels); without the mesh term the body runs on the 43 key vertices + joints only.

GPU tensors only for the SMPL terms (BodyModel has no CPU fallback).  `smpl_batch_size` keeps the reference's meaning as an upper
bound: more rows of one gender than that raise the reference's exception.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .body_model import BodyModel
from .ops import rotation_matrix_to_angle_axis
from .tables import CONTACT_INDS, KEYPT_VERTS, NUM_BODY_JOINTS, SMPL_JOINTS

BETA_SIZE = 16
CONTACT_THRESH = 0.5
# humor/body_model/utils.py:3: where <gender>/model.npz is looked up unless smplh_path is given
SMPLH_PATH = './body_models/smplh'
NJ = len(SMPL_JOINTS)

# constructor keyword -> default (humor_loss.py:19-41); every value is the weight of its term, 0 switches the term off
_DEFAULTS = {
    'kl_loss': 1.0, 'kl_loss_anneal_start': 0, 'kl_loss_anneal_end': 0, 'kl_loss_cycle_len': -1,
    'regr_trans_loss': 1.0, 'regr_trans_vel_loss': 1.0, 'regr_root_orient_loss': 1.0, 'regr_root_orient_vel_loss': 1.0,
    'regr_pose_loss': 1.0, 'regr_pose_vel_loss': 1.0, 'regr_joint_loss': 1.0, 'regr_joint_vel_loss': 1.0,
    'regr_joint_orient_vel_loss': 1.0, 'regr_vert_loss': 1.0, 'regr_vert_vel_loss': 1.0,
    'contacts_loss': 0.0, 'contacts_vel_loss': 0.0,
    'smpl_joint_loss': 0.0, 'smpl_mesh_loss': 0.0, 'smpl_joint_consistency_loss': 0.0, 'smpl_vert_consistency_loss': 0.0,
}
# key of the pred / gt dictionaries -> constructor keyword of its regression weight (humor_loss.py:72-84)
_REGRESSION = (('trans', 'regr_trans_loss'), ('trans_vel', 'regr_trans_vel_loss'), ('root_orient', 'regr_root_orient_loss'),
               ('root_orient_vel', 'regr_root_orient_vel_loss'), ('pose_body', 'regr_pose_loss'), ('pose_body_vel', 'regr_pose_vel_loss'),
               ('joints', 'regr_joint_loss'), ('joints_vel', 'regr_joint_vel_loss'), ('joints_orient_vel', 'regr_joint_orient_vel_loss'),
               ('verts', 'regr_vert_loss'), ('verts_vel', 'regr_vert_vel_loss'))


def _mse(a, b):
    return F.mse_loss(a, b, reduction='none').mean()


class HumorLoss(nn.Module):

    def __init__(self, smpl_batch_size=480, smplh_path=None, _lib_override=None, **weights):
        super(HumorLoss, self).__init__()
        unknown = sorted(set(weights) - set(_DEFAULTS))
        if unknown:
            raise TypeError('HumorLoss got unexpected keyword argument(s): %s' % ', '.join(unknown))
        w = dict(_DEFAULTS, **weights)
        self.kl_loss_weight = w['kl_loss']
        self.kl_loss_anneal_start, self.kl_loss_anneal_end = w['kl_loss_anneal_start'], w['kl_loss_anneal_end']
        self.kl_loss_cycle_len = w['kl_loss_cycle_len']
        self.use_kl_cycle = self.kl_loss_cycle_len > 0                   # cyclic annealing takes precedence over the linear ramp
        self.use_kl_anneal = (not self.use_kl_cycle) and self.kl_loss_anneal_end > self.kl_loss_anneal_start
        self.contacts_loss_weight, self.contacts_vel_loss_weight = w['contacts_loss'], w['contacts_vel_loss']
        self.regr_loss_weight_dict = {key: w[kw] for key, kw in _REGRESSION}
        self.smpl_joint_loss_weight, self.smpl_mesh_loss_weight = w['smpl_joint_loss'], w['smpl_mesh_loss']
        self.smpl_joint_consistency_loss_weight = w['smpl_joint_consistency_loss']
        self.smpl_vert_consistency_loss_weight = w['smpl_vert_consistency_loss']
        self.smpl_batch_size = smpl_batch_size
        self._lib = _lib_override
        self.use_smpl_losses = (self.smpl_joint_loss_weight + self.smpl_mesh_loss_weight + self.smpl_joint_consistency_loss_weight +
                                self.smpl_vert_consistency_loss_weight) > 0.0
        if self.use_smpl_losses:
            root = SMPLH_PATH if smplh_path is None else smplh_path
            # only the mesh term reads all 6890 vertices; without it the body model runs on the 43 key vertices + joints (the
            # wave-per-frame subset kernels) and returns exactly the key vertices as `v`
            self._dense = self.smpl_mesh_loss_weight > 0.0
            subset = None if self._dense else list(KEYPT_VERTS)
            make = lambda g: BodyModel(bm_path=os.path.join(root, g, 'model.npz'), num_betas=BETA_SIZE, batch_size=smpl_batch_size,
                                       vertex_subset=subset, _lib_override=_lib_override)
            self.male_bm, self.female_bm = make('male'), make('female')

    # ---- the terms ---------------------------------------------------------------------------------------------------------------
    def _kl_anneal_weight(self, cur_epoch):
        """Linear ramp between the two epochs, or a cycle whose first half ramps and second half holds (humor_loss.py:129-143)."""
        if not (self.use_kl_anneal or self.use_kl_cycle):
            return 1.0
        if self.use_kl_cycle:
            epoch, start, end = cur_epoch % self.kl_loss_cycle_len, 0, self.kl_loss_cycle_len // 2
        else:
            epoch, start, end = cur_epoch, self.kl_loss_anneal_start, self.kl_loss_anneal_end
        if epoch < start:
            return 0.0
        ramp = (epoch - start) / (end - start)
        return 1.0 if ramp > 1.0 else ramp

    def _contact_terms(self, pred_dict, gt_dict, stats):
        """BCE on the contact logits + confusion-matrix accuracies; predicted-contact-weighted joint speed (humor_loss.py:177-226)."""
        total = 0.0
        if self.contacts_loss_weight > 0.0:
            if 'contacts' in gt_dict and 'contacts' in pred_dict:
                logits, labels = pred_dict['contacts'], gt_dict['contacts']
                bce = F.binary_cross_entropy_with_logits(logits, labels, reduction='none').mean()
                stats['contacts_loss'] = bce
                total = total + self.contacts_loss_weight * bce
                hit, truth = torch.sigmoid(logits) > CONTACT_THRESH, labels.to(torch.bool)
                tp, fp = (hit & truth).sum().float(), (hit & ~truth).sum().float()
                fn, tn = (~hit & truth).sum().float(), (~hit & ~truth).sum().float()
                stats['contacts_acc'] = (tp + tn) / (tp + fp + fn + tn)
                stats['contacts_pos_acc'] = tp / (tp + fn)
                stats['contacts_neg_acc'] = tn / (tn + fp)
            else:
                print('Cannot compute contact loss without contact pred/gt! Skipping...')
        if self.contacts_vel_loss_weight > 0.0:
            if 'contacts' in pred_dict and 'joints_vel' in pred_dict:
                speed = torch.norm(pred_dict['joints_vel'].reshape((-1, NJ, 3))[:, CONTACT_INDS, :], dim=-1)
                cv = (torch.sigmoid(pred_dict['contacts']) * (speed ** 2)).mean()
                stats['contacts_vel_loss'] = cv
                total = total + self.contacts_vel_loss_weight * cv
            else:
                print('Cannot compute contact vel loss without contact and joints_vel pred! Skipping...')
        return total

    def _smpl_bodies(self, pred_dict, gt_dict, gender, betas):
        """Joints and vertices of the predicted and the ground-truth bodies, rows in the reference's order (all male rows, then all
        female rows), plus that row order."""
        try:
            pred_trans, pred_orient, pred_pose = pred_dict['trans'], pred_dict['root_orient'], pred_dict['pose_body']
            gt_trans, gt_orient, gt_pose = gt_dict['trans'], gt_dict['root_orient'], gt_dict['pose_body']
        except KeyError:
            raise KeyError('In order to use SMPL losses must have trans, root_orient, and pose_body in pred and gt dicts!')
        B = pred_trans.size(0)
        # rotation matrices -> axis-angle, prediction and ground truth in ONE kernel launch each way
        R_all = torch.cat([pred_orient.reshape(B, 1, 9), pred_pose.reshape(B, NUM_BODY_JOINTS, 9),
                           gt_orient.reshape(B, 1, 9), gt_pose.reshape(B, NUM_BODY_JOINTS, 9)], dim=1)
        aa_all = rotation_matrix_to_angle_axis(R_all.reshape(-1, 3, 3), self._lib).reshape(B, 2 * NJ * 3)
        pred_aa, gt_aa = aa_all[:, :NJ * 3], aa_all[:, NJ * 3:]
        gender = np.asarray(gender)
        order, pieces = [], []
        for name, bm in (('male', self.male_bm), ('female', self.female_bm)):
            idx = np.nonzero(gender[:, 0] == name)[0]
            if idx.size == 0:
                continue
            if idx.size > self.smpl_batch_size:
                raise Exception('SMPL model batch size not large enough to accomodate!')
            rows = torch.from_numpy(idx).to(pred_trans.device)
            order.append(rows)
            n = idx.size
            aa = torch.cat([pred_aa.index_select(0, rows), gt_aa.index_select(0, rows)], dim=0)
            tr = torch.cat([pred_trans.index_select(0, rows), gt_trans.index_select(0, rows)], dim=0)
            be = betas.index_select(0, rows)
            body = bm(pose_body=aa[:, 3:], betas=torch.cat([be, be], dim=0), root_orient=aa[:, :3], trans=tr)
            pieces.append((body.Jtr[:n, :NJ], body.Jtr[n:, :NJ], body.v[:n], body.v[n:]))
        stack = lambda i: torch.cat([p[i] for p in pieces], dim=0)
        return stack(0), stack(1), stack(2), stack(3), torch.cat(order, dim=0)

    # ------------------------------------------------------------------------------------------------------------------------------
    def forward(self, pred_dict, gt_dict, cur_epoch, gender=None, betas=None):
        '''
        All data in the dictionaries is B x D (humor_loss.py:106-116).
        '''
        loss = 0.0
        stats_dict = dict()

        # KL(posterior || prior) with its annealing weight
        if self.kl_loss_weight > 0.0:
            kl = self.kl_normal(*pred_dict['posterior_distrib'], *pred_dict['prior_distrib']).mean()
            anneal = self._kl_anneal_weight(cur_epoch)
            loss = loss + anneal * self.kl_loss_weight * kl
            stats_dict.update(kl_loss=kl, kl_anneal_weight=anneal, kl_weighted_loss=loss)

        # mean-squared regression of every state component present in the ground truth (humor_loss.py:155-175)
        for key in gt_dict.keys():
            weight = self.regr_loss_weight_dict.get(key, 0.0)
            if weight > 0.0:
                term = _mse(pred_dict[key], gt_dict[key])
                stats_dict[key + '_loss'] = term
                loss = loss + weight * term

        loss = loss + self._contact_terms(pred_dict, gt_dict, stats_dict)

        # terms that need the SMPL bodies (humor_loss.py:228-345)
        if self.use_smpl_losses:
            if gender is None or betas is None:
                raise Exception('Must pass gender and betas to MotionVAE loss to use SMPL losses!')
            pred_joints, gt_joints, pred_mesh, gt_mesh, order = self._smpl_bodies(pred_dict, gt_dict, gender, betas)
            B = pred_dict['trans'].size(0)           # the reference's reshape size (humor_loss.py:236); `order` then picks the gendered rows
            if order.numel() != B:
                raise ValueError('HumorLoss SMPL terms: every row must be male or female (got %d of %d rows)' % (order.numel(), B))
            smpl_terms = []
            if self.smpl_joint_loss_weight > 0.0:
                smpl_terms.append(('smpl_joint_loss', self.smpl_joint_loss_weight, _mse(pred_joints, gt_joints)))
            if self.smpl_mesh_loss_weight > 0.0:
                smpl_terms.append(('smpl_mesh_loss', self.smpl_mesh_loss_weight, _mse(pred_mesh, gt_mesh)))
            if self.smpl_joint_consistency_loss_weight > 0.0:
                if 'joints' not in pred_dict:
                    raise KeyError('Must regress joints in order to use smpl joint consistency loss!')
                regressed = pred_dict['joints'].reshape((B, NJ, -1)).index_select(0, order)
                smpl_terms.append(('smpl_joint_consistency_loss', self.smpl_joint_consistency_loss_weight, _mse(pred_joints, regressed)))
            if self.smpl_vert_consistency_loss_weight > 0.0:
                if 'verts' not in pred_dict:
                    raise KeyError('Must regress verts in order to use smpl vert consistency loss!')
                key_verts = pred_mesh[:, KEYPT_VERTS, :] if self._dense else pred_mesh      # the subset body returns exactly the key vertices
                regressed = pred_dict['verts'].reshape((B, len(KEYPT_VERTS), -1)).index_select(0, order)
                smpl_terms.append(('smpl_vert_consistency_loss', self.smpl_vert_consistency_loss_weight, _mse(key_verts, regressed)))
            for name, weight, term in smpl_terms:
                stats_dict[name] = term
                loss = loss + weight * term

        if self.kl_loss_weight > 0.0:
            stats_dict['reconstr_weighted_loss'] = loss - stats_dict['kl_weighted_loss']
        return loss, stats_dict

    def zero_pad_tensors(self, pad_list, pad_size):
        '''B x D tensors padded with zero rows (humor_loss.py:352-361); kept for callers, unused by forward.'''
        return [torch.cat([t, t.new_zeros((pad_size, t.size(1)))], dim=0) for t in pad_list]

    def kl_normal(self, qm, qv, pm, pv):
        """KL(q || p) of diagonal normals given means / variances, summed over the last dim (humor_loss.py:364-378)."""
        element_wise = 0.5 * (torch.log(pv) - torch.log(qv) + qv / pv + (qm - pm).pow(2) / pv - 1)
        return element_wise.sum(-1)

    def log_normal(self, x, m, v):
        """log N(x; m, v) summed over the last dim (humor_loss.py:380-392)."""
        return torch.sum(-torch.log(torch.sqrt(v)) - np.log(np.sqrt(2 * np.pi)) - ((x - m) ** 2 / (2 * v)), dim=-1)

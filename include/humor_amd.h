/*
 * humor_amd.h -- C ABI of libhumor_amd.so: the MI355X (gfx950) hot path of HuMoR test-time optimisation.
 *
 * The reference (davrempe/humor) has no FFI for this path: the boundary is three Python classes
 * (BodyModel / HumorModel / MotionOptimizer).  This header is what those classes bind
 * instead of the chain of ATen ops they run today; every entry point names the reference interface it
 * replaces.  Conventions:
 *   - plain C, no torch types; all tensor arguments are raw DEVICE pointers to contiguous row-major fp32
 *     (int32 for index tables) unless marked HOST; the caller owns every buffer;
 *   - the library owns only opaque handles (model constants packed for the kernels, resident in HBM);
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous w.r.t. the host and never
 *     synchronises the device (no hidden hipDeviceSynchronize, safe under stream capture);
 *   - every function returns an int status: 0 = HA_OK, otherwise an HA_ERR_* code; ha_last_error()
 *     returns a thread-local message.  Nothing aborts: the Python layer raises RuntimeError so the
 *     reference's skip-the-batch handling (humor/fitting/run_fitting.py:437-439) keeps working;
 *   - one host thread per device (the multi-GPU runner is one process per GPU).
 */
#ifndef HUMOR_AMD_H
#define HUMOR_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HA_OK 0
#define HA_ERR_INVALID_ARG 1   /* bad shape / null pointer / unsupported size */
#define HA_ERR_HIP 2           /* a HIP runtime call or kernel launch failed */
#define HA_ERR_UNSUPPORTED 3   /* valid request this build cannot serve */

const char* ha_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int ha_abi_version(void);
/* Writes the device's gcnArchName (e.g. "gfx950:sramecc+:xnack-") into buf. */
int ha_device_arch(int device, char* buf, int buflen);
/* Development knobs for kernel launch variants (A/B measurements), process-wide: "skin_variant" (-1 = auto), "layer_spb" (K-slices per block of the roll-out layer kernel, 0 = default), "layer_finish" (0 never / 1 auto / 2 always use the GroupNorm finishing pass), "gemm_rm" (row tiles per wave of the batched prior GEMM: 0 by size / 1 / 2), "rollout_persist" / "rollout_persist_bwd" (0 = launch chain; 1 = persistent kernels, 3 = with write-through publishes), "rollout_pipe" / "rollout_pipe_bwd" (batches of more than 32 sequences: 1 = the layer-parallel pipelined persistent kernels, 0 = launch chain; set the adjoint knobs before the forward call), "rollout_persist_inject" (test hook: 1 = the next persistent forwards drop one CU of team 0, so that the failure path -- NaN results, error word -- can be exercised), "cu_poison" (test hook: != 0 = every persistent roll-out launch is preceded by a kernel that fills the LDS and the vector registers of every CU with a bit pattern -- 1 = a quiet NaN, else the value itself -- so that a read of state the kernel did not write shows on every box). */
int ha_tune_set(const char* key, int value);
/* Test support: fills the LDS and every VGPR / AGPR of all CUs with `pattern` (0 = quiet NaN) on `stream`; with `surviving_words` non-null it then
 * synchronises the stream and counts the LDS words (of 256 x 40960) a following kernel still finds holding the pattern -- 0 means this box clears
 * LDS between kernels and the hook cannot show anything. */
int ha_debug_cu_poison(unsigned int pattern, unsigned int* surviving_words, void* stream);

/* ------------------------------------------------------------------------------------------------
 * SMPL / SMPL+H body model  (replaces smplx==0.1.28 `lbs`, `SMPLH.forward`, `VertexJointSelector` as
 * called by humor/body_model/body_model.py:61-68, 78-91, and the gathers of
 * humor/fitting/motion_optimizer.py:1093-1100)
 * ---------------------------------------------------------------------------------------------- */
typedef struct ha_smpl_model ha_smpl_model;

/* Packs and uploads the model constants.  All array arguments are HOST pointers in the layout of the
 * reference's model.npz (body_model.py:37-48): v_template[V,3], shapedirs[V,3,NB], posedirs[V,3,P]
 * with P=(J-1)*9, J_regressor[J,V], weights[V,J] (dense; packed to <=max-nnz-per-row sparse form),
 * parents[J] (parents[0] ignored / treated as root).  `device` is the HIP device ordinal. */
int ha_smpl_model_create(ha_smpl_model** out, int device, int V, int J, int NB,
                         const float* v_template, const float* shapedirs, const float* posedirs,
                         const float* J_regressor, const float* weights, const int32_t* parents);
int ha_smpl_model_destroy(ha_smpl_model* m);
/* Queries: what = 0:V 1:J 2:NB 3:max skinning influences per vertex 4:tree depth 5:V padded (multiple of 64)
 *          6:number of vertex subsets defined 7:P (pose-blend basis size) */
int ha_smpl_model_info(const ha_smpl_model* m, int what, int* value);

/* Defines vertex subset `slot` (1..7; slot 0 is "all vertices") from HOST ids[n].  Kernels can then be
 * asked to evaluate only that subset (the fitting losses consume <=64 of the 6890 vertices:
 * VertexJointSelector's 21 + KEYPT_VERTS' 43, body_model/utils.py:17-19). */
int ha_smpl_model_define_subset(ha_smpl_model* m, int slot, const int32_t* ids, int n);

/* Forward for N frames on vertex subset `slot` (0 = all).
 *   pose   [N, J*3] axis-angle (global|body|lhand|rhand as SMPLH.forward concatenates them)
 *   n_active_joints: joints >= n_active are treated as exactly zero pose (R = I, pose feature 0);
 *                    22 reproduces pose_hand=None of BodyModel.forward, J = general case
 *   betas  [N, NB]; transl [N,3] or NULL
 * outputs
 *   verts  [N, n_subset, 3]  (subset order; slot 0: [N,V,3])
 *   joints [N, J, 3]         posed joints + transl
 * optional workspace outputs (NULL to skip), needed by the tiled dense path and for debugging:
 *   A_out  [N, J, 12]  relative joint transforms (3x4 row-major)
 * `algo`: 0 = auto, 1 = wave-per-frame VALU kernel, 2 = tiled MFMA pose-blend + streaming skinning
 *         (slot 0 only; needs A_out, ws_vposed [N,V,3](+4) and ws_coeff [Kc, Npad] sized by ha_smpl_workspace),
 *         3 = the pose-blend GEMM with the skinning in its epilogue (forward-only callers: v_posed -- the dense adjoint's input --
 *         is never written; slot 0, needs verts, A_out and ws_coeff, ws_vposed may be NULL; J <= 53; same bits as algo 2). */
int ha_smpl_forward(const ha_smpl_model* m, int slot, int N, int n_active_joints,
                    const float* pose, const float* betas, const float* transl,
                    float* verts, float* joints, float* A_out,
                    float* ws_vposed, float* ws_coeff, int algo, void* stream);

/* Element counts (floats) of the algo-2 workspaces for N frames: *vposed = N*V*3 + 4, *coeff = Kc_pad*Npad. */
int ha_smpl_workspace(const ha_smpl_model* m, int N, int n_active_joints, int64_t* vposed, int64_t* coeff);

/* Backward of ha_smpl_forward for subset `slot`.  Forward intermediates are recomputed from
 * (pose, betas), nothing needs to be stashed.
 *   g_verts  [N, n_subset, 3] or NULL;  g_joints [N, J, 3] or NULL
 * outputs (each may be NULL): g_pose [N, J*3] (entries of joints >= n_active are written as 0),
 *   g_betas [N, NB], g_transl [N, 3]. */
int ha_smpl_backward(const ha_smpl_model* m, int slot, int N, int n_active_joints,
                     const float* pose, const float* betas,
                     const float* g_verts, const float* g_joints,
                     float* g_pose, float* g_betas, float* g_transl, void* stream);

/* Vertex-subset evaluation whose first n_head vertices are delivered as extra joints (BodyModel(use_vtx_selector=True): the
 * reference's Jtr = cat(joints, v[:, selector]), humor/body_model/body_model.py:97-99, without the cat / slice copies around the
 * kernel): joints_ext [N, J + n_head, 3] receives the J joints followed by the first n_head vertices of subset `slot`,
 * verts_tail [N, n - n_head, 3] the remaining ones (may be NULL when n_head == n).  The backward call reads the gradients in the
 * same two tensors (either may be NULL = zero). */
int ha_smpl_forward_split(const ha_smpl_model* m, int slot, int N, int n_active, const float* pose, const float* betas,
                          const float* transl, int n_head, float* joints_ext, float* verts_tail, void* stream);
int ha_smpl_backward_split(const ha_smpl_model* m, int slot, int N, int n_active, const float* pose, const float* betas,
                           int n_head, const float* g_joints_ext, const float* g_verts_tail, float* g_pose, float* g_betas,
                           float* g_transl, void* stream);

/* "Parts" form of the split evaluation (ABI 2; what the stage-3 composites of humor_amd/stage3.py call, so that no ATen launch sits
 * between the kernels of an objective evaluation):
 *   - the pose arrives as root [N,3] + body [N,(n_active-1)*3] (BodyModel.forward's arguments, humor/body_model/body_model.py:72-80,
 *     without the cat in front of the kernel); joints >= n_active rest;
 *   - one shape row serves betas_div consecutive frames: betas [N / betas_div, NB] (the reference expands betas over the frames,
 *     humor/fitting/motion_optimizer.py:1087; here neither the expanded copy nor its summed gradient's expand node exist -- g_betas is
 *     still per frame [N, NB], ha_seq_sum_add folds it);
 *   - forward with n_head == 0 and verts_tail == NULL evaluates the J joints alone (no vertex is blended or skinned); the backward call
 *     skips the vertex phase whenever no vertex can carry a gradient (g_verts_tail NULL and no head row inside gj_rows);
 *   - backward: only the first gj_rows rows of g_joints_ext exist (0 = all J + n_head), row stride gj_stride joints per frame (0 = J +
 *     n_head); the pose gradient leaves as g_root [N,3] + g_body [N,(n_active-1)*3]; add_root / add_body / add_betas [N,NB] /
 *     add_transl (each may be NULL) are added to the corresponding output in the kernel: the gradient another reader of the same
 *     input produced, instead of an accumulation launch behind the call. */
int ha_smpl_forward_parts(const ha_smpl_model* m, int slot, int N, int n_active, const float* root, const float* body,
                          const float* betas, int betas_div, const float* transl, int n_head, float* joints_ext, float* verts_tail,
                          void* stream);
int ha_smpl_backward_parts(const ha_smpl_model* m, int slot, int N, int n_active, const float* root, const float* body,
                           const float* betas, int betas_div, int n_head, const float* g_joints_ext, int gj_rows, int gj_stride,
                           const float* g_verts_tail, const float* add_root, const float* add_body, const float* add_betas,
                           const float* add_transl, float* g_root, float* g_body, float* g_betas, float* g_transl, void* stream);
/* out [B,W] = sum_t src [B,T,W] + add1 [B,W] + add2 [B,W] (addends may be NULL; W <= 64): per-frame shape gradients back to one row
 * per sequence together with the gradients the other readers of the shape rows produced, in one launch. */
int ha_seq_sum_add(int B, int T, int W, const float* src, const float* add1, const float* add2, float* out, void* stream);

/* Backward of the DENSE forward (slot 0, every vertex carries a gradient: the point-cloud / chamfer term of
 * humor/fitting/fitting_loss.py:378-396 back through BodyModel): g_verts [N, V, 3] is required; v_posed [N, V, 3] and
 * A [N, J, 12] are the forward's `ws_vposed` / `A_out`.  The vertex phase runs as batched kernels over all frames
 * (dL/dv_posed streaming pass; dL/dA and dL/dcoeff = dL/dv_posed x Pd^T on the fp32 MFMA units), then the per-frame
 * kinematic-chain adjoint.  Same outputs as ha_smpl_backward; `ws` holds ha_smpl_backward_dense_workspace floats. */
int ha_smpl_backward_dense_workspace(const ha_smpl_model* m, int N, int n_active_joints, int64_t* ws_floats);
int ha_smpl_backward_dense(const ha_smpl_model* m, int N, int n_active_joints, const float* pose, const float* betas,
                           const float* g_verts, const float* g_joints, const float* v_posed, const float* A, float* ws,
                           float* g_pose, float* g_betas, float* g_transl, void* stream);

/* The streaming linear-blend-skinning kernel on its own (the HBM-roofline kernel):
 * verts[n,v,:] = (sum_i w[v,i] * A[n, idx[v,i]]) * [v_posed[n,v,:]; 1] + transl[n].
 * v_posed and verts are [N, V, 3] (v_posed allocated with the +4 floats ha_smpl_workspace reports). */
int ha_lbs_skin(const ha_smpl_model* m, int N, const float* v_posed, const float* A, const float* transl,
                float* verts, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Rotation conversions (replace humor/utils/transforms.py:139-170 batch_rodrigues and :243-389
 * rotation_matrix_to_angle_axis, forward and backward)
 * ---------------------------------------------------------------------------------------------- */
int ha_rodrigues_fwd(int n, const float* aa /*[n,3]*/, float* R /*[n,9]*/, void* stream);
int ha_rodrigues_bwd(int n, const float* aa, const float* gR /*[n,9]*/, float* g_aa /*[n,3]*/, void* stream);
int ha_rotmat_to_aa_fwd(int n, const float* R /*[n,9]*/, float* aa /*[n,3]*/, void* stream);
int ha_rotmat_to_aa_bwd(int n, const float* R, const float* g_aa, float* gR, void* stream);
/* 6-D representation -> rotation matrix (humor/utils/transforms.py:201-220 rot6d_to_rotmat; the decoder of VPoser and
 * HumorModel(out_rot_rep='6d') end in it): x [n,6] viewed as [3,2] (columns a1, a2) -> R [n,9] = [b1 | b2 | b1 x b2]. */
int ha_rot6d_to_rotmat_fwd(int n, const float* x /*[n,6]*/, float* R /*[n,9]*/, void* stream);
int ha_rot6d_to_rotmat_bwd(int n, const float* x, const float* gR /*[n,9]*/, float* gx /*[n,6]*/, void* stream);
/* 9-D representation -> rotation matrix (humor/utils/transforms.py:222-241 rot9d_to_rotmat: U diag(1, 1, det(U V^T)) V^T of the SVD
 * of the 3x3; HumorModel(out_rot_rep='9d'), humor/models/humor_model.py:476-484).  The backward pass is the derivative of that
 * projection itself (finite for equal singular values, where the autograd of torch.svd is not). */
int ha_rot9d_to_rotmat_fwd(int n, const float* x /*[n,9]*/, float* R /*[n,9]*/, void* stream);
int ha_rot9d_to_rotmat_bwd(int n, const float* x, const float* gR /*[n,9]*/, float* gx /*[n,9]*/, void* stream);

/* ------------------------------------------------------------------------------------------------
 * HuMoR CVAE roll-out (replaces HumorModel.roll_out / sample_step / prior / decode / MLP.forward /
 * apply_world2local_trans, humor/models/humor_model.py:407-498, 696-772, 785-1059, 1206-1241, and
 * compute_world2aligned_mat, humor/utils/transforms.py:17-42) for the fitting configuration:
 * in_rot_rep='mat', out_rot_rep='aa', steps_in=1, 'smpl+joints(+contacts)', conditional prior.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ha_humor_net ha_humor_net;

/* One MLP = Linear -> [GroupNorm(16) -> ReLU -> (cat z) -> Linear]*.  HOST arrays, reference
 * state_dict layout: w[i] is [out_i, in_i (+skip)] row-major, gamma/beta are the GroupNorm affine of
 * the activation feeding Linear i (i >= 1). */
typedef struct ha_mlp_desc {
  int n_linear;              /* number of Linear layers (<= 8) */
  int in_dim;                /* input width of Linear 0 (including the skip part) */
  int skip_dim;              /* width of the input tail re-concatenated before every later Linear (0 = none) */
  int out_dims[8];
  const float* w[8];
  const float* b[8];
  const float* gn_gamma[8];  /* index i>=1: affine of GroupNorm before Linear i */
  const float* gn_beta[8];
} ha_mlp_desc;

/* decoder: [339+48] -> 1024 -> 1024 -> 512 -> 216 with z skip; prior: 339 -> 1024 x4 -> 96. */
int ha_humor_net_create(ha_humor_net** out, int device, const ha_mlp_desc* decoder, const ha_mlp_desc* prior);
int ha_humor_net_destroy(ha_humor_net* net);

/* Floats of stash needed per (sequence, step) for the backward pass, and total workspace floats for
 * a roll-out of B sequences x S steps. */
int ha_humor_rollout_workspace(const ha_humor_net* net, int B, int S, int64_t* stash_floats);

/* Forward roll-out.
 *   past_in0 [B,339]  initial (canonical-frame) input state, layout: trans3|trans_vel3|rootR9|root_vel3|
 *                     bodyR189|joints66|joints_vel66
 *   z_seq    [B,S,48]
 * outputs
 *   world    [B,S,348] world-frame predicted states (past_in layout + 9 contact logits)
 *   prior_mu [B,S,48], prior_var [B,S,48] (NULL to skip the prior network, G10)
 *   stash    workspace kept for ha_humor_rollout_backward */
int ha_humor_rollout_forward(const ha_humor_net* net, int B, int S, const float* past_in0, const float* z_seq,
                             float* world, float* prior_mu, float* prior_var, float* stash, void* stream);

/* Sampling roll-out (HumorModel.roll_out with z_seq=None, humor_model.py:1029-1047; test_humor.py:224): at every step
 * z_t = prior_mu_t + eps_t * sqrt(prior_var_t) (eps_seq [B,S,48]; NULL = use the prior mean, `use_mean=True`).
 * Outputs as ha_humor_rollout_forward plus the sampled latents z_out [B,S,48].  Forward only. */
int ha_humor_rollout_sample(const ha_humor_net* net, int B, int S, const float* past_in0, const float* eps_seq,
                            float* world, float* prior_mu, float* prior_var, float* z_out, float* stash, void* stream);

/* Backward roll-out: given gradients of the outputs, produce gradients of the inputs.
 *   g_world [B,S,348], g_prior_mu / g_prior_var [B,S,48] (NULL = zero)
 * outputs g_past_in0 [B,339], g_z_seq [B,S,48].  `stash` is the forward's, consumed read-only except
 * for scratch regions reserved by ha_humor_rollout_workspace. */
int ha_humor_rollout_backward(const ha_humor_net* net, int B, int S, const float* z_seq,
                              const float* g_world, const float* g_prior_mu, const float* g_prior_var,
                              float* stash, float* g_past_in0, float* g_z_seq, void* stream);

/* ha_humor_rollout_backward with an addend: g_z_add [B,S,48] (or NULL) is added to g_z_seq -- the gradient another reader of z_seq (the
 * motion-prior loss term) produced -- inside the adjoint's last kernel instead of by an accumulation launch behind the call. */
int ha_humor_rollout_backward_ex(const ha_humor_net* net, int B, int S, const float* z_seq,
                                 const float* g_world, const float* g_prior_mu, const float* g_prior_var,
                                 float* stash, float* g_past_in0, float* g_z_seq, const float* g_z_add, void* stream);

/* Per-network options.  "output_delta" (default 1): the decoder emits residuals that are composed with the input state
 * (HumorModel(output_delta=True), humor/models/humor_model.py:460-494); 0: it emits the next state itself and only its rotations are
 * converted (humor_model.py:331-347).  Networks with output_delta = 0 run the launch chain. */
int ha_humor_net_set_option(ha_humor_net* net, const char* key, int value);

/* Persistent roll-out (ha_tune_set "rollout_persist" != 0; B <= 32: weight-stationary kernels, 32 < B: pipelined kernels): state of the one-launch path of this
 * network.  *available = 1 when the network / device qualify and no launch has reported a failure; *error_word = the kernel's
 * host-mapped error word (0 = none; 0x1xx / 0x3xx an XCD received more than its 32 blocks (forward / adjoint; pipelined kernels: 0x5xx / 0x7xx),
 * 0x2xx / 0x4xx a team's bounded wait ran out (forward / adjoint; pipelined kernels: 0x6xx / 0x8xx): that team's output rows -- world states; dL/dpast_in0 and dL/dz -- are filled with NaN),
 * meaningful once the stream of the last roll-out has been synchronised; *launches = persistent forwards issued so far for this
 * network in the low 32 bits, persistent adjoints in the high 32 bits.  After a failure the library uses the launch chain. */
int ha_humor_persist_status(const ha_humor_net* net, int* available, unsigned int* error_word, int64_t* launches);
/* Acknowledges a failure the caller has read through ha_humor_persist_status and handled itself (MotionOptimizer aborts the fit it happened in --
 * the only detection path for closures replayed from a hipGraph, which never pass an entry point): without it the NEXT roll-out entry point on this
 * network would return the failure once more, and abort a fit whose evaluations all ran validly on the launch chain.  No-op when the word is 0. */
int ha_humor_persist_ack(const ha_humor_net* net);

/* ------------------------------------------------------------------------------------------------
 * Frozen MLPs on N independent rows (no weight gradients): VPoser v1.0's decoder / encoder as MotionOptimizer.latent2pose /
 * pose2latent call them in every closure (humor/fitting/motion_optimizer.py:1041-1063; Linear + LeakyReLU(0.2), eval-mode
 * BatchNorm folded into the Linear layers by the host, the decoder followed by 6-D -> rotation matrix -> axis-angle), and
 * HuMoR's posterior encoder (humor/models/humor_model.py:180-190, 400-405: Linear + GroupNorm(16) + ReLU on [past | next]).
 * ---------------------------------------------------------------------------------------------- */
typedef struct ha_mlp ha_mlp;
#define HA_MLP_GN_RELU 0        /* GroupNorm(16) + ReLU before every Linear after the first (desc.gn_gamma / gn_beta) */
#define HA_MLP_LEAKY_RELU 1     /* LeakyReLU(slope) between the Linear layers */
#define HA_MLP_TAIL_NONE 0      /* y [N, out] */
#define HA_MLP_TAIL_ROT6D_AA 1  /* every 6 outputs are a 6-D rotation: y [N, out/6, 3] axis-angle (transforms.py:201-220, 243-389) */
int ha_mlp_create(ha_mlp** out, int device, const ha_mlp_desc* desc /* skip_dim = 0 */, int act, float slope);
int ha_mlp_destroy(ha_mlp* mlp);
int ha_mlp_workspace(const ha_mlp* mlp, int N, int64_t* ws_floats);
/* x [N, in] -> y; `ws` (ha_mlp_workspace floats) keeps what ha_mlp_backward needs. */
int ha_mlp_forward(const ha_mlp* mlp, int N, const float* x, int tail, float* y, float* ws, void* stream);
/* g_y (shape of y) -> g_x [N, in]; `ws` is the forward's (its scratch part is overwritten). */
int ha_mlp_backward(const ha_mlp* mlp, int N, const float* g_y, int tail, float* ws, float* g_x, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fitting objective: the data / regularisation terms of FittingLoss.root_fit / smpl_fit / motion_fit
 * (humor/fitting/fitting_loss.py:94-309; term definitions :317-484, 504-516; gmof fitting_utils.py:250-258;
 * perspective_projection fitting_utils.py:647-676 with identity extrinsics) and their gradients, in one pass.
 * Not covered (stay with the caller): the init-state GMM term (:416-429), points3d / chamfer, the cross-batch
 * `prev_batch_overlap_res` terms, irregular (per-pair different) overlaps are covered through `overlap[b]`.
 * ---------------------------------------------------------------------------------------------- */
#define HA_FIT_J2D 0            /* joints2d_loss: conf^2 * gmof(projected - observed), OpenPose BODY_25 */
#define HA_FIT_J3D 1            /* joints3d_loss on the camera-frame SMPL joints (inf = invisible) */
#define HA_FIT_V3D 2            /* verts3d_loss on the key vertices */
#define HA_FIT_J3D_RO 3         /* joints3d_loss on the roll-out joints ('joints3d_rollout') */
#define HA_FIT_POSE_PRIOR 4     /* sum latent_pose^2 */
#define HA_FIT_SHAPE_PRIOR 5    /* sum betas^2 (x nsteps in the loss) */
#define HA_FIT_SMOOTH 6         /* joints3d_smooth_loss */
#define HA_FIT_MOTION_PRIOR 7   /* -sum log N(z; prior_mu, prior_var), or sum z^2 without a conditional prior */
#define HA_FIT_JOINT_CONSIST 8  /* 0.5 sum (SMPL joints - roll-out joints)^2 */
#define HA_FIT_BONE_LEN 9       /* bone_length_loss on the roll-out joints (SMPL_PARENTS, body_model/utils.py:9) */
#define HA_FIT_CONTACT_VEL 10
#define HA_FIT_CONTACT_H 11     /* contact_height_loss, threshold 0.08 */
#define HA_FIT_FLOOR_REG 12     /* floor_reg_loss (x nsteps in the loss) */
#define HA_FIT_OV_VPOS 13       /* overlap consistency of consecutive sub-sequences: key-vertex positions ... */
#define HA_FIT_OV_VVEL 14       /* ... their frame differences ... */
#define HA_FIT_OV_BETAS 15      /* ... betas ... */
#define HA_FIT_OV_FLOOR 16      /* ... floor plane (fitting_loss.py:135-157, 211-215, 296-300) */
#define HA_FIT_NTERMS 17

/* All pointers are DEVICE pointers to contiguous fp32 (int32 for tables); NULL = absent.  B sub-sequences x T frames. */
typedef struct ha_fit_args {
  int B, T;
  /* predictions */
  const float* cam_jtr; int nj;        /* [B,T,nj,3] camera-frame SMPL joints (+ selected vertices), nj in [22,128] */
  const float* cam_verts; int nv;      /* [B,T,nv,3] camera-frame key vertices */
  const float* pri_joints; int pri_nj; /* [B,T,pri_nj,3] prior-frame SMPL joints (the first 22 are read; pri_nj >= 22) */
  const float* ro_joints;              /* [B,T,22,3] roll-out joints */
  const float* contacts_conf;          /* [B,T,22] */
  const float* latent_pose; int dlp;   /* [B,T,dlp] */
  const float* betas; int nb;          /* [B,nb] */
  const float* latent_motion; const float* prior_mu; const float* prior_var; int S; int dz;   /* [B,S,dz], S <= T */
  const float* floor;                  /* [B,3] */
  /* halo of the multi-GPU sharding: the sequence before local sequence 0 lives on the previous rank */
  const float* prev_tail;              /* [T,nv,3] its predicted key vertices */
  const float* prev_betas;             /* [nb] */
  const float* prev_floor;             /* [3] */
  /* observations and tables */
  const float* obs_j2d;                /* [B,T,25,3] (x, y, confidence) */
  const int32_t* smpl2op;              /* [25] index into the nj joints (smpl_to_openpose, body_model/utils.py:53-56) */
  const float* op_mask;                /* [25] 0 for OP_IGNORE_JOINTS (fitting_utils.py:679), else 1 */
  const float* cam_f; const float* cam_c;   /* [B,2] focal lengths, principal point */
  float sigma;                         /* joints2d_sigma */
  const float* obs_j3d;                /* [B,T,22,3], +-inf = invisible */
  const float* obs_v3d;                /* [B,T,nv,3] */
  const float* obs_floor;              /* [B,4] (a,b,c,d) */
  const int32_t* overlap;              /* [B] frames sequence b shares with its predecessor (0 = none); NULL = no overlap terms */
  /* weights of the current stage, indexed by HA_FIT_* (0 = term off) */
  float w[HA_FIT_NTERMS];
  float nsteps;
  /* outputs: unweighted term values, the weighted loss, and d(loss)/d(input) for every input given (fully written) */
  float* terms;                        /* [HA_FIT_NTERMS] */
  float* loss;                         /* [1] */
  float* g_cam_jtr; float* g_cam_verts; float* g_pri_joints; float* g_ro_joints; float* g_contacts_conf;
  float* g_latent_pose; float* g_betas; float* g_latent_motion; float* g_prior_mu; float* g_prior_var; float* g_floor;
  float* g_prev_tail; float* g_prev_betas; float* g_prev_floor;
  float* partial;                      /* workspace [B*T, HA_FIT_NTERMS] */
  /* optional (ABI 2): the stage-3 init-state prior (ha_gmm_nll run by the caller just before, on the same stream) folded into the
   * objective: loss += gmm_w * sum_b gmm_nll[b]; gmm_total[0] = sum_b gmm_nll[b]; and gmm_w * gmm_gx [B, gmm_D] is handed out segment
   * by segment (as ha_gmm_args lays x_b out): segment 0 (66 floats: the 22 joints of frame 0 of pri_joints) is added to frame 0 of
   * g_pri_joints by the loss kernel itself when gmm_g[0] != NULL; segment s >= 1 of row b is stored to gmm_g[s] + b * gmm_g_stride[s]
   * (gmm_g_acc is reserved, pass 0). */
  const float* gmm_nll; const float* gmm_gx; float gmm_w; int gmm_D; int gmm_nseg;
  float* gmm_g[4]; int gmm_seg_width[4]; int gmm_g_stride[4]; int gmm_g_acc[4];
  float* gmm_total;
} ha_fit_args;

int ha_fit_loss(const ha_fit_args* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Stage-3 set-up of one closure evaluation with an optimised floor: compute_cam2prior (humor/fitting/fitting_utils.py:149-190),
 * the forward apply_cam2prior of the key frame (humor/fitting/motion_optimizer.py:678-742) and the initial roll-out state
 * (motion_optimizer.py:905-942), from ONE camera-frame SMPL evaluation (the prior-frame joints are its rigid image).
 *   inputs  floor [B,3] (normal * offset), trans0 / root0 [B,3] (camera frame, axis-angle), pose0 [B,63], jcam [B,22,3] camera-
 *           frame SMPL joints of frame 0, trans_vel [B,3], joints_vel [B,22,3], root_orient_vel [B,3]
 *   outputs past_in [B,339], trans_p / root_p [B,3] (prior frame), joints_p [B,22,3], c2p_R [B,9], c2p_t [B,3], root_height [B]
 * Backward: g_* of the outputs may be NULL (= zero); every input gradient is fully written.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ha_fit_pre_args {
  int B;
  const float* floor; const float* trans0; const float* root0; const float* pose0; const float* jcam;
  const float* trans_vel; const float* joints_vel; const float* root_orient_vel;
  float* past_in; float* trans_p; float* root_p; float* joints_p; float* c2p_R; float* c2p_t; float* root_height;
  const float* g_past_in; const float* g_trans_p; const float* g_root_p; const float* g_joints_p; const float* g_c2p_R;
  const float* g_c2p_t; const float* g_root_height;
  float* g_floor; float* g_trans0; float* g_root0; float* g_pose0; float* g_jcam; float* g_trans_vel; float* g_joints_vel;
  float* g_root_orient_vel;
  /* optional (ABI 2).  jcam_stride: floats between the sequences' rows of jcam (0 = 66; 219 reads the first 22 rows of a [B,73,3] joint
   * tensor in place).  Backward addends (each may be NULL): the gradient another reader of floor / pose0 / the velocities produced,
   * added to the corresponding output in the kernel instead of by an accumulation launch behind it. */
  int jcam_stride;
  const float* add_floor; const float* add_pose0; const float* add_trans_vel; const float* add_joints_vel; const float* add_root_orient_vel;
} ha_fit_pre_args;
int ha_fit_pre_forward(const ha_fit_pre_args* args, void* stream);
int ha_fit_pre_backward(const ha_fit_pre_args* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Roll-out post-processing (replaces the op chain of MotionOptimizer.rollout_latent_motion after roll_out,
 * humor/fitting/motion_optimizer.py:950-1019, and apply_cam2prior(inverse=True), :678-742): R -> axis-angle of the root and
 * body rotations, the optimised frame 0 prepended (T = S + 1 frames), contact logits -> confidences / labels on the 22 SMPL
 * joints (CONTACT_INDS, amass_utils.py:21-23; frame 0 repeats frame 1), and the root trajectory mapped back into the camera
 * frame (cam_root = aa(R^T rodrigues(root)), cam_trans = R^T (trans - trans_0) - t) when c2p_R / c2p_t are given.
 * The same struct serves both directions; gradient inputs may be NULL (= zero), every gradient output is fully written.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ha_rollout_post_args {
  int B, S;
  const float* world;                               /* [B,S,348] world-frame roll-out output */
  const float* trans0; const float* root0; const float* pose0; const float* joints0;   /* frame 0: [B,3] [B,3] [B,63] [B,22,3] */
  const float* c2p_R; const float* c2p_t;           /* [B,9], [B,3] or NULL */
  /* forward outputs (inputs of the backward call) */
  float* trans; float* root_orient; float* pose_body; float* joints;   /* [B,T,3] [B,T,3] [B,T,63] [B,T,22,3] */
  float* contacts_conf; float* contacts;            /* [B,T,22] */
  float* cam_trans; float* cam_root_orient;         /* [B,T,3] or NULL */
  /* backward: gradients of the outputs ... */
  const float* g_trans; const float* g_root_orient; const float* g_pose_body; const float* g_joints; const float* g_contacts_conf;
  const float* g_cam_trans; const float* g_cam_root_orient;
  /* ... and of the inputs */
  float* g_world; float* g_trans0; float* g_root0; float* g_pose0; float* g_joints0; float* g_c2p_R; float* g_c2p_t;
  float* partial;                                   /* workspace [B*T, 15] */
} ha_rollout_post_args;
int ha_rollout_post_forward(const ha_rollout_post_args* args, void* stream);
int ha_rollout_post_backward(const ha_rollout_post_args* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The SMPL body of a frame under a second root pose (replaces the second BodyModel evaluation of a stage-3 closure,
 * humor/fitting/motion_optimizer.py:584 -- same pose_body and betas as :573, camera-frame root instead of the prior-frame one).
 * Every SMPL output point is R (x - J0) + J0 + t with x independent of the root (humor/body_model/body_model.py:146-153), so
 *     X' = Q (X - p) + p - t + t',   Q = rodrigues(root2) rodrigues(root)^T,   p = joints[:, 0]  (= J0 + t).
 *   inputs  joints [N,J,3] / verts [N,V,3] (V may be 0) of the first evaluation, root / trans [N,3] it was made with,
 *           root2 / trans2 [N,3] the second root pose (axis-angle)
 *   outputs joints2 [N,J,3], verts2 [N,V,3]
 * Backward: g_joints2 / g_verts2 may be NULL (= zero); g_joints, g_verts, g_root, g_trans, g_root2, g_trans2 are fully written.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ha_rigid_image_args {
  int N, J, V;
  const float* joints; const float* verts; const float* root; const float* trans; const float* root2; const float* trans2;
  float* joints2; float* verts2;
  const float* g_joints2; const float* g_verts2;
  float* g_joints; float* g_verts; float* g_root; float* g_trans; float* g_root2; float* g_trans2;
  /* backward, optional (ABI 2): addends of g_joints / g_verts -- the gradient another reader of joints / verts produced, added in
   * the kernel instead of by an accumulation launch behind it */
  const float* g_joints_add; const float* g_verts_add;
} ha_rigid_image_args;
int ha_rigid_image_forward(const ha_rigid_image_args* args, void* stream);
int ha_rigid_image_backward(const ha_rigid_image_args* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Init-state prior of stage 3 (replaces FittingLoss.init_motion_prior_loss, humor/fitting/fitting_loss.py:416-429: the negative
 * log-density of frame 0's joints | joint velocities | root velocity | root angular velocity under the Gaussian mixture loaded by
 * run_fitting.py:248-261), value and gradient in two launches.
 *   x_b      the nseg (<= 4) segments side by side: seg[s] + b * seg_stride[s], seg_width[s] floats each (D in total, D <= 256)
 *   means [K,D]; Linv [K,D,D] the inverses of the Cholesky factors of the covariances (row-major) and LinvT their transposes;
 *   cst [K] = log(weight) - log det(L) - D/2 log(2 pi);  K <= 64
 *   workspaces lp [B,K], gpart [B,K,D];  outputs nll [B] = -logsumexp_k lp, g_x [B,D] = d nll_b / d x_b
 * ---------------------------------------------------------------------------------------------- */
typedef struct ha_gmm_args {
  int B, K, D, nseg;
  const float* seg[4]; int seg_width[4]; int seg_stride[4];
  const float* means; const float* Linv; const float* LinvT; const float* cst;
  float* lp; float* gpart; float* nll; float* g_x;
} ha_gmm_args;
int ha_gmm_nll(const ha_gmm_args* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Chamfer distance (replaces humor/utils/chamfer_distance/chamfer_distance.cu: ChamferDistanceKernelLauncher :140-157 and
 * ChamferDistanceGradKernelLauncher :189-208, i.e. chamfer_distance.py's cd.forward_cuda / cd.backward_cuda).
 *   xyz1 [b,n,3], xyz2 [b,m,3];  dist1[b,n] / idx1[b,n]: squared distance to, and index of, the nearest point of xyz2 for every
 *   point of xyz1 (ties -> lowest index); dist2 / idx2 [b,m] the other direction.  Distances are (dx*dx + dy*dy) + dz*dz in
 *   fp32 without FMA contraction = the reference's CPU path (chamfer_distance.cpp:59-87) bit for bit.
 * Backward: grad_xyz1 [b,n,3] and grad_xyz2 [b,m,3] are fully written (no pre-zeroing needed).
 * ---------------------------------------------------------------------------------------------- */
int ha_chamfer_forward(int b, int n, const float* xyz1, int m, const float* xyz2, float* dist1, int32_t* idx1, float* dist2,
                       int32_t* idx2, void* stream);
int ha_chamfer_backward(int b, int n, const float* xyz1, int m, const float* xyz2, const float* grad_dist1, const int32_t* idx1,
                        const float* grad_dist2, const int32_t* idx2, float* grad_xyz1, float* grad_xyz2, void* stream);

/* ------------------------------------------------------------------------------------------------
 * L-BFGS direction (the two-loop recursion of torch.optim.LBFGS.step, which the reference drives from
 * humor/fitting/motion_optimizer.py:233-254, 284-310, 461-512) in coefficient form: given the Gram matrix
 * G [2h,2h] of the stored pairs M = [s slots (h rows) ; y slots (h rows)], Mg = M g [2h] and the initial
 * Hessian scale, writes coef [2h] such that  d = M^T coef - h_diag * g.  `order` (HOST, num_old entries) lists the
 * physical slots from the oldest to the newest pair; slots not listed get coefficient 0.  hist <= 128.  The scale is `h_diag`,
 * or -- when h_diag_dev is non-NULL -- the float it points to in device memory (no host round trip for ys / yy).
 * ---------------------------------------------------------------------------------------------- */
int ha_lbfgs_coeffs(int hist, int num_old, const int32_t* order, const float* G, const float* Mg, float h_diag,
                    const float* h_diag_dev, float* coef, void* stream);

/* The rest of one L-BFGS inner iteration between two closure evaluations (torch/optim/lbfgs.py: y = g - g_prev, s = t d, the
 * curvature test y.s > 1e-10, H = y.s / y.y, the direction, g.d, max|d| -- ~25 small launches and three reads of the history),
 * as four launches with a fixed summation order (the replicated multi-GPU optimiser needs bit-identical directions on all ranks).
 * M is the history [rows_total][n], row-major: rows 0..h-1 the s slots, h..2h-1 the y slots, row 2h the current gradient.
 *   ha_lbfgs_gram         P [rows][3] = M[r] . (M[i0], M[i1], M[i2]) for r < rows, one read of M; `part` is scratch of
 *                         ha_lbfgs_gram_workspace floats
 *   ha_lbfgs_pair_coeffs  ha_lbfgs_coeffs after installing the pair in `slot` from P (Gram rows / columns, Mg = P[:,2]) with the scale
 *                         H = y.s / y.y taken from P; additionally coef[2h] = -H (so that d = M[:2h+1]^T coef) and scal = (y.s, y.y)
 *   ha_lbfgs_scalars      out[0] = a.b, out[1] = max|a|, out[2] = sum|a|, out[3] = *extra (0 if NULL): what the line search reads
 *                         after a closure evaluation (a = gradient, b = direction, extra = loss) or a direction update */
int ha_lbfgs_gram(int n, int rows, const float* M, int i0, int i1, int i2, float* part, float* P, void* stream);
int ha_lbfgs_gram_workspace(int n, int rows, int64_t* part_floats);
int ha_lbfgs_pair_coeffs(int hist, int num_old, const int32_t* order, int slot, const float* P, float* G, float* Mg, float* coef,
                         float* scal, void* stream);
int ha_lbfgs_scalars(int n, const float* a, const float* b, const float* extra, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HUMOR_AMD_H */

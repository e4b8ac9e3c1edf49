"""Fixed index tables of the fitting path.  These are data (bit-exact parity is required), cited to
the reference file that defines each one."""

# humor/body_model/utils.py:5-8
SMPL_JOINTS = {'hips': 0, 'leftUpLeg': 1, 'rightUpLeg': 2, 'spine': 3, 'leftLeg': 4, 'rightLeg': 5,
               'spine1': 6, 'leftFoot': 7, 'rightFoot': 8, 'spine2': 9, 'leftToeBase': 10, 'rightToeBase': 11,
               'neck': 12, 'leftShoulder': 13, 'rightShoulder': 14, 'head': 15, 'leftArm': 16, 'rightArm': 17,
               'leftForeArm': 18, 'rightForeArm': 19, 'leftHand': 20, 'rightHand': 21}
# humor/body_model/utils.py:9 -- used only by the bone-length loss (NOT the LBS kinematic tree)
SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 12, 12, 13, 14, 16, 17, 18, 19]
NUM_BODY_JOINTS = len(SMPL_JOINTS) - 1

# humor/body_model/utils.py:17-19 -- the 43 "virtual marker" vertices
KEYPT_VERTS = [4404, 920, 3076, 3169, 823, 4310, 1010, 1085, 4495, 4569, 6615, 3217, 3313, 6713,
               6785, 3383, 6607, 3207, 1241, 1508, 4797, 4122, 1618, 1569, 5135, 5040, 5691, 5636,
               5404, 2230, 2173, 2108, 134, 3645, 6543, 3123, 3024, 4194, 1306, 182, 3694, 4294, 744]

# humor/datasets/amass_utils.py:21-23
CONTACT_ORDERING = ['hips', 'leftLeg', 'rightLeg', 'leftFoot', 'rightFoot', 'leftToeBase', 'rightToeBase', 'leftHand', 'rightHand']
CONTACT_INDS = [SMPL_JOINTS[j] for j in CONTACT_ORDERING]

# humor/body_model/utils.py:53-56: smpl_to_openpose('smplh', use_hands=False, use_face=False, 'coco25')
SMPLH_TO_OPENPOSE25 = [52, 12, 17, 19, 21, 16, 18, 20, 0, 2, 5, 8, 1, 4, 7, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62]
# humor/fitting/fitting_utils.py:678-682
OP_NUM_JOINTS = 25
OP_IGNORE_JOINTS = [1, 9, 12]
OP_EDGE_LIST = [[1, 8], [1, 2], [1, 5], [2, 3], [3, 4], [5, 6], [6, 7], [8, 9], [9, 10], [10, 11], [8, 12], [12, 13], [13, 14],
                [1, 0], [0, 15], [15, 17], [0, 16], [16, 18], [14, 19], [19, 20], [14, 21], [11, 22], [22, 23], [11, 24]]

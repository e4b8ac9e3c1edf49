"""ORACLE (test infrastructure only).  Imports the *unmodified* reference Python from /root/reference/humor
in THIS container so the restatements in oracle/ can be validated against it and golden vectors can be
generated (oracle/make_golden.py).  /root/reference does not exist on the GPU box: nothing that runs
there may call into this module -- ``available()`` is the guard.

Two in-memory shims make the reference importable:
  * ``smplx``  -> oracle/lbs_restated.py (the un-vendored smplx==0.1.28 arithmetic, restated)
  * ``cv2``    -> empty module (imported by fitting_utils / transforms but unused on the path)
"""
import os
import sys
import types

REF_ROOT = '/root/reference/humor'
_loaded = {}


def available():
    return os.path.isdir(REF_ROOT)


def _install_shims():
    from oracle import lbs_restated as L

    if 'smplx' not in sys.modules or not getattr(sys.modules['smplx'], '_humor_amd_oracle_shim', False):
        smplx = types.ModuleType('smplx')
        smplx._humor_amd_oracle_shim = True

        class SMPL(L.SMPLHLayer):
            NUM_JOINTS = 23
            SHAPE_SPACE_DIM = 300

        class SMPLH(L.SMPLHLayer):
            pass

        class SMPLX(L.SMPLHLayer):
            NUM_JOINTS = 54

        smplx.SMPL, smplx.SMPLH, smplx.SMPLX = SMPL, SMPLH, SMPLX
        vid = types.ModuleType('smplx.vertex_ids')
        vid.vertex_ids = L.VERTEX_IDS
        utils = types.ModuleType('smplx.utils')

        class Struct(object):
            def __init__(self, **kwargs):
                for key, val in kwargs.items():
                    setattr(self, key, val)

        utils.Struct = Struct
        smplx.vertex_ids, smplx.utils = vid, utils
        sys.modules['smplx'] = smplx
        sys.modules['smplx.vertex_ids'] = vid
        sys.modules['smplx.utils'] = utils
    if 'cv2' not in sys.modules:
        sys.modules['cv2'] = types.ModuleType('cv2')


def load():
    """Returns a namespace with the reference modules: transforms, humor_model, body_model,
    motion_optimizer, fitting_loss, fitting_utils, amass_utils, bm_utils."""
    if _loaded:
        return types.SimpleNamespace(**_loaded)
    if not available():
        raise RuntimeError('reference tree not present (expected only inside the build container)')
    _install_shims()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import importlib
    names = {
        'transforms': 'utils.transforms',
        'humor_model': 'models.humor_model',
        'body_model': 'body_model.body_model',
        'bm_utils': 'body_model.utils',
        'amass_utils': 'datasets.amass_utils',
        'fitting_utils': 'fitting.fitting_utils',
        'fitting_loss': 'fitting.fitting_loss',
        'motion_optimizer': 'fitting.motion_optimizer',
    }
    for short, mod in names.items():
        _loaded[short] = importlib.import_module(mod)
    return types.SimpleNamespace(**_loaded)

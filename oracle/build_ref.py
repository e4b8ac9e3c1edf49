"""ORACLE (test infrastructure only).  Builds the part of the reference that is compiled code on the fitting path -- the CPU
nearest-neighbour search / gradient of humor/utils/chamfer_distance/chamfer_distance.cpp:59-187 -- from the sources WHERE THEY LIE
under /root/reference into oracle/_ref/ (git-ignored, travels to the GPU box with the snapshot).  Nothing is copied into the repo.

    python -m oracle.build_ref        # build container only (needs /root/reference)

``load()`` returns the compiled module (functions ``forward(xyz1, xyz2, dist1, dist2, idx1, idx2)`` and ``backward(...)`` exactly as
chamfer_distance.py:27, 52 calls them) or None when neither the sources nor a previous build are available."""
import glob
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = '/root/reference/humor/utils/chamfer_distance/chamfer_distance.cpp'
OUT_DIR = os.path.join(HERE, '_ref')
NAME = 'humor_ref_cd'


def _built():
    hits = glob.glob(os.path.join(OUT_DIR, NAME + '*.so'))
    return hits[0] if hits else None


def build(verbose=False):
    if _built():
        return _built()
    if not os.path.exists(REF_SRC):
        return None
    os.makedirs(OUT_DIR, exist_ok=True)
    from torch.utils.cpp_extension import load
    load(name=NAME, sources=[REF_SRC, os.path.join(HERE, 'chamfer_ref_stub.cpp')], build_directory=OUT_DIR, verbose=verbose,
         extra_cflags=['-O2', '-ffp-contract=off'])
    return _built()


def load():
    so = _built() or build()
    if so is None:
        return None
    import torch  # noqa: F401  (libtorch must be loaded before the extension)
    spec = importlib.util.spec_from_file_location(NAME, so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == '__main__':
    print(build(verbose='-v' in sys.argv))

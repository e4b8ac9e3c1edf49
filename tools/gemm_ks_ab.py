import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from humor_amd import synth, _lib
dev = torch.device('cuda:0')
npz = synth.write_smplh_npz('/tmp/model_ks.npz', seed=0)
lib = _lib.get_lib()
for ks in (0, 2, 3, 0, 2, 3):
    lib.call('ha_tune_set', b'gemm_ks', ks)
    fc = bench.FitClosure(dev, npz, 1, 0, None, use_graphs=False)
    for _ in range(5): fc.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): fc.step()
    torch.cuda.synchronize(); print('gemm_ks', ks, 'closure %.3f ms' % ((time.perf_counter() - t0) * 25))

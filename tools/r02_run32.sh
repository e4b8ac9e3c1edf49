R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_run32
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_smpl_gpu.py -q -x > $OUT/pytest_smpl.txt 2>&1; tail -5 $OUT/pytest_smpl.txt | cut -c1-300
timeout 300 python - > $OUT/dense_fwd.txt 2>&1 <<'PY'
import os, sys, tempfile, torch
sys.path.insert(0, os.getcwd())
from humor_amd import synth
from humor_amd.body_model import BodyModel
import bench
dev = torch.device('cuda:0')
npz = synth.write_smplh_npz(os.path.join(tempfile.mkdtemp(), 'm.npz'))
for N in (1920, 7680, 30720):
    B = N // 60
    root, body, trans = synth.smooth_pose_sequence(B, 60, seed=1)
    args = dict(root_orient=root.reshape(N, 3).to(dev), pose_body=body.reshape(N, 63).to(dev), trans=trans.reshape(N, 3).to(dev), betas=torch.randn(N, 16, device=dev))
    for algo in (2, 3, 2, 3):
        bm = BodyModel(npz, num_betas=16, use_vtx_selector=True, algo=algo)
        with torch.no_grad():
            ms = bench.time_events(lambda: bm(**args), iters=20, warm=10)
        print(f'N={N} algo={algo}: dense SMPL forward {ms:.4f} ms = {N * 6890 / ms / 1e6:.1f} G verts/s', flush=True)
PY
cat $OUT/dense_fwd.txt | grep "N=" 

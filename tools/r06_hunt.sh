cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_hunt
O=gpurun_out/r06_hunt
timeout 300 python tools/nan_hunt.py --smoke --poison nan --cu-poison 1 > $O/smoke_paths_cupoison.txt 2>&1; tail -5 $O/smoke_paths_cupoison.txt
timeout 600 python tools/nan_hunt.py --poison nan --cu-poison 1 --grid 8 > $O/hunt_cupoison.txt 2>&1; tail -25 $O/hunt_cupoison.txt
timeout 600 python tools/nan_hunt.py --poison nan --cu-poison 0xffffffff --grid 5 > $O/hunt_cupoison_ff.txt 2>&1; tail -12 $O/hunt_cupoison_ff.txt
timeout 600 python tools/nan_hunt.py --poison nan --cu-poison 1 --cases 9x3,17x5,31x2,32x4,33x3,40x2,70x3,130x2,260x2,288x2 > $O/hunt_cupoison_pipe.txt 2>&1; tail -20 $O/hunt_cupoison_pipe.txt
HUMOR_AMD_CU_POISON=1 timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_poison.txt 2>&1; tail -60 $O/pytest_poison.txt | cut -c1-250

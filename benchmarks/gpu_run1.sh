set -e
export TMPDIR=/tmp
timeout -k 10 300 python bench.py --cpu-steps 0 > gpurun_out/b_default.log 2>&1 || { tail -20 gpurun_out/b_default.log; exit 1; }
tail -1 gpurun_out/b_default.log
timeout -k 10 300 python bench.py --cpu-steps 0 --no-fuse-opt > gpurun_out/b_nofuse.log 2>&1
tail -1 gpurun_out/b_nofuse.log
timeout -k 10 400 python -u -m pytest tests/test_trajectory_gpu.py -x -q --timeout 120 --timeout-method thread > gpurun_out/t_traj.log 2>&1 || { tail -30 gpurun_out/t_traj.log; exit 1; }
tail -1 gpurun_out/t_traj.log

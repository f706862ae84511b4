set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w $R/gpurun_out/kt
cd $R
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- python3 bench.py --steps 3 --warmup 1 --cpu-steps 0 --streams 1 > gpurun_out/pmc_f.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python3 bench.py --steps 3 --warmup 1 --cpu-steps 0 --streams 1 > gpurun_out/pmc_w.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -o kt -- python3 bench.py --steps 10 --warmup 3 --cpu-steps 0 > gpurun_out/kt.log 2>&1
tail -1 gpurun_out/kt.log
ls -la gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/kt

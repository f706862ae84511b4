#!/bin/bash
source benchmarks/gpu_steps.sh
step r3_t_new 900 python -u -m pytest tests/test_fullsize_gpu.py tests/test_wgemm_gpu.py -x -v --timeout 600 --timeout-method thread -k "nb_decoder or whole_paired or just_below"
step r3_b_c2 400 python bench.py --steps 100
step r3_b_c2_dp2 400 python bench.py --gpus 2 --steps 5 --warmup 2 --roofline-steps 3
step r3_b_c3 500 python bench.py --workload C3 --steps 30
step r3_b_c4 400 python bench.py --workload C4 --steps 50
step r3_b_c5 600 python bench.py --workload C5 --steps 20

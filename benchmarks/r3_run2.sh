#!/bin/bash
source benchmarks/gpu_steps.sh
step r3_t_tail 600 python -u -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -x -q --timeout 600 --timeout-method thread -k "wgrad or fcnn_layer or k2prime or whole_paired"
step r3_b2_c2 400 python bench.py --steps 200
SCPAIR_DW_TAIL=0 step r3_b2_c2_notail 400 python bench.py --steps 200 --cpu-steps 0 --no-extras
step r3_b2_c4 400 python bench.py --workload C4 --steps 100 --cpu-steps 0

set -e
B="--cpu-steps 0 --steps 200 --no-extras --roofline-steps 0"
for v in 1 "" 1 ""; do
  SCPAIR_ADAM_NOCACHE=$v timeout -k 10 200 python3 bench.py $B > gpurun_out/host_ab.json 2>/dev/null
  python3 -c "
import json
d=json.loads(open('gpurun_out/host_ab.json').read().strip().splitlines()[-1]); print('nocache=$v', round(d['value']), round(d['ms_per_step'],3), d['config']['host_enqueue_ms_per_step'], d['config']['host_enqueue_ms_by_phase'])"
done

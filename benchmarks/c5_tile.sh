set -e
B="--workload C5 --cpu-steps 0 --steps 30 --no-extras --roofline-steps 3"
for t in 128 256; do
  SCPAIR_DW_TILE=$t timeout -k 10 300 python3 bench.py $B > gpurun_out/c5_tile_$t.log 2> /dev/null
  python3 -c "
import json
d=json.loads(open('gpurun_out/c5_tile_$t.log').read().strip().splitlines()[-1]); r=d['roofline']
print($t, round(d['value']), round(d['ms_per_step'],3), round(r['avg_ms_per_step_in_kernel'],3), round(r['frac'],3))"
done

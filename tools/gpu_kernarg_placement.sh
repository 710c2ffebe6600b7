#!/bin/bash
# experiment: kernel-argument placement (HIP_FORCE_DEV_KERNARG) and its effect on the step kernel
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
for v in 0 1; do
  HIP_FORCE_DEV_KERNARG=$v timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-cold > $OUT/env_kernarg_$v.json 2> $OUT/env_kernarg_$v.err || tail -3 $OUT/env_kernarg_$v.err
  python3 -c "
import json
d=json.load(open('gpurun_out/env_kernarg_$v.json')); r=d['roofline']
print('HIP_FORCE_DEV_KERNARG=$v value %.0f'%d['value'], 'resident %.0f'%d['resident_batch']['value'], 'us/launch %.1f'%r['kernel_avg_us_hip_events'], d.get('critical_path'))
"
done

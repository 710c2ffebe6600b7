#!/bin/bash
# experiment: sample tiles per batch-side workgroup of the pipelined step (fewer, larger partial pre-gradients)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 200 python -m pytest tests/test_hip_parity.py -m gpu -q -k "counters or stepwise" > $OUT/tiles_tests.log 2>&1; tail -1 $OUT/tiles_tests.log
for t in 0 2 3; do
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-cold --pipe-tiles $t > $OUT/tiles_$t.json 2> $OUT/tiles_$t.err || tail -3 $OUT/tiles_$t.err
  python3 -c "
import json
d=json.load(open('gpurun_out/tiles_$t.json')); r=d['roofline']
print('pipe-tiles $t value %.0f'%d['value'], 'resident %.0f'%d['resident_batch']['value'], 'us/launch %.1f'%r['kernel_avg_us_hip_events'], 'final acc', d['final_accuracy'])
"
done

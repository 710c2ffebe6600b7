#!/bin/bash
# round 3, first measurement: GPU suite with the mixed-precision SVD, then C3 with and without it
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 500 python -m pytest tests -m gpu -q > $OUT/r3a_tests.log 2>&1
tail -3 $OUT/r3a_tests.log
grep -E "^(FAILED|ERROR)" $OUT/r3a_tests.log | head -30
for mode in mixed f64; do
  if [ $mode = f64 ]; then FL="--svd-f64"; else FL=""; fi
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 $FL > $OUT/r3a_bench_$mode.json 2> $OUT/r3a_bench_$mode.err || { tail -5 $OUT/r3a_bench_$mode.err; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/r3a_bench_$mode.json')); r=d['roofline']
print('$mode value %.0f'%d['value'], 'cold %.0f'%d['cold_start']['value'], 'resident %.0f'%d['resident_batch']['value'], 'step us %.1f'%r['kernel_avg_us_hip_events'], d['jacobi'], d.get('critical_path'))
"
done

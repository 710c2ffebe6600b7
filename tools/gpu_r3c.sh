#!/bin/bash
# round 3: persistent sweep -- its own parity test first (bounded), then the whole GPU suite, then the bench with and without it
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 200 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "persistent" > $OUT/r3c_persist_test.log 2>&1
rc=$?
tail -15 $OUT/r3c_persist_test.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python -m pytest tests -m gpu -q > $OUT/r3c_tests.log 2>&1
tail -3 $OUT/r3c_tests.log
grep -E "^(FAILED|ERROR)" $OUT/r3c_tests.log | head -30
for mode in persist perstep; do
  if [ $mode = perstep ]; then FL="--per-step"; else FL=""; fi
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 $FL > $OUT/r3c_bench_$mode.json 2> $OUT/r3c_bench_$mode.err || { tail -5 $OUT/r3c_bench_$mode.err; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/r3c_bench_$mode.json')); r=d['roofline']
print('$mode value %.0f'%d['value'], 'cold %.0f'%d['cold_start']['value'], 'resident %.0f'%d['resident_batch']['value'], 'step us %.1f'%r.get('step_avg_us_hip_events', r['kernel_avg_us_hip_events']), d['jacobi'], 'finite', d['finite'], 'acc', d['final_accuracy'])
"
done

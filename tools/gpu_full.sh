#!/bin/bash
# whole GPU suite, smoke, then the driver's bench command on c3 and the other configurations
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 800 python -m pytest tests -m gpu -q > $OUT/full_tests.log 2>&1
rc=$?
tail -3 $OUT/full_tests.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|E  )" $OUT/full_tests.log | head -20; exit $rc; }
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/full_smoke.log 2>&1 || { tail -5 $OUT/full_smoke.log; exit 1; }
tail -1 $OUT/full_smoke.log
for cfg in c3 c2 c5; do
  if [ $cfg = c5 ]; then SW="--steps 6 --warmup 2"; else SW="--steps 20 --warmup 5"; fi
  timeout -k 10 400 python3 bench.py --config $cfg $SW > $OUT/full_bench_$cfg.json 2> $OUT/full_bench_$cfg.err || { tail -5 $OUT/full_bench_$cfg.err; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/full_bench_$cfg.json')); r=d['roofline']
print('$cfg value %.0f'%d['value'], 'cold %.0f'%d['cold_start']['value'], 'resident %.0f'%d['resident_batch']['value'], 'cpu %.2f'%d['cpu_baseline']['value'], 'frac %.5f'%r['frac'], d['jacobi'])
"
done
# an early training window (passes 3-8) and the reference-policy case
timeout -k 10 400 python3 bench.py --steps 6 --warmup 2 > $OUT/full_bench_c3_default.json 2> $OUT/full_bench_c3_default.err || { tail -5 $OUT/full_bench_c3_default.err; exit 1; }
timeout -k 10 400 python3 bench.py --policy reference --steps 20 --warmup 5 > $OUT/full_bench_c3_refpolicy.json 2> $OUT/full_bench_c3_refpolicy.err || { tail -5 $OUT/full_bench_c3_refpolicy.err; exit 1; }
python3 -c "
import json
for f in ('c3_default','c3_refpolicy'):
    d=json.load(open('gpurun_out/full_bench_%s.json'%f)); print(f, 'value %.0f'%d['value'], 'cold %.0f'%d['cold_start']['value'], 'resident %.0f'%d['resident_batch']['value'], 'cpu %.2f'%d['cpu_baseline']['value'])
"

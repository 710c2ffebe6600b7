#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $OUT/r2_t8.log 2>&1
tail -4 $OUT/r2_t8.log
timeout -k 10 300 python3 bench.py --config c5 --steps 6 --warmup 2 --cpu-steps 10 > $OUT/r2_bench_c5.json 2> $OUT/r2_bench_c5.err
timeout -k 10 200 python3 bench.py --config c2 --steps 20 --warmup 5 --cpu-steps 40 > $OUT/r2_bench_c2.json 2> $OUT/r2_bench_c2.err
timeout -k 10 200 python3 bench.py --config c3 --policy reference --steps 20 --warmup 5 --cpu-steps 40 --ref-form-budget 0 > $OUT/r2_bench_c3ref.json 2> $OUT/r2_bench_c3ref.err
python3 - <<'PY'
import json
for f in ('r2_bench_c5','r2_bench_c2','r2_bench_c3ref'):
    try:
        d=json.load(open('gpurun_out/%s.json'%f))
        print(f, 'value %.0f'%d['value'], d['jacobi'], 'cold', d.get('cold_start',{}).get('value'), 'resident', d.get('resident_batch',{}).get('value'), 'cpu', d.get('cpu_baseline',{}).get('value'), d['roofline']['frac'])
    except Exception as e:
        print(f, 'ERR', e, open('gpurun_out/%s.err'%f).read()[-800:])
PY

#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $OUT/r2_t6.log 2>&1
tail -4 $OUT/r2_t6.log
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 > $OUT/r2_bench_b.json 2> $OUT/r2_bench_b.err
tail -c 300 $OUT/r2_bench_b.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r2_bench_b.json'))
print('value %.0f'%d['value'], d['jacobi'], 'cold', d['cold_start']['value'], 'resident', d['resident_batch']['value'])
print(d.get('critical_path'))
PY
TNML_LIB=build_exp/libtnml_fine.so timeout -k 10 200 python tools/probe_step.py 64 20 5000 2 > $OUT/r2_probe_fine12.txt 2>&1; grep -A11 "back-to-back" $OUT/r2_probe_fine12.txt

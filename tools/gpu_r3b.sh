#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
STEPS=24 timeout -k 10 300 python tools/probe_mixed_sweep.py > $OUT/r3_probe_mixed_sweep.txt 2>&1
tail -8 $OUT/r3_probe_mixed_sweep.txt | cut -c1-260
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 > $OUT/r3b_bench_mixed.json 2> $OUT/r3b_bench_mixed.err || { tail -5 $OUT/r3b_bench_mixed.err; exit 1; }
python3 -c "
import json
d=json.load(open('gpurun_out/r3b_bench_mixed.json')); r=d['roofline']
print('mixed value %.0f'%d['value'], 'cold %.0f'%d['cold_start']['value'], d['cold_start']['mixed_precision'], 'resident %.0f'%d['resident_batch']['value'], 'step us %.1f'%r['kernel_avg_us_hip_events'], d['jacobi'], d.get('critical_path'))
"

#!/bin/bash
# final validation of the round: whole GPU suite, smoke, the driver's bench command, then the C3 / C2 profile passes
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 700 python -m pytest tests -m gpu -q > $OUT/r2_final_tests.log 2>&1
tail -3 $OUT/r2_final_tests.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r2_smoke.log 2>&1; tail -1 $OUT/r2_smoke.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/r2_bench_c3.json 2> $OUT/r2_bench_c3.err
timeout -k 10 300 python3 bench.py > $OUT/r2_bench_c3_default.json 2> $OUT/r2_bench_c3_default.err
python3 -c "
import json
for f in ('r2_bench_c3','r2_bench_c3_default'):
    d=json.load(open('gpurun_out/%s.json'%f)); print(f, 'value %.0f'%d['value'], 'cold %.0f'%d['cold_start']['value'], 'resident %.0f'%d['resident_batch']['value'], 'cpu %.1f'%d['cpu_baseline']['value'], d['roofline']['frac'], d['cpu_baseline_reference_form']['forms'])
"
bash tools/run_profiles.sh c3 > $OUT/r2_prof_c3.log 2>&1
bash tools/run_profiles.sh c2 > $OUT/r2_prof_c2.log 2>&1
head -6 $OUT/prof_summary_c3.txt $OUT/prof_summary_c2.txt

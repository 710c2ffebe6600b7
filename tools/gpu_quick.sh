#!/bin/bash
# quick validation of a step-kernel change: core parity tests, the fine-stamp probe (if an experiment build is present), one bench line
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 500 python -m pytest tests/test_hip_parity.py -m gpu -q > $OUT/q_tests.log 2>&1
rc=$?
tail -3 $OUT/q_tests.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|E  )" $OUT/q_tests.log | head; exit $rc; }
if [ -f build_exp/libtnml_fs.so ]; then
  TNML_LIB=build_exp/libtnml_fs.so timeout -k 10 200 python3 tools/probe_step.py > $OUT/q_probe.txt 2>&1
  grep -A10 "back-to-back" $OUT/q_probe.txt | cut -c1-400 | grep -v "wide kernel WG0\|per wave\|per-wave"
fi
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 > $OUT/q_bench.json 2> $OUT/q_bench.err || { tail -5 $OUT/q_bench.err; exit 1; }
python3 -c "
import json
d=json.load(open('gpurun_out/q_bench.json')); r=d['roofline']
print('value %.0f'%d['value'], 'cold %.0f'%d['cold_start']['value'], 'resident %.0f'%d['resident_batch']['value'], 'us/launch %.1f'%r['kernel_avg_us_hip_events'], 'whole-run %.1f'%r.get('step_kernel_avg_us_hip_events_whole_run',0), d['jacobi'], d.get('critical_path'))
"

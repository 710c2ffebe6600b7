#!/bin/bash
# the rocprofv3 passes behind profiles/ for the three single-GPU configurations (tools/run_profiles.sh each)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
for cfg in c3 c2; do
  bash tools/run_profiles.sh $cfg > $OUT/prof_$cfg.log 2>&1 || { tail -5 $OUT/prof_$cfg.log; exit 1; }
  head -8 $OUT/prof_summary_$cfg.txt
done
bash tools/run_profiles.sh c5 > $OUT/prof_c5.log 2>&1 || { tail -5 $OUT/prof_c5.log; exit 1; }
head -8 $OUT/prof_summary_c5.txt

#!/bin/bash
# production timeline of one sweep: rocprofv3 kernel trace of back-to-back passes against the per-step Jacobi statistics
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/tl_trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_trace -- python3 $ROOT/tools/probe_timeline.py --trace > $OUT/tl_trace.log 2>&1 || { tail -5 $OUT/tl_trace.log; exit 1; }
cd $ROOT
timeout -k 10 300 python3 tools/probe_timeline.py --rounds $OUT/tl_rounds.json > $OUT/tl_rounds.log 2>&1 || { tail -5 $OUT/tl_rounds.log; exit 1; }
T=$(find $OUT/tl_trace -name '*_kernel_trace.csv' | head -1)
python3 tools/probe_timeline.py --merge $T $OUT/tl_rounds.json | tee $OUT/tl_merge${TL_CLASSIC:+_classic}.txt
find $OUT/tl_trace -name '*_kernel_trace.csv' -delete

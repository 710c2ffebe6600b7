#!/bin/bash
# On the GPU box: the rocprofv3 passes behind profiles/ (kernel trace + stats of the DRIVER's bench command, then one --pmc
# pass per counter group on a short run).  Output under gpurun_out/prof_*; summarise with tools/rocprof_summary.py.
# Usage: tools/run_profiles.sh [config] [extra bench flags]
# The passes are chained with && : after a pass that was killed at its limit nothing further runs on that GPU.  The --pmc
# passes drain the stream every 4 sweep steps (--sync-interval): rocprofv3's counter collection keeps per-dispatch state and
# was overrun by the tens of thousands of queued launches of a C5 sweep in round 1 (SIGSEGV / malformed AQL packet).
CFG=${1:-c3}
shift
EXTRA="$@"
# kernel-trace pass = the driver's command; c5 (0.3 s per pass, 14 launches per step) with 6 / 2 passes
if [ "$CFG" = c5 ]; then KT="--steps 6 --warmup 2"; else KT="--steps 20 --warmup 5"; fi
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_kt_$CFG $OUT/prof_fetch_$CFG $OUT/prof_write_$CFG $OUT/prof_mfma_$CFG
# (--event-handoffs: under --pmc only one kernel runs at a time -- a kernel that polls for another stream's sequence number would wait
#  for a producer that cannot start)
SHORT="--config $CFG --steps 2 --warmup 1 --cpu-steps 0 --no-kernel-profile --no-cold --no-resident --sync-interval 4 --event-handoffs $EXTRA"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kt_$CFG -- python3 $ROOT/bench.py --config $CFG $KT --cpu-steps 0 $EXTRA > $OUT/prof_kt_$CFG.json 2> $OUT/prof_kt_$CFG.err && \
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_fetch_$CFG -- python3 $ROOT/bench.py $SHORT > /dev/null 2> $OUT/prof_fetch_$CFG.err && \
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_write_$CFG -- python3 $ROOT/bench.py $SHORT > /dev/null 2> $OUT/prof_write_$CFG.err && \
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $OUT/prof_mfma_$CFG -- python3 $ROOT/bench.py $SHORT > /dev/null 2> $OUT/prof_mfma_$CFG.err
echo "passes rc $?"
cd $ROOT
python3 tools/rocprof_summary.py $CFG > $OUT/prof_summary_$CFG.txt 2>&1
# gpurun brings back at most 64 MiB: the raw per-dispatch traces (hundreds of thousands of rows at C5) stay on the box
find $OUT/prof_kt_$CFG $OUT/prof_fetch_$CFG $OUT/prof_write_$CFG $OUT/prof_mfma_$CFG -name '*_kernel_trace.csv' -delete 2>/dev/null
find $OUT/prof_fetch_$CFG $OUT/prof_write_$CFG $OUT/prof_mfma_$CFG -name '*_counter_collection.csv' -delete 2>/dev/null
tail -30 $OUT/prof_summary_$CFG.txt

#!/bin/bash
# round 2, GPU call 1: new parity tests on the existing kernels, then C5 with launch checking, then one C5 --pmc pass
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_true_shapes_gpu.py -m gpu -x -q -s > $OUT/r2_t1.log 2>&1
echo "pytest rc $?" >> $OUT/r2_t1.log
tail -5 $OUT/r2_t1.log
timeout -k 10 200 python3 bench.py --config c5 --steps 1 --warmup 1 --cpu-steps 0 --no-kernel-profile --no-cold --check-launches > $OUT/r2_c5_check.json 2> $OUT/r2_c5_check.err && \
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/r2_pmc_write_c5 -- python3 $ROOT/bench.py --config c5 --steps 1 --warmup 0 --cpu-steps 0 --no-kernel-profile --no-cold --sync-interval 4 > $OUT/r2_pmc_write_c5.json 2> $OUT/r2_pmc_write_c5.err ; echo "pmc rc $?" )
tail -3 $OUT/r2_c5_check.err $OUT/r2_pmc_write_c5.err

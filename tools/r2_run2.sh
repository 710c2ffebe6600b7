#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 500 python -m pytest tests/test_true_shapes_gpu.py -m gpu -q -s -k "true_shape or n196" > $OUT/r2_t5.log 2>&1
tail -4 $OUT/r2_t5.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/r2_bench_pipe.json 2> $OUT/r2_bench_pipe.err
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --classic --cpu-steps 0 --no-cold > $OUT/r2_bench_classic.json 2> $OUT/r2_bench_classic.err
tail -3 $OUT/r2_bench_pipe.err $OUT/r2_bench_classic.err

#!/usr/bin/env bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -x -q > gpurun_out/r2s2_t11.log 2>&1
tail -4 gpurun_out/r2s2_t11.log
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_fp32g
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o fp32 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > $OUT/bench.json 2> $OUT/err.log
rm -f $OUT/*kernel_trace.csv $OUT/*.db
cut -c60-200 $OUT/bench.json

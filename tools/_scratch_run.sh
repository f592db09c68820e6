#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_pack_cache.py tests/test_gpu_bf16s.py tests/test_gpu_bf16s_v3.py tests/test_gpu_bf16.py tests/test_gpu_unet.py -m gpu -x -q > gpurun_out/pk_test.log 2>&1 || { tail -30 gpurun_out/pk_test.log; exit 1; }
tail -2 gpurun_out/pk_test.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pk -o pk -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > /tmp/prof_pk.log 2>&1
grep -h "pack_many\|head_\|c3\|direct" $(find /tmp/prof_pk -name '*kernel_stats.csv' | head -1) | cut -c1-140
tail -1 /tmp/prof_pk.log | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pk2 -o pk -- python3 $GRAFT_REPO_ROOT/bench.py --dtype bf16 --size 512 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > /tmp/prof_pk2.log 2>&1
grep -h "pack_many\|head_\|c3\|direct" $(find /tmp/prof_pk2 -name '*kernel_stats.csv' | head -1) | cut -c1-140
tail -1 /tmp/prof_pk2.log | cut -c1-200

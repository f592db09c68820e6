#!/usr/bin/env bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_interunet.py tests/test_gpu_smallunet.py -x -q > gpurun_out/r2s2_t19.log 2>&1
tail -25 gpurun_out/r2s2_t19.log

#!/usr/bin/env bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_gunet.py -x -q -k "conv_context or vgg or context" > gpurun_out/r2s2_t6.log 2>&1
tail -25 gpurun_out/r2s2_t6.log

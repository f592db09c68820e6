#!/usr/bin/env bash
set -o pipefail
mkdir -p gpurun_out
UNETK_DIST_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2s2_2rank.json 2> gpurun_out/r2s2_2rank.err || tail -20 gpurun_out/r2s2_2rank.err
cut -c1-700 gpurun_out/r2s2_2rank.json

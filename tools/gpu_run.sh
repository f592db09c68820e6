#!/usr/bin/env bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_lits.py tests/test_gpu_evaluator.py -q > gpurun_out/r2_lits.log 2>&1
tail -25 gpurun_out/r2_lits.log

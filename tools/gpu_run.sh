#!/usr/bin/env bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_lits.py -q -k "entry" > gpurun_out/r2_entry.log 2>&1
tail -30 gpurun_out/r2_entry.log

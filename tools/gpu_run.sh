#!/usr/bin/env bash
# Scratch runner for one-off GPU commands:  gpurun -- 'bash tools/gpu_run.sh'.  Edit, run, do not rely on its content.
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -5

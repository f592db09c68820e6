#!/usr/bin/env bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_gunet.py tests/test_gpu_lgnet.py tests/test_gpu_lits.py tests/test_gpu_lits_eval.py tests/test_gpu_evaluator.py tests/test_gpu_tf_checkpoint.py -q > gpurun_out/r2_gunet.log 2>&1
tail -30 gpurun_out/r2_gunet.log

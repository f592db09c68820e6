#!/usr/bin/env bash
# Scratch runner for one-off GPU commands:  gpurun -- 'bash tools/gpu_run.sh'.  Edit, run, do not rely on its content.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/full_gpu.log 2>&1; tail -3 gpurun_out/full_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --steps 10 --warmup 3 > gpurun_out/bench_n1_final.json 2> gpurun_out/bench_n1_final.err; tail -c 1500 gpurun_out/bench_n1_final.json | head -c 1500

#!/usr/bin/env bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2_gpu_tests_b.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_gpu_tests_b.log; tail -4 gpurun_out/r2_gpu_tests_b.log
python bench.py --dtype bf16 --size 512 --batch 8 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_b16s_512_final.json 2> gpurun_out/r2_b16s_512.err || tail -5 gpurun_out/r2_b16s_512.err
cut -c1-330 gpurun_out/r2_b16s_512_final.json
python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2_b16s_256.json 2>> gpurun_out/r2_b16s_512.err; cut -c90-330 gpurun_out/r2_b16s_256.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2_fp32_check.json 2>> gpurun_out/r2_b16s_512.err; cut -c90-330 gpurun_out/r2_fp32_check.json

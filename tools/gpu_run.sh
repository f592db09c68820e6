#!/usr/bin/env bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2s2_full.log 2>&1
tail -5 gpurun_out/r2s2_full.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2s2_fp32.json 2> gpurun_out/r2s2.err || tail -5 gpurun_out/r2s2.err
cut -c60-200 gpurun_out/r2s2_fp32.json
python bench.py --dtype bf16 --size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2s2_bf16.json 2> gpurun_out/r2s2.err || tail -5 gpurun_out/r2s2.err
cut -c60-200 gpurun_out/r2s2_bf16.json
python bench.py --model UNet3D --size 96 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2s2_u3d1.json 2> gpurun_out/r2s2.err || tail -5 gpurun_out/r2s2.err
cut -c60-200 gpurun_out/r2s2_u3d1.json

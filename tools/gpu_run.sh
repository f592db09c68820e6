#!/usr/bin/env bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_ops3d.py tests/test_gpu_unet3d.py tests/test_gpu_fullsize.py tests/test_gpu_ops.py -x -q > gpurun_out/r2s2_t18.log 2>&1
tail -3 gpurun_out/r2s2_t18.log
python bench.py --model UNet3D --size 96 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --detail > gpurun_out/r2s2_u3d1n.json 2> gpurun_out/r2s2.err || tail -5 gpurun_out/r2s2.err
cut -c60-200 gpurun_out/r2s2_u3d1n.json

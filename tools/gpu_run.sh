#!/usr/bin/env bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ops3d.py tests/test_gpu_unet3d.py tests/test_gpu_unet3d_v2.py tests/test_gpu_smallunet.py tests/test_gpu_interunet.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r2s2_t12.log 2>&1
tail -6 gpurun_out/r2s2_t12.log
python bench.py --model UNet3D --size 96 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --detail > gpurun_out/r2s2_u3d1e.json 2> gpurun_out/r2s2.err || tail -5 gpurun_out/r2s2.err
cut -c60-200 gpurun_out/r2s2_u3d1e.json

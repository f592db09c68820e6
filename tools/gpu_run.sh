#!/usr/bin/env bash
set -o pipefail
mkdir -p gpurun_out
for w8 in 0 1; do for sk in 0 1; do
UNETK_LIN_W8=$w8 UNETK_LIN_SK=$sk python bench.py --model UNet3D --size 96 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --detail > gpurun_out/r2s2_u3d_w${w8}_sk${sk}.json 2> gpurun_out/r2s2_u3d.err || tail -5 gpurun_out/r2s2_u3d.err
echo "w8=$w8 sk=$sk"; cut -c60-130 gpurun_out/r2s2_u3d_w${w8}_sk${sk}.json
done; done
timeout -k 10 900 python -m pytest tests/test_gpu_unet.py tests/test_dp.py tests/test_gpu_unet3d.py tests/test_gpu_golden.py tests/test_gpu_tf_checkpoint.py -x -q > gpurun_out/r2s2_t4.log 2>&1
tail -6 gpurun_out/r2s2_t4.log

#!/usr/bin/env bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_pack_cache.py tests/test_gpu_unet.py tests/test_gpu_unet3d.py tests/test_gpu_bf16s.py tests/test_gpu_bf16.py tests/test_dp.py tests/test_gpu_tf_checkpoint.py tests/test_gpu_golden.py tests/test_gpu_smallunet.py -x -q > gpurun_out/r2s2_t15.log 2>&1
tail -12 gpurun_out/r2s2_t15.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2s2_fp32i.json 2> gpurun_out/r2s2.err || tail -5 gpurun_out/r2s2.err
cut -c60-200 gpurun_out/r2s2_fp32i.json
python bench.py --dtype bf16 --size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2s2_bf16i.json 2> gpurun_out/r2s2.err || tail -5 gpurun_out/r2s2.err
cut -c60-200 gpurun_out/r2s2_bf16i.json
python bench.py --model UNet3D --size 96 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2s2_u3d1i.json 2> gpurun_out/r2s2.err || tail -5 gpurun_out/r2s2.err
cut -c60-200 gpurun_out/r2s2_u3d1i.json

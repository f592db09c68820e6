#!/usr/bin/env bash
set -o pipefail
mkdir -p gpurun_out
python bench.py --model UNet3D --size 96 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --detail > gpurun_out/r2s2_u3d1l.json 2> gpurun_out/r2s2.err || tail -5 gpurun_out/r2s2.err
cut -c60-200 gpurun_out/r2s2_u3d1l.json
python bench.py --model UNet3D --size 96 --batch 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2s2_u3d2l.json 2> gpurun_out/r2s2.err || tail -5 gpurun_out/r2s2.err
cut -c60-200 gpurun_out/r2s2_u3d2l.json

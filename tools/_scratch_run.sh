#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py tests/test_gpu_bf16s.py tests/test_gpu_golden.py tests/test_gpu_gunet.py -m gpu -x -q > gpurun_out/hd_test.log 2>&1 || { tail -30 gpurun_out/hd_test.log; exit 1; }
tail -2 gpurun_out/hd_test.log
for f in 1 2; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/hd_fp32_$f.json 2>gpurun_out/hd_err.log
  python bench.py --dtype bf16 --size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/hd_bf16_$f.json 2>gpurun_out/hd_err.log
  python - <<PY
import json
for m in ('fp32','bf16'):
    d=json.loads(open('gpurun_out/hd_%s_$f.json'%m).read().strip().splitlines()[-1])
    print(m, 'run $f', d['value'], d['ms_per_step'], [(k['kernel'][:24], k['avg_launch_ms']) for k in d['hbm_kernels'] if 'head' in k['kernel']])
PY
done

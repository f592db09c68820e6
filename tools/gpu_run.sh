#!/usr/bin/env bash
# Scratch runner for one-off GPU commands:  gpurun -- 'bash tools/gpu_run.sh'.  Edit, run, do not rely on its content.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_bf16s.py -m gpu -x -q 2>&1 | tail -3
python bench.py --dtype bf16 --size 512 --batch 8 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/e2.json 2> gpurun_out/e1.err
python - <<PY
import json
d=json.loads(open("gpurun_out/e2.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["whole_step_frac_of_dtype_peak"])
for k in d["kernels"][:9]:
    print("%-80s n %3d %.4f ms %7.1f TF tot %.3f" % (k["kernel"][:80], k["launches"], k["avg_launch_ms"], k["achieved_tflops"], k["total_ms_per_step"]))
PY

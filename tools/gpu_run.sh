#!/usr/bin/env bash
# Scratch runner for one-off GPU commands:  gpurun -- 'bash tools/gpu_run.sh'.  Edit, run, do not rely on its content.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ops3d.py -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --detail > gpurun_out/e1.json 2> gpurun_out/e1.err
python - <<PY
import json
d=json.loads(open("gpurun_out/e1.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for k in d["kernels"]:
    if "wgrad_kernel" in k["kernel"]: print("%-75s %.4f ms %.1f TF" % (k["kernel"], k["avg_launch_ms"], k["achieved_tflops"]))
PY

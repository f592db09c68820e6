set -o pipefail
mkdir -p gpurun_out/r3d
timeout -k 10 900 python -m pytest tests/test_gpu_bf16_e2e.py -q -s > gpurun_out/r3d/test_e2e.log 2>&1; echo "e2e rc=$?"; grep -n "trajectory\|signed mean\|passed\|failed\|Error\|assert" gpurun_out/r3d/test_e2e.log | head -20

set -e
mkdir -p gpurun_out/r5f
timeout -k 10 900 python -m pytest tests/test_gpu_unet3d.py -x -q > gpurun_out/r5f/pytest.log 2>&1 || { tail -60 gpurun_out/r5f/pytest.log; exit 1; }
tail -3 gpurun_out/r5f/pytest.log
bash tools/ab_run.sh s2lin "--model UNet3D --size 96 --batch 1 --steps 8 --warmup 2" base:UNETK_S2LIN=0 base:UNETK_S2LIN=1
ROUNDS=1 bash tools/ab_run.sh s2lin_d "--model UNet3D --size 96 --batch 1 --steps 8 --warmup 2 --detail" base:UNETK_S2LIN=0 base:UNETK_S2LIN=1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab/s2lin_d/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['value'])
    for r in d['kernels']:
        if 's12' in r['kernel'] or 's22' in r['kernel']: print('   %-60s %.4f ms %6.1f TF' % (r['kernel'], r['avg_launch_ms'], r['achieved_tflops']))
PY

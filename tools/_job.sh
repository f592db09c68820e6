# scratch job file for `gpurun -- 'bash tools/_job.sh'` (overwritten per experiment; the round's experiments are recorded in profiles/)
set -e
mkdir -p gpurun_out/job
timeout -k 10 500 python -m pytest tests/test_gpu_ops3d.py tests/test_gpu_unet3d.py tests/test_gpu_unet3d_v2.py -q -m gpu -x > gpurun_out/job/pytest.log 2>&1 || { tail -60 gpurun_out/job/pytest.log; exit 1; }
tail -3 gpurun_out/job/pytest.log
ROUNDS=2 bash tools/ab_run.sh kskip2 "--model UNet3D --size 96 --batch 1 --steps 8 --warmup 2" base:UNETK_KSKIP=0 base:UNETK_KSKIP=3
ROUNDS=1 bash tools/ab_run.sh kskip2b "--model UNet3D --size 96 --batch 2 --steps 6 --warmup 2" base:UNETK_KSKIP=0 base:UNETK_KSKIP=3
ROUNDS=1 bash tools/ab_run.sh kskip2c "--model UNet3D --depth 10 --size 256 --batch 4 --steps 6 --warmup 2" base:UNETK_KSKIP=0 base:UNETK_KSKIP=3

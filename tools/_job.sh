# scratch job file for `gpurun -- 'bash tools/_job.sh'` (overwritten per experiment; the round's experiments are recorded in profiles/)
set -e
mkdir -p gpurun_out/job
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ops3d.py tests/test_gpu_unet.py tests/test_gpu_unet3d.py tests/test_gpu_smallunet.py -q -m gpu -x > gpurun_out/job/pytest.log 2>&1 || { tail -30 gpurun_out/job/pytest.log; exit 1; }
tail -1 gpurun_out/job/pytest.log
ROUNDS=3 bash tools/ab_run.sh zp2_head "--steps 20 --warmup 5" prev base
ROUNDS=2 bash tools/ab_run.sh zp2_u3d "--model UNet3D --size 96 --batch 1 --steps 8 --warmup 2" prev base
ROUNDS=2 bash tools/ab_run.sh zp2_gunet "--model GUNet --size 256 --batch 8 --steps 10 --warmup 3" prev base
ROUNDS=1 bash tools/ab_run.sh zp2_infer "--mode infer --steps 24 --warmup 8" prev base

set -e
mkdir -p gpurun_out/r5k
timeout -k 10 900 python -m pytest tests/test_gpu_side_wgrad.py tests/test_gpu_ops.py tests/test_gpu_unet3d.py tests/test_gpu_unet.py -x -q > gpurun_out/r5k/pytest.log 2>&1 || { tail -40 gpurun_out/r5k/pytest.log; exit 1; }
tail -3 gpurun_out/r5k/pytest.log
ROUNDS=2 bash tools/ab_run.sh sdec_u3d "--model UNet3D --size 96 --batch 1 --steps 8 --warmup 2" base:UNETK_SIDE_DECONV=0 base:UNETK_SIDE_DECONV=1
ROUNDS=2 bash tools/ab_run.sh sdec_unet "--steps 10 --warmup 3" base:UNETK_SIDE_DECONV=0 base:UNETK_SIDE_DECONV=1
ROUNDS=1 bash tools/ab_run.sh sdec_gunet "--model GUNet --batch 8 --steps 10 --warmup 3" base:UNETK_SIDE_DECONV=0 base:UNETK_SIDE_DECONV=1 base:UNETK_SIDE_DECONV=0 base:UNETK_SIDE_DECONV=1
ROUNDS=1 bash tools/ab_run.sh sdec_u3d10 "--model UNet3D --depth 10 --size 256 --batch 4 --steps 8 --warmup 2" base:UNETK_SIDE_DECONV=0 base:UNETK_SIDE_DECONV=1

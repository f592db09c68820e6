set -e
mkdir -p gpurun_out/r5l
timeout -k 10 900 python -m pytest tests/test_gpu_pack_cache.py tests/test_gpu_unet.py tests/test_gpu_side_wgrad.py tests/test_gpu_bf16_e2e.py -x -q > gpurun_out/r5l/pytest.log 2>&1 || { tail -40 gpurun_out/r5l/pytest.log; exit 1; }
tail -3 gpurun_out/r5l/pytest.log
ROUNDS=2 bash tools/ab_run.sh pf_u3d "--model UNet3D --size 96 --batch 1 --steps 8 --warmup 2" base:UNETK_PACK_PREFETCH=0 base:UNETK_PACK_PREFETCH=1
ROUNDS=2 bash tools/ab_run.sh pf_gunet "--model GUNet --batch 8 --steps 10 --warmup 3" base:UNETK_PACK_PREFETCH=0 base:UNETK_PACK_PREFETCH=1
ROUNDS=2 bash tools/ab_run.sh pf_unet "--steps 10 --warmup 3" base:UNETK_PACK_PREFETCH=0 base:UNETK_PACK_PREFETCH=1
ROUNDS=2 bash tools/ab_run.sh pf_bf16 "--dtype bf16 --size 512 --batch 8 --steps 20 --warmup 5 --no-kernel-events" base:UNETK_PACK_PREFETCH=0 base:UNETK_PACK_PREFETCH=1

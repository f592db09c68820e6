# scratch job file for `gpurun -- 'bash tools/_job.sh'` (overwritten per experiment; the round's experiments are recorded in profiles/)
set -e
mkdir -p gpurun_out/job
timeout -k 10 500 python -m pytest tests/test_gpu_ops3d.py tests/test_gpu_unet3d.py tests/test_gpu_wgrad_stacked.py -q -m gpu -x > gpurun_out/job/pytest.log 2>&1 || { tail -60 gpurun_out/job/pytest.log; exit 1; }
tail -3 gpurun_out/job/pytest.log
ROUNDS=3 bash tools/ab_run.sh s2w6 "--model UNet3D --size 96 --batch 1 --steps 8 --warmup 2" base:UNETK_WG_S2W6=0 base:UNETK_WG_S2W6=1
UNETK_WG_S2W6=0 python bench.py --model UNet3D --size 96 --batch 1 --steps 4 --warmup 2 --no-cpu-baseline --detail > gpurun_out/job/detail_w6_0.json
UNETK_WG_S2W6=1 python bench.py --model UNet3D --size 96 --batch 1 --steps 4 --warmup 2 --no-cpu-baseline --detail > gpurun_out/job/detail_w6_1.json

set -e
mkdir -p gpurun_out/r5e
timeout -k 10 900 python -m pytest tests/test_gpu_semantics_r4.py tests/test_gpu_evaluator.py tests/test_dp.py tests/test_gpu_unet3d.py tests/test_gpu_side_wgrad.py -x -q > gpurun_out/r5e/pytest.log 2>&1 || { tail -40 gpurun_out/r5e/pytest.log; exit 1; }
tail -3 gpurun_out/r5e/pytest.log
for r in 1 2; do
  timeout -k 10 300 python bench.py --dp-rehearsal --dtype bf16 --size 512 --batch 8 --steps 20 --warmup 4 --no-cpu-baseline --no-kernel-events > gpurun_out/r5e/dp_bf16_$r.json 2> gpurun_out/r5e/dp_bf16_$r.err
  python - gpurun_out/r5e/dp_bf16_$r.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
dp=d['data_parallel']
print("dp bf16: ms %.3f compute_only %.3f eff %.4f exposed %s buckets %s fired %s" % (d['ms_per_step'], dp['compute_only_ms_per_step'], dp['dp_efficiency_vs_compute_only'], dp['allreduce_exposed_ms'], dp['buckets'], dp['buckets_fired_in_backward']))
PY
done
bash tools/ab_run.sh side3d "--model UNet3D --size 96 --batch 1 --steps 8 --warmup 2" base:UNETK_SIDE_WGRAD3D_FIRST=0 base:UNETK_SIDE_WGRAD3D_FIRST=1
bash tools/ab_run.sh side3d_b2 "--model UNet3D --size 96 --batch 2 --steps 8 --warmup 2" base:UNETK_SIDE_WGRAD3D_FIRST=0 base:UNETK_SIDE_WGRAD3D_FIRST=1

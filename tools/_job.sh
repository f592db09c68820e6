set -e
mkdir -p gpurun_out/r5b
timeout -k 10 600 python -m pytest tests/test_gpu_train_dice.py tests/test_gpu_lits_loader.py -x -q -s --durations=5 > gpurun_out/r5b/pytest.log 2>&1 || { tail -50 gpurun_out/r5b/pytest.log; exit 1; }
tail -15 gpurun_out/r5b/pytest.log
grep -E "Dice|UNet" gpurun_out/r5b/pytest.log | head -10
timeout -k 10 400 python bench.py --steps 10 --warmup 3 > gpurun_out/r5b/bench.json 2> gpurun_out/r5b/bench.err
python -c "
import json
d=json.loads(open('gpurun_out/r5b/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('dice_vs_oracle'), d['cpu_baseline']['wall_s'])
"

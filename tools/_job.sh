# scratch job file for `gpurun -- 'bash tools/_job.sh'` (overwritten per experiment; the round's experiments are recorded in profiles/)
set -e
mkdir -p gpurun_out/job
timeout -k 10 600 python -m pytest tests/test_bench_launch.py tests/test_dp.py -q -m gpu -x > gpurun_out/job/pytest.log 2>&1 || { tail -30 gpurun_out/job/pytest.log; exit 1; }
tail -2 gpurun_out/job/pytest.log
for r in 1 2; do
timeout -k 10 300 python bench.py --dp-rehearsal --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/job/dp_fp32_$r.json
timeout -k 10 300 python bench.py --dp-rehearsal --dtype bf16 --size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/job/dp_bf16_$r.json
timeout -k 10 300 python bench.py --dp-rehearsal --dtype bf16 --size 512 --batch 8 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events > gpurun_out/job/dp_bf16_noev_$r.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/job/dp_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); p=d['data_parallel']
    print(f, d['ms_per_step'], p['compute_only_ms_per_step'], p.get('dp_again_ms_per_step'), p['dp_efficiency_vs_compute_only'], p.get('dp_efficiency_drift_cancelled'), p.get('dp_again_error'))
PY

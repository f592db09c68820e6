# scratch job file for `gpurun -- 'bash tools/_job.sh'` (overwritten per experiment; the round's experiments are recorded in profiles/)
set -e
mkdir -p gpurun_out/job
timeout -k 10 1100 python -m pytest tests -q -m gpu --durations=15 > gpurun_out/job/pytest.log 2>&1 || { tail -30 gpurun_out/job/pytest.log; exit 1; }
tail -22 gpurun_out/job/pytest.log
python -c "import __graft_entry__ as g; g.smoke()"
SECONDS=0
python bench.py > gpurun_out/job/bench_default.json
echo "default bench.py took $SECONDS s"
python -c "
import json; d=json.loads(open('gpurun_out/job/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['cpu_baseline'], d.get('dice_vs_oracle'))"

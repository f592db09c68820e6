# scratch job file for `gpurun -- 'bash tools/_job.sh'` (overwritten per experiment; the round's experiments are recorded in profiles/)
set -e
mkdir -p gpurun_out/job
timeout -k 10 600 python -m pytest tests/test_gpu_gunet_combos.py -q -m gpu > gpurun_out/job/pytest.log 2>&1 || { tail -60 gpurun_out/job/pytest.log; exit 1; }
tail -3 gpurun_out/job/pytest.log

# scratch job file for `gpurun -- 'bash tools/_job.sh'` (overwritten per experiment; the round's experiments are recorded in profiles/)
set -e
mkdir -p gpurun_out/job
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/job/pytest.log 2>&1 || { tail -30 gpurun_out/job/pytest.log; exit 1; }
tail -2 gpurun_out/job/pytest.log
python -c "import __graft_entry__ as g; g.smoke()"

# scratch job file for `gpurun -- 'bash tools/_job.sh'` (overwritten per experiment; the round's experiments are recorded in profiles/)
set -e
mkdir -p gpurun_out/job
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/job/pytest.log 2>&1 || { tail -30 gpurun_out/job/pytest.log; exit 1; }
tail -3 gpurun_out/job/pytest.log
python -c "import __graft_entry__ as g; g.smoke()"
ROUNDS=2 bash tools/ab_run.sh lin2d_b "--model GUNet --size 256 --batch 8 --steps 10 --warmup 3" base:UNETK_LIN_2D=0 base

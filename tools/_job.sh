set -e
mkdir -p gpurun_out/refresh
T="timeout -k 10 300"
$T python bench.py --dp-rehearsal --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/refresh/bench_n1_dp_rehearsal.json
$T python bench.py --dp-rehearsal --dtype bf16 --size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/refresh/bench_bf16_512_bs8_dp_rehearsal.json
$T python bench.py --dp-rehearsal --dtype bf16 --size 512 --batch 8 --steps 20 --warmup 4 --no-cpu-baseline --no-kernel-events > gpurun_out/refresh/bench_bf16_512_bs8_dp_rehearsal_noevents.json
python - <<'PY'
import json
for n in ('bench_n1_dp_rehearsal','bench_bf16_512_bs8_dp_rehearsal','bench_bf16_512_bs8_dp_rehearsal_noevents'):
    d=json.loads(open('gpurun_out/refresh/%s.json'%n).read().strip().splitlines()[-1]); dp=d['data_parallel']
    print(n, d['ms_per_step'], dp['compute_only_ms_per_step'], dp['dp_efficiency_vs_compute_only'], dp['buckets'], dp['allreduce_exposed_ms'])
PY

set -o pipefail
mkdir -p gpurun_out/r3c
timeout -k 10 600 python -m pytest tests/test_gpu_bf16s_v3.py -q > gpurun_out/r3c/test_v3.log 2>&1; echo "v3 tests rc=$?"; tail -4 gpurun_out/r3c/test_v3.log
for f in 0 1 2; do
UNETK_V3_FLAGS=$f timeout -k 10 300 python bench.py --dtype bf16 --size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --detail > gpurun_out/r3c/bench_detail_f$f.json 2> gpurun_out/r3c/bench_f$f.err; echo "bench rc=$?"
UNETK_V3_FLAGS=$f timeout -k 10 300 python bench.py --dtype bf16 --size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3c/bench_plain_f$f.json 2>> gpurun_out/r3c/bench_f$f.err; echo "bench rc=$?"
done
python - <<'PY'
import json
for f in (0,1,2):
    d=json.load(open('gpurun_out/r3c/bench_plain_f%d.json'%f)); print(f, d['value'], d['ms_per_step'], d.get('whole_step_frac_of_dtype_peak'))
PY

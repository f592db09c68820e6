set -o pipefail
mkdir -p gpurun_out/r3j
timeout -k 10 900 python -m pytest tests/test_gpu_lits.py tests/test_gpu_ops.py tests/test_gpu_bf16s.py tests/test_gpu_unet.py tests/test_bench_launch.py -m gpu -q > gpurun_out/r3j/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r3j/tests.log
for dt in bf16 fp32; do
if [ $dt = bf16 ]; then A="--dtype bf16 --size 512 --batch 8"; else A=""; fi
timeout -k 10 300 python bench.py $A --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3j/bench_$dt.json 2> gpurun_out/r3j/bench_$dt.err; echo "bench $dt rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/r3j/bench_$dt.json')); print("$dt", d['value'], d['ms_per_step'], d.get('whole_step_frac_of_dtype_peak'), d['roofline']['kernel'], d['roofline']['frac'])
for k in d.get('hbm_kernels',[]): print("   ", k['kernel'], k['avg_launch_ms'], k['achieved_gbps'])
PY
done

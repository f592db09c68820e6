set -e
bash tools/ab_run.sh wg16 "--dtype bf16 --size 512 --batch 8 --steps 10 --warmup 3 --detail" base wg16
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab/wg16/*_r2.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    def tot(pred): return sum(x['total_ms_per_step'] for x in d['kernels'] if pred(x['kernel']))
    print(f.split('/')[-1], d['ms_per_step'], 'wgrad', round(tot(lambda k:'wgrad_bf16s' in k),3), '<8>', round(tot(lambda k:'kernel<8' in k),3), '<4>', round(tot(lambda k:'<4>' in k),3))
    for r in d['kernels']:
        if 'wgrad_bf16s' in r['kernel'] and ('64->64' in r['kernel'] or '512->512' in r['kernel'] or '128->128' in r['kernel']): print('   ', r['kernel'], r['avg_launch_ms'])
PY
ROUNDS=2 bash tools/ab_run.sh wg16_plain "--dtype bf16 --size 512 --batch 8 --steps 20 --warmup 5 --no-kernel-events" base wg16

set -e
ROUNDS=1 bash tools/ab_run.sh lin2d_gunet2 "--model GUNet --batch 8 --steps 10 --warmup 3 --detail" base base:UNETK_LIN_2D=1 base:UNETK_LIN_2D=15 base:UNETK_LIN_2D=7 base
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab/lin2d_gunet2/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], d['value'])
    for r in d['kernels']:
        if '8x32x32' in r['kernel'] and ('fwd' in r['kernel'] or 'dgrad' in r['kernel']) and 'deconv' not in r['kernel'] and 'pw_' not in r['kernel']: print('   %-75s %.4f %6.1f' % (r['kernel'], r['avg_launch_ms'], r['achieved_tflops']))
PY

mkdir -p gpurun_out/r5j
sample() { while true; do rocm-smi --showpower --showclocks --csv 2>/dev/null | tail -n +2 | head -2 | tr '\n' ' ' >> "$1"; echo >> "$1"; sleep 0.2; done; }
for rep in 1 2; do
for v in 0 1; do
  sample gpurun_out/r5j/smi_ldw${v}_$rep.csv & SP=$!
  UNETK_LIB=$PWD/ab/ldw/libunetk.so UNETK_V3_LDW=$v timeout -k 10 300 python bench.py --dtype bf16 --size 512 --batch 8 --steps 600 --warmup 10 --no-cpu-baseline --no-kernel-events > gpurun_out/r5j/bench_ldw${v}_$rep.json 2> gpurun_out/r5j/bench_ldw${v}_$rep.err
  kill $SP; wait $SP 2>/dev/null
done
done
python - <<'PY'
import re,statistics,json,glob
for f in sorted(glob.glob('gpurun_out/r5j/smi_*.csv')):
    rows=[]
    for l in open(f):
        m=re.findall(r'\((\d+)Mhz\)', l); p=re.findall(r',([\d.]+)\s*$', l.strip())
        if len(m)>=3 and p: rows.append((int(m[2]), float(p[0])))
    busy=[r for r in rows if r[1]>900]
    d=json.loads(open(f.replace('smi_','bench_').replace('.csv','.json')).read().strip().splitlines()[-1])
    print(f.split('/')[-1], 'ms/step %.3f' % d['ms_per_step'], 'n', len(busy), 'sclk median', statistics.median([b[0] for b in busy]), 'power median', statistics.median([b[1] for b in busy]))
PY

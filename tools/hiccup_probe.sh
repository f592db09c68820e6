#!/usr/bin/env bash
# Is this box one of those on which a timed step now and then takes 0.5 s (profiles/r05_step_hiccups.txt)?  Three headline runs to find
# out; on an affected box, alternate the side-stream filter gradients on / off.  gpurun -- 'bash tools/hiccup_probe.sh'
set -e
OUT=gpurun_out/hic2; mkdir -p $OUT
run() { e=$1; shift; env $e timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events "$@" 2>/dev/null | tail -1; }
val() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f %.0f' % (d['value'], d['ms_per_step_max_hipevents']), d.get('settle'))"; }
hit=0
for i in 1 2 3; do r=$(run X=1 --settle-seconds 0 | val); echo "detect $i: $r"; m=$(echo $r | cut -d' ' -f2); [ "$m" -gt 150 ] && hit=1; done
for i in 1 2 3; do r=$(run X=1 | val); echo "with settle phase $i: $r"; done
[ $hit = 0 ] && { echo "no hiccup on this box"; exit 0; }
for i in 1 2 3 4 5; do
  echo "side auto: $(run UNETK_SIDE_WGRAD=auto | val)    side off: $(run UNETK_SIDE_WGRAD=0 | val)"
done

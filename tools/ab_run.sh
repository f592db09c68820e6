#!/usr/bin/env bash
# Same-call A/B on the GPU box: tools/ab_run.sh <out-tag> "<bench.py args>" <variant> [<variant> ...]
# A variant is `name` (library ab/<name>/libunetk.so; `base` = the tree's own library) optionally followed by `:ENV=VAL,ENV=VAL`.
# Every variant runs ROUNDS times, interleaved; prints value / ms_per_step per run.  Outputs in gpurun_out/ab/<out-tag>/.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
TAG="$1"; ARGS="$2"; shift 2
OUT="$ROOT/gpurun_out/ab/$TAG"; mkdir -p "$OUT"
cd "$ROOT"
for r in $(seq 1 "${ROUNDS:-2}"); do
  for v in "$@"; do
    name="${v%%:*}"; envs=""
    if [[ "$v" == *:* ]]; then envs="${v#*:}"; fi
    lib="$ROOT/ab/$name/libunetk.so"; [[ "$name" == base ]] && lib="$ROOT/boxsegliver_amd/lib/libunetk.so"
    f="$OUT/${v//[:=,]/_}_r$r.json"
    env UNETK_LIB="$lib" $(echo "$envs" | tr ',' ' ') timeout -k 10 300 python bench.py $ARGS --no-cpu-baseline > "$f" 2> "$f.err" || { echo "$v failed"; tail -5 "$f.err"; exit 1; }
    python - "$f" "$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
hb = {r['kernel']: r['total_ms_per_step'] for r in d.get('hbm_kernels', [])}
print("%-40s value %9.2f  ms %8.3f  hbm_passes %.3f" % (sys.argv[2], d['value'], d['ms_per_step'], sum(hb.values())))
PY
  done
done

#!/usr/bin/env bash
# Time the parts of the round-3 bf16 conv loop (csrc/conv_igemm_bf16s.hip) on the GPU box: builds a PROBE copy of the library
# (-DUNETK_V3_PROBE: switches that skip parts of the loop and give wrong results) over the scratch copy of the tree, runs the
# by-layer bench once per switch against that copy.  gpurun -- 'bash tools/probe_v3.sh'; outputs in gpurun_out/probe_v3/.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out/probe_v3"
mkdir -p "$OUT"
cd "$ROOT"
# the probe library is built into directories of its own and loaded through UNETK_LIB (boxsegliver_amd/_abi.py): the tree's
# real libunetk.so and build/ objects are never touched, whatever this script dies of
PROBE_DIR="$(mktemp -d /tmp/unetk_probe.XXXXXX)"
trap 'rm -rf "$PROBE_DIR"' EXIT
UNETK_EXTRA_FLAGS=-DUNETK_V3_PROBE UNETK_OUT_DIR="$PROBE_DIR/lib" UNETK_OBJ_DIR="$PROBE_DIR/obj" \
  bash boxsegliver_amd/csrc/build.sh > "$OUT/build.log" 2>&1
for f in ${PROBE_FLAGS:-0 4 8 16 32 20 60}; do
  UNETK_LIB="$PROBE_DIR/lib/libunetk.so" UNETK_V3_FLAGS=$f timeout -k 10 300 python bench.py --dtype bf16 --size 512 --batch 8 --steps 6 --warmup 2 --no-cpu-baseline --detail \
    > "$OUT/bench_detail_f$f.json" 2> "$OUT/bench_f$f.err" || echo "flag $f failed"
  echo "flag $f done"
done

#!/usr/bin/env bash
# bf16 MFMA-loop probe (tools/mfma_mix_bf16.hip) on the GPU box: timings on random and on zero data, then the matrix-pipe busy
# share / held clock per variant from a PMC pass.  gpurun -- 'bash tools/probe_bf16.sh'; outputs in gpurun_out/probe_bf16/.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out/probe_bf16"
mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value "$ROOT/tools/mfma_mix_bf16.hip" -o /tmp/mfma_mix_bf16
timeout -k 10 120 /tmp/mfma_mix_bf16 > "$OUT/mix_random.txt"
cat "$OUT/mix_random.txt"
timeout -k 10 120 /tmp/mfma_mix_bf16 z > "$OUT/mix_zeros.txt"
cat "$OUT/mix_zeros.txt"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc" -o mix -- /tmp/mfma_mix_bf16 > "$OUT/pmc.log" 2>&1
cd "$ROOT"
python tools/pmc_mfma.py "$(find "$OUT/pmc" -name '*counter_collection.csv' | head -1)" "$OUT/mix_pmc_busy.txt"
rm -rf "$OUT/pmc"

set -e
mkdir -p gpurun_out/r5h
ROUNDS=3 bash tools/ab_run.sh ldw_plain "--dtype bf16 --size 512 --batch 8 --steps 20 --warmup 5 --no-kernel-events" base:UNETK_V3_LDW=0 base:UNETK_V3_LDW=1
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  export UNETK_V3_LDW=$v
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$ROOT/gpurun_out/r5h/pmc_ldw$v" -o mfma -- \
    python3 "$ROOT/bench.py" --dtype bf16 --size 512 --batch 8 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > "$ROOT/gpurun_out/r5h/pmc_ldw$v.log" 2>&1
  python3 "$ROOT/tools/pmc_mfma.py" "$(find "$ROOT/gpurun_out/r5h/pmc_ldw$v" -name '*counter_collection.csv' | head -1)" "$ROOT/gpurun_out/r5h/pmc_mfma_busy_ldw$v.txt" | head -14
  rm -rf "$ROOT/gpurun_out/r5h/pmc_ldw$v"
done

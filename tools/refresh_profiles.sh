#!/usr/bin/env bash
# Regenerate everything profiles/ holds, on the GPU box (run through gpurun from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh bench'     (then the same with `prof`)
# Outputs land in gpurun_out/refresh/ (merged back by gpurun); copy the summaries into profiles/rNN_*.
# Each step runs to completion before the next starts (&&): a failed GPU step stops the script.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out/refresh"
mkdir -p "$OUT"
cd "$ROOT"
T="timeout -k 10 300"
PART="${1:-all}"        # bench | prof | all  (one gpurun call is at most 1200 s: run the two parts in two calls)
if [[ "$PART" != "prof" ]]; then

$T python bench.py --steps 20 --warmup 5 > "$OUT/bench_n1.json"            # the driver's own command line
echo "bench_n1 done"
$T python bench.py --steps 20 --warmup 5 --no-kernel-events --no-cpu-baseline > "$OUT/bench_n1_noevents.json"
$T python bench.py --steps 10 --warmup 3 --detail --no-cpu-baseline > "$OUT/bench_n1_by_layer.json"
$T python bench.py --model GUNet --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_gunet_bs8.json"
$T python bench.py --model UNet3D --size 96 --batch 2 --steps 8 --warmup 2 --no-cpu-baseline > "$OUT/bench_unet3d_96_bs2.json"
$T python bench.py --model UNet3D --size 96 --batch 1 --steps 8 --warmup 2 --no-cpu-baseline > "$OUT/bench_unet3d_96_bs1.json"
# by layer: --detail runs UNet3D on ONE stream (its filter gradients otherwise overlap the input gradients and stretch them)
$T python bench.py --model UNet3D --size 96 --batch 1 --steps 8 --warmup 2 --detail --no-cpu-baseline > "$OUT/bench_unet3d_96_bs1_by_layer.json"
# the reference's own 3-D training shape (threed_script/201_unet_v1.sh:26): 10 x 256 x 256 patches, bs 4 (and bs 1 = its 4-GPU layout)
$T python bench.py --model UNet3D --depth 10 --size 256 --batch 4 --steps 8 --warmup 2 --no-cpu-baseline > "$OUT/bench_unet3d_10x256_bs4.json"
$T python bench.py --model UNet3D --depth 10 --size 256 --batch 4 --steps 8 --warmup 2 --detail --no-cpu-baseline > "$OUT/bench_unet3d_10x256_bs4_by_layer.json"
$T python bench.py --model UNet3D --depth 10 --size 256 --batch 1 --steps 8 --warmup 2 --no-cpu-baseline > "$OUT/bench_unet3d_10x256_bs1.json"
for m in UNetInter LGNet SmallUNet InterUNet; do
  $T python bench.py --model $m --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_${m}_bs8.json"
done
echo "fp32 benches done"
$T python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_bf16_256_bs32.json"
$T python bench.py --dtype bf16 --size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_bf16_512_bs8.json"
$T python bench.py --dtype bf16 --model GUNet --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_bf16_gunet_bs8.json"
$T python bench.py --dtype bf16c --size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_bf16c_512_bs8.json"
$T python bench.py --dp-rehearsal --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_n1_dp_rehearsal.json"
$T python bench.py --dp-rehearsal --dtype bf16 --size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_bf16_512_bs8_dp_rehearsal.json"
$T python bench.py --dtype bf16 --model GUNet --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > "$OUT/bench_bf16_gunet_bs8_noevents.json"
echo "bf16 benches done"
# inference: the volume evaluator's slab loop, fused conv + affine + ReLU [+ pool] epilogue vs the two-pass path
$T python bench.py --mode infer --steps 24 --warmup 8 > "$OUT/bench_infer_bs32.json"
UNETK_FUSE_EVAL=0 $T python bench.py --mode infer --steps 24 --warmup 8 > "$OUT/bench_infer_bs32_twopass.json"
$T python bench.py --mode infer --mirror --steps 24 --warmup 8 > "$OUT/bench_infer_bs32_mirror.json"
$T python bench.py --mode infer --dtype bf16 --size 512 --batch 8 --steps 24 --warmup 8 > "$OUT/bench_infer_bf16_512_bs8.json"
UNETK_FUSE_EVAL=0 $T python bench.py --mode infer --dtype bf16 --size 512 --batch 8 --steps 24 --warmup 8 > "$OUT/bench_infer_bf16_512_bs8_twopass.json"
$T python bench.py --mode infer --model GUNet --batch 8 --steps 24 --warmup 8 > "$OUT/bench_infer_gunet_bs8.json"
echo "inference benches done"
$T python tools/bench_lits_load.py --slices 2000 --out "$OUT/lits_load.json" > /dev/null
$T tools/probe_ext_events.bin > "$OUT/probe_ext_events.txt"
$T tools/mfma_tile_bf16.bin > "$OUT/mfma_tile_bf16_random.txt"
$T tools/mfma_tile_bf16.bin z > "$OUT/mfma_tile_bf16_zeros.txt"
echo "loader + probes done"
fi
if [[ "$PART" == "bench" ]]; then exit 0; fi

# rocprofv3: kernel trace + stats (own run), then the two PMC passes (own runs, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_fp32" -o fp32 -- \
  python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --single-stream > "$OUT/prof_fp32.log" 2>&1
echo "kernel trace fp32 done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bf16" -o bf16 -- \
  python3 "$ROOT/bench.py" --dtype bf16 --size 512 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --single-stream > "$OUT/prof_bf16.log" 2>&1
echo "kernel trace bf16 (512x512 bs 8) done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_u3d" -o u3d -- \
  python3 "$ROOT/bench.py" --model UNet3D --size 96 --batch 1 --steps 5 --warmup 2 --detail --no-cpu-baseline --no-kernel-events > "$OUT/prof_u3d.log" 2>&1
echo "kernel trace UNet3D done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- \
  python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > "$OUT/pmc_fetch.log" 2>&1
echo "pmc fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- \
  python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > "$OUT/pmc_write.log" 2>&1
echo "pmc write done"
cd "$ROOT"
F=$(find "$OUT/pmc_fetch" -name '*counter_collection.csv' | head -1)
W=$(find "$OUT/pmc_write" -name '*counter_collection.csv' | head -1)
python tools/pmc_summary.py "$F" "$W" "$OUT/pmc_traffic.json" | tee "$OUT/pmc_summary.txt"
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_bf16" -o fetch -- \
  python3 "$ROOT/bench.py" --dtype bf16 --size 512 --batch 8 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > "$OUT/pmc_fetch_bf16.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_bf16" -o write -- \
  python3 "$ROOT/bench.py" --dtype bf16 --size 512 --batch 8 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > "$OUT/pmc_write_bf16.log" 2>&1
echo "pmc bf16 done"
# matrix-pipe busy share and held clock per kernel (own pass: SQ + GRBM counters only), fp32 headline and bf16 512x512
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -o mfma -- \
  python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > "$OUT/pmc_mfma.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma_bf16" -o mfma -- \
  python3 "$ROOT/bench.py" --dtype bf16 --size 512 --batch 8 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > "$OUT/pmc_mfma_bf16.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma_u3d" -o mfma -- \
  python3 "$ROOT/bench.py" --model UNet3D --size 96 --batch 1 --steps 2 --warmup 1 --detail --no-cpu-baseline --no-kernel-events > "$OUT/pmc_mfma_u3d.log" 2>&1
echo "pmc mfma done"
cd "$ROOT"
# LDS bank conflicts per kernel (own pass): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, fp32 headline
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_ldsc" -o lds -- \
  python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > "$OUT/pmc_ldsc.log" 2>&1
cd "$ROOT"
python tools/pmc_lds.py "$(find "$OUT/pmc_ldsc" -name '*counter_collection.csv' | head -1)" "$OUT/pmc_lds.txt" > /dev/null
rm -rf "$OUT/pmc_ldsc"
python tools/pmc_mfma.py "$(find "$OUT/pmc_mfma" -name '*counter_collection.csv' | head -1)" "$OUT/pmc_mfma_busy.txt"
python tools/pmc_mfma.py "$(find "$OUT/pmc_mfma_bf16" -name '*counter_collection.csv' | head -1)" "$OUT/pmc_mfma_busy_bf16.txt"
python tools/pmc_mfma.py "$(find "$OUT/pmc_mfma_u3d" -name '*counter_collection.csv' | head -1)" "$OUT/pmc_mfma_busy_unet3d.txt"
F=$(find "$OUT/pmc_fetch_bf16" -name '*counter_collection.csv' | head -1)
W=$(find "$OUT/pmc_write_bf16" -name '*counter_collection.csv' | head -1)
python tools/pmc_summary.py "$F" "$W" "$OUT/pmc_traffic_bf16.json" | tee "$OUT/pmc_summary_bf16.txt"
# HBM traffic of configs[4] (UNet3D 96^3, one patch) and configs[3] (GUNet bs 8): fills roofline.traffic of those lines
cd /tmp
for cfg in "unet3d --model UNet3D --size 96 --batch 1" "gunet --model GUNet --batch 8"; do
  set -- $cfg; tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_${c}_$tag" -o pmc -- \
      python3 "$ROOT/bench.py" "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --single-stream > "$OUT/pmc_${c}_$tag.log" 2>&1
  done
  F=$(find "$OUT/pmc_FETCH_SIZE_$tag" -name '*counter_collection.csv' | head -1)
  W=$(find "$OUT/pmc_WRITE_SIZE_$tag" -name '*counter_collection.csv' | head -1)
  python "$ROOT/tools/pmc_summary.py" "$F" "$W" "$OUT/pmc_traffic_$tag.json" > "$OUT/pmc_summary_$tag.txt"
  rm -rf "$OUT/pmc_FETCH_SIZE_$tag" "$OUT/pmc_WRITE_SIZE_$tag"
  echo "pmc traffic $tag done"
done
cd "$ROOT"
cp "$(find "$OUT/prof_fp32" -name '*kernel_stats.csv' | head -1)" "$OUT/bench_kernel_stats.csv"
cp "$(find "$OUT/prof_bf16" -name '*kernel_stats.csv' | head -1)" "$OUT/bench_bf16_kernel_stats.csv"
cp "$(find "$OUT/prof_u3d" -name '*kernel_stats.csv' | head -1)" "$OUT/bench_unet3d_kernel_stats.csv"
# keep the merge-back small: drop the raw traces
rm -rf "$OUT/prof_fp32" "$OUT/prof_bf16" "$OUT/prof_u3d" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_fetch_bf16" "$OUT/pmc_write_bf16" "$OUT/pmc_mfma" "$OUT/pmc_mfma_bf16" "$OUT/pmc_mfma_u3d"
head -8 "$OUT/bench_kernel_stats.csv"

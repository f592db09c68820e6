#!/usr/bin/env bash
# Build a variant copy of libunetk.so for a same-call A/B on the GPU box:  tools/ab_build.sh <name> "<extra hipcc flags>"
# -> ab/<name>/libunetk.so (git-ignored, travels with gpurun; objects under build/ab_<name>).  Loaded through UNETK_LIB.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
NAME="$1"; FLAGS="${2:-}"
UNETK_EXTRA_FLAGS="$FLAGS" UNETK_OUT_DIR="$ROOT/ab/$NAME" UNETK_OBJ_DIR="$ROOT/build/ab_$NAME" bash "$ROOT/boxsegliver_amd/csrc/build.sh"

#!/usr/bin/env bash
# Build libunetk.so (HIP kernels + C ABI) for gfx950.  Cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="${UNETK_OUT_DIR:-$ROOT/boxsegliver_amd/lib}"      # probe builds (tools/probe_v3.sh) go to directories of their own
OBJ="${UNETK_OBJ_DIR:-$ROOT/build}"
mkdir -p "$OUT" "$OBJ"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$HERE -Wall -Wno-unused-function ${UNETK_EXTRA_FLAGS:-}"
objs=()
pids=()
for src in conv_igemm conv_igemm_lin conv_igemm_bf16 conv_igemm_bf16s conv_wgrad conv_wgrad_bf16s conv3d deconv norm reduce pool head weights lits fc conv1d pack optim prof; do
  o="$OBJ/$src.o"
  if [[ ! -f "$o" || "$HERE/$src.hip" -nt "$o" || "$HERE/common.h" -nt "$o" || "$HERE/pack.h" -nt "$o" || "$ROOT/include/unetk.h" -nt "$o" ]]; then
    rm -f "$o"                                  # a failed compile must not leave a stale object for the link
    "$HIPCC" $FLAGS -c "$HERE/$src.hip" -o "$o" &
    pids+=("$!")
  fi
  objs+=("$o")
done
for pid in "${pids[@]:-}"; do                   # a bare `wait` returns 0 even when a job failed
  [[ -z "$pid" ]] || wait "$pid" || { echo "build.sh: a compile job failed" >&2; exit 1; }
done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/libunetk.so" "${objs[@]}"
echo "built $OUT/libunetk.so"

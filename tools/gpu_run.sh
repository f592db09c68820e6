#!/usr/bin/env bash
set -o pipefail
mkdir -p gpurun_out
python bench.py --model UNet3D --size 96 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --detail > gpurun_out/r2_u3d_bs1.json 2> gpurun_out/r2_u3d.err || tail -5 gpurun_out/r2_u3d.err
cut -c1-300 gpurun_out/r2_u3d_bs1.json
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_u3d
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o u3d -- python3 $GRAFT_REPO_ROOT/bench.py --model UNet3D --size 96 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > $OUT/bench.json 2> $OUT/err.log
rm -f $OUT/*kernel_trace.csv $OUT/*.db

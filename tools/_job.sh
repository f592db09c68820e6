# scratch job file for `gpurun -- 'bash tools/_job.sh'` (overwritten per experiment; the round's experiments are recorded in profiles/)
set -e
mkdir -p gpurun_out/job
ROUNDS=2 bash tools/ab_run.sh omp_gunetbf16 "--dtype bf16 --model GUNet --batch 8 --steps 20 --warmup 5 --no-kernel-events" base:OMP_NUM_THREADS=128,MKL_NUM_THREADS=128 base
ROUNDS=2 bash tools/ab_run.sh omp_u3d "--model UNet3D --size 96 --batch 1 --steps 8 --warmup 2 --no-kernel-events" base:OMP_NUM_THREADS=128,MKL_NUM_THREADS=128 base
ROUNDS=2 bash tools/ab_run.sh omp_bf16 "--dtype bf16 --size 512 --batch 8 --steps 20 --warmup 5 --no-kernel-events" base:OMP_NUM_THREADS=128,MKL_NUM_THREADS=128 base
ROUNDS=1 bash tools/ab_run.sh omp_infer "--mode infer --model GUNet --batch 8 --steps 24 --warmup 8" base:OMP_NUM_THREADS=128,MKL_NUM_THREADS=128 base

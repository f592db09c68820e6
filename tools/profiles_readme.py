"""Write profiles/rNN_README.md from the files tools/refresh_profiles.sh produced (after they were copied to profiles/rNN_*).

Usage: python tools/profiles_readme.py r02 > profiles/r02_README.md"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r02"


def J(name):
    with open(os.path.join(ROOT, "profiles", "{}_{}.json".format(R, name))) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def table(name, top=16):
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "{}_{}.csv".format(R, name)))))
    out = ["| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
    for r in rows[:top]:
        k = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        out.append("| `{}` | {} | {:.1f} | {:.2f} | {} |".format(k, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                            float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
    return "\n".join(out)


def pct(d):
    return 100.0 * d["whole_step_frac_of_dtype_peak"]


h, hb, hc, h256, g, gb = J("bench_n1"), J("bench_bf16_512_bs8"), J("bench_bf16c_512_bs8"), J("bench_bf16_256_bs32"), \
    J("bench_gunet_bs8"), J("bench_bf16_gunet_bs8")
u1, u2 = J("bench_unet3d_96_bs1"), J("bench_unet3d_96_bs2")
others = [J("bench_{}_bs8".format(m)) for m in ("UNetInter", "LGNet", "SmallUNet", "InterUNet")]
rf, rfb = h["roofline"], hb["roofline"]
mf = open(os.path.join(ROOT, "profiles", R + "_pmc_mfma_busy.txt")).read().rstrip()
mfb = "\n".join(open(os.path.join(ROOT, "profiles", R + "_pmc_mfma_busy_bf16.txt")).read().rstrip().splitlines()[:12])

print("""# Round 2 profiles (one MI355X, gfx950, ROCm 7.2)

All produced on the GPU box by `tools/refresh_profiles.sh` (ONE `gpurun` call, i.e. one device; the script holds the exact
command lines) with the kernels of the commit that adds this file; this text is written from those files by
`tools/profiles_readme.py`.  Boxes of the pool differ by 3-5 % on the same binary (`MI355X_MICROARCH.md`, DVFS notes: devices
differ): other calls of this session read 414.4-419.4 slices/s (fp32 headline) and 534-551 slices/s (bf16 512x512).

* `{R}_bench_n1.json` -- `python bench.py --steps 10 --warmup 3` (headline, cfg1: UNet 256x256x3 bs 32 fp32, fwd+bwd+TF-Adam):
  **{hv} slices/s, {hms} ms/step = {htf} TFLOP/s = {hp:.1f} % of the fp32 peak**
  (round 1: 402.0, 79.60 ms, 73.8 %); dominant kernel `{rk}` {ra} TFLOP/s = {rfp:.1f} %,
  {rg} GFLOP and {rms} ms per launch (HIP events on the launch stream inside the timed region); `traffic` {tr} bytes = launch-weighted
  mean over the tile configuration's instantiations in `{R}_pmc_traffic.json` (the PMC file of the PREVIOUS refresh of the same kernels: the bench reads the
  committed one, the passes of this refresh came after it in the script);
  `cpu_baseline` (oracle port) at cfg1-shaped bs 2 and cfg0 (2 classes), median of 5 steps.
* `{R}_bench_kernel_stats.csv` -- `rocprofv3 --kernel-trace --stats` of `bench.py --steps 5 --warmup 2 --no-cpu-baseline`
  (1 variable-creating eval forward + 2 warm-up + 5 timed steps).  `<2,2,4,2,...>` / `<4,1,2,2,...>` = the 16 x 16 pixel tiles
  (256 x 128, 256 x 64); `<...,0>` plain epilogue, `<...,2>` input gradient + the producing unit's norm-backward reduction;
  `pack_many_kernel` = every filter re-layout of a step in one launch.
* `{R}_bench_n1_by_layer.json` -- `--detail`: one row per (kernel, layer shape).
* `{R}_pmc_traffic.json` / `{R}_pmc_traffic_bf16.json` -- `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`
  (separate passes) of `bench.py --steps 2 --warmup 1` / `bench.py --dtype bf16 --size 512 --batch 8 --steps 2 --warmup 1`,
  summarised by `tools/pmc_summary.py` (read side x2: the gfx950 FETCH_SIZE correction).
* `{R}_pmc_mfma_busy.txt` / `{R}_pmc_mfma_busy_bf16.txt` -- `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` (own pass) of the
  same two commands, summarised by `tools/pmc_mfma.py`: the clock each kernel held and the share of cycles its matrix pipes
  were busy (DESIGN.md 5.0 "Where the last 10 % go"); `{R}_pmc_mfma_busy_unet3d.txt`: the same pass for UNet3D 96^3 at one
  patch; `{R}_mfma_mix.txt`: output of the stand-alone MFMA-loop probe `tools/mfma_mix.hip`.
* **bf16 storage mode** (`--dtype bf16`): `{R}_bench_bf16_512_bs8.json` (configs[2] per-GPU shape):
  **{bv} slices/s, {bms} ms/step = {btf} TFLOP/s = {bp:.1f} % of the dense bf16 peak** on this box (537.6-548.7 slices/s = 24.7-25.2 %
  on other boxes of the session; round 1, fp32 storage: 324.1 slices/s, 24.7 ms, 14.9 % -- kept as `--dtype bf16c`:
  `{R}_bench_bf16c_512_bs8.json`, {cv} slices/s); dominant kernel `{bk}` {ba} TFLOP/s = {bfp:.1f} %;
  `{R}_bench_bf16_256_bs32.json` headline shape in bf16: {h256v} slices/s ({h256ms} ms/step);
  `{R}_bench_bf16_gunet_bs8.json` {gbv} slices/s; `{R}_bench_bf16_kernel_stats.csv` -- rocprofv3 stats of the 512x512 bs 8 run.
* `{R}_bench_gunet_bs8.json` -- GUNet + guide, instance norm, bs 8 (configs[3] per-GPU shape): {gv} slices/s ({gp:.1f} %;
  round 1: 355.2, 65.2 %); `{R}_bench_{{UNetInter,LGNet,SmallUNet,InterUNet}}_bs8.json`: {ov} slices/s.
* `{R}_bench_unet3d_96_bs1.json` / `_bs2.json` (+ `{R}_bench_unet3d_kernel_stats.csv` for bs 1): UNet3D 96^3:
  **{u1v} patches/s at one patch per GPU ({u1ms} ms, {u1p:.1f} %; round 1: 41.7, 24.0 ms, 42.7 %)**,
  {u2v} at bs 2 ({u2p:.1f} %; round 1: 47.8, 49.1 %).
* `{R}_bench_2rank_gloo_rehearsal.json` -- `UNETK_DIST_BACKEND=gloo python bench.py --gpus 2 --batch 8` on the ONE GPU of the
  test box: the self-launching multi-rank path (parent spawns the ranks before any HIP call); not a scaling number.

## Matrix-pipe busy share and held clock (fp32 headline)

```
{mf}
```

bf16 512x512 bs 8:

```
{mfb}
```

## fp32 headline run (rocprofv3 --stats)

{t1}

## bf16 storage, 512x512 bs 8

{t2}

## UNet3D 96^3, one patch

{t3}""".format(R=R, hv=h["value"], hms=h["ms_per_step"], htf=h["whole_step_tflops"], hp=pct(h), rk=rf["kernel"], ra=rf["achieved"],
               rfp=100 * rf["frac"], rg=rf["avg_launch_gflop"], tr=rf.get("traffic"), rms=rf["avg_launch_ms"], bv=hb["value"], bms=hb["ms_per_step"],
               btf=hb["whole_step_tflops"], bp=pct(hb), cv=hc["value"], bk=rfb["kernel"], ba=rfb["achieved"], bfp=100 * rfb["frac"],
               h256v=h256["value"], h256ms=h256["ms_per_step"], gbv=gb["value"], gv=g["value"], gp=pct(g),
               ov=" / ".join(str(o["value"]) for o in others), u1v=u1["value"], u1ms=u1["ms_per_step"], u1p=pct(u1),
               u2v=u2["value"], u2p=pct(u2), mf=mf, mfb=mfb, t1=table("bench_kernel_stats"), t2=table("bench_bf16_kernel_stats"),
               t3=table("bench_unet3d_kernel_stats")))

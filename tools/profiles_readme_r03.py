"""Write profiles/r03_README.md from the round-3 files under profiles/ (copied there from gpurun_out/ after
tools/refresh_profiles.sh, tools/probe_bf16.sh and tools/probe_v3.sh ran on the GPU box).

Usage: python tools/profiles_readme_r03.py > profiles/r03_README.md"""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = "r03"


def J(name):
    with open(os.path.join(ROOT, "profiles", "{}_{}.json".format(R, name))) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def table(name, top=14):
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "{}_{}.csv".format(R, name)))))
    out = ["| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
    for r in rows[:top]:
        k = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        out.append("| `{}` | {} | {:.1f} | {:.2f} | {} |".format(k, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                            float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
    return "\n".join(out)


def kern_table(d, top=12):
    out = ["| kernel (bench tag) | launches | avg ms | TFLOP/s | ms / step |", "|---|---|---|---|---|"]
    for k in d["kernels"][:top]:
        out.append("| `{}` | {} | {} | {} | {} |".format(k["kernel"], k["launches"], k["avg_launch_ms"], k["achieved_tflops"],
                                                         k["total_ms_per_step"]))
    return "\n".join(out)


def hbm_table(d):
    out = ["| HBM-bound pass | launches | avg ms | MB / launch | GB/s | of 8 TB/s | ms / step |", "|---|---|---|---|---|---|---|"]
    for k in d["hbm_kernels"]:
        out.append("| `{}` | {} | {} | {} | {} | {:.1f} % | {} |".format(k["kernel"], k["launches"], k["avg_launch_ms"], k["avg_launch_mbytes"],
                                                                       k["achieved_gbps"], 100 * k["frac_of_hbm_peak"], k["total_ms_per_step"]))
    return "\n".join(out)


def txt(name):
    return open(os.path.join(ROOT, "profiles", "{}_{}".format(R, name))).read().rstrip()


h, hb, hc, h256, g, gb, gbn = J("bench_n1"), J("bench_bf16_512_bs8"), J("bench_bf16c_512_bs8"), J("bench_bf16_256_bs32"), \
    J("bench_gunet_bs8"), J("bench_bf16_gunet_bs8"), J("bench_bf16_gunet_bs8_noevents")
u1, u2 = J("bench_unet3d_96_bs1"), J("bench_unet3d_96_bs2")
others = [J("bench_{}_bs8".format(m)) for m in ("UNetInter", "LGNet", "SmallUNet", "InterUNet")]
rf = h["roofline"]
cb = h["cpu_baseline"]
conv = json.load(open(os.path.join(ROOT, "profiles", R + "_probe_conv_loop_parts.json")))
wg = json.load(open(os.path.join(ROOT, "profiles", R + "_probe_wgrad_loop_parts.json")))


def parts_table(p, pick):
    cols = list(p["columns"])
    out = ["| layer | " + " | ".join(p["columns"][c] for c in cols) + " |", "|---|" + "---|" * len(cols)]
    for name, v in p["layers"].items():
        if any(s in name for s in pick):
            out.append("| `{}` | ".format(name.split(" [")[1].rstrip("]")) + " | ".join("{:.4f}".format(v[c]) for c in cols) + " |")
    return "\n".join(out)


print("""# Round 3 profiles (one MI355X, gfx950, ROCm 7.2)

Written by `tools/profiles_readme_r03.py` from the files next to it.  `{R}_bench_*`, `{R}_pmc_*`, `{R}_*kernel_stats.csv` come from ONE
`gpurun` call of `tools/refresh_profiles.sh` (one device; the script holds the command lines); `{R}_mfma_mix_bf16_*` from
`tools/probe_bf16.sh`; `{R}_probe_*_loop_parts.json` from `tools/probe_v3.sh` (a `-DUNETK_V3_PROBE` build, timing only);
`{R}_pmc_sq_bf16.json` from two `rocprofv3 --pmc` passes of eight SQ counters each.  Boxes of the pool differ by 3-6 % on the
same binary (matrix-heavy kernels most): other calls of this round read 13.0-14.1 ms for the bf16 step below and
75.2-77.4 ms for the fp32 headline.

## Headline (BASELINE.json configs[1]) -- matrix kernels unchanged; pool passes folded into the norm passes, first layer on the matrix pipe

`{R}_bench_n1.json` -- `python bench.py --steps 10 --warmup 3`: **{hv} slices/s, {hms} ms/step = {htf} TFLOP/s = {hp:.1f} % of the
fp32 peak** (round 2: 416.70 / 76.795 ms; round 1: 402.0).  Dominant kernel `{rk}`: {ra} TFLOP/s = {rfp:.1f} % of 157.3,
{rg} GFLOP and {rms} ms per launch (HIP events on the launch stream inside the timed region), HBM traffic {tr} bytes per launch
(`{ts}`).  `cpu_baseline` (oracle port, {cores} torch threads): {cv} slices/s at bs 2, **{c8} slices/s at bs 8** (new),
{c0} at configs[0] (2 classes).  New in the line: `hbm_kernels` (HBM-bound passes by algorithmic bytes, timed in three extra
steps OUTSIDE the timed region) and, for N > 1 or `--dp-rehearsal`, `data_parallel`.

{hbmf}

`{R}_bench_n1_dp_rehearsal.json` / `{R}_bench_bf16_512_bs8_dp_rehearsal.json` -- the same commands with `--dp-rehearsal`: the
data-parallel path (gradient buckets, RCCL all-reduce from inside backward, stream joins) in a world of ONE rank, what the
driver's N > 1 runs add on top of the step: {dpf} (fp32) and {dpb} (bf16).

## bf16 storage mode (BASELINE.json configs[2] shape: 512x512 bs 8 per GPU)

`{R}_bench_bf16_512_bs8.json`: **{bv} slices/s, {bms} ms/step = {btf} TFLOP/s = {bp:.1f} % of the dense bf16 peak** (round 2: 551.16 /
14.515 ms = 25.3 %; round 1: 324.1).  `{R}_bench_bf16_256_bs32.json` (headline shape in bf16): {b256} slices/s.
`{R}_bench_bf16c_512_bs8.json` (bf16 operands, fp32 storage): {bc} slices/s.  GUNet bs 8 in bf16: {gbn} slices/s
(`{R}_bench_bf16_gunet_bs8_noevents.json`, `--no-kernel-events`; with the per-launch events this 5 ms step is host-bound:
{gbe} in `{R}_bench_bf16_gunet_bs8.json`).

{bk}

{hbmb}

Matrix-pipe busy share and held clock (`{R}_pmc_mfma_busy_bf16.txt`: `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`):

```
{mfb}
```

### Where the bf16 kernels' time goes

`{R}_mfma_mix_bf16_random.txt` (the K loop of the conv kernel as a stand-alone program, features switched on one at a time;
random bf16 operands; `_zeros.txt` = the same on zeros, `_pmc_busy.txt` = busy share and clock per variant):

```
{mix}
```

`{R}_mfma_mix_bf16_staging.txt` (`/tmp/mfma_mix_bf16 d`, built with `-DRDS=16 | 12 | 8`: what 16 KiB staged per CU and step cost the
loop by each path, at three LDS read volumes; 12 reads per wave and step is the shipped kernel's):

```
{stg}
```

`{R}_probe_wgrad_spread.json`: the filter gradient's five requests per wave and tile in a row (shipped) against one behind each
MFMA group: {spread}.

`{R}_probe_conv_loop_parts.json` -- the REAL round-3 forward / input-gradient kernel with parts of its loop switched off (ms per
launch; columns 0 / 64 / 128 / 32 from a later call than 4 / 8 / 16 / 20 / 60):

{pc}

`{R}_probe_wgrad_loop_parts.json` -- the same for the filter gradient (kernel + slab reduction):

{pw}

## fp32, other shapes (unchanged kernels)

| bench | slices (patches) / s | ms / step | of fp32 peak |
|---|---|---|---|
| GUNet + guide, IN, bs 8 (configs[3] per-GPU shape) | {gv} | {gms} | {gp:.1f} % |
| UNet3D 96^3, bs 1 (configs[4] per-GPU shape) | {u1v} | {u1ms} | {u1p:.1f} % |
| UNet3D 96^3, bs 2 | {u2v} | {u2ms} | {u2p:.1f} % |
| UNetInter / LGNet / SmallUNet / InterUNet, bs 8 | {ov} | | |

## fp32 headline run (rocprofv3 --stats, `{R}_bench_kernel_stats.csv`)

{t32}

## bf16 storage, 512x512 bs 8 (`{R}_bench_bf16_kernel_stats.csv`)

{t16}""".format(
    R=R, hv=h["value"], hms=h["ms_per_step"], htf=h["whole_step_tflops"], hp=100 * h["whole_step_frac_of_dtype_peak"],
    rk=rf["kernel"], ra=rf["achieved"], rfp=100 * rf["frac"], rg=rf["avg_launch_gflop"], rms=rf["avg_launch_ms"], tr=rf["traffic"],
    ts=rf.get("traffic_source", "-"), cores=cb["cores"], cv=cb["value"], c8=cb["bs8"]["value"], c0=cb["cfg0"]["value"],
    hbmf=hbm_table(h), bv=hb["value"], bms=hb["ms_per_step"], btf=hb["whole_step_tflops"], bp=100 * hb["whole_step_frac_of_dtype_peak"],
    b256=h256["value"], bc=hc["value"], gbn=gbn["value"], gbe=gb["value"], bk=kern_table(hb), hbmb=hbm_table(hb),
    dpf=json.dumps(J("bench_n1_dp_rehearsal")["data_parallel"]), dpb=json.dumps(J("bench_bf16_512_bs8_dp_rehearsal")["data_parallel"]),
    stg=txt("mfma_mix_bf16_staging.txt"),
    spread=", ".join("{} {:.4f} / {:.4f} ms".format(k.split(" [")[1].rstrip("]"), v["0"], v["2048"]) for k, v in list(json.load(open(
        os.path.join(ROOT, "profiles", R + "_probe_wgrad_spread.json")))["layers"].items())[:4]),
    mfb="\n".join(txt("pmc_mfma_busy_bf16.txt").splitlines()[:13]), mix=txt("mfma_mix_bf16_random.txt"),
    pc=parts_table(conv, ("64x64 1024->512", "512x512 64->64]", "256x256 128->128]", "512x512 64->128")),
    pw=parts_table(wg, ("256x256 128->128", "512x512 64->64", "64x64 1024->512")),
    gv=g["value"], gms=g["ms_per_step"], gp=100 * g["whole_step_frac_of_dtype_peak"],
    u1v=u1["value"], u1ms=u1["ms_per_step"], u1p=100 * u1["whole_step_frac_of_dtype_peak"],
    u2v=u2["value"], u2ms=u2["ms_per_step"], u2p=100 * u2["whole_step_frac_of_dtype_peak"],
    ov=" / ".join(str(o["value"]) for o in others), t32=table("bench_kernel_stats"), t16=table("bench_bf16_kernel_stats")))

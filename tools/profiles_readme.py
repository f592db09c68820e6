"""Write profiles/rNN_README.md from the round's files under profiles/ (copied there from gpurun_out/refresh/ after the two
`tools/refresh_profiles.sh bench | prof` calls ran on the GPU box).  One generator for every round from 4 on; the READMEs of
rounds 1-3 are frozen documents.

Usage: python tools/profiles_readme.py r04 > profiles/r04_README.md"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r04"


def J(name):
    with open(os.path.join(ROOT, "profiles", "{}_{}.json".format(R, name))) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def stats_rows(name):
    return {r["Name"]: r for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "{}_{}.csv".format(R, name))))}


def table(name, top=14):
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "{}_{}.csv".format(R, name)))))
    out = ["| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
    for r in rows[:top]:
        k = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        out.append("| `{}` | {} | {:.1f} | {:.2f} | {} |".format(k, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                            float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
    return "\n".join(out)


def kern_table(d, top=12):
    out = ["| op (bench tag) | launches | avg ms | TFLOP/s | ms / step |", "|---|---|---|---|---|"]
    for k in d["kernels"][:top]:
        out.append("| `{}` | {} | {} | {} | {} |".format(k["kernel"], k["launches"], k["avg_launch_ms"], k["achieved_tflops"],
                                                         k["total_ms_per_step"]))
    return "\n".join(out)


def hbm_table(d):
    out = ["| HBM-bound pass | launches | avg ms | MB / launch | GB/s | of 8 TB/s | ms / step |", "|---|---|---|---|---|---|---|"]
    for k in d["hbm_kernels"]:
        out.append("| `{}` | {} | {} | {} | {} | {:.1f} % | {} |".format(k["kernel"], k["launches"], k["avg_launch_ms"], k.get("avg_launch_mbytes", "-"),
                                                                       k["achieved_gbps"], 100 * k.get("frac_of_hbm_peak", k["achieved_gbps"] / 8000.0), k["total_ms_per_step"]))
    return "\n".join(out)


def agree_table(d, stats, top=8):
    """bench.py's per-kernel averages (start / stop events bound to the dispatch) next to rocprofv3 --kernel-trace --stats of the
    same command line in its own profiled run."""
    out = ["| kernel (as rocprofv3 names it) | bench calls | bench avg ms | rocprofv3 calls | rocprofv3 avg ms | diff |", "|---|---|---|---|---|---|"]
    for r in d["kernel_trace"][:top]:
        q = stats.get(r["name"])
        if q is None:
            continue
        a = float(q["AverageNs"]) / 1e6
        out.append("| `{}` | {} | {} | {} | {:.4f} | {:+.1f} % |".format(
            r["name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], r["calls"], r["avg_ms"], q["Calls"], a,
            100.0 * (r["avg_ms"] - a) / a))
    return "\n".join(out)


def txt(name):
    return open(os.path.join(ROOT, "profiles", "{}_{}".format(R, name))).read().rstrip()


def pct(d):
    return 100 * d["whole_step_frac_of_dtype_peak"]


h, hn, hb, hc, h256 = J("bench_n1"), J("bench_n1_noevents"), J("bench_bf16_512_bs8"), J("bench_bf16c_512_bs8"), J("bench_bf16_256_bs32")
g, gb, gbn = J("bench_gunet_bs8"), J("bench_bf16_gunet_bs8"), J("bench_bf16_gunet_bs8_noevents")
u1, u2, r4, r1 = J("bench_unet3d_96_bs1"), J("bench_unet3d_96_bs2"), J("bench_unet3d_10x256_bs4"), J("bench_unet3d_10x256_bs1")
u1d, r4d = J("bench_unet3d_96_bs1_by_layer"), J("bench_unet3d_10x256_bs4_by_layer")
inf, inf2, infm, infb, infb2, infg = J("bench_infer_bs32"), J("bench_infer_bs32_twopass"), J("bench_infer_bs32_mirror"), \
    J("bench_infer_bf16_512_bs8"), J("bench_infer_bf16_512_bs8_twopass"), J("bench_infer_gunet_bs8")
others = [J("bench_{}_bs8".format(m)) for m in ("UNetInter", "LGNet", "SmallUNet", "InterUNet")]
ll = json.load(open(os.path.join(ROOT, "profiles", R + "_lits_load.json")))
rf, cb = h["roofline"], h["cpu_baseline"]
st32 = stats_rows("bench_kernel_stats")

R5 = """
## Round 5: what changed in how the line is produced, and this round's same-call A/Bs

* **Filter gradients on a second stream** are now the default for the fp32 2-D units too (`ops._Side`, queued BEFORE the unit's input
  gradient so that the two kernels really share the chip): headline 75.40 / 75.24 -> 74.16 ms in a same-call A/B; UNet3D 96^3 at one
  patch 20.51 -> 19.97 ms from the same re-ordering; GUNet bs 8 no change; bf16 storage loses (13.28-13.33 -> 13.42 ms: kept off).
  `bench.py` runs its TRACED steps (at most three) on one stream, so the kernel tables, `roofline` and `kernels_fit_step` describe
  single-stream steps while `value` is over all timed steps (`side_stream_filter_gradients`, `traced_steps_single_stream`); the rocprofv3
  runs of this set use `--single-stream` for the same reason (their per-kernel averages are compared with the line's below).
* `r05_unet3d_s2lin_ab.txt` -- UNet3D's stride-2 layers with small output planes: space-to-depth copy + the linear-pixel kernel with
  grouped taps (stream-K), and the (2,2,2) input gradient as one launch: 20.31 / 20.39 -> 19.91 / 19.95 ms.
* `r05_dp_rehearsal_probe.txt` -- where the data-parallel path lost 2-2.7 % in a world of one: not in the ~100 Python hooks
  (efficiency 0.9995 with the collectives skipped) but ~30-40 us of stream hand-over per COLLECTIVE; 8 -> 3 collectives per step.
* `r05_probe_ilv_epilogue.txt` (+ `.patch`) -- the 64-channel bf16 kernel's tile epilogue drained inside the next tile's K loop:
  parity-green, 13-18 % SLOWER per launch.  Not kept.
* `r05_probe_loader_waves.txt` (+ `.patch`) -- four loader waves for the same kernel: the kernel is 9-17 % FASTER, the step 2 % SLOWER
  (every other matrix kernel of the step runs 4 % slower behind it: clock).  Not kept.
* `r05_probe_norm_mirror.txt` -- norm-pass traversal order probe (no gain).
* `r05_unet3d_kskip_ab.txt` -- UNet3D's channel padding (30 / 60 / 120 / 240 in 32 / 64 / 128 / 256) no longer contracted: one bit per
  8-channel group of Cin / Cout rides on every padded filter (`unetk_conv3d_desc.cin_live8 / cout_live8`, ABI 10); the linear-pixel
  kernel leaves dead 16-channel chunks out of K and runs half the MFMAs of half-live ones: 20.51 -> 20.15 ms (- 1.7 %) in a same-call A/B.
* `r05_gunet_lin2d_ab.txt` -- the 2-D nets' starved 16 x 16 planes at 8 slices per GPU: 128-pixel blocks + stream-K over all tiles
  (`UNETK_LIN_2D`, default 6 now): bridge layers 93 -> 107 TFLOP/s, GUNet bs 8 370.3 -> 373.3 slices/s.
* `r05_probe_wgrad_rounds.txt` -- one round of larger blocks instead of two for the stacked-plane filter gradient (a dispatch model says
  - 5 %): no gain beside the main stream's kernels.  Not kept.
* `r05_probe_lin_prefetch.txt` (+ `.patch`) -- **the register-prefetch loops waited for their own requests**: rocprofv3's FETCH_SIZE on
  UNet3D (`r05_pmc_traffic_unet3d.json`: 820 MB per launch beyond L2 on layers with 35 MB of operands) -> a same-panel timing probe (the
  kernel waits for those loads) -> the ISA (`s_waitcnt vmcnt(0)` at the end of every exec-masked request block, in front of the tap's
  MFMAs) -> unconditional requests (zero page by pointer select, NOT a const array: that made them flat loads) pinned by
  `__builtin_amdgcn_sched_barrier`: headline 430.5 -> 434.7 slices/s, UNet3D 96^3 49.9 -> 51.85 patches/s, GUNet bs 8 373.6 -> 380.2,
  inference bs 32 1385 -> 1407 (same-call A/Bs against the library before); `tools/asm_lint.py` counts such waits and flat loads.
* `r05_asm_lint.txt` -- `tools/asm_lint.py` over the final sources: per MFMA kernel, flat loads and prefetch waits in front of the MFMAs
  (zero for the tiled, plain linear-pixel and wgrad kernels; the remaining ones are listed in DESIGN.md 4).
* `r05_pmc_lds*.txt` (`tools/pmc_lds.py`) -- LDS bank-conflict share per kernel: the tiled conv 34.7 % -> 6.1 % after padding its halo rows
  to 384 floats (`r05_pmc_lds_before_rowpad.txt`); no change of the step.
* `r05_probe_wgrad_valu.txt` -- instructions per MFMA (SQ_INSTS_*): the stacked-plane filter gradient ran 5.1 VALU per MFMA (four run-time
  integer divisions per staged piece) -> divisions by reciprocal: 98.8 -> 102.4 / 106.6 -> 111.5 TFLOP/s.
* `r05_step_hiccups.txt`, `r05_bench_n1_noevents_hiccup.json` -- one-step GPU gaps (0.4-1 s) in the 2nd-4th process on a fresh box, and
  `bench.py`'s settle phase.
* `r05_pmc_traffic_unet3d.json`, `r05_pmc_traffic_gunet.json` -- FETCH_SIZE / WRITE_SIZE passes of configs[4] and configs[3]: their
  bench lines carry `roofline.traffic` too.
* The CPU thread pools follow the cgroup's quota (`boxsegliver_amd/utils/hostcpu.py`: the box shows 256 CPUs and grants 16):
  `cpu_baseline` 0.38 -> 2.0 slices/s on 16 threads, the GPU test suite 525-600 s -> 150 s.
* The bridge's filter gradient (6 x 6 output planes) on 4 x 6 stride-2 tiles: 0.309 -> 0.151 ms (`r05_bench_unet3d_96_bs1_by_layer.json`
  `conv3d_wgrad [k3s22 ...]`; same-call step 20.19 -> 20.11 ms).
"""

print("""# Round {rn} profiles (one MI355X, gfx950, ROCm 7.2)
{r5}

Written by `tools/profiles_readme.py {R}` from the files next to it.  Everything here comes from two `gpurun` calls on the final tree of
the round: `tools/refresh_profiles.sh bench` (`{R}_bench_*.json`, `{R}_lits_load.json`, the probe outputs) and
`tools/refresh_profiles.sh prof` (`{R}_*kernel_stats.csv`, `{R}_pmc_*`) -- the script holds the command lines.  Boxes of the pool differ
by 3-6 % on matrix-heavy kernels.

## How a kernel is timed (since round 4)

`bench.py` no longer brackets a C-ABI call with `hipEventRecord` pairs (such a bracket swallows the host's time whenever the stream has
drained: `{R}_probe_ext_events.txt`, `tools/probe_ext_events.hip` -- a 1.000 ms kernel reads 4.1-4.4 ms behind a 3 ms host stall).  The
library itself traces its launches (`csrc/prof.hip`, `unetk_prof_*`): on the traced steps ({tr_note}) each launch goes through
`hipExtLaunchKernelGGL` with a start and a stop event BOUND TO THE DISPATCH, so a duration is the GPU's own begin -> end interval of that
kernel -- the quantity `rocprofv3 --kernel-trace` reports -- whatever the host does around the launch:

```
{ext}
```

The line carries the cross-checks the review asked for: `step_ms_event_steps` {ev} / `step_ms_plain_steps` {pl} ms (a traced launch costs
~7 us of host time and a few us on the GPU: traced steps are {evd:.2f} ms slower, which is why only a few steps are traced),
`host_ms_per_step` {host} (the host enqueues a {pl}-ms step in {host} ms: the GPU never waits for it), `gpu_kernel_ms_per_step` {gk}
(every kernel of the library, own GPU time, per traced step), `sum_kernels_plus_hbm_kernels_ms` {sk} and **`kernels_fit_step`: {fit}**.
The same command without tracing (`{R}_bench_n1_noevents.json`): {hnv} slices/s.

bench.py's per-kernel averages against `rocprofv3 --kernel-trace --stats` of the same command line (`{R}_bench_kernel_stats.csv`, its
own profiled run, possibly another box of the pool):

{agree}

## Headline (BASELINE.json configs[1])

`{R}_bench_n1.json` -- `python bench.py --steps 20 --warmup 5` (the driver's command): **{hv} slices/s, {hms} ms/step = {htf} TFLOP/s =
{hp:.1f} % of the fp32 peak** (round 3: 424.86 / 75.32 ms on its box; round 2: 416.70; round 1: 402.0).  Dominant kernel
`{rk}`: {ra} TFLOP/s = **{rfp:.1f} %** of 157.3, {rg} GFLOP and {rms} ms per launch over {rl} launches, HBM traffic {tr} bytes per
launch (`{ts}`).  `cpu_baseline` (oracle port, {cores} torch threads, {cw} s of CPU work in all -- round 3: 77 s): {cv} slices/s at bs 2,
{c0} at configs[0] (2 classes).

{hk}

{hbmf}

`{R}_bench_n1_dp_rehearsal.json` / `{R}_bench_bf16_512_bs8_dp_rehearsal.json` -- `--dp-rehearsal`: the data-parallel path (gradient
buckets in arrival order across both flat buffers, RCCL all-reduce from inside backward) in a world of ONE rank: {dpf} (fp32) and
{dpb} (bf16).  `bucket_launch_progress` = the share of the variables that had arrived when each bucket went out.

## Inference (new: `bench.py --mode infer`; evaluators/evaluator_liver.py's slab loop, forward only)

Eval-mode forward with conv + (scale, shift) + ReLU [+ 2 x 2 max-pool] as ONE kernel per unit (`unetk_conv3x3_fwd_affine`) against the
two-pass path (`UNETK_FUSE_EVAL=0`: conv, then `norm_apply_relu[_pool]`):

| bench | fused | two passes | forward TFLOP/s (fused) | of the dtype's peak |
|---|---|---|---|---|
| UNet 256 x 256, bs 32, fp32 | **{iv} slices/s** ({ims} ms / slab) | {i2v} ({i2ms} ms) | {itf} | {ip:.1f} % |
| same with mirror TTA (4 forwards per slab) | {imv} slices/s | | {imtf} | {imp:.1f} % |
| UNet 512 x 512, bs 8, bf16 storage | **{ibv} slices/s** ({ibms} ms) | {ib2v} ({ib2ms} ms) | {ibtf} | {ibp:.1f} % |
| GUNet + guide (instance norm: statistics of the slab itself, nothing to fuse), bs 8 | {igv} slices/s | | {igtf} | {igp:.1f} % |

{ik}

## UNet3D

| bench | patches / s | ms / step | of fp32 peak |
|---|---|---|---|
| 96^3, bs 1 (configs[4] per-GPU shape) | {u1v} | {u1ms} | {u1p:.1f} % |
| 96^3, bs 2 | {u2v} | {u2ms} | {u2p:.1f} % |
| **10 x 256 x 256, bs 4** (the reference's own 3-D training shape, `threed_script/201_unet_v1.sh:26`) | **{r4v}** | {r4ms} | {r4p:.1f} % |
| 10 x 256 x 256, bs 1 (its 4-GPU mirrored layout) | {r1v} | {r1ms} | {r1p:.1f} % |

UNet3D runs its conv3d filter gradients on a second stream beside the input gradients (bit-identical results; 96^3 at one patch:
{u1dms} ms on one stream -> {u1ms} ms); kernels then overlap and stretch each other, so the by-layer tables come from `--detail` runs,
which use ONE stream (`{R}_bench_unet3d_96_bs1_by_layer.json`: {u1dv} patches/s, `{R}_bench_unet3d_10x256_bs4_by_layer.json`: {r4dv}).

`{R}_bench_unet3d_10x256_bs4_by_layer.json`, the ops below 100 TFLOP/s:

{r4k}

`{R}_bench_unet3d_96_bs1_by_layer.json`, the ops below 80 TFLOP/s:

{u1k}

Matrix-pipe busy share and held clock at one 96^3 patch (`{R}_pmc_mfma_busy_unet3d.txt`):

```
{mfu}
```

## bf16 storage mode (BASELINE.json configs[2] shape: 512 x 512 bs 8 per GPU)

`{R}_bench_bf16_512_bs8.json`: **{bv} slices/s, {bms} ms/step = {btf} TFLOP/s = {bp:.1f} % of the dense bf16 peak** (matrix kernels unchanged
since round 3: 613.26 on its box; round 5's two schedule probes of the 64-channel kernel are below).  `{R}_bench_bf16_256_bs32.json`: {b256} slices/s; `{R}_bench_bf16c_512_bs8.json` (bf16 operands, fp32
storage): {bc}; GUNet bs 8 in bf16: {gbn} slices/s untraced, {gbe} traced.

{bk}

`{R}_mfma_tile_bf16_random.txt` / `_zeros.txt` (`tools/mfma_tile_bf16.hip`) -- the structural alternative the round-3 review asked to be
built or refuted: a 128 x 128 accumulator tile per wave in AGPRs at ONE wave per SIMD (16 LDS reads per 64 MFMAs) against the shipped
64 x 128 tile at two waves per SIMD (12 per 32), same block tile, LDS image, staging volume and barrier:

```
{tile}
```

With everything a real loop needs (staging + barrier) the big tile is 3 % SLOWER on random data (6 % on zeros): LDS operand bandwidth is
not what holds the loop, and the second wave per SIMD is what hides each wave's request stalls.  Not built.

Matrix-pipe busy share (`{R}_pmc_mfma_busy_bf16.txt`):

```
{mfb}
```

## fp32, other shapes

| bench | slices / s | ms / step | of fp32 peak |
|---|---|---|---|
| GUNet + guide, IN, bs 8 (configs[3] per-GPU shape) | {gv} | {gms} | {gp:.1f} % |
| UNetInter / LGNet / SmallUNet / InterUNet, bs 8 | {ov} | | |

## LiTS loader (`{R}_lits_load.json`, `tools/bench_lits_load.py`)

{lls} slice pairs (512 x 512 16-bit image + 8-bit label PNG, all five row-filter types mixed per row) from /dev/shm into the resident
store: **{llr} pairs/s** ({llt} s, {llth} inflate threads, chunk {llc}), peak host RSS {rss0} -> {rss1} GB (staging buffers {stg} GB);
the un-filter kernel alone: {kim:.0f} images/s ({kms} ms per 256 16-bit images).  Round 3's loader ran Average / Paeth rows through a
per-byte Python loop (~0.3-0.5 s per slice).

## fp32 headline run (rocprofv3 --stats, `{R}_bench_kernel_stats.csv`)

{t32}

## bf16 storage, 512 x 512 bs 8 (`{R}_bench_bf16_kernel_stats.csv`)

{t16}

## UNet3D 96^3 bs 1 (`{R}_bench_unet3d_kernel_stats.csv`)

{t3d}""".format(
    rn=int(R[1:]), r5=R5 if R >= "r05" else "", tr_note=("at most three of the timed steps, run on ONE stream" if R >= "r05" else "every fourth timed step"),
    R=R, ext=txt("probe_ext_events.txt"), ev=h["step_ms_event_steps"], pl=h["step_ms_plain_steps"],
    evd=h["step_ms_event_steps"] - h["step_ms_plain_steps"], host=h["host_ms_per_step"], gk=h["gpu_kernel_ms_per_step"],
    sk=h["sum_kernels_plus_hbm_kernels_ms"], fit=h["kernels_fit_step"], hnv=hn["value"], agree=agree_table(h, st32),
    hv=h["value"], hms=h["ms_per_step"], htf=h["whole_step_tflops"], hp=pct(h), rk=rf["kernel"], ra=rf["achieved"], rfp=100 * rf["frac"],
    rg=rf["avg_launch_gflop"], rms=rf["avg_launch_ms"], rl=rf["launches"], tr=rf["traffic"], ts=rf.get("traffic_source", "-"),
    cores=cb["cores"], cw=cb["wall_s"], cv=cb["value"], c0=cb["cfg0"]["value"], hk=kern_table(h), hbmf=hbm_table(h),
    dpf=json.dumps(J("bench_n1_dp_rehearsal")["data_parallel"]), dpb=json.dumps(J("bench_bf16_512_bs8_dp_rehearsal")["data_parallel"]),
    iv=inf["value"], ims=inf["ms_per_step"], i2v=inf2["value"], i2ms=inf2["ms_per_step"], itf=inf["forward_tflops"],
    ip=100 * inf["forward_frac_of_dtype_peak"], imv=infm["value"], imtf=infm["forward_tflops"], imp=100 * infm["forward_frac_of_dtype_peak"],
    ibv=infb["value"], ibms=infb["ms_per_step"], ib2v=infb2["value"], ib2ms=infb2["ms_per_step"], ibtf=infb["forward_tflops"],
    ibp=100 * infb["forward_frac_of_dtype_peak"], igv=infg["value"], igtf=infg["forward_tflops"], igp=100 * infg["forward_frac_of_dtype_peak"],
    ik=kern_table(inf, 8),
    u1v=u1["value"], u1ms=u1["ms_per_step"], u1p=pct(u1), u2v=u2["value"], u2ms=u2["ms_per_step"], u2p=pct(u2),
    r4v=r4["value"], r4ms=r4["ms_per_step"], r4p=pct(r4), r1v=r1["value"], r1ms=r1["ms_per_step"], r1p=pct(r1),
    r4k=kern_table({"kernels": [k for k in r4d["kernels"] if k["achieved_tflops"] < 100.0]}, 14),
    u1k=kern_table({"kernels": [k for k in u1d["kernels"] if k["achieved_tflops"] < 80.0]}, 16),
    u1dms=u1d["ms_per_step"], u1dv=u1d["value"], r4dv=r4d["value"],
    mfu="\n".join(txt("pmc_mfma_busy_unet3d.txt").splitlines()[:16]),
    bv=hb["value"], bms=hb["ms_per_step"], btf=hb["whole_step_tflops"], bp=pct(hb), b256=h256["value"], bc=hc["value"],
    gbn=gbn["value"], gbe=gb["value"], bk=kern_table(hb), tile=txt("mfma_tile_bf16_random.txt") + "\n" + txt("mfma_tile_bf16_zeros.txt"),
    mfb="\n".join(txt("pmc_mfma_busy_bf16.txt").splitlines()[:12]),
    gv=g["value"], gms=g["ms_per_step"], gp=pct(g), ov=" / ".join(str(o["value"]) for o in others),
    lls=ll["slices"], llr=ll["slices_per_s"], llt=ll["load_seconds"], llth=ll["threads"], llc=ll["chunk"], rss0=ll["peak_rss_gb_before"],
    rss1=ll["peak_rss_gb_after"], stg=ll["staging_gb"], kim=ll["unfilter_kernel_images_per_s"], kms=ll["unfilter_kernel_ms_per_256_images_16bit"],
    t32=table("bench_kernel_stats"), t16=table("bench_bf16_kernel_stats"), t3d=table("bench_unet3d_kernel_stats")))

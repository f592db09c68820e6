#!/usr/bin/env python
"""Benchmark of the U-Net hot path on MI355X: CT slices/s for one full training step
(forward + backward + TF-Adam update [+ RCCL gradient all-reduce when N > 1]) of the 2-D UNet,
Liver+Tumor, 256x256x3, bs 32 per GPU, fp32 (BASELINE.json configs[1]) on synthetic LiTS-shaped data.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W          # self-launching: spawns one rank per GPU (see launch())
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events around every launch of
the conv kernels during the timed region; `cpu_baseline` times the CPU oracle (a restatement, not
the TensorFlow reference -- TF 1.13 cannot run here) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from boxsegliver_amd.utils import hostcpu  # noqa: E402  (no torch inside)

# before torch is imported: the intra-op pool follows the CPUs this process may USE (cgroup quota), not the host's count -- the
# oracle legs (cpu_baseline, dice_vs_oracle) ran 128 threads on a 16-core quota before round 5 (boxsegliver_amd/utils/hostcpu.py)
_USER_SET_OMP = "OMP_NUM_THREADS" in os.environ
hostcpu.size_thread_pools()


def launch(n_gpus, argv):
    """`python bench.py --gpus N` from a plain shell (no torchrun in front, WORLD_SIZE unset): this parent -- which has
    NOT touched HIP (torch is not even imported yet) -- starts `python -m torch.distributed.run` with one fresh child
    process per GPU, relays rank 0's JSON line on stdout (everything else goes to stderr) and exits non-zero if any rank
    failed or no line came back.  The reference is single-process multi-GPU (MirroredStrategy,
    utils/distribution_utils.py:85-98); one process per GPU over RCCL is this package's design."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it)
    if not _USER_SET_OMP:                                   # the ranks share this node's CPU quota
        env["OMP_NUM_THREADS"] = env["MKL_NUM_THREADS"] = str(max(1, hostcpu.usable_cpus() // n_gpus))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for out in proc.stdout:
        if out.startswith('{"metric"'):
            line = out
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if rc != 0:
        raise SystemExit("bench.py: a rank failed (torch.distributed.run exit code {})".format(rc))
    if line is None:
        raise SystemExit("bench.py: the ranks finished but rank 0 printed no result line")
    sys.stdout.write(line)
    sys.stdout.flush()


def _want_launch(argv):
    if "WORLD_SIZE" in os.environ:
        return 0
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=1)
    n = ap.parse_known_args(argv)[0].gpus
    return n if n > 1 else 0


if __name__ == "__main__" and _want_launch(sys.argv[1:]):
    launch(_want_launch(sys.argv[1:]), sys.argv[1:])
    sys.exit(0)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# SURVEY.md 8d / BASELINE.md 3: conv / deconv / 1x1 FLOPs only, 2 per MAC, bwd = dgrad + wgrad
GFLOP_PER_SLICE_FWD_BWD = 288.828
FP32_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32 matrix (= vector) peak
HBM_PEAK_BPS = 8.0e12             # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable by a float4 copy)
BF16_PEAK_TFLOPS = 2516.6         # dense bf16 MFMA: 256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4 GHz (16x the fp32 rate)
METRIC = "CT slices/sec/node (fwd+bwd) UNet 256×256 bs=32; Dice vs ref"


def make_args(bs, num_gpus, size):
    return argparse.Namespace(
        classes=["Liver", "Tumor"], batch_size=bs * num_gpus, num_gpus=num_gpus, im_height=size, im_width=size,
        im_channel=3, normalizer="batch_norm", without_norm=False, weight_init="xavier", weight_decay_rate=1e-6,
        bias_decay=False, loss_type="xentropy", loss_weight_type="numerical", loss_numeric_w=[0.2, 0.4, 4.4],
        loss_proportion_decay=1000, metrics_train=["Dice"], img_grad=False, tag="bench", seed=1234,
        learning_rate=1e-3, learning_policy="plateau", lr_decay_step=100000, lr_decay_rate=0.2,
        num_of_total_steps=600000, lr_power=0.9, lr_end=0.0, lr_decay_boundaries=None, lr_custom_values=None,
        optimizer="Adam", eval_per_epoch=False, noise_scale=0.05, synthetic_batches=2)


YML = dict(init_channels=64, num_down_samples=4, ret_prob=False, ret_pred=True, build_metrics=True,
           build_summaries=False)


def _cpu_oracle_steps(size, n_classes, bs, warm, timed):
    """Median seconds per fwd + bwd + TF-Adam step of the CPU oracle (numpy / PyTorch-CPU restatement; test infrastructure
    used here only as the timed CPU baseline) at `bs` slices of size x size x 3, n_classes classes."""
    from boxsegliver_amd.data.synthetic import make_batch
    from oracle import solver as osolver
    from oracle import unet2d
    net = unet2d.UNet2DOracle(3, n_classes)
    params = unet2d.init_params(net.specs, seed=1234)
    images, labels, _ = make_batch(bs, size, size, 3, n_classes, 1234)
    images, labels = torch.from_numpy(images), torch.from_numpy(labels).long()
    opt = osolver.TFAdam(0.9, 0.99, 1e-8)
    numeric_w = [0.2, 0.4, 4.4][:n_classes]
    kw = dict(loss_type="xentropy", loss_weight_type="numerical", numeric_w=numeric_w, weight_decay_rate=1e-6)
    times = []
    for i in range(warm + timed):
        t0 = time.perf_counter()
        _, _, _, grads, stats = net.loss_and_grads(params, images, labels, **kw)
        opt.step({k: params[k].numpy() for k in grads}, {k: g.numpy() for k, g in grads.items()}, 1e-3)
        for k, v in stats.items():
            params[k] = v
        if i >= warm:
            times.append(time.perf_counter() - t0)
    return statistics.median(times)


def cpu_baseline(size, full=True):
    """SURVEY.md 8d "CPU baseline beside it": the reference's own TF-1.13 CPU path cannot run (TensorFlow is not
    installable), so the labelled substitute is the oracle on this host's cores, fwd+bwd+TF-Adam: the
    configs[1]-shaped workload at bs 2 (3 classes; median of 2 timed steps after one warm-up) = `value`, and BASELINE.json
    configs[0] (Liver only = 2 classes, bs 2; one timed step after one warm-up), and the configs[1] shape at bs 8 (SURVEY.md 8d
    "bs 2 and bs 8") -- ~15 s of CPU work in all since the thread pool follows the CPUs the process may use (round 5: 16 threads
    on a 16-core grant run the bs-2 step in 0.93 s; the 128 throttled threads before took 5.2 s), so the default run is not
    mostly oracle."""
    threads = torch.get_num_threads()
    bs = 2
    t_start = time.perf_counter()
    dt1 = _cpu_oracle_steps(size, 3, bs, 1, 2)
    dt0 = _cpu_oracle_steps(size, 2, bs, 1, 1)
    out = {"value": round(bs / dt1, 4), "unit": "slices/s", "cores": threads, "kind": "port",
           "sample": "oracle (PyTorch-CPU restatement, not TF): UNet {0}x{0}x3 3-class bs {1} (configs[1] shape at bs 2), "
                     "fwd+bwd+Adam, median of 2 timed steps after 1 warm-up, {2:.2f} s/step, {3} torch threads, "
                     "host os.cpu_count()={4}, CPUs granted to this process {5}".format(size, bs, dt1, threads, os.cpu_count(),
                                                                                          hostcpu.usable_cpus()),
           "cfg0": {"value": round(bs / dt0, 4), "unit": "slices/s",
                    "sample": "same, BASELINE.json configs[0]: Liver only (2 classes) bs 2, 1 timed step after 1 warm-up, "
                              "{:.2f} s/step".format(dt0)}}
    if full:
        dt8 = _cpu_oracle_steps(size, 3, 8, 1, 2)
        out["bs8"] = {"value": round(8 / dt8, 4), "unit": "slices/s",
                      "sample": "same workload at bs 8, median of 2 timed steps after 1 warm-up, {:.2f} s/step".format(dt8)}
    out["wall_s"] = round(time.perf_counter() - t_start, 1)
    return out


def dice_vs_oracle(steps=120, size=64, bs=8):
    """BASELINE.json metric "... Dice vs ref": part of the cpu_baseline leg (the only place bench.py touches the oracle), outside
    the timed region.  The headline network (UNet.yml sizes, 3 classes, batch norm, numerical loss weights) is trained for `steps`
    TF-Adam steps at size x size, bs `bs`, from identical variables on the same learnable synthetic stream, once on the HIP path
    (fp32) and once on the oracle in float64 (on the device: the oracle is a torch restatement; as the CHECKER only); then
    Liver/Dice and Tumor/Dice (loss_metrics.py:261-301) on held-out batches.  `max_abs_diff` is what north_star bounds by 1e-3;
    tests/test_gpu_train_dice.py asserts it on a longer run."""
    import numpy as np
    from boxsegliver_amd.NetworksV2.UNet import UNet
    from boxsegliver_amd.core.solver import Solver
    from oracle import train_parity as tp
    from oracle import unet2d
    t0 = time.perf_counter()
    args = make_args(bs, 1, size)
    args.learning_policy, args.tag = "period_step", "dice_vs_oracle"
    train, held = tp.stream(12, bs, size, 2026), tp.stream(4, bs, size, 99)
    dev = lambda b: {"images": torch.from_numpy(b[0]).cuda(), "labels": torch.from_numpy(b[1]).cuda()}   # noqa: E731
    model = UNet(args)
    model(dev(train[0]), "eval", **YML)
    net = unet2d.UNet2DOracle(3, 3, init_channels=YML["init_channels"], num_down_samples=YML["num_down_samples"])
    params = unet2d.init_params(net.specs, seed=11)
    model.params.load_state(params)
    solver = Solver(args)
    for s in range(steps):
        solver(model(dev(train[s % len(train)]), "train", **YML), model)
    hip = {"Liver/Dice": [], "Tumor/Dice": []}
    for b in held:
        with torch.no_grad():
            model(dev(b), "train", **YML)                 # batch statistics, as oracle/train_parity.heldout
        for k in hip:
            hip[k].append(float(model.metrics_dict[k]))
    hip = {k: float(np.mean(v)) for k, v in hip.items()}
    kw = dict(loss_type="xentropy", loss_weight_type="numerical", numeric_w=args.loss_numeric_w,
              weight_decay_rate=args.weight_decay_rate)
    p_ref, _ = tp.train(net, params, train, steps, args.learning_rate, kw, device="cuda", dtype=torch.float64)
    ref, _, _ = tp.heldout(net, p_ref, held, ["Background", "Liver", "Tumor"], device="cuda", dtype=torch.float64)
    return {"max_abs_diff": round(max(abs(hip[k] - ref[k]) for k in ref), 6), "hip_fp32": {k: round(v, 6) for k, v in hip.items()},
            "oracle_f64": {k: round(v, 6) for k, v in ref.items()}, "tolerance": 1e-3,
            "sample": "UNet (64 ch, 4 levels, BN) {0}x{0}x3 bs {1}, {2} TF-Adam steps from identical variables on the same synthetic "
                      "stream; held-out Liver/Dice, Tumor/Dice (batch statistics); oracle in float64 on the device".format(size, bs, steps),
            "wall_s": round(time.perf_counter() - t0, 1)}


def infer_main(a, args, model, inputs_of, data, gflop_unit, workload_name, rank, world):
    """--mode infer: EvaluateVolume's per-slab work (evaluators/evaluator_liver.py `_slab_probability` / `_predict_case`, the
    device-side restatement of the reference's evaluator_liver.py:616-678,704-766) on resident synthetic slabs.  One step =
    one slab of --batch slices: forward in EVAL mode (moving statistics; conv + norm + ReLU [+ pool] fused per unit), softmax,
    mirror variants un-flipped and averaged on the device; every --case-slabs steps a case ends: concatenation, argmax
    (unetk_head_predict) and ONE uint8 device -> host copy.  Forward-only algorithmic FLOPs = a third of fwd+bwd."""
    from boxsegliver_amd import ops
    from boxsegliver_amd.evaluators import evaluator_liver as ev
    from boxsegliver_amd.core import models
    args.eval_mirror, args.random_flip = bool(a.mirror), (3 if a.mirror else 0)
    params = {"args": args, "model": type(model), "model_instances": [model], "model_args": (),
              "model_kwargs": dict(models.get_model_params(args, build_metrics=False)["model_kwargs"])}
    evaluator = ev.EvaluateVolume(estimator=None, model_dir="/nonexistent", params=params)
    feats = []
    for _ in range(2):
        inp = inputs_of(next(data))
        inp.pop("labels", None)
        feats.append(inp)
    n_fwd = 1 + len(evaluator.mirror_variants)
    slabs = []

    def one_step(i):
        slabs.append(evaluator._slab_probability(model, feats[i % 2]))
        if len(slabs) == a.case_slabs:
            volume = torch.cat(slabs)
            amax, _ = ops.head_predict(volume.contiguous(), volume.shape[-1], want_preds=False)
            host = amax.view(volume.shape[:-1]).cpu()           # the case's segmentation: one uint8 copy (a host sync per case)
            del slabs[:]
            return host
        return None

    prof_list = [] if (rank == 0 and not a.no_kernel_events) else None
    ev_stride = 1 if a.steps <= 4 else max(4, -(-a.steps // 3))      # at most three traced steps (they run single-stream and carry the events: ~3 % slower)
    n_ev_steps = len(range(0, a.steps, ev_stride))
    for j in range(a.warmup):
        one_step(j)
    del slabs[:]
    if prof_list is not None:
        ops.profile_begin(4096 * n_ev_steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        if prof_list is not None:
            ops.profile_on(prof_list if i % ev_stride == 0 else None)
        one_step(i)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ops.profile_on(None)
    trace_ms, trace_names = ops.profile_read() if prof_list is not None else ([], [])
    ms = elapsed / a.steps * 1e3
    slices = a.batch * a.steps / elapsed
    fwd_gflop = gflop_unit / 3.0                               # fwd + dgrad + wgrad are three equal contractions
    tfl = slices * n_fwd * fwd_gflop / 1e3
    wl = (workload_name or "UNet 2D Liver+Tumor {0}x{0}x3 bs={1}/GPU fp32").format(a.size, a.batch)
    if a.dtype == "bf16":
        wl = wl.replace(" fp32", " bf16-MFMA + bf16 activation storage")
    out = {"metric": "CT slices/sec (volume evaluation, forward only{})".format(", mirror TTA x4" if a.mirror else ""),
           "value": round(slices, 2), "unit": "slices/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32" if a.dtype == "fp32" else "bf16", "data": "synthetic",
           "config": {"workload": wl + ", EvaluateVolume slab loop (eval mode, moving statistics), {} forward(s) per slab, "
                                     "a case = {} slabs".format(n_fwd, a.case_slabs),
                      "global_batch": a.batch, "parallelism": "dp1", "forwards_per_slab": n_fwd},
           "forward_tflops": round(tfl, 2), "forward_frac_of_fp32_peak": round(tfl / FP32_PEAK_TFLOPS, 4),
           "forward_frac_of_dtype_peak": round(tfl / (FP32_PEAK_TFLOPS if a.dtype == "fp32" else BF16_PEAK_TFLOPS), 4),
           "fused_eval_epilogue": bool(ops.FUSE_EVAL)}
    if prof_list:
        agg, hbm = {}, {}
        for tag, flops, i0, i1, nbytes in prof_list:
            secs = sum(trace_ms[i0:i1]) * 1e-3
            if flops > 0:
                d = agg.setdefault(tag, [0, 0.0, 0.0])
                d[0] += 1; d[1] += flops; d[2] += secs
            elif nbytes:
                h = hbm.setdefault(tag, [0, 0, 0.0])
                h[0] += 1; h[1] += nbytes; h[2] += secs
        out["gpu_kernel_ms_per_step"] = round(sum(trace_ms) / n_ev_steps, 3)
        out["kernels"] = sorted([{"kernel": t, "launches": c, "avg_launch_ms": round(sc / c * 1e3, 4),
                                  "achieved_tflops": round(f / sc / 1e12, 2), "total_ms_per_step": round(sc / n_ev_steps * 1e3, 3)}
                                 for t, (c, f, sc) in agg.items() if sc > 0], key=lambda k: -k["total_ms_per_step"])
        out["hbm_kernels"] = sorted([{"kernel": t, "launches": c, "avg_launch_ms": round(sc / c * 1e3, 4),
                                      "achieved_gbps": round(nb / sc / 1e9, 1), "total_ms_per_step": round(sc / n_ev_steps * 1e3, 3)}
                                     for t, (c, nb, sc) in hbm.items() if sc > 0], key=lambda k: -k["total_ms_per_step"])
        top = out["kernels"][0]
        kpeak = BF16_PEAK_TFLOPS if "bf16" in top["kernel"] else FP32_PEAK_TFLOPS
        out["roofline"] = {"bound": "mfma", "kernel": top["kernel"], "achieved": top["achieved_tflops"], "peak": kpeak,
                           "unit": "TFLOP/s", "frac": round(top["achieved_tflops"] / kpeak, 4), "traffic": None}
    print(json.dumps(out, ensure_ascii=False), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="slices per GPU per step")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--depth", type=int, default=0,
                    help="UNet3D: patch depth (default = --size, i.e. a cube); the reference's own 3-D script trains "
                         "--depth 10 --size 256 --batch 4 (threed_script/201_unet_v1.sh:26)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="kept for old command lines: the bs 8 leg is part of the default since round 5")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip per-launch HIP-event timing")
    ap.add_argument("--settle-seconds", type=float, default=45.0,
                    help="upper bound of the untimed settle phase in front of the warm-up (blocks of ten steps until two in a "
                         "row show no one-step hiccup of the box); 0 = none")
    ap.add_argument("--single-stream", action="store_true",
                    help="every step on ONE stream (filter gradients otherwise run beside input gradients on a second stream): "
                         "what the traced steps do anyway; for rocprofv3 runs whose per-kernel averages are compared with the line's")
    ap.add_argument("--detail", action="store_true", help="break the kernel table down by layer shape")
    ap.add_argument("--model", default="UNet", choices=["UNet", "GUNet", "UNet3D", "UNetInter", "LGNet", "SmallUNet", "InterUNet"],
                    help="UNet = the headline workload (BASELINE.json configs[1]); GUNet / UNet3D = configs[3] / [4] "
                         "(use --batch 8 / --size 96 --batch 1..4)")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16", "bf16c"],
                    help="fp32 = the headline configuration (exact fp32 MFMA); bf16 = BASELINE.json configs[2]'s mode: bf16 "
                         "matrix cores + bf16 storage of activations / activation gradients, fp32 accumulate / statistics / "
                         "master weights (use --size 512 --batch 8); bf16c = bf16 MFMA operands only, fp32 storage (round 1)")
    ap.add_argument("--mode", default="train", choices=["train", "infer"],
                    help="train = the headline metric (fwd+bwd+optimiser); infer = the volume evaluator's slab loop "
                         "(evaluators/evaluator_liver.py:616-766: forward with moving statistics, softmax, mirror un-flip / "
                         "averaging, per-case argmax + one uint8 copy to the host) on synthetic slabs: slices/s, forward only")
    ap.add_argument("--mirror", action="store_true", help="--mode infer: --eval_mirror --random_flip 3 (the un-mirrored slab + "
                                                          "three flipped variants, averaged: evaluator_liver.py:648-655)")
    ap.add_argument("--case-slabs", type=int, default=8, help="--mode infer: slabs (steps) per case; a case ends with cat + argmax + D2H")
    ap.add_argument("--dp-rehearsal", action="store_true",
                    help="with --gpus 1: run the data-parallel code path (process group on RCCL, bucketed all-reduce launched from "
                         "backward, 1/N in the optimiser) in a world of ONE rank -- the production path on the one GPU a test box has")
    ap.add_argument("--launch-check", action="store_true",
                    help="self-test of the multi-rank launch only (no kernels, no GPU needed): every rank joins the process "
                         "group, rank 0 prints the ranks it saw as a `launch-check` line")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit("bench.py --gpus {} inside a {}-rank launch (WORLD_SIZE): the two must agree".format(a.gpus, world))
    if a.launch_check:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        seen = [rank]
        if world > 1:
            dist.init_process_group(backend="gloo")
            got = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(got, torch.tensor([rank], dtype=torch.int64))
            seen = sorted(int(g.item()) for g in got)
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": "launch-check", "value": len(seen), "unit": "ranks", "n_gpus": a.gpus,
                              "n_ranks_seen": len(seen), "ranks": seen}), flush=True)
        return
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (the product has no CPU path)"
    # one rank per GPU; the modulo only matters for rehearsals of the multi-rank path on a box with fewer GPUs than
    # ranks (UNETK_DIST_BACKEND=gloo: RCCL refuses two ranks on one device)
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    backend = os.environ.get("UNETK_DIST_BACKEND", "nccl")                               # "nccl" = RCCL over xGMI
    dp_on = world > 1 or a.dp_rehearsal
    if a.dp_rehearsal and world == 1:
        import socket as _socket
        s_ = _socket.socket()
        s_.bind(("127.0.0.1", 0))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(s_.getsockname()[1]))
        s_.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend=backend, rank=0, world_size=1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl" and torch.cuda.device_count() < world:
            raise SystemExit("bench.py --gpus {}: only {} GPU(s) visible; RCCL needs one device per rank "
                             "(UNETK_DIST_BACKEND=gloo rehearses the multi-rank path on fewer)".format(world, torch.cuda.device_count()))
        dist.init_process_group(backend=backend)

    from boxsegliver_amd import ops
    from boxsegliver_amd.core import models
    from boxsegliver_amd.core.solver import Solver
    from boxsegliver_amd.data import synthetic
    from boxsegliver_amd.utils.distribution_utils import DistributionStrategy

    global YML
    args = make_args(a.batch, world, a.size)
    args.model, args.model_config = a.model, None
    args.compute_dtype = a.dtype
    input_fn = synthetic.input_fn
    gflop_unit, workload_name = GFLOP_PER_SLICE_FWD_BWD * (a.size / 256.0) ** 2, None
    if a.model == "GUNet":          # BASELINE.json configs[3]: spatial guide, instance norm (002_gnet.sh:39)
        args.use_spatial, args.use_context, args.guide_channel, args.normalizer = True, False, 1, "instance_norm"
        args.side_dropout, args.dropout, args.use_se, args.fix = 0.5, None, False, False
        gflop_unit = 288.9 * (a.size / 256.0) ** 2
        workload_name = "GUNet + 1-ch spatial guide {0}x{0}x3 bs={1}/GPU fp32 instance_norm (BASELINE.json configs[3])"
    elif a.model == "UNet3D":       # BASELINE.json configs[4]: 96^3 patches, instance norm (201_unet_v1.sh:39)
        depth = a.depth or a.size
        args.classes, args.im_channel, args.im_depth, args.normalizer = ["NF"], 1, depth, "instance_norm"
        args.loss_numeric_w, args.use_spatial, args.weight_decay_rate = [1.0, 1.0], False, 3e-5
        input_fn = synthetic.input_fn_3d
        gflop_unit = 1612.92 * (depth * a.size * a.size / 96.0 ** 3)       # every level's voxel count scales with D * H * W
        workload_name = "UNet3D " + str(depth) + "x{0}x{0}x1 bs={1}/GPU fp32 instance_norm " + \
            ("(BASELINE.json configs[4])" if depth == a.size == 96 else
             "(the reference's 3-D training shape, threed_script/201_unet_v1.sh)" if (depth, a.size) == (10, 256) else "(off-config shape)")
    elif a.model in ("UNetInter", "LGNet", "SmallUNet", "InterUNet"):     # the other MODEL_ZOO nets: guide as a second input
        args.use_spatial, args.use_context, args.guide_channel, args.normalizer = True, False, 1, "instance_norm"
        args.side_dropout, args.dropout, args.use_se, args.fix, args.mid_cat = 0.5, None, False, False, False
        gflop_unit = None                                                 # algorithmic FLOPs summed from the kernel events
        workload_name = a.model + " + 1-ch guide {0}x{0}x3 bs={1}/GPU fp32 instance_norm (not a BASELINE.json config)"
    YML = models.get_model_params(args, build_metrics=True)["model_kwargs"]
    # UNet3D runs its filter gradients on a second stream beside the input gradients (ops.SIDE_WGRAD3D_VOXELS): kernels then
    # overlap in time and stretch each other, so their durations no longer add up to the step.  --detail (the by-layer table)
    # switches that off to attribute time to layers; `value` of a --detail run is therefore the single-stream step.
    # Round 5: the fp32 2-D units do the same (ops._Side: + 1.6 % on the headline step), and every TRACED step (each fourth) runs
    # on one stream, so the kernel tables, the roofline block and kernels_fit_step describe single-stream steps while `value`
    # is over all steps.
    prec_flag = {"fp32": 0, "bf16c": 1, "bf16": 2}[a.dtype]
    one_stream = bool(a.detail or a.single_stream)
    side3d = (not one_stream) and ((a.model == "UNet3D" and ops.SIDE_WGRAD3D_VOXELS > 0) or
                                 (a.model != "UNet3D" and ops.side_wgrad_on(prec_flag)))
    if one_stream:
        ops.side_streams_pause(True)
    params = {"args": args, "rank": rank, "device": torch.device("cuda", torch.cuda.current_device())}
    data = input_fn("train", params)
    model = {c.__name__: c for c in models.MODEL_ZOO}[a.model](args)
    solver = Solver(args)
    strategy = DistributionStrategy("mirrored", world, rank) if dp_on else None
    solver.strategy = strategy
    solver.dp_rehearsal = bool(a.dp_rehearsal)

    def inputs_of(batch):
        features, labels = batch
        inp = {k: v for k, v in features.items() if k != "names"}
        inp["labels"] = labels
        return inp

    model(inputs_of(next(data)), "eval", **YML)                                   # create variables
    if a.mode == "infer":
        return infer_main(a, args, model, inputs_of, data, gflop_unit, workload_name, rank, world)
    if strategy is not None:
        strategy.broadcast_(list(model.params.flat.values()))                      # identical replicas

    def one_step():
        loss = model(inputs_of(next(data)), "train", **YML)
        solver(loss, model)
        return loss

    # The roofline block: on every fourth step of the timed region the library's kernel trace is on (csrc/prof.hip): each
    # launch then carries start / stop events bound to its DISPATCH (hipExtLaunchKernelGGL), so a kernel's duration is the GPU's
    # own begin -> end interval -- what rocprofv3 --kernel-trace reports -- and no host gap can land in it (round 3 bracketed
    # the C-ABI call with hipEventRecord pairs, which swallow the host's time whenever the stream has drained:
    # tools/probe_ext_events.hip).  A traced launch costs ~7 us more host time, hence the sampling; the events are created
    # before the timed region, sized by the launch count of the LAST warm-up step (traced for that purpose).
    prof_list = [] if (rank == 0 and not a.no_kernel_events) else None
    ev_stride = 1 if a.steps <= 4 else max(4, -(-a.steps // 3))      # at most three traced steps (they run single-stream and carry the events: ~3 % slower)
    n_ev_steps = len(range(0, a.steps, ev_stride))
    ops.PROFILE_SHAPES = bool(a.detail)
    launches_per_step = 4096
    # Settle phase (in front of the W warm-up steps, untimed like them; N ranks agree block by block): on a box of this pool the second to fourth
    # GPU process after its start see 0.4-1 s gaps on the GPU's own timeline every second or two, whatever the library
    # (profiles/r05_step_hiccups.txt: detect 437 / 274 / 266, then 437 for every later process).  Blocks of ten steps with an event
    # per step until two blocks in a row show no step above three times the block's median (and 100 ms above it), or `--settle-seconds` (default 45) have
    # passed.  The timed region below is unchanged: exactly K steps between barrier + synchronize.
    settle = {"steps": 0, "hiccups": 0, "seconds": 0.0}
    if a.settle_seconds > 0 and a.mode == "train":
        t_settle, clean, timed_out = time.perf_counter(), 0, False
        while clean < 2 and not timed_out:
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
            evs[0].record()
            for i in range(10):
                one_step()
                evs[i + 1].record()
            torch.cuda.synchronize()
            ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(10)]
            med = statistics.median(ms)
            bad = sum(1 for v in ms if v > 3.0 * med and v - med > 100.0)      # the gaps are 400 ms and more; launch jitter of tiny steps is not
            timed_out = time.perf_counter() - t_settle >= a.settle_seconds
            if world > 1:       # every rank takes the same blocks (the steps carry the gradient all-reduce): agree on what was seen
                flags = torch.tensor([float(bad), 1.0 if timed_out else 0.0], dtype=torch.float32,
                                     device="cuda" if backend == "nccl" else "cpu")
                dist.all_reduce(flags, op=dist.ReduceOp.MAX)
                bad, timed_out = int(flags[0].item()), bool(flags[1].item() > 0)
            settle["steps"] += 10
            settle["hiccups"] += bad
            clean = 0 if bad else clean + 1
        settle["seconds"] = round(time.perf_counter() - t_settle, 2)
    for j in range(a.warmup):
        if j == a.warmup - 1 and prof_list is not None:
            ops.profile_begin(0)
            ops.profile_on([])
        one_step()
    if prof_list is not None:
        ops.profile_on(None)
        torch.cuda.synchronize()
        if a.warmup > 0:
            launches_per_step = ops._abi.lib().unetk_prof_mark()
        ops.profile_begin(launches_per_step * n_ev_steps + 64)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]   # per-step HIP events (no host sync)
    host_ms = []
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(a.steps):
        if prof_list is not None:
            ops.profile_on(prof_list if i % ev_stride == 0 else None)
            # traced steps run on ONE stream: the filter gradients otherwise run beside the input gradients on a second
            # stream (ops._Side) and the two stretch each other -- durations that no longer add up to the step.  The other
            # steps run as production does; `value` is over all of them.
            ops.side_streams_pause(one_stream or i % ev_stride == 0)
        h0 = time.perf_counter()
        loss = one_step()
        host_ms.append((time.perf_counter() - h0) * 1e3)
        marks[i + 1].record()
    torch.cuda.synchronize()
    own = time.perf_counter() - t0                  # this rank's own time for its K steps
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    prof = prof_list
    ops.profile_on(None)
    ops.side_streams_pause(one_stream)
    loss_val = float(loss.detach())
    trace_ms, trace_names = ops.profile_read() if prof is not None else ([], [])
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)]
    rank_ms, n_ranks_seen = [own / a.steps * 1e3], 1
    if world > 1:
        n_ranks_seen = dist.get_world_size()
        tdev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)      # the job's time = the slowest rank's
        elapsed = float(t.item())
        mine = torch.tensor([own / a.steps * 1e3], dtype=torch.float64, device=tdev)
        every = [torch.zeros_like(mine) for _ in range(n_ranks_seen)]
        dist.all_gather(every, mine)
        rank_ms = [float(v.item()) for v in every]

    # ---- data-parallel diagnostics (N > 1): what the buckets did, how long the compute stream waited for them, and this
    # rank's step WITHOUT the all-reduce (3 extra untimed-by-the-metric steps, replicas may drift afterwards: the run is over)
    dp_diag = None
    if dp_on:
        bk = getattr(solver, "_buckets", None)
        dp_diag = {"buckets": len(bk.buckets) if bk is not None else 0,
                   "buckets_fired_in_backward": bk.last["fired_in_backward"] if bk is not None and bk.last else None,
                   "bucket_launch_progress": bk.last["launch_progress"] if bk is not None and bk.last else None,
                   "bucket_bytes": bk.last["bucket_bytes"] if bk is not None and bk.last else None,
                   "allreduce_exposed_ms": round(bk.exposed_ms(), 4) if bk is not None and bk.exposed_ms() is not None else None,
                   "allreduce_bytes": int(sum(g.numel() for g in model.params.grad.values()) * 4)}
        solver.strategy = None
        if bk is not None:
            bk.remove()
            solver._buckets = None
        # EXACTLY the timed region again -- the same number of steps, the same steps traced (events, one stream) -- without the
        # all-reduce path: the difference must be that path alone.  (Round 4 ran four steps with one traced: with three traced
        # single-stream steps of ten in the timed region the two shares no longer matched and the ratio read 0.95.)
        one_step()
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for j in range(a.steps):
            if prof is not None:
                ops.profile_on([] if j % ev_stride == 0 else None)
                ops.side_streams_pause(one_stream or j % ev_stride == 0)
            one_step()
        torch.cuda.synchronize()
        dp_diag["compute_only_ms_per_step"] = round((time.perf_counter() - tc) / a.steps * 1e3, 3)
        ops.profile_on(None)
        ops.side_streams_pause(one_stream)
        dp_diag["backend"] = backend
        dist.barrier()
        # ... and the data-parallel region once more BEHIND it (not part of the metric): DP / compute-only / DP -- the mean of the
        # two DP timings against the compute-only one in between cancels the drift of a clock-governed step over the run (the bf16
        # step moves by 1-2 % within one process; single comparisons read 0.977-0.993 on different boxes)
        try:
            if world > 1:                                # rehearsal only: a diagnostic must not add collectives to a real N-rank run
                raise RuntimeError("skipped at world > 1")
            solver.strategy = strategy
            one_step()                                   # re-creates the buckets
            torch.cuda.synchronize()
            dist.barrier()
            tc = time.perf_counter()
            for j in range(a.steps):
                if prof is not None:
                    ops.profile_on([] if j % ev_stride == 0 else None)
                    ops.side_streams_pause(one_stream or j % ev_stride == 0)
                one_step()
            torch.cuda.synchronize()
            dist.barrier()
            dp_diag["dp_again_ms_per_step"] = round((time.perf_counter() - tc) / a.steps * 1e3, 3)
        except Exception as e:                           # diagnostics only: never lose the line over them
            if world == 1:
                dp_diag["dp_again_error"] = repr(e)[:200]
        ops.profile_on(None)
        ops.side_streams_pause(one_stream)

    if rank == 0:
        ms = elapsed / a.steps * 1e3
        slices = a.batch * world * a.steps / elapsed
        if workload_name is None:
            cfg = "configs[1]" if (a.size == 256 and a.batch == 32 and a.dtype == "fp32") else \
                ("configs[2] shape" if (a.size == 512 and a.batch == 8) else "off-config shape")
            workload_name = "UNet 2D Liver+Tumor {0}x{0}x3 bs={1}/GPU fp32 (BASELINE.json " + cfg + ")"
        wl = workload_name.format(a.size, a.batch) + ", fwd+bwd+TF-Adam" + ("+{} grad all-reduce".format("RCCL" if backend == "nccl" else backend) if world > 1 else "")
        if a.dtype == "bf16":
            wl = wl.replace(" fp32", " bf16-MFMA + bf16 activation storage / fp32 accumulate, statistics, master weights")
        elif a.dtype == "bf16c":
            wl = wl.replace(" fp32", " bf16-MFMA operands / fp32 storage")
        peak = FP32_PEAK_TFLOPS if a.dtype == "fp32" else BF16_PEAK_TFLOPS
        if gflop_unit is None:      # conv / deconv algorithmic FLOPs (2 per MAC, fwd + dgrad + wgrad) of one unit, from the events
            gflop_unit = (sum(e[1] for e in prof) / n_ev_steps / a.batch / 1e9) if prof else 0.0
        out = {
            "metric": METRIC if a.model == "UNet" else "{} units/sec/node (fwd+bwd)".format(a.model),
            "value": round(slices, 2), "unit": "slices/s" if a.model != "UNet3D" else "patches/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "kernel_event_steps": (n_ev_steps if prof is not None else 0),
            "ms_per_step_median_hipevents": round(statistics.median(step_ms), 3),
            # a hiccup of the box (one step far above the median: seen as 0.5 s GPU-side gaps on some boxes of the pool, whatever
            # the library) shows here and in `value`, which is wall time over all K steps by contract
            "ms_per_step_max_hipevents": round(max(step_ms), 3), "ms_sum_steps_hipevents": round(sum(step_ms), 3),
            "settle": settle,
            "n_ranks_seen": n_ranks_seen,
            "rank_ms_per_step_min": round(min(rank_ms), 3), "rank_ms_per_step_max": round(max(rank_ms), 3),
            "dist_backend": (backend if world > 1 else None), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if a.dtype == "fp32" else "bf16", "data": "synthetic",
            "compute_dtype_flag": a.dtype,
            "config": {"workload": wl, "global_batch": a.batch * world, "parallelism": "dp{}".format(world),
                       "classes": len(args.classes) + 1, "final_loss": round(loss_val, 5)},
            "whole_step_tflops": round(slices * gflop_unit / 1e3, 2),
            "whole_step_frac_of_fp32_peak": round(slices * gflop_unit / 1e3 / (FP32_PEAK_TFLOPS * world), 4),
            "whole_step_frac_of_dtype_peak": round(slices * gflop_unit / 1e3 / (peak * world), 4),
            "side_stream_filter_gradients": bool(side3d),      # True: on the untraced steps filter gradients run beside input gradients
            "traced_steps_single_stream": True,
        }
        if dp_diag is not None:
            dp_diag["dp_efficiency_vs_compute_only"] = round(dp_diag["compute_only_ms_per_step"] / ms, 4)
            if dp_diag.get("dp_again_ms_per_step"):      # drift-cancelled: compute-only against the mean of the DP regions around it
                dp_diag["dp_efficiency_drift_cancelled"] = round(
                    dp_diag["compute_only_ms_per_step"] / (0.5 * (ms + dp_diag["dp_again_ms_per_step"])), 4)
            out["data_parallel"] = dp_diag
        ev_steps = [step_ms[i] for i in range(a.steps) if i % ev_stride == 0]
        plain_steps = [step_ms[i] for i in range(a.steps) if i % ev_stride != 0] or ev_steps
        out["step_ms_event_steps"] = round(statistics.median(ev_steps), 3)      # steps whose launches carried events
        out["step_ms_plain_steps"] = round(statistics.median(plain_steps), 3)   # steps without
        # host time to ENQUEUE a step (no sync inside): far below the step time = the GPU never waits for the host
        host_plain = [host_ms[i] for i in range(a.steps) if i % ev_stride != 0] or host_ms
        out["host_ms_per_step"] = round(statistics.median(host_plain), 3)
        out["host_ms_per_traced_step"] = round(statistics.median([host_ms[i] for i in range(a.steps) if i % ev_stride == 0]), 3)
        if prof:
            # op brackets -> (tag, FLOPs, bytes, kernel time = sum of the dispatches the call launched, its longest kernel)
            agg, hbm, by_name, main_of = {}, {}, {}, {}
            covered = 0.0
            for tag, flops, i0, i1, nbytes in prof:
                ms_in = trace_ms[i0:i1]
                secs = sum(ms_in) * 1e-3
                covered += secs
                if flops > 0:
                    d = agg.setdefault(tag, [0, 0.0, 0.0])
                    d[0] += 1
                    d[1] += flops
                    d[2] += secs
                    if ms_in:       # the call's matrix kernel = its longest dispatch, when that IS the call (a filter gradient and
                        j = max(range(len(ms_in)), key=ms_in.__getitem__)     # its slab reduction; not the three GEMMs of a
                        if ms_in[j] >= 0.8 * sum(ms_in):                      # transposed conv's backward: those stay by op tag)
                            main_of[i0 + j] = flops
                if nbytes:                                    # HBM-bound passes: algorithmic bytes (each operand once) / time
                    h = hbm.setdefault(tag, [0, 0, 0.0])
                    h[0] += 1
                    h[1] += nbytes
                    h[2] += secs
            for j, (ms_j, name) in enumerate(zip(trace_ms, trace_names)):
                r = by_name.setdefault(name, [0, 0.0, 0.0])
                r[0] += 1
                r[1] += ms_j
                r[2] += main_of.get(j, 0.0)
            total_kernel_ms = sum(trace_ms) / n_ev_steps
            out["traced_launches_per_step"] = len(trace_ms) // n_ev_steps
            out["gpu_kernel_ms_per_step"] = round(total_kernel_ms, 3)           # every kernel of the library, own GPU time
            out["gpu_kernel_ms_outside_op_brackets"] = round(total_kernel_ms - covered * 1e3 / n_ev_steps, 3)
            out["hbm_kernels"] = sorted(
                [{"kernel": tag, "launches": cnt, "avg_launch_ms": round(secs / cnt * 1e3, 4), "avg_launch_mbytes": round(nb / cnt / 1e6, 2),
                  "achieved_gbps": round(nb / secs / 1e9, 1), "frac_of_hbm_peak": round(nb / secs / HBM_PEAK_BPS, 4),
                  "total_ms_per_step": round(secs / n_ev_steps * 1e3, 3)} for tag, (cnt, nb, secs) in hbm.items() if secs > 0],
                key=lambda k: -k["total_ms_per_step"])
            kern = []
            for tag, (cnt, flops, secs) in agg.items():
                kern.append({"kernel": tag, "launches": cnt, "avg_launch_ms": round(secs / cnt * 1e3, 4),
                             "avg_launch_gflop": round(flops / cnt / 1e9, 3),
                             "achieved_tflops": round(flops / secs / 1e12, 2), "total_ms_per_step": round(secs / n_ev_steps * 1e3, 3)})
            kern.sort(key=lambda k: -k["total_ms_per_step"])
            # the same launches by their REAL kernel names (one row per instantiation, as rocprofv3 --stats prints them)
            trace = []
            for name, (cnt, ms_sum, flops) in by_name.items():
                row = {"name": name, "calls": cnt, "avg_ms": round(ms_sum / cnt, 4), "total_ms_per_step": round(ms_sum / n_ev_steps, 3)}
                if flops > 0:
                    row["avg_gflop"] = round(flops / cnt / 1e9, 3)
                    row["achieved_tflops"] = round(flops / ms_sum / 1e9, 2)
                trace.append(row)
            trace.sort(key=lambda r: -r["total_ms_per_step"])
            sum_tables = sum(k["total_ms_per_step"] for k in kern) + sum(k["total_ms_per_step"] for k in out["hbm_kernels"]
                                                                         if k["kernel"] not in agg)
            out["sum_kernels_plus_hbm_kernels_ms"] = round(sum_tables, 3)
            # against the traced steps' own duration (they run on one stream; the untraced ones may overlap kernels and be shorter)
            out["kernels_fit_step"] = bool(sum_tables <= out["step_ms_event_steps"] and total_kernel_ms <= out["step_ms_event_steps"])
            top = next(r for r in trace if "achieved_tflops" in r)         # the matrix kernel with the most time in the step
            kpeak = BF16_PEAK_TFLOPS if "bf16" in top["name"] else FP32_PEAK_TFLOPS
            out["roofline"] = {"bound": "mfma", "kernel": top["name"], "achieved": top["achieved_tflops"],
                               "peak": kpeak, "unit": "TFLOP/s",
                               "frac": round(top["achieved_tflops"] / kpeak, 4), "traffic": None,
                               "avg_launch_ms": top["avg_ms"], "avg_launch_gflop": top["avg_gflop"], "launches": top["calls"],
                               "timing": "start/stop events bound to each dispatch (hipExtLaunchKernelGGL) on the launch stream, "
                                         "{} of the {} timed steps".format(n_ev_steps, a.steps)}
            out["kernels"] = kern
            out["kernel_trace"] = trace[:24]
            # HBM traffic per launch of the dominant kernel: PMC counters cannot be read from inside the process, so it
            # comes from the committed rocprofv3 --pmc summary of this same command (profiles/rNN_pmc_traffic*.json,
            # tools/pmc_summary.py; newest round wins) and ONLY when that file names this exact kernel -- else null.
            try:
                import glob
                # one PMC pass pair per measured configuration (tools/refresh_profiles.sh prof): (model, dtype, size, batch) -> file suffix
                pmc_cfgs = {("UNet", "fp32", 256, 32, 0): "", ("UNet", "bf16", 512, 8, 0): "_bf16",
                            ("UNet3D", "fp32", 96, 1, 0): "_unet3d", ("GUNet", "fp32", 256, 8, 0): "_gunet"}
                suffix = pmc_cfgs.get((a.model, a.dtype, a.size, a.batch, a.depth))
                shape_ok = suffix is not None
                files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic" + (suffix or "") + ".json")))
                pmc = json.load(open(files[-1]))["kernels"] if files and shape_ok else {}
                # the PMC table is keyed like rocprofv3's kernel names minus "void ", the anonymous namespace and the arguments
                want = top["name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                hit = pmc.get(want)
                key = (hit["launches"], hit["launches"] * hit["hbm_bytes_per_launch_corrected"]) if hit else None
                if key and key[0] and shape_ok:
                    out["roofline"]["traffic"] = round(key[1] / key[0])
                    out["roofline"]["traffic_source"] = "profiles/" + os.path.basename(files[-1]) + \
                        " (rocprofv3 --pmc, FETCH_SIZE x2 + WRITE_SIZE)"
            except (OSError, KeyError, ValueError, IndexError):
                pass
        if not a.no_cpu_baseline and world == 1 and a.model == "UNet":
            out["cpu_baseline"] = cpu_baseline(a.size)
            if a.dtype == "fp32":
                out["dice_vs_oracle"] = dice_vs_oracle()
        print(json.dumps(out, ensure_ascii=False), flush=True)
    if dp_on:
        dist.destroy_process_group()


def _main_with_clean_stdout():
    """stdout carries the ONE JSON line and nothing else: libraries that print to file descriptor 1 on their own (RCCL's version
    banner at communicator creation, amdgpu.ids notices) are sent to stderr while the benchmark runs."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    real = os.fdopen(saved, "w", buffering=1)
    sys.stdout = real
    try:
        main()
    finally:
        sys.stdout.flush()


if __name__ == "__main__":
    _main_with_clean_stdout()

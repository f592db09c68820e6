"""Data-parallel path: one process per GPU, gradients of the two flat buffers summed
with one all-reduce each, 1/N folded into the optimiser, replica-local BN statistics.

* CPU (gloo, world_size 2): the host logic -- DistributionStrategy collectives and
  Solver.apply_gradients' sum-then-1/N semantics (TF MirroredStrategy: loss x 1/N, gradients summed,
  core/estimator.py:570-578) with a test double for the HIP Adam kernel.
* GPU (-m gpu; 2 ranks sharing the one card, gloo over device tensors): a real 2-replica UNet run
  equals the single-process emulation that applies BN per half-batch (the DP parity definition)."""
import argparse
import math
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _solver_args():
    return argparse.Namespace(
        learning_rate=1e-3, learning_policy="period_step", lr_decay_step=100000, lr_decay_rate=0.1,
        num_of_total_steps=1000, lr_power=0.9, lr_end=1e-6, lr_decay_boundaries=None, lr_custom_values=None,
        optimizer="Adam")


class _FakeStore(object):
    def __init__(self, n):
        self.flat = {"reg": torch.linspace(-1, 1, n), "noreg": torch.ones(8)}
        self.grad = {"reg": torch.zeros(n), "noreg": torch.zeros(8)}


def _cpu_adam(p, g, m, v, lr_t, b1, b2, eps, gscale=1.0, l2=0.0, decoupled_wd=0.0):
    gg = g * gscale + l2 * p
    m += (1 - b1) * (gg - m)
    v += (1 - b2) * (gg * gg - v)
    p.mul_(1.0 - decoupled_wd)
    p -= lr_t * m / (v.sqrt() + eps)


def _cpu_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from boxsegliver_amd import ops
    from boxsegliver_amd.core.solver import Solver
    from boxsegliver_amd.utils.distribution_utils import DistributionStrategy
    ops.adam_step = _cpu_adam                       # test double for the HIP kernel (no GPU here)
    strategy = DistributionStrategy("mirrored", world, rank)
    store = _FakeStore(37)
    if rank != 0:
        store.flat["reg"].add_(5.0)                 # replicas start different ...
    strategy.broadcast_(list(store.flat.values()))  # ... and are made identical
    solver = Solver(_solver_args())
    solver.strategy = strategy
    for step in range(3):
        gen = torch.Generator().manual_seed(100 * step + rank)
        store.grad["reg"].copy_(torch.randn(37, generator=gen))
        store.grad["noreg"].copy_(torch.randn(8, generator=gen))
        solver.apply_gradients(store, 1e-2, 1e-3)
    mean_loss = strategy.reduce_mean(torch.tensor(float(rank + 1)))
    torch.save({"reg": store.flat["reg"], "noreg": store.flat["noreg"], "mean": mean_loss},
               os.path.join(out_dir, "r{}.pt".format(rank)))
    dist.destroy_process_group()


def test_dp_host_logic_gloo_world2():
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_cpu_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        r = [torch.load(os.path.join(d, "r{}.pt".format(i))) for i in range(world)]
    assert torch.equal(r[0]["reg"], r[1]["reg"]) and torch.equal(r[0]["noreg"], r[1]["noreg"])
    assert r[0]["mean"].item() == pytest.approx(1.5)
    # single-process reference: Adam on the MEAN of the replica gradients
    p = {"reg": torch.linspace(-1, 1, 37), "noreg": torch.ones(8)}
    st = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in p.items()}
    for step in range(3):
        gs = []
        for rank in range(world):
            gen = torch.Generator().manual_seed(100 * step + rank)
            gs.append((torch.randn(37, generator=gen), torch.randn(8, generator=gen)))
        t = step + 1
        lr_t = 1e-3 * math.sqrt(1 - 0.99 ** t) / (1 - 0.9 ** t)
        _cpu_adam(p["reg"], (gs[0][0] + gs[1][0]) / 2, *st["reg"], lr_t, 0.9, 0.99, 1e-8, 1.0, 1e-2)
        _cpu_adam(p["noreg"], (gs[0][1] + gs[1][1]) / 2, *st["noreg"], lr_t, 0.9, 0.99, 1e-8, 1.0, 0.0)
    np.testing.assert_allclose(r[0]["reg"].numpy(), p["reg"].numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(r[0]["noreg"].numpy(), p["noreg"].numpy(), rtol=1e-6, atol=1e-7)


SPECS = [("T/a/weights", (3, 3, 2, 4), "conv_w"), ("T/a/beta", (4,), "beta"), ("T/b/weights", (3, 3, 4, 5), "conv_w"),
         ("T/b/gamma", (5,), "gamma"), ("T/unused/weights", (1, 1, 5, 7), "conv_w"), ("T/c/weights", (2, 2, 3, 5), "deconv_w"),
         ("T/c/biases", (3,), "bias"), ("T/a/moving_mean", (4,), "moving_mean")]


class _ToyModel(object):
    """A ParamStore-backed stand-in for a plugin: plain torch ops on the CPU, one variable never used."""

    def __init__(self):
        from boxsegliver_amd.NetworksV2.base import ParamStore
        self.params = ParamStore(SPECS, torch.device("cpu"))
        self.params.initialize("xavier", seed=3)

    def _get_regularizer(self):
        return 1e-2, None

    def loss(self, seed):
        gen = torch.Generator().manual_seed(seed)
        p, total = self.params, 0.0
        for name in ("T/a/weights", "T/a/beta", "T/b/weights", "T/b/gamma", "T/c/weights", "T/c/biases"):
            x = torch.randn(p[name].shape, generator=gen)
            total = total + ((p[name] * x).sum() + 0.3) ** 2
        return total


def _bucket_worker(rank, world, port, out_dir, overlap):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from boxsegliver_amd import ops
    from boxsegliver_amd.core.solver import Solver
    from boxsegliver_amd.utils.distribution_utils import DistributionStrategy
    ops.adam_step = _cpu_adam
    model = _ToyModel()
    solver = Solver(_solver_args())
    solver.strategy = DistributionStrategy("mirrored", world, rank)
    solver.overlap_allreduce = overlap
    solver.bucket_bytes = 256                      # several buckets per buffer
    fired = []
    for step in range(3):
        solver(model.loss(10 * step + rank), model)
        if overlap:
            fired.append(list(solver._buckets._fired))
    torch.save({"flat": model.params.flat, "n_buckets": len(solver._buckets.buckets) if overlap else 0, "fired": fired},
               os.path.join(out_dir, "r{}.pt".format(rank)))
    dist.destroy_process_group()


def test_bucketed_overlapped_allreduce_equals_plain_allreduce_gloo_world2():
    """GradBuckets: hooks fire the per-bucket all-reduce during backward; the result is bit-identical to one
    all-reduce per buffer after backward, covers a variable that never receives a gradient, and equals Adam on
    the mean of the replica gradients."""
    world, res = 2, {}
    for overlap in (True, False):
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(_bucket_worker, args=(world, _free_port(), d, overlap), nprocs=world, join=True)
            res[overlap] = [torch.load(os.path.join(d, "r{}.pt".format(i))) for i in range(world)]
    assert res[True][0]["n_buckets"] >= 3          # filter buckets; the norm parameters ride in the last one (round 5)
    assert all(all(f) for f in res[True][0]["fired"])
    for g in ("reg", "noreg", "stats"):
        assert torch.equal(res[True][0]["flat"][g], res[True][1]["flat"][g])
        assert torch.equal(res[True][0]["flat"][g], res[False][0]["flat"][g])
    # single-process reference
    model = _ToyModel()
    st = {g: (torch.zeros_like(model.params.flat[g]), torch.zeros_like(model.params.flat[g])) for g in ("reg", "noreg")}
    for step in range(3):
        grads = {g: torch.zeros_like(model.params.flat[g]) for g in ("reg", "noreg")}
        for rank in range(world):
            model.params.zero_grad()
            model.loss(10 * step + rank).backward()
            for g in grads:
                grads[g] += model.params.grad[g] / world
        t = step + 1
        lr_t = 1e-3 * math.sqrt(1 - 0.99 ** t) / (1 - 0.9 ** t)
        with torch.no_grad():
            _cpu_adam(model.params.flat["reg"], grads["reg"], *st["reg"], lr_t, 0.9, 0.99, 1e-8, 1.0, 1e-2)
            _cpu_adam(model.params.flat["noreg"], grads["noreg"], *st["noreg"], lr_t, 0.9, 0.99, 1e-8, 1.0, 0.0)
    for g in ("reg", "noreg"):
        np.testing.assert_allclose(res[True][0]["flat"][g].numpy(), model.params.flat[g].detach().numpy(), rtol=1e-5, atol=1e-7)


# --------------------------------------------------------- in-place gradient sink x buckets
def _sink_base():
    from boxsegliver_amd import ops
    return ops._Op


class _SinkDot(_sink_base()):
    """CPU stand-in for a libunetk op: y = <x, w>; its backward writes dL/dw straight into the variable's slot of the flat
    gradient buffer when ops._take grants it (the ops._GradSink protocol of the HIP ops), else returns it to autograd."""

    @staticmethod
    def forward(ctx, w, x):
        from boxsegliver_amd import ops
        ctx.save_for_backward(x)
        ctx.sink = ops.grad_sink(w, ctx)
        return (w * x).sum()

    @staticmethod
    def backward(ctx, dy):
        from boxsegliver_amd import ops
        (x,) = ctx.saved_tensors
        slot = ops._take(ctx.sink)
        dw = dy * x
        if slot is not None:
            slot.copy_(dw)
        return ops._ret(dw, slot), None


class _SinkModel(_ToyModel):
    """T/a/weights is SHARED by two ops of one step; the order of the terms (= the order in which backward reaches the
    variables, = the order in which buckets complete) depends on the rank."""

    def loss(self, seed, rank=0):
        gen = torch.Generator().manual_seed(seed)
        p = self.params
        uses = [("T/a/weights", 0.7), ("T/a/beta", 1.0), ("T/b/weights", 1.0), ("T/a/weights", -1.3), ("T/b/gamma", 1.0),
                ("T/c/weights", 1.0), ("T/c/biases", 1.0)]
        xs = [torch.randn(p[name].shape, generator=gen) for name, _ in uses]
        terms = [(_SinkDot.apply(p[name], x) * c + 0.3) ** 2 for (name, c), x in zip(uses, xs)]
        order = list(range(len(terms)))
        order = order[rank % len(order):] + order[:rank % len(order)]          # rank-dependent graph order
        total = 0.0
        for i in order:
            total = total + terms[i]
        return total


def _sink_worker(rank, world, port, out_dir, overlap):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from boxsegliver_amd import ops
    from boxsegliver_amd.core.solver import Solver
    from boxsegliver_amd.utils.distribution_utils import DistributionStrategy
    ops.adam_step = _cpu_adam
    model = _SinkModel()
    solver = Solver(_solver_args())
    solver.strategy = DistributionStrategy("mirrored", world, rank)
    solver.overlap_allreduce = overlap
    solver.bucket_bytes = 64                       # nearly one bucket per variable: every arrival order matters
    written, granted, real_take = [], [], ops._take

    def counting_take(slot):
        out = real_take(slot)
        granted.append(out is not None)
        return out
    ops._take = counting_take
    for step in range(3):
        del granted[:]
        solver(model.loss(10 * step + rank, rank), model)
        written.append(sum(granted))
    torch.save({"flat": model.params.flat, "written": written,
                "diag": {k: v for k, v in solver._buckets.last.items() if k != "events"} if overlap else None},
               os.path.join(out_dir, "r{}.pt".format(rank)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_shared_variable_and_uneven_arrival_order_under_buckets_gloo(world):
    """(1) A variable used by two ops must not 'arrive' at its bucket on the first in-place write: its slot is not taken,
    autograd sums both contributions and the post-accumulate hook fires after the sum.  (2) Ranks reach their variables in
    different orders: buckets are still all-reduced in index order on every rank (no mismatched collectives).  Overlapped
    == plain all-reduce after backward == Adam on the mean gradient of a single process."""
    res = {}
    for overlap in (True, False):
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(_sink_worker, args=(world, _free_port(), d, overlap), nprocs=world, join=True)
            res[overlap] = [torch.load(os.path.join(d, "r{}.pt".format(i))) for i in range(world)]
    for g in ("reg", "noreg"):
        for r in range(1, world):
            assert torch.equal(res[True][0]["flat"][g], res[True][r]["flat"][g])
        if world == 2:
            assert torch.equal(res[True][0]["flat"][g], res[False][0]["flat"][g])
        else:       # four addends: a ring all-reduce sums a bucket's elements in another rank order than the whole buffer's
            np.testing.assert_allclose(res[True][0]["flat"][g].numpy(), res[False][0]["flat"][g].numpy(), rtol=1e-5, atol=1e-8)
    # single-use variables were written in place (5 of them), the shared one was not
    assert res[True][0]["written"] == [5, 5, 5]
    assert res[True][0]["diag"]["buckets"] >= 4          # one bucket per filter variable; gamma / beta ride in the last (round 5)
    model = _SinkModel()
    st = {g: (torch.zeros_like(model.params.flat[g]), torch.zeros_like(model.params.flat[g])) for g in ("reg", "noreg")}
    from boxsegliver_amd import ops
    for step in range(3):
        grads = {g: torch.zeros_like(model.params.flat[g]) for g in ("reg", "noreg")}
        for rank in range(world):
            model.params.zero_grad()
            ops._GradSink.enabled = False                       # reference: plain autograd accumulation
            try:
                model.loss(10 * step + rank, rank).backward()
            finally:
                ops._GradSink.enabled = True
            for g in grads:
                grads[g] += model.params.grad[g] / world
        t = step + 1
        lr_t = 1e-3 * math.sqrt(1 - 0.99 ** t) / (1 - 0.9 ** t)
        with torch.no_grad():
            _cpu_adam(model.params.flat["reg"], grads["reg"], *st["reg"], lr_t, 0.9, 0.99, 1e-8, 1.0, 1e-2)
            _cpu_adam(model.params.flat["noreg"], grads["noreg"], *st["noreg"], lr_t, 0.9, 0.99, 1e-8, 1.0, 0.0)
    for g in ("reg", "noreg"):
        np.testing.assert_allclose(res[True][0]["flat"][g].numpy(), model.params.flat[g].detach().numpy(), rtol=1e-5, atol=1e-7)


def test_grad_sink_scopes_and_use_counts():
    """ops._GradSink bookkeeping: a slot is granted only for a single recorded use; a no_grad forward is not a use; one
    store's zero_grad does not re-open another store's slots."""
    from boxsegliver_amd import ops
    a, b = _ToyModel(), _ToyModel()
    a.params.zero_grad(); b.params.zero_grad()
    w = a.params["T/a/weights"]
    x = torch.ones_like(w)
    ops._GradSink.uses.clear()
    with torch.no_grad():
        _SinkDot.apply(w, x)                                    # evaluation forward: not a use
    assert ops._GradSink.uses.get(w.grad.data_ptr(), 0) == 0
    _SinkDot.apply(w, x).backward()
    assert w.grad.data_ptr() in ops._GradSink.written and torch.equal(w.grad, x)
    _SinkDot.apply(w, 2 * x).backward()                         # a second step without zero_grad must ACCUMULATE
    assert torch.equal(w.grad, 3 * x)
    wb = b.params["T/b/weights"]
    _SinkDot.apply(wb, torch.ones_like(wb)).backward()
    a.params.zero_grad()                                        # re-opens a's slots only
    assert w.grad.data_ptr() not in ops._GradSink.written and wb.grad.data_ptr() in ops._GradSink.written
    (_SinkDot.apply(w, x) + _SinkDot.apply(w, 4 * x)).backward()   # shared inside one step: autograd sums
    assert torch.equal(w.grad, 5 * x) and w.grad.data_ptr() not in ops._GradSink.written


# ----------------------------------------------------------------------------------------- GPU
YML = dict(init_channels=64, num_down_samples=2, ret_prob=False, ret_pred=True, build_metrics=True)


def _unet_args(bs, num_gpus):
    return argparse.Namespace(
        classes=["Liver", "Tumor"], batch_size=bs, num_gpus=num_gpus, im_height=32, im_width=32, im_channel=3,
        normalizer="batch_norm", without_norm=False, weight_init="xavier", weight_decay_rate=1e-5, bias_decay=False,
        loss_type="xentropy", loss_weight_type="numerical", loss_numeric_w=[0.2, 0.4, 4.4], loss_proportion_decay=1000,
        metrics_train=["Dice"], img_grad=False, tag="dp", seed=4321, **vars(_solver_args()))


def _shard(rank):
    from boxsegliver_amd.data.synthetic import make_batch
    images, labels, _ = make_batch(2, 32, 32, 3, 3, 900 + rank)
    return {"images": torch.from_numpy(images).cuda(), "labels": torch.from_numpy(labels).cuda()}


def _gpu_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from boxsegliver_amd.NetworksV2.UNet import UNet
    from boxsegliver_amd.core.solver import Solver
    from boxsegliver_amd.utils.distribution_utils import DistributionStrategy
    args = _unet_args(4, world)                      # global batch 4 -> 2 per replica
    model = UNet(args)
    assert model.bs == 2
    inputs = _shard(rank)
    model(inputs, "eval", **YML)
    strategy = DistributionStrategy("mirrored", world, rank)
    if rank != 0:
        model.params.flat["reg"].mul_(0.5)          # prove the broadcast makes replicas identical
    strategy.broadcast_(list(model.params.flat.values()))
    solver = Solver(args)
    solver.strategy = strategy
    losses = []
    for _ in range(2):
        loss = model(inputs, "train", **YML)
        losses.append(strategy.reduce_mean(loss.detach()).item())
        solver(loss, model)
    torch.cuda.synchronize()
    torch.save({"flat": {k: v.cpu() for k, v in model.params.flat.items()}, "losses": losses},
               os.path.join(out_dir, "r{}.pt".format(rank)))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_dp_two_replicas_equal_per_chunk_bn_emulation():
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_gpu_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        r = [torch.load(os.path.join(d, "r{}.pt".format(i))) for i in range(world)]
    for g in ("reg", "noreg"):
        assert torch.equal(r[0]["flat"][g], r[1]["flat"][g])            # replicas stay in lock-step
    assert not torch.equal(r[0]["flat"]["stats"], r[1]["flat"]["stats"])  # BN moving stats are replica-local
    assert r[0]["losses"] == r[1]["losses"]

    # single-process emulation: same initial variables (seed), BN applied per half-batch, mean gradient
    from boxsegliver_amd.NetworksV2.UNet import UNet
    from boxsegliver_amd.core.solver import Solver
    args = _unet_args(2, 1)
    model = UNet(args)
    shards = [_shard(0), _shard(1)]
    model(shards[0], "eval", **YML)
    solver = Solver(args)
    emu_losses = []
    for _ in range(2):
        grads, losses = [], []
        stats0 = None
        for i, sh in enumerate(shards):
            before = model.params.flat["stats"].clone()
            model.params.zero_grad()
            loss = model(sh, "train", **YML)
            loss.backward()
            grads.append({k: v.clone() for k, v in model.params.grad.items()})
            losses.append(loss.item())
            if i == 0:
                stats0 = model.params.flat["stats"].clone()     # rank 0's replica-local moving stats
            model.params.flat["stats"].copy_(before)
        model.params.flat["stats"].copy_(stats0)
        for k in model.params.grad:
            model.params.grad[k].copy_((grads[0][k] + grads[1][k]) * 0.5)
        lr = solver._get_model_learning_rate()
        solver.apply_gradients(model.params, args.weight_decay_rate, lr)
        emu_losses.append(sum(losses) / 2)
    torch.cuda.synchronize()
    np.testing.assert_allclose(r[0]["losses"], emu_losses, rtol=1e-6)
    for g in ("reg", "noreg", "stats"):
        np.testing.assert_allclose(r[0]["flat"][g].numpy(), model.params.flat[g].cpu().numpy(), rtol=2e-5, atol=2e-7)


def _rccl_worker(rank, world, port, out_dir):
    """RCCL itself (backend "nccl") with the one GPU of the test box: a world of 1 cannot test the sum, but it
    runs the exact production code path -- per-bucket async all-reduce launched from the autograd thread's hooks
    on RCCL's stream, finish() joining the compute stream -- which gloo does not."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    from boxsegliver_amd.NetworksV2.UNet import UNet
    from boxsegliver_amd.utils.distribution_utils import DistributionStrategy, GradBuckets
    args = _unet_args(2, 1)
    model = UNet(args)
    inputs = _shard(0)
    model(inputs, "eval", **YML)
    strategy = DistributionStrategy("mirrored", world, rank)
    strategy.broadcast_(list(model.params.flat.values()))
    model.params.zero_grad()
    model(inputs, "train", **YML).backward()
    plain = {k: v.clone() for k, v in model.params.grad.items()}
    buckets = GradBuckets(model.params, strategy, bucket_bytes=1 << 20)
    for _ in range(2):
        model.params.zero_grad()
        buckets.arm()
        model(inputs, "train", **YML).backward()
        fired_in_backward = sum(buckets._fired)
        buckets.finish()
    torch.cuda.synchronize()
    same = all(torch.equal(plain[k], model.params.grad[k]) for k in plain)
    mean = strategy.reduce_mean(torch.tensor(3.0, device="cuda")).item()
    torch.save({"same": same, "n": len(buckets.buckets), "fired": fired_in_backward, "mean": mean},
               os.path.join(out_dir, "r0.pt"))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_bucketed_allreduce_from_backward_hooks_world1():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_rccl_worker, args=(1, _free_port(), d), nprocs=1, join=True)
        r = torch.load(os.path.join(d, "r0.pt"))
    assert r["same"] and r["n"] >= 3 and r["fired"] == r["n"] and r["mean"] == 3.0


def _rccl_worker_n(rank, world, port, out_dir):
    """RCCL with N > 1 ranks, one GPU each: the bucketed all-reduce launched from backward hooks must leave in every
    rank's gradient buffers the SUM over ranks of the plain per-rank gradients (different shards per rank), and a whole
    Solver step must keep the replicas bit-identical."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    from boxsegliver_amd.NetworksV2.UNet import UNet
    from boxsegliver_amd.core.solver import Solver
    from boxsegliver_amd.utils.distribution_utils import DistributionStrategy, GradBuckets
    args = _unet_args(2 * world, world)
    model = UNet(args)
    inputs = _shard(rank)
    model(inputs, "eval", **YML)
    strategy = DistributionStrategy("mirrored", world, rank)
    if rank != 0:
        model.params.flat["reg"].mul_(0.5)
    strategy.broadcast_(list(model.params.flat.values()))
    model.params.zero_grad()
    model(inputs, "train", **YML).backward()
    want = {k: v.clone() for k, v in model.params.grad.items()}
    for v in want.values():
        dist.all_reduce(v, op=dist.ReduceOp.SUM)                      # plain all-reduce after backward
    buckets = GradBuckets(model.params, strategy, bucket_bytes=1 << 20)
    fired = 0
    for _ in range(2):
        model.params.zero_grad()
        buckets.arm()
        model(inputs, "train", **YML).backward()
        fired = sum(buckets._fired)
        buckets.finish()
    torch.cuda.synchronize()
    same = all(torch.equal(want[k], model.params.grad[k]) for k in want)
    buckets.remove()
    solver = Solver(args)
    solver.strategy = strategy
    for _ in range(2):
        solver(model(inputs, "train", **YML), model)
    torch.cuda.synchronize()
    torch.save({"same": same, "n": len(buckets.buckets), "fired": fired,
                "flat": {k: v.cpu() for k, v in model.params.flat.items()}},
               os.path.join(out_dir, "r{}.pt".format(rank)))
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank; this box has fewer than 2")
def test_rccl_bucketed_allreduce_two_ranks():
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_rccl_worker_n, args=(world, _free_port(), d), nprocs=world, join=True)
        r = [torch.load(os.path.join(d, "r{}.pt".format(i))) for i in range(world)]
    for x in r:
        assert x["same"] and x["n"] >= 3 and x["fired"] == x["n"]
    for g in ("reg", "noreg"):
        assert torch.equal(r[0]["flat"][g], r[1]["flat"][g])


# ------------------------------------------------------------ control hooks under data parallelism
def _plateau_worker(rank, world, port, out_dir):
    import json
    import types
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from boxsegliver_amd.core import hooks
    from boxsegliver_amd.core.estimator import _RunContext
    from boxsegliver_amd.core.solver import Solver
    from boxsegliver_amd.utils.distribution_utils import DistributionStrategy
    a = _solver_args()
    a.learning_policy, a.lr_decay_rate, a.lr_end = "plateau", 0.1, 1e-6
    solver = Solver(a)
    save_dir = os.path.join(out_dir, "model")            # ONE directory shared by the ranks, as in a real job
    est = types.SimpleNamespace(params={"solver": solver}, _train_distribution=DistributionStrategy("mirrored", world, rank),
                                model_dir=save_dir)
    hook = hooks.ReduceLROnPlateauHook(save_dir, lr_patience=1, tr_patience=3, min_delta=0.01, every_n_steps=1,
                                       moving_average=0.0)
    ctx = _RunContext(types.SimpleNamespace(estimator=est))
    lrs, stopped_at = [], None
    gen = torch.Generator().manual_seed(17 + rank)
    for i in range(40):
        solver.global_step = i + 3
        lr = solver._get_model_learning_rate()
        lrs.append(lr)
        # rank-local losses that disagree about "improved by min_delta": rank 0 keeps improving, rank 1 stagnates
        loss = (1.0 / (i + 1) if rank == 0 else 0.5) + 0.01 * float(torch.rand(1, generator=gen))
        hook.after_run(ctx, types.SimpleNamespace(loss=torch.tensor(loss), train_op=lr,
                                                  model=types.SimpleNamespace(metrics_dict={})))
        if ctx.stop_requested:
            stopped_at = i
            break
    with open(os.path.join(out_dir, "r{}.json".format(rank)), "w") as f:
        json.dump({"lrs": lrs, "stopped_at": stopped_at, "state": hook._state()}, f)
    dist.barrier()
    dist.destroy_process_group()


def test_plateau_hook_is_collective_under_dp_gloo_world2():
    """Different per-rank losses must yield the SAME learning-rate sequence and the SAME stop step on every rank (decided
    on rank 0 from the replica-mean loss and broadcast); lr_schedule is written once, by rank 0."""
    import json
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_plateau_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        r = [json.load(open(os.path.join(d, "r{}.json".format(i)))) for i in range(world)]
        sched = json.load(open(os.path.join(d, "model", "lr_schedule")))
    assert r[0]["lrs"] == r[1]["lrs"] and r[0]["stopped_at"] == r[1]["stopped_at"]
    assert r[0]["state"] == r[1]["state"]
    assert min(r[0]["lrs"]) < 1e-3                     # the mean loss did plateau: the rate was decayed at least once
    assert r[0]["stopped_at"] is not None              # ... and training stopped, on both ranks at the same step
    assert sched["best"] == pytest.approx(r[0]["state"]["best"])


def test_bucket_layout_three_buckets_with_a_small_tail_and_norm_parameters_in_the_last():
    """Round 5: one gradient allocation [noreg | reg]; the filter gradients are cut from the end into buckets of >= bucket_bytes, a
    last bucket is cut where no more than tail_bytes of filters remain, and every gamma / beta rides in that last bucket -- one
    contiguous range starting at the beginning of the allocation (utils/distribution_utils.GradBuckets)."""
    from boxsegliver_amd.NetworksV2.base import ParamStore
    from boxsegliver_amd.utils.distribution_utils import DistributionStrategy, GradBuckets
    specs = []
    for i, n in enumerate([100, 200, 3000, 5000, 7000, 4000]):            # forward order: small first layers, big deep ones
        specs += [("L{}/weights".format(i), (n,), "conv_w"), ("L{}/gamma".format(i), (8,), "gamma"), ("L{}/beta".format(i), (8,), "beta")]
    store = ParamStore(specs, torch.device("cpu"))
    n_noreg = store.grad["noreg"].numel()
    assert store.grad["noreg"].data_ptr() == store.gbuf.data_ptr()
    assert store.grad["reg"].data_ptr() == store.gbuf.data_ptr() + 4 * n_noreg
    b = GradBuckets(store, DistributionStrategy("one_device", 1, 0), bucket_bytes=8000 * 4, tail_bytes=400 * 4)
    try:
        spans = [(lo, hi) for lo, hi, _ in b.buckets]
        # contiguous cover of the whole allocation, launch order = from the end
        assert spans[-1][0] == 0 and spans[0][1] == store.gbuf.numel()
        assert all(spans[i][0] == spans[i + 1][1] for i in range(len(spans) - 1))
        # the tail: at most tail_bytes of filters + every norm parameter
        assert spans[-1][1] - n_noreg <= 400 and spans[-1][1] - n_noreg == 300        # L0 + L1
        assert all(b._bucket_of["L{}/{}".format(i, k)] == len(spans) - 1 for i in range(6) for k in ("gamma", "beta"))
        assert b._bucket_of["L5/weights"] == 0 and b._bucket_of["L1/weights"] == len(spans) - 1
        assert len(spans) == 3 and sum(n for _, _, n in b.buckets) == len(specs)
    finally:
        b.remove()

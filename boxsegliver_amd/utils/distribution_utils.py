"""Data-parallel helpers -- mirror of the reference's utils/distribution_utils.py.

The reference replicates the model in ONE process with tf.contrib.distribute.MirroredStrategy and an
NCCL all-reduce (distribution_utils.py:27-104).  The MI355X design is one process per GPU with
torch.distributed (backend "nccl" == RCCL over xGMI); gradients of the two flat parameter buffers are
summed with ONE all-reduce each and scaled by 1/world inside the optimiser kernel, which is exactly
TF's "per-replica loss x 1/N, gradients summed" (core/estimator.py:570-578).
Batch-norm statistics stay replica-local, as in the reference (no SyncBN).
"""
import os

import torch
import torch.distributed as dist

from .. import ops


def per_device_batch_size(batch_size, num_gpus):
    """distribution_utils.py:107-134 (same error text)."""
    if num_gpus <= 1:
        return batch_size
    remainder = batch_size % num_gpus
    if remainder:
        err = ('When running with multiple GPUs, batch size '
               'must be a multiple of the number of available GPUs. Found {} '
               'GPUs with a batch size of {}; try --batch_size={} instead.'
               ).format(num_gpus, batch_size, batch_size - remainder)
        raise ValueError(err)
    return int(batch_size / num_gpus)


class DistributionStrategy(object):
    """What get_distribution_strategy returns: a handle on the process group."""

    def __init__(self, name, world_size, rank):
        self.name = name
        self.num_replicas_in_sync = world_size
        self.rank = rank

    def all_reduce_sum_(self, tensors):
        if self.num_replicas_in_sync > 1:
            works = [dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True) for t in tensors]
            for w in works:
                w.wait()
        return tensors

    def reduce_mean(self, value):
        """strategy.reduce(MEAN, loss) -- core/estimator.py:576,585."""
        if self.num_replicas_in_sync > 1:
            v = value.detach().clone().reshape(1)
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
            return (v / self.num_replicas_in_sync).reshape(())
        return value

    def broadcast_object(self, obj, src=0):
        """A small picklable host value from `src` to every rank (control decisions: learning-rate / stop flags)."""
        if self.num_replicas_in_sync > 1:
            box = [obj]
            dist.broadcast_object_list(box, src=src, device=torch.device("cuda", torch.cuda.current_device())
                                       if dist.get_backend() == "nccl" else None)
            return box[0]
        return obj

    def broadcast_(self, tensors, src=0):
        if self.num_replicas_in_sync > 1:
            for t in tensors:
                dist.broadcast(t, src=src)
            ops.bump_param_gen()              # the variables changed behind torch's version counters' back


_PROBE = os.environ.get("UNETK_DP_PROBE", "")      # measurement switches of GradBuckets (never set in production)


class GradBuckets(object):
    """Gradient all-reduce overlapped with backward.

    Each flat gradient buffer (ParamStore.grad) is cut into contiguous buckets of about `bucket_bytes`, walking
    the variables from the END of the buffer (= forward order reversed: backward finishes the logits / decoder
    gradients first).  A post-accumulate hook on every variable counts arrivals; the moment a bucket's last
    gradient has landed its slice is all-reduced asynchronously -- on RCCL's own stream, ordered after the
    kernels already queued on the compute stream -- while backward keeps running on the encoder.  finish()
    launches whatever did not fire (variables that received no gradient) and makes the compute stream wait.
    The reference packs into num_packs=2 tensors (distribution_utils.py:94-95) and leaves the overlap to TF."""

    def __init__(self, store, strategy, bucket_bytes=64 << 20, tail_bytes=4 << 20):
        self.store, self.strategy = store, strategy
        self.buckets = []                     # (lo, hi, n_vars) in elements of store.gbuf, in LAUNCH order = expected completion order
        self._bucket_of = {}
        self._hooks = []
        # ParamStore keeps both gradient buffers in ONE allocation, [noreg | reg].  One walk over the variables in reverse forward
        # order (= the order their gradients arrive in) cuts the filter gradients ("reg") from the end into buckets of about
        # `bucket_bytes`; the norm parameters (gamma / beta: 47 KB in all for the U-Net) all ride in the LAST bucket -- the first
        # layers' filters, whose slice starts where the noreg buffer ends, so the bucket is one contiguous range.  Round 4 gave
        # the norm parameters met between two cuts a collective of their own: 8 collectives per step, and on a 13 ms step each
        # costs ~30 us on the GPU whatever its size (the stream hand-over: measured in a world of one, where no byte moves:
        # profiles/r05_dp_rehearsal_probe.txt) -- now 3 for the U-Net: 86 MB (the decoder and the deep levels: complete when 55 % of
        # the variables have arrived), 35 MB, and a last bucket that is cut where no more than `tail_bytes` of filters remain, so
        # that what cannot overlap backward (it completes with the very last gradient) stays small: 2.2 MB + gamma / beta.
        gbuf = store.gbuf
        n_noreg = store.grad["noreg"].numel()
        assert store.grad["noreg"].data_ptr() == gbuf.data_ptr() and store.grad["reg"].data_ptr() == gbuf.data_ptr() + 4 * n_noreg
        names = list(reversed(store.trainable_names()))
        hi = store.grad["reg"].numel()
        cur, noreg = [], []

        def close(final):
            nonlocal hi, cur
            members = cur + (noreg if final else [])
            if not members:
                return
            lo = 0 if final else store.where[cur[-1]][1]
            assert all(store.where[a][1] > store.where[b][1] for a, b in zip(cur, cur[1:])), "offsets follow the specs"
            for m in members:
                self._bucket_of[m] = len(self.buckets)
            self.buckets.append((0 if final else n_noreg + lo, n_noreg + hi, len(members)))
            hi, cur = lo, []

        for k, name in enumerate(names):
            grp, off = store.where[name][0], store.where[name][1]
            (cur if grp == "reg" else noreg).append(name)
            final = k == len(names) - 1
            if final or (grp == "reg" and ((hi - off) * 4 >= bucket_bytes or (off * 4 <= tail_bytes < hi * 4))):
                close(final)
        self._sink_keys = []
        self.history, self.last = [], None
        for name in store.trainable_names():
            t = store.tensors[name]
            hook = self._make_hook(name, self._bucket_of[name])
            self._hooks.append(t.register_post_accumulate_grad_hook(hook))
            # gradients the backward kernels write straight into the flat buffer (ops._GradSink) never pass through
            # autograd's accumulation: the op calls the same arrival hook itself
            grp, off, n, shp, _ = store.where[name]
            key = store.grad[grp].data_ptr() + off * 4
            ops._GradSink.hooks[key] = (lambda h=hook: h(None))
            self._sink_keys.append(key)
        self.arm()

    def _make_hook(self, name, b):
        def hook(_param):
            # a variable arrives ONCE per step: autograd also runs the post-accumulate hook of a variable whose
            # gradient the kernels wrote in place (it sees an undefined gradient), after the op's own call
            if _PROBE == "late":               # measurement only: nothing is launched from backward, finish() sends every bucket
                return
            if self._armed and name not in self._arrived:
                self._arrived.add(name)
                self._pending[b] -= 1
                if self._pending[b] == 0:
                    self._ready[b] = True
                    self._launch_ready()
        return hook

    def _launch_ready(self):
        # Collectives of one process group must be issued in the SAME order on every rank.  Arrival order is a property
        # of each rank's autograd walk (and a variable may receive no gradient on one rank only), so buckets are launched
        # strictly by index: a complete bucket waits for the lower-numbered ones.
        while self._next < len(self.buckets) and self._ready[self._next]:
            self._launch(self._next)
            self._next += 1

    def _launch(self, b):
        lo, hi, _ = self.buckets[b]
        self._fired[b] = True
        self._launch_pos[b] = len(self._arrived)      # how many variables had arrived when this bucket went out
        ops.side_join()                       # filter gradients queued on the side stream (ops._Side) must be in the bucket
        if _PROBE == "nolaunch":              # measurement only (bench.py --dp-rehearsal): the hooks' cost without the collective
            return
        self._works.append(dist.all_reduce(self.store.gbuf[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def arm(self):
        """Call before backward of every step."""
        self._pending = [b[2] for b in self.buckets]
        self._fired = [False] * len(self.buckets)
        self._launch_pos = [None] * len(self.buckets)
        self._ready = [False] * len(self.buckets)
        self._next = 0
        self._works = []
        self._arrived = set()
        self._armed = True

    def finish(self):
        """Call after backward: every bucket reduced and visible to the compute stream.  Leaves the step's diagnostics in
        `last` (buckets, how many were launched from inside backward, and -- on a GPU -- HIP events bracketing the wait on
        the compute stream: the all-reduce time backward did not hide; read with exposed_ms() after a synchronize)."""
        self._armed = False
        n_vars = len(self._bucket_of)
        # launched while gradients were still outstanding, i.e. with backward work left to hide the transfer behind (a bucket
        # launched from the very last arrival's hook is "in backward" by the clock but overlaps nothing)
        in_backward = sum(1 for b in range(len(self.buckets)) if self._fired[b] and self._launch_pos[b] < n_vars)
        launch_pos = list(self._launch_pos)
        for b in range(len(self.buckets)):
            self._ready[b] = True
        self._launch_ready()
        ev = None
        if torch.cuda.is_available() and self.store.grad["reg"].is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w in self._works:
            w.wait()
        if ev is not None:
            ev[1].record()
        self._works = []
        self.last = {"buckets": len(self.buckets), "fired_in_backward": in_backward, "events": ev,
                     # per bucket: the fraction of the variables that had arrived when it was launched (None = by finish())
                     "launch_progress": [None if p is None else round(p / float(max(n_vars, 1)), 4) for p in launch_pos],
                     "bucket_bytes": [(hi - lo) * 4 for lo, hi, _ in self.buckets]}
        self.history.append(self.last)
        if len(self.history) > 64:
            del self.history[0]

    def exposed_ms(self):
        """Mean time (ms) the compute stream spent waiting in finish() over the recorded steps (None without events).
        Synchronises the events' stream."""
        ts = []
        for h in self.history:
            if h["events"] is not None:
                h["events"][1].synchronize()
                ts.append(h["events"][0].elapsed_time(h["events"][1]))
        return sum(ts) / len(ts) if ts else None

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for k in self._sink_keys:
            ops._GradSink.hooks.pop(k, None)
        self._sink_keys = []


def init_process_group_from_env(backend=None):
    """One process per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from torchrun."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend)
    return world, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def get_distribution_strategy(distribution_strategy="default", num_gpus=0, num_workers=1,
                              all_reduce_alg=None, session_config=None):
    """distribution_utils.py:27-104.  `off` / num_gpus < 2 -> None; `mirrored`/`default` -> data
    parallel over the initialised process group; `parameter_server` is accepted by the reference
    flag parser but unused by any script -> NotImplementedError; multi-worker -> NotImplementedError
    (:68-69)."""
    if num_gpus < 0:
        raise ValueError("`num_gpus` can not be negative.")
    distribution_strategy = distribution_strategy.lower()
    if distribution_strategy == "off":
        if num_gpus > 1 or num_workers > 1:
            raise ValueError("When {} GPUs and  {} workers are specified, distribution_strategy flag "
                             "cannot be set to 'off'.".format(num_gpus, num_workers))
        return None
    if num_workers > 1:
        raise NotImplementedError("multi-worker training is not supported (reference: distribution_utils.py:68-69)")
    if distribution_strategy == "one_device" or num_gpus < 2:
        if num_gpus > 1:
            raise ValueError("`OneDeviceStrategy` can not be used for more than one device.")
        return DistributionStrategy("one_device", 1, 0)
    if distribution_strategy in ("mirrored", "default"):
        world, rank, _ = init_process_group_from_env()
        if world != num_gpus:
            raise ValueError("--num_gpus {} but WORLD_SIZE is {}: launch one process per GPU "
                             "(python -m torch.distributed.run --nproc-per-node {})".format(num_gpus, world, num_gpus))
        return DistributionStrategy("mirrored", world, rank)
    if distribution_strategy == "parameter_server":
        raise NotImplementedError("parameter_server strategy is not used by the reference's scripts")
    raise ValueError("Unrecognized Distribution Strategy: %r" % distribution_strategy)

"""Training flags, learning-rate policies and the optimiser -- mirror of the reference's core/solver.py.

The optimiser state lives in flat fp32 device buffers and ONE HIP kernel launch per buffer applies
TF-1.13 Adam / Momentum semantics (csrc/optim.hip): lr_t = lr*sqrt(1-b2^t)/(1-b1^t),
theta -= lr_t*m/(sqrt(v)+eps) with defaults beta1 .9, **beta2 .99**, eps 1e-8 (solver.py:205-207),
plus the slim.l2_regularizer gradient wd*w.  Under data parallelism the two gradient buffers are
all-reduced (sum) over RCCL and scaled by 1/world inside the same kernel.
"""
import logging
import math

import numpy as np
import torch

from .. import ops
from ..config import CustomKeys
from ..utils import distribution_utils

log = logging.getLogger("boxsegliver_amd")


def add_arguments(parser):
    """core/solver.py:23-82 (names / defaults verbatim)."""
    group = parser.add_argument_group(title="Training Arguments")
    group.add_argument("--learning_rate", type=float, default=1e-3,
                       help="Base learning rate for model training (default: %(default)f)")
    group.add_argument("--learning_policy", type=str, default="period_step",
                       choices=["custom_step", "period_step", "poly", "plateau"],
                       help="Learning rate policy for training (default: %(default)s)")
    group.add_argument("--num_of_steps", type=int, default=0, help="Number of steps for training")
    group.add_argument("--num_of_total_steps", type=int, default=1000, help="Number of total steps for training")
    group.add_argument("--lr_decay_boundaries", type=int, nargs="*",
                       help="For \"custom_step\" policy. Use the specified learning rate at the given boundaries.")
    group.add_argument("--lr_custom_values", type=float, nargs="+",
                       help="For \"custom_step\" policy. Make sure len(lr_custom_values) - len(lr_decay_boundaries) = 1")
    group.add_argument("--lr_decay_step", type=int, default=1e5,
                       help="For \"period_step\" policy. Decay the base learning rate at a fixed step (default: %(default)d)")
    group.add_argument("--lr_decay_rate", type=float, default=0.1,
                       help="For \"period_step\" and \"plateau\" policy. Learning rate decay rate (default: %(default)f)")
    group.add_argument("--lr_power", type=float, default=0.9,
                       help="For \"poly\" policy. Polynomial power (default: %(default)f)")
    group.add_argument("--lr_end", type=float, default=1e-6,
                       help="For \"poly\" and \"plateau\" policy. The minimal end learning rate (default: %(default)f)")
    group.add_argument("--optimizer", type=str, default="Adam", choices=["Adam", "Momentum", "AdamW"],
                       help="Optimizer for training (default: %(default)s)")
    group.add_argument("--lr_warm_up", action="store_true",
                       help="Warm up with a low learning rate to stabilize parameters")
    group.add_argument("--slow_start_step", type=int, default=1000,
                       help="Training model with small learning rate for few steps")
    group.add_argument("--slow_start_lr", type=float, default=1e-4, help="Learning rate employed during slow start")
    group.add_argument("--adam_beta1", type=float)
    group.add_argument("--adam_beta2", type=float)
    group.add_argument("--adam_eps", type=float)
    group.add_argument("--mm_mm", type=float)
    group.add_argument("--mm_nesterov", action="store_true")
    group.add_argument("--lr_patience", type=int, default=30, help="Learning rate patience for decay (unit: epoch)")


def get_solver_params(args, warm_up=False, slow_start_step=None, slow_start_learning_rate=None):
    """core/solver.py:85-108"""
    optimizer_params = {}
    if args.adam_beta1:
        optimizer_params["beta1"] = args.adam_beta1
    if args.adam_beta2:
        optimizer_params["beta2"] = args.adam_beta2
    if args.adam_eps:
        optimizer_params["epsilon"] = args.adam_eps
    if args.mm_mm:
        optimizer_params["momentum"] = args.mm_mm
    if args.mm_nesterov:
        optimizer_params["use_nesterov"] = True
    params = {"solver": Solver(args, optimizer_params=optimizer_params or None)}
    if warm_up:
        if slow_start_step is None or slow_start_learning_rate is None:
            raise ValueError("If warm up is True, arguments \"slow_start_step\" and "
                             "\"slow_start_learning_rate\" should be given")
        params["solver_kwargs"] = {"slow_start_step": slow_start_step,
                                   "slow_start_learning_rate": slow_start_learning_rate}
    else:
        params["solver_kwargs"] = {}
    return params


class Solver(object):
    def __init__(self, args, name=None, optimizer_params=None):
        self._args = args
        self.name = name or "Optimizer"
        self.global_step = 0
        self.learning_policy = args.learning_policy
        self.base_learning_rate = args.learning_rate
        self.learning_rate_decay_step = args.lr_decay_step
        self.learning_rate_decay_rate = args.lr_decay_rate
        self.num_of_total_steps = args.num_of_total_steps
        self.learning_power = args.lr_power
        self.end_learning_rate = args.lr_end
        self.learning_rate_decay_boundaries = args.lr_decay_boundaries
        self.learning_rate_custom_values = args.lr_custom_values
        self.optimizer = args.optimizer.lower()
        self.optimizer_params = optimizer_params
        self.plateau_lr = None              # the `learning_rate/value` variable of plateau_decay (:246-254)
        self.collections = {CustomKeys.LEARNING_RATE: None}
        self._state = None
        self.strategy = None                # set by the estimator for data parallelism
        self.overlap_allreduce = True       # bucketed all-reduce launched from backward (distribution_utils.GradBuckets)
        self.bucket_bytes = 64 << 20
        self._buckets = None
        self.dp_rehearsal = False           # run the data-parallel path (buckets, all-reduce) in a world of ONE (bench.py --dp-rehearsal)

    @property
    def args(self):
        return self._args

    # ------------------------------------------------------------------ solver.py:138-202
    def _get_model_learning_rate(self, slow_start_step=0, slow_start_learning_rate=1e-4):
        gs = self.global_step
        if self.learning_policy == "period_step":
            lr = self.base_learning_rate * self.learning_rate_decay_rate ** math.floor(gs / self.learning_rate_decay_step)
        elif self.learning_policy == "custom_step":
            lr = self.learning_rate_custom_values[-1]
            for b, v in zip(self.learning_rate_decay_boundaries, self.learning_rate_custom_values):
                if gs <= b:
                    lr = v
                    break
        elif self.learning_policy == "poly":
            g = min(gs, self.num_of_total_steps)
            lr = (self.base_learning_rate - self.end_learning_rate) * \
                (1 - g / self.num_of_total_steps) ** self.learning_power + self.end_learning_rate
        elif self.learning_policy == "plateau":
            if self.plateau_lr is None:
                self.plateau_lr = self.base_learning_rate
            lr = self.plateau_lr
        else:
            raise ValueError('Not supported learning policy.')
        if slow_start_step > 0 and gs < slow_start_step:
            lr = slow_start_learning_rate
        self.collections[CustomKeys.LEARNING_RATE] = lr
        return lr

    def update_plateau_lr(self):
        """The LR_UPDATE_OPS op of plateau_decay (solver.py:251): lr <- max(lr*factor, min_lr)."""
        if self.plateau_lr is None:
            self.plateau_lr = self.base_learning_rate
        self.plateau_lr = max(self.plateau_lr * self.learning_rate_decay_rate, self.end_learning_rate)
        return self.plateau_lr

    # ------------------------------------------------------------------ solver.py:204-219
    def _optimizer_hparams(self):
        if self.optimizer == "adam":
            return self.optimizer_params or {"beta1": 0.9, "beta2": 0.99}
        if self.optimizer == "momentum":
            return self.optimizer_params or {"momentum": 0.9}
        if self.optimizer == "adamw":
            # solver.py:212-216: contrib_opt.AdamWOptimizer(weight_decay=--weight_decay_rate, beta1 .9, beta2 .99);
            # decoupled decay var -= weight_decay * var on EVERY variable, on top of the L2 regularisers
            return self.optimizer_params or {"weight_decay": self._args.weight_decay_rate, "beta1": 0.9, "beta2": 0.99}
        raise ValueError("Not supported optimizer: " + self.optimizer)

    def _ensure_state(self, store):
        if self._state is None:
            self._state = {}
            for grp in ("reg", "noreg"):
                if self.optimizer in ("adam", "adamw"):
                    self._state[grp] = (torch.zeros_like(store.flat[grp]), torch.zeros_like(store.flat[grp]))
                else:
                    self._state[grp] = (torch.zeros_like(store.flat[grp]),)
        return self._state

    def state_dict(self):
        st = {"global_step": self.global_step, "plateau_lr": self.plateau_lr}
        if self._state is not None:
            st["slots"] = {g: [t.detach().cpu() for t in ts] for g, ts in self._state.items()}
        return st

    def load_state_dict(self, st, store):
        self.global_step = int(st.get("global_step", 0))
        self.plateau_lr = st.get("plateau_lr")
        if "slots" in st:
            state = self._ensure_state(store)
            for g, ts in st["slots"].items():
                for dst, src in zip(state[g], ts):
                    dst.copy_(src)

    @torch.no_grad()
    def load_variable_slots(self, store, getter, global_step=0, plateau_lr=None):
        """Optimiser state from per-variable slots (a TensorFlow checkpoint, core/estimator.restore_variables):
        getter(variable name, slot name) -> ndarray (TF shape) or None, slot names "Adam" / "Adam_1" (m, v) or "Momentum";
        plateau_lr = the `Optimizer/learning_rate/value` variable of plateau_decay (solver.py:246-254) when the file has
        one.  Channel-padded stores map the TF shapes onto their padded layout (store.write_slot).  The bias-correction
        powers are not read: they are functions of global_step here as in TF (beta^t).  Returns the number of slots read."""
        self.global_step = int(global_step)
        if plateau_lr is not None:
            self.plateau_lr = float(plateau_lr)
        names = ("Adam", "Adam_1") if self.optimizer in ("adam", "adamw") else ("Momentum",)
        if not hasattr(store, "where"):
            return 0
        state = self._ensure_state(store)
        loaded, missing = 0, []
        for name in store.trainable_names():
            grp = store.where[name][0]
            for k, slot in enumerate(names):
                v = getter(name, slot)
                if v is not None:
                    store.write_slot(state[grp][k], name, np.ascontiguousarray(v, dtype=np.float32))
                    loaded += 1
                else:
                    missing.append(name + "/" + slot)
        if loaded and missing:
            log.warning("optimiser slots missing in the checkpoint for %d entries (kept at zero), e.g. %s",
                        len(missing), missing[:3])
        return loaded

    def variable_slots(self, store):
        """The way back: {(variable name, slot name): tensor in the TF shape} of the live optimiser state."""
        out = {}
        if self._state is None or not hasattr(store, "where"):
            return out
        names = ("Adam", "Adam_1") if self.optimizer in ("adam", "adamw") else ("Momentum",)
        for name in store.trainable_names():
            grp = store.where[name][0]
            for k, slot in enumerate(names):
                out[(name, slot)] = store.read_slot(self._state[grp][k], name)
        return out

    def apply_gradients(self, store, l2, lr):
        """One optimiser step on the flat buffers; gradients are already in store.grad."""
        world = self.strategy.num_replicas_in_sync if self.strategy is not None else 1
        if world > 1 or (self.dp_rehearsal and self.strategy is not None):
            if self._buckets is not None and self._buckets.store is store:
                self._buckets.finish()              # buckets were all-reduced while backward ran
            else:
                self.strategy.all_reduce_sum_([store.grad["reg"], store.grad["noreg"]])
        gscale = 1.0 / world
        hp = self._optimizer_hparams()
        state = self._ensure_state(store)
        t = self.global_step + 1
        for grp, wd in (("reg", l2 or 0.0), ("noreg", 0.0)):
            if self.optimizer in ("adam", "adamw"):
                b1, b2 = hp.get("beta1", 0.9), hp.get("beta2", 0.999)
                eps = hp.get("epsilon", 1e-8)
                lr_t = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
                m, v = state[grp]
                dwd = float(hp.get("weight_decay", 0.0) or 0.0) if self.optimizer == "adamw" else 0.0
                ops.adam_step(store.flat[grp], store.grad[grp], m, v, lr_t, b1, b2, eps, gscale, wd, dwd)
            else:
                (acc,) = state[grp]
                ops.momentum_step(store.flat[grp], store.grad[grp], acc, lr, hp.get("momentum", 0.9),
                                  hp.get("use_nesterov", False), gscale, wd)
        self.global_step += 1

    def __call__(self, loss, model=None, *args, **kwargs):
        """solver.py:221-243: learning rate -> optimiser -> minimize(loss) (BN moving-stat updates
        already ran inside the forward kernels, the eager counterpart of the UPDATE_OPS dependency).
        Returns the learning rate used (the reference returns the train_op)."""
        if model is None:
            raise ValueError("Solver needs the model whose variables it updates")
        lr_params = {}
        if "slow_start_step" in kwargs:
            lr_params["slow_start_step"] = kwargs.pop("slow_start_step")
        if "slow_start_learning_rate" in kwargs:
            lr_params["slow_start_learning_rate"] = kwargs.pop("slow_start_learning_rate")
        lr = self._get_model_learning_rate(**lr_params)
        store = model.params
        store.zero_grad()
        world = self.strategy.num_replicas_in_sync if self.strategy is not None else 1
        if (world > 1 or (self.dp_rehearsal and self.strategy is not None)) and self.overlap_allreduce and hasattr(store, "tensors"):
            if self._buckets is None or self._buckets.store is not store:
                if self._buckets is not None:
                    self._buckets.remove()
                self._buckets = distribution_utils.GradBuckets(store, self.strategy, self.bucket_bytes)
            self._buckets.arm()
        loss.backward()
        w_reg, _ = model._get_regularizer()
        self.apply_gradients(store, w_reg, lr)
        return lr

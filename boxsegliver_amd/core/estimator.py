"""Train / eval entry points -- mirror of the reference's core/estimator.py `CustomEstimator`.

Kept: constructor `(model_fn, model_dir, config, params, warm_start_from)` (:187), `train(input_fn,
steps, hooks, max_steps, saving_listeners)` (:234-261), `evaluate_online(session, predict_keys, steps,
yield_single_examples)` (:334-357), `predict(input_fn, predict_keys, hooks, checkpoint_path,
latest_filename, yield_single_examples)` (:281-332), properties `model_dir / params / config`;
resume from `model_dir` (:249-253), NaN-loss abort (:676), checkpoint status file `checkpoint`
(:694-719), per-step log line `loss, lr, step, <Class>/<Metric> (x it/s)` (core/hooks.py:536-543).

Replaced: the TF graph/session machinery.  The hot loop (`mon_sess.run([train_op, loss])`, :756-757)
is an eager loop that enqueues HIP kernels on the current stream; data parallelism is one process
per GPU (torch.distributed / RCCL) instead of in-graph MirroredStrategy (:528-619).
"""
import json
import logging
import math
import os
import time
from pathlib import Path

import numpy as np
import torch

from .. import ops
from ..NetworksV2.base import ModeKeys
from ..utils import tf_checkpoint

log = logging.getLogger("boxsegliver_amd")


class RunConfig(object):
    """The subset of tf.estimator.RunConfig the reference sets (entry/main.py:139-147)."""

    def __init__(self, model_dir=None, save_checkpoints_steps=5000, keep_checkpoint_max=1, log_step_count_steps=500,
                 train_distribute=None, tf_random_seed=None):
        self.model_dir = model_dir
        self.save_checkpoints_steps = save_checkpoints_steps
        self.keep_checkpoint_max = keep_checkpoint_max
        self.log_step_count_steps = log_step_count_steps
        self.train_distribute = train_distribute
        self.tf_random_seed = tf_random_seed


class SessionRunHook(object):
    """Minimal tf.train.SessionRunHook protocol for the hooks of core/hooks.py."""

    def begin(self):
        pass

    def after_create_session(self, session, coord=None):
        pass

    def before_run(self, run_context):
        pass

    def after_run(self, run_context, run_values):
        pass

    def end(self, session):
        pass


class _RunContext(object):
    def __init__(self, session):
        self.session = session
        self.stop_requested = False

    def request_stop(self):
        self.stop_requested = True


class _Session(object):
    """What hooks / evaluate_online receive in place of a tf.Session."""

    def __init__(self, estimator):
        self.estimator = estimator


class NanLossDuringTrainingError(RuntimeError):
    def __str__(self):
        return "NaN loss during training."


def _load_global_step_from_checkpoint_dir(model_dir, status_file="checkpoint"):
    """core/estimator.py:52-59 (a TensorFlow checkpoint directory: the bundle's `global_step` variable)."""
    try:
        with open(os.path.join(model_dir, status_file)) as f:
            return int(json.load(f)["global_step"])
    except (OSError, ValueError, KeyError):
        pass
    try:
        prefix = tf_checkpoint.get_checkpoint_state(model_dir, status_file)
        if prefix and os.path.exists(prefix + ".index"):
            return int(tf_checkpoint.CheckpointReader(prefix).get_tensor("global_step"))
    except (OSError, ValueError, KeyError):
        pass
    return 0


def restore_variables(path, model, solver=None, root_scope=None, strict=True):
    """Load `path` into model.params (and the solver): a checkpoint of this package (torch file) or a TensorFlow V2
    checkpoint prefix written by the reference's tf.train.Saver -- the variable names are the same (UNet.py:203-205), the
    Adam / Momentum slots are "Optimizer/<variable>/Adam", ".../Adam_1" / ".../Momentum" (core/solver.py:232), the step is
    `global_step`.  root_scope: the model scope inside the file when it differs from model.name (core/models.py:151-177)."""
    rename = (lambda n: n) if not root_scope or root_scope == model.name else \
        (lambda n: root_scope + n[len(model.name):] if n.startswith(model.name) else n)
    if os.path.exists(str(path) + ".index"):
        reader = tf_checkpoint.CheckpointReader(path)
        state, missing = {}, []
        for name in model.params.state_dict():
            if reader.has_tensor(rename(name)):
                state[name] = torch.from_numpy(np.ascontiguousarray(reader.get_tensor(rename(name)), dtype=np.float32))
            else:
                missing.append(name)
        if missing and strict:
            raise KeyError("{}: {} variables missing, e.g. {}".format(path, len(missing), missing[:3]))
        model.params.load_state(state, strict=False)
        if solver is not None:
            def slot(name, k):
                key = "Optimizer/{}/{}".format(rename(name), k)
                return reader.get_tensor(key) if reader.has_tensor(key) else None
            step = int(reader.get_tensor("global_step")) if reader.has_tensor("global_step") else 0
            lr_key = "{}/learning_rate/value".format(solver.name)           # plateau_decay's variable (solver.py:246-254)
            plateau = float(reader.get_tensor(lr_key)) if reader.has_tensor(lr_key) else None
            solver.load_variable_slots(model.params, slot, step, plateau_lr=plateau)
        return
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    variables = ckpt["variables"]
    if root_scope and root_scope != model.name:
        variables = {model.name + k[len(root_scope):] if k.startswith(root_scope) else k: v for k, v in variables.items()}
    model.params.load_state(variables, strict=strict)
    if solver is not None and ckpt.get("solver"):
        solver.load_state_dict(ckpt["solver"], model.params)


def save_tf_checkpoint(prefix, model, solver=None, root_scope=None):
    """The way back: write model (and solver) state as a TensorFlow V2 checkpoint with the names tf.train.Saver uses in the
    reference's graph -- variables under their graph names, `global_step`, and for Adam "Optimizer/<variable>/Adam{,_1}" +
    "Optimizer/beta{1,2}_power" (= beta^(t+1) as TF keeps them), for Momentum "Optimizer/<variable>/Momentum" -- so the
    reference's tooling (its evaluators / exporters under TF) can consume a model trained here.  Returns the prefix."""
    ren = (lambda n: n) if not root_scope or root_scope == model.name else \
        (lambda n: root_scope + n[len(model.name):] if n.startswith(model.name) else n)
    out = {ren(k): v.numpy() for k, v in model.params.state_dict().items()}
    if solver is not None:
        t = int(solver.global_step)
        out["global_step"] = np.int64(t)
        adam = solver.optimizer in ("adam", "adamw")
        if adam:
            hp = solver._optimizer_hparams()
            out["Optimizer/beta1_power"] = np.float32(hp.get("beta1", 0.9) ** (t + 1))
            out["Optimizer/beta2_power"] = np.float32(hp.get("beta2", 0.999) ** (t + 1))
        if solver.learning_policy == "plateau":        # the non-trainable `learning_rate/value` variable (solver.py:246-254)
            lr = solver.plateau_lr if solver.plateau_lr is not None else solver.base_learning_rate
            out["{}/learning_rate/value".format(solver.name)] = np.float32(lr)
        for (name, slot), v in solver.variable_slots(model.params).items():
            out["Optimizer/{}/{}".format(ren(name), slot)] = v.numpy()
    return tf_checkpoint.write_checkpoint(prefix, out)


class CustomEstimator(object):
    def __init__(self, model_fn, model_dir=None, config=None, params=None, warm_start_from=None):
        if model_fn is None:
            raise ValueError('model_fn must be provided to Estimator.')
        self._config = config or RunConfig(model_dir=model_dir)
        if model_dir:
            self._config.model_dir = model_dir
        self._model_dir = self._config.model_dir
        self._train_distribution = self._config.train_distribute
        self._model_fn = model_fn
        self._params = params or {}
        self._warm_start_from = warm_start_from
        self.predictions = None
        self._eval_iter_fn = None
        self._nan_flag = None          # int32[2] on the device: (a NaN loss was seen, at step) -- _train_model
        self.double_dataloader_modes = self._params.get("double_dataloader_modes", None)
        if self.double_dataloader_modes and len(self.double_dataloader_modes) != 2:
            raise ValueError("double_dataloader_modes need a list of 2 elements for specifying input_fn modes")

    @property
    def model_dir(self):
        return self._model_dir

    @property
    def config(self):
        return self._config

    @property
    def params(self):
        return self._params

    # ------------------------------------------------------------------ checkpoints
    def _model(self):
        inst = self._params.get("model_instances")
        return inst[0] if inst else None

    def checkpoint_path(self, checkpoint_path=None, latest_filename=None):
        if checkpoint_path:
            return checkpoint_path
        # this package's JSON status file or TensorFlow's text CheckpointState (a model_dir trained by the reference)
        return tf_checkpoint.get_checkpoint_state(self._model_dir, latest_filename)

    def save_checkpoint(self, status_file="checkpoint", tag="model.ckpt"):
        model, solver = self._model(), self._params.get("solver")
        if model is None or model.params is None or not self._model_dir:
            return None
        rank = self._train_distribution.rank if self._train_distribution else 0
        if rank != 0:
            return None
        os.makedirs(self._model_dir, exist_ok=True)
        step = solver.global_step if solver else 0
        fname = "{}-{}.pt".format(tag, step)
        torch.save({"variables": model.params.state_dict(), "solver": solver.state_dict() if solver else None},
                   os.path.join(self._model_dir, fname))
        status = os.path.join(self._model_dir, status_file)
        # the previous entry: this package's JSON status file, or TensorFlow's text CheckpointState after a resume from
        # a model_dir the reference trained (get_checkpoint_state reads both)
        old = tf_checkpoint.get_checkpoint_state(self._model_dir, status_file)
        with open(status, "w") as f:
            json.dump({"model_checkpoint_path": fname, "global_step": step}, f)
        if old and os.path.basename(old) != fname and self._config.keep_checkpoint_max == 1 and os.path.isfile(old) \
                and old.endswith(".pt"):                      # never a TensorFlow bundle (prefix.index / .data-*)
            try:
                os.remove(old)
            except OSError:
                pass
        return fname

    def _restore(self, path, model, solver=None):
        restore_variables(path, model, solver)

    # ------------------------------------------------------------------ model_fn plumbing
    def _call_model_fn(self, features, labels, mode, config=None):
        return self._model_fn(features, labels, mode, self._params, config or self._config)

    def _maybe_restore(self, model, solver):
        if getattr(self, "_restored", False) or model.params is None:
            return
        self._restored = True
        path = self.checkpoint_path()
        if path and tf_checkpoint.checkpoint_exists(path):
            self._restore(path, model, solver)                     # auto-resume (:738-741)
        else:
            if self._warm_start_from:
                self._restore(self._warm_start_from, model, None)  # warm start (:649-652)
            from . import models as models_lib
            init_fn = models_lib.init_model(model, self._params["args"])   # --load_weights (core/models.py:160-185,258)
            if init_fn is not None:
                init_fn(None, None)
        strategy = self._train_distribution
        if strategy is not None and strategy.num_replicas_in_sync > 1:
            strategy.broadcast_(list(model.params.flat.values()))  # identical replicas

    # ------------------------------------------------------------------ train
    def train(self, input_fn, steps=None, hooks=None, max_steps=None, saving_listeners=None):
        if (steps is not None) and (max_steps is not None):
            raise ValueError('Can not provide both steps and max_steps.')
        if steps is not None and steps <= 0:
            raise ValueError('Must specify steps > 0, given: {}'.format(steps))
        if max_steps is not None and max_steps <= 0:
            raise ValueError('Must specify max_steps > 0, given: {}'.format(max_steps))
        if max_steps is not None:
            start_step = _load_global_step_from_checkpoint_dir(self._model_dir) if self._model_dir else 0
            if max_steps <= start_step:
                log.info('Skipping training since max_steps has already saved.')
                return self
        loss = self._train_model(input_fn, list(hooks or []), steps, max_steps)
        log.info('Loss for final step: %s.', loss)
        return self

    def _train_model(self, input_fn, hooks, steps, max_steps):
        args = self._params["args"]
        solver = self._params["solver"]
        solver.strategy = self._train_distribution
        train_iter = iter(input_fn(ModeKeys.TRAIN, self._params))
        if getattr(args, "eval_per_epoch", False):
            self._eval_iter_fn = lambda: iter(input_fn("eval_online", self._params))
        session = _Session(self)
        for h in hooks:
            h.begin()
        for h in hooks:
            h.after_create_session(session)
        ctx = _RunContext(session)
        log_step = max(int(getattr(args, "log_step", 500) or 500), 1)
        save_steps = self._config.save_checkpoints_steps
        done, last_loss = 0, None
        t_last, it_last = time.time(), 0
        first = True
        while not ctx.stop_requested:
            try:
                features, labels = next(train_iter)
            except StopIteration:
                break
            if first:
                # run one forward to create variables, then restore / broadcast before the first update
                first = False
                model_peek = self._params.get("model_instances")
                if not model_peek:
                    self._params["model_instances"] = [self._params["model"](args)]
                model = self._params["model_instances"][0]
                if model.params is None:
                    model(dict({k: v for k, v in features.items() if k != "names"}, labels=labels), ModeKeys.EVAL,
                          **{k: v for k, v in self._params.get("model_kwargs", {}).items()
                             if k not in ("build_metrics", "build_summaries")})
                self._maybe_restore(model, solver)
                if max_steps is not None and solver.global_step >= max_steps:
                    break
            for h in hooks:
                h.before_run(ctx)
            spec = self._call_model_fn(features, labels, ModeKeys.TRAIN)
            self.predictions = spec.predictions
            done += 1
            step = solver.global_step
            # NanTensorHook(loss) (reference core/estimator.py:676) runs EVERY step; here a one-thread kernel sets a sticky
            # device flag and the host reads it where it synchronises anyway (log steps, before each checkpoint)
            if spec.loss.is_cuda:
                if self._nan_flag is None:
                    self._nan_flag = torch.zeros(2, dtype=torch.int32, device=spec.loss.device)
                ops.nan_watch(spec.loss.detach().to(torch.float32).reshape(1), self._nan_flag, step)
            if step % log_step == 0 or done == 1:
                # the logged loss and "<Class>/<Metric>" scalars are replica MEANS (strategy.reduce, :576,:585); log steps
                # are the same on every rank, so the collective is safe; host sync only at log steps
                dp = self._train_distribution if (self._train_distribution is not None and
                                                  self._train_distribution.num_replicas_in_sync > 1) else None
                red = (lambda v: dp.reduce_mean(v.detach().to(torch.float32))) if dp is not None else (lambda v: v)
                loss_val = float(red(spec.loss).detach())
                if math.isnan(loss_val):
                    raise NanLossDuringTrainingError()          # NanTensorHook, estimator.py:676
                self._raise_if_nan_seen(dp)                     # a NaN at any step since the last poll
                last_loss = loss_val
                vals = {"loss": loss_val, "lr": spec.train_op, "step": step}
                for k, v in spec.model.metrics_dict.items():
                    vals[k] = float(red(v) if torch.is_tensor(v) else v)
                now = time.time()
                msg = ", ".join(("%s = %.4g" if k == "step" else "%s = %.3g") % (k, vals[k]) for k in sorted(vals))
                if done > 1:
                    msg += " ({:.3g} it/s)".format((done - it_last) / max(now - t_last, 1e-9))
                t_last, it_last = now, done
                log.info(msg)
            for h in hooks:
                h.after_run(ctx, spec)
            if save_steps and step % save_steps == 0:
                self._raise_if_nan_seen(self._dp_or_none())     # never checkpoint a model a NaN step has gone through
                self.save_checkpoint()
            if steps is not None and done >= steps:
                break
            if max_steps is not None and step >= max_steps:
                break
        self._raise_if_nan_seen(self._dp_or_none())         # before the hooks' final evaluation / best-checkpoint save
        for h in hooks:
            h.end(session)
        if last_loss is None and done:
            last_loss = float(spec.loss)
        self._raise_if_nan_seen(self._dp_or_none())
        self.save_checkpoint()
        return last_loss

    def poll_nan(self):
        """For hooks, on EVERY rank, before work that would consume or persist the variables (an evaluation, a best-checkpoint
        save): raises NanLossDuringTrainingError if any step since the last poll saw a NaN loss (the flag is max-reduced under
        data parallelism, so all ranks stop together).  One small device -> host read."""
        self._raise_if_nan_seen(self._dp_or_none())

    def _dp_or_none(self):
        d = self._train_distribution
        return d if (d is not None and d.num_replicas_in_sync > 1) else None

    def _raise_if_nan_seen(self, dp):
        """Poll the sticky NaN flag (one small device -> host read; every rank calls this at the same steps, so under data
        parallelism the flag is max-reduced first and all ranks stop together)."""
        if self._nan_flag is None:
            return
        flag = self._nan_flag
        if dp is not None:
            flag = flag.clone()
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
        seen, at = (int(v) for v in flag.tolist())
        if seen:
            log.error("NaN loss at step %d", at)
            raise NanLossDuringTrainingError()

    # ------------------------------------------------------------------ eval
    def evaluate_online(self, session=None, predict_keys=None, steps=None, yield_single_examples=False):
        """Evaluate from inside training with the live variables and is_training=False (moving BN
        statistics), as the reference does by re-running the train graph with the eval iterator handle."""
        if self._eval_iter_fn is None:
            raise ValueError("evaluate_online needs --eval_per_epoch (an `eval_online` input_fn mode)")
        counter = 0
        for features, labels in self._eval_iter_fn():
            if steps is not None and counter >= steps:
                break
            spec = self._call_model_fn(features, labels, ModeKeys.EVAL)
            preds = spec.predictions if predict_keys is None else {k: spec.predictions[k] for k in predict_keys}
            counter += 1
            if not yield_single_examples:
                yield preds
            else:
                n = features["images"].shape[0]
                for i in range(n):
                    yield {k: (v[i] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == n else v)
                           for k, v in preds.items()}

    def predict(self, input_fn, predict_keys=None, hooks=None, checkpoint_path=None, latest_filename=None,
                yield_single_examples=True):
        path = self.checkpoint_path(checkpoint_path, latest_filename)
        if checkpoint_path and not tf_checkpoint.checkpoint_exists(path):
            raise FileNotFoundError("Missing checkpoint file {}".format(path))     # a TF prefix exists only as .index/.data-*
        restored = False
        for features, labels in input_fn(ModeKeys.EVAL, self._params):
            if not restored:
                restored = True
                if not self._params.get("model_instances"):
                    self._params["model_instances"] = [self._params["model"](self._params["args"])]
                model = self._params["model_instances"][0]
                if model.params is None:
                    self._call_model_fn(features, labels, ModeKeys.EVAL)
                if path and tf_checkpoint.checkpoint_exists(path):
                    self._restore(path, model, None)
            spec = self._call_model_fn(features, labels, ModeKeys.EVAL)
            preds = spec.predictions
            if isinstance(predict_keys, list):
                keys = predict_keys + list(spec.model.metrics_dict.keys())
                preds = {k: preds[k] for k in keys if k in preds}
            elif predict_keys is not None:
                raise TypeError("predict_keys must be None(for 3d eval) or a list(for 2d eval, "
                                "for example [\"Names\", \"Indices\"])")
            if not yield_single_examples:
                yield preds
            else:
                n = features["images"].shape[0]
                for i in range(n):
                    yield {k: (v[i] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == n else v)
                           for k, v in preds.items()}

    def evaluate(self, evaluator, input_fn, hooks=None, checkpoint_path=None, latest_filename=None, cases=None):
        """core/estimator.py:263-279: delegate to the evaluator, dump eval_results_{2d,3d}.txt."""
        checkpoint_path = self.checkpoint_path(checkpoint_path, latest_filename)
        results = evaluator.run(input_fn, hooks=hooks, checkpoint_path=checkpoint_path, cases=cases)
        suffix = "_3d" if getattr(self._params["args"], "eval_3d", False) else "_2d"
        with (Path(self.model_dir) / "eval_results{}.txt".format(suffix)).open("w") as f:
            json.dump(results, f)
        return results

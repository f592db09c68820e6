"""Training-control hooks -- host-side mirror of the reference's core/hooks.py on the estimator of
this package: `LogLearningRateHook` (:471-518), `ReduceLROnPlateauHook` (:581-723) and `EvaluatorHook` (:79-285).

The reference's hooks talk to a tf.Session (fetch tensors in before_run, read them in after_run); here after_run receives
the step's EstimatorSpec (loss tensor on the device, learning rate, model) and `run_context.session.estimator` is the
CustomEstimator, so a hook can read the solver's global step, decay the plateau learning rate or save a checkpoint.
Host syncs (`float(loss)`) happen only on a hook's trigger steps.  Pure control plane: no kernels here.
"""
import json
import logging
import time
from pathlib import Path

import numpy as np

from ..evaluators import evaluator_base
from .estimator import SessionRunHook

log = logging.getLogger("boxsegliver_amd")


class SecondOrStepTimer(object):
    """basic_session_run_hooks.SecondOrStepTimer: triggers every N steps or every N seconds (exactly one given)."""

    def __init__(self, every_secs=None, every_steps=None):
        if (every_secs is None) == (every_steps is None):
            raise ValueError("Exactly one of every_secs and every_steps should be provided.")
        self._every_secs, self._every_steps = every_secs, every_steps
        self._last_time, self._last_step = None, None

    def should_trigger_for_step(self, step):
        if self._last_step is None:
            return True
        if self._last_step == step:
            return False
        if self._every_secs is not None and time.time() >= self._last_time + self._every_secs:
            return True
        if self._every_steps is not None and step >= self._last_step + self._every_steps:
            return True
        return False

    def update_last_triggered_step(self, step):
        now = time.time()
        elapsed = (None, None) if self._last_time is None else (now - self._last_time, step - self._last_step)
        self._last_time, self._last_step = now, step
        return elapsed

    def last_triggered_step(self):
        return self._last_step


def _global_step(run_context):
    return run_context.session.estimator.params["solver"].global_step


def _strategy(session):
    """The data-parallel strategy of the session's estimator when it spans more than one rank, else None.  The reference
    runs ONE process for all GPUs (MirroredStrategy), so its hooks see replica-reduced values and act once; with one
    process per GPU every rank runs the hooks, and anything that changes the training state must be made collective."""
    st = getattr(getattr(session, "estimator", None), "_train_distribution", None)
    return st if st is not None and st.num_replicas_in_sync > 1 else None


def _poll_nan(session):
    """Estimator.poll_nan on every rank (fake sessions of the host tests have no estimator / no such method)."""
    poll = getattr(getattr(session, "estimator", None), "poll_nan", None)
    if poll is not None:
        poll()


class LogLearningRateHook(SessionRunHook):
    """core/hooks.py:471-518: log (and record) the learning rate every N steps."""

    def __init__(self, prefix, every_n_steps=100, every_n_secs=None, output_dir=None, do_logging=True, summary_writer=None):
        self._timer = SecondOrStepTimer(every_steps=every_n_steps, every_secs=every_n_secs)
        self._summary_tag = "{}/learning rate".format(prefix)
        self.do_logging = do_logging
        self.records = []                       # (step, lr): stands in for the TF summary writer

    def after_run(self, run_context, run_values):
        step = _global_step(run_context)
        if self._timer.should_trigger_for_step(step):
            self._timer.update_last_triggered_step(step)
            self._log_and_record(float(run_values.train_op), step)

    def _log_and_record(self, lr, step):
        self.records.append((step, lr))
        if self.do_logging:
            log.info(self._summary_tag + ": {:.6f}".format(lr))


class ReduceLROnPlateauHook(SessionRunHook):
    """core/hooks.py:581-723: every N steps fold the monitored value (default the loss) into a moving average; when it
    has not improved by min_delta for more than lr_patience checks, run the solver's plateau update
    (lr <- max(lr * decay_rate, lr_end), solver.py:246-254); stop training when it has not improved for tr_patience
    checks and the learning rate is already at its floor.  State survives restarts in <save_dir>/lr_schedule."""

    def __init__(self, save_dir, monitor="total_loss", lr_patience=30, tr_patience=50, mode="min", min_delta=0.0005,
                 cooldown=0, moving_average=0.95, every_n_steps=200, every_n_secs=None):
        self.save_dir = save_dir
        self.monitor = monitor
        self.lr_patience, self.tr_patience = lr_patience, tr_patience
        self.mode, self.min_delta, self.cooldown = mode, min_delta, cooldown
        self.cooldown_counter = 0
        self.lr_wait = self.tr_wait = 0
        self.alpha = moving_average
        self.total_loss_MA = None
        self.lr_threshold = 1e-6
        self._reset()
        self.load_lr_schedule()
        self.inc_tr_patience = self.tr_patience // 2
        self._timer = SecondOrStepTimer(every_steps=every_n_steps, every_secs=every_n_secs)

    def _reset(self):
        if self.mode not in ["min", "max"]:
            raise ValueError("Learning Rate Plateau Reducing mode %s is unknown, fallback to auto mode." % self.mode)
        if self.mode == "min":
            self.monitor_op = lambda a, b: np.less(a, b - self.min_delta)
            self.best = np.inf
        else:
            self.monitor_op = lambda a, b: np.greater(a, b + self.min_delta)
            self.best = -np.inf
        self.cooldown_counter = 0
        self.wait = 0

    def in_cooldown(self):
        return self.cooldown_counter > 0

    def _monitored(self, run_values):
        if self.monitor == "total_loss":
            loss = run_values.loss
            return float(loss.detach()) if hasattr(loss, "detach") else float(loss)
        return float(run_values.model.metrics_dict[self.monitor])

    def after_run(self, run_context, run_values):
        step = _global_step(run_context)
        strategy = _strategy(run_context.session)
        if strategy is not None and self._timer._every_secs is not None:
            raise ValueError("ReduceLROnPlateauHook under data parallelism needs a step-based trigger (every_n_steps): a "
                             "time-based one fires on different steps on different ranks")
        if self._timer.should_trigger_for_step(step) and step > 2:
            self._timer.update_last_triggered_step(step)
            old_lr = float(run_values.train_op)
            current = self._monitored(run_values)
            if strategy is None:
                self.try_update_lr(run_context.session, current)
                stop = self.check_stop(old_lr)
            else:
                # the reference monitors the replica-MEAN loss (core/estimator.py:576) and keeps ONE learning-rate
                # variable (ONLY_FIRST_REPLICA, solver.py:249-250): reduce the value, decide on rank 0, broadcast the
                # decision -- every rank then applies the same lr and stops at the same step
                import torch
                device = run_values.loss.device if hasattr(run_values.loss, "device") else "cpu"
                current = float(strategy.reduce_mean(torch.tensor(current, dtype=torch.float64, device=device)))
                solver = run_context.session.estimator.params["solver"]
                decision = None
                if strategy.rank == 0:
                    self.try_update_lr(run_context.session, current)
                    decision = {"plateau_lr": solver.plateau_lr, "stop": bool(self.check_stop(old_lr)), "state": self._state()}
                decision = strategy.broadcast_object(decision, src=0)
                if strategy.rank != 0:
                    solver.plateau_lr = decision["plateau_lr"]
                    self._set_state(decision["state"])
                stop = decision["stop"]
            if stop:
                run_context.request_stop()

    def _state(self):
        return {"best": float(self.best), "total_loss_MA": self.total_loss_MA, "tr_wait": self.tr_wait,
                "lr_wait": self.lr_wait, "cooldown_counter": self.cooldown_counter}

    def _set_state(self, s):
        self.best, self.total_loss_MA = s["best"], s["total_loss_MA"]
        self.tr_wait, self.lr_wait, self.cooldown_counter = s["tr_wait"], s["lr_wait"], s["cooldown_counter"]

    def load_lr_schedule(self):
        f = Path(self.save_dir) / "lr_schedule"
        if f.exists():
            with f.open() as fh:
                s = json.load(fh)
            self.best, self.total_loss_MA = s["best"], s["total_loss_MA"]
            self.tr_wait, self.lr_wait, self.cooldown_counter = s["tr_wait"], s["lr_wait"], s["cooldown_counter"]

    def save_lr_schedule(self):
        Path(self.save_dir).mkdir(parents=True, exist_ok=True)
        with (Path(self.save_dir) / "lr_schedule").open("w") as fh:
            json.dump({"best": float(self.best), "total_loss_MA": float(self.total_loss_MA), "tr_wait": self.tr_wait,
                       "lr_wait": self.lr_wait, "lr_patience": self.lr_patience, "lr_threshold": float(self.lr_threshold),
                       "tr_patience": self.tr_patience, "cooldown_counter": self.cooldown_counter, "mode": self.mode}, fh)

    def try_update_lr(self, session, current):
        if self.total_loss_MA is None:
            self.total_loss_MA = current
        else:
            self.total_loss_MA = self.alpha * self.total_loss_MA + (1 - self.alpha) * current
        if self.in_cooldown():
            self.cooldown_counter -= 1
            self.lr_wait = 0
        log.info("*** total_loss_MA={:.3g}, last_best={:.3g}, wait {} epochs/tr, {} epochs/lr"
                 .format(self.total_loss_MA, self.best, self.tr_wait, self.lr_wait))
        if self.monitor_op(self.total_loss_MA, self.best):
            self.best = self.total_loss_MA
            self.lr_wait = self.tr_wait = 0
        elif not self.in_cooldown():
            self.lr_wait += 1
            self.tr_wait += 1
            if self.lr_wait > self.lr_patience:
                log.info("*** Decay learning rate. Total loss MA: {:.3g}".format(self.total_loss_MA))
                session.estimator.params["solver"].update_plateau_lr()          # the LR_UPDATE_OPS op
                self.cooldown_counter = self.cooldown
                self.lr_wait = 0
        self.save_lr_schedule()

    def check_stop(self, old_lr):
        if self.tr_wait <= self.tr_patience:
            return False
        elif old_lr > self.lr_threshold:
            self.tr_wait -= self.inc_tr_patience
            return False
        return True


class EvaluatorHook(SessionRunHook):
    """core/hooks.py:79-285: evaluate every N steps (and at the end) with `evaluator.run_with_session(session)`, keep
    the better result by `compare_fn`, and -- with save_best -- save the variables under the status file
    `checkpoint_best` (or `checkpoint_best_<interval end>` with save_interval) next to a `best_result` json."""

    def __init__(self, evaluator, checkpoint_dir=None, compare_fn=None, prefix=None, eval_n_secs=None, eval_n_steps=None,
                 saver=None, checkpoint_basename="best_model.ckpt", save_best=False, save_interval=0):
        if not isinstance(evaluator, evaluator_base.EvaluateBase):
            raise TypeError("`evaluator` must be an EvaluateBase instance")
        self._summary_tag = prefix + "/Eval/{}" if prefix else "Eval/{}"
        self._evaluator = evaluator
        self._compare_fn = compare_fn
        self._checkpoint_dir = checkpoint_dir
        self._timer = SecondOrStepTimer(every_secs=eval_n_secs, every_steps=eval_n_steps)
        self._save_best, self._save_interval = save_best, save_interval
        self._better_result = None
        self._basename = checkpoint_basename
        self._need_save = False
        self._last_interval_step = 0
        self.summaries = []                    # (step, {tag: value}): stands in for the TF summary writer
        if self._save_best:
            if self._save_interval:
                saved = [-1] + [int(x.stem.split("_")[-1]) for x in Path(checkpoint_dir).glob("best_result_*")]
                self._last_interval_step = max(saved)
                best_file = self._best_file("best_result_{}".format(max(saved)))
            else:
                best_file = self._best_file()
            if best_file.exists():
                with best_file.open() as f:
                    self._better_result = json.load(f)
                log.info("Best result records '%s' loaded!", best_file)

    def _best_file(self, name="best_result"):
        return Path(self._checkpoint_dir) / name

    def after_run(self, run_context, run_values):
        step = _global_step(run_context)
        if self._timer.should_trigger_for_step(step):
            self._timer.update_last_triggered_step(step)
            if self._evaluate(run_context.session, step):
                run_context.request_stop()

    def end(self, session):
        last_step = session.estimator.params["solver"].global_step
        if last_step != self._timer.last_triggered_step():
            self._evaluate(session, last_step)

    def _evaluate(self, session, step):
        _poll_nan(session)                  # never evaluate -- or save as "best" -- variables a NaN step has gone through
        results = self._evaluator.run_with_session(session)
        if self._save_interval and (step // self._save_interval != self._last_interval_step // self._save_interval):
            self._better_result = None                                  # new interval
        if not self._better_result or self._compare_fn(results, self._better_result):
            self._better_result = results
            self._need_save = True
        self.summaries.append((step, {self._summary_tag.format(k): v for k, v in results.items()}))
        if not self._save_best or not self._need_save:
            return False
        self._need_save = False
        if self._save_interval:
            end_point = (step // self._save_interval + 1) * self._save_interval
            status, best = "checkpoint_best_{}".format(end_point), "best_result_{}".format(end_point)
            self._last_interval_step = step
        else:
            status, best = "checkpoint_best", "best_result"
        strategy = _strategy(session)
        if strategy is not None and strategy.rank != 0:
            return False                      # every rank evaluates (lock-step), rank 0 alone writes checkpoint / best_result
        log.info("Saving (best) checkpoints for %d into %s (%s).", step - 1, self._checkpoint_dir, status)
        session.estimator.save_checkpoint(status_file=status, tag=self._basename)
        with self._best_file(best).open("w") as f:
            json.dump({k: (int(v) if isinstance(v, (np.integer,)) else float(v)) for k, v in self._better_result.items()}, f)
        return False


class EvaluatorHookV2(SessionRunHook):
    """core/hooks.py:288-468: evaluate every N steps; keep a MOVING AVERAGE of every metric and save the variables (status
    file `checkpoint_best`, `best_result` json = {"ma_results", "ma_best_result"}) whenever the mean of the averaged metrics
    beats the best so far under `compare_fn` (default: larger is better).  As in the reference the first trigger only
    evaluates (the averages start with the second: "update moving average for 1 trigger delay", :386-388), and the
    evaluation at the end of training does not count as a trigger."""

    def __init__(self, evaluator, checkpoint_dir=None, compare_fn=lambda x, y: x > y, prefix=None, eval_n_secs=None,
                 eval_n_steps=None, saver=None, checkpoint_basename="best_model.ckpt", save_best=False, ma_alpha=0.9):
        if not isinstance(evaluator, evaluator_base.EvaluateBase):
            raise TypeError("`evaluator` must be an EvaluateBase instance")
        self._summary_tag = prefix + "/Eval/{}" if prefix else "Eval/{}"
        self._evaluator = evaluator
        self._compare_fn = compare_fn
        self._checkpoint_dir = checkpoint_dir
        self._timer = SecondOrStepTimer(every_secs=eval_n_secs, every_steps=eval_n_steps)
        self._save_best = save_best
        self._basename = checkpoint_basename
        self._ma_results = None
        self._ma_best_result = None
        self.ma_alpha = ma_alpha
        self._trigger_counter = 0
        self._need_save = False
        self.summaries = []                    # (step, {tag: value}): stands in for the TF summary writer
        if self._save_best:
            best_file = self._best_file()
            if best_file.exists():
                with best_file.open() as f:
                    data = json.load(f)
                self._ma_results, self._ma_best_result = data["ma_results"], data["ma_best_result"]
                log.info("Load previous best result records: %s", data)

    def _best_file(self, name="best_result"):
        return Path(self._checkpoint_dir) / name

    def after_run(self, run_context, run_values):
        step = _global_step(run_context)
        if self._timer.should_trigger_for_step(step):
            self._timer.update_last_triggered_step(step)
            self._trigger_counter += 1
            if self._evaluate(run_context.session, step):
                run_context.request_stop()

    def end(self, session):
        last_step = session.estimator.params["solver"].global_step
        if last_step != self._timer.last_triggered_step():
            self._evaluate(session, last_step)

    def _evaluate(self, session, step):
        _poll_nan(session)
        results = self._evaluator.run_with_session(session)
        if self._trigger_counter <= 1:
            return False
        if self._ma_results is None:
            self._ma_results = {k: float(v) for k, v in results.items()}
            self._ma_best_result = float(np.mean(list(results.values())))
            self._need_save = True
        else:
            self._ma_results = {k: float(self.ma_alpha * v + (1 - self.ma_alpha) * results[k])
                                for k, v in self._ma_results.items()}
            new_avg = float(np.mean(list(self._ma_results.values())))
            if self._compare_fn(new_avg, self._ma_best_result):
                self._ma_best_result = new_avg
                self._need_save = True
        self.summaries.append((step, {self._summary_tag.format(k): v for k, v in self._ma_results.items()}))
        if not (self._save_best and self._need_save):
            return False
        self._need_save = False
        strategy = _strategy(session)
        if strategy is not None and strategy.rank != 0:
            return False                      # every rank evaluates (lock-step), rank 0 alone writes
        log.info("Saving (best) checkpoints for %d into %s (checkpoint_best).", step - 1, self._checkpoint_dir)
        session.estimator.save_checkpoint(status_file="checkpoint_best", tag=self._basename)
        with self._best_file().open("w") as f:
            json.dump({"ma_results": self._ma_results, "ma_best_result": float(self._ma_best_result)}, f)
        return False

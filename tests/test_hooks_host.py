"""CPU: training-control hooks (reference core/hooks.py:79-285,471-518,581-723) on fake sessions."""
import argparse
import json
import types

import numpy as np
import pytest

from boxsegliver_amd.core import hooks
from boxsegliver_amd.core.estimator import _RunContext
from boxsegliver_amd.core.solver import Solver
from boxsegliver_amd.evaluators.evaluator_base import EvaluateBase


def _solver(policy="plateau"):
    return Solver(argparse.Namespace(
        learning_rate=1e-3, learning_policy=policy, lr_decay_step=100000, lr_decay_rate=0.1, num_of_total_steps=1000,
        lr_power=0.9, lr_end=1e-6, lr_decay_boundaries=None, lr_custom_values=None, optimizer="Adam"))


class _FakeEstimator(object):
    def __init__(self, solver, model_dir):
        self.params = {"solver": solver}
        self.model_dir = str(model_dir)
        self.saved = []

    def save_checkpoint(self, status_file="checkpoint", tag="model.ckpt"):
        self.saved.append((status_file, tag, self.params["solver"].global_step))
        return tag


def _ctx(est):
    return _RunContext(types.SimpleNamespace(estimator=est))


def _spec(loss, lr):
    return types.SimpleNamespace(loss=loss, train_op=lr, model=types.SimpleNamespace(metrics_dict={}))


def test_second_or_step_timer():
    t = hooks.SecondOrStepTimer(every_steps=3)
    assert t.should_trigger_for_step(1)
    t.update_last_triggered_step(1)
    assert not t.should_trigger_for_step(1) and not t.should_trigger_for_step(3) and t.should_trigger_for_step(4)
    with pytest.raises(ValueError):
        hooks.SecondOrStepTimer()


def test_reduce_lr_on_plateau_decays_then_stops_and_persists(tmp_path):
    solver = _solver()
    est = _FakeEstimator(solver, tmp_path)
    hook = hooks.ReduceLROnPlateauHook(str(tmp_path), lr_patience=1, tr_patience=3, min_delta=0.01, every_n_steps=1,
                                       moving_average=0.0)
    ctx = _ctx(est)
    lrs, stopped_at = [], None
    losses = [1.0, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5]
    for i, loss in enumerate(losses):
        solver.global_step = i + 3                            # the hook ignores steps <= 2 (:637)
        lr = solver._get_model_learning_rate()
        lrs.append(lr)
        hook.after_run(ctx, _spec(loss, lr))
        if ctx.stop_requested:
            stopped_at = i
            break
    # best 1.0 -> 0.5 (improved), then every SECOND stagnating check decays by lr_decay_rate down to lr_end
    assert lrs[:2] == [1e-3, 1e-3] and lrs[3] == pytest.approx(1e-3) and lrs[4] == pytest.approx(1e-4)
    assert min(lrs) >= 1e-6 and solver.plateau_lr == pytest.approx(max(1e-3 * 0.1 ** ((len(lrs) - 2) // 2), 1e-6), rel=1e-6)
    assert stopped_at is not None or hook.tr_wait <= hook.tr_patience     # stops only once lr <= lr_threshold
    state = json.load(open(str(tmp_path / "lr_schedule")))
    assert state["best"] == pytest.approx(0.5) and state["mode"] == "min"
    again = hooks.ReduceLROnPlateauHook(str(tmp_path), lr_patience=1, tr_patience=3, every_n_steps=1)
    assert again.best == pytest.approx(0.5) and again.lr_wait == hook.lr_wait and again.tr_wait == hook.tr_wait


def test_plateau_stop_rule():
    hook = hooks.ReduceLROnPlateauHook("/tmp/does-not-matter-{}".format(np.random.randint(1 << 30)), tr_patience=4)
    hook.tr_wait = 5
    assert hook.check_stop(old_lr=1e-3) is False and hook.tr_wait == 3        # lr above threshold: more patience
    hook.tr_wait = 5
    assert hook.check_stop(old_lr=1e-7) is True
    with pytest.raises(ValueError):
        hooks.ReduceLROnPlateauHook("/tmp/x", mode="auto")


class _ScriptedEvaluator(EvaluateBase):
    def __init__(self, results):
        super(_ScriptedEvaluator, self).__init__()
        self.results = list(results)
        self.calls = 0

    def run_with_session(self, session):
        r = self.results[min(self.calls, len(self.results) - 1)]
        self.calls += 1
        return dict(r)


def test_evaluator_hook_tracks_best_and_saves(tmp_path):
    from boxsegliver_amd.evaluators.evaluator_liver import _compare
    solver = _solver("period_step")
    est = _FakeEstimator(solver, tmp_path)
    ev = _ScriptedEvaluator([{"Liver/Dice": 0.80, "Tumor/Dice": 0.30}, {"Liver/Dice": 0.78, "Tumor/Dice": 0.50},
                             {"Liver/Dice": 0.85, "Tumor/Dice": 0.20}])
    cmp_fn = lambda cur, ori: _compare(cur, ori, primary_metric="Liver/Dice")
    hook = hooks.EvaluatorHook(ev, checkpoint_dir=str(tmp_path), compare_fn=cmp_fn, eval_n_steps=2, save_best=True)
    ctx = _ctx(est)
    for step in range(1, 7):
        solver.global_step = step
        hook.after_run(ctx, _spec(0.0, 1e-3))
    assert ev.calls == 3                                                    # steps 1, 3, 5
    assert est.saved == [("checkpoint_best", "best_model.ckpt", 1), ("checkpoint_best", "best_model.ckpt", 5)]
    assert json.load(open(str(tmp_path / "best_result"))) == {"Liver/Dice": 0.85, "Tumor/Dice": 0.20}
    hook.end(ctx.session)                                                   # last step 6 != last trigger 5 -> evaluates
    assert ev.calls == 4 and len(hook.summaries) == 4 and "Eval/Liver/Dice" in hook.summaries[0][1]
    # a restarted hook loads the best record
    again = hooks.EvaluatorHook(ev, checkpoint_dir=str(tmp_path), compare_fn=cmp_fn, eval_n_steps=2, save_best=True)
    assert again._better_result == {"Liver/Dice": 0.85, "Tumor/Dice": 0.20}
    with pytest.raises(TypeError):
        hooks.EvaluatorHook(object(), checkpoint_dir=str(tmp_path), eval_n_steps=1)


def test_log_learning_rate_hook_records():
    solver = _solver("period_step")
    est = _FakeEstimator(solver, "/tmp")
    hook = hooks.LogLearningRateHook("Liver", every_n_steps=2, do_logging=False)
    ctx = _ctx(est)
    for step in range(1, 6):
        solver.global_step = step
        hook.after_run(ctx, _spec(0.0, 1e-3 * step))
    assert [s for s, _ in hook.records] == [1, 3, 5] and hook.records[1][1] == pytest.approx(3e-3)


def test_evaluator_hook_v2_moving_average_best_checkpoint(tmp_path):
    """core/hooks.py:288-468: the first trigger only evaluates; from the second on every metric is averaged
    (ma <- alpha ma + (1 - alpha) new) and the variables are saved whenever mean(ma) improves; `best_result` keeps both."""
    class _Ev(EvaluateBase):
        def __init__(self, seq):
            super(_Ev, self).__init__()
            self.seq, self.calls = list(seq), 0

        def run_with_session(self, session):
            r = self.seq[min(self.calls, len(self.seq) - 1)]
            self.calls += 1
            return dict(r)

    seq = [{"Liver/Dice": 0.1, "Tumor/Dice": 0.1}, {"Liver/Dice": 0.8, "Tumor/Dice": 0.4}, {"Liver/Dice": 0.9, "Tumor/Dice": 0.6},
           {"Liver/Dice": 0.2, "Tumor/Dice": 0.2}, {"Liver/Dice": 0.2, "Tumor/Dice": 0.2}]
    solver = _solver("period_step")
    est = _FakeEstimator(solver, tmp_path)
    ev = _Ev(seq)
    hook = hooks.EvaluatorHookV2(ev, checkpoint_dir=str(tmp_path), eval_n_steps=2, save_best=True, ma_alpha=0.5)
    ctx = _ctx(est)
    for step in range(1, 10):
        solver.global_step = step
        hook.after_run(ctx, _spec(0.5, 1e-3))
    # triggers at steps 1, 3, 5, 7, 9 -> 5 evaluations; the first is discarded
    assert ev.calls == 5
    ma1 = {"Liver/Dice": 0.8, "Tumor/Dice": 0.4}                                  # second trigger initialises
    ma2 = {k: 0.5 * ma1[k] + 0.5 * seq[2][k] for k in ma1}                        # 0.85, 0.5  -> mean 0.675 > 0.6: save
    ma3 = {k: 0.5 * ma2[k] + 0.5 * seq[3][k] for k in ma1}                        # mean 0.4375: no save
    ma4 = {k: 0.5 * ma3[k] + 0.5 * seq[4][k] for k in ma1}
    assert [s for s, _, _ in [(x[2], 0, 0) for x in est.saved]] == [3, 5]        # saved at the 2nd and 3rd trigger only
    assert all(x[0] == "checkpoint_best" and x[1] == "best_model.ckpt" for x in est.saved)
    best = json.load(open(str(tmp_path / "best_result")))
    assert best["ma_best_result"] == pytest.approx(0.675) and best["ma_results"] == pytest.approx(ma2)
    assert hook._ma_results == pytest.approx(ma4)
    assert [s for s, _ in hook.summaries] == [3, 5, 7, 9]
    # a fresh hook resumes from best_result; end() evaluates once more when the last step was not a trigger
    hook2 = hooks.EvaluatorHookV2(_Ev([{"Liver/Dice": 1.0, "Tumor/Dice": 1.0}]), checkpoint_dir=str(tmp_path), eval_n_steps=100,
                                  save_best=True, ma_alpha=0.5)
    assert hook2._ma_best_result == pytest.approx(0.675) and hook2._ma_results == pytest.approx(ma2)
    solver.global_step = 11
    hook2.end(ctx.session)
    assert hook2._evaluator.calls == 1 and len(est.saved) == 2                     # trigger counter 0: evaluated, not averaged
    with pytest.raises(TypeError):
        hooks.EvaluatorHookV2(object())

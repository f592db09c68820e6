"""GPU: two small semantic points of the reference the round-3 review found untested / not restated.

* `--bias_decay` (NetworksV2/base.py:128-135, literal reading: biases carry the weights' L2 regulariser UNLESS the flag is
  given) -- the branch with the flag set, whole step and Adam trajectory against the oracle.
* NanTensorHook(loss) (core/estimator.py:676) fires at EVERY step: here a sticky device flag written by a one-thread kernel
  each step and polled where the host synchronises anyway -- a NaN between two log steps must stop the run before the
  next checkpoint is written."""
import os

import numpy as np
import pytest
import torch

from oracle import solver as osolver

pytestmark = pytest.mark.gpu


def test_bias_decay_flag_given_matches_oracle_and_moves_biases_out_of_the_regulariser():
    import test_gpu_unet as t
    from boxsegliver_amd.core.solver import Solver
    images, labels = t.synth(2, 32, 32, 3)
    losses = {}
    for flag in (True, False):
        args = t.make_args(bias_decay=flag, weight_decay_rate=1e-2)           # large enough for the term to matter at 1e-4
        model, inputs = t.build(args, images, labels)
        net, params = t.oracle_for(args)
        for name, _, kind in net.specs:
            if kind == "bias":
                params[name] = params[name] * 20.0                             # |b| ~ 2: sum b^2 is visible in the loss
        model.params.load_state(params)
        bias_names = [n for n, _, k in net.specs if k == "bias"]
        assert bias_names
        in_reg = set(n for n in model.params.tensors if model.params.where[n][0] == "reg")
        p = {k: v.clone() for k, v in params.items()}
        opt = osolver.TFAdam(0.9, 0.99, 1e-8)
        solver = Solver(args)
        got, ref = [], []
        for step in range(3):
            total, data, _, grads, new_stats = net.loss_and_grads(p, torch.from_numpy(images), torch.from_numpy(labels).long(),
                                                                  **t.loss_kwargs(args))
            ref.append(total.item())
            if step == 0:
                losses[flag] = (total.item(), data.item(), sum(float((params[n].double() ** 2).sum()) for n in bias_names))
            opt.step({k: p[k].numpy() for k in grads}, {k: g.numpy() for k, g in grads.items()}, 1e-3)
            for k, v in new_stats.items():
                p[k] = v
            loss = model(inputs, "train", **t.YML)
            got.append(loss.item())
            solver(loss, model)
        np.testing.assert_allclose(got, ref, rtol=1e-4)
        # after three optimiser steps the biases themselves agree (their update carries wd * b only without the flag)
        sd = model.params.state_dict()
        for n in bias_names:
            np.testing.assert_allclose(sd[n].cpu().numpy(), p[n].numpy(), rtol=0, atol=2e-4)     # three Adam steps move a bias by ~3e-3
        assert all((n in in_reg) == (not flag) for n in bias_names)
    # the two branches differ by exactly 0.5 * wd * sum(b^2) (slim.l2_regularizer = wd * tf.nn.l2_loss)
    (tot_t, data_t, sb), (tot_f, data_f, _) = losses[True], losses[False]
    assert abs(data_t - data_f) < 1e-6
    assert abs((tot_f - tot_t) - 0.5 * 1e-2 * sb) < 1e-4 and (tot_f - tot_t) > 1e-3


def test_nan_loss_between_log_steps_stops_training_before_the_next_checkpoint(tmp_path):
    import test_gpu_unet as t
    from boxsegliver_amd.NetworksV2.UNet import UNet
    from boxsegliver_amd.core import estimator as est
    from boxsegliver_amd.core import models
    from boxsegliver_amd.core.solver import Solver
    from boxsegliver_amd.data import synthetic

    class Poison(object):                       # a hook: after the 2nd step one weight becomes NaN -> step 3's loss is NaN
        def __init__(self):
            self.n = 0

        def begin(self):
            pass

        def after_create_session(self, session):
            pass

        def before_run(self, ctx):
            pass

        def after_run(self, ctx, spec):
            self.n += 1
            if self.n == 2:
                spec.model.params["UNet/Encode1/Repeat/convolution2d_1/weights"].data.view(-1)[0] = float("nan")
                from boxsegliver_amd import ops
                ops.PARAM_GEN += 1                # the variable changed behind the pack cache's back

        def end(self, session):
            pass

    args = t.make_args(batch_size=2, im_height=32, im_width=32, model="UNet", noise_scale=0.05, synthetic_batches=2,
                       log_step=1000)                                      # log steps: 1 only (done == 1), then never
    params = {"args": args, "model": UNet, "model_kwargs": dict(t.YML), "model_args": (), "solver": Solver(args),
              "solver_kwargs": {}}
    e = est.CustomEstimator(models.model_fn, str(tmp_path), est.RunConfig(model_dir=str(tmp_path), save_checkpoints_steps=4),
                            params)
    with pytest.raises(est.NanLossDuringTrainingError):
        e.train(synthetic.input_fn, steps=8, hooks=[Poison()])
    seen, at = (int(v) for v in e._nan_flag.tolist())
    assert seen == 1 and at == 3                                           # the first NaN step, not the polling step
    assert not [f for f in os.listdir(str(tmp_path)) if f.startswith("model.ckpt")]     # step 4's checkpoint was refused
    # and a healthy run of the same length does checkpoint at step 4 and at the end
    args2 = t.make_args(batch_size=2, im_height=32, im_width=32, model="UNet", noise_scale=0.05, synthetic_batches=2,
                        log_step=1000)
    params2 = {"args": args2, "model": UNet, "model_kwargs": dict(t.YML), "model_args": (), "solver": Solver(args2),
               "solver_kwargs": {}}
    d2 = tmp_path / "ok"
    e2 = est.CustomEstimator(models.model_fn, str(d2), est.RunConfig(model_dir=str(d2), save_checkpoints_steps=4), params2)
    e2.train(synthetic.input_fn, steps=5)
    assert [int(v) for v in e2._nan_flag.tolist()] == [0, 0]
    assert any(f.startswith("model.ckpt-5") for f in os.listdir(str(d2)))

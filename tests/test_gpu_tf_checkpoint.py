"""GPU: a model_dir / weights file in the REFERENCE's on-disk format (TensorFlow V2 checkpoint written by tf.train.Saver:
variables under their graph names, Adam slots "Optimizer/<variable>/Adam{,_1}", `global_step`, the text `checkpoint`
status file) resumes training here exactly where an uninterrupted run continues (core/estimator.py:52-59,646-741), and
--load_weights / --weights_scope (core/models.py:151-185) initialise a model under another scope from it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(tmp_path, **over):
    import test_gpu_unet as t
    from boxsegliver_amd.NetworksV2.UNet import UNet
    from boxsegliver_amd.core.solver import Solver
    from boxsegliver_amd.data import synthetic
    args = t.make_args(batch_size=2, im_height=32, im_width=32, model="UNet", noise_scale=0.05, synthetic_batches=2,
                       log_step=100, model_dir=str(tmp_path / "run"), load_weights=None, load_weights_version="checkpoint",
                       weights_scope=None, **over)
    images, labels, _ = synthetic.make_batch(2, 32, 32, 3, 3, 77)
    inputs = {"images": torch.from_numpy(images).cuda(), "labels": torch.from_numpy(labels).cuda()}
    return t, UNet, Solver, args, inputs


def _export_tf(prefix, model, solver, scope=None):
    """What tf.train.Saver would have written for this state (core/estimator.save_tf_checkpoint)."""
    from boxsegliver_amd.core import estimator as est
    from boxsegliver_amd.utils import tf_checkpoint as tfc
    out = est.save_tf_checkpoint(prefix, model, solver, root_scope=scope)
    names = tfc.CheckpointReader(out).get_variable_to_shape_map()
    root = scope or model.name
    assert "global_step" in names and "Optimizer/beta1_power" in names and names["Optimizer/beta2_power"] == []
    assert names["Optimizer/{}/AdjustChannels/weights/Adam_1".format(root)] == [1, 1, 64, 3]
    assert names["{}/Encode1/Repeat/convolution2d_1/BatchNorm/moving_variance".format(root)] == [64]
    assert not any(k.endswith("moving_mean/Adam") for k in names)          # statistics have no slots
    return out


def test_resume_from_reference_format_checkpoint_is_exact(tmp_path):
    from boxsegliver_amd.core import estimator as est
    t, UNet, Solver, args, inputs = _setup(tmp_path)
    (tmp_path / "run").mkdir()
    model, solver = UNet(args), Solver(args)
    for _ in range(3):
        solver(model(inputs, "train", **t.YML), model)
    prefix = _export_tf(tmp_path / "run" / "model.ckpt-3", model, solver)
    (tmp_path / "run" / "checkpoint").write_text('model_checkpoint_path: "model.ckpt-3"\nall_model_checkpoint_paths: "model.ckpt-3"\n')
    cont = []
    for _ in range(2):
        loss = model(inputs, "train", **t.YML)
        cont.append(loss.item())
        solver(loss, model)
    # a fresh process: the estimator finds the TF status file, reads step and variables
    assert est._load_global_step_from_checkpoint_dir(str(tmp_path / "run")) == 3
    e = est.CustomEstimator(lambda *a, **k: None, str(tmp_path / "run"), est.RunConfig(model_dir=str(tmp_path / "run")),
                            {"args": args})
    assert e.checkpoint_path() == prefix
    model2, solver2 = UNet(args), Solver(args)
    model2(inputs, "eval", **t.YML)                                      # creates the variables
    e._maybe_restore(model2, solver2)
    assert solver2.global_step == 3
    resumed = []
    for _ in range(2):
        loss = model2(inputs, "train", **t.YML)
        resumed.append(loss.item())
        solver2(loss, model2)
    assert resumed == cont                                               # bit-identical continuation
    for k, v in model.params.state_dict().items():
        assert torch.equal(v, model2.params.state_dict()[k]), k
    # a variable missing from the file is an error, not a silent default
    from boxsegliver_amd.utils import tf_checkpoint as tfc
    partial = {k: v for k, v in tfc.read_checkpoint(prefix).items() if "AdjustChannels/biases" not in k}
    tfc.write_checkpoint(tmp_path / "partial", partial)
    with pytest.raises(KeyError):
        est.restore_variables(str(tmp_path / "partial"), model2)


def test_load_weights_with_scope_from_reference_checkpoint_dir(tmp_path):
    from boxsegliver_amd.core import estimator as est
    from boxsegliver_amd.core import models
    t, UNet, Solver, args, inputs = _setup(tmp_path)
    src, solver = UNet(args), Solver(args)
    solver(src(inputs, "train", **t.YML), src)
    wdir = tmp_path / "pretrained"
    wdir.mkdir()
    _export_tf(wdir / "model.ckpt-1", src, solver, scope="OldNet")
    (wdir / "checkpoint_best").write_text('model_checkpoint_path: "model.ckpt-1"\n')
    assert models._find_root_scope(str(wdir / "model.ckpt-1")) == "OldNet"
    args.load_weights, args.load_weights_version = "pretrained", "checkpoint_best"     # a directory next to model_dir
    dst = UNet(args)
    dst(inputs, "eval", **t.YML)
    init_fn = models.init_model(dst, args)
    init_fn(None, None)
    for k, v in src.params.state_dict().items():
        assert torch.equal(v, dst.params.state_dict()[k]), k
    # through the estimator: no checkpoint in model_dir -> the init_fn runs after variable creation
    e = est.CustomEstimator(lambda *a, **k: None, str(tmp_path / "run"), est.RunConfig(model_dir=str(tmp_path / "run")),
                            {"args": args})
    dst2, solver2 = UNet(args), Solver(args)
    dst2(inputs, "eval", **t.YML)
    e._maybe_restore(dst2, solver2)
    assert solver2.global_step == 0 and torch.equal(dst2.params.state_dict()["UNet/Encode1/Repeat/convolution2d_1/weights"],
                                                    src.params.state_dict()["UNet/Encode1/Repeat/convolution2d_1/weights"])
    args.load_weights = "nowhere"
    with pytest.raises(FileNotFoundError):
        models.init_model(dst, args)
    args.load_weights = None
    assert models.init_model(dst, args) is None


def test_offline_eval_and_predict_restore_a_tf_prefix_and_refuse_missing_ones(tmp_path):
    """`model.ckpt-3` exists only as .index / .data-* files -- evaluator.run and estimator.predict must
    restore it (not silently score the fresh initialisation) and must raise FileNotFoundError for a path that is not a
    checkpoint (reference evaluators/evaluator_liver.py:705-708)."""
    from boxsegliver_amd.core import estimator as est
    from boxsegliver_amd.core import models
    from boxsegliver_amd.data import synthetic
    from boxsegliver_amd.evaluators import evaluator_liver as ev
    t, UNet, Solver, args, inputs = _setup(tmp_path, eval_mirror=False, random_flip=0, metrics_eval=["Dice"],
                                           use_global_dice=False, pred_type="pred", mode="eval", eval_num=-1, save_path=None)
    (tmp_path / "run").mkdir()
    src, solver = UNet(args), Solver(args)
    for _ in range(2):
        solver(src(inputs, "train", **t.YML), src)
    prefix = _export_tf(tmp_path / "run" / "model.ckpt-2", src, solver)
    want = src.params.state_dict()

    params = {"args": args, "model": UNet, "model_kwargs": dict(t.YML), "model_args": (), "eval_cases": [(7, 4)]}
    e = est.CustomEstimator(models.model_fn, str(tmp_path / "run"), est.RunConfig(model_dir=str(tmp_path / "run")), params)
    evaluator = ev.get_evaluator("Volume", estimator=e, model_dir=str(tmp_path / "run"), params=params)
    evaluator.run(synthetic.input_fn_eval_volumes, checkpoint_path=prefix)
    got = evaluator._model().params.state_dict()
    for k, v in want.items():
        assert torch.equal(v, got[k]), k
    with pytest.raises(FileNotFoundError):
        evaluator.run(synthetic.input_fn_eval_volumes, checkpoint_path=str(tmp_path / "run" / "model.ckpt-9"))

    params2 = {"args": args, "model": UNet, "model_kwargs": dict(t.YML), "model_args": ()}
    e2 = est.CustomEstimator(models.model_fn, str(tmp_path / "run"), est.RunConfig(model_dir=str(tmp_path / "run")), params2)
    next(e2.predict(synthetic.input_fn, checkpoint_path=prefix))
    got2 = params2["model_instances"][0].params.state_dict()
    for k, v in want.items():
        assert torch.equal(v, got2[k]), k
    with pytest.raises(FileNotFoundError):
        next(e2.predict(synthetic.input_fn, checkpoint_path=str(tmp_path / "run" / "model.ckpt-9")))


def test_save_after_resuming_from_a_tf_model_dir_and_plateau_lr_travels(tmp_path):
    """ADVICE r1 (medium x2): after a resume from a reference model_dir the status file is TensorFlow's text
    CheckpointState -- the next save must replace it (no JSONDecodeError) and leave the TF bundle on disk; the plateau
    learning-rate variable `Optimizer/learning_rate/value` (solver.py:246-254) is exported and restored."""
    import json
    import os
    from boxsegliver_amd.core import estimator as est
    from boxsegliver_amd.utils import tf_checkpoint as tfc
    t, UNet, Solver, args, inputs = _setup(tmp_path, learning_policy="plateau", lr_decay_rate=0.2)
    run = tmp_path / "run"
    run.mkdir()
    model, solver = UNet(args), Solver(args)
    solver(model(inputs, "train", **t.YML), model)
    solver.update_plateau_lr()                                         # 1e-3 -> 2e-4, as the plateau hook would
    prefix = est.save_tf_checkpoint(run / "model.ckpt-1", model, solver)
    assert float(tfc.CheckpointReader(prefix).get_tensor("Optimizer/learning_rate/value")) == pytest.approx(2e-4)
    (run / "checkpoint").write_text('model_checkpoint_path: "model.ckpt-1"\nall_model_checkpoint_paths: "model.ckpt-1"\n')

    params = {"args": args, "solver": Solver(args)}
    e = est.CustomEstimator(lambda *a, **k: None, str(run), est.RunConfig(model_dir=str(run)), params)
    model2 = UNet(args)
    model2(inputs, "eval", **t.YML)
    params["model_instances"] = [model2]
    e._maybe_restore(model2, params["solver"])
    assert params["solver"].global_step == 1 and params["solver"].plateau_lr == pytest.approx(2e-4)
    assert params["solver"]._get_model_learning_rate() == pytest.approx(2e-4)
    fname = e.save_checkpoint()                                        # used to raise JSONDecodeError here
    assert fname == "model.ckpt-1.pt" and json.load(open(str(run / "checkpoint")))["global_step"] == 1
    assert os.path.exists(prefix + ".index") and os.path.exists(prefix + ".data-00000-of-00001")   # TF bundle kept
    params["solver"](model2(inputs, "train", **t.YML), model2)
    assert e.save_checkpoint() == "model.ckpt-2.pt" and not os.path.exists(str(run / "model.ckpt-1.pt"))
    assert os.path.exists(prefix + ".index")

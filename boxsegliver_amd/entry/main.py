"""Command-line entry point -- the drop-in for the reference's entry/main.py (UNet / UNet3D / ...) and, through
entry/main_g.py, entry/main_g.py (GUNet):

    python -m boxsegliver_amd.entry.main <subcommand> --mode train --tag NAME --model UNet --classes Liver Tumor ...

The first positional argument selects the dataset pipeline and evaluator exactly as in the reference (entry/main.py:53-77:
only_liver | liver | nf | nf_inter | nf_3d; entry/main_g.py:55-73: liver | nf | nf2 | nf_inter); every flag of the six
argument groups the reference merges (config, core.models, core.solver, loss_metrics, <pipeline>, <evaluator>) is accepted
with the same name and default, so the shipped run scripts' flag lists parse unchanged (tests/test_entry_host.py does that
with run_scripts/template/001_unet.sh, scripts/102_gnet_v1.sh and threed_script/201_unet_v1.sh's lists).

What differs, by design: multi-GPU is ONE PROCESS PER GPU (launch with `python -m torch.distributed.run --nproc-per-node N
-m boxsegliver_amd.entry.main ... --distribution_strategy mirrored --num_gpus N`; the reference replicates inside one
process), and the NF sub-commands run on synthetic tensors with the NF contract (the dataset is private; data/nf.py)."""
import argparse
import functools
import logging
import sys
from pathlib import Path

from .. import config, loss_metrics
from ..NetworksV2.base import ModeKeys
from ..core import estimator as estimator_lib
from ..core import hooks, models, solver
from ..utils import distribution_utils

log = logging.getLogger("boxsegliver_amd")

KEEP_CHECKPOINT_MAX = 1                     # entry/main.py:42


def _liver(pipeline):
    from ..data import flagsets, lits
    from ..evaluators import evaluator_liver

    def add(parser):
        flagsets.add_arguments(parser, pipeline)
        parser.add_argument("--lits_root", type=str, default="data/LiTS", help="where png/, meta.json and k_folds.txt live")
        parser.add_argument("--seed", type=int, default=1234)
    return add, lits.input_fn, lits.input_fn_eval, evaluator_liver


def _nf(pipeline):
    from ..data import nf
    from ..evaluators import evaluator_liver            # EvaluateVolume is shared (evaluator_nf adds nothing the hot path uses)
    return nf.add_arguments_for(pipeline), nf.input_fn, nf.input_fn, evaluator_liver


# sub-command -> pipeline: entry/main.py:53-77 and entry/main_g.py:55-73
SUBCOMMANDS = {
    False: {"only_liver": lambda: _liver("liver_li"), "liver": lambda: _liver("liver"), "nf": lambda: _nf("nf"),
            "nf_inter": lambda: _nf("nf_g_simply"), "nf_3d": lambda: _nf("nf_3d")},
    True: {"liver": lambda: _liver("liver_g"), "nf": lambda: _nf("nf_g"), "nf2": lambda: _nf("nf_iin"),
           "nf_inter": lambda: _nf("nf_g_simply")},
}


def get_arguments(argv, guided=False):
    """entry/main.py:45-85: assemble the parser from the six groups; returns (args, subcommand, pipeline tuple)."""
    table = SUBCOMMANDS[bool(guided)]
    if len(argv) < 1:
        raise ValueError("Please choice first argument from [{}]".format(", ".join(table)))
    sub = argv[0]
    if sub not in table and sub not in ("-h", "--help"):
        raise ValueError("First argument must be choose from [{}], got {}".format(", ".join(table), sub))
    parser = argparse.ArgumentParser(prog="boxsegliver_amd.entry." + ("main_g" if guided else "main"))
    config.add_arguments(parser)
    models.add_arguments(parser)
    solver.add_arguments(parser)
    loss_metrics.add_arguments(parser)
    if sub in ("-h", "--help"):
        parser.parse_args(["--help"])
    pipe = table[sub]()
    pipe[0](parser)
    pipe[3].add_arguments(parser)
    args = parser.parse_args(argv[1:])
    config.check_args(args, parser)
    config.fill_default_args(args)
    return args, sub, pipe


def _setup_logging(args):
    """entry/main.py:101-114: <model_dir>/logs/<mode>_<tag> (or --out_file) next to the console."""
    log_dir = Path(args.model_dir) / "logs"
    log_dir.mkdir(parents=True, exist_ok=True)
    handler = logging.FileHandler(str(log_dir / (args.out_file or "{}_{}".format(args.mode, args.tag))))
    handler.setFormatter(logging.Formatter("%(asctime)s %(levelname)s %(message)s"))
    log.addHandler(handler)
    log.setLevel(logging.INFO)


def run(args, sub, pipe, guided=False, dataset_params=None):
    _, input_fn, input_fn_eval, evaluator_lib = pipe
    _setup_logging(args)
    log.debug(args)
    if args.num_gpus < 2:
        args.distribution_strategy = "off"
    strategy = distribution_utils.get_distribution_strategy(distribution_strategy=args.distribution_strategy,
                                                            num_gpus=args.num_gpus, num_workers=1,
                                                            all_reduce_alg=args.all_reduce_alg)
    extra = dict(dataset_params or {})
    if hasattr(args, "lits_root"):
        extra.setdefault("lits_root", args.lits_root)
    if strategy is not None:
        extra.setdefault("rank", strategy.rank)
        extra.setdefault("strategy", strategy)       # data/lits.SliceStore: each rank decodes 1/N of the dataset, then exchange

    if args.mode == ModeKeys.TRAIN:
        run_config = estimator_lib.RunConfig(model_dir=args.model_dir, train_distribute=strategy, save_checkpoints_steps=5000,
                                             keep_checkpoint_max=KEEP_CHECKPOINT_MAX, log_step_count_steps=args.log_step)
        params = {"args": args}
        params.update(extra)
        params.update(models.get_model_params(args, build_metrics=True, build_summaries=bool(args.log_step)))
        params.update(solver.get_solver_params(args, warm_up=args.lr_warm_up, slow_start_step=args.slow_start_step,
                                               slow_start_learning_rate=args.slow_start_lr))
        if args.eval_per_epoch:
            params["double_dataloader_modes"] = [ModeKeys.TRAIN, "eval_online"]
        if guided:
            params["save_best_ckpt"] = args.save_best
        estimator = estimator_lib.CustomEstimator(models.model_fn, args.model_dir, run_config, params, args.warm_start_from)
        train_hooks = [hooks.LogLearningRateHook(prefix=args.summary_prefix, every_n_steps=args.log_step,
                                                 output_dir=args.model_dir, do_logging=False)]
        if args.learning_policy == "plateau":
            plateau_kw = dict(min_delta=args.min_delta) if guided else dict(tr_patience=50, min_delta=1e-4)
            train_hooks.append(hooks.ReduceLROnPlateauHook(args.model_dir, lr_patience=args.lr_patience,
                                                           every_n_steps=args.batches_per_epoch, **plateau_kw))
        if args.eval_per_epoch:
            kw = dict(use_sg_reduce_fp=False) if guided else {}
            evaluator = evaluator_lib.get_evaluator(args.evaluator, estimator=estimator, **kw)
            if guided and not args.save_interval:            # entry/main_g.py:174-178
                train_hooks.append(hooks.EvaluatorHookV2(evaluator, checkpoint_dir=estimator.model_dir, prefix=args.summary_prefix,
                                                         eval_n_steps=args.batches_per_epoch, save_best=args.save_best))
            else:
                train_hooks.append(hooks.EvaluatorHook(
                    evaluator, checkpoint_dir=estimator.model_dir,
                    compare_fn=functools.partial(evaluator.compare, primary_metric=args.primary_metric,
                                                 secondary_metric=args.secondary_metric),
                    prefix=args.summary_prefix, eval_n_steps=args.batches_per_epoch, save_best=args.save_best,
                    save_interval=args.save_interval))
        steps, max_steps = (args.num_of_steps, None) if args.num_of_steps > 0 else (None, args.num_of_total_steps)
        try:
            estimator.train(input_fn, hooks=train_hooks, steps=steps, max_steps=max_steps)
        except KeyboardInterrupt:
            log.info("Main process terminated by user.")
        finally:
            log.info("Clean up!")
        log.info("Process end.")
        return estimator

    if args.mode in (ModeKeys.EVAL, ModeKeys.PREDICT):
        params = {"args": args}
        params.update(extra)
        params.update(models.get_model_params(args))
        estimator = estimator_lib.CustomEstimator(models.model_fn, args.model_dir,
                                                  estimator_lib.RunConfig(model_dir=args.model_dir), params)
        evaluator = evaluator_lib.get_evaluator(args.evaluator, estimator=estimator, model_dir=args.model_dir, params=params)
        ckpt = estimator.checkpoint_path(args.ckpt_path, args.load_status_file if not args.eval_final else None)
        if not ckpt:
            raise FileNotFoundError("Missing checkpoint file in {} with status_file {}".format(
                args.model_dir, args.load_status_file if not args.eval_final else None))
        return evaluator.run(input_fn_eval, checkpoint_path=ckpt, save=args.save_predict)

    raise ValueError("--mode {} is not built (TF-Serving export is out of scope, SURVEY.md 2 row 15)".format(args.mode))


def main(argv=None, guided=False):
    args, sub, pipe = get_arguments(list(sys.argv[1:] if argv is None else argv), guided)
    run(args, sub, pipe, guided)
    return 0


if __name__ == "__main__":
    sys.exit(main())

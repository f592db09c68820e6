"""Global + device flags -- mirror of the reference's config.py:26-133 (names, defaults, choices verbatim)."""
from pathlib import Path


class CustomKeys(object):
    LEARNING_RATE = "learning_rate"
    LOSS_MEAN = "total_loss_mean"
    LR_UPDATE_OPS = "lr_update_ops"


def add_arguments(parser):
    group = parser.add_argument_group(title="Global Arguments")
    group.add_argument("--mode", type=str, choices=["train", "eval", "infer", "export"], required=True,
                       help="Model mode for train/val/test")
    group.add_argument("--tag", type=str, required=True, help="Configuration tag(like UID)")
    group.add_argument("--model_dir", type=str, default="", help="Directory to save model parameters, graph and etc")
    group.add_argument("-s", "--save_predict", action="store_true", help="Save prediction to file")
    group.add_argument("--warm_start_from", type=str, help="Warm start the model from a checkpoint")
    group.add_argument("-l", "--load_status_file", type=str, default="checkpoint",
                       help="Status file to locate checkpoint file. Use for restore parameters.")
    group.add_argument("--out_file", type=str, help="Logging file name to replace default.")
    group.add_argument("--summary_prefix", type=str, help="A string that will be prepend to the summary tags.")
    group.add_argument("--save_best", action="store_true", help="Save best checkpoint")
    group.add_argument("--save_interval", type=int, default=0, help="Save best checkpoint in each interval")
    group.add_argument("--log_step", type=int, default=500, help="Log running information per `log_step`")
    group.add_argument("--min_delta", type=float, default=5e-4, help="min_delta for pleatau lr strategy")

    group = parser.add_argument_group(title="Device Arguments")
    group.add_argument("--distribution_strategy", type=str, default="off",
                       choices=['off', 'default', 'one_device', 'mirrored', 'parameter_server'], required=False,
                       help="A string specify which distribution strategy to use (default: %(default)s)")
    group.add_argument("--num_gpus", type=int, default=1, required=False, help="Number of gpus to run this model")
    group.add_argument("--all_reduce_alg", type=str, default="", choices=["", "hierarchical_copy", "nccl"],
                       required=False, help="Specify which algorithm to use when performing all-reduce")
    group.add_argument("--device_mem_frac", type=float, default=0., required=False,
                       help="Used for per_process_gpu_memory_fraction")
    group.add_argument("--fix", action="store_true", help="Remove norm+relu in spatial guide module")


def check_args(args, parser):
    """config.py:96-125"""
    if hasattr(args, "loss_weight_type"):
        if args.loss_weight_type == "numerical":
            if not args.loss_numeric_w:
                raise parser.error("loss_weight_type==numerical need parameter: --loss_numeric_w")
            if len(args.loss_numeric_w) != len(args.classes) + 1:
                raise parser.error("Asserting len(args.loss_numeric_w) = len(args.classes) + 1 failed!")
        elif args.loss_weight_type == "proportion":
            if not args.loss_proportion_decay:
                raise parser.error("loss_weight_type==proportion need parameter: --loss_proportion_decay")
    for key in ("primary_metric", "secondary_metric"):
        val = getattr(args, key, None)
        if val:
            parts = val.split("/")
            if len(parts) == 2:
                if parts[0] not in args.classes or parts[1] not in args.metrics_eval:
                    raise ValueError("Wrong {}: {}".format(key, val))
    if not args.summary_prefix:
        args.summary_prefix = args.tag


def fill_default_args(args):
    """config.py:128-133: model_dir defaults to <package>/model_dir/<tag>."""
    if not args.model_dir:
        model_dir = Path(__file__).parent / "model_dir"
        args.model_dir = str(model_dir / args.tag)

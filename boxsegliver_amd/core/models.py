"""Model registry, model flags and model_fn -- mirror of the reference's core/models.py.

Registry + `--model/--classes` plugin surface: models.py:36-89; yml lookup `get_model_params`:
:92-118; `model_fn` contract: :224-281.
"""
import copy
import logging
import os
from collections import namedtuple
from pathlib import Path

import yaml

from ..NetworksV2.GUNet import GUNet
from ..NetworksV2.UNet import UNet
from ..NetworksV2.UNet3D import UNet3D
from ..NetworksV2.UNetInter import UNetInter
from ..NetworksV2.SmallUNet import SmallUNet
from ..NetworksV2.LGNet import LGNet
from ..NetworksV2.InterUNet import InterUNet
from ..NetworksV2.base import ModeKeys

log = logging.getLogger("boxsegliver_amd")

# Available models (reference models.py:36-38 lists UNet, GUNet, UNetInter, LGNet, UNet3D, SmallUNet,
# InterUNet (DenseUNet is commented out there); this build ships all seven.
MODEL_ZOO = [
    UNet, GUNet, UNetInter, LGNet, UNet3D, SmallUNet, InterUNet,
]

EstimatorSpec = namedtuple("EstimatorSpec", ["mode", "loss", "train_op", "predictions", "model"])


def add_arguments(parser):
    """core/models.py:41-89 (names / defaults verbatim)."""
    group = parser.add_argument_group(title="Model Arguments")
    group.add_argument("--model", type=str, choices=[cls.__name__ for cls in MODEL_ZOO], required=True,
                       help="Model backbone")
    group.add_argument("--model_config", type=str, required=False, help="Model configuration. (default: <model>.yml)")
    group.add_argument("--classes", type=str, nargs="+", required=True, help="Class names of the objects")
    group.add_argument("--batch_size", type=int, default=8, required=False, help="Model batch size (default: %(default)d)")
    group.add_argument("--weight_init", type=str, default="xavier", choices=["trunc_norm", "xavier"], required=False,
                       help="Model variable initialization method (default: %(default)s)")
    group.add_argument("--normalizer", type=str, default="batch_norm", choices=["batch_norm", "instance_norm"],
                       required=False, help="Normalization method (default: %(default)s)")
    group.add_argument("--cls_branch", action="store_true", required=False, help="Classify branch")
    group.add_argument("--load_weights", type=str, required=False,
                       help="Initialize model parameters from this given ckpt file.")
    group.add_argument("--load_weights_version", type=str, default="checkpoint", help="Used for latest_filename")
    group.add_argument("--weights_scope", type=str, required=False,
                       help="Network scope of the weights in the given ckpt file")
    group.add_argument("--without_norm", action="store_true", required=False, help="Conv without batch normalization")
    group.add_argument("--batches_per_epoch", type=int, default=2000, help="Number of batches per epoch")
    group.add_argument("--eval_per_epoch", action="store_true")
    group.add_argument("--dropout", type=float, help="Dropout for backbone networks")
    group.add_argument("--img_grad", action="store_true", help="Use image gradients")
    # not a reference flag: arithmetic of the 3x3 contractions on MI355X (BASELINE.json configs[1] fp32 / configs[2] bf16)
    group.add_argument("--compute_dtype", type=str, default="fp32", choices=["fp32", "bf16", "bf16c"],
                       help="fp32: exact fp32 MFMA (default). bf16: bf16 MFMA + bf16 storage of activations / activation "
                            "gradients, fp32 accumulate / statistics / master weights. bf16c: bf16 MFMA operands only, "
                            "fp32 storage")
    group.add_argument("--mid_cat", action="store_true", help="Concat guide to middle layers")


def get_model_params(args, build_metrics=False, build_summaries=False):
    """core/models.py:92-118"""
    params = dict()
    zoo = {cls.__name__: cls for cls in MODEL_ZOO}
    if args.model not in zoo:
        raise NameError("name '{}' is not defined".format(args.model))
    params["model"] = zoo[args.model]

    if not getattr(args, "model_config", None):
        args.model_config = args.model + ".yml"
    model_config_path = Path(__file__).parent.parent / "NetworksV2" / args.model_config
    if not model_config_path.exists():
        model_config_path = model_config_path.parent / "ext_config" / args.model_config
        if not model_config_path.exists():
            model_config_path = None
    if model_config_path:
        with model_config_path.open() as f:
            params["model_kwargs"] = yaml.load(f, Loader=yaml.Loader) or {}
    else:
        params["model_kwargs"] = {}
    params["model_kwargs"]["build_metrics"] = build_metrics
    params["model_kwargs"]["build_summaries"] = build_summaries
    return params


def _find_root_scope(ckpt_filename):
    """core/models.py:151-157: the model scope of a checkpoint = the second component of its optimiser slot names."""
    from ..utils import tf_checkpoint
    if os.path.exists(str(ckpt_filename) + ".index"):
        variables = list(tf_checkpoint.CheckpointReader(ckpt_filename).get_variable_to_shape_map())
    else:
        import torch
        variables = list(torch.load(ckpt_filename, map_location="cpu", weights_only=False)["variables"])
        return variables[0].split("/")[0] if variables else None       # this package's files hold no slot names
    for var in variables:
        if var.startswith("Optimizer") and not var.endswith("power"):
            return var.split("/")[1]
    return None


def init_model(model, args):
    """core/models.py:160-185 (--load_weights / --load_weights_version / --weights_scope): an init_fn(scaffold, session)
    that loads the model's variables -- renamed from `model.name` to the checkpoint's root scope -- from a checkpoint file
    or the latest one of a directory next to model_dir; TensorFlow V2 checkpoints of the reference or files of this
    package.  None without --load_weights."""
    if not getattr(args, "load_weights", None):
        return None
    from ..utils import tf_checkpoint
    from .estimator import restore_variables
    weights_dir = Path(args.model_dir).parent / args.load_weights
    ckpt_filename = args.load_weights
    if weights_dir.is_dir():
        latest = tf_checkpoint.get_checkpoint_state(weights_dir, getattr(args, "load_weights_version", None))
        if latest:
            ckpt_filename = latest
    if not tf_checkpoint.checkpoint_exists(ckpt_filename):
        raise FileNotFoundError("ckpt_filename {} doesn't exist".format(ckpt_filename))
    root_scope = getattr(args, "weights_scope", None) or _find_root_scope(ckpt_filename)
    log.info("Create init_fn with checkpoint: " + str(ckpt_filename))

    def init_fn(scaffold, session):
        _ = scaffold, session
        restore_variables(ckpt_filename, model, None, root_scope=root_scope)

    return init_fn


def model_fn(features, labels, mode, params, config=None):
    """core/models.py:224-281.  The model instance is created once and cached in
    params["model_instances"] (the reference appends one per graph build)."""
    features = copy.copy(features)
    images = features.pop("images")
    if labels is None and "labels" in features:
        labels = features.pop("labels")
    inputs = {"images": images, "labels": labels}
    inputs.update(features)

    args = params["args"]
    if not params.get("model_instances"):
        params["model_instances"] = [params["model"](args)]
    model = params["model_instances"][0]
    model_args = params.get("model_args", ())
    model_kwargs = params.get("model_kwargs", {})

    loss = model(inputs, mode, *model_args, **model_kwargs)

    train_op = None
    if mode == ModeKeys.TRAIN:
        solver = params["solver"]
        solver_args = params.get("solver_args", ())
        solver_kwargs = params.get("solver_kwargs", {})
        train_op = solver(loss, model, *solver_args, **solver_kwargs)

    predictions = None
    if getattr(args, "eval_per_epoch", False) or mode == ModeKeys.EVAL:
        predictions = features
        predictions["labels"] = labels
        predictions.update(model.predictions)
        predictions.update(model.metrics_dict)

    return EstimatorSpec(mode=mode, loss=loss, train_op=train_op, predictions=predictions, model=model)

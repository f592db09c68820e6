"""Model plugin base class -- host-side mirror of the reference's NetworksV2/base.py:33-197.

Same contract: `Model(args, name=None)`; `loss = model(inputs, mode, **yml_kwargs)` builds
network -> loss (TRAIN only) -> metrics (if build_metrics) and returns the total loss or None;
callers read `predictions`, `metrics_dict`, `probability`, `_layers["logits"]`, `classes`,
`feed_dict`, `name`.  The reference builds a TF graph once; here every call executes eagerly on
the current HIP stream through boxsegliver_amd.ops (libunetk kernels).
"""
from collections import OrderedDict

import torch

from .. import ops


class ModeKeys(object):
    """tensorflow_estimator ModeKeys values (reference base.py:25)."""
    TRAIN = "train"
    EVAL = "eval"
    PREDICT = "infer"


def _check_size_type(size):
    """base.py:28-31"""
    if size < 0:
        return None
    return size


def _align4(n):
    return (n + 3) // 4 * 4


class ParamStore(object):
    """All variables of one model in three flat fp32 device buffers:
      reg    -- trainable, L2-regularised   (conv / deconv weights, biases unless --bias_decay)
      noreg  -- trainable, not regularised  (norm gamma / beta)
      stats  -- non-trainable moving statistics
    Tensors handed out are 16-byte aligned views, so one optimiser launch / one all-reduce covers
    a whole buffer.  Keys are the reference's TF variable names (UNet.py:203-205)."""

    def __init__(self, specs, device, bias_decay=False):
        self.specs = list(specs)
        self.device = device
        layout = {"reg": [], "noreg": [], "stats": []}
        sizes = {"reg": 0, "noreg": 0, "stats": 0}
        self.where = {}
        for name, shape, kind in self.specs:
            n = 1
            for s in shape:
                n *= s
            if kind in ("conv_w", "deconv_w") or (kind == "bias" and not bias_decay):
                grp = "reg"
            elif kind in ("moving_mean", "moving_var"):
                grp = "stats"
            else:
                grp = "noreg"
            self.where[name] = (grp, sizes[grp], n, tuple(shape), kind)
            layout[grp].append(name)
            sizes[grp] += _align4(n)
        self.flat = {g: torch.zeros(max(sizes[g], 4), dtype=torch.float32, device=device) for g in sizes}
        for g in ("reg", "noreg"):                        # filters that are views of these keep their packed forms (ops._PackCache)
            ops.register_param_buffer(self.flat[g])
        # ONE allocation [noreg | reg] for the gradients: the norm parameters' gradients lie directly in front of the first layers'
        # filter gradients, so the data-parallel path's LAST bucket (distribution_utils.GradBuckets) carries them in the same
        # collective, and zero_grad is one fill
        n_noreg, n_reg = self.flat["noreg"].numel(), self.flat["reg"].numel()
        self.gbuf = torch.zeros(n_noreg + n_reg, dtype=torch.float32, device=device)
        self.grad = {"noreg": self.gbuf[:n_noreg], "reg": self.gbuf[n_noreg:]}
        self.tensors = OrderedDict()
        for name, shape, kind in self.specs:
            grp, off, n, shp, _ = self.where[name]
            t = self.flat[grp][off:off + n].view(shp)
            if grp != "stats":
                t.requires_grad_(True)
                t.grad = self.grad[grp][off:off + n].view(shp)
            self.tensors[name] = t

    def __getitem__(self, name):
        return self.tensors[name]

    def get(self, name, default=None):
        return self.tensors.get(name, default)

    def trainable_names(self):
        return [n for n in self.tensors if self.where[n][0] != "stats"]

    def num_trainable(self):
        return sum(self.where[n][2] for n in self.trainable_names())

    def zero_grad(self):
        ops.new_step(self.grad.values())      # the backward kernels may write each of THIS store's gradient slots in place once per step
        self.gbuf.zero_()
        for name in self.trainable_names():
            grp, off, n, shp, _ = self.where[name]
            t = self.tensors[name]
            if t.grad is None or t.grad.data_ptr() != self.grad[grp].data_ptr() + off * 4:
                t.grad = self.grad[grp][off:off + n].view(shp)

    @torch.no_grad()
    def initialize(self, weight_init="xavier", seed=None):
        """base.py:137-151: xavier = Glorot uniform (slim.xavier_initializer), trunc_norm sigma .01;
        biases 0; BN gamma 1, beta 0, moving_mean 0, moving_variance 1."""
        gen = torch.Generator().manual_seed(1234 if seed is None else seed)
        for name, shape, kind in self.specs:
            t = self.tensors[name]
            if kind in ("conv_w", "deconv_w"):
                if weight_init == "xavier":
                    rf = shape[0] * shape[1]
                    limit = (6.0 / (rf * shape[2] + rf * shape[3])) ** 0.5
                    v = (torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1) * limit
                elif weight_init == "trunc_norm":
                    v = torch.empty(shape, dtype=torch.float64)
                    torch.nn.init.trunc_normal_(v, mean=0.0, std=0.01, a=-0.02, b=0.02, generator=gen)
                else:
                    raise ValueError("Not supported weight initializer: " + str(weight_init))
                t.copy_(v.to(torch.float32))
            elif kind == "fc_w":        # slim.fully_connected default: Glorot uniform, [in, out], no regulariser
                limit = (6.0 / (shape[0] + shape[1])) ** 0.5
                t.copy_(((torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1) * limit).to(torch.float32))
            elif kind == "conv1d_w":    # slim.conv1d default: Glorot uniform on [k, in, out], no regulariser (slim_nets.py:69 ...)
                limit = (6.0 / (shape[0] * shape[1] + shape[0] * shape[2])) ** 0.5
                t.copy_(((torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1) * limit).to(torch.float32))
            elif kind == "fc_b_one":    # final_biases_initializer=tf.ones_initializer() of the vgg context models (GUNet.py:74)
                t.fill_(1.0)
            elif kind == "fc_w_he":     # tf.keras.initializers.he_normal: truncated normal, var 2/fan_in (GUNet.py:59)
                std = (2.0 / shape[0]) ** 0.5 / 0.87962566103423978
                v = torch.empty(shape, dtype=torch.float64)
                torch.nn.init.trunc_normal_(v, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=gen)
                t.copy_(v.to(torch.float32))
            elif kind in ("gamma", "moving_var"):
                t.fill_(1.0)
            else:
                t.zero_()

    @torch.no_grad()
    def load_state(self, state, strict=True):
        for name in self.tensors:
            if name in state:
                v = state[name]
                v = v if torch.is_tensor(v) else torch.as_tensor(v)
                self.tensors[name].copy_(v.to(torch.float32).reshape(self.tensors[name].shape))
            elif strict:
                raise KeyError("missing variable " + name)

    def state_dict(self):
        return OrderedDict((n, t.detach().cpu().clone()) for n, t in self.tensors.items())

    # optimiser slots (Adam m / v, Momentum accumulator) are flat buffers laid out like self.flat[grp]; checkpoints speak
    # per-variable tensors in the TF shapes ("Optimizer/<variable>/Adam", core/estimator.py)
    def read_slot(self, flat, name):
        """The slice of `flat` (a buffer mirroring self.flat[group of name]) that belongs to variable `name`, TF shape."""
        _, off, n, shp, _ = self.where[name]
        return flat[off:off + n].view(shp).detach().cpu().clone()

    @torch.no_grad()
    def write_slot(self, flat, name, value):
        _, off, n, shp, _ = self.where[name]
        v = value if torch.is_tensor(value) else torch.as_tensor(value)
        flat[off:off + n].view(shp).copy_(v.to(torch.float32).reshape(shp))


class BaseNet(object):
    def __init__(self, args):
        self._name = "Base"
        self._mode = ModeKeys.TRAIN
        self._args = args
        self._inputs = {}
        self._layers = {}
        self._image_summaries = {}
        self.classes = ["Background"]
        self.metrics_dict = {}
        self.predictions = {}
        self.key_collections = {}
        self._is_training = False
        self._feed_dict = {}
        self.ret_prob = False
        self.ret_pred = False
        self.probability = None
        self.params = None          # ParamStore, created at first call
        self.loss_terms = {}

    # --- properties mirrored from base.py:54-110
    @property
    def name(self):
        return self._name

    @name.setter
    def name(self, new_name):
        if new_name and isinstance(new_name, str):
            self._name = new_name

    @property
    def mode(self):
        return self._mode

    @mode.setter
    def mode(self, new_mode):
        if new_mode in [ModeKeys.TRAIN, ModeKeys.EVAL, ModeKeys.PREDICT]:
            self._mode = new_mode
            # base.py:77-78: is_training is a placeholder defaulting to False that the TRAIN loop
            # feeds True; eagerly that is simply "mode == TRAIN"
            self._is_training = (new_mode == ModeKeys.TRAIN)
            self._feed_dict["is_training"] = self._is_training

    @property
    def is_training(self):
        return self._is_training

    @property
    def args(self):
        return self._args

    @property
    def compute_bf16(self):
        """--compute_dtype (not a reference flag; BASELINE.json configs[2]) as the precision value of include/unetk.h:
        0 `fp32` (exact fp32 MFMA); 1 `bf16c`: the contractions round their operands to bf16 for the bf16 matrix cores,
        tensors in HBM stay fp32 (round 1's mode); 2 `bf16`: bf16 matrix cores AND bf16 storage of activations and
        activation gradients -- configs[2]'s "bf16 activations / weights-compute, fp32 master / accum / stats".
        Statistics, master weights, weight gradients and the optimiser are fp32 in every mode."""
        name = str(getattr(self._args, "compute_dtype", "fp32") or "fp32").lower()
        if name in ("bf16", "bfloat16", "bf16s"):
            return 2
        if name in ("bf16c", "bf16_compute"):
            return 1
        if name in ("fp32", "float32"):
            return 0
        raise ValueError("unknown --compute_dtype {!r} (fp32 | bf16 | bf16c)".format(name))

    @property
    def storage_dtype(self):
        """dtype of the activation tensors the network allocates (concat buffers)."""
        import torch
        return torch.bfloat16 if self.compute_bf16 == 2 else torch.float32

    @property
    def num_classes(self):
        return len(self.classes)

    @property
    def layers(self):
        return self._layers

    @property
    def feed_dict(self):
        return self._feed_dict

    @property
    def metrics(self):
        return dict(self.metrics_dict)

    def _net_arg_scope(self, *args, **kwargs):
        raise NotImplementedError

    def _build_summaries(self):
        raise NotImplementedError

    def _build_network(self, *args, **kwargs):
        raise NotImplementedError

    def _build_loss(self):
        raise NotImplementedError

    def _build_metrics(self):
        raise NotImplementedError

    # --- base.py:128-178
    def _get_regularizer(self):
        """Returns (weight l2 scale, bias l2 scale); None = unregularised.  Literal reading of
        base.py:131: biases share the weight regulariser UNLESS --bias_decay is passed."""
        wd = getattr(self.args, "weight_decay_rate", 0) or 0
        if wd > 0:
            w_reg = wd
            b_reg = None if getattr(self.args, "bias_decay", False) else w_reg
        else:
            w_reg, b_reg = None, None
        return w_reg, b_reg

    def _get_initializer(self):
        if self.args.weight_init not in ("trunc_norm", "xavier"):
            raise ValueError("Not supported weight initializer: " + str(self.args.weight_init))
        return self.args.weight_init, "zeros"

    def _get_normalization(self, freeze=None):
        if self.args.normalizer == "batch_norm":
            params = {"scale": True, "eps": 1e-3, "decay": 0.999}
            if freeze is None:
                params["is_training"] = self.is_training
            elif not freeze:
                params["is_training"] = True
            else:
                params.update({"is_training": False, "trainable": False})
            return "batch_norm", params
        elif self.args.normalizer == "instance_norm":
            return "instance_norm", {"eps": 1e-6}
        raise ValueError("Not supported normalization function: " + str(self.args.normalizer))

    def _get_weights_params(self):
        w_params = {"tag": self.args.tag}
        if self.args.loss_weight_type == "numerical":
            w_params["numeric_w"] = self.args.loss_numeric_w
        elif self.args.loss_weight_type == "proportion":
            if self.args.loss_proportion_decay > 0:
                w_params["proportion_decay"] = self.args.loss_proportion_decay
        return w_params

    def __call__(self, inputs, mode, *args, **kwargs):
        """base.py:180-197"""
        self._inputs = inputs
        self.mode = mode
        self.metrics_dict = {}
        self.predictions = {}
        self._net_arg_scope()
        self._build_network(*args, **kwargs)
        ret = None
        if self.mode == ModeKeys.TRAIN:
            ret = self._build_loss()
        if kwargs.get("build_metrics", False):
            self._build_metrics()
        if kwargs.get("build_summaries", False):
            self._build_summaries()
        return ret

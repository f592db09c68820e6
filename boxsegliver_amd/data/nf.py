"""The neurofibroma (NF) sub-commands of the reference's entry points (`nf`, `nf2`, `nf_inter`, `nf_3d`).

The NF MRI dataset is private and its pipelines (DataLoader/NF/*.py, ~4900 lines of tf.data / GeodisTK code) are outside
this package's scope: what matters to the hot path is the tensor contract -- images [bs, (D,) H, W, C]
z-scored floats, labels int32 {0, 1}, optional `sp_guide` [bs, H, W, guide_channel], `context` [bs, L].  The sub-commands
therefore parse the reference's flag groups verbatim (data/flagsets.py) and are served by the synthetic generators of
data/synthetic.py with that contract; pointing them at real NF data raises."""
from . import flagsets, synthetic


def add_arguments_for(pipeline):
    def add_arguments(parser):
        flagsets.add_arguments(parser, pipeline)
        group = parser.add_argument_group(title="Synthetic data (this package)")
        group.add_argument("--nf_root", type=str, default=None, help="NF dataset root (private data: not supported)")
        group.add_argument("--synthetic_batches", type=int, default=2, help="distinct synthetic batches kept on device")
        group.add_argument("--seed", type=int, default=1234)
    return add_arguments


def input_fn(mode, params):
    args = params["args"]
    if getattr(args, "nf_root", None):
        raise NotImplementedError("the NF dataset pipelines are out of scope (private data, SURVEY.md 2 row 19); without "
                                  "--nf_root the sub-command runs on synthetic tensors with the NF contract")
    if getattr(args, "model", "") == "UNet3D":
        return synthetic.input_fn_3d(mode, params)
    return synthetic.input_fn(mode, params)

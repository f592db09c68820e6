"""The input-pipeline flag groups of the reference, as DATA.

The reference's first positional argument selects an input pipeline module whose `add_arguments` contributes a flag group
(entry/main.py:53-80, entry/main_g.py:55-76).  The flag names, types and defaults are part of the drop-in surface (the
shipped run scripts pass them verbatim), so they are kept exactly; what is not kept is nine near-identical argparse
functions -- one table of flag definitions and one table saying which pipeline uses which flags.

Sources (reference repo): DataLoader/Liver/input_pipeline.py:54-70, input_pipeline_li.py:53-65, input_pipeline_g.py:71-125;
DataLoader/NF/input_pipeline.py, input_pipeline_g.py, input_pipeline_g_simply.py, input_pipeline_iin.py, input_pipeline_3d.py.
"""
import math

S = "store_true"
# flag -> argparse keyword arguments (the same for every pipeline unless PIPELINES overrides the default)
FLAGS = {
    "--test_fold": dict(type=int, default=2),
    "--im_depth": dict(type=int, default=10),
    "--im_height": dict(type=int, default=256),
    "--im_width": dict(type=int, default=256),
    "--im_channel": dict(type=int, default=3),
    "--filter_size": dict(type=int, default=0, help="Filter tumors small than the given size"),
    "--noise_scale": dict(type=float, default=0.1),
    "--zoom_scale": dict(type=float, nargs=2, default=(1.0, 1.4)),
    "--random_flip": dict(type=int, default=1, help="Random flip while training. 0: no flip, 1: flip left/right, 2: flip "
                                                     "up/down, 3: both (3-D pipelines: bit 2 flips the slices)"),
    "--eval_in_patches": dict(action=S),
    "--eval_num_batches_per_epoch": dict(type=int, default=100),
    "--eval_mirror": dict(action=S),
    "--liver_percent": dict(type=float, default=0.66),
    "--tumor_percent": dict(type=float, default=0.5),
    "--downsampling": dict(flags=("-ds", "--downsampling"), action=S),
    "--side_dropout": dict(type=float, default=0.5, help="Dropout used in the context sub-network"),
    "--use_context": dict(action=S),
    "--context_list": dict(type=str, nargs="+", help="Paired context information: name length [name length ...]"),
    "--hist_noise": dict(action=S),
    "--hist_noise_scale": dict(type=float, default=0.002),
    "--hist_scale": dict(type=float, default=20),
    "--glcm": dict(action=S),
    "--glcm_features": dict(type=str, nargs="+"),
    "--glcm_distance": dict(type=int, nargs="+", default=[1, 2, 3]),
    "--glcm_angle": dict(type=float, nargs="+", default=[0., math.pi * 0.25, math.pi * 0.5, math.pi * 0.75]),
    "--glcm_noise": dict(action=S),
    "--use_zscore": dict(action=S),
    "--use_gamma": dict(action=S),
    "--gamma_range": dict(type=float, nargs="+", default=[0.7, 1.5]),
    "--use_spatial": dict(action=S),
    "--local_enhance": dict(action=S),
    "--geodesic": dict(action=S),
    "--spatial_random": dict(type=float, default=1.),
    "--spatial_inner_random": dict(action=S),
    "--center_random_ratio": dict(type=float, default=0.2),
    "--stddev_random_ratio": dict(type=float, default=0.4),
    "--eval_no_sp": dict(action=S),
    "--min_std": dict(type=float, default=2.),
    "--stddev": dict(type=float, default=3.),
    "--save_sp_guide": dict(action=S),
    "--use_se": dict(action=S),
    "--eval_discount": dict(type=float, default=0.85),
    "--eval_no_p": dict(action=S),
    "--real_sp": dict(type=str),
    "--guide_scale": dict(type=float, default=5.),
    "--guide_channel": dict(type=int, default=2),
    "--fp_sample": dict(action=S),
    "--sample_neg": dict(type=float, default=0.),
    "--fp_version": dict(type=int, default=1, choices=[1, 2]),
    "--use_cascade": dict(action=S),
    "--cascade_binary": dict(action=S),
    "--use_2d": dict(action=S),
    "--model_2d": dict(type=str),
    "--model_2d_config": dict(type=str),
    "--ckpt_2d": dict(type=str),
    "--ct_conv": dict(default=1),
    "--case_id": dict(type=int),
    "--pos": dict(type=int, nargs="+"),
    "--ct_base": dict(type=int, default=[32], nargs="+"),
}

_COMMON = ["--test_fold", "--im_height", "--im_width", "--im_channel"]
_CTX = ["--side_dropout", "--use_context", "--context_list", "--hist_noise", "--hist_noise_scale", "--hist_scale", "--glcm"]
_SP = ["--use_spatial", "--spatial_random", "--spatial_inner_random", "--center_random_ratio", "--stddev_random_ratio", "--eval_no_sp"]
# pipeline -> (flags in the reference's order, {flag: default override})
PIPELINES = {
    "liver": (_COMMON + ["--filter_size", "--noise_scale", "--zoom_scale", "--random_flip", "--eval_in_patches",
                         "--eval_num_batches_per_epoch", "--eval_mirror", "--liver_percent", "--tumor_percent"], {}),
    "liver_li": (_COMMON + ["--noise_scale", "--zoom_scale", "--random_flip", "--eval_num_batches_per_epoch", "--eval_mirror"], {}),
    "liver_g": (_COMMON + ["--filter_size", "--noise_scale", "--zoom_scale", "--random_flip", "--eval_in_patches",
                           "--eval_num_batches_per_epoch", "--eval_mirror"] + _CTX +
                ["--glcm_features", "--glcm_distance", "--glcm_angle", "--glcm_noise"] + _SP +
                ["--min_std", "--save_sp_guide", "--use_se", "--eval_discount", "--real_sp"], {}),
    "nf": (_COMMON + ["--filter_size", "--noise_scale", "--zoom_scale", "--random_flip", "--eval_in_patches",
                      "--eval_num_batches_per_epoch", "--eval_mirror"], {"--zoom_scale": (1.0, 1.25)}),
    "nf_g": (_COMMON + ["--filter_size", "--noise_scale", "--zoom_scale", "--random_flip", "--eval_in_patches",
                        "--eval_num_batches_per_epoch", "--eval_mirror"] + _CTX + ["--glcm_noise", "--use_zscore", "--use_gamma"] +
             _SP + ["--min_std", "--save_sp_guide", "--use_se", "--eval_discount", "--eval_no_p", "--real_sp", "--guide_scale"],
             {"--zoom_scale": (1.0, 1.25)}),
    "nf_g_simply": (_COMMON + ["--noise_scale", "--zoom_scale", "--random_flip", "--eval_in_patches", "--eval_num_batches_per_epoch",
                               "--eval_mirror", "--tumor_percent", "--downsampling"] + _CTX +
                    ["--glcm_noise", "--use_zscore", "--use_gamma", "--use_spatial", "--local_enhance", "--geodesic",
                     "--spatial_random", "--spatial_inner_random", "--center_random_ratio", "--stddev_random_ratio", "--eval_no_sp",
                     "--stddev", "--save_sp_guide", "--use_se", "--eval_discount", "--eval_no_p", "--real_sp", "--guide_scale",
                     "--guide_channel", "--fp_sample", "--sample_neg", "--fp_version"], {"--zoom_scale": (1.0, 1.25)}),
    "nf_iin": (_COMMON + ["--filter_size", "--noise_scale", "--zoom_scale", "--random_flip", "--use_zscore", "--use_gamma",
                          "--gamma_range", "--eval_in_patches", "--eval_num_batches_per_epoch", "--eval_mirror", "--side_dropout",
                          "--use_context", "--use_spatial", "--spatial_random", "--eval_no_sp", "--min_std", "--save_sp_guide",
                          "--use_se", "--eval_discount", "--eval_no_p", "--real_sp", "--guide_scale", "--ct_conv", "--case_id",
                          "--pos", "--ct_base"], {"--zoom_scale": (1.0, 1.25)}),
    "nf_3d": (["--test_fold", "--im_depth", "--im_height", "--im_width", "--im_channel", "--zoom_scale", "--random_flip",
               "--eval_in_patches", "--eval_num_batches_per_epoch", "--eval_mirror", "--tumor_percent", "--use_spatial",
               "--local_enhance", "--eval_no_sp", "--stddev", "--save_sp_guide", "--eval_no_p", "--guide_channel", "--fp_sample",
               "--sample_neg", "--use_cascade", "--cascade_binary", "--use_2d", "--downsampling", "--model_2d",
               "--model_2d_config", "--ckpt_2d"],
              {"--im_channel": 1, "--zoom_scale": (1.0, 1.25), "--stddev": [1, 3., 3.]}),
}


def add_arguments(parser, pipeline):
    """The flag group the reference's `<pipeline module>.add_arguments(parser)` contributes."""
    names, overrides = PIPELINES[pipeline]
    group = parser.add_argument_group(title="Input Pipeline Arguments")
    for name in names:
        kw = dict(FLAGS[name])
        flags = kw.pop("flags", (name,))
        if name in overrides:
            kw["default"] = overrides[name]
            if name == "--stddev" and isinstance(overrides[name], list):
                kw["nargs"] = "+"
        group.add_argument(*flags, **kw)
    return group

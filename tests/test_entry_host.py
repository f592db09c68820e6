"""CPU: the command-line entry points (boxsegliver_amd.entry.main / main_g; reference entry/main.py:45-85,122-208 and
entry/main_g.py): sub-command dispatch, the merged flag surface, and -- where the reference tree is present (build container)
-- the flag lists of its shipped run scripts parsed VERBATIM."""
import os
import re
import shlex

import pytest

from boxsegliver_amd.entry import main as entry

REF = "/root/reference"


def _script_invocations(path):
    """[(entry script, [argv after it])] of every `python ./entry/<x>.py ...` command in a reference run script, with the
    shell substitutions the scripts use resolved (tag from the script name, one GPU)."""
    text = open(path).read()
    base = os.path.basename(path)[:-3]
    out = []
    for m in re.finditer(r"python\s+\./entry/(\w+)\.py\s+(.*?)\$@", text, re.S):
        body = m.group(2).replace("\\\n", " ")
        body = body.replace('${BASE_NAME%".sh"}', base).replace("${#GPU_IDS[@]}", "1")
        out.append((m.group(1), shlex.split(body)))
    return out


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree (build container only)")
@pytest.mark.parametrize("script,expect", [
    ("run_scripts/template/001_unet.sh", dict(model="UNet", classes=["Liver", "Tumor"], batch_size=8, learning_policy="plateau",
                                              loss_numeric_w=[0.2, 0.4, 4.4], primary_metric="Tumor/Dice")),
    ("scripts/102_gnet_v1.sh", dict(model="GUNet", classes=["NF"], batch_size=16, normalizer="instance_norm", use_spatial=True,
                                    distribution_strategy="mirrored", summary_prefix="nf")),
    ("threed_script/201_unet_v1.sh", dict(model="UNet3D", classes=["NF"], im_depth=10, batch_size=4, tumor_percent=0.75,
                                          random_flip=7, log_step=125)),
])
def test_reference_run_scripts_parse_verbatim(script, expect):
    calls = _script_invocations(os.path.join(REF, script))
    assert calls, script
    parsed = 0
    for which, argv in calls:
        if which not in ("main", "main_g"):
            continue                                           # main_eval_3d.py etc.: NF research drivers, out of scope
        args, sub, _ = entry.get_arguments(argv, guided=(which == "main_g"))
        parsed += 1
        assert args.tag == os.path.basename(script)[:-3] and sub == argv[0]
        if args.mode == "train":
            for k, v in expect.items():
                assert getattr(args, k) == v, (k, getattr(args, k), v)
            assert args.model_dir.endswith("model_dir/" + args.tag)
    assert parsed >= 1


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree (build container only)")
def test_every_shipped_script_that_uses_main_or_main_g_parses():
    n = 0
    for d in ("run_scripts/template", "scripts", "threed_script"):
        for name in sorted(os.listdir(os.path.join(REF, d))):
            if not name.endswith(".sh"):
                continue
            for which, argv in _script_invocations(os.path.join(REF, d, name)):
                if which in ("main", "main_g") and argv and argv[0] in entry.SUBCOMMANDS[which == "main_g"]:
                    entry.get_arguments(argv, guided=(which == "main_g"))
                    n += 1
    assert n >= 20


def test_subcommand_dispatch_and_flag_groups():
    args, sub, pipe = entry.get_arguments(
        "liver --mode train --tag t1 --model UNet --classes Liver Tumor --loss_weight_type numerical "
        "--loss_numeric_w 0.2 0.4 4.4 --eval_per_epoch --evaluator Volume --save_best --test_fold 2".split())
    assert sub == "liver" and args.liver_percent == 0.66 and args.tumor_percent == 0.5 and args.zoom_scale == (1.0, 1.4)
    assert args.lits_root == "data/LiTS" and args.evaluator == "Volume" and args.eval_num == -1
    assert args.learning_rate == 1e-3 and args.optimizer == "Adam" and args.weight_decay_rate == 1e-5      # solver / loss groups
    a2, _, _ = entry.get_arguments("only_liver --mode eval --tag t --model UNet --classes Liver".split())
    assert not hasattr(a2, "liver_percent") and not hasattr(a2, "filter_size")          # input_pipeline_li.py:53-65
    a3, _, _ = entry.get_arguments("nf_3d --mode train --tag t --model UNet3D --classes NF --stddev 1 3 3".split())
    assert a3.im_channel == 1 and a3.im_depth == 10 and a3.guide_channel == 2 and a3.stddev == [1.0, 3.0, 3.0]
    g, sub, _ = entry.get_arguments("nf2 --mode train --tag t --model GUNet --classes NF --ct_base 16 32".split(), guided=True)
    assert sub == "nf2" and g.ct_base == [16, 32] and g.gamma_range == [0.7, 1.5]
    with pytest.raises(ValueError):
        entry.get_arguments(["nf_3d", "--mode", "train"], guided=True)                 # not a main_g sub-command
    with pytest.raises(ValueError):
        entry.get_arguments(["kidney"])
    with pytest.raises(ValueError):
        entry.get_arguments([])
    with pytest.raises(SystemExit):                                                     # a required flag is missing
        entry.get_arguments("liver --mode train --model UNet --classes Liver".split())

"""Every model configuration the package ships resolves through the reference's lookup rule (core/models.py:92-118:
<model>.yml, then NetworksV2/, then NetworksV2/ext_config/) and names a model of the registry; the GUNet variants only
use options the host mirror implements (context_model "fc" or the 1-D VGG trunks of GUNet_DE_VGG16{B,D}.yml)."""
import argparse
from pathlib import Path

import pytest

from boxsegliver_amd.core import models

NETS = Path(models.__file__).parent.parent / "NetworksV2"
CONFIGS = sorted(p.name for p in list(NETS.glob("*.yml")) + list((NETS / "ext_config").glob("*.yml")))


def test_registry_matches_reference_names():
    names = [cls.__name__ for cls in models.MODEL_ZOO]
    assert names == ["UNet", "GUNet", "UNetInter", "LGNet", "UNet3D", "SmallUNet", "InterUNet"]   # reference models.py:36-38
    parser = argparse.ArgumentParser()
    models.add_arguments(parser)
    ns = parser.parse_args(["--model", "UNetInter", "--classes", "Liver", "Tumor"])
    assert ns.model == "UNetInter" and ns.classes == ["Liver", "Tumor"] and ns.batch_size == 8
    with pytest.raises(SystemExit):
        parser.parse_args(["--model", "DenseUNet", "--classes", "Liver"])


@pytest.mark.parametrize("cfg", CONFIGS)
def test_shipped_config_resolves(cfg):
    model = cfg.split(".")[0].split("_")[0].replace("GUNetV2", "GUNet")
    args = argparse.Namespace(model=model, model_config=cfg)
    params = models.get_model_params(args, build_metrics=True)
    kw = params["model_kwargs"]
    assert params["model"].__name__ == model and kw["build_metrics"] is True and kw["ret_pred"] is True
    if model == "LGNet":
        assert len(kw["mod_layers"]) == 2 and all(br == sorted(br) for br in kw["mod_layers"])
    if model in ("GUNet",):
        want = {"GUNet_DE_VGG16B.yml": "vgg16B", "GUNet_DE_VGG16D.yml": "vgg16D"}.get(cfg, "fc")
        assert kw["context_model"] == want and kw["mod_layers"] == [1, 2, 3, 4] and len(kw["context_fc_channels"]) == 2
        if want != "fc":
            assert kw["context_conv_init_channels"] == 2
    if model == "UNet3D":
        assert kw["num_pool_layers"] in (4, 5) and kw["init_channels"] == 30 and kw["max_channels"] == 320


def test_default_config_name_and_missing_config():
    args = argparse.Namespace(model="UNet", model_config=None)
    params = models.get_model_params(args)
    assert args.model_config == "UNet.yml" and params["model_kwargs"]["init_channels"] == 64
    args = argparse.Namespace(model="UNet", model_config="does_not_exist.yml")
    assert models.get_model_params(args)["model_kwargs"] == {"build_metrics": False, "build_summaries": False}
    with pytest.raises(NameError):
        models.get_model_params(argparse.Namespace(model="DenseUNet", model_config=None))

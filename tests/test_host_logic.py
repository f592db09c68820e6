"""CPU tests: the C-ABI library loads and exports every symbol include/unetk.h declares (no compute
calls without a GPU), argument validation, and the host-side mirror of the reference interface
(flags, registry, LR policies, parameter store, data contract)."""
import argparse
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from boxsegliver_amd import _abi, config, loss_metrics
from boxsegliver_amd.NetworksV2 import UNet as unet_mod
from boxsegliver_amd.NetworksV2.base import ParamStore
from boxsegliver_amd.core import models, solver
from boxsegliver_amd.data import synthetic
from boxsegliver_amd.utils import distribution_utils
from oracle import solver as osolver
from oracle import unet2d

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_all_exported_and_bound():
    hdr = open(os.path.join(ROOT, "include", "unetk.h")).read()
    declared = set(re.findall(r"\b(unetk_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"unetk_conv_desc", "unetk_deconv_desc", "unetk_head_desc", "unetk_norm_desc", "unetk_conv3d_desc", "unetk_deconv3d_desc"}
    lib = _abi.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_abi.EXPORTED_SYMBOLS)
    assert lib.unetk_abi_version() == _abi.ABI_VERSION == 10


def test_abi_argument_validation_without_gpu():
    lib = _abi.lib()
    d = _abi.ConvDesc(2, 8, 16, 64, 64, 64, 64)
    assert lib.unetk_conv3x3_fwd(ctypes.byref(d), None, None, None, None, None) == -1       # UNETK_E_BADARG
    assert lib.unetk_conv3x3_stat_rows(ctypes.byref(d)) == 2
    bad = _abi.ConvDesc(2, 8, 16, 64, 64, 32, 64)                                            # x_stride < Cin
    assert lib.unetk_conv3x3_stat_rows(ctypes.byref(bad)) == -1
    odd = _abi.ConvDesc(2, 8, 16, 30, 60, 30, 60)                                            # UNet3D-like channels
    assert lib.unetk_conv3x3_wgrad_ws_bytes(ctypes.byref(odd)) == 0
    assert b"workspace" in lib.unetk_error_string(-3)
    hd = _abi.HeadDesc()
    hd.N, hd.HW, hd.C, hd.ncls = 2, 64, 64, 9                                                # > UNETK_MAX_CLASSES
    assert lib.unetk_head_ws_bytes(ctypes.byref(hd)) == 0


def test_ops_refuse_cpu_tensors():
    from boxsegliver_amd import ops
    with pytest.raises(_abi.UnetkError):
        ops.conv3x3_pack(torch.zeros(3, 3, 16, 64))


def _parser():
    p = argparse.ArgumentParser()
    config.add_arguments(p)
    models.add_arguments(p)
    solver.add_arguments(p)
    loss_metrics.add_arguments(p)
    synthetic.add_arguments(p)
    return p


def test_flag_surface_matches_reference_run_script():
    # flags of run_scripts/template/001_unet.sh:11-36 that belong to these groups
    argv = ("--mode train --tag 001_unet --model UNet --classes Liver Tumor --im_height 256 --im_width 256 "
            "--im_channel 3 --noise_scale 0.05 --random_flip 3 --num_of_total_steps 600000 "
            "--loss_weight_type numerical --loss_numeric_w 0.2 0.4 4.4 --batches_per_epoch 2000 --batch_size 8 "
            "--weight_decay_rate 0.000001 --learning_policy plateau --learning_rate 0.001 --lr_end 0 "
            "--lr_decay_rate 0.2 --eval_per_epoch --save_best").split()
    p = _parser()
    args = p.parse_args(argv)
    config.check_args(args, p)
    config.fill_default_args(args)
    assert args.model_dir.endswith(os.path.join("model_dir", "001_unet"))
    assert args.normalizer == "batch_norm" and args.weight_init == "xavier" and args.optimizer == "Adam"
    assert args.log_step == 500 and args.distribution_strategy == "off" and args.num_gpus == 1
    assert args.loss_type == "xentropy" and args.metrics_train == ["Dice"] and args.summary_prefix == "001_unet"
    with pytest.raises(SystemExit):
        bad = p.parse_args(argv + ["--loss_numeric_w", "1", "2"])
        config.check_args(bad, p)


def test_model_registry_and_yml():
    p = _parser()
    args = p.parse_args("--mode train --tag t --model UNet --classes Liver".split())
    params = models.get_model_params(args, build_metrics=True)
    assert params["model"].__name__ == "UNet" and args.model_config == "UNet.yml"
    assert params["model_kwargs"] == {"init_channels": 64, "num_down_samples": 4, "ret_prob": False, "ret_pred": True,
                                      "build_metrics": True, "build_summaries": False}
    m = params["model"](args)
    assert m.classes == ["Background", "Liver"] and m.num_classes == 2 and m.name == "UNet"
    with pytest.raises(SystemExit):
        p.parse_args("--mode train --tag t --model NoSuchNet --classes Liver".split())


def test_param_specs_match_oracle_and_counts():
    a = unet_mod.param_specs(3, 3, 64, 4, "batch_norm", False, "UNet")
    b = unet2d.param_specs(3, 3)
    assert [(n, tuple(s), k) for n, s, k in a] == [(n, tuple(s), k) for n, s, k in b]
    store = ParamStore(unet_mod.param_specs(3, 3, 8, 2, "batch_norm", False, "UNet"), torch.device("cpu"))
    store.initialize("xavier", seed=1)
    assert store.num_trainable() == sum(int(np.prod(s)) for _, s, k in store.specs if k in unet2d.TRAINABLE_KINDS)
    w = store["UNet/Encode1/Repeat/convolution2d_1/weights"]
    assert w.shape == (3, 3, 3, 8) and w.data_ptr() % 16 == 0
    assert float(w.abs().max()) <= (6.0 / (27 + 72)) ** 0.5 + 1e-7
    assert torch.all(store["UNet/Encode1/Repeat/convolution2d_1/BatchNorm/moving_variance"] == 1)
    # gradient views alias the flat gradient buffer; biases are regularised unless --bias_decay
    store.zero_grad()
    w.grad += 1.0
    assert store.grad["reg"].sum().item() == w.numel()
    assert store.where["UNet/AdjustChannels/biases"][0] == "reg"
    store2 = ParamStore(store.specs, torch.device("cpu"), bias_decay=True)
    assert store2.where["UNet/AdjustChannels/biases"][0] == "noreg"
    sd = store.state_dict()
    store2.load_state(sd)
    assert torch.equal(store2[store.specs[0][0]], store[store.specs[0][0]])


def test_padded_store_hands_out_live_channel_masks():
    """PaddedParamStore marks, per 8-channel group, where a padded UNet3D filter holds real channels (the conv kernels skip
    the rest of their contraction axis): 240 in 256, 120 in 128, and the two padded halves of a concat input."""
    from boxsegliver_amd.NetworksV2 import UNet3D as u3
    from boxsegliver_amd.NetworksV2.padded import PaddedParamStore
    specs, pads = u3.param_specs(1, 3, 30, 4, 320, "instance_norm", "UNet3D")
    store = PaddedParamStore(specs, pads, torch.device("cpu"))
    m = store["UNet3D/conv_e3/conv2/weights"].unetk_live8            # 240 -> 240 in 256 x 256
    assert m == ((1 << 30) - 1, (1 << 30) - 1)
    m = store["UNet3D/conv_d3/conv1/weights"].unetk_live8            # concat(240, 240) in 512
    assert m[0] == ((1 << 30) - 1) | (((1 << 30) - 1) << 32)
    m = store["UNet3D/conv_d2/conv1/weights"].unetk_live8            # concat(120, 120) in 256 -> 120 in 128
    assert m == (((1 << 15) - 1) | (((1 << 15) - 1) << 16), (1 << 15) - 1)
    m = store["UNet3D/conv_e0/conv2/weights"].unetk_live8            # 30 in 32: every group holds a real channel
    assert m == (0xF, 0xF)
    assert not hasattr(store["UNet3D/conv_d3/up/weights"], "unetk_live8")
    # every physical entry outside the mask's groups is zero after initialisation (the promise behind the skip)
    store.initialize("xavier", seed=0)
    w = store["UNet3D/conv_d3/conv1/weights"]
    dead = [c for c in range(512) if not (w.unetk_live8[0] >> (c // 8)) & 1]
    assert dead == list(range(240, 256)) + list(range(496, 512)) and float(w.detach()[:, :, :, dead].abs().max()) == 0.0


def test_thread_pools_follow_the_cpus_the_process_may_use(tmp_path):
    """boxsegliver_amd/utils/hostcpu.py: the pool size is the minimum of the host's count, the affinity mask and the cgroup quota,
    the ranks of a node share it, and a value the user exported wins."""
    from boxsegliver_amd.utils import hostcpu
    n = hostcpu.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    env = {}
    assert hostcpu.size_thread_pools(env) == n and env == {"OMP_NUM_THREADS": str(n), "MKL_NUM_THREADS": str(n)}
    env = {"LOCAL_WORLD_SIZE": "4"}
    assert hostcpu.size_thread_pools(env) == max(1, n // 4) and env["OMP_NUM_THREADS"] == str(max(1, n // 4))
    env = {"OMP_NUM_THREADS": "3"}
    hostcpu.size_thread_pools(env)
    assert env["OMP_NUM_THREADS"] == "3"
    # the test session itself runs sized (tests/conftest.py)
    assert torch.get_num_threads() <= n


def test_reciprocal_division_of_the_stacked_plane_filter_gradient_is_exact():
    """csrc/common.h unetk_fdiv (round 5): floor(n / d) from a float32 reciprocal -- estimate, one multiply-subtract, one correction --
    restated in numpy and compared with integer division over the whole range the kernel uses (n < 2^20 virtual rows / planes by
    the plan's guard; divisors H + 1, planes per sample, planes per address group, and the 2^30 of a dense address map)."""
    n = np.arange(0, 1 << 20, dtype=np.int64)
    for d in (1, 2, 3, 7, 13, 25, 33, 49, 65, 96, 97, 192, 1000, 4097, 1 << 19, (1 << 20) - 1, 1 << 30):
        rcp = np.float32(1.0) / np.float32(d)
        q = (n.astype(np.float32) * rcp).astype(np.int64)                  # (int)((float)n * rcp): truncation
        r = n - q * d
        q = q + (r >= d).astype(np.int64) - (r < 0).astype(np.int64)
        assert np.array_equal(q, n // d), d


def test_halo_row_stride_of_the_tiled_conv_is_free_of_lds_bank_conflicts():
    """csrc/conv_igemm.hip halo_row_f (round 5).  A ds_read_b128 is served in four fixed 16-lane groups, each lane taking four
    consecutive banks of 64 (MI355X_MICROARCH.md, LDS).  The A fragment's lane l reads pixel (row l >> 4, column l & 15) of a
    two-row patch at row_stride * row + 20 * column floats: with the unpadded row (18 pixels x 20 floats = 360) two windows of a
    group coincide -- the 35 % conflict share rocprofv3 counted -- with rows padded to 384 floats none do."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]

    def cycles(row_stride, ps=20):
        worst = 0
        for g in groups:
            banks = {}
            for l in g:
                off = (l >> 4) * row_stride + (l & 15) * ps
                for b in range(4):
                    banks.setdefault((off + b) % 64, set()).add(off)
            worst = max(worst, max(len(v) for v in banks.values()))
        return worst

    assert cycles(18 * 20) == 2                      # rounds 1-4
    padded = (18 * 20 + 63) // 64 * 64
    assert padded == 384 and cycles(padded) == 1
    # the B fragment (lane l at 4 l floats) never conflicted
    for g in groups:
        assert len({(4 * l + b) % 64 for l in g for b in range(4)}) == 64


def test_solver_lr_policies_match_oracle():
    p = _parser()
    for extra, kw in [
        ("--learning_policy period_step --lr_decay_step 10 --lr_decay_rate 0.5",
         dict(policy="period_step", decay_step=10, decay_rate=0.5)),
        ("--learning_policy custom_step --lr_decay_boundaries 5 9 --lr_custom_values 1.0 0.1 0.01",
         dict(policy="custom_step", boundaries=[5, 9], values=[1.0, 0.1, 0.01])),
        ("--learning_policy poly --num_of_total_steps 20 --lr_power 0.9 --lr_end 1e-6",
         dict(policy="poly", total_steps=20, power=0.9, end_lr=1e-6)),
    ]:
        args = p.parse_args(("--mode train --tag t --model UNet --classes Liver " + extra).split())
        s = solver.Solver(args)
        for gs in (0, 4, 5, 6, 9, 10, 19, 20, 25):
            s.global_step = gs
            assert s._get_model_learning_rate() == pytest.approx(osolver.learning_rate(global_step=gs, base_lr=1e-3, **kw))
            assert s._get_model_learning_rate(slow_start_step=7, slow_start_learning_rate=1e-4) == \
                pytest.approx(osolver.learning_rate(global_step=gs, base_lr=1e-3, slow_start_step=7, **kw))
    args = p.parse_args("--mode train --tag t --model UNet --classes Liver --learning_policy plateau "
                        "--lr_decay_rate 0.2 --lr_end 0".split())
    s = solver.Solver(args)
    assert s._get_model_learning_rate() == 1e-3
    assert s.update_plateau_lr() == pytest.approx(2e-4) and s._get_model_learning_rate() == pytest.approx(2e-4)
    assert solver.get_solver_params(args)["solver"]._optimizer_hparams() == {"beta1": 0.9, "beta2": 0.99}


def test_per_device_batch_size_and_strategy():
    assert distribution_utils.per_device_batch_size(32, 1) == 32
    assert distribution_utils.per_device_batch_size(64, 8) == 8
    with pytest.raises(ValueError, match="must be a multiple"):
        distribution_utils.per_device_batch_size(4, 8)       # cfg4 of BASELINE.json: reference raises too
    assert distribution_utils.get_distribution_strategy("off", 1) is None
    with pytest.raises(ValueError):
        distribution_utils.get_distribution_strategy("off", 2)
    with pytest.raises(NotImplementedError):
        distribution_utils.get_distribution_strategy("mirrored", 2, num_workers=2)


def test_synthetic_input_contract():
    images, labels, names = synthetic.make_batch(4, 64, 64, 3, 3, seed=1234)
    assert images.shape == (4, 64, 64, 3) and images.dtype == np.float32
    assert labels.shape == (4, 64, 64) and labels.dtype == np.int32 and set(np.unique(labels)) <= {0, 1, 2}
    frac1 = (labels >= 1).mean()
    assert 0.15 < frac1 < 0.30 and 0.002 < (labels == 2).mean() < 0.02
    i2, l2, _ = synthetic.make_batch(4, 64, 64, 3, 3, seed=1234)
    assert np.array_equal(images, i2) and np.array_equal(labels, l2)
    assert set(np.unique(synthetic.make_batch(2, 32, 32, 3, 2)[1])) <= {0, 1}


def test_profiles_readme_is_generated_from_the_committed_profiles():
    """profiles/rNN_README.md (round 4 on) is written by tools/profiles_readme.py from the bench lines, rocprofv3 tables and PMC
    summaries next to it: regenerating it must give the committed text (numbers in the prose cannot drift from the files)."""
    import glob
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rounds = sorted(os.path.basename(p)[:3] for p in glob.glob(os.path.join(root, "profiles", "r*_README.md")))
    rounds = [r for r in rounds if r >= "r04"]
    assert rounds
    for r in rounds:
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "profiles_readme.py"), r], stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True, cwd=root)
        assert out.returncode == 0, out.stderr[-2000:]
        with open(os.path.join(root, "profiles", r + "_README.md")) as f:
            assert out.stdout == f.read(), r


def test_input_gradient_pack_item_order_is_a_bijection():
    """csrc/pack.h dgrad_item: the order in which the threads of unetk_pack_many take the items of an input-gradient pack (QL
    lanes share a filter row's cache line) must visit every item exactly once, for every layer shape the nets have -- restated
    here (the device function is four integer divisions) and checked as a permutation; the packed bytes themselves are
    compared on the GPU (tests/test_gpu_pack_cache.py, tests/test_gpu_bf16s*.py)."""
    import numpy as np

    def dgrad_item(i, cin, q, ql):
        nl = 64 // ql
        if q % ql or cin % nl:
            return i
        l, blk = i & 63, i >> 6
        q_l, n_l = l % ql, l // ql
        nt_n = cin // nl
        nt, rest = blk % nt_n, blk // nt_n
        qt, t = rest % (q // ql), rest // (q // ql)
        return (t * q + qt * ql + q_l) * cin + nt * nl + n_l

    for cin, cout in [(64, 64), (128, 64), (64, 128), (256, 512), (1024, 1024), (3, 64), (20, 36), (64, 32)]:
        for ql, per in ((8, 4), (4, 8)):                      # fp32 packs: 4 floats per item, 8 lanes per row; bf16: 8 and 4
            if cout % per:
                continue
            q = cout // per
            total = 9 * q * cin
            idx = np.arange(total, dtype=np.int64)
            out = np.array([dgrad_item(int(i), cin, q, ql) for i in idx[:: max(1, total // 20000)]])
            assert out.min() >= 0 and out.max() < total
            if total <= 200000:
                full = np.array([dgrad_item(int(i), cin, q, ql) for i in idx])
                assert np.array_equal(np.sort(full), idx), (cin, cout, ql)

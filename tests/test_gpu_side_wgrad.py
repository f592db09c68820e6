"""GPU: filter gradients on the second HIP stream (ops._Side, UNETK_SIDE_WGRAD=1) -- the backward of slim.conv2d's
kernel variable (NetworksV2/UNet.py:79,85,94) is off the critical chain of the backward pass, so it may run beside the
HBM-bound passes.  Same kernels, same operands: one training step's loss and EVERY gradient must be bit-identical with
the option on and off, in fp32 and in the bf16 storage mode, and stay so over a few optimiser steps (a missing
stream join would let Adam read a filter gradient that is still being written)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(dtype, side, steps):
    import test_gpu_unet as t
    from boxsegliver_amd import ops
    from boxsegliver_amd.core.solver import Solver
    saved = (ops._Side.enabled, ops._Side.mode)
    ops._Side.enabled, ops._Side.mode = side, ("1" if side else "0")      # "1": every precision (the default takes fp32 units only)
    try:
        bs, size = 4, 128
        args = t.make_args(batch_size=bs, im_height=size, im_width=size, compute_dtype=dtype, learning_rate=1e-3)
        images, labels = t.synth(bs, size, size, 3, seed=9)
        model, batch = t.build(args, images, labels)
        solver = Solver(args)
        losses, grads = [], None
        for s in range(steps):
            loss = model(batch, "train", **t.YML)
            solver(loss, model)
            losses.append(float(loss))
            if s == 0:
                torch.cuda.synchronize()
                grads = {k: v.clone() for k, v in model.params.grad.items()}
        torch.cuda.synchronize()
        return losses, grads, {k: v.clone() for k, v in model.params.flat.items()}
    finally:
        ops._Side.enabled, ops._Side.mode = saved


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_side_stream_filter_gradients_are_bit_identical(dtype):
    from boxsegliver_amd import ops
    l0, g0, p0 = _run(dtype, False, 4)
    l1, g1, p1 = _run(dtype, True, 4)
    assert ops._Side.stream is not None, "the side stream was never used"
    assert l0 == l1, (l0, l1)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k


def _run3d(voxels, steps):
    import test_gpu_unet3d as t
    from boxsegliver_amd import ops
    from boxsegliver_amd.NetworksV2.UNet3D import UNet3D
    from boxsegliver_amd.core.solver import Solver
    from boxsegliver_amd.data.synthetic import make_batch_3d
    old = ops.SIDE_WGRAD3D_VOXELS
    ops.SIDE_WGRAD3D_VOXELS = voxels
    try:
        args = t.make_args(batch_size=1, im_depth=16, im_height=64, im_width=64, learning_rate=1e-3)
        images, labels, _ = make_batch_3d(1, 16, 64, 64, 1, 2, 31)
        inputs = {"images": torch.from_numpy(images).cuda(), "labels": torch.from_numpy(labels).cuda()}
        model = UNet3D(args)
        solver = Solver(args)
        losses, grads = [], None
        for s in range(steps):
            loss = model(inputs, "train", **t.YML)
            solver(loss, model)
            losses.append(float(loss))
            if s == 0:
                torch.cuda.synchronize()
                grads = {k: v.clone() for k, v in model.params.grad.items()}
        torch.cuda.synchronize()
        return losses, grads, {k: v.clone() for k, v in model.params.flat.items()}
    finally:
        ops.SIDE_WGRAD3D_VOXELS = old


def test_unet3d_filter_gradients_beside_the_input_gradient_are_bit_identical():
    """Round 4: every conv3d unit's filter gradient runs on the side stream beside its input gradient (ops.SIDE_WGRAD3D_VOXELS,
    the default).  Same kernels and operands: losses, gradients and variables over four optimiser steps equal the
    single-stream run bit for bit (a missing join would let Adam or the next step's pack read a gradient still being written)."""
    l0, g0, p0 = _run3d(0, 4)
    l1, g1, p1 = _run3d(1 << 30, 4)
    assert l0 == l1, (l0, l1)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k

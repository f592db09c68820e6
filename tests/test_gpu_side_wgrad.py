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
    ops._Side.enabled = side
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
        ops._Side.enabled = False


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

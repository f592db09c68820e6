"""Debug: for every conv unit's backward, recompute BN-bwd / dgrad / wgrad in fp64 on the CPU from the
SAME operands the HIP kernels saw, and print each kernel's own error (isolates it from upstream noise)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_unet as T
from boxsegliver_amd import ops
from oracle import tf_ops

args = T.make_args(loss_type=sys.argv[1] if len(sys.argv) > 1 else "dice", loss_weight_type=sys.argv[2] if len(sys.argv) > 2 else "none")
images, labels = T.synth(2, 32, 32, 3)
model, inputs = T.build(args, images, labels)
net, params = T.oracle_for(args)
model.params.load_state(params)
ops.DEBUG_CAPTURE = []
model.params.zero_grad()
loss = model(inputs, "train", **T.YML)
loss.backward()
torch.cuda.synchronize()
print("%-3s %-22s %9s %9s %9s %9s %9s | %9s %9s" % ("#", "shape", "dy", "dgamma", "dbeta", "dw", "dx", "|dz|max", "minvar"))
for i, c in enumerate(ops.DEBUG_CAPTURE):
    y = c["y"].detach().cpu().double().requires_grad_(True)
    g = c["gamma"].detach().cpu().double().requires_grad_(True)
    b = c["beta"].detach().cpu().double().requires_grad_(True)
    dz = c["dz"].detach().cpu().double()
    z, _, _ = tf_ops.batch_norm(y, g, b, torch.zeros_like(g), torch.ones_like(g), True)
    torch.relu(z).backward(dz)
    var = y.detach().var(dim=(0, 1, 2), unbiased=False)
    e_dy = T.rel(c["dy"].cpu().numpy(), y.grad.numpy())
    e_dg = T.rel(c["dgamma"].cpu().numpy(), g.grad.numpy())
    e_db = T.rel(c["dbeta"].cpu().numpy(), b.grad.numpy())
    # conv grads from the HIP dy
    x = c["x"].detach().cpu().double().contiguous().requires_grad_(True)
    wname = None
    dyh = c["dy"].detach().cpu().double()
    cin, cout = x.shape[3], dyh.shape[3]
    # find the weight by shape & order: reconstruct from params in backward order is messy; use autograd with a dummy w
    w = c["w"].cpu().double().requires_grad_(True)
    out = tf_ops.conv_nd_same(x, w)
    out.backward(dyh)
    e_dw = T.rel(c["dw"].cpu().numpy(), w.grad.numpy())
    e_dx = T.rel(c["dx"].cpu().numpy(), x.grad.numpy()) if c["dx"] is not None else float("nan")
    print("%-3d %-22s %9.2e %9.2e %9.2e %9.2e %9.2e | %9.2e %9.2e" % (i, "%dx%d %d->%d" % (x.shape[1], x.shape[2], cin, cout), e_dy, e_dg, e_db, e_dw, e_dx, dz.abs().max().item(), var.min().item()))

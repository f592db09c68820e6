"""Debug: per-layer activation and activation-gradient errors, HIP vs fp64 oracle vs fp32 CPU oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_unet as T

loss_type, w_type = sys.argv[1] if len(sys.argv) > 1 else "dice", sys.argv[2] if len(sys.argv) > 2 else "none"
args = T.make_args(loss_type=loss_type, loss_weight_type=w_type)
images, labels = T.synth(2, 32, 32, 3)
model, inputs = T.build(args, images, labels)
net, params = T.oracle_for(args)
model.params.load_state(params)

def run_oracle(dtype):
    q = {}
    for k, v in params.items():
        t = v.to(dtype).clone()
        if net.kinds[k] in T.unet2d.TRAINABLE_KINDS:
            t.requires_grad_(True)
        q[k] = t
    taps = {}
    total, _, logits, _ = net.loss(q, torch.from_numpy(images).to(dtype), torch.from_numpy(labels).long(), taps=taps, **T.loss_kwargs(args))
    for t in taps.values():
        t.retain_grad()
    total.backward()
    return taps

t64, t32 = run_oracle(torch.float64), run_oracle(torch.float32)
model._taps = {}
model.params.zero_grad()
loss = model(inputs, "train", **T.YML)
grads = {}
for k, t in model._taps.items():
    t.register_hook(lambda g, k=k: grads.__setitem__(k, g.detach().clone()))
loss.backward()
torch.cuda.synchronize()
print("%-45s %10s %10s | %10s %10s" % ("layer", "act hip", "act cpu32", "dact hip", "dact cpu32"))
for k in model._taps:
    a64 = t64[k].detach().numpy(); a32 = t32[k].detach().numpy()
    ah = model._taps[k].detach().cpu().numpy()
    g64 = t64[k].grad.numpy(); g32 = t32[k].grad.numpy(); gh = grads[k].cpu().numpy()
    print("%-45s %10.2e %10.2e | %10.2e %10.2e" % (k[5:], T.rel(ah, a64), T.rel(a32, a64), T.rel(gh, g64), T.rel(g32, g64)))

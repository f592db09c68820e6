"""Debug: per-parameter gradient error table, HIP vs fp64 oracle vs fp32 CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_unet as T

loss_type, w_type = sys.argv[1] if len(sys.argv) > 1 else "dice", sys.argv[2] if len(sys.argv) > 2 else "none"
args = T.make_args(loss_type=loss_type, loss_weight_type=w_type)
images, labels = T.synth(2, 32, 32, 3)
model, inputs = T.build(args, images, labels)
net, params = T.oracle_for(args)
model.params.load_state(params)
_, _, logits, grads, _ = net.loss_and_grads(params, torch.from_numpy(images), torch.from_numpy(labels).long(), **T.loss_kwargs(args))
p64 = {k: v.double() for k, v in params.items()}
_, _, logits64, grads64, _ = net.loss_and_grads(p64, torch.from_numpy(images).double(), torch.from_numpy(labels).long(), **T.loss_kwargs(args))
model.params.zero_grad()
loss = model(inputs, "train", **T.YML)
loss.backward()
print("logits err hip %.2e cpu32 %.2e" % (T.rel(model.layers["logits"].cpu().numpy(), logits64.numpy()), T.rel(logits.numpy(), logits64.numpy())))
for name in model.params.trainable_names():
    g = model.params[name].grad.cpu().numpy()
    r = T.rel(g, grads64[name].numpy()); rc = T.rel(grads[name].numpy(), grads64[name].numpy())
    flag = " <<<" if r > 3 * rc + 2e-4 else ""
    print("%-60s hip %.2e cpu32 %.2e |g|max %.2e%s" % (name[5:], r, rc, np.abs(grads64[name].numpy()).max(), flag))

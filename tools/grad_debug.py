"""dev probe: per-parameter gradient comparison HIP vs oracle at full size (python tools/grad_debug.py [precision] [n_images] [depth] [proposals])."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import wsovod_ref as R
from wsovod_amd.data import make_batch
from wsovod_amd.testing import build_hot_path_model

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 18
props = int(sys.argv[4]) if len(sys.argv) > 4 else 512
K = 20
host = make_batch(n, props, K, seed=4321)
cfg, model = build_hot_path_model(seed=0, depth=depth, K=K, precision=prec, device="cuda:0")
model.train()
for m in model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.eval()
sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
keys = [k for k, p in model.named_parameters() if p.requires_grad]
batch = [{"image": x["image"].cuda(), "proposals": x["proposals"].to("cuda"), "instances": x["instances"],
          "height": x["height"], "width": x["width"]} for x in host]
losses = model(batch)
sum(losses.values()).backward()
torch.cuda.synchronize()
got = {k: p.grad.detach().float().cpu() for k, p in model.named_parameters() if p.requires_grad}
for k in keys:
    sd[k].requires_grad_(True)
ol, inter = R.train_forward(sd, R.batch_from_inputs(host), depth=depth, num_classes=K)
grads = torch.autograd.grad(sum(ol.values()), [sd[k] for k in keys] + [inter["box_features"]], allow_unused=True)
print({k: (float(v), float(ol[k])) for k, v in losses.items()})
for k, g in zip(keys, grads):
    a = got[k]
    if g is None:
        print(k, "oracle None"); continue
    rel = float((a - g).norm() / g.norm().clamp(min=1e-30))
    cos = float((a.flatten() @ g.flatten()) / (a.norm() * g.norm()).clamp(min=1e-30))
    print(f"{k:55s} |g|={float(g.norm()):.4e} |hip|={float(a.norm()):.4e} rel_l2={rel:.3e} cos={cos:.8f} maxabs={float((a-g).abs().max()):.3e}")
gbf = grads[len(keys)]
print("oracle d box_features norm", float(gbf.norm()))
# where is the fc1 difference: per output row norms
k = "roi_heads.box_head.fc1.weight"
d = (got[k] - grads[keys.index(k)])
rn = d.norm(dim=1)
top = torch.topk(rn, 8)
print("fc1 diff row norms top:", top.values.tolist(), top.indices.tolist(), "total", float(d.norm()))
gr = grads[keys.index(k)].norm(dim=1)
print("fc1 oracle row norms at those rows:", gr[top.indices].tolist())

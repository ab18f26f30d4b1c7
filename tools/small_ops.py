"""Where do the tiny ATen launches of one training step come from?  torch.profiler with python stacks; prints, per
ATen device kernel family, launches per step and the nearest wsovod_amd / bench frame.  (python tools/small_ops.py [batch])"""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import ProfilerActivity, profile
from wsovod_amd.data import make_batch
from wsovod_amd.engine import HotPathTrainer, build_optimizer
from wsovod_amd.testing import build_hot_path_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg, model = build_hot_path_model(seed=0, precision="bf16", device="cuda:0")
model.train()
tr = HotPathTrainer(model, build_optimizer(cfg, model))
host = make_batch(B, 512, 20, seed=1)
batch = [{"image": x["image"].cuda(), "proposals": x["proposals"].to("cuda"), "instances": x["instances"],
          "height": x["height"], "width": x["width"]} for x in host]
for _ in range(5):
    tr.run_step(batch)
torch.cuda.synchronize()
STEPS = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(STEPS):
        tr.run_step(batch)
    torch.cuda.synchronize()
agg = collections.Counter()
dur = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or ev.cpu_children and any(
            c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
        continue
    where = "?"
    for fr in ev.stack or []:
        if "wsovod_amd" in fr or "bench.py" in fr:
            where = fr.split("/root/repo/")[-1] if "/root/repo/" in fr else fr[-90:]
            break
    agg[(ev.name, where)] += 1
    dur[(ev.name, where)] += ev.device_time_total
tot = 0.0
for key, us in sorted(dur.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{us / STEPS:8.1f} us/step  {agg[key] / STEPS:6.1f} x  {key[0]:<28} {key[1]}")
    tot += us / STEPS
print(f"listed ATen device time: {tot:.1f} us/step")

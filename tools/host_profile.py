"""cProfile of the host side of one training step at 1 image/step (where the step is host-bound)."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsovod_amd.data import make_batch
from wsovod_amd.engine import HotPathTrainer, build_optimizer
from wsovod_amd.testing import build_hot_path_model
cfg, model = build_hot_path_model(seed=0, precision="bf16", device="cuda:0")
model.train()
opt = build_optimizer(cfg, model)
tr = HotPathTrainer(model, opt)
host = make_batch(1, 512, 20, seed=1)
batch = [{"image": x["image"].cuda(), "proposals": x["proposals"].to("cuda"), "instances": x["instances"], "height": x["height"], "width": x["width"]} for x in host]
for _ in range(10): tr.run_step(batch)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(50): tr.run_step(batch)
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)

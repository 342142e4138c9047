import os, sys, faulthandler
faulthandler.enable()
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import test_gpu_capture as T
from equiformer_amd.graph import EdgeGraph
from equiformer_amd.capture import CapturedTrainStep
which = sys.argv[1]
m, opt, d = T._train_setup(0.0)
pos_a, z_a, y_a = d["pos"].clone(), d["z"].clone(), d["y"].clone()
pos, z, y = pos_a.clone(), z_a.clone(), y_a.clone()
def fl(g):
    return (m(None, pos, d["batch"], z, graph=g).squeeze() - y).abs().mean()
def build(into):
    return EdgeGraph.from_radius(pos, d["batch"], 5.0, num_graphs=6, into=into)
cs = CapturedTrainStep(opt, fl, min_eager=3)
for it in range(6):
    if "p" in which: pos.copy_(pos_a)
    if "z" in which: z.copy_(z_a)
    if "y" in which: y.copy_(y_a)
    loss = cs.step(build)
    print(it, float(loss), flush=True)
print("ok", which)

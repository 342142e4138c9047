import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import test_gpu_capture as T
from equiformer_amd.graph import EdgeGraph
from equiformer_amd.capture import CapturedTrainStep
hist = []
for use_graph in (False, True):
    m, opt, d = T._train_setup(0.0)
    def fl(g):
        return (m(None, d["pos"], d["batch"], d["z"], graph=g).squeeze() - d["y"]).abs().mean()
    def build(into):
        return EdgeGraph.from_radius(d["pos"], d["batch"], 5.0, num_graphs=6, into=into)
    cs = CapturedTrainStep(opt, fl, min_eager=3)
    h = []
    for it in range(6):
        for gr in opt.param_groups: gr["lr"] = 1e-3 * (1 + 0.1 * it)
        if use_graph: loss = cs.step(build)
        else:
            opt.zero_grad(set_to_none=True); loss = fl(build(None)); loss.backward(); opt.step(); loss = loss.detach()
        torch.cuda.synchronize()
        h.append((float(loss), opt.flat_p.clone(), opt.flat_m.clone(), opt.flat_v.clone(), None if opt._hyper_dev is None else opt._hyper_dev.cpu().tolist()))
    hist.append(h)
rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
for it in range(6):
    e, g = hist[0][it], hist[1][it]
    print(it, "loss %.6f %.6f  p %.2e m %.2e v %.2e  hyper %s" % (e[0], g[0], rel(g[1], e[1]), rel(g[2], e[2]), rel(g[3], e[3]), g[4]))

import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
import torch
import test_gpu_model as T
from equiformer_amd import ops
for mode in ("split", "fp32", "split6", "split"):
    with ops.matrix_mode(mode):
        try:
            _, worst = T._md17_fixture_case("md17_l2_bench8", "graph_attention_transformer_nonlinear_exp_l2_md17", True)
            print(mode, ["%s %.2e" % (n.replace("blocks.", "b"), e) for e, n in worst[:4]], flush=True)
        except AssertionError as ex:
            print(mode, "ASSERT", str(ex)[:200])
# the same without the radial bank (per-module radial MLPs: the round-4 second-order path)
from equiformer_amd.nets import graph_attention_transformer as G
G._Trunk.use_radial_bank = False
_, worst = T._md17_fixture_case("md17_l2_bench8", "graph_attention_transformer_nonlinear_exp_l2_md17", True)
print("split, radial bank off", ["%s %.2e" % (n.replace("blocks.", "b"), e) for e, n in worst[:4]], flush=True)

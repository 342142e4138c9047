#!/usr/bin/env python
"""Write the full-size oracle fixtures the GPU tests load (tests/golden/fullsize/*.npz; format: tests/golden/fullsize.py).

    python tests/golden/make_fullsize_golden.py            # every case (CPU, fp64 oracle: ~15 minutes on 8 cores)
    python tests/golden/make_fullsize_golden.py qm9_l2_bench md17_l3_second_order
    python tests/golden/make_fullsize_golden.py --check qm9_l2_bench     # recompute and compare with the stored file

Each case = one BASELINE.json configuration at the size the bench (or the reference script) runs it: the oracle
(`oracle/nets.py`, the CPU restatement pinned to the reference's own model code by tests/test_reference_pin.py) in fp64, weights
from the model's own initialisation under torch.manual_seed(0), inputs from the seeded synthetic generators of
equiformer_amd/synthetic.py.  The GPU tests rebuild the same weights and inputs from the same seeds (the
oracle's state_dict is loaded into the HIP model), so a fixture stores outputs only.
"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for _p in (HERE, ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import fullsize  # noqa: E402


# ------------------------------------------------------------------------------------------------------------- inputs
def qm9_bench_batch():
    from equiformer_amd.synthetic import qm9_like_batch
    return qm9_like_batch(128, 18, side=6.5, seed=11)


def oc20_bench_batch(B=16, Na=78, seed=1000):
    """the bench's OC20 workload (bench.py:_oc20_batch): slab + adsorbate shaped structures in an 11 x 11 x 30 A cell"""
    g = torch.Generator().manual_seed(seed)
    cell = torch.diag(torch.tensor([11.0, 11.0, 30.0]))[None].repeat(B, 1, 1)
    pos = (torch.rand(B * Na, 3, generator=g) * torch.tensor([1.0, 1.0, 0.45])) @ cell[0]
    return dict(pos=pos, batch=torch.arange(B).repeat_interleave(Na), cell=cell,
                atomic_numbers=torch.randint(1, 84, (B * Na,), generator=g), tags=torch.randint(0, 3, (B * Na,), generator=g),
                natoms=torch.full((B,), Na))


def md17_probe(frames, seed):
    """cotangents of the force loss L = <a, E> + <B, F>"""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(frames, 1, generator=g, dtype=torch.float64), torch.randn(frames * 21, 3, generator=g, dtype=torch.float64)


# -------------------------------------------------------------------------------------------------------------- cases
def case_qm9_l2_bench():
    """BASELINE configs #1/#2 at the bench batch: energies of 128 molecules + every parameter gradient of the L1 loss"""
    from oracle import nets as onets
    torch.manual_seed(0)
    ref = onets.graph_attention_transformer_nonlinear_l2("5x0e", 5.0).double().eval()
    d = qm9_bench_batch()
    # Molecules do not interact (no edge crosses a molecule, every normalisation is per node): the batch is evaluated in chunks of
    # 32 molecules and the gradient of the MEAN loss is the sum of the chunks' gradients of sum(|y - t|) / 128 -- the same
    # numbers as one 128-molecule pass up to fp64 summation order, at a quarter of its 60 GB of autograd state.
    ys, acc = [], None
    params = list(ref.parameters())
    for lo in range(0, 128, 32):
        sel = slice(lo * 18, (lo + 32) * 18)
        y = ref(None, d["pos"][sel].double(), d["batch"][sel] - lo, d["z"][sel])
        part = torch.autograd.grad((y.squeeze() - d["y"][lo:lo + 32].double()).abs().sum() / 128.0, params, allow_unused=True)
        acc = list(part) if acc is None else [a if b is None else (b if a is None else a + b) for a, b in zip(acc, part)]
        ys.append(y.detach())
    return dict(molecules=128, atoms=18, seed=11), {"energy": torch.cat(ys)}, {n: g for (n, _), g in zip(ref.named_parameters(), acc)}


def _md17(name, frames, seed, probe_seed, grads):
    from equiformer_amd.synthetic import md17_aspirin_batch
    from oracle import nets as onets
    torch.manual_seed(0)
    ref = onets.model_entrypoint(name)("64x0e", 5.0, num_basis=32).double().train()
    d = md17_aspirin_batch(frames, seed=seed)
    a, B = md17_probe(frames, probe_seed)
    params = list(ref.parameters())
    Es, Fs, acc = [], [], None
    for f in range(frames):  # frame by frame (frames do not interact; L = <a, E> + <B, F> is a sum over frames): bounded memory
        sel = slice(21 * f, 21 * (f + 1))
        E, F = ref(d["z"][sel], d["pos"][sel].double(), d["batch"][sel] - f)
        if grads:
            part = torch.autograd.grad((a[f:f + 1] * E).sum() + (B[sel] * F).sum(), params, allow_unused=True)
            acc = list(part) if acc is None else [x if y is None else (y if x is None else x + y) for x, y in zip(acc, part)]
        Es.append(E.detach())
        Fs.append(F.detach())
    g = {n: t for (n, _), t in zip(ref.named_parameters(), acc)} if grads else {}
    return dict(frames=frames, seed=seed, probe_seed=probe_seed), {"energy": torch.cat(Es), "forces": torch.cat(Fs)}, g


def case_md17_l3_second_order():
    """BASELINE config #4 at full size, one aspirin frame: E, F and the force-loss gradient of every parameter (the
    oracle's double backward)"""
    return _md17("graph_attention_transformer_nonlinear_exp_l3_md17", 1, 4, 1, True)


def case_md17_l2_bench8():
    """BASELINE config #3 at the bench batch (8 frames): E, F and the force-loss gradients"""
    return _md17("graph_attention_transformer_nonlinear_exp_l2_md17", 8, 1000, 2, True)


def case_md17_l3_bench5():
    """BASELINE config #4 at the bench batch (5 frames): E and F"""
    return _md17("graph_attention_transformer_nonlinear_exp_l3_md17", 5, 1000, 3, False)


def case_oc20_bench16():
    """BASELINE config #5 at the bench batch (16 structures x 78 atoms, periodic graph): energies"""
    from oracle import nets as onets, pbc
    torch.manual_seed(0)
    ref = onets.oc20_l1_256_nonlinear().double().eval()
    d = oc20_bench_batch()
    ei, coff, nb = pbc.radius_graph_pbc(d["pos"], d["cell"], [78] * 16, 5.0, 500)
    _, _, off = pbc.get_pbc_distances(d["pos"].double(), ei, d["cell"].double(), coff, nb)
    with torch.no_grad():
        y = ref(d["atomic_numbers"], d["tags"], d["pos"].double(), d["batch"], edge_index=ei, offsets=off)
    return dict(structures=16, atoms=78, seed=1000, edges=int(ei.shape[1])), {"energy": y}, {}


CASES = {
    "qm9_l2_bench": case_qm9_l2_bench,
    "md17_l3_second_order": case_md17_l3_second_order,
    "md17_l2_bench8": case_md17_l2_bench8,
    "md17_l3_bench5": case_md17_l3_bench5,
    "oc20_bench16": case_oc20_bench16,
}


def check(case):
    """recompute `case` and compare with the stored fixture (fp64 against fp64: 1e-9)"""
    meta, outs, grads = CASES[case]()
    _, souts, sgr = fullsize.load(case)
    for k, v in outs.items():
        err = float((v.detach().double() - souts[k]).abs().max() / souts[k].abs().max())
        assert err < 1e-9, (case, k, err)
    if grads:
        fullsize.compare_summary(grads, sgr, 1e-9, what=case + ": ")
    return True


def main(argv):
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    do_check = "--check" in argv
    names = [a for a in argv if not a.startswith("--")] or list(CASES)
    for name in names:
        t0 = time.perf_counter()
        if do_check:
            check(name)
            print("%-24s matches the stored fixture (%.0f s)" % (name, time.perf_counter() - t0), flush=True)
            continue
        meta, outs, grads = CASES[name]()
        p = fullsize.save(name, meta, outs, grads)
        print("%-24s -> %s (%.1f KB, %d gradient tensors, %.0f s)" % (name, os.path.relpath(p, ROOT), os.path.getsize(p) / 1e3,
                                                                    len(grads), time.perf_counter() - t0), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])

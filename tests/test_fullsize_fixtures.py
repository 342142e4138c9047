"""The full-size oracle fixtures of tests/golden/fullsize/ (CPU only): the summary format detects what it must, the committed
files carry every case the generator knows, and the generator reproduces them (two of the five cases are recomputed here --
the fp64 oracle at full size -- the other three take 1-8 minutes each and are recomputed with
`python tests/golden/make_fullsize_golden.py --check`)."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import fullsize  # noqa: E402
import make_fullsize_golden as mk  # noqa: E402


def _tensors(seed):
    g = torch.Generator().manual_seed(seed)
    return {"small": torch.randn(37, 11, generator=g, dtype=torch.float64),
            "large": torch.randn(300, 200, generator=g, dtype=torch.float64) * 1e-3,
            "absent": None,
            "zero": torch.zeros(5000, dtype=torch.float64)}


def test_summary_roundtrip_and_what_it_detects():
    ref = _tensors(1)
    summ = fullsize.summarize(ref)
    assert "all::small" in summ and "smp::large" in summ and "prj::large" in summ and "none::absent" in summ
    assert summ["smp::large"].shape == (fullsize.SAMPLE,) and summ["prj::large"].shape == (fullsize.NPROJ,)
    # the same tensors in fp32 pass at the fp32 level
    worst = fullsize.compare_summary({k: (None if v is None else v.float()) for k, v in ref.items()}, summ, 1e-6)
    assert worst and worst[0][0] < 1e-6
    # an error of 1e-3 of the largest entry in ONE element of the large tensor (not necessarily a sampled one) shows in the
    # projections; a uniform relative error shows everywhere
    bad = {k: (None if v is None else v.clone()) for k, v in ref.items()}
    bad["large"].view(-1)[12345] += 1e-3 * float(ref["large"].abs().max()) * 50
    with pytest.raises(AssertionError):
        fullsize.compare_summary(bad, summ, 1e-4)
    bad = {k: (None if v is None else v * (1 + 3e-4)) for k, v in ref.items()}
    with pytest.raises(AssertionError):
        fullsize.compare_summary(bad, summ, 1e-4)
    # structure: a gradient where the oracle has none
    bad = dict(ref, absent=torch.ones(3, dtype=torch.float64))
    with pytest.raises(AssertionError):
        fullsize.compare_summary(bad, summ, 1e-4)


def test_every_case_has_a_committed_fixture():
    for case in mk.CASES:
        meta, outs, grads = fullsize.load(case)
        assert "energy" in outs and torch.isfinite(outs["energy"]).all(), case
        assert os.path.getsize(fullsize.path(case)) < 4 << 20, case  # "small fixtures": a few MB at most
    m, o, g = fullsize.load("qm9_l2_bench")
    assert int(m["molecules"]) == 128 and o["energy"].shape == (128, 1)
    assert sum(1 for k in g if k.startswith("max::")) == 214  # every parameter tensor of the 3 531 715-parameter model
    m, o, g = fullsize.load("md17_l2_bench8")
    assert o["forces"].shape == (8 * 21, 3) and sum(1 for k in g if k.startswith("max::")) == 210
    assert fullsize.load("oc20_bench16")[1]["energy"].shape[0] == 16


@pytest.mark.parametrize("case", ["oc20_bench16", "md17_l3_second_order"])
def test_generator_reproduces_the_committed_fixture(case):
    """fp64 oracle recomputed here (11 s / ~25 s on 8 cores) against the stored file at 1e-9"""
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    assert mk.check(case)

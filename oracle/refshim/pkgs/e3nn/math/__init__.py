"""e3nn.math stand-in: normalize2mom (nets/fast_activation.py:25) and perm.inverse (nets/tensor_product_rescale.py:229)."""
import torch

from oracle import e3 as _e3
from . import perm  # noqa: F401


class normalize2mom(torch.nn.Module):
    """e3nn/math/_normalize_activation.py: f scaled so that its second moment under N(0,1) is 1; the constant is e3nn's
    Monte-Carlo estimate (1e6 samples, manual_seed(0), fp64) -- oracle.e3.normalize2mom_const."""

    def __init__(self, f, dtype=None, device=None):
        super().__init__()
        cst = _e3.normalize2mom_const(f)
        self._is_id = abs(cst - 1) < 1e-4
        self.f = f
        self.cst = cst

    def forward(self, x):
        return self.f(x) if self._is_id else self.f(x).mul(self.cst)

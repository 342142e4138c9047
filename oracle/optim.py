"""TEST INFRASTRUCTURE (oracle): CPU restatement of the optimizer-side arithmetic of the reference's train loop.

The reference builds `torch.optim.AdamW` over the name-based groups of optim_factory.py:27-42,126-127, optionally
clips with timm's `dispatch_clip_grad(mode='norm')` = `torch.nn.utils.clip_grad_norm_` (engine.py:76-78) and updates
`timm.utils.ModelEmaV2` (engine.py:89-90; timm 0.4.12, un-vendored).  The formulas below restate those libraries;
tests/test_optim.py pins them against the torch build in this image (torch.optim.AdamW, clip_grad_norm_), which is the
very dependency the reference calls.
"""
import numpy as np


def adamw_step(p, g, m, v, step, lr, beta1, beta2, eps, weight_decay):
    """One torch.optim.AdamW step (amsgrad=False, maximize=False) in float64 numpy; returns (p, m, v)."""
    p = p * (1.0 - lr * weight_decay)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = np.sqrt(v) / np.sqrt(bc2) + eps
    return p - (lr / bc1) * m / denom, m, v


def clip_coef(grads, max_norm):
    """clip_grad_norm_(norm_type=2): gradients are scaled by min(1, max_norm / (total_norm + 1e-6))."""
    total = np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads))
    return min(1.0, max_norm / (total + 1e-6)), total


def ema_update(ema, p, decay):
    """ModelEmaV2: ema = decay * ema + (1 - decay) * model."""
    return decay * ema + (1.0 - decay) * p

"""e3nn.nn.Gate / Activation are only named by a class the models never build
(nets/tensor_product_rescale.py:195-221; the models use nets/fast_activation.py)."""
from . import models  # noqa: F401


class Activation:
    def __init__(self, *a, **kw):
        raise NotImplementedError("e3nn.nn.Activation is not on the reference's model path")


class Gate(Activation):
    pass

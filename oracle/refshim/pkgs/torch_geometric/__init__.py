"""torch_geometric 2.0.3 stand-in: utils.softmax / utils.degree / nn.inits.glorot / nn.global_*_pool (the last two are
only imported by norm layers no shipped configuration uses)."""
from . import utils, nn  # noqa: F401

from torch_scatter import scatter
from . import inits  # noqa: F401


def global_mean_pool(x, batch, size=None):
    return scatter(x, batch, dim=0, dim_size=size, reduce="mean")


def global_max_pool(x, batch, size=None):
    return scatter(x, batch, dim=0, dim_size=size, reduce="max")

"""Caller side of the boundary: the batch format the drivers hand to the model, without torch_geometric.

The reference's loaders are `torch_geometric.loader.DataLoader` (main_qm9.py:9,204-216; main_md17.py) over PyG `Data`
objects (datasets/pyg/qm9.py:281-284): a batch is a PyG `Batch` -- node-level tensors concatenated along dim 0, a
`batch` vector (graph id per node, ascending), `ptr` (node offsets), per-graph tensors such as `y` concatenated along
dim 0, and every attribute whose name contains "index" concatenated along the LAST dim with the node offset of its graph
added (PyG `Data.__cat_dim__` / `__inc__`; the QM9 samples carry the dense pair list `edge_d_index` / `edge_d_attr`,
which the model ignores, engine.py:63-66).  torch_geometric is an un-vendored dependency (pyg 2.0.3); its collation
rules are restated here (and, independently, in oracle/collate.py for the tests).

`DataLoader` is a `torch.utils.data.DataLoader` with that collation as `collate_fn`: same constructor as PyG's
(`dataset, batch_size, shuffle, follow_batch, exclude_keys, **kwargs`), works with `DistributedSampler`, worker
processes and `pin_memory=True` (`Batch.pin_memory`), so `for data in loader: data = data.to(device)` of engine.py:58-59
runs unchanged; `.to(device, non_blocking=True)` on a pinned batch overlaps the copy with the previous step.
"""
import torch


class Data:
    """Attribute bag of tensors describing one graph (subset of torch_geometric.data.Data used by the drivers)."""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    def keys(self):
        return [k for k, v in self.__dict__.items() if not k.startswith("_") and v is not None]

    def __getitem__(self, key):
        return getattr(self, key)

    def __contains__(self, key):
        return key in self.keys()

    @property
    def num_nodes(self):
        for k in ("pos", "x", "z", "atomic_numbers"):
            v = getattr(self, k, None)
            if torch.is_tensor(v):
                return int(v.shape[0])
        raise ValueError("cannot infer the number of nodes (no pos / x / z)")

    def _apply(self, fn):
        for k in self.keys():
            v = getattr(self, k)
            if torch.is_tensor(v):
                setattr(self, k, fn(v))
        return self

    def to(self, device, non_blocking=False):
        return self._apply(lambda t: t.to(device, non_blocking=non_blocking))

    def pin_memory(self):
        return self._apply(lambda t: t.pin_memory())

    def __repr__(self):
        items = ", ".join("%s=%s" % (k, list(getattr(self, k).shape) if torch.is_tensor(getattr(self, k)) else
                                     repr(getattr(self, k))) for k in self.keys())
        return "%s(%s)" % (type(self).__name__, items)


class Batch(Data):
    """Concatenation of `Data` objects; adds `batch`, `ptr`, `num_graphs` and `natoms`."""

    @staticmethod
    def from_data_list(data_list, follow_batch=(), exclude_keys=()):
        if len(data_list) == 0:
            raise ValueError("empty batch")
        keys = [k for k in data_list[0].keys() if k not in exclude_keys]
        sizes = [d.num_nodes for d in data_list]
        out = Batch()
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + n)
        for k in keys:
            vals = [getattr(d, k) for d in data_list]
            if torch.is_tensor(vals[0]):
                if "index" in k:  # PyG: __cat_dim__ = -1, __inc__ = num_nodes
                    vals = [v + o for v, o in zip(vals, offs)]
                    setattr(out, k, torch.cat(vals, dim=-1))
                elif vals[0].dim() == 0:
                    setattr(out, k, torch.stack(vals))
                else:
                    setattr(out, k, torch.cat(vals, dim=0))
            elif isinstance(vals[0], (int, float)):
                setattr(out, k, torch.tensor(vals))
            else:
                setattr(out, k, vals)  # names etc. stay Python lists
            if k in follow_batch and torch.is_tensor(vals[0]):
                setattr(out, k + "_batch", torch.cat([torch.full((v.shape[0],), i, dtype=torch.long)
                                                     for i, v in enumerate(vals)]))
        nat = torch.tensor(sizes, dtype=torch.long)
        out.batch = torch.repeat_interleave(torch.arange(len(sizes)), nat)
        out.ptr = torch.tensor(offs, dtype=torch.long)
        out.natoms = nat
        out._num_graphs = len(sizes)
        return out

    @property
    def num_graphs(self):
        return self._num_graphs

    def to(self, device, non_blocking=False):
        super().to(device, non_blocking=non_blocking)
        return self


class Collater:
    def __init__(self, follow_batch=(), exclude_keys=()):
        self.follow_batch, self.exclude_keys = tuple(follow_batch or ()), tuple(exclude_keys or ())

    def __call__(self, data_list):
        return Batch.from_data_list(data_list, self.follow_batch, self.exclude_keys)


class DataLoader(torch.utils.data.DataLoader):
    """Drop-in for torch_geometric.loader.DataLoader [ref: main_qm9.py:204-216]."""

    def __init__(self, dataset, batch_size=1, shuffle=False, follow_batch=None, exclude_keys=None, **kwargs):
        kwargs.pop("collate_fn", None)
        super().__init__(dataset, batch_size, shuffle, collate_fn=Collater(follow_batch, exclude_keys), **kwargs)

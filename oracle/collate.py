"""TEST INFRASTRUCTURE (oracle): PyG 2.0.3 `Batch.from_data_list` collation rules, written as plain loops.

The reference batches with torch_geometric.loader.DataLoader (main_qm9.py:9,204-216); torch_geometric is an
un-vendored dependency that cannot be imported here (dependency restatement, collation is not on the arithmetic path), so this restates its documented rules:
tensors are concatenated along dim 0, except attributes whose name contains "index", which are concatenated along the
last dim after adding the number of nodes of the preceding graphs; `batch[i]` = graph of node i; `ptr` = node offsets.
"""
import torch


def collate(samples):
    """samples: list of dicts of tensors with at least 'pos' -> dict"""
    out = {}
    n_before = 0
    batch, ptr = [], [0]
    for gi, s in enumerate(samples):
        n = s["pos"].shape[0]
        for k, v in s.items():
            if "index" in k:
                out.setdefault(k, []).append(v + n_before)
            else:
                out.setdefault(k, []).append(v)
        batch.extend([gi] * n)
        n_before += n
        ptr.append(n_before)
    res = {k: torch.cat(v, dim=-1 if "index" in k else 0) for k, v in out.items()}
    res["batch"] = torch.tensor(batch, dtype=torch.long)
    res["ptr"] = torch.tensor(ptr, dtype=torch.long)
    return res

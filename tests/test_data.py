"""Batch collation (caller side of the boundary): equiformer_amd.data vs the loop restatement in oracle/collate.py and
hand-written known answers; loader behaviour with samplers, workers and pinned memory."""
import pytest
import torch

from equiformer_amd.data import Batch, Data, DataLoader
from oracle import collate as ocollate


def _samples(sizes, seed=0):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i, n in enumerate(sizes):
        idx = torch.arange(n)
        dst = torch.repeat_interleave(idx, n)[None]
        src = idx.repeat(n)[None]
        pos = torch.randn(n, 3, generator=g)
        out.append(dict(x=torch.randn(n, 5, generator=g), pos=pos, z=torch.randint(1, 10, (n,), generator=g),
                        y=torch.randn(1, 19, generator=g), edge_d_index=torch.cat([dst, src]),
                        edge_d_attr=(pos[dst[0]] - pos[src[0]]).norm(dim=1)))
    return out


def test_collation_matches_oracle_and_known_answer():
    sizes = [3, 1, 5, 2]
    samples = _samples(sizes)
    b = Batch.from_data_list([Data(**s) for s in samples])
    ref = ocollate.collate(samples)
    for k, v in ref.items():
        assert torch.equal(getattr(b, k), v), k
    assert b.batch.tolist() == [0, 0, 0, 1, 2, 2, 2, 2, 2, 3, 3]
    assert b.ptr.tolist() == [0, 3, 4, 9, 11] and b.natoms.tolist() == sizes and b.num_graphs == 4
    assert b.y.shape == (4, 19) and b.pos.shape == (11, 3) and b.edge_d_index.shape == (2, 9 + 1 + 25 + 4)
    # the pair list of graph 2 starts at node offset 4
    assert b.edge_d_index[:, 9 + 1].tolist() == [4, 4] and int(b.edge_d_index.max()) == 10
    # every pair stays inside its graph
    assert (b.batch[b.edge_d_index[0]] == b.batch[b.edge_d_index[1]]).all()


def test_exclude_keys_and_follow_batch_and_python_attributes():
    samples = _samples([2, 3])
    ds = [Data(name="mol%d" % i, index=i, **s) for i, s in enumerate(samples)]
    b = Batch.from_data_list(ds, follow_batch=("edge_d_attr",), exclude_keys=("edge_d_index",))
    assert not hasattr(b, "edge_d_index") and b.name == ["mol0", "mol1"] and b.index.tolist() == [0, 1]
    assert b.edge_d_attr_batch.tolist() == [0] * 4 + [1] * 9


def test_loader_with_distributed_sampler_workers_and_pinning():
    sizes = [1 + (i * 7) % 5 for i in range(23)]
    ds = [Data(**s) for s in _samples(sizes, 3)]
    seen = []
    for rank in range(2):
        sampler = torch.utils.data.DistributedSampler(ds, num_replicas=2, rank=rank, shuffle=True, seed=1)
        loader = DataLoader(ds, batch_size=4, sampler=sampler, drop_last=True, num_workers=2 if rank == 0 else 0)
        n = 0
        for data in loader:
            assert data.num_graphs == 4 and data.batch[-1] == 3 and data.ptr[-1] == data.pos.shape[0]
            seen.append(data.pos.shape[0])
            n += 1
        assert n == 3  # 12 samples per rank, batches of 4
    assert len(seen) == 6
    if torch.cuda.is_available():
        loader = DataLoader(ds, batch_size=5, pin_memory=True)
        data = next(iter(loader))
        assert data.pos.is_pinned()
        data = data.to("cuda", non_blocking=True)
        assert data.batch.is_cuda


@pytest.mark.gpu
def test_collated_batch_feeds_the_model():
    from equiformer_amd import nets
    from equiformer_amd.synthetic import qm9_like_batch
    dev = torch.device("cuda:0")
    d = qm9_like_batch(3, 9, side=4.5, seed=5)
    ds = [Data(pos=d["pos"][i * 9:(i + 1) * 9], z=d["z"][i * 9:(i + 1) * 9], y=d["y"][i:i + 1, None]) for i in range(3)]
    loader = DataLoader(ds, batch_size=3, pin_memory=True)
    data = next(iter(loader)).to(dev, non_blocking=True)
    torch.manual_seed(0)
    model = nets.model_entrypoint("graph_attention_transformer_nonlinear_l2")(irreps_in="5x0e", radius=5.0,
                                                                               num_basis=32).to(dev).eval()
    with torch.no_grad():
        a = model(f_in=None, pos=data.pos, batch=data.batch, node_atom=data.z)
        b = model(f_in=None, pos=d["pos"].to(dev), batch=d["batch"].to(dev), node_atom=d["z"].to(dev))
    assert torch.equal(a, b) and a.shape == (3, 1)

"""Device-side collate / MaskAtom over an HBM-resident dataset (csrc/loader.hip, data/resident.py)
against the host collate that restates BatchMasking.from_data_list (chem/batch.py:17-52) and MaskAtom
(chem/util.py:225-277).  Integer / byte work: every comparison is bit-exact."""
import numpy as np
import pytest
import torch

from pretrain_gnns_amd.data import Data, resident, synthetic
from oracle import hostdata

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _chem_graphs(count, seed=0):
    rng = np.random.default_rng(seed)
    return [synthetic.zinc_like_graph(rng) for _ in range(count)]


def _same(batch, want, keys=("x", "edge_index", "edge_attr", "batch")):
    for k in keys:
        got, ref = getattr(batch, k).cpu(), getattr(want, k)
        assert got.dtype == ref.dtype and got.shape == ref.shape, k
        assert torch.equal(got, ref), k


@pytest.mark.parametrize("batch_size", [1, 7, 64, 2500])
def test_collate_matches_host_collate_chem(batch_size):
    graphs = _chem_graphs(50, seed=1)
    ds = resident.ResidentDataset.from_graphs(graphs, DEV, relabel=False)
    ids = np.random.default_rng(batch_size).integers(0, len(graphs), size=batch_size)  # with repeats; > 1024 ids = scan carry
    out = ds.collate(ids)
    ds.check(out)
    _same(out, hostdata.collate([graphs[i] for i in ids]))
    assert out.num_graphs == batch_size


def test_collate_matches_host_collate_bio():
    rng = np.random.default_rng(2)
    graphs = [synthetic.ppi_like_graph(rng) for _ in range(12)]
    ds = resident.ResidentDataset.from_graphs(graphs, DEV, relabel=False)
    ids = [3, 3, 0, 11, 7]
    out = ds.collate(ids)
    ds.check(out)
    _same(out, hostdata.collate([graphs[i] for i in ids]))


def test_collate_ragged_and_empty_graphs():
    """a single-atom molecule (no bonds), a two-atom one, and regular ones, in every position"""
    lone = Data(x=torch.tensor([[5, 0]]), edge_index=torch.zeros(2, 0, dtype=torch.int64),
                edge_attr=torch.zeros(0, 2, dtype=torch.int64))
    pair = Data(x=torch.tensor([[6, 0], [7, 1]]), edge_index=torch.tensor([[0, 1], [1, 0]]),
                edge_attr=torch.tensor([[1, 0], [1, 0]]))
    graphs = [lone, pair] + _chem_graphs(3, seed=3)
    ds = resident.ResidentDataset.from_graphs(graphs, DEV, relabel=False)
    for ids in ([0], [0, 0, 0], [0, 1, 2], [2, 0, 3, 0, 1], [1, 0]):
        out = ds.collate(ids)
        ds.check(out)
        _same(out, hostdata.collate([graphs[i] for i in ids]))


def test_bad_graph_ids_are_refused():
    ds = resident.ResidentDataset.from_graphs(_chem_graphs(4), DEV, relabel=False)
    with pytest.raises(IndexError):
        ds.collate([0, 4])
    with pytest.raises(ValueError):
        ds.collate([])


def test_mask_atoms_properties_and_determinism():
    graphs = _chem_graphs(40, seed=4)
    ds = resident.ResidentDataset.from_graphs(graphs, DEV, relabel=False)
    ids = np.arange(40)[::-1].copy()
    plain = ds.collate(ids)
    a = ds.collate(ids, mask_rate=0.15, seed=11)
    b = ds.collate(ids, mask_rate=0.15, seed=11)
    c = ds.collate(ids, mask_rate=0.15, seed=12)
    ds.check(a)
    assert torch.equal(a.masked_atom_indices, b.masked_atom_indices) and torch.equal(a.x, b.x)
    assert not torch.equal(a.masked_atom_indices, c.masked_atom_indices)
    idx = a.masked_atom_indices.cpu()
    sizes = [graphs[i].x.size(0) for i in ids]
    off = np.concatenate([[0], np.cumsum(sizes)])
    pos = 0
    for g, n in enumerate(sizes):
        k = int(n * 0.15 + 1)  # chem/util.py:232
        mine = idx[pos:pos + k]
        pos += k
        assert len(set(mine.tolist())) == k  # distinct
        assert int(mine.min()) >= off[g] and int(mine.max()) < off[g + 1]  # inside its own graph
    assert pos == idx.numel()
    # labels are the original rows, masked rows carry the mask token, every other row is untouched
    assert torch.equal(a.mask_node_label.cpu(), plain.x.cpu()[idx])
    x = a.x.cpu()
    assert torch.equal(x[idx], torch.tensor([[119, 0]]).repeat(idx.numel(), 1))
    keep = torch.ones(x.size(0), dtype=torch.bool)
    keep[idx] = False
    assert torch.equal(x[keep], plain.x.cpu()[keep])
    # the draw belongs to the graph, not to its position in the batch
    d = ds.collate(ids[::-1].copy(), mask_rate=0.15, seed=11)
    first_graph = int(ids[0])
    k0 = int(sizes[0] * 0.15 + 1)
    local_a = (a.masked_atom_indices[:k0] - 0).cpu()
    n_before = sum(graphs[i].x.size(0) for i in ids[::-1][:-1])
    local_d = (d.masked_atom_indices[-k0:] - n_before).cpu()
    assert torch.equal(local_a, local_d), first_graph


def test_mask_atoms_is_uniform():
    """every atom of a 20-atom graph is picked with frequency k/n = 4/20 over many seeds"""
    g = Data(x=torch.tensor([[5, 0]] * 20), edge_index=torch.zeros(2, 0, dtype=torch.int64),
             edge_attr=torch.zeros(0, 2, dtype=torch.int64))
    ds = resident.ResidentDataset.from_graphs([g], DEV, relabel=False)
    counts = torch.zeros(20)
    trials = 600
    for seed in range(trials):
        counts[ds.collate([0], mask_rate=0.15, seed=seed).masked_atom_indices.cpu()] += 1
    freq = counts / trials
    assert float(counts.sum()) == trials * 4
    assert float(freq.min()) > 0.13 and float(freq.max()) < 0.27, freq  # 0.2 +- 4 sigma (sigma = 0.016)


@pytest.mark.parametrize("mask_edge", [False, True])
def test_explicit_indices_match_host_maskatom(mask_edge):
    """MaskAtom's debugging hook (masked_atom_indices given): the device result equals the host
    restatement applied per graph, then collated -- node labels, masked rows, bond labels, bond rows."""
    graphs = _chem_graphs(16, seed=5)
    rng = np.random.default_rng(6)
    masked, picks = [], []
    off = 0
    for g in graphs:
        n = g.x.size(0)
        idx = rng.choice(n, int(n * 0.15 + 1), replace=False).astype(np.int64)
        d = g.clone()
        d.mask_node_label = d.x[idx].clone()
        d.masked_atom_indices = torch.from_numpy(idx)
        d.x[idx] = torch.tensor([119, 0])
        if mask_edge:
            ei = d.edge_index.numpy()
            connected = np.nonzero(np.isin(ei[0], idx) | np.isin(ei[1], idx))[0]
            first = connected[::2]
            d.mask_edge_label = d.edge_attr[first].clone()
            d.edge_attr[connected] = torch.tensor([5, 0])
            d.connected_edge_indices = torch.from_numpy(first.astype(np.int64))
        masked.append(d)
        picks.append(idx + off)
        off += n
    want = hostdata.collate(masked)
    ds = resident.ResidentDataset.from_graphs(graphs, DEV, relabel=False)
    out = ds.collate(np.arange(16), masked_atom_indices=np.concatenate(picks), mask_edge=mask_edge)
    ds.check(out)
    keys = ["x", "edge_index", "edge_attr", "batch", "masked_atom_indices", "mask_node_label"]
    if mask_edge:
        keys += ["connected_edge_indices", "mask_edge_label"]
    _same(out, want, keys)


def test_bio_mask_edge_matches_host_and_properties():
    """bio MaskEdge (bio/util.py:77-102): explicit pairs bit-exact vs the host restatement; random draw:
    k = int(E/2 * rate + 1) distinct even indices inside each graph's edge range, both directions
    overwritten with the mask row, labels = original rows."""
    rng = np.random.default_rng(8)
    graphs = [synthetic.ppi_like_graph(rng) for _ in range(6)]
    ds = resident.ResidentDataset.from_graphs(graphs, DEV, relabel=False)
    masked = [hostdata.mask_edges(g, rng) for g in graphs]
    want = hostdata.collate(masked)
    out = ds.collate(np.arange(6), masked_edge_idx=want.masked_edge_idx)
    ds.check(out)
    _same(out, want, ("x", "edge_index", "edge_attr", "batch", "masked_edge_idx", "mask_edge_label"))

    plain = ds.collate(np.arange(6))
    a = ds.collate(np.arange(6), mask_rate=0.15, seed=5)
    b = ds.collate(np.arange(6), mask_rate=0.15, seed=5)
    ds.check(a)
    assert torch.equal(a.masked_edge_idx, b.masked_edge_idx) and torch.equal(a.edge_attr, b.edge_attr)
    idx = a.masked_edge_idx.cpu()
    eoff = np.concatenate([[0], np.cumsum([g.edge_index.size(1) for g in graphs])])
    pos = 0
    for g in range(6):
        k = int((eoff[g + 1] - eoff[g]) // 2 * 0.15 + 1)
        mine = idx[pos:pos + k]
        pos += k
        assert len(set(mine.tolist())) == k and bool((mine % 2 == 0).all())
        assert int(mine.min()) >= eoff[g] and int(mine.max()) < eoff[g + 1]
    assert pos == idx.numel()
    assert torch.equal(a.mask_edge_label.cpu(), plain.edge_attr.cpu()[idx])
    row = torch.zeros(9)
    row[8] = 1
    attr = a.edge_attr.cpu()
    assert torch.equal(attr[idx], row.repeat(idx.numel(), 1)) and torch.equal(attr[idx + 1], row.repeat(idx.numel(), 1))
    keep = torch.ones(attr.size(0), dtype=torch.bool)
    keep[idx] = False
    keep[idx + 1] = False
    assert torch.equal(attr[keep], plain.edge_attr.cpu()[keep])


def test_resident_batch_drives_the_train_step_identically():
    """same graphs + same masked atoms => the train step sees bit-identical inputs => identical loss"""
    from pretrain_gnns_amd import train
    from pretrain_gnns_amd.chem import model as hmodel
    host = hostdata.chem_masking_batch(32, seed=9)
    rng = np.random.default_rng(9)  # chem_masking_batch draws graph, mask, graph, mask, ... from this stream
    graphs = []
    for _ in range(32):
        g = synthetic.zinc_like_graph(rng)
        hostdata.mask_atoms(g, rng)
        graphs.append(g)
    ds = resident.ResidentDataset.from_graphs(graphs, DEV, relabel=False)
    dev_batch = ds.collate(np.arange(32), masked_atom_indices=host.masked_atom_indices)
    _same(dev_batch, host, ("x", "edge_index", "edge_attr", "batch", "masked_atom_indices", "mask_node_label"))
    losses = []
    for batch in (host.to(DEV), dev_batch):
        torch.manual_seed(0)
        mods = [hmodel.GNN(5, 300).to(DEV), torch.nn.Linear(300, 119).to(DEV), torch.nn.Linear(300, 4).to(DEV)]
        opts = [torch.optim.Adam(m.parameters(), lr=1e-3) for m in mods]
        losses.append([train.chem_masking_step(mods, opts, batch)[0] for _ in range(2)])
    assert losses[0] == losses[1]


def test_loader_epoch_covers_dataset_once():
    graphs = _chem_graphs(37, seed=7)
    ds = resident.ResidentDataset.from_graphs(graphs, DEV, relabel=False)
    loader = resident.ResidentLoader(ds, batch_size=8, shuffle=True, seed=3, mask_rate=0.15)
    seen_nodes, batches = 0, 0
    for batch in loader:
        ds.check(batch)
        seen_nodes += batch.x.size(0)
        batches += 1
        assert batch.masked_atom_indices.numel() == batch.mask_node_label.size(0) > 0
    assert batches == len(loader) == 5
    assert seen_nodes == sum(g.x.size(0) for g in graphs)


def test_loader_epochs_through_the_pinned_id_staging_match_the_host_permutation():
    """five epochs back to back without a synchronisation in between: the graph ids of every epoch go up through one of two
    pinned staging buffers (no pipeline stall at the epoch boundary), and every batch must still be the collate of exactly
    ``batch_ids(epoch)`` -- node counts per batch against the host's, epoch after epoch, also after a buffer is reused"""
    graphs = _chem_graphs(50, seed=17)
    sizes = np.array([g.x.size(0) for g in graphs])
    ds = resident.ResidentDataset.from_graphs(graphs, DEV, relabel=False)
    loader = resident.ResidentLoader(ds, batch_size=16, shuffle=True, seed=9)
    got, want = [], []
    for epoch in range(5):
        want.append([int(sizes[ids].sum()) for ids in loader.batch_ids(epoch)])
        got.append([b.batch for b in loader])  # device tensors only: nothing waits for the GPU inside the epochs
    for epoch in range(5):
        assert [int(b.numel()) for b in got[epoch]] == want[epoch]
        for b, ids in zip(got[epoch], loader.batch_ids(epoch)):
            counts = torch.bincount(b.cpu(), minlength=ids.size).numpy()
            assert np.array_equal(counts, sizes[ids])


CTX_KEYS = ("x_substruct", "edge_index_substruct", "edge_attr_substruct", "x_context", "edge_index_context",
            "edge_attr_context", "center_substruct_idx", "overlap_context_substruct_idx", "batch_overlapped_context",
            "overlapped_context_size")


@pytest.mark.parametrize("k,l1,l2", [(5, 4, 7), (3, 2, 5), (1, 0, 1), (4, 1, 3)])
def test_substruct_context_matches_host_extraction(k, l1, l2):
    """ExtractSubstructureContextPair + BatchSubstructContext on the device == the host restatement
    (data/synthetic.py) applied per graph with the same roots, then collated -- every field, bit for bit,
    including which graphs are dropped for lack of a context / overlap (chem/batch.py:169)."""
    graphs = _chem_graphs(40, seed=11)
    lone = Data(x=torch.tensor([[5, 0]]), edge_index=torch.zeros(2, 0, dtype=torch.int64),
                edge_attr=torch.zeros(0, 2, dtype=torch.int64))
    graphs.insert(7, lone)  # an isolated atom: no context, must be dropped
    rng = np.random.default_rng(12)
    ids = rng.permutation(len(graphs))[:33]
    roots = [int(rng.integers(0, graphs[i].x.size(0))) for i in ids]
    want = hostdata.collate_substruct_context(
        [hostdata.extract_substruct_context(graphs[i], rng, k=k, l1=l1, l2=l2, root=r) for i, r in zip(ids, roots)])
    ds = resident.ResidentDataset.from_graphs(graphs, DEV, relabel=False)
    out = ds.collate_substruct_context(ids, k=k, l1=l1, l2=l2, roots=roots)
    ds.check(out)
    _same(out, want, CTX_KEYS)


def _big_graph(n, chords, seed):
    """a chain of n atoms with `chords` extra bonds (both directions stored, as the datasets have them)"""
    rng = np.random.default_rng(seed)
    pairs = [(i, i + 1) for i in range(n - 1)]
    while len(pairs) < n - 1 + chords:
        a, b = (int(v) for v in rng.integers(0, n, size=2))
        if a != b:
            pairs.append((a, b))
    src = [a for a, b in pairs] + [b for a, b in pairs]
    dst = [b for a, b in pairs] + [a for a, b in pairs]
    e = len(src)
    return Data(x=torch.stack([torch.from_numpy(rng.integers(0, 119, size=n)), torch.from_numpy(rng.integers(0, 3, size=n))], 1),
                edge_index=torch.tensor([src, dst], dtype=torch.int64),
                edge_attr=torch.stack([torch.from_numpy(rng.integers(0, 4, size=e)), torch.from_numpy(rng.integers(0, 3, size=e))], 1))


def test_substruct_context_on_graphs_beyond_the_lds_window():
    """k_ctx_plan keeps the BFS distances of a graph in its wave's LDS slice up to 1 024 nodes and a lane's first two bonds in
    registers (round 5); beyond, distances stay in global memory and further bonds are re-read: a 1 500-atom chain with chords
    (global path), a 900-atom one with 700 bonds (LDS path, eleven bonds per lane) and molecules in one batch, against the host
    extraction, every field bit for bit"""
    graphs = _chem_graphs(6, seed=31) + [_big_graph(1500, 40, 32), _big_graph(900, 250, 33), _big_graph(1024, 5, 34), _big_graph(1025, 5, 35)]
    rng = np.random.default_rng(36)
    ids = np.array([6, 0, 7, 3, 8, 9, 6, 5])
    roots = [int(rng.integers(0, graphs[i].x.size(0))) for i in ids]
    for k, l1, l2 in ((5, 4, 7), (9, 6, 12)):
        want = hostdata.collate_substruct_context(
            [hostdata.extract_substruct_context(graphs[i], rng, k=k, l1=l1, l2=l2, root=r) for i, r in zip(ids, roots)])
        ds = resident.ResidentDataset.from_graphs(graphs, DEV, relabel=False)
        out = ds.collate_substruct_context(ids, k=k, l1=l1, l2=l2, roots=roots)
        ds.check(out)
        _same(out, want, CTX_KEYS)


def test_substruct_context_all_graphs_dropped():
    """k < l1: substructure and context rings cannot overlap, every graph is dropped -> empty batch"""
    ds = resident.ResidentDataset.from_graphs(_chem_graphs(9, seed=14), DEV, relabel=False)
    out = ds.collate_substruct_context(np.arange(9), k=2, l1=3, l2=4, seed=1)
    ds.check(out)
    assert out.num_graphs == 0
    assert all(getattr(out, key).numel() == 0 for key in CTX_KEYS)
    assert out.edge_index_substruct.shape == (2, 0) and out.x_context.shape == (0, 2)


def test_substruct_context_random_roots_and_training_step():
    """random roots: reproducible per (seed, graph), valid (inside the graph), and the batch drives the
    context-prediction train step"""
    from pretrain_gnns_amd import train
    from pretrain_gnns_amd.chem import model as hmodel
    graphs = _chem_graphs(64, seed=13)
    ds = resident.ResidentDataset.from_graphs(graphs, DEV, relabel=False)
    ids = np.arange(64)
    a = ds.collate_substruct_context(ids, seed=3)
    b = ds.collate_substruct_context(ids, seed=3)
    c = ds.collate_substruct_context(ids, seed=4)
    assert torch.equal(a._roots, b._roots) and not torch.equal(a._roots, c._roots)
    sizes = torch.tensor([g.x.size(0) for g in graphs])
    assert bool(((a._roots.cpu() >= 0) & (a._roots.cpu() < sizes)).all())
    for key in CTX_KEYS:
        assert torch.equal(getattr(a, key), getattr(b, key)), key
    # same roots on the host give the same batch
    rng = np.random.default_rng(0)
    want = hostdata.collate_substruct_context(
        [hostdata.extract_substruct_context(g, rng, root=int(r)) for g, r in zip(graphs, a._roots.cpu())])
    _same(a, want, CTX_KEYS)
    torch.manual_seed(0)
    ms, mc = hmodel.GNN(5, 300).to(DEV), hmodel.GNN(3, 300).to(DEV)
    os_, oc = torch.optim.Adam(ms.parameters(), lr=1e-3), torch.optim.Adam(mc.parameters(), lr=1e-3)
    loss, acc = train.chem_contextpred_step(ms, mc, os_, oc, a)
    assert loss == loss and 0.0 <= acc <= 1.0


def test_substruct_context_loader_plans_ahead_with_the_same_batches(monkeypatch):
    """ResidentLoader(substruct_context=...) enqueues the plan of batch t + 1 (and the copy of its totals to pinned memory) in front of
    the work the consumer enqueues for batch t -- plan_substruct_context / fill_substruct_context, csrc/loader.hip unchanged -- so the one
    sync per step waits for nothing.  Same batches bit for bit as plan and fill back to back (PGNN_CTX_PIPELINE=0), over two epochs,
    with device work enqueued between the batches and a short last batch."""
    graphs = _chem_graphs(70, seed=17)
    ds = resident.ResidentDataset.from_graphs(graphs, DEV, relabel=False)
    runs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("PGNN_CTX_PIPELINE", flag)
        loader = resident.ResidentLoader(ds, 16, shuffle=True, seed=5, drop_last=False, substruct_context=(5, 4, 7))
        got = []
        for _ in range(2):
            for batch in loader:
                torch.randn(512, 512, device=DEV) @ torch.randn(512, 512, device=DEV)  # (what a train step would put between two batches)
                got.append({key: getattr(batch, key).clone() for key in CTX_KEYS})
        runs.append(got)
    monkeypatch.delenv("PGNN_CTX_PIPELINE")
    assert len(runs[0]) == len(runs[1]) == 2 * 5
    for a, b in zip(*runs):
        for key in CTX_KEYS:
            assert torch.equal(a[key], b[key]), key


def test_bio_resident_loader_drives_the_masking_step():
    """PPI-shaped dataset resident in HBM -> device MaskEdge batches -> bio masking train step"""
    from pretrain_gnns_amd import train
    from pretrain_gnns_amd.bio import model as hbio
    rng = np.random.default_rng(21)
    graphs = [synthetic.ppi_like_graph(rng) for _ in range(24)]
    ds = resident.ResidentDataset.from_graphs(graphs, DEV, relabel=False)
    loader = resident.ResidentLoader(ds, batch_size=8, shuffle=True, seed=2, mask_rate=0.15)
    torch.manual_seed(0)
    mods = [hbio.GNN(3, 300).to(DEV), torch.nn.Linear(300, 7).to(DEV)]
    opts = [torch.optim.Adam(m.parameters(), lr=1e-3) for m in mods]
    losses = []
    for batch in loader:
        ds.check(batch)
        assert batch.masked_edge_idx.numel() == batch.mask_edge_label.size(0) > 0
        loss, acc = train.bio_masking_step(mods, opts, batch)
        losses.append(loss)
        assert loss == loss and 0.0 <= acc <= 1.0
    assert len(losses) == 3


def test_from_inmemory_dataset_layout():
    """the (data, slices) pair an InMemoryDataset keeps (chem/loader.py: concatenated tensors + per-key slice
    vectors) maps one to one onto the resident dataset"""
    graphs = _chem_graphs(11, seed=31)

    class DataLike:
        pass

    data = DataLike()
    data.x = torch.cat([g.x for g in graphs], 0)
    data.edge_index = torch.cat([g.edge_index for g in graphs], 1)  # graph-local node ids, as InMemoryDataset stores them
    data.edge_attr = torch.cat([g.edge_attr for g in graphs], 0)
    ns = torch.tensor(np.cumsum([0] + [g.x.size(0) for g in graphs]))
    es = torch.tensor(np.cumsum([0] + [g.edge_index.size(1) for g in graphs]))
    slices = {"x": ns, "edge_index": es, "edge_attr": es}
    ds = resident.ResidentDataset.from_inmemory(data, slices, DEV, relabel=False)
    ids = [10, 0, 4, 4]
    out = ds.collate(ids)
    ds.check(out)
    _same(out, hostdata.collate([graphs[i] for i in ids]))
    with pytest.raises(ValueError):
        resident.ResidentDataset(data.x, data.edge_index, data.edge_attr, ns[:-1], es, DEV)


@pytest.mark.parametrize("order", ["smiles", "permuted"])
def test_relabelled_dataset_gives_the_same_sums_in_permuted_rows(order):
    """ResidentDataset(relabel=True) (data/relabel.py): the loader renumbers every graph's atoms once so that
    the aggregation kernel finds its source rows in LDS; only labels change, edge_index keeps its column order, so the GIN
    aggregation of chem/model.py:37-52 -- forward instance (CSR by destination) AND transposed instance (CSR by source, the
    backward's) -- gives BIT-identical rows, permuted: equal to the kernel on the dataset as fed, and equal to the oracle's
    sequential index_add_ on the dataset as fed.  The renumbered batch leaves (next to) no edge outside the kernel's window."""
    from oracle import chem as ochem
    from pretrain_gnns_amd import ops
    from pretrain_gnns_amd.data import relabel

    rng = np.random.default_rng(11)
    make = synthetic.zinc_like_graph_smiles if order == "smiles" else (lambda r: synthetic.zinc_like_graph(r, permute=True))
    graphs = [make(rng) for _ in range(300)]
    ids = rng.permutation(300)[:257]
    fed = resident.ResidentDataset.from_graphs(graphs, DEV, relabel=False)
    ren = resident.ResidentDataset.from_graphs(graphs, DEV, relabel=True)
    b0, b1 = fed.collate(ids), ren.collate(ids)
    perm = torch.from_numpy(ren.batch_rows_in_original_order(ids)).to(DEV)  # row j of b1 = row perm[j] of b0
    assert torch.equal(b1.x, b0.x[perm]) and torch.equal(b1.batch, b0.batch[perm]) and torch.equal(b1.edge_attr, b0.edge_attr)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(perm.numel(), device=DEV)
    assert torch.equal(b1.edge_index, inv[b0.edge_index])  # the same edges in the same columns, new labels

    def miss(b):
        lo = (b.edge_index[0] // 8) * 8 - 8
        return float(((b.edge_index[1] < lo) | (b.edge_index[1] >= lo + 24)).float().mean())
    assert miss(b1) < 0.003 < miss(b0)
    n = b0.x.size(0)
    torch.manual_seed(0)
    conv = ochem.GINConv(300)
    x0 = torch.randn(n, 300)
    want = conv.aggregate(x0, b0.edge_index.cpu(), b0.edge_attr.cpu()).detach()  # the oracle, on the batch as fed
    e1, e2 = conv.edge_embedding1.weight.detach().to(DEV), conv.edge_embedding2.weight.detach().to(DEV)
    x0 = x0.to(DEV)
    g0 = ops.build_chem_graph(b0.edge_index, b0.edge_attr, n)
    g1 = ops.build_chem_graph(b1.edge_index, b1.edge_attr, n)
    y0 = ops.ChemAggregate.apply(x0, e1, e2, g0)
    y1 = ops.ChemAggregate.apply(x0[perm].contiguous(), e1, e2, g1)
    assert torch.equal(y0.cpu(), want)
    assert torch.equal(y1, y0[perm])
    # the transposed instance (what the backward runs: sums over OUT-edges, self row last)
    lib, sp = ops.load(), ops.stream_ptr()
    t0, t1 = torch.empty(n, 300, device=DEV), torch.empty(n, 300, device=DEV)
    x1 = x0[perm].contiguous()
    ops.check(lib.pgnn_neighbor_sum(x0.data_ptr(), 300, g0.out_ptr.data_ptr(), g0.out_dst.data_ptr(), None, t0.data_ptr(), 300, n, 300, sp), "nsum")
    ops.check(lib.pgnn_neighbor_sum(x1.data_ptr(), 300, g1.out_ptr.data_ptr(), g1.out_dst.data_ptr(), None, t1.data_ptr(), 300, n, 300, sp), "nsum")
    assert torch.equal(t1, t0[perm])


def test_relabelled_dataset_drives_the_masking_step():
    """the masking train step on a renumbered dataset: MaskAtom draws, the head's gather and the loss act on rows of the
    renumbered batch -- the same graphs, the same number of masked atoms per graph, finite loss, accuracy in [0, 1]; and with the
    SAME atoms masked (explicit indices mapped through the renumbering) the node embeddings are the as-fed ones, permuted, to
    fp32 rounding (BatchNorm's column sums meet the rows in another order)"""
    from pretrain_gnns_amd.chem import model as hmodel

    rng = np.random.default_rng(3)
    graphs = [synthetic.zinc_like_graph_smiles(rng) for _ in range(96)]
    fed = resident.ResidentDataset.from_graphs(graphs, DEV, relabel=False)
    ren = resident.ResidentDataset.from_graphs(graphs, DEV, relabel=True)
    ids = np.arange(96)
    b0 = fed.collate(ids, mask_rate=0.15, seed=4)
    perm = torch.from_numpy(ren.batch_rows_in_original_order(ids)).to(DEV)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(perm.numel(), device=DEV)
    b1 = ren.collate(ids, masked_atom_indices=inv[b0.masked_atom_indices])
    assert torch.equal(b1.x, b0.x[perm]) and torch.equal(b1.mask_node_label, b0.mask_node_label)
    torch.manual_seed(0)
    net = hmodel.GNN(5, 300).to(DEV).eval()
    with torch.no_grad():
        h0 = net(b0.x, b0.edge_index, b0.edge_attr)
        h1 = net(b1.x, b1.edge_index, b1.edge_attr)
    assert torch.equal(h1, h0[perm])  # eval mode: no batch statistics, every row is a function of its own neighbourhood
    net.train()
    h0 = net(b0.x, b0.edge_index, b0.edge_attr)
    h1 = net(b1.x, b1.edge_index, b1.edge_attr)
    torch.testing.assert_close(h1, h0[perm], rtol=1e-4, atol=1e-4)
    b2 = ren.collate(ids, mask_rate=0.15, seed=4)
    assert b2.masked_atom_indices.numel() == b0.masked_atom_indices.numel()
    assert torch.equal(torch.bincount(b2.batch[b2.masked_atom_indices], minlength=96), torch.bincount(b0.batch[b0.masked_atom_indices], minlength=96))


def _struct_equal(a, b):
    n, e = a.n, a.e
    assert (a.n, a.e, a.kind, a.gcn) == (b.n, b.e, b.kind, b.gcn)
    for name, cnt in (("in_ptr", n + 1), ("out_ptr", n + 1), ("in_src", e), ("out_dst", e), ("in_code", e), ("dinv", n), ("cfeat", n)):
        assert torch.equal(getattr(a, name)[:cnt], getattr(b, name)[:cnt]), name


@pytest.mark.parametrize("relabel", [False, True])
def test_loader_structure_by_offset_add_equals_the_per_batch_build(relabel):
    """SURVEY 8f rank 1, the CSR half: the structure ``collate`` attaches (``pgnn_collate_structure``: slices of ONE structure over the
    whole dataset, shifted) is bit for bit what ``pgnn_chem_graph_build`` computes from the collated COO -- both CSRs, bond codes,
    normalisers, per-node bond counts -- on ragged batches: a lone atom (no bonds) in every position, a 700-bond hub, repeated
    graphs, more than 1024 graphs (the offset scan's carry)."""
    from pretrain_gnns_amd import ops
    lone = Data(x=torch.tensor([[5, 0]]), edge_index=torch.zeros(2, 0, dtype=torch.int64), edge_attr=torch.zeros(0, 2, dtype=torch.int64))
    pair = Data(x=torch.tensor([[6, 0], [7, 1]]), edge_index=torch.tensor([[0, 1], [1, 0]]), edge_attr=torch.tensor([[1, 0], [1, 0]]))
    hub_n = 351
    spokes = torch.arange(1, hub_n)
    hub_ei = torch.stack([torch.stack([torch.zeros_like(spokes), spokes], 1).view(-1), torch.stack([spokes, torch.zeros_like(spokes)], 1).view(-1)])
    hub = Data(x=torch.tensor([[5, 0]]).repeat(hub_n, 1), edge_index=hub_ei,
               edge_attr=torch.stack([torch.arange(hub_ei.size(1)) // 2 % 4, torch.arange(hub_ei.size(1)) // 2 % 3], 1))
    graphs = [lone, pair, hub] + _chem_graphs(40, seed=5)
    ds = resident.ResidentDataset.from_graphs(graphs, DEV, relabel=relabel)
    rng = np.random.default_rng(9)
    for ids in ([0], [0, 0], [2], [1, 0, 2, 0, 5, 5, 0], list(range(len(graphs))), rng.integers(0, len(graphs), size=1500)):
        out = ds.collate(ids, mask_rate=0.15, seed=3)
        ds.check(out)
        got = ops.attached_graph("chem", out.edge_index, out.edge_attr, out.x.size(0), False)
        assert got is not None
        want = ops.build_chem_graph(out.edge_index, out.edge_attr, out.x.size(0))
        _struct_equal(got, want)
        assert ops.build_chem_graph(out.edge_index, out.edge_attr, out.x.size(0), reuse=True) is got
        assert ops.attached_graph("chem", out.edge_index, out.edge_attr, out.x.size(0), True) is None       # a GCN model builds its own
        assert ops.attached_graph("chem", out.edge_index.clone(), out.edge_attr, out.x.size(0), False) is None  # another tensor
    plain = ds.collate([3, 4], structure=False)
    assert ops.attached_graph("chem", plain.edge_index, plain.edge_attr, plain.x.size(0), False) is None
    edged = ds.collate([3, 4, 2], mask_rate=0.15, seed=1, mask_edge=True)  # bond masking rewrites edge_attr: nothing attached
    assert ops.attached_graph("chem", edged.edge_index, edged.edge_attr, edged.x.size(0), False) is None
    out = ds.collate([3, 4])
    out.edge_attr[0, 0] = 2  # written to after the hand-over: stale, must not be used
    assert ops.attached_graph("chem", out.edge_index, out.edge_attr, out.x.size(0), False) is None


def test_model_forward_uses_the_loader_structure_and_gives_the_same_bits():
    """GNN.forward(x, edge_index, edge_attr) on a loader batch (structure attached) == on clones of the same tensors (structure
    built from the COO): embeddings and every gradient bit-identical, and the attached structure is really the one used."""
    from pretrain_gnns_amd import ops
    from pretrain_gnns_amd.chem import model as hchem
    ds = resident.ResidentDataset.from_graphs(_chem_graphs(64, seed=8), DEV)
    b = ds.collate(np.arange(64), mask_rate=0.15, seed=2)
    torch.manual_seed(3)
    m = hchem.GNN(5, 300).to(DEV)
    calls = []
    real = ops._build_graph
    ops._build_graph = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        y1 = m(b.x, b.edge_index, b.edge_attr)
        assert not calls
        y1.square().sum().backward()
        g1 = [p.grad.clone() for p in m.parameters()]
        m.zero_grad(set_to_none=True)
        y2 = m(b.x, b.edge_index.clone(), b.edge_attr.clone())
        assert calls
        y2.square().sum().backward()
    finally:
        ops._build_graph = real
    assert torch.equal(y1, y2)
    for a, p in zip(g1, m.parameters()):
        assert torch.equal(a, p.grad)


def test_prefetching_loader_hands_out_the_same_batches(monkeypatch):
    """ResidentLoader collates batch t + 1 on a side stream while step t runs (PGNN_LOADER_PREFETCH, default on): every batch, its
    masks and its attached structure are bit-identical to the in-line collate's, across two epochs, and a train loop over either
    loader ends in the same parameters."""
    from pretrain_gnns_amd import ops, optim, train as ptrain
    from pretrain_gnns_amd.chem import model as hchem
    ds = resident.ResidentDataset.from_graphs(_chem_graphs(100, seed=21), DEV)
    runs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("PGNN_LOADER_PREFETCH", flag)
        loader = resident.ResidentLoader(ds, 16, shuffle=True, seed=4, mask_rate=0.15)
        torch.manual_seed(5)
        mods = [hchem.GNN(5, 300).to(DEV), torch.nn.Linear(300, 119).to(DEV), torch.nn.Linear(300, 4).to(DEV)]
        opts = optim.Adam.shared([m.parameters() for m in mods], lr=1e-3)
        seen = []
        for _ in range(2):
            for b in loader:
                g = ops.attached_graph("chem", b.edge_index, b.edge_attr, b.x.size(0), False)
                assert g is not None
                seen.append([t.clone() for t in (b.x, b.edge_index, b.edge_attr, b.batch, b.masked_atom_indices, b.mask_node_label,
                                                 g.in_ptr, g.in_src[:g.e], g.in_code[:g.e], g.out_ptr, g.out_dst[:g.e], g.cfeat)])
                ptrain.chem_masking_step(mods, opts, b)
                ds.check(b)
        torch.cuda.synchronize()
        runs[flag] = (seen, [p.detach().clone() for p in mods[0].parameters()])
    assert len(runs["0"][0]) == len(runs["1"][0]) == 14
    for a, b in zip(runs["0"][0], runs["1"][0]):
        for ta, tb in zip(a, b):
            assert torch.equal(ta, tb)
    for pa, pb in zip(runs["0"][1], runs["1"][1]):
        assert torch.equal(pa, pb)

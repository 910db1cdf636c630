"""Seeded synthetic batches with the exact field layout the reference's collate code
hands to the hot path (SURVEY.md §8d is the binding definition of the shapes).

* ZINC-2M-shaped molecule graphs  -> ``BatchMasking`` layout (chem/batch.py:17-52) after a
  ``MaskAtom`` transform (chem/util.py:225-277), or the ``BatchSubstructContext`` layout
  (chem/batch.py:141-210) after an ``ExtractSubstructureContextPair`` transform
  (chem/util.py:96-149), or the plain PyG ``Batch`` layout used by chem/finetune.py:32.
* PPI-ego-shaped graphs -> bio ``BatchMasking`` layout (bio/batch.py:70-106) after ``MaskEdge``
  (bio/util.py:77-102), with ``center_node_idx`` for bio ``GNN_graphpred`` (bio/model.py:343).

Conventions kept from the reference data: int64 indices; the two directions of a bond are
adjacent columns of ``edge_index`` and carry identical attributes (chem/loader.py:83-86,
chem/util.py:212-213).  Everything here is host-side numpy; tensors are returned on CPU.
"""
import collections

import numpy as np
import torch

from .batch import Data

ATOM_MASK_TOKEN = 119  # chem/pretrain_masking.py:122 (num_atom_type = 119)
BOND_MASK_TOKEN = 5    # chem/pretrain_masking.py:122 (num_edge_type = 5)

_ATOM_TYPES = np.array([5, 6, 7, 8, 15, 16])
_ATOM_P = np.array([0.72, 0.12, 0.10, 0.02, 0.02, 0.02])
_CHIRAL_P = np.array([0.95, 0.025, 0.025])
_BOND_TYPES = np.array([0, 1, 2, 3])
_BOND_P = np.array([0.62, 0.08, 0.01, 0.29])
_BOND_DIR_P = np.array([0.97, 0.015, 0.015])


# ----------------------------------------------------------------------------- molecules
def zinc_like_graph(rng, parent_window=3, permute=False):
    """One ZINC-shaped molecule: random tree + a few ring closures, max degree 5.  SURVEY.md 8d's generator is the default
    (parent of atom i uniform in [i - 3, i)); ``parent_window`` widens that range and ``permute`` relabels the atoms at random --
    atom orders less local than the survey's, for bench.py's aggregation robustness leg (an RDKit atom order, chem/loader.py:53-100,
    is not the survey's generator)."""
    n = int(np.clip(np.rint(rng.normal(26.6, 6.0)), 6, 60))
    local = np.arange(1, n)
    parent = local - rng.integers(1, np.minimum(local, parent_window) + 1)
    bonds = [(int(p), int(c)) for p, c in zip(parent, local)]
    deg = np.bincount(np.concatenate([parent, local]), minlength=n)
    have = set(bonds)
    for _ in range(int(round(0.125 * n + 1))):
        u = int(rng.integers(0, n - 2))
        v = min(u + int(rng.integers(2, 7)), n - 1)
        if v - u < 2 or (u, v) in have or deg[u] >= 4 or deg[v] >= 4:
            continue
        have.add((u, v))
        bonds.append((u, v))
        deg[u] += 1
        deg[v] += 1
    b = np.asarray(bonds, dtype=np.int64)
    nb = b.shape[0]
    edge_index = np.empty((2, 2 * nb), dtype=np.int64)
    edge_index[0, 0::2], edge_index[1, 0::2] = b[:, 0], b[:, 1]
    edge_index[0, 1::2], edge_index[1, 1::2] = b[:, 1], b[:, 0]
    battr = np.stack([rng.choice(_BOND_TYPES, nb, p=_BOND_P), rng.choice(3, nb, p=_BOND_DIR_P)], axis=1)
    edge_attr = np.repeat(battr, 2, axis=0).astype(np.int64)
    x = np.stack([rng.choice(_ATOM_TYPES, n, p=_ATOM_P), rng.choice(3, n, p=_CHIRAL_P)], axis=1).astype(np.int64)
    if permute:  # new label of atom a = perm[a]; both directions of a bond stay adjacent
        perm = rng.permutation(n)
        edge_index = perm[edge_index]
        x_new = np.empty_like(x)
        x_new[perm] = x
        x = x_new
    return Data(x=torch.from_numpy(x), edge_index=torch.from_numpy(edge_index), edge_attr=torch.from_numpy(edge_attr))


def mask_atoms_at(data, idx, mask_edge=False, atom_token=ATOM_MASK_TOKEN, bond_token=BOND_MASK_TOKEN):
    """MaskAtom.__call__(data, masked_atom_indices) (chem/util.py:207-277) on a copy, for GIVEN atoms: label = the
    original feature rows in the order of ``idx``; masked rows := [atom_token, 0]; with ``mask_edge`` the bonds
    touching a masked atom := [bond_token, 0], their labels / indices taken from the first direction of each pair
    (``connected_edge_indices[::2]``, :255-268)."""
    data = data.clone()
    idx = torch.as_tensor(idx, dtype=torch.long)
    data.mask_node_label = data.x[idx].clone()
    data.masked_atom_indices = idx.clone()
    data.x[idx] = torch.tensor([atom_token, 0])
    if mask_edge:
        ei = data.edge_index.numpy()
        hit = np.isin(ei[0], idx.numpy()) | np.isin(ei[1], idx.numpy())
        connected = np.nonzero(hit)[0]
        if connected.size:
            first = connected[::2]
            data.mask_edge_label = data.edge_attr[first].clone()
            data.edge_attr[connected] = torch.tensor([bond_token, 0])
            data.connected_edge_indices = torch.from_numpy(first.astype(np.int64))
        else:
            data.mask_edge_label = torch.empty((0, 2), dtype=torch.int64)
            data.connected_edge_indices = torch.empty((0,), dtype=torch.int64)
    return data


def mask_atoms(data, rng, mask_rate=0.15, mask_edge=False):
    """MaskAtom with its random draw (chem/util.py:225-231): k = int(n*rate + 1) distinct atoms."""
    n = data.x.size(0)
    idx = rng.choice(n, int(n * mask_rate + 1), replace=False).astype(np.int64)
    return mask_atoms_at(data, idx, mask_edge)


def _bfs_dist(n, edge_index, root):
    adj = [[] for _ in range(n)]
    for u, v in zip(edge_index[0].tolist(), edge_index[1].tolist()):
        adj[u].append(v)
    dist = np.full(n, -1, dtype=np.int64)
    dist[root] = 0
    q = collections.deque([root])
    while q:
        u = q.popleft()
        for v in adj[u]:
            if dist[v] < 0:
                dist[v] = dist[u] + 1
                q.append(v)
    return dist


def _induced(data, keep):
    """Sub-graph on the sorted node set ``keep`` with nodes renumbered by rank (the effect of
    reset_idxes, chem/util.py:175-185); bond pairs keep their original relative order."""
    n = data.x.size(0)
    new_id = np.full(n, -1, dtype=np.int64)
    new_id[keep] = np.arange(keep.size)
    ei = data.edge_index.numpy()
    sel = (new_id[ei[0]] >= 0) & (new_id[ei[1]] >= 0)
    return (data.x[keep], torch.from_numpy(new_id[ei[:, sel]]), data.edge_attr[torch.from_numpy(sel)], new_id)


def extract_substruct_context(data, rng, k=5, l1=4, l2=7, root=None):
    """ExtractSubstructureContextPair (chem/util.py:96-149): substructure = nodes within k hops
    of a random root; context = nodes with l1 < dist <= l2; overlap = their intersection, indexed
    in the context graph's numbering.  Attributes are absent when the sets are empty."""
    data = data.clone()
    n = data.x.size(0)
    root = int(rng.integers(0, n)) if root is None else root
    dist = _bfs_dist(n, data.edge_index.numpy(), root)
    reach = dist >= 0
    sub = np.nonzero(reach & (dist <= k))[0]
    ctx = np.nonzero(reach & (dist > l1) & (dist <= l2))[0]
    if sub.size:
        data.x_substruct, data.edge_index_substruct, data.edge_attr_substruct, sub_id = _induced(data, sub)
        data.center_substruct_idx = torch.tensor([int(sub_id[root])])
    if ctx.size:
        data.x_context, data.edge_index_context, data.edge_attr_context, ctx_id = _induced(data, ctx)
        overlap = np.intersect1d(sub, ctx)
        if overlap.size:
            data.overlap_context_substruct_idx = torch.from_numpy(ctx_id[overlap])
    return data


# ----------------------------------------------------------------------------- PPI ego nets
def ppi_like_graph(rng):
    """One PPI-ego-shaped graph: G(n,p) with ~9.2 n undirected edges, 9-dim 0/1 edge attrs whose
    first 7 bits are the evidence channels (bio/loader.py:57-59), x = ones[n,1]."""
    n = int(np.clip(np.rint(rng.normal(39.8, 12.0)), 10, 120))
    p = min(0.9, 2 * 9.2 / (n - 1))
    iu, ju = np.triu_indices(n, 1)
    pick = rng.random(iu.size) < p
    if not pick.any():
        pick[0] = True
    u, v = iu[pick], ju[pick]
    ne = u.size
    bits = (rng.random((ne, 7)) < 0.3)
    empty = ~bits.any(axis=1)
    bits[empty, rng.integers(0, 7, int(empty.sum()))] = True
    attr = np.zeros((ne, 9), dtype=np.float32)
    attr[:, :7] = bits
    edge_index = np.empty((2, 2 * ne), dtype=np.int64)
    edge_index[0, 0::2], edge_index[1, 0::2] = u, v
    edge_index[0, 1::2], edge_index[1, 1::2] = v, u
    return Data(x=torch.ones(n, 1, dtype=torch.float32), edge_index=torch.from_numpy(edge_index),
                edge_attr=torch.from_numpy(np.repeat(attr, 2, axis=0)), center_node_idx=torch.tensor([0]))


def mask_edges_at(data, first):
    """MaskEdge.__call__(data, masked_edge_indices) (bio/util.py:55-110) on a copy, for GIVEN first-direction edge
    indices: label = original attr of the first direction; both directions := [0]*8 + [1]."""
    data = data.clone()
    first = torch.as_tensor(first, dtype=torch.long)
    data.masked_edge_idx = first.clone()
    data.mask_edge_label = data.edge_attr[first].clone()
    mask_row = torch.zeros(9)
    mask_row[8] = 1
    data.edge_attr[torch.cat([first, first + 1])] = mask_row
    return data


def mask_edges(data, rng, mask_rate=0.15):
    """MaskEdge with its random draw (bio/util.py:77-86): k = int(E/2*rate + 1) undirected edges."""
    num_edges = data.edge_index.size(1) // 2
    first = 2 * rng.choice(num_edges, int(num_edges * mask_rate + 1), replace=False).astype(np.int64)
    return mask_edges_at(data, first)


def bio_extract_substruct_context(data, l1=1):
    """bio ExtractSubstructureContextPair(l1, center=True) (bio/util.py:123-209): the substructure is the whole ego
    net; the context is the sub-graph induced on the nodes MORE than l1 hops from the centre node (unreachable
    ones included), its edge attributes rebuilt as [w1..w7, 0, 0] (bio/loader.py:56-68, 134); every context node is
    an overlap node.  Kept nodes are numbered by node index and bonds keep their original relative order (the
    reference takes both orders from networkx; tests compare as labelled graphs)."""
    data = data.clone()
    n = data.x.size(0)
    root = int(data.center_node_idx.item())
    dist = _bfs_dist(n, data.edge_index.numpy(), root)
    data.x_substruct, data.edge_attr_substruct = data.x, data.edge_attr
    data.edge_index_substruct, data.center_substruct_idx = data.edge_index, data.center_node_idx
    ctx = np.nonzero((dist < 0) | (dist > (l1 if l1 != 0 else -1)))[0]
    if ctx.size:
        data.x_context, data.edge_index_context, ea, ctx_id = _induced(data, ctx)
        ea = ea.clone()
        ea[:, 7:] = 0
        data.edge_attr_context = ea
        data.overlap_context_substruct_idx = torch.arange(ctx.size)
    return data


# ----------------------------------------------------------------------------- collate
_NODE_OFFSET_KEYS = ("edge_index", "masked_atom_indices", "center_node_idx", "negative_edge_index")
_EDGE_OFFSET_KEYS = ("connected_edge_indices", "masked_edge_idx")


def collate(graphs, shift_center=True):
    """BatchMasking.from_data_list (chem/batch.py:17-52, bio/batch.py:70-106): concatenate every
    key, shifting node-index keys by the node cumsum and edge-index keys by the edge cumsum.
    ``center_node_idx`` is shifted by bio BatchFinetune (bio/batch.py:41) but NOT by bio BatchMasking
    (bio/batch.py:93-96): ``shift_center=False`` gives the latter."""
    keys = sorted(set().union(*[set(g.keys) for g in graphs]))
    cols = {k: [] for k in keys}
    batch_vec, node_off, edge_off = [], 0, 0
    for i, g in enumerate(graphs):
        n = g.x.size(0)
        batch_vec.append(torch.full((n,), i, dtype=torch.long))
        for k in g.keys:
            item = getattr(g, k)
            if k in _NODE_OFFSET_KEYS and (shift_center or k != "center_node_idx"):
                item = item + node_off
            elif k in _EDGE_OFFSET_KEYS:
                item = item + edge_off
            cols[k].append(item)
        node_off += n
        edge_off += g.edge_index.size(1)
    out = Data()
    for k in keys:
        setattr(out, k, torch.cat(cols[k], dim=-1 if k in ("edge_index", "negative_edge_index") else 0))
    out.batch = torch.cat(batch_vec)
    return out.contiguous()


def collate_substruct_context(graphs):
    """BatchSubstructContext.from_data_list (chem/batch.py:141-210): graphs without a context are
    skipped (:169); substruct and context graphs are offset independently."""
    sub_keys = ("center_substruct_idx", "edge_attr_substruct", "edge_index_substruct", "x_substruct")
    ctx_keys = ("overlap_context_substruct_idx", "edge_attr_context", "edge_index_context", "x_context")
    shifted = {"edge_index_substruct", "edge_index_context", "overlap_context_substruct_idx", "center_substruct_idx"}
    cols = {k: [] for k in sub_keys + ctx_keys}
    overlap_batch, overlap_size = [], []
    off_sub = off_ctx = used = 0
    for g in graphs:
        if not hasattr(g, "x_context") or not hasattr(g, "overlap_context_substruct_idx"):
            continue
        m = len(g.overlap_context_substruct_idx)
        overlap_batch.append(torch.full((m,), used, dtype=torch.long))
        overlap_size.append(m)
        for k in sub_keys:
            item = getattr(g, k)
            cols[k].append(item + off_sub if k in shifted else item)
        for k in ctx_keys:
            item = getattr(g, k)
            cols[k].append(item + off_ctx if k in shifted else item)
        off_sub += g.x_substruct.size(0)
        off_ctx += g.x_context.size(0)
        used += 1
    out = Data()
    for k in sub_keys + ctx_keys:
        setattr(out, k, torch.cat(cols[k], dim=-1 if k.startswith("edge_index") else 0))
    out.batch_overlapped_context = torch.cat(overlap_batch)
    out.overlapped_context_size = torch.tensor(overlap_size, dtype=torch.long)
    return out.contiguous()


# ----------------------------------------------------------------------------- batch makers
def chem_masking_batch(num_graphs, seed=0, mask_rate=0.15, mask_edge=False):
    rng = np.random.default_rng(seed)
    return collate([mask_atoms(zinc_like_graph(rng), rng, mask_rate, mask_edge) for _ in range(num_graphs)])


def chem_plain_batch(num_graphs, seed=0):
    rng = np.random.default_rng(seed)
    return collate([zinc_like_graph(rng) for _ in range(num_graphs)])


def chem_finetune_batch(num_graphs, num_tasks=12, seed=0, missing=0.2):
    """labelled batch in the layout of the MoleculeNet datasets of chem/loader.py: per graph a row of
    ``num_tasks`` labels in {-1, +1}, 0 where the label is missing; ``y`` is their concatenation."""
    rng = np.random.default_rng(seed)
    graphs = []
    for _ in range(num_graphs):
        g = zinc_like_graph(rng)
        y = rng.choice([-1, 1], size=num_tasks)
        y[rng.random(num_tasks) < missing] = 0
        g.y = torch.from_numpy(y.astype(np.int64))
        graphs.append(g)
    return collate(graphs)


def negative_edges(data, rng):
    """NegativeEdge (chem/util.py:22-44): up to E/2 distinct directed non-bonded, non-loop atom pairs drawn
    from 5*E uniform candidates, in draw order."""
    data = data.clone()
    n, e = data.x.size(0), data.edge_index.size(1)
    have = set(zip(data.edge_index[0].tolist(), data.edge_index[1].tolist()))
    cand = rng.integers(0, n, size=(2, 5 * e))
    picked, seen = [], set()
    for i in range(5 * e):
        u, v = int(cand[0, i]), int(cand[1, i])
        if u != v and (u, v) not in have and (u, v) not in seen:
            seen.add((u, v))
            picked.append(i)
        if len(picked) == e // 2:
            break
    data.negative_edge_index = torch.from_numpy(cand[:, picked].astype(np.int64)).reshape(2, -1)
    return data


def chem_edgepred_batch(num_graphs, seed=0):
    """BatchAE layout (chem/batch.py:58-121): plain graphs + per-graph negative pairs shifted by the node cumsum."""
    rng = np.random.default_rng(seed)
    out = collate([negative_edges(zinc_like_graph(rng), rng) for _ in range(num_graphs)])
    return out


def chem_contextpred_batch(num_graphs, seed=0, num_layer=5, csize=3):
    """k = num_layer, l1 = num_layer-1, l2 = l1+csize (chem/pretrain_contextpred.py:145-152)."""
    rng = np.random.default_rng(seed)
    l1 = num_layer - 1
    return collate_substruct_context(
        [extract_substruct_context(zinc_like_graph(rng), rng, num_layer, l1, l1 + csize) for _ in range(num_graphs)])


def bio_masking_batch(num_graphs, seed=0, mask_rate=0.15):
    rng = np.random.default_rng(seed)
    return collate([mask_edges(ppi_like_graph(rng), rng, mask_rate) for _ in range(num_graphs)])


def tile_batch(batch, times):
    """Replicate a collated masking batch ``times`` times (offsetting every index key) -- a cheap
    way to reach roofline-sized inputs (>= 16384 graphs) without 16k python graph builds."""
    n, e = batch.x.size(0), batch.edge_index.size(1)
    g = int(batch.batch[-1].item()) + 1
    out = Data()
    for k in batch.keys:
        v = getattr(batch, k)
        if k in _NODE_OFFSET_KEYS:
            off, dim = n, -1
        elif k in _EDGE_OFFSET_KEYS:
            off, dim = e, 0
        elif k == "batch":
            off, dim = g, 0
        else:
            off, dim = 0, 0
        if k in ("masked_atom_indices", "center_node_idx"):
            dim = 0
        setattr(out, k, torch.cat([v + i * off if off else v for i in range(times)], dim=dim))
    return out.contiguous()

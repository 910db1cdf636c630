"""Host-side (numpy / torch CPU) restatements of the reference's data transforms and collate code -- TEST INFRASTRUCTURE, like the
rest of ``oracle/``: the checkers the device-side loader (pretrain_gnns_amd/csrc/loader.hip, pretrain_gnns_amd/data/resident.py) is
compared with, and the input builders of the CPU tests and of bench.py's ``cpu_baseline`` legs.  Nothing under ``pretrain_gnns_amd/``
imports this module.

* ``MaskAtom`` (chem/util.py:207-277), ``MaskEdge`` (bio/util.py:54-102), ``ExtractSubstructureContextPair`` (chem/util.py:96-185,
  bio/util.py:105-190), ``NegativeEdge`` (chem/util.py:22-44)
* ``BatchMasking`` / ``BatchSubstructContext`` / ``BatchAE`` ``from_data_list`` (chem/batch.py:17-52,141-210,58-121; bio/batch.py:70-106)
* seeded batch makers over the synthetic generators of ``pretrain_gnns_amd.data.synthetic``

Each is pinned to the reference's own classes, run unmodified through ``oracle/refshim``, by tests/test_cpu_reference.py.
"""
import collections

import numpy as np
import torch

from pretrain_gnns_amd.data.batch import Data
from pretrain_gnns_amd.data.synthetic import ATOM_MASK_TOKEN, BOND_MASK_TOKEN, ppi_like_graph, tile_batch, zinc_like_graph  # noqa: F401

_NODE_OFFSET_KEYS = ("edge_index", "masked_atom_indices", "center_node_idx", "negative_edge_index")
_EDGE_OFFSET_KEYS = ("connected_edge_indices", "masked_edge_idx")



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

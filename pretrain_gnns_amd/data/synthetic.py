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
chem/util.py:212-213).

The graph generators are host-side numpy.  The batch makers run the PRODUCT's loader: the graphs go to the GPU once
(``ResidentDataset``) and collate + masking / substructure extraction happen there (csrc/loader.hip), so they need the HIP
library and a GPU.  The host restatements of the same transforms (the checkers, and the builders of CPU-side inputs) live in
``oracle/hostdata.py``, outside the package.
"""
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


def _smiles_order_bonds(rng, n_target, p_ring=0.8, p_ring_branch=0.5, p_big=0.85, p_chain_branch=0.3):
    """atoms numbered in SMILES parse order, bonds in the order a parser meets them (the chain bond when an atom is written, the
    ring-closure bond when the closing digit is) -- the order RDKit keeps for ``mol.GetAtoms()`` / ``mol.GetBonds()`` and therefore
    the row order of ``x`` and the column order of ``edge_index`` that chem/loader.py:53-100 produces.  A writer puts a substituent
    in parentheses right behind the atom that carries it and the REST of the ring behind the substituent, so the bond back to the
    ring (and the ring closure) can span the whole subtree: in ``CC(C)(C)c1ccc2occ(CC(=O)Nc3ccccc3F)c2c1`` the bond ``c(`` -- ``c2``
    spans 14 atoms.  Parameters tuned against a sample of ZINC250k SMILES run through a token-level parser: bonds / atom 1.08,
    |u - v| = 1 for 0.74 of the bonds, >= 20 for 0.02-0.03, 0.05-0.06 of the directed edges outside the aggregation kernel's window."""
    bonds, deg = [], []

    def new_atom(parent):
        i = len(deg)
        deg.append(0)
        if parent is not None:
            bonds.append((parent, i))
            deg[parent] += 1
            deg[i] += 1
        return i

    def chain(parent, budget):
        tail = parent
        while budget > 0:
            if budget >= 5 and rng.random() < p_ring and (tail is None or deg[tail] < 4):
                r = min(6 if rng.random() < 0.75 else 5, budget)
                first = new_atom(tail)
                budget -= 1
                prev = first
                for k in range(1, r):
                    a = new_atom(prev)
                    budget -= 1
                    prev = a
                    avail = budget - (r - 1 - k)  # the ring's remaining atoms are written behind the substituent
                    if k < r - 1 and avail > 0 and rng.random() < p_ring_branch:
                        b = int(rng.integers(1, avail + 1)) if rng.random() < p_big else min(avail, int(rng.integers(1, 3)))
                        before = len(deg)
                        chain(a, b)
                        budget -= len(deg) - before
                bonds.append((first, prev))  # the closing digit
                deg[first] += 1
                deg[prev] += 1
                tail = prev
            else:
                a = new_atom(tail)
                budget -= 1
                if budget > 0 and rng.random() < p_chain_branch and deg[a] < 3:
                    b = min(budget, int(rng.integers(1, 4)))
                    before = len(deg)
                    chain(a, b)
                    budget -= len(deg) - before
                tail = a

    chain(None, n_target)
    return len(deg), np.asarray(bonds, dtype=np.int64)


def zinc_like_graph_smiles(rng):
    """One ZINC-shaped molecule with its atoms in SMILES parse order (``_smiles_order_bonds``) -- the order the reference's loader
    feeds the model (chem/loader.py:53-100), unlike SURVEY 8d's "parent within 3 rows" tree: branches return far from their parent.
    Same size law, attribute distributions and edge layout (both directions of a bond adjacent) as ``zinc_like_graph``."""
    n_target = int(np.clip(np.rint(rng.normal(26.6, 6.0)), 6, 60))
    n, b = _smiles_order_bonds(rng, n_target)
    nb = b.shape[0]
    edge_index = np.empty((2, 2 * nb), dtype=np.int64)
    edge_index[0, 0::2], edge_index[1, 0::2] = b[:, 0], b[:, 1]
    edge_index[0, 1::2], edge_index[1, 1::2] = b[:, 1], b[:, 0]
    battr = np.stack([rng.choice(_BOND_TYPES, nb, p=_BOND_P), rng.choice(3, nb, p=_BOND_DIR_P)], axis=1)
    edge_attr = np.repeat(battr, 2, axis=0).astype(np.int64)
    x = np.stack([rng.choice(_ATOM_TYPES, n, p=_ATOM_P), rng.choice(3, n, p=_CHIRAL_P)], axis=1).astype(np.int64)
    return Data(x=torch.from_numpy(x), edge_index=torch.from_numpy(edge_index), edge_attr=torch.from_numpy(edge_attr))


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


# ----------------------------------------------------------------------------- batch makers (device-side collate)
def _resident(graphs, device):
    from .resident import ResidentDataset
    return ResidentDataset.from_graphs(graphs, device)


def chem_masking_batch(num_graphs, seed=0, mask_rate=0.15, mask_edge=False, device="cuda"):
    """``num_graphs`` molecules collated and MaskAtom'ed on the device (BatchMasking layout, chem/batch.py:17-52)"""
    rng = np.random.default_rng(seed)
    ds = _resident([zinc_like_graph(rng) for _ in range(num_graphs)], device)
    return ds.collate(np.arange(num_graphs), mask_rate=mask_rate, seed=seed, mask_edge=mask_edge)


def chem_plain_batch(num_graphs, seed=0, device="cuda"):
    rng = np.random.default_rng(seed)
    return _resident([zinc_like_graph(rng) for _ in range(num_graphs)], device).collate(np.arange(num_graphs))


def chem_contextpred_batch(num_graphs, seed=0, num_layer=5, csize=3, device="cuda"):
    """k = num_layer, l1 = num_layer-1, l2 = l1+csize (chem/pretrain_contextpred.py:145-152); BatchSubstructContext layout"""
    rng = np.random.default_rng(seed)
    l1 = num_layer - 1
    ds = _resident([zinc_like_graph(rng) for _ in range(num_graphs)], device)
    return ds.collate_substruct_context(np.arange(num_graphs), k=num_layer, l1=l1, l2=l1 + csize, seed=seed)


def bio_masking_batch(num_graphs, seed=0, mask_rate=0.15, device="cuda"):
    """``num_graphs`` PPI-ego-shaped graphs collated and MaskEdge'd on the device (bio/batch.py:70-106, bio/util.py:77-102)"""
    rng = np.random.default_rng(seed)
    ds = _resident([ppi_like_graph(rng) for _ in range(num_graphs)], device)
    return ds.collate(np.arange(num_graphs), mask_rate=mask_rate, seed=seed)


def chem_aggregation_batch(graphs, order="smiles", relabel=True, seed=123, base_graphs=2048, device="cuda"):
    """the batch behind bench.py's aggregation roofline (and tools/agg_bench.py's PMC passes): ``base_graphs`` molecules in the
    given atom order -- "smiles" = SMILES parse order, what chem/loader.py:53-100 feeds the model; "survey" = SURVEY 8d's tree
    (parent within 3 rows); "permuted" = survey graphs with their atoms shuffled -- loaded as the product loads a dataset
    (``ResidentDataset``, ``relabel`` = its once-per-dataset Cuthill-McKee renumbering), collated, and tiled to ``graphs`` graphs.
    Returns (batch, info): info holds the fraction of directed edges outside the aggregation kernel's LDS window before / after."""
    from . import relabel as _relabel
    from .resident import ResidentDataset
    rng = np.random.default_rng(seed)
    make = {"smiles": zinc_like_graph_smiles, "survey": zinc_like_graph, "permuted": lambda r: zinc_like_graph(r, permute=True)}[order]
    gl = [make(rng) for _ in range(base_graphs)]
    ns = np.cumsum([0] + [g.x.size(0) for g in gl])
    es = np.cumsum([0] + [g.edge_index.size(1) for g in gl])
    ei = torch.cat([g.edge_index for g in gl], 1).numpy()
    ds = ResidentDataset.from_graphs(gl, device, relabel=relabel)
    info = {"atom_order": order, "relabelled_by_loader": bool(relabel),
            "out_of_window_edge_fraction_as_fed": round(_relabel.window_miss_fraction(ei, ns, es), 4)}
    base = ds.collate(np.arange(len(gl)))
    big = tile_batch(base, max(1, graphs // base_graphs)).to(device)
    dst, src = big.edge_index[0], big.edge_index[1]
    lo = (dst // 8) * 8 - 8
    info["out_of_window_edge_fraction"] = round(float(((src < lo) | (src >= lo + 24)).float().mean()), 4)
    return big, info


_NODE_OFFSET_KEYS = ("edge_index", "masked_atom_indices", "center_node_idx", "negative_edge_index")
_EDGE_OFFSET_KEYS = ("connected_edge_indices", "masked_edge_idx")


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

"""Reader of tests/golden/ref_*.npz -- outputs of the REFERENCE'S OWN code (oracle/refshim/make_fixtures.py).

`load(name)` returns the nested dict the generator saved, arrays as torch tensors (integers widened back to
int64, the 0/1 bio edge attributes back to float32).  `raw_graphs(fx["raw"])` rebuilds the per-graph objects.
"""
import os

import numpy as np
import torch

from pretrain_gnns_amd.data import synthetic
from oracle import hostdata

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_cache = {}


def _tensor(key, a):
    if a.ndim == 0:
        return a.item()
    t = torch.from_numpy(np.ascontiguousarray(a))
    if t.dtype == torch.int32:
        t = t.long()
    leaf = key.rsplit("/", 1)[-1]
    if t.dtype == torch.uint8 and (leaf.startswith("edge_attr") or leaf == "mask_edge_label"):
        t = t.float()
    return t


def load(name):
    if name not in _cache:
        tree = {}
        with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
            for key in z.files:
                node = tree
                parts = key.split("/")
                for p in parts[:-1]:
                    node = node.setdefault(p, {})
                node[parts[-1]] = _tensor(key, z[key])
        _cache[name] = tree
    return _cache[name]


class Bag:
    """attribute container with the few Data methods the step functions use"""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    @property
    def keys(self):
        return [k for k, v in self.__dict__.items() if v is not None]

    def __getitem__(self, k):
        return getattr(self, k)

    def to(self, device):
        return Bag(**{k: (v.to(device) if torch.is_tensor(v) else v) for k, v in self.__dict__.items()})

    def clone(self):
        return Bag(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.__dict__.items()})


def batch(tree):
    return Bag(**{k: v for k, v in tree.items() if torch.is_tensor(v)})


def raw_graphs(raw, bio=False):
    ns, es = raw["node_slices"].tolist(), raw["edge_slices"].tolist()
    out = []
    for i in range(len(ns) - 1):
        g = Bag(x=raw["x"][ns[i]:ns[i + 1]].clone(), edge_index=raw["edge_index"][:, es[i]:es[i + 1]].clone(),
                edge_attr=raw["edge_attr"][es[i]:es[i + 1]].clone())
        if bio:
            g.x = g.x.float()
            g.center_node_idx = torch.tensor([0])
        out.append(g)
    return out


def ragged(tree, i):
    s = tree["slices"].tolist()
    return tree["values"][s[i]:s[i + 1]]


def unpack_params(tree):
    """name -> (norm, positions or None, values) from pack_params"""
    out = {}
    for key, v in tree.items():
        name, kind = key.rsplit("|", 1)
        out.setdefault(name, {})[kind] = v
    return out


def check_params(named, tree, what, rtol, norm_rtol=None):
    """compare every tensor of `named` (after `what`) with the packed reference values, elementwise on the stored
    entries: |err| <= rtol * (|ref| + 1e-2 * max|ref of this tensor| + 1e-1 * max|ref of any tensor|), and on the
    L2 norm.  The last term is the noise floor for tensors that are mathematically zero (the gradient of a bias
    that feeds a BatchNorm is pure fp32 rounding, ~1e-7 of the largest gradient)."""
    ref = unpack_params(tree)
    top = max(float((r["full"] if "full" in r else r["val"]).abs().max()) for r in ref.values())
    seen = 0
    for name, t in named:
        t = what(t)
        if t is None or name not in ref:
            continue
        r = ref[name]
        flat = t.detach().reshape(-1).cpu()
        got, want = (flat, r["full"]) if "full" in r else (flat[r["pos"]], r["val"])
        scale = float(want.abs().max())
        err = (got - want).abs()
        bound = rtol * (want.abs() + 1e-2 * scale + 1e-1 * top)
        assert bool((err <= bound).all()), "%s: max err %.3e (tensor scale %.3e, global %.3e)" % (name, float(err.max()), scale, top)
        n = float(flat.double().norm())
        assert abs(n - r["norm"]) <= (norm_rtol or rtol) * (r["norm"] + top), "%s norm %g vs %g" % (name, n, r["norm"])
        seen += 1
    assert seen > 0
    return seen


# ----------------------------------------------------------------------------- rebuilding the reference's inputs
def masked_batches(fx, tag, mask_edge, collate=hostdata.collate):
    """rebuild the batches the reference's loader produced: raw graphs + the stored per-graph atom choices through
    the HOST restatements (hostdata.mask_atoms semantics with explicit indices, hostdata.collate)"""
    raw = raw_graphs(fx["raw"])
    counts, local = fx[tag]["mask_counts"].tolist(), fx[tag]["mask_local"]
    graphs, pos = [], 0
    for g, k in zip(raw, counts):
        d = synthetic.Data(x=g.x, edge_index=g.edge_index, edge_attr=g.edge_attr)
        graphs.append(hostdata.mask_atoms_at(d, local[pos:pos + k], mask_edge=mask_edge))
        pos += k
    bs = int(fx["batch_size"])
    return [collate(graphs[i:i + bs]) for i in range(0, len(graphs), bs)]


def context_graphs(fx, k=5, l1=4, l2=7):
    raw = raw_graphs(fx["raw"])
    return [hostdata.extract_substruct_context(synthetic.Data(x=g.x, edge_index=g.edge_index, edge_attr=g.edge_attr), None, k, l1, l2,
                                                root=int(r)) for g, r in zip(raw, fx["roots"].tolist())]


def bfs(fx, i):
    g = raw_graphs(fx["raw"])[i]
    d = hostdata._bfs_dist(g.x.size(0), g.edge_index.numpy(), int(fx["roots"][i]))
    return np.where(d < 0, 10 ** 6, d)


def assert_same_labelled_graphs(fx, graphs, want):
    """per graph: map both numberings back to molecule atom ids and compare node features, the edge multiset
    (with attributes), the centre and the overlap set"""
    used = [i for i, g in enumerate(graphs) if hasattr(g, "x_context")]
    ns = np.cumsum([0] + [graphs[i].x_substruct.size(0) for i in used])
    nc = np.cumsum([0] + [graphs[i].x_context.size(0) for i in used])
    raw = raw_graphs(fx["raw"])
    eis, eic = want["edge_index_substruct"], want["edge_index_context"]
    for j, i in enumerate(used):
        for part, order, off, ei_all, ea_all, x_all in (
                ("substruct", ragged(fx["sub_order"], i), ns, eis, want["edge_attr_substruct"], want["x_substruct"]),
                ("context", ragged(fx["ctx_order"], i), nc, eic, want["edge_attr_context"], want["x_context"])):
            lo, hi = int(off[j]), int(off[j + 1])
            assert hi - lo == len(order)
            assert torch.equal(x_all[lo:hi], raw[i].x[order])  # reference numbering -> atom ids
            sel = (ei_all[0] >= lo) & (ei_all[0] < hi)
            ref_edges = sorted((int(order[u - lo]), int(order[v - lo]), tuple(a.tolist()))
                               for (u, v), a in zip(ei_all[:, sel].t().tolist(), ea_all[sel]))
            mine = graphs[i]
            kept = np.sort(order.numpy())
            mei = getattr(mine, "edge_index_" + part)
            my_edges = sorted((int(kept[u]), int(kept[v]), tuple(a.tolist()))
                              for (u, v), a in zip(mei.t().tolist(), getattr(mine, "edge_attr_" + part)))
            assert ref_edges == my_edges, (i, part)


def bio_batches(fx):
    raw = raw_graphs(fx["raw"], bio=True)
    counts, local = fx["mask_counts"].tolist(), fx["mask_local"]
    graphs, pos = [], 0
    for g, k in zip(raw, counts):
        d = synthetic.Data(x=g.x, edge_index=g.edge_index, edge_attr=g.edge_attr, center_node_idx=g.center_node_idx)
        graphs.append(hostdata.mask_edges_at(d, local[pos:pos + k]))
        pos += k
    bs = int(fx["batch_size"])
    return [hostdata.collate(graphs[i:i + bs], shift_center=False) for i in range(0, len(graphs), bs)]


def assert_same_bio_context(fx, raw, graphs, want):
    """the context graphs as labelled graphs: reference numbering (networkx order, stored) vs node-index order"""
    used = [i for i, g in enumerate(graphs) if hasattr(g, "x_context")]
    nc = np.cumsum([0] + [graphs[i].x_context.size(0) for i in used])
    ei_all, ea_all = want["edge_index_context"], want["edge_attr_context"]
    ov = want["overlap_context_substruct_idx"]
    assert sorted(ov.tolist()) == list(range(int(nc[-1])))  # every context node is an overlap node
    for j, i in enumerate(used):
        order = ragged(fx["ctx_order"], i)
        lo, hi = int(nc[j]), int(nc[j + 1])
        assert hi - lo == len(order)
        sel = (ei_all[0] >= lo) & (ei_all[0] < hi)
        ref_edges = sorted((int(order[u - lo]), int(order[v - lo]), tuple(a.tolist()))
                           for (u, v), a in zip(ei_all[:, sel].t().tolist(), ea_all[sel]))
        kept = np.sort(order.numpy())
        g = graphs[i]
        my_edges = sorted((int(kept[u]), int(kept[v]), tuple(a.tolist()))
                          for (u, v), a in zip(g.edge_index_context.t().tolist(), g.edge_attr_context))
        assert ref_edges == my_edges, i




def edgepred_batches(fx, bio=False):
    """raw graphs + the reference's stored NegativeEdge draws -> BatchAE layout through the host collate (bio/batch.py:147-154 shifts
    edge_index and negative_edge_index only: ``center_node_idx`` stays graph-local)"""
    raw = raw_graphs(fx["raw"], bio=bio)
    graphs = []
    for i, g in enumerate(raw):
        neg = ragged(fx["neg"], i).reshape(-1, 2).t().contiguous()
        extra = {"center_node_idx": g.center_node_idx} if bio else {}
        graphs.append(synthetic.Data(x=g.x, edge_index=g.edge_index, edge_attr=g.edge_attr, negative_edge_index=neg, **extra))
    bs = int(fx["batch_size"])
    return [hostdata.collate(graphs[i:i + bs], shift_center=False) for i in range(0, len(graphs), bs)]


def plain_batches(fx, bio=False):
    raw = raw_graphs(fx["raw"], bio=bio)
    graphs = [synthetic.Data(x=g.x, edge_index=g.edge_index, edge_attr=g.edge_attr) for g in raw]
    bs = int(fx["batch_size"])
    return [hostdata.collate(graphs[i:i + bs]) for i in range(0, len(graphs), bs)]


def bio_finetune_batches(fx):
    """BatchFinetune layout (bio/batch.py:4-50): centre node indices shifted by the node offset, labels concatenated"""
    raw = raw_graphs(fx["raw"], bio=True)
    graphs = [synthetic.Data(x=g.x, edge_index=g.edge_index, edge_attr=g.edge_attr, center_node_idx=g.center_node_idx, go_target_downstream=y)
              for g, y in zip(raw, fx["y"])]
    bs = int(fx["batch_size"])
    return [hostdata.collate(graphs[i:i + bs], shift_center=True) for i in range(0, len(graphs), bs)]

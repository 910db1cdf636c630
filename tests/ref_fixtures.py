"""Reader of tests/golden/ref_*.npz -- outputs of the REFERENCE'S OWN code (oracle/refshim/make_fixtures.py).

`load(name)` returns the nested dict the generator saved, arrays as torch tensors (integers widened back to
int64, the 0/1 bio edge attributes back to float32).  `raw_graphs(fx["raw"])` rebuilds the per-graph objects.
"""
import os

import numpy as np
import torch

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

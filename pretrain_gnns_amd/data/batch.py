"""Minimal attribute container standing in for ``torch_geometric.data.Data`` / the
reference's ``Batch*`` classes (chem/batch.py:4-52,124-228; bio/batch.py:58-121).

Only what the hot path and the reference ``train()`` bodies touch is provided:
attribute access, ``keys``, ``to(device)``, ``contiguous()``, ``num_nodes`` and
``num_graphs``.  The device-side collate that fills it is ``resident.py`` (csrc/loader.hip); the host restatement is ``oracle/hostdata.py``.
"""
import torch


class Data:
    def __init__(self, **fields):
        for k, v in fields.items():
            setattr(self, k, v)

    @property
    def keys(self):
        return [k for k, v in self.__dict__.items() if v is not None and not k.startswith("_")]

    def __getitem__(self, key):
        return getattr(self, key)

    def __setitem__(self, key, value):
        setattr(self, key, value)

    def __contains__(self, key):
        return key in self.keys

    def __iter__(self):
        for k in sorted(self.keys):
            yield k, getattr(self, k)

    def apply(self, fn):
        for k in self.keys:
            v = getattr(self, k)
            if torch.is_tensor(v):
                setattr(self, k, fn(v))
        return self

    def to(self, device, non_blocking=False):
        return self.apply(lambda t: t.to(device, non_blocking=non_blocking))

    def contiguous(self):
        return self.apply(lambda t: t.contiguous())

    def clone(self):
        out = self.__class__()
        for k in self.keys:
            v = getattr(self, k)
            setattr(out, k, v.clone() if torch.is_tensor(v) else v)
        return out

    @property
    def num_nodes(self):
        x = getattr(self, "x", None)
        return None if x is None else x.size(0)

    @property
    def num_edges(self):
        ei = getattr(self, "edge_index", None)
        return None if ei is None else ei.size(1)

    @property
    def num_graphs(self):
        known = self.__dict__.get("_num_graphs")  # set by collates that know it (avoids a device sync)
        if known is not None:
            return known
        b = getattr(self, "batch", None)
        return None if b is None else int(b[-1].item()) + 1

    def __repr__(self):
        body = ", ".join("%s=%s" % (k, list(v.shape) if torch.is_tensor(v) else v) for k, v in self)
        return "%s(%s)" % (self.__class__.__name__, body)

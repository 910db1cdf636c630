"""Stand-ins for the third-party modules the reference imports, so that its UNMODIFIED sources
(/root/reference/{chem,bio}/*.py) can be imported and executed in this container.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  This file is the only restated surface between
the reference's own code and the fixtures under tests/golden/ref_*: everything else that runs when
those fixtures are generated is the reference's source, byte for byte.

What is restated, from the published sources of the versions pinned in /root/reference/requirements.txt
(torch-geometric==1.0.3, torch-scatter==1.1.2), restricted to what the reference calls:

  torch_scatter.scatter_add / scatter_mean / scatter_max   (chem/model.py:6,78; via PyG's scatter_)
  torch_geometric.utils.{add_self_loops, degree, softmax, scatter_}
  torch_geometric.nn.MessagePassing.propagate              (chem/model.py:49,101; bio/model.py:52,114)
  torch_geometric.nn.{global_add_pool, global_mean_pool, global_max_pool, GlobalAttention, Set2Set}
  torch_geometric.nn.inits.{uniform, glorot, zeros}
  torch_geometric.data.{Data, Batch, DataLoader, Dataset, InMemoryDataset}

Inert placeholders (imported by the reference at module top, never reached on the hot path):
  rdkit.*, tensorboardX.SummaryWriter.
"""
import inspect
import math
import re
import sys
import types

import torch


# ----------------------------------------------------------------------------- torch_scatter 1.1.2
def _gen(src, index, dim, out, dim_size, fill_value):
    """torch_scatter/utils/gen.py: broadcast a 1-D index over src, allocate out filled with fill_value."""
    dim = range(src.dim())[dim]
    if index.dim() == 1:
        index_size = [1] * src.dim()
        index_size[dim] = src.size(dim)
        index = index.view(index_size).expand_as(src)
    if out is None:
        dim_size = int(index.max().item()) + 1 if dim_size is None else dim_size
        out_size = list(src.size())
        out_size[dim] = dim_size
        out = src.new_full(out_size, fill_value)
    return src, out, index, dim


def scatter_add(src, index, dim=-1, out=None, dim_size=None, fill_value=0):
    src, out, index, dim = _gen(src, index, dim, out, dim_size, fill_value)
    return out.scatter_add_(dim, index, src)  # CPU: sequential in index order


def scatter_mean(src, index, dim=-1, out=None, dim_size=None, fill_value=0):
    out = scatter_add(src, index, dim, out, dim_size, fill_value)
    count = scatter_add(torch.ones_like(src), index, dim, None, out.size(dim))
    return out / count.clamp(min=1)


def scatter_max(src, index, dim=-1, out=None, dim_size=None, fill_value=0):
    src, out, index, dim = _gen(src, index, dim, out, dim_size, fill_value)
    out = out.scatter_reduce(dim, index, src, reduce="amax", include_self=True)
    return out, None  # the argmax output is never used by the reference


# ----------------------------------------------------------------------------- torch_geometric.utils
def maybe_num_nodes(index, num_nodes=None):
    return int(index.max().item()) + 1 if num_nodes is None else num_nodes


def add_self_loops(edge_index, num_nodes=None):
    """1.0.3: returns ONE tensor; the N loops are appended after the real edges."""
    num_nodes = maybe_num_nodes(edge_index, num_nodes)
    loop = torch.arange(0, num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    loop = loop.unsqueeze(0).repeat(2, 1)
    return torch.cat([edge_index, loop], dim=1)


def degree(index, num_nodes=None, dtype=None):
    num_nodes = maybe_num_nodes(index, num_nodes)
    out = torch.zeros((num_nodes,), dtype=dtype, device=index.device)
    return out.scatter_add_(0, index, out.new_ones((index.size(0))))


def softmax(src, index, num_nodes=None):
    num_nodes = maybe_num_nodes(index, num_nodes)
    out = src - scatter_max(src, index, dim=0, dim_size=num_nodes)[0][index]
    out = out.exp()
    return out / (scatter_add(out, index, dim=0, dim_size=num_nodes)[index] + 1e-16)


def scatter_(name, src, index, dim_size=None):
    assert name in ["add", "mean", "max"]
    op = {"add": scatter_add, "mean": scatter_mean, "max": scatter_max}[name]
    fill_value = -1e38 if name == "max" else 0
    out = op(src, index, 0, None, dim_size, fill_value)
    if isinstance(out, tuple):
        out = out[0]
    if name == "max":
        out[out == fill_value] = 0
    return out


# ----------------------------------------------------------------------------- torch_geometric.nn
class MessagePassing(torch.nn.Module):
    """1.0.3 message passing: `_i` arguments are gathered with edge_index[0], `_j` with
    edge_index[1]; messages are reduced at edge_index[0]."""

    def __init__(self):
        super(MessagePassing, self).__init__()
        self.message_args = inspect.getfullargspec(self.message)[0][1:]
        self.update_args = inspect.getfullargspec(self.update)[0][2:]

    def propagate(self, aggr, edge_index, **kwargs):
        assert aggr in ["add", "mean", "max"]
        kwargs["edge_index"] = edge_index
        size = None
        message_args = []
        for arg in self.message_args:
            if arg[-2:] == "_i":
                tmp = kwargs[arg[:-2]]
                size = tmp.size(0)
                message_args.append(tmp[edge_index[0]])
            elif arg[-2:] == "_j":
                tmp = kwargs[arg[:-2]]
                size = tmp.size(0)
                message_args.append(tmp[edge_index[1]])
            else:
                message_args.append(kwargs[arg])
        update_args = [kwargs[arg] for arg in self.update_args]
        out = self.message(*message_args)
        out = scatter_(aggr, out, edge_index[0], dim_size=size)
        return self.update(out, *update_args)

    def message(self, x_j):
        return x_j

    def update(self, aggr_out):
        return aggr_out


def global_add_pool(x, batch, size=None):
    size = int(batch.max().item()) + 1 if size is None else size
    return scatter_("add", x, batch, dim_size=size)


def global_mean_pool(x, batch, size=None):
    size = int(batch.max().item()) + 1 if size is None else size
    return scatter_("mean", x, batch, dim_size=size)


def global_max_pool(x, batch, size=None):
    size = int(batch.max().item()) + 1 if size is None else size
    return scatter_("max", x, batch, dim_size=size)


class GlobalAttention(torch.nn.Module):
    def __init__(self, gate_nn, nn=None):
        super(GlobalAttention, self).__init__()
        self.gate_nn = gate_nn
        self.nn = nn

    def forward(self, x, batch, size=None):
        x = x.unsqueeze(-1) if x.dim() == 1 else x
        size = int(batch[-1].item()) + 1 if size is None else size
        gate = self.gate_nn(x).view(-1, 1)
        x = self.nn(x) if self.nn is not None else x
        gate = softmax(gate, batch, size)
        return scatter_add(gate * x, batch, dim=0, dim_size=size)


class Set2Set(torch.nn.Module):
    def __init__(self, in_channels, processing_steps, num_layers=1):
        super(Set2Set, self).__init__()
        self.in_channels = in_channels
        self.out_channels = 2 * in_channels
        self.processing_steps = processing_steps
        self.num_layers = num_layers
        self.lstm = torch.nn.LSTM(self.out_channels, self.in_channels, num_layers)
        self.lstm.reset_parameters()

    def forward(self, x, batch):
        batch_size = int(batch.max().item()) + 1
        h = (x.new_zeros((self.num_layers, batch_size, self.in_channels)),
             x.new_zeros((self.num_layers, batch_size, self.in_channels)))
        q_star = x.new_zeros(batch_size, self.out_channels)
        for _ in range(self.processing_steps):
            q, h = self.lstm(q_star.unsqueeze(0), h)
            q = q.view(batch_size, self.in_channels)
            e = (x * q[batch]).sum(dim=-1, keepdim=True)
            a = softmax(e, batch, batch_size)
            r = scatter_add(a * x, batch, dim=0, dim_size=batch_size)
            q_star = torch.cat([q, r], dim=-1)
        return q_star


def uniform(size, tensor):
    bound = 1.0 / math.sqrt(size)
    if tensor is not None:
        tensor.data.uniform_(-bound, bound)


def glorot(tensor):
    stdv = math.sqrt(6.0 / (tensor.size(-2) + tensor.size(-1)))
    if tensor is not None:
        tensor.data.uniform_(-stdv, stdv)


def zeros(tensor):
    if tensor is not None:
        tensor.data.fill_(0)


# ----------------------------------------------------------------------------- torch_geometric.data
class Data(object):
    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, pos=None):
        self.x = x
        self.edge_index = edge_index
        self.edge_attr = edge_attr
        self.y = y
        self.pos = pos

    @staticmethod
    def from_dict(dictionary):
        data = Data()
        for key, item in dictionary.items():
            data[key] = item
        return data

    def __getitem__(self, key):
        return getattr(self, key)

    def __setitem__(self, key, item):
        setattr(self, key, item)

    @property
    def keys(self):
        return [key for key in self.__dict__.keys() if self[key] is not None]

    def __len__(self):
        return len(self.keys)

    def __contains__(self, key):
        return key in self.keys

    def __iter__(self):
        for key in sorted(self.keys):
            yield key, self[key]

    def __call__(self, *keys):
        for key in sorted(self.keys) if not keys else keys:
            if self[key] is not None:
                yield key, self[key]

    def cat_dim(self, key, item):
        return -1 if bool(re.search("(index|face)", key)) else 0

    def cumsum(self, key, item):
        return bool(re.search("(index|face)", key))

    @property
    def num_nodes(self):
        for key, item in self("x", "pos"):
            return item.size(self.cat_dim(key, item))
        if self.edge_index is not None:
            return maybe_num_nodes(self.edge_index)
        return None

    @property
    def num_edges(self):
        for key, item in self("edge_index", "edge_attr"):
            return item.size(self.cat_dim(key, item))
        return None

    @property
    def num_features(self):
        return 1 if self.x.dim() == 1 else self.x.size(1)

    def apply(self, func, *keys):
        for key, item in self(*keys):
            if torch.is_tensor(item):
                self[key] = func(item)
        return self

    def contiguous(self, *keys):
        return self.apply(lambda x: x.contiguous(), *keys)

    def to(self, device, *keys):
        return self.apply(lambda x: x.to(device), *keys)

    def __repr__(self):
        info = ["{}={}".format(key, list(item.size())) for key, item in self if torch.is_tensor(item)]
        return "{}({})".format(self.__class__.__name__, ", ".join(info))


class Batch(Data):
    def __init__(self, batch=None, **kwargs):
        super(Batch, self).__init__(**kwargs)
        self.batch = batch

    @staticmethod
    def from_data_list(data_list):
        keys = [set(data.keys) for data in data_list]
        keys = list(set.union(*keys))
        assert "batch" not in keys
        batch = Batch()
        for key in keys:
            batch[key] = []
        batch.batch = []
        cumsum = 0
        for i, data in enumerate(data_list):
            num_nodes = data.num_nodes
            batch.batch.append(torch.full((num_nodes,), i, dtype=torch.long))
            for key in data.keys:
                item = data[key]
                item = item + cumsum if data.cumsum(key, item) else item
                batch[key].append(item)
            cumsum += num_nodes
        for key in keys:
            item = batch[key][0]
            if torch.is_tensor(item):
                batch[key] = torch.cat(batch[key], dim=data_list[0].cat_dim(key, item))
            elif isinstance(item, (int, float)):
                batch[key] = torch.tensor(batch[key])
            else:
                raise ValueError("Unsupported attribute type.")
        batch.batch = torch.cat(batch.batch, dim=-1)
        return batch.contiguous()

    @property
    def num_graphs(self):
        return self.batch[-1].item() + 1


class DataLoader(torch.utils.data.DataLoader):
    def __init__(self, dataset, batch_size=1, shuffle=True, **kwargs):
        super(DataLoader, self).__init__(dataset, batch_size, shuffle,
                                         collate_fn=lambda data_list: Batch.from_data_list(data_list), **kwargs)


class Dataset(torch.utils.data.Dataset):
    """placeholder base: the reference's dataset classes are only imported, never instantiated here"""


class InMemoryDataset(Dataset):
    pass


# ----------------------------------------------------------------------------- inert placeholders
class _Inert(types.ModuleType):
    """a module whose every attribute is another inert object (rdkit, tensorboardX)"""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        child = _Inert(self.__name__ + "." + name)
        setattr(self, name, child)
        return child

    def __call__(self, *a, **k):
        return _Inert(self.__name__ + "()")


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


def build_modules():
    """name -> module for everything install() puts into sys.modules"""
    here = sys.modules[__name__]
    mods = {}
    mods["torch_scatter"] = _module("torch_scatter", scatter_add=scatter_add, scatter_mean=scatter_mean,
                                    scatter_max=scatter_max)
    utils_convert = _module("torch_geometric.utils.convert")
    utils = _module("torch_geometric.utils", add_self_loops=add_self_loops, degree=degree, softmax=softmax,
                    scatter_=scatter_, convert=utils_convert)
    inits = _module("torch_geometric.nn.inits", uniform=uniform, glorot=glorot, zeros=zeros)
    nn = _module("torch_geometric.nn", MessagePassing=MessagePassing, global_add_pool=global_add_pool,
                 global_mean_pool=global_mean_pool, global_max_pool=global_max_pool,
                 GlobalAttention=GlobalAttention, Set2Set=Set2Set, inits=inits)
    data = _module("torch_geometric.data", Data=Data, Batch=Batch, DataLoader=DataLoader, Dataset=Dataset,
                   InMemoryDataset=InMemoryDataset)
    tg = _module("torch_geometric", utils=utils, nn=nn, data=data, __version__="1.0.3")
    tg.__path__ = []
    mods.update({"torch_geometric": tg, "torch_geometric.utils": utils, "torch_geometric.utils.convert": utils_convert,
                 "torch_geometric.nn": nn, "torch_geometric.nn.inits": inits, "torch_geometric.data": data})
    for name in ("rdkit", "rdkit.Chem", "rdkit.Chem.AllChem", "rdkit.Chem.Descriptors", "rdkit.Chem.rdMolDescriptors",
                 "rdkit.DataStructs", "rdkit.Chem.Scaffolds", "rdkit.Chem.Scaffolds.MurckoScaffold", "tensorboardX"):
        m = _Inert(name)
        m.__path__ = []
        mods[name] = m
    for name in list(mods):  # parents expose children as attributes
        if "." in name and name.split(".")[0] in ("rdkit",):
            parent, child = name.rsplit(".", 1)
            setattr(mods[parent], child, mods[name])
    del here
    return mods

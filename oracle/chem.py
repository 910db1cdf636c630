"""CPU oracle of the chem (molecule) GNN stack.  Test infrastructure only.

Restates /root/reference/chem/model.py: GINConv :15-55, GCNConv :58-104, GNN :206-290,
GNN_graphpred :293-369, with torch_geometric 1.0.3 semantics from oracle/pyg_semantics.py.
Module/parameter names equal the reference's so its checkpoints strict-load.
GAT / GraphSAGE / attention / set2set are outside the hot path and absent.
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import pyg_semantics as pyg

# vocabulary sizes, chem/model.py:9-13
NUM_ATOM_TYPE = 120
NUM_CHIRALITY_TAG = 3
NUM_BOND_TYPE = 6
NUM_BOND_DIRECTION = 3
SELF_LOOP_BOND_TYPE = 4  # chem/model.py:43


def _with_self_loops(edge_index, edge_attr, num_nodes):
    """chem/model.py:39-45: N self loops appended; their attribute row is [4, 0]."""
    ei = pyg.add_self_loops(edge_index, num_nodes)
    loop_attr = torch.zeros(num_nodes, 2, dtype=edge_attr.dtype, device=edge_attr.device)
    loop_attr[:, 0] = SELF_LOOP_BOND_TYPE
    return ei, torch.cat([edge_attr, loop_attr], dim=0)


class _BondEmbedding:
    """mixin: edge_embedding1(attr[:,0]) + edge_embedding2(attr[:,1]) (chem/model.py:47)."""

    def bond_embedding(self, edge_attr):
        return self.edge_embedding1(edge_attr[:, 0]) + self.edge_embedding2(edge_attr[:, 1])


class GINConv(_BondEmbedding, nn.Module):
    """chem/model.py:15-55."""

    def __init__(self, emb_dim, aggr="add"):
        # construction order of the reference (mlp first, then the two embeddings) is kept so
        # that torch.manual_seed(k) gives identical initial weights.
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(emb_dim, 2 * emb_dim), nn.ReLU(), nn.Linear(2 * emb_dim, emb_dim))
        self.edge_embedding1 = nn.Embedding(NUM_BOND_TYPE, emb_dim)
        self.edge_embedding2 = nn.Embedding(NUM_BOND_DIRECTION, emb_dim)
        nn.init.xavier_uniform_(self.edge_embedding1.weight.data)
        nn.init.xavier_uniform_(self.edge_embedding2.weight.data)
        self.aggr = aggr

    def aggregate(self, x, edge_index, edge_attr):
        ei, ea = _with_self_loops(edge_index, edge_attr, x.size(0))
        ee = self.bond_embedding(ea)
        return pyg.propagate_add(ei, x, ee, lambda x_j, e: x_j + e, x.size(0))  # :49-52

    def forward(self, x, edge_index, edge_attr):
        return self.mlp(self.aggregate(x, edge_index, edge_attr))  # update, :54-55


class GCNConv(_BondEmbedding, nn.Module):
    """chem/model.py:58-104."""

    def __init__(self, emb_dim, aggr="add"):
        super().__init__()
        self.emb_dim = emb_dim
        self.linear = nn.Linear(emb_dim, emb_dim)
        self.edge_embedding1 = nn.Embedding(NUM_BOND_TYPE, emb_dim)
        self.edge_embedding2 = nn.Embedding(NUM_BOND_DIRECTION, emb_dim)
        nn.init.xavier_uniform_(self.edge_embedding1.weight.data)
        nn.init.xavier_uniform_(self.edge_embedding2.weight.data)
        self.aggr = aggr

    @staticmethod
    def norm(edge_index, num_nodes, dtype):
        """:73-82 (edge_index already holds the self loops)."""
        w = torch.ones(edge_index.size(1), dtype=dtype, device=edge_index.device)
        row, col = edge_index
        deg = pyg.scatter_add(w, row, num_nodes)
        dis = deg.pow(-0.5)
        dis[dis == float("inf")] = 0
        return dis[row] * w * dis[col]

    def forward(self, x, edge_index, edge_attr):
        ei, ea = _with_self_loops(edge_index, edge_attr, x.size(0))
        ee = self.bond_embedding(ea)
        nrm = self.norm(ei, x.size(0), x.dtype)
        x = self.linear(x)  # :99
        return pyg.propagate_add(ei, x, ee, lambda x_j, e: nrm.view(-1, 1) * (x_j + e), x.size(0))  # :103-104


class GraphSAGEConv(_BondEmbedding, nn.Module):
    """chem/model.py:165-202: mean over (x_j W^T + b + e_ij) incl. the self loop, then L2-normalise."""

    def __init__(self, emb_dim, aggr="mean"):
        super().__init__()
        self.emb_dim = emb_dim
        self.linear = nn.Linear(emb_dim, emb_dim)
        self.edge_embedding1 = nn.Embedding(NUM_BOND_TYPE, emb_dim)
        self.edge_embedding2 = nn.Embedding(NUM_BOND_DIRECTION, emb_dim)
        nn.init.xavier_uniform_(self.edge_embedding1.weight.data)
        nn.init.xavier_uniform_(self.edge_embedding2.weight.data)
        self.aggr = aggr

    def forward(self, x, edge_index, edge_attr):
        ei, ea = _with_self_loops(edge_index, edge_attr, x.size(0))
        ee = self.bond_embedding(ea)
        x = self.linear(x)  # :194
        out = pyg.propagate_mean(ei, x, ee, lambda x_j, e: x_j + e, x.size(0))  # :196-199
        return F.normalize(out, p=2, dim=-1)  # update, :201-202


class GATConv(nn.Module):
    """chem/model.py:107-162: 2-head attention over (W x_j + b + e_ij) incl. the self loop, heads averaged."""

    def __init__(self, emb_dim, heads=2, negative_slope=0.2, aggr="add"):
        super().__init__()
        self.aggr, self.emb_dim, self.heads, self.negative_slope = aggr, emb_dim, heads, negative_slope
        self.weight_linear = nn.Linear(emb_dim, heads * emb_dim)
        self.att = nn.Parameter(torch.Tensor(1, heads, 2 * emb_dim))
        self.bias = nn.Parameter(torch.Tensor(emb_dim))
        self.edge_embedding1 = nn.Embedding(NUM_BOND_TYPE, heads * emb_dim)
        self.edge_embedding2 = nn.Embedding(NUM_BOND_DIRECTION, heads * emb_dim)
        nn.init.xavier_uniform_(self.edge_embedding1.weight.data)
        nn.init.xavier_uniform_(self.edge_embedding2.weight.data)
        pyg.glorot_(self.att)
        self.bias.data.zero_()

    def forward(self, x, edge_index, edge_attr):
        n = x.size(0)
        ei, ea = _with_self_loops(edge_index, edge_attr, n)
        ee = (self.edge_embedding1(ea[:, 0]) + self.edge_embedding2(ea[:, 1])).view(-1, self.heads, self.emb_dim)
        xh = self.weight_linear(x).view(-1, self.heads, self.emb_dim)
        x_i, x_j = xh[ei[0]], xh[ei[1]] + ee  # :147-150
        alpha = (torch.cat([x_i, x_j], dim=-1) * self.att).sum(dim=-1)
        alpha = pyg.softmax(F.leaky_relu(alpha, self.negative_slope), ei[0], n)
        out = pyg.scatter_add(x_j * alpha.view(-1, self.heads, 1), ei[0], n)
        return out.mean(dim=1) + self.bias  # update, :158-162


class GNN(nn.Module):
    """chem/model.py:206-290."""

    def __init__(self, num_layer, emb_dim, JK="last", drop_ratio=0, gnn_type="gin"):
        super().__init__()
        self.num_layer, self.drop_ratio, self.JK = num_layer, drop_ratio, JK
        if num_layer < 2:
            raise ValueError("Number of GNN layers must be greater than 1.")
        self.x_embedding1 = nn.Embedding(NUM_ATOM_TYPE, emb_dim)
        self.x_embedding2 = nn.Embedding(NUM_CHIRALITY_TAG, emb_dim)
        nn.init.xavier_uniform_(self.x_embedding1.weight.data)
        nn.init.xavier_uniform_(self.x_embedding2.weight.data)
        conv = {"gin": GINConv, "gcn": GCNConv, "graphsage": GraphSAGEConv, "gat": GATConv}[gnn_type]
        self.gnns = nn.ModuleList([conv(emb_dim) for _ in range(num_layer)])
        self.batch_norms = nn.ModuleList([nn.BatchNorm1d(emb_dim) for _ in range(num_layer)])

    def forward(self, *argv):
        if len(argv) == 3:
            x, edge_index, edge_attr = argv
        elif len(argv) == 1:
            x, edge_index, edge_attr = argv[0].x, argv[0].edge_index, argv[0].edge_attr
        else:
            raise ValueError("unmatched number of arguments.")
        h = self.x_embedding1(x[:, 0]) + self.x_embedding2(x[:, 1])  # :264
        h_list = [h]
        for layer in range(self.num_layer):
            h = self.gnns[layer](h_list[layer], edge_index, edge_attr)
            h = self.batch_norms[layer](h)
            if layer != self.num_layer - 1:
                h = F.relu(h)
            h = F.dropout(h, self.drop_ratio, training=self.training)
            h_list.append(h)
        if self.JK == "concat":
            return torch.cat(h_list, dim=1)
        if self.JK == "last":
            return h_list[-1]
        stacked = torch.stack(h_list, dim=0)
        if self.JK == "max":
            return stacked.max(dim=0)[0]
        if self.JK == "sum":  # reference quirk (:286-288): returns row 0 of the layer sum
            return stacked.sum(dim=0)[0]
        raise ValueError(self.JK)


class GNN_graphpred(nn.Module):
    """chem/model.py:293-369."""

    def __init__(self, num_layer, emb_dim, num_tasks, JK="last", drop_ratio=0, graph_pooling="mean", gnn_type="gin"):
        super().__init__()
        self.num_layer, self.drop_ratio, self.JK = num_layer, drop_ratio, JK
        self.emb_dim, self.num_tasks = emb_dim, num_tasks
        if num_layer < 2:
            raise ValueError("Number of GNN layers must be greater than 1.")
        self.gnn = GNN(num_layer, emb_dim, JK, drop_ratio, gnn_type=gnn_type)
        pools = {"sum": pyg.global_add_pool, "mean": pyg.global_mean_pool, "max": pyg.global_max_pool}
        width = (num_layer + 1) * emb_dim if JK == "concat" else emb_dim
        self.mult = 1
        if graph_pooling in pools:
            self.pool = pools[graph_pooling]
        elif graph_pooling == "attention":  # :329-333
            self.pool = pyg.GlobalAttention(gate_nn=nn.Linear(width, 1))
        elif graph_pooling[:-1] == "set2set":  # :334-339, :344-345
            self.pool = pyg.Set2Set(width, int(graph_pooling[-1]))
            self.mult = 2
        else:
            raise ValueError("Invalid graph pooling type.")
        self.graph_pred_linear = nn.Linear(self.mult * width, num_tasks)

    def from_pretrained(self, model_file):
        self.gnn.load_state_dict(torch.load(model_file, map_location="cpu"))

    def forward(self, *argv):
        if len(argv) == 4:
            x, edge_index, edge_attr, batch = argv
        elif len(argv) == 1:
            d = argv[0]
            x, edge_index, edge_attr, batch = d.x, d.edge_index, d.edge_attr, d.batch
        else:
            raise ValueError("unmatched number of arguments.")
        return self.graph_pred_linear(self.pool(self.gnn(x, edge_index, edge_attr), batch))

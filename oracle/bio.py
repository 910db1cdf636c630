"""CPU oracle of the bio (PPI ego-network) GNN stack.  Test infrastructure only.

Restates /root/reference/bio/model.py: GINConv :11-58, GCNConv :61-114, GNN :227-290,
GNN_graphpred :293-347.  Differences from chem: dense 9->D edge encoder on float attrs (:27,47),
self-loop attr = one-hot index 7 (:42-43), layer-0 nodes re-embedded from a 2-row table (:30-33,
49-50), GIN message = concat[x_j, e] (:54-55), BatchNorm *inside* the GIN mlp (:24), no outer
BatchNorm in GNN.forward (:273-290), graph head on concat[mean-pool, centre-node] (:338-347).
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import pyg_semantics as pyg

NUM_EDGE_FEATURES = 9
SELF_LOOP_FEATURE = 7  # bio/model.py:43


def _with_self_loops(edge_index, edge_attr, num_nodes):
    ei = pyg.add_self_loops(edge_index, num_nodes)
    loop_attr = torch.zeros(num_nodes, NUM_EDGE_FEATURES, dtype=edge_attr.dtype, device=edge_attr.device)
    loop_attr[:, SELF_LOOP_FEATURE] = 1
    return ei, torch.cat([edge_attr, loop_attr], dim=0)


class _InputLayerMixin:
    def _make_edge_and_input(self, emb_dim, input_layer):
        self.edge_encoder = nn.Linear(NUM_EDGE_FEATURES, emb_dim)
        self.input_layer = input_layer
        if input_layer:
            self.input_node_embeddings = nn.Embedding(2, emb_dim)
            nn.init.xavier_uniform_(self.input_node_embeddings.weight.data)

    def _maybe_embed_input(self, x):
        if self.input_layer:  # bio/model.py:49-50
            return self.input_node_embeddings(x.to(torch.int64).view(-1))
        return x


class GINConv(_InputLayerMixin, nn.Module):
    """bio/model.py:11-58."""

    def __init__(self, emb_dim, aggr="add", input_layer=False):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(2 * emb_dim, 2 * emb_dim), nn.BatchNorm1d(2 * emb_dim), nn.ReLU(),
                                 nn.Linear(2 * emb_dim, emb_dim))
        self._make_edge_and_input(emb_dim, input_layer)
        self.aggr = aggr

    def aggregate(self, x, edge_index, edge_attr):
        ei, ea = _with_self_loops(edge_index, edge_attr, x.size(0))
        ee = self.edge_encoder(ea)
        x = self._maybe_embed_input(x)
        return pyg.propagate_add(ei, x, ee, lambda x_j, e: torch.cat([x_j, e], dim=1), x.size(0))

    def forward(self, x, edge_index, edge_attr):
        return self.mlp(self.aggregate(x, edge_index, edge_attr))


class GCNConv(_InputLayerMixin, nn.Module):
    """bio/model.py:61-114."""

    def __init__(self, emb_dim, aggr="add", input_layer=False):
        super().__init__()
        self.emb_dim = emb_dim
        self.linear = nn.Linear(emb_dim, emb_dim)
        self._make_edge_and_input(emb_dim, input_layer)
        self.aggr = aggr

    @staticmethod
    def norm(edge_index, num_nodes, dtype):
        w = torch.ones(edge_index.size(1), dtype=dtype, device=edge_index.device)
        row, col = edge_index
        deg = pyg.scatter_add(w, row, num_nodes)
        dis = deg.pow(-0.5)
        dis[dis == float("inf")] = 0
        return dis[row] * w * dis[col]

    def forward(self, x, edge_index, edge_attr):
        ei, ea = _with_self_loops(edge_index, edge_attr, x.size(0))
        ee = self.edge_encoder(ea)
        x = self._maybe_embed_input(x)
        nrm = self.norm(ei, x.size(0), x.dtype)
        x = self.linear(x)
        return pyg.propagate_add(ei, x, ee, lambda x_j, e: nrm.view(-1, 1) * (x_j + e), x.size(0))


class GraphSAGEConv(_InputLayerMixin, nn.Module):
    """bio/model.py:183-224."""

    def __init__(self, emb_dim, aggr="mean", input_layer=False):
        super().__init__()
        self.emb_dim = emb_dim
        self.linear = nn.Linear(emb_dim, emb_dim)
        self._make_edge_and_input(emb_dim, input_layer)
        self.aggr = aggr

    def forward(self, x, edge_index, edge_attr):
        ei, ea = _with_self_loops(edge_index, edge_attr, x.size(0))
        ee = self.edge_encoder(ea)
        x = self._maybe_embed_input(x)
        x = self.linear(x)
        out = pyg.propagate_mean(ei, x, ee, lambda x_j, e: x_j + e, x.size(0))
        return F.normalize(out, p=2, dim=-1)


class GATConv(_InputLayerMixin, nn.Module):
    """bio/model.py:117-181."""

    def __init__(self, emb_dim, heads=2, negative_slope=0.2, aggr="add", input_layer=False):
        super().__init__()
        self.aggr, self.emb_dim, self.heads, self.negative_slope = aggr, emb_dim, heads, negative_slope
        self.weight_linear = nn.Linear(emb_dim, heads * emb_dim)
        self.att = nn.Parameter(torch.Tensor(1, heads, 2 * emb_dim))
        self.bias = nn.Parameter(torch.Tensor(emb_dim))
        self.edge_encoder = nn.Linear(NUM_EDGE_FEATURES, heads * emb_dim)
        self.input_layer = input_layer
        if input_layer:
            self.input_node_embeddings = nn.Embedding(2, emb_dim)
            nn.init.xavier_uniform_(self.input_node_embeddings.weight.data)
        pyg.glorot_(self.att)
        self.bias.data.zero_()

    def forward(self, x, edge_index, edge_attr):
        n = x.size(0)
        ei, ea = _with_self_loops(edge_index, edge_attr, n)
        ee = self.edge_encoder(ea).view(-1, self.heads, self.emb_dim)
        x = self._maybe_embed_input(x)
        xh = self.weight_linear(x).view(-1, self.heads, self.emb_dim)
        x_i, x_j = xh[ei[0]], xh[ei[1]] + ee
        alpha = (torch.cat([x_i, x_j], dim=-1) * self.att).sum(dim=-1)
        alpha = pyg.softmax(F.leaky_relu(alpha, self.negative_slope), ei[0], n)
        out = pyg.scatter_add(x_j * alpha.view(-1, self.heads, 1), ei[0], n)
        return out.mean(dim=1) + self.bias


class GNN(nn.Module):
    """bio/model.py:227-290 (JK last / sum)."""

    def __init__(self, num_layer, emb_dim, JK="last", drop_ratio=0, gnn_type="gin"):
        super().__init__()
        self.num_layer, self.drop_ratio, self.JK = num_layer, drop_ratio, JK
        if num_layer < 2:
            raise ValueError("Number of GNN layers must be greater than 1.")
        conv = {"gin": GINConv, "gcn": GCNConv, "graphsage": GraphSAGEConv, "gat": GATConv}[gnn_type]
        self.gnns = nn.ModuleList([conv(emb_dim, input_layer=(layer == 0)) for layer in range(num_layer)])

    def forward(self, x, edge_index, edge_attr):
        h_list = [x]
        for layer in range(self.num_layer):
            h = self.gnns[layer](h_list[layer], edge_index, edge_attr)
            if layer != self.num_layer - 1:
                h = F.relu(h)
            h = F.dropout(h, self.drop_ratio, training=self.training)
            h_list.append(h)
        if self.JK == "last":
            return h_list[-1]
        if self.JK == "sum":  # reference quirk (:286-288): row 0 of the sum over layers 1..L
            return torch.stack(h_list[1:], dim=0).sum(dim=0)[0]
        raise ValueError(self.JK)


class GNN_graphpred(nn.Module):
    """bio/model.py:293-347."""

    def __init__(self, num_layer, emb_dim, num_tasks, JK="last", drop_ratio=0, graph_pooling="mean", gnn_type="gin"):
        super().__init__()
        self.num_layer, self.drop_ratio, self.JK = num_layer, drop_ratio, JK
        self.emb_dim, self.num_tasks = emb_dim, num_tasks
        if num_layer < 2:
            raise ValueError("Number of GNN layers must be greater than 1.")
        self.gnn = GNN(num_layer, emb_dim, JK, drop_ratio, gnn_type=gnn_type)
        pools = {"sum": pyg.global_add_pool, "mean": pyg.global_mean_pool, "max": pyg.global_max_pool}
        if graph_pooling in pools:
            self.pool = pools[graph_pooling]
        elif graph_pooling == "attention":  # :331-332
            self.pool = pyg.GlobalAttention(gate_nn=nn.Linear(emb_dim, 1))
        else:
            raise ValueError("Invalid graph pooling type.")
        self.graph_pred_linear = nn.Linear(2 * emb_dim, num_tasks)

    def from_pretrained(self, model_file):
        self.gnn.load_state_dict(torch.load(model_file, map_location="cpu"))

    def forward(self, data):
        h = self.gnn(data.x, data.edge_index, data.edge_attr)
        graph_rep = torch.cat([self.pool(h, data.batch), h[data.center_node_idx]], dim=1)
        return self.graph_pred_linear(graph_rep)

"""Drop-in for the reference's ``bio/model.py`` class surface, backed by the gfx950 HIP kernels.

Same class names, constructor signatures, forward signatures and state-dict keys as
/root/reference/bio/model.py (GINConv :11-58, GCNConv :61-114, GNN :227-290, GNN_graphpred
:293-347).  Differences from the chem stack that matter for the kernels: edge attributes are nine
0/1 floats pushed through a dense ``edge_encoder`` (folded here into a per-node 10-vector computed
once per batch, so no [E,300] edge embedding is ever materialised); the GIN message is the concat
[x_j, e_ij] (aggregation width 2D); BatchNorm sits inside the mlp; layer 0 re-embeds the constant
node feature from a 2-row table.  GPU tensors only; the CPU restatement lives in oracle/.
"""
import torch
import torch.nn.functional as F

import os

from pretrain_gnns_amd import attention, ops

_STACK_CALL = os.environ.get("PGNN_STACK_CALL", "1") != "0"  # GNN.forward of a GIN / JK="last" model as one library call


def _edge_and_input(module, emb_dim, input_layer):
    module.edge_encoder = torch.nn.Linear(9, emb_dim)
    module.input_layer = input_layer
    if input_layer:
        module.input_node_embeddings = torch.nn.Embedding(2, emb_dim)
        torch.nn.init.xavier_uniform_(module.input_node_embeddings.weight.data)


def _embed_input(module, x):
    if module.input_layer:
        return ops.Embed.apply(x.to(torch.int64).view(-1), module.input_node_embeddings.weight, None)
    return x


class GINConv(torch.nn.Module):
    """mlp(sum_j [x_j, enc(e_ij)] + [x_i, enc(e_selfloop)]), mlp = Linear-BN-ReLU-Linear."""

    def __init__(self, emb_dim, aggr="add", input_layer=False):
        super().__init__()
        if aggr != "add":
            raise NotImplementedError("only aggr='add' is on the HIP path")
        self.mlp = torch.nn.Sequential(torch.nn.Linear(2 * emb_dim, 2 * emb_dim), torch.nn.BatchNorm1d(2 * emb_dim),
                                       torch.nn.ReLU(), torch.nn.Linear(2 * emb_dim, emb_dim))
        _edge_and_input(self, emb_dim, input_layer)
        self.aggr = aggr

    def forward(self, x, edge_index, edge_attr, graph=None):
        x = _embed_input(self, x)
        if graph is None:
            graph = ops.build_bio_graph(edge_index, edge_attr, x.size(0), gcn=False)
        agg = ops.BioAggregate.apply(x, self.edge_encoder.weight, self.edge_encoder.bias, graph)
        h = ops.linear(agg, self.mlp[0])
        h = ops.batch_norm(h, self.mlp[1], relu=True)
        return ops.linear(h, self.mlp[3])


class GCNConv(torch.nn.Module):
    def __init__(self, emb_dim, aggr="add", input_layer=False):
        super().__init__()
        if aggr != "add":
            raise NotImplementedError("only aggr='add' is on the HIP path")
        self.emb_dim = emb_dim
        self.linear = torch.nn.Linear(emb_dim, emb_dim)
        _edge_and_input(self, emb_dim, input_layer)
        self.aggr = aggr

    def forward(self, x, edge_index, edge_attr, graph=None):
        x = _embed_input(self, x)
        if graph is None:
            graph = ops.build_bio_graph(edge_index, edge_attr, x.size(0), gcn=True)
        h = ops.linear(x, self.linear)
        return ops.BioAggregate.apply(h, self.edge_encoder.weight, self.edge_encoder.bias, graph)


class GraphSAGEConv(torch.nn.Module):
    """bio/model.py:183-224: L2-normalised mean over (W x_j + b + enc(e_ij)), self loop included."""

    def __init__(self, emb_dim, aggr="mean", input_layer=False):
        super().__init__()
        if aggr != "mean":
            raise NotImplementedError("only aggr='mean' is on the HIP path")
        self.emb_dim = emb_dim
        self.linear = torch.nn.Linear(emb_dim, emb_dim)
        _edge_and_input(self, emb_dim, input_layer)
        self.aggr = aggr

    def forward(self, x, edge_index, edge_attr, graph=None):
        x = _embed_input(self, x)
        if graph is None:
            graph = ops.build_bio_graph(edge_index, edge_attr, x.size(0), gcn=False)
        h = ops.linear(x, self.linear)
        total = ops.BioSumAggregate.apply(h, self.edge_encoder.weight, self.edge_encoder.bias, graph)
        return ops.MeanL2Normalize.apply(total, graph)


class GATConv(torch.nn.Module):
    """bio/model.py:117-181: weight_linear on the MFMA GEMM, then message / edge soft-max / aggregate / update as the HIP
    kernels of csrc/attention.hip (ops.BioGATAggregate; the edge_encoder output [E, 2D] is never formed)."""

    def __init__(self, emb_dim, heads=2, negative_slope=0.2, aggr="add", input_layer=False):
        super().__init__()
        if aggr != "add":
            raise NotImplementedError("only aggr='add' is implemented")
        self.aggr, self.emb_dim, self.heads, self.negative_slope = aggr, emb_dim, heads, negative_slope
        self.weight_linear = torch.nn.Linear(emb_dim, heads * emb_dim)
        self.att = torch.nn.Parameter(torch.Tensor(1, heads, 2 * emb_dim))
        self.bias = torch.nn.Parameter(torch.Tensor(emb_dim))
        self.edge_encoder = torch.nn.Linear(9, heads * emb_dim)
        self.input_layer = input_layer
        if input_layer:
            self.input_node_embeddings = torch.nn.Embedding(2, emb_dim)
            torch.nn.init.xavier_uniform_(self.input_node_embeddings.weight.data)
        self.reset_parameters()

    def reset_parameters(self):
        attention.glorot_(self.att)
        self.bias.data.zero_()

    def forward(self, x, edge_index, edge_attr, graph=None):
        x = _embed_input(self, x)
        xh = ops.linear(x, self.weight_linear)
        if self.heads == 2:
            if graph is None:
                graph = ops.build_bio_graph(edge_index, edge_attr, x.size(0))
            feat = ops.bio_slot_features(graph, edge_index, edge_attr)
            return ops.BioGATAggregate.apply(xh, self.att, self.bias, self.edge_encoder.weight, self.edge_encoder.bias, graph, feat,
                                             self.negative_slope)
        ee = self.edge_encoder(edge_attr.to(torch.float32))
        self_emb = self.edge_encoder.weight[:, 7] + self.edge_encoder.bias  # self-loop attr = one-hot index 7
        return attention.gat_propagate(xh, edge_index, ee, self_emb, self.att, self.bias, self.heads, self.negative_slope)


class GNN(torch.nn.Module):
    """bio/model.py:227-290: ``num_layer`` convs with ReLU between them (no outer BatchNorm);
    JK in last|sum; gnn_type in gin|gcn|graphsage|gat (HIP kernels)."""

    def __init__(self, num_layer, emb_dim, JK="last", drop_ratio=0, gnn_type="gin"):
        super().__init__()
        self.num_layer = num_layer
        self.drop_ratio = drop_ratio
        self.JK = JK
        self.gnn_type = gnn_type
        if self.num_layer < 2:
            raise ValueError("Number of GNN layers must be greater than 1.")

        self.gnns = torch.nn.ModuleList()
        for layer in range(num_layer):
            input_layer = layer == 0
            if gnn_type == "gin":
                self.gnns.append(GINConv(emb_dim, aggr="add", input_layer=input_layer))
            elif gnn_type == "gcn":
                self.gnns.append(GCNConv(emb_dim, input_layer=input_layer))
            elif gnn_type == "graphsage":
                self.gnns.append(GraphSAGEConv(emb_dim, input_layer=input_layer))
            elif gnn_type == "gat":
                self.gnns.append(GATConv(emb_dim, input_layer=input_layer))
            else:
                raise ValueError("unknown gnn_type %r" % (gnn_type,))

    def forward(self, x, edge_index, edge_attr):
        graph = ops.build_bio_graph(edge_index, edge_attr, x.size(0), gcn=(self.gnn_type == "gcn"))
        if (_STACK_CALL and self.gnn_type == "gin" and self.JK == "last" and (self.drop_ratio == 0 or not self.training)
                and not any(getattr(c.mlp[1], "pgnn_exact", False) for c in self.gnns)):
            # the whole network as one library call per direction (ops.BioGINStack)
            return ops.bio_gin_stack(_embed_input(self.gnns[0], x), graph, list(self.gnns))
        h_list = [x]
        for layer in range(self.num_layer):
            h = self.gnns[layer](h_list[layer], edge_index, edge_attr, graph)
            if layer != self.num_layer - 1:
                h = F.relu(h)
            if self.drop_ratio > 0:
                h = F.dropout(h, self.drop_ratio, training=self.training)
            h_list.append(h)

        if self.JK == "last":
            node_representation = h_list[-1]
        elif self.JK == "sum":
            # reference quirk (bio/model.py:286-288): row 0 of the sum over layers 1..L
            node_representation = torch.sum(torch.stack(h_list[1:], dim=0), dim=0)[0]
        else:
            raise ValueError("unknown JK mode %r" % (self.JK,))
        return node_representation


def global_add_pool(x, batch, size=None):
    return ops.global_add_pool(x, batch, size)


def global_mean_pool(x, batch, size=None):
    return ops.global_mean_pool(x, batch, size)


def global_max_pool(x, batch, size=None):
    return ops.global_max_pool(x, batch, size)


class GNN_graphpred(torch.nn.Module):
    """bio/model.py:293-347: head on concat[pool(h), h[center_node_idx]]."""

    def __init__(self, num_layer, emb_dim, num_tasks, JK="last", drop_ratio=0, graph_pooling="mean", gnn_type="gin"):
        super().__init__()
        self.num_layer = num_layer
        self.drop_ratio = drop_ratio
        self.JK = JK
        self.emb_dim = emb_dim
        self.num_tasks = num_tasks
        if self.num_layer < 2:
            raise ValueError("Number of GNN layers must be greater than 1.")

        self.gnn = GNN(num_layer, emb_dim, JK, drop_ratio, gnn_type=gnn_type)

        if graph_pooling == "sum":
            self.pool = global_add_pool
        elif graph_pooling == "mean":
            self.pool = global_mean_pool
        elif graph_pooling == "max":
            self.pool = global_max_pool
        elif graph_pooling == "attention":
            self.pool = attention.GlobalAttention(gate_nn=torch.nn.Linear(emb_dim, 1))
        else:
            raise ValueError("Invalid graph pooling type.")

        self.graph_pred_linear = torch.nn.Linear(2 * self.emb_dim, self.num_tasks)

    def from_pretrained(self, model_file):
        self.gnn.load_state_dict(torch.load(model_file, map_location=lambda storage, loc: storage))

    def forward(self, data):
        x, edge_index, edge_attr, batch = data.x, data.edge_index, data.edge_attr, data.batch
        node_representation = self.gnn(x, edge_index, edge_attr)
        pooled = self.pool(node_representation, batch)
        center_node_rep = node_representation[data.center_node_idx]
        return self.graph_pred_linear(torch.cat([pooled, center_node_rep], dim=1))


if __name__ == "__main__":
    pass

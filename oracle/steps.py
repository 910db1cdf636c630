"""Restatement of the reference's per-batch ``train()`` bodies.  Test infrastructure only.

Each function runs ONE optimisation step on ONE batch exactly as the loop body of the cited
reference function does, and returns the scalars that loop accumulates.  They are written
against duck-typed ``model`` objects, so the same function drives the CPU oracle modules and
(from tests / bench.py only) the HIP-backed modules -- that is how "the reference train()
calls the new GNN unchanged" is checked.
"""
import torch
import torch.nn.functional as F

from . import pyg_semantics as pyg


def compute_accuracy(pred, target):
    """chem/pretrain_masking.py:30-31."""
    return float(torch.sum(torch.max(pred.detach(), dim=1)[1] == target).cpu().item()) / len(pred)


def chem_masking_step(model_list, optimizer_list, batch, mask_edge=False):
    """chem/pretrain_masking.py:47-76 (loop body of train())."""
    model, linear_pred_atoms, linear_pred_bonds = model_list
    node_rep = model(batch.x, batch.edge_index, batch.edge_attr)
    pred_node = linear_pred_atoms(node_rep[batch.masked_atom_indices])
    loss = F.cross_entropy(pred_node.double(), batch.mask_node_label[:, 0])
    acc_node = compute_accuracy(pred_node, batch.mask_node_label[:, 0])
    acc_edge = 0.0
    if mask_edge:
        masked_edge_index = batch.edge_index[:, batch.connected_edge_indices]
        edge_rep = node_rep[masked_edge_index[0]] + node_rep[masked_edge_index[1]]
        pred_edge = linear_pred_bonds(edge_rep)
        loss = loss + F.cross_entropy(pred_edge.double(), batch.mask_edge_label[:, 0])
        acc_edge = compute_accuracy(pred_edge, batch.mask_edge_label[:, 0])
    for opt in optimizer_list:
        opt.zero_grad()
    loss.backward()
    for opt in optimizer_list:
        opt.step()
    return float(loss.detach().cpu().item()), acc_node, acc_edge


def bio_masking_step(model_list, optimizer_list, batch):
    """bio/pretrain_masking.py:39-64 (loop body of train()); CE in float32 (:58)."""
    model, linear_pred_edges = model_list
    node_rep = model(batch.x, batch.edge_index, batch.edge_attr)
    masked_edge_index = batch.edge_index[:, batch.masked_edge_idx]
    edge_rep = node_rep[masked_edge_index[0]] + node_rep[masked_edge_index[1]]
    pred_edge = linear_pred_edges(edge_rep)
    edge_label = torch.argmax(batch.mask_edge_label, dim=1)
    acc_edge = compute_accuracy(pred_edge, edge_label)
    for opt in optimizer_list:
        opt.zero_grad()
    loss = F.cross_entropy(pred_edge, edge_label)
    loss.backward()
    for opt in optimizer_list:
        opt.step()
    return float(loss.detach().cpu().item()), acc_edge


def cycle_index(num, shift):
    """chem/pretrain_contextpred.py:36-39: roll-left by ``shift``."""
    arr = torch.arange(num) + shift
    arr[-shift:] = torch.arange(shift)
    return arr


def contextpred_logits(model_substruct, model_context, batch, neg_samples=1, pool=pyg.global_mean_pool, mode="cbow"):
    """chem/pretrain_contextpred.py:54-81 (= bio/pretrain_contextpred.py:49-76): cbow and skipgram branches."""
    substruct_rep = model_substruct(batch.x_substruct, batch.edge_index_substruct,
                                    batch.edge_attr_substruct)[batch.center_substruct_idx]
    overlapped_node_rep = model_context(batch.x_context, batch.edge_index_context,
                                        batch.edge_attr_context)[batch.overlap_context_substruct_idx]
    if mode == "cbow":
        context_rep = pool(overlapped_node_rep, batch.batch_overlapped_context)
        neg_context_rep = torch.cat(
            [context_rep[cycle_index(len(context_rep), i + 1).to(context_rep.device)] for i in range(neg_samples)], dim=0)
        pred_pos = torch.sum(substruct_rep * context_rep, dim=1)
        pred_neg = torch.sum(substruct_rep.repeat((neg_samples, 1)) * neg_context_rep, dim=1)
    elif mode == "skipgram":  # :69-81: every overlap node against its own (pos) / the next graph's (neg) centre
        sizes = batch.overlapped_context_size
        expanded = torch.cat([substruct_rep[i].repeat((int(sizes[i]), 1)) for i in range(len(substruct_rep))], dim=0)
        pred_pos = torch.sum(expanded * overlapped_node_rep, dim=1)
        shifted_all = []
        for i in range(neg_samples):
            shifted = substruct_rep[cycle_index(len(substruct_rep), i + 1)]
            shifted_all.append(torch.cat([shifted[j].repeat((int(sizes[j]), 1)) for j in range(len(shifted))], dim=0))
        pred_neg = torch.sum(torch.cat(shifted_all, dim=0) * overlapped_node_rep.repeat((neg_samples, 1)), dim=1)
    else:
        raise ValueError("Invalid mode!")
    return pred_pos, pred_neg


def chem_contextpred_step(model_substruct, model_context, optimizer_substruct, optimizer_context, batch,
                          neg_samples=1, pool=pyg.global_mean_pool, mode="cbow"):
    """chem/pretrain_contextpred.py:51-100 and bio/pretrain_contextpred.py:46-95 (loop body of train())."""
    pred_pos, pred_neg = contextpred_logits(model_substruct, model_context, batch, neg_samples, pool, mode)
    loss_pos = F.binary_cross_entropy_with_logits(pred_pos.double(), torch.ones_like(pred_pos).double())
    loss_neg = F.binary_cross_entropy_with_logits(pred_neg.double(), torch.zeros_like(pred_neg).double())
    optimizer_substruct.zero_grad()
    optimizer_context.zero_grad()
    loss = loss_pos + neg_samples * loss_neg
    loss.backward()
    optimizer_substruct.step()
    optimizer_context.step()
    balanced = float(loss_pos.detach().cpu().item() + loss_neg.detach().cpu().item())
    acc = 0.5 * (float(torch.sum(pred_pos > 0).detach().cpu().item()) / len(pred_pos)
                 + float(torch.sum(pred_neg < 0).detach().cpu().item()) / len(pred_neg))
    return balanced, acc


def chem_finetune_step(model, optimizer, batch):
    """chem/finetune.py:27-49 (loop body of train()): multi-task BCE-with-logits in float64 over the
    non-null labels (y in {-1, 0 = missing, +1})."""
    pred = model(batch.x, batch.edge_index, batch.edge_attr, batch.batch)
    y = batch.y.view(pred.shape).to(torch.float64)
    is_valid = y ** 2 > 0
    loss_mat = F.binary_cross_entropy_with_logits(pred.double(), (y + 1) / 2, reduction="none")
    loss_mat = torch.where(is_valid, loss_mat, torch.zeros(loss_mat.shape).to(loss_mat.device).to(loss_mat.dtype))
    optimizer.zero_grad()
    loss = torch.sum(loss_mat) / torch.sum(is_valid)
    loss.backward()
    optimizer.step()
    return float(loss.detach().cpu().item())


def roc_auc(labels01, scores):
    """rank-based ROC-AUC with midranks for ties (= sklearn.metrics.roc_auc_score, which
    chem/finetune.py:73 calls)."""
    import numpy as np
    labels01, scores = np.asarray(labels01, dtype=np.float64), np.asarray(scores, dtype=np.float64)
    order = np.argsort(scores, kind="mergesort")
    ranks = np.empty(len(scores), dtype=np.float64)
    s = scores[order]
    i = 0
    while i < len(s):
        j = i
        while j + 1 < len(s) and s[j + 1] == s[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    pos = labels01 == 1
    n_pos, n_neg = pos.sum(), (~pos).sum()
    return float((ranks[pos].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def chem_eval(model, batches):
    """chem/finetune.py:52-77: eval-mode forward over the loader, mean ROC-AUC over the tasks that have
    both classes, computed on the valid (non-zero) labels."""
    import numpy as np
    model.eval()
    y_true, y_scores = [], []
    for batch in batches:
        with torch.no_grad():
            pred = model(batch.x, batch.edge_index, batch.edge_attr, batch.batch)
        y_true.append(batch.y.view(pred.shape))
        y_scores.append(pred)
    y_true = torch.cat(y_true, dim=0).cpu().numpy()
    y_scores = torch.cat(y_scores, dim=0).cpu().numpy()
    roc_list = []
    for i in range(y_true.shape[1]):
        if np.sum(y_true[:, i] == 1) > 0 and np.sum(y_true[:, i] == -1) > 0:
            is_valid = y_true[:, i] ** 2 > 0
            roc_list.append(roc_auc((y_true[is_valid, i] + 1) / 2, y_scores[is_valid, i]))
    return sum(roc_list) / len(roc_list)


def chem_masking_epoch(model_list, optimizer_list, loader, mask_edge=False, device=None):
    """chem/pretrain_masking.py:37-78 (train()): one pass over ``loader``; the three averages divide by
    the LAST step index, not the number of steps (:78) -- preserved."""
    for m in model_list:
        m.train()
    loss_accum = acc_node_accum = acc_edge_accum = 0.0
    step = 0
    for step, batch in enumerate(loader):
        if device is not None:
            batch = batch.to(device)
        loss, acc_node, acc_edge = chem_masking_step(model_list, optimizer_list, batch, mask_edge)
        loss_accum += loss
        acc_node_accum += acc_node
        acc_edge_accum += acc_edge
    return loss_accum / step, acc_node_accum / step, acc_edge_accum / step


def bio_finetune_step(model, optimizer, batch):
    """bio/finetune.py:25-37 (loop body of train()): bio GNN_graphpred on the whole batch object (it reads
    center_node_idx), BCE-with-logits in float64 against go_target_downstream viewed as [graphs, tasks]."""
    pred = model(batch)
    y = batch.go_target_downstream.view(pred.shape).to(torch.float64)
    optimizer.zero_grad()
    loss = F.binary_cross_entropy_with_logits(pred.double(), y)
    loss.backward()
    optimizer.step()
    return float(loss.detach().cpu().item())


def bio_eval(model, batches):
    """bio/finetune.py:40-65: eval-mode scores of every batch, then one ROC-AUC per task (nan where a task has a single class)."""
    import numpy as np
    model.eval()
    y_true, y_scores = [], []
    for batch in batches:
        with torch.no_grad():
            pred = model(batch)
        y_true.append(batch.go_target_downstream.view(pred.shape).detach().cpu())
        y_scores.append(pred.detach().cpu())
    y_true, y_scores = torch.cat(y_true, dim=0).numpy(), torch.cat(y_scores, dim=0).numpy()
    roc_list = []
    for i in range(y_true.shape[1]):
        if np.sum(y_true[:, i] == 1) > 0 and np.sum(y_true[:, i] == 0) > 0:
            roc_list.append(roc_auc(y_true[:, i], y_scores[:, i]))
        else:
            roc_list.append(np.nan)
    return np.array(roc_list)


def chem_edgepred_step(model, optimizer, batch):
    """chem/pretrain_edgepred.py:32-46 (loop body of train()): dot-product scores of the bonded pairs (one
    direction of every bond) against sampled non-bonded pairs, BCE-with-logits in float32."""
    node_emb = model(batch.x, batch.edge_index, batch.edge_attr)
    positive_score = torch.sum(node_emb[batch.edge_index[0, ::2]] * node_emb[batch.edge_index[1, ::2]], dim=1)
    negative_score = torch.sum(node_emb[batch.negative_edge_index[0]] * node_emb[batch.negative_edge_index[1]], dim=1)
    optimizer.zero_grad()
    loss = (F.binary_cross_entropy_with_logits(positive_score, torch.ones_like(positive_score))
            + F.binary_cross_entropy_with_logits(negative_score, torch.zeros_like(negative_score)))
    loss.backward()
    optimizer.step()
    acc = (torch.sum(positive_score > 0) + torch.sum(negative_score < 0)).to(torch.float32) / float(2 * len(positive_score))
    return float(loss.detach().cpu().item()), float(acc.detach().cpu().item())


class Discriminator(torch.nn.Module):
    """chem/pretrain_deepgraphinfomax.py:30-42: bilinear score x^T W s; W ~ U(-1/sqrt(D), 1/sqrt(D))."""

    def __init__(self, hidden_dim):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.Tensor(hidden_dim, hidden_dim))
        bound = 1.0 / (hidden_dim ** 0.5)
        self.weight.data.uniform_(-bound, bound)

    def forward(self, x, summary):
        return torch.sum(x * torch.matmul(summary, self.weight), dim=1)


def chem_infomax_step(gnn, discriminator, optimizer, batch, pool=pyg.global_mean_pool):
    """chem/pretrain_deepgraphinfomax.py:61-84 (loop body of train()): node embeddings scored against the
    sigmoid of their own graph's mean-pooled summary (positive) and of the next graph's (negative)."""
    node_emb = gnn(batch.x, batch.edge_index, batch.edge_attr)
    summary_emb = torch.sigmoid(pool(node_emb, batch.batch))
    positive_expanded = summary_emb[batch.batch]
    shifted = summary_emb[cycle_index(len(summary_emb), 1).to(summary_emb.device)]
    negative_expanded = shifted[batch.batch]
    positive_score = discriminator(node_emb, positive_expanded)
    negative_score = discriminator(node_emb, negative_expanded)
    optimizer.zero_grad()
    loss = (F.binary_cross_entropy_with_logits(positive_score, torch.ones_like(positive_score))
            + F.binary_cross_entropy_with_logits(negative_score, torch.zeros_like(negative_score)))
    loss.backward()
    optimizer.step()
    acc = (torch.sum(positive_score > 0) + torch.sum(negative_score < 0)).to(torch.float32) / float(2 * len(positive_score))
    return float(loss.detach().cpu().item()), float(acc.detach().cpu().item())


def bio_edgepred_step(model, optimizer, batch):
    """bio/pretrain_edgepred.py:26-40: the same loop body as the chem script's, on bio/model.py's GNN."""
    return chem_edgepred_step(model, optimizer, batch)


def bio_infomax_step(gnn, discriminator, optimizer, batch, pool=pyg.global_mean_pool):
    """bio/pretrain_deepgraphinfomax.py:59-81: the same loop body as the chem script's, on bio/model.py's GNN."""
    return chem_infomax_step(gnn, discriminator, optimizer, batch, pool)

"""Host-side mirror of the reference's per-batch ``train()`` loop bodies, for the HIP-backed models.

The reference scripts can call the drop-in ``model.GNN`` unchanged; these functions exist so that
``bench.py`` and users without rdkit/torch_geometric can drive the same step without the scripts'
dataset plumbing.  Same statements, same order, same dtypes (float64 losses) as
chem/pretrain_masking.py:47-76, bio/pretrain_masking.py:39-64, chem/pretrain_contextpred.py:51-100.
"""
import torch
import torch.nn.functional as F

from . import ops


def compute_accuracy(pred, target):
    return float(torch.sum(torch.max(pred.detach(), dim=1)[1] == target).cpu().item()) / len(pred)


def _fusable_head(linear, node_rep):
    """plain nn.Linear on the GPU with up to 128 classes: the fused masking head of csrc/head.hip applies"""
    return (type(linear) is torch.nn.Linear and node_rep.is_cuda and node_rep.dtype == torch.float32 and linear.out_features <= 128
            and linear.in_features % 4 == 0 and linear.in_features <= 2048)


def _fusable_edge_head(linear, node_rep, label):
    """plain nn.Linear onto 4 (bond types) or 7 (PPI edge types) classes on the GPU: the edge head of csrc/edgehead.hip applies"""
    return (type(linear) is torch.nn.Linear and node_rep.is_cuda and node_rep.dtype == torch.float32 and linear.out_features in (4, 7)
            and linear.in_features % 4 == 0 and linear.in_features <= 1024 and label.numel() > 0
            and (label.dtype == torch.int64 or (label.dtype == torch.float32 and label.dim() == 2 and label.size(1) >= linear.out_features)))


def _correct(pred, target):
    """numerator of compute_accuracy, left on the device"""
    return torch.sum(torch.max(pred.detach(), dim=1)[1] == target)


_ones = {}


def _unit_grad(loss):
    """d loss / d loss for ``loss.backward``: a cached float64 one per device (torch builds a new ones_like every call)"""
    key = (loss.device.type, loss.device.index, loss.dtype)
    t = _ones.get(key)
    if t is None:
        t = _ones[key] = torch.ones((), dtype=loss.dtype, device=loss.device)
    return t


def _backward(loss, unit=True):
    """``loss.backward()`` on the CALLING thread.  By default the autograd engine hands a CUDA graph to a per-device worker thread
    and the caller sleeps until it is done: two thread hand-overs (and a GIL hand-over per Python-implemented node) per step.  The
    steps here enqueue ~100 launches in ~1 ms of host time, and on a host-bound box that detour is 0.3 ms of it (tools/host_ab.py:
    1.03-1.16 -> 0.71-0.84 ms per step with the GPU out of the way).  Same gradients; only the thread that walks the graph changes."""
    with torch.autograd.set_multithreading_enabled(False):
        if unit:
            loss.backward(_unit_grad(loss))
        else:
            loss.backward()


def epoch_accumulator(device):
    """float64 [4] on ``device``: the running sums train() of chem/pretrain_masking.py:72-76 keeps on the host --
    [sum of loss, sum of node accuracy, sum of edge accuracy, steps] -- for ``readback="epoch"``."""
    return torch.zeros(4, dtype=torch.float64, device=device)


def chem_masking_step(model_list, optimizer_list, batch, mask_edge=False, readback="end", accum=None):
    """One iteration of chem/pretrain_masking.py:46-76.

    readback="inline" reads accuracy and loss back exactly where the reference does (a device->host
    sync between forward and backward, another after the optimizer).  readback="end" (default) computes
    the same three numbers from the same tensors but fetches them with ONE transfer after
    ``optimizer.step()``, so the GPU is not left idle mid-step waiting for Python to resume.
    readback="epoch" does not fetch at all: the step adds its three numbers to ``accum`` (an
    ``epoch_accumulator``) on the device -- the same float64 additions, in the same order, the reference's
    ``loss_accum += ...`` lines do on the host -- and returns None, so the host enqueues the next step while
    this one still runs; ``chem_masking_epoch`` reads the sums back once."""
    model, linear_pred_atoms, linear_pred_bonds = model_list
    node_rep = model(batch.x, batch.edge_index, batch.edge_attr)
    if readback not in ("inline", "end", "epoch"):
        raise ValueError("readback must be 'inline', 'end' or 'epoch'")
    inline, deferred = readback == "inline", readback == "epoch"
    if deferred and accum is None:
        raise ValueError("readback='epoch' needs accum=epoch_accumulator(device)")
    packed = None
    if not inline and _fusable_head(linear_pred_atoms, node_rep):
        # the three statements below as one launch per direction (ops.MaskedHead: same dtypes, float64 soft-max and loss)
        loss, acc_node, packed = ops.masked_head(node_rep, batch.masked_atom_indices, linear_pred_atoms,
                                                 batch.mask_node_label[:, 0], with_metrics=True,
                                                 accum=accum if deferred and not mask_edge else None)
        n_node = batch.masked_atom_indices.numel()
    else:
        pred_node = linear_pred_atoms(node_rep[batch.masked_atom_indices])
        loss = F.cross_entropy(pred_node.double(), batch.mask_node_label[:, 0])
        acc_node = compute_accuracy(pred_node, batch.mask_node_label[:, 0]) if inline else _correct(pred_node, batch.mask_node_label[:, 0])
        n_node = len(pred_node)
    n_edge = 1
    acc_edge = 0.0 if (inline or not mask_edge) else None
    if mask_edge:
        masked_edge_index = batch.edge_index[:, batch.connected_edge_indices]
        edge_label = batch.mask_edge_label[:, 0]
        if not inline and masked_edge_index.size(1) > 0 and _fusable_edge_head(linear_pred_bonds, node_rep, edge_label):
            # the bond head (:60-66) on [nodes, 4]-wide data (ops.EdgeHead: float64 soft-max and loss as `pred_edge.double()` here)
            loss_edge, acc_edge, _ = ops.edge_head(node_rep, masked_edge_index, linear_pred_bonds, edge_label.contiguous(), float64=True)
            loss = loss + loss_edge
            n_edge = masked_edge_index.size(1)
        else:
            edge_rep = node_rep[masked_edge_index[0]] + node_rep[masked_edge_index[1]]
            pred_edge = linear_pred_bonds(edge_rep)
            loss = loss + F.cross_entropy(pred_edge.double(), edge_label)
            n_edge = len(pred_edge)
            acc_edge = compute_accuracy(pred_edge, edge_label) if inline else _correct(pred_edge, edge_label)
    for opt in optimizer_list:
        opt.zero_grad()
    # (Measured and NOT kept, profiles/r04/deferred_head_grads_ab.txt: the head's own weight / bias gradients -- 26 us at the head of
    # the backward, needed by the optimizers only -- on an auxiliary stream, joined behind backward(): bit-identical, 0.993-1.000
    # against 0.976-0.989 ms; a third stream beside the network's two costs more than it hides.)
    _backward(loss)
    for opt in optimizer_list:
        opt.step()
    if inline:
        return float(loss.cpu().item()), acc_node, acc_edge
    if deferred:
        if packed is None or mask_edge:  # (the fused head added its own numbers inside its forward launch)
            if not torch.is_tensor(acc_edge):
                acc_edge = torch.zeros((), dtype=torch.long, device=loss.device)
            accum += torch.stack([loss.detach().double(), acc_node.double() / n_node, acc_edge.double() / n_edge,
                                  torch.ones((), dtype=torch.float64, device=loss.device)])
        return None
    if packed is not None and not mask_edge:  # (loss, correct) already side by side on the device
        vals = packed.cpu().tolist()
        return vals[0], vals[1] / n_node, 0.0
    if not torch.is_tensor(acc_edge):
        acc_edge = torch.zeros((), dtype=torch.long, device=loss.device)
    vals = torch.stack([loss.detach(), acc_node.double(), acc_edge.double()]).cpu().tolist()
    return vals[0], vals[1] / n_node, vals[2] / n_edge


def chem_masking_epoch(model_list, optimizer_list, loader, mask_edge=False, device=None, readback="epoch"):
    """train() of chem/pretrain_masking.py:37-78: one pass over ``loader`` (a ResidentLoader yields batches
    that already live on the GPU; host batches are moved with ``.to(device)``).  Returns the reference's
    three epoch averages, which divide by the LAST step index rather than the step count (:78).
    The function's only outputs are the epoch sums, so by default (readback="epoch") they are accumulated on
    the device and fetched once; "end" / "inline" fetch every step and add on the host."""
    for m in model_list:
        m.train()
    loss_accum = acc_node_accum = acc_edge_accum = 0.0
    step = 0
    accum = None
    for step, batch in enumerate(loader):
        if device is not None:
            batch = batch.to(device)
        if readback == "epoch":
            if accum is None:
                accum = epoch_accumulator(batch.x.device)
            chem_masking_step(model_list, optimizer_list, batch, mask_edge, readback, accum)
            continue
        loss, acc_node, acc_edge = chem_masking_step(model_list, optimizer_list, batch, mask_edge, readback)
        loss_accum += loss
        acc_node_accum += acc_node
        acc_edge_accum += acc_edge
    if accum is not None:
        loss_accum, acc_node_accum, acc_edge_accum, _ = accum.cpu().tolist()
    return loss_accum / step, acc_node_accum / step, acc_edge_accum / step


class GraphedChemMaskingStep:
    """The same masking train step captured ONCE into a HIP graph and replayed (torch.cuda.CUDAGraph;
    the library's launches, memsets and its side-stream fork/join are ordinary stream work and are
    captured like torch's own kernels).  Valid only while the batch keeps its shape -- node / edge /
    masked-atom counts are baked into the captured launches -- so it serves fixed-shape replay
    (benchmarks, bucketed/padded loaders); variable-shape training uses ``chem_masking_step``.
    Optimizers must be built with ``capturable=True``.  Host-side scalars are frozen at capture time, which
    includes the fused-dropout seeds: use it with ``drop_ratio == 0`` only."""

    def __init__(self, model_list, optimizer_list, batch, mask_edge=False, warmup=3, readback="end"):
        self.model_list, self.optimizer_list, self.batch, self.mask_edge = model_list, optimizer_list, batch, mask_edge
        if readback not in ("end", "epoch"):
            raise ValueError("readback must be 'end' or 'epoch'")
        # readback="epoch": the captured step adds its numbers to self.accum (epoch_accumulator) and a replay fetches nothing;
        # sums() reads and clears them
        self.accum = epoch_accumulator(batch.x.device) if readback == "epoch" else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._core()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._core()
        if self.accum is not None:
            torch.cuda.synchronize()
            self.accum.zero_()  # drop what the warm-up and the capture pass added

    def _core(self):
        model, linear_pred_atoms, linear_pred_bonds = self.model_list
        b = self.batch
        node_rep = model(b.x, b.edge_index, b.edge_attr)
        fused = _fusable_head(linear_pred_atoms, node_rep)
        if fused:
            loss, acc_node = ops.masked_head(node_rep, b.masked_atom_indices, linear_pred_atoms, b.mask_node_label[:, 0],
                                             accum=self.accum if not self.mask_edge else None)
        else:
            pred_node = linear_pred_atoms(node_rep[b.masked_atom_indices])
            loss = F.cross_entropy(pred_node.double(), b.mask_node_label[:, 0])
            acc_node = _correct(pred_node, b.mask_node_label[:, 0])
        acc_edge = torch.zeros((), dtype=torch.long, device=loss.device)
        self.n_node, self.n_edge = b.masked_atom_indices.numel(), 1
        if self.mask_edge:
            mei = b.edge_index[:, b.connected_edge_indices]
            pred_edge = linear_pred_bonds(node_rep[mei[0]] + node_rep[mei[1]])
            loss = loss + F.cross_entropy(pred_edge.double(), b.mask_edge_label[:, 0])
            acc_edge = _correct(pred_edge, b.mask_edge_label[:, 0])
            self.n_edge = len(pred_edge)
        for opt in self.optimizer_list:
            opt.zero_grad(set_to_none=True)
        _backward(loss)
        for opt in self.optimizer_list:
            opt.step()
        out = torch.stack([loss.detach(), acc_node.double(), acc_edge.double()])
        if self.accum is not None and (not fused or self.mask_edge):
            self.accum += torch.stack([out[0], out[1] / self.n_node, out[2] / self.n_edge, torch.ones_like(out[0])])
        return out

    def __call__(self):
        self.graph.replay()
        if self.accum is not None:
            return None
        vals = self.out.cpu().tolist()
        return vals[0], vals[1] / self.n_node, vals[2] / self.n_edge

    def sums(self):
        """readback="epoch": (sum of loss, sum of node accuracy, sum of edge accuracy, replays) since the last call"""
        vals = self.accum.cpu().tolist()
        self.accum.zero_()
        return vals


def bio_masking_step(model_list, optimizer_list, batch, readback="end", accum=None):
    """One iteration of bio/pretrain_masking.py:29-60.  readback as in ``chem_masking_step``: "inline" syncs where the
    reference does (accuracy between forward and backward, loss after the optimizer), "end" (default) fetches the same two
    numbers with one transfer after ``optimizer.step()``, "epoch" adds them to ``accum`` (``epoch_accumulator``: [loss sum,
    unused, edge-accuracy sum, steps]) on the device and returns None."""
    if readback not in ("inline", "end", "epoch"):
        raise ValueError("readback must be 'inline', 'end' or 'epoch'")
    if readback == "epoch" and accum is None:
        raise ValueError("readback='epoch' needs accum=epoch_accumulator(device)")
    model, linear_pred_edges = model_list
    node_rep = model(batch.x, batch.edge_index, batch.edge_attr)
    masked_edge_index = batch.edge_index[:, batch.masked_edge_idx]
    if readback != "inline" and _fusable_edge_head(linear_pred_edges, node_rep, batch.mask_edge_label):
        # the statements below as two launches forward / ten backward on [*, 7]-wide data (ops.EdgeHead: fp32 soft-max and loss as here)
        deferred = readback == "epoch"
        loss, correct, packed = ops.edge_head(node_rep, masked_edge_index, linear_pred_edges, batch.mask_edge_label,
                                              accum=accum if deferred else None, accum_slot=2)
        for opt in optimizer_list:
            opt.zero_grad()
        _backward(loss)
        for opt in optimizer_list:
            opt.step()
        if deferred:
            return None
        vals = packed.cpu().tolist()
        return vals[0], vals[1] / masked_edge_index.size(1)
    edge_rep = node_rep[masked_edge_index[0]] + node_rep[masked_edge_index[1]]
    pred_edge = linear_pred_edges(edge_rep)
    edge_label = torch.argmax(batch.mask_edge_label, dim=1)
    inline = readback == "inline"
    acc_edge = compute_accuracy(pred_edge, edge_label) if inline else _correct(pred_edge, edge_label)
    for opt in optimizer_list:
        opt.zero_grad()
    loss = F.cross_entropy(pred_edge, edge_label)
    _backward(loss)
    for opt in optimizer_list:
        opt.step()
    if inline:
        return float(loss.cpu().item()), acc_edge
    if readback == "epoch":
        zero = torch.zeros((), dtype=torch.float64, device=loss.device)
        accum += torch.stack([loss.detach().double(), zero, acc_edge.double() / len(pred_edge), zero + 1.0])
        return None
    vals = torch.stack([loss.detach().double(), acc_edge.double()]).cpu().tolist()
    return vals[0], vals[1] / len(pred_edge)


def bio_masking_epoch(model_list, optimizer_list, loader, device=None, readback="epoch"):
    """train() of bio/pretrain_masking.py:29-66: one pass over ``loader``; returns (loss_accum / (step + 1), acc_accum / (step + 1))
    -- the bio script divides by the step COUNT (:66), the chem one by the last step index.  Sums on the device by default, as
    in ``chem_masking_epoch``."""
    for m in model_list:
        m.train()
    loss_accum = acc_accum = 0.0
    step = 0
    accum = None
    for step, batch in enumerate(loader):
        if device is not None:
            batch = batch.to(device)
        if readback == "epoch":
            if accum is None:
                accum = epoch_accumulator(batch.x.device)
            bio_masking_step(model_list, optimizer_list, batch, readback, accum)
            continue
        loss, acc = bio_masking_step(model_list, optimizer_list, batch, readback)
        loss_accum += loss
        acc_accum += acc
    if accum is not None:
        loss_accum, _, acc_accum, _ = accum.cpu().tolist()
    return loss_accum / (step + 1), acc_accum / (step + 1)


def cycle_index(num, shift):
    arr = torch.arange(num) + shift
    arr[-shift:] = torch.arange(shift)
    return arr


def contextpred_logits(model_substruct, model_context, batch, neg_samples=1, mode="cbow", pool=None, node_reps=None):
    """chem/pretrain_contextpred.py:54-81 (identical in bio/pretrain_contextpred.py:49-76): the positive and
    negative dot-product scores.  cbow: centre embedding against the pooled overlap embeddings of its own graph /
    of the graph `shift` places later (cycle_index).  skipgram: every overlap node against its own / the shifted
    centre; the reference's per-graph ``.repeat`` loops are one ``repeat_interleave`` here (a pure gather: same bits).
    ``node_reps`` = (substructure, context) node embeddings the caller has already computed (the two networks then do NOT
    run again: a second training-mode forward would move the BatchNorm running statistics twice)."""
    pool = ops.global_mean_pool if pool is None else pool
    if node_reps is None:
        node_reps = (model_substruct(batch.x_substruct, batch.edge_index_substruct, batch.edge_attr_substruct),
                     model_context(batch.x_context, batch.edge_index_context, batch.edge_attr_context))
    substruct_rep = node_reps[0][batch.center_substruct_idx]
    overlapped_node_rep = node_reps[1][batch.overlap_context_substruct_idx]
    if mode == "cbow":
        context_rep = pool(overlapped_node_rep, batch.batch_overlapped_context)
        neg_context_rep = torch.cat([context_rep[cycle_index(len(context_rep), i + 1).to(context_rep.device)]
                                     for i in range(neg_samples)], dim=0)
        pred_pos = torch.sum(substruct_rep * context_rep, dim=1)
        pred_neg = torch.sum(substruct_rep.repeat((neg_samples, 1)) * neg_context_rep, dim=1)
    elif mode == "skipgram":
        sizes = batch.overlapped_context_size.to(substruct_rep.device)
        pred_pos = torch.sum(torch.repeat_interleave(substruct_rep, sizes, dim=0) * overlapped_node_rep, dim=1)
        shifted = [torch.repeat_interleave(substruct_rep[cycle_index(len(substruct_rep), i + 1).to(substruct_rep.device)], sizes, dim=0)
                   for i in range(neg_samples)]
        pred_neg = torch.sum(torch.cat(shifted, dim=0) * overlapped_node_rep.repeat((neg_samples, 1)), dim=1)
    else:
        raise ValueError("Invalid mode!")
    return pred_pos, pred_neg


def chem_contextpred_step(model_substruct, model_context, optimizer_substruct, optimizer_context, batch, neg_samples=1,
                          mode="cbow", pool=None, readback="end", accum=None):
    """One iteration of chem/pretrain_contextpred.py:51-100 (bio/pretrain_contextpred.py:46-95 is the same body);
    defaults = the reference's defaults (cbow, mean context pooling, one negative sample).  readback="epoch": the two numbers
    train() adds up on the host (:99-100) are added to ``accum`` (``epoch_accumulator``: [balanced loss sum, accuracy sum, -,
    steps]) on the device instead -- same float64 operations -- and nothing is fetched; returns None."""
    if readback not in ("end", "epoch"):
        raise ValueError("readback must be 'end' or 'epoch'")
    if readback == "epoch" and accum is None:
        raise ValueError("readback='epoch' needs accum=epoch_accumulator(device)")
    node_reps = None
    if mode == "cbow" and pool is None and batch.x_substruct.is_cuda and 1 <= neg_samples <= 8:
        # the reference's defaults: the whole loss in two launches forward and one back (csrc/contextpred.hip) instead of ~50
        # torch launches a few hundred elements long, between which the GPU idled
        # (Measured and NOT kept, profiles/r04/wgrad2p_and_ctx_two_streams_ab.txt: the context network on a second stream beside the
        # substructure network -- bit-identical, 1.66-1.70 against 1.64-1.65 ms per step: each network's backward already runs on
        # two streams, and four streams of 6 k-row kernels only stretch one another.)
        hs = model_substruct(batch.x_substruct, batch.edge_index_substruct, batch.edge_attr_substruct)
        hc = model_context(batch.x_context, batch.edge_index_context, batch.edge_attr_context)
        node_reps = (hs, hc)  # an ineligible shape falls through to the torch loss ON THESE embeddings (no second forward)
        if ops.contextpred_loss_eligible(hs, hc, batch.center_substruct_idx, neg_samples):
            loss, vals = ops.contextpred_loss(hs, batch.center_substruct_idx, hc, batch.overlap_context_substruct_idx,
                                              batch.batch_overlapped_context, neg_samples, accum if readback == "epoch" else None)
            optimizer_substruct.zero_grad()
            optimizer_context.zero_grad()
            _backward(loss)
            optimizer_substruct.step()
            optimizer_context.step()
            if readback == "epoch":
                return None
            v = vals.cpu().tolist()
            return v[0] + v[1], 0.5 * (v[2] + v[3])
    pred_pos, pred_neg = contextpred_logits(model_substruct, model_context, batch, neg_samples, mode, pool, node_reps)
    loss_pos = F.binary_cross_entropy_with_logits(pred_pos.double(), torch.ones_like(pred_pos).double())
    loss_neg = F.binary_cross_entropy_with_logits(pred_neg.double(), torch.zeros_like(pred_neg).double())
    optimizer_substruct.zero_grad()
    optimizer_context.zero_grad()
    loss = loss_pos + neg_samples * loss_neg
    _backward(loss)
    optimizer_substruct.step()
    optimizer_context.step()
    vals = torch.stack([loss_pos.detach(), loss_neg.detach(), torch.sum(pred_pos > 0).double() / len(pred_pos),
                        torch.sum(pred_neg < 0).double() / len(pred_neg)])
    if readback == "epoch":
        one = torch.ones((), dtype=torch.float64, device=vals.device)
        accum += torch.stack([vals[0] + vals[1], 0.5 * (vals[2] + vals[3]), one - one, one])
        return None
    vals = vals.cpu().tolist()
    return vals[0] + vals[1], 0.5 * (vals[2] + vals[3])


def chem_contextpred_epoch(model_substruct, model_context, optimizer_substruct, optimizer_context, loader, neg_samples=1, mode="cbow",
                           pool=None, device=None, readback="epoch"):
    """train() of chem/pretrain_contextpred.py:43-102 (bio: :39-97): one pass over ``loader``; returns (balanced_loss_accum / step,
    acc_accum / step) with the reference's divisor, the last step index.  Sums on the device by default, one fetch per epoch."""
    model_substruct.train()
    model_context.train()
    loss_accum = acc_accum = 0.0
    step = 0
    accum = None
    for step, batch in enumerate(loader):
        if device is not None:
            batch = batch.to(device)
        if readback == "epoch":
            if accum is None:
                accum = epoch_accumulator(batch.x_substruct.device)
            chem_contextpred_step(model_substruct, model_context, optimizer_substruct, optimizer_context, batch, neg_samples, mode, pool,
                                  readback, accum)
            continue
        loss, acc = chem_contextpred_step(model_substruct, model_context, optimizer_substruct, optimizer_context, batch, neg_samples,
                                          mode, pool, readback)
        loss_accum += loss
        acc_accum += acc
    if accum is not None:
        loss_accum, acc_accum, _, _ = accum.cpu().tolist()
    return loss_accum / step, acc_accum / step


bio_contextpred_step = chem_contextpred_step  # bio/pretrain_contextpred.py:39-102: same loop body, bio GNN classes
bio_contextpred_epoch = chem_contextpred_epoch


def chem_finetune_step(model, optimizer, batch):
    """One iteration of chem/finetune.py:27-49: ``GNN_graphpred`` forward (dropout active), float64
    BCE-with-logits over the non-null labels (y in {-1, 0 = missing, +1}), backward, optimizer step."""
    pred = model(batch.x, batch.edge_index, batch.edge_attr, batch.batch)
    y = batch.y.view(pred.shape).to(torch.float64)
    is_valid = y ** 2 > 0
    loss_mat = F.binary_cross_entropy_with_logits(pred.double(), (y + 1) / 2, reduction="none")
    loss_mat = torch.where(is_valid, loss_mat, torch.zeros_like(loss_mat))
    optimizer.zero_grad()
    loss = torch.sum(loss_mat) / torch.sum(is_valid)
    _backward(loss, unit=False)
    optimizer.step()
    return float(loss.detach().cpu().item())


def _roc_auc(labels01, scores):
    """ROC-AUC with midranks for ties (what sklearn.metrics.roc_auc_score returns; chem/finetune.py:73)"""
    import numpy as np
    from scipy.stats import rankdata
    labels01 = np.asarray(labels01)
    ranks = rankdata(np.asarray(scores, dtype=np.float64))
    pos = labels01 == 1
    n_pos, n_neg = int(pos.sum()), int((~pos).sum())
    return float((ranks[pos].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def chem_eval(model, batches):
    """chem/finetune.py:52-77: eval-mode scores for every batch, then the mean ROC-AUC over the tasks
    that have both classes, on their valid (non-zero) labels."""
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
            roc_list.append(_roc_auc((y_true[is_valid, i] + 1) / 2, y_scores[is_valid, i]))
    return sum(roc_list) / len(roc_list)


def bio_finetune_step(model, optimizer, batch):
    """One iteration of bio/finetune.py:25-37: bio ``GNN_graphpred`` on the batch object (mean / sum pooling concatenated with
    the centre node's row, bio/model.py:338-347), float64 BCE-with-logits against ``go_target_downstream`` viewed as
    [graphs, tasks], backward, optimizer step."""
    pred = model(batch)
    y = batch.go_target_downstream.view(pred.shape).to(torch.float64)
    optimizer.zero_grad()
    loss = F.binary_cross_entropy_with_logits(pred.double(), y)
    _backward(loss, unit=False)
    optimizer.step()
    return float(loss.detach().cpu().item())


def bio_eval(model, batches):
    """bio/finetune.py:40-65: eval-mode scores for every batch, then one ROC-AUC per task (nan where a task shows a single
    class) -- the array the reference's eval() returns."""
    import numpy as np
    model.eval()
    y_true, y_scores = [], []
    for batch in batches:
        with torch.no_grad():
            pred = model(batch)
        y_true.append(batch.go_target_downstream.view(pred.shape))
        y_scores.append(pred)
    y_true = torch.cat(y_true, dim=0).cpu().numpy()
    y_scores = torch.cat(y_scores, dim=0).cpu().numpy()
    roc_list = []
    for i in range(y_true.shape[1]):
        if np.sum(y_true[:, i] == 1) > 0 and np.sum(y_true[:, i] == 0) > 0:
            roc_list.append(_roc_auc(y_true[:, i], y_scores[:, i]))
        else:
            roc_list.append(np.nan)
    return np.array(roc_list)


def chem_edgepred_step(model, optimizer, batch):
    """One iteration of chem/pretrain_edgepred.py:32-46: dot-product scores of the bonded atom pairs (one
    direction per bond) against the batch's sampled non-bonded pairs, BCE-with-logits (float32)."""
    node_emb = model(batch.x, batch.edge_index, batch.edge_attr)
    positive_score = torch.sum(node_emb[batch.edge_index[0, ::2]] * node_emb[batch.edge_index[1, ::2]], dim=1)
    negative_score = torch.sum(node_emb[batch.negative_edge_index[0]] * node_emb[batch.negative_edge_index[1]], dim=1)
    optimizer.zero_grad()
    loss = (F.binary_cross_entropy_with_logits(positive_score, torch.ones_like(positive_score))
            + F.binary_cross_entropy_with_logits(negative_score, torch.zeros_like(negative_score)))
    _backward(loss, unit=False)
    optimizer.step()
    acc = (torch.sum(positive_score > 0) + torch.sum(negative_score < 0)).to(torch.float32) / float(2 * len(positive_score))
    out = torch.stack([loss.detach(), acc]).cpu().tolist()
    return out[0], out[1]


class Discriminator(torch.nn.Module):
    """Bilinear node-vs-summary score of chem/pretrain_deepgraphinfomax.py:30-42 (same parameter name, same
    U(-1/sqrt(D), 1/sqrt(D)) initialisation as torch_geometric's ``uniform``)."""

    def __init__(self, hidden_dim):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.Tensor(hidden_dim, hidden_dim))
        self.reset_parameters()

    def reset_parameters(self):
        bound = 1.0 / (self.weight.size(0) ** 0.5)
        self.weight.data.uniform_(-bound, bound)

    def forward(self, x, summary):
        return torch.sum(x * torch.matmul(summary, self.weight), dim=1)


class Infomax(torch.nn.Module):
    """chem/pretrain_deepgraphinfomax.py:44-50: holder of the GNN, the discriminator and the mean pooling."""

    def __init__(self, gnn, discriminator):
        super().__init__()
        self.gnn, self.discriminator = gnn, discriminator
        self.loss = torch.nn.BCEWithLogitsLoss()
        self.pool = ops.global_mean_pool


def chem_infomax_step(model, optimizer, batch):
    """One iteration of chem/pretrain_deepgraphinfomax.py:61-84 on an ``Infomax`` model."""
    node_emb = model.gnn(batch.x, batch.edge_index, batch.edge_attr)
    summary_emb = torch.sigmoid(model.pool(node_emb, batch.batch))
    positive_expanded = summary_emb[batch.batch]
    shifted = summary_emb[cycle_index(len(summary_emb), 1).to(summary_emb.device)]
    negative_expanded = shifted[batch.batch]
    positive_score = model.discriminator(node_emb, positive_expanded)
    negative_score = model.discriminator(node_emb, negative_expanded)
    optimizer.zero_grad()
    loss = model.loss(positive_score, torch.ones_like(positive_score)) + model.loss(negative_score, torch.zeros_like(negative_score))
    _backward(loss, unit=False)
    optimizer.step()
    acc = (torch.sum(positive_score > 0) + torch.sum(negative_score < 0)).to(torch.float32) / float(2 * len(positive_score))
    out = torch.stack([loss.detach(), acc]).cpu().tolist()
    return out[0], out[1]


def bio_edgepred_step(model, optimizer, batch):
    """One iteration of bio/pretrain_edgepred.py:26-40: the loop body is the chem script's (scores of one direction
    of every PPI edge against the NegativeEdge pairs of bio/util.py:16-44) on a bio ``GNN``."""
    return chem_edgepred_step(model, optimizer, batch)


def bio_infomax_step(model, optimizer, batch):
    """One iteration of bio/pretrain_deepgraphinfomax.py:59-81 on an ``Infomax`` model holding a bio ``GNN``."""
    return chem_infomax_step(model, optimizer, batch)


def _pair_epoch(step, model, optimizer, loader, device, count_last_index):
    acc_sum = loss_sum = 0.0
    steps = 0
    for batch in loader:
        loss, acc = step(model, optimizer, batch.to(device) if device is not None else batch)
        loss_sum += loss
        acc_sum += acc
        steps += 1
    div = steps - 1 if count_last_index else steps
    return acc_sum / div, loss_sum / div


def chem_edgepred_epoch(model, optimizer, loader, device=None):
    """chem/pretrain_edgepred.py:25-52 train(): returns (accuracy, loss) sums divided by the LAST step index (:52 divides by
    ``step``, not ``step + 1``) -- kept, so that logged curves line up with the reference's."""
    return _pair_epoch(chem_edgepred_step, model, optimizer, loader, device, True)


def chem_infomax_epoch(model, optimizer, loader, device=None):
    """chem/pretrain_deepgraphinfomax.py:52-90 train(): the same last-index divisor (:90)."""
    return _pair_epoch(chem_infomax_step, model, optimizer, loader, device, True)


def bio_edgepred_epoch(model, optimizer, loader, device=None):
    """bio/pretrain_edgepred.py:20-43 train(): sums divided by the step count (:43)."""
    return _pair_epoch(bio_edgepred_step, model, optimizer, loader, device, False)


def bio_infomax_epoch(model, optimizer, loader, device=None):
    """bio/pretrain_deepgraphinfomax.py:52-84 train(): sums divided by the step count (:84)."""
    return _pair_epoch(bio_infomax_step, model, optimizer, loader, device, False)

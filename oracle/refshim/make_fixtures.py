"""Generate tests/golden/ref_*.npz by RUNNING THE REFERENCE'S OWN CODE.  Test infrastructure only.

    python -m oracle.refshim.make_fixtures          (build container only: needs /root/reference)

Everything numerical or structural in these fixtures is produced by the unmodified sources under
/root/reference (imported through oracle/refshim, third-party modules replaced by the stand-ins of
pyg103.py): `model.GNN` / `GNN_graphpred` forward + backward, `batch.BatchMasking` /
`BatchSubstructContext.from_data_list`, `util.MaskAtom` / `MaskEdge` / `ExtractSubstructureContextPair`,
`util.NegativeEdge` / `batch.BatchAE`, bio `batch.BatchFinetune`, and the `train()` / `eval()` functions of pretrain_masking.py,
pretrain_contextpred.py and finetune.py (chem and bio), chem pretrain_edgepred.py and pretrain_deepgraphinfomax.py.  Only the RAW synthetic graphs (SURVEY.md §8d shapes) come from this repository's
generator; they are stored in the fixture, so the tests never regenerate them.

The fixtures are what `/root/reference` leaves behind for the GPU box, where it does not exist:
  tests/test_cpu_reference.py   oracle == fixtures (everywhere) and fixtures == live reference (here)
  tests/test_gpu_reference.py   HIP path == fixtures, through the C ABI
"""
import argparse
import os
import random
import sys

import numpy as np
import torch

from oracle import refshim
from pretrain_gnns_amd.data import synthetic

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
BIG = 40000          # parameter tensors above this many elements are stored as a seeded sample
SAMPLE = 4096


# ----------------------------------------------------------------------------- helpers
def _np(v):
    if torch.is_tensor(v):
        v = v.detach().cpu().numpy()
    v = np.asarray(v)
    if v.dtype == np.int64 and v.size and np.abs(v).max() < 2 ** 31:
        v = v.astype(np.int32)  # the loader of tests/ref_fixtures.py widens these back to int64
    return v


def save(name, tree):
    flat = {}

    def walk(prefix, node):
        if isinstance(node, dict):
            for k, v in node.items():
                walk(prefix + "/" + str(k) if prefix else str(k), v)
        else:
            flat[prefix] = _np(node)

    walk("", tree)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **flat)
    print("%-34s %8.1f KB  %d arrays" % (name + ".npz", os.path.getsize(path) / 1024.0, len(flat)))


def sample_positions(numel, tag):
    g = np.random.default_rng(abs(hash_str(tag)) % (2 ** 32))
    return np.sort(g.choice(numel, SAMPLE, replace=False))


def hash_str(s):
    h = 2166136261
    for ch in s.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


def pack_params(named, what):
    """every tensor's fp64 L2 norm; small tensors in full, large ones at SAMPLE seeded flat positions"""
    out = {}
    for n, t in named:
        t = what(t)
        if t is None:
            continue
        t = t.detach().reshape(-1)
        out[n + "|norm"] = np.float64(t.double().norm().item())
        if t.numel() <= BIG:
            out[n + "|full"] = t
        else:
            pos = sample_positions(t.numel(), n)
            out[n + "|pos"] = pos
            out[n + "|val"] = t[torch.from_numpy(pos)]
    return out


def rows_sample(t, tag, k=256):
    g = np.random.default_rng(hash_str(tag))
    rows = np.sort(g.choice(t.size(0), min(k, t.size(0)), replace=False))
    return {"rows": rows, "vals": t.detach()[torch.from_numpy(rows)], "colsum": t.detach().double().sum(0),
            "abssum": np.float64(t.detach().double().abs().sum().item())}


def raw_pack(graphs, keys=("x", "edge_index", "edge_attr")):
    """concatenated (data, slices) storage of the raw graphs, local node ids"""
    out = {"node_slices": np.cumsum([0] + [g.x.size(0) for g in graphs]),
           "edge_slices": np.cumsum([0] + [g.edge_index.size(1) for g in graphs])}
    for k in keys:
        out[k] = torch.cat([getattr(g, k) for g in graphs], dim=-1 if k == "edge_index" else 0)
    if out["edge_attr"].dtype == torch.float32:  # bio: 0/1 flags
        out["edge_attr"] = out["edge_attr"].to(torch.uint8)
    return out


def batch_pack(batch):
    out = {}
    for k in batch.keys:
        v = batch[k]
        if torch.is_tensor(v):
            out[k] = v.to(torch.uint8) if (v.dtype == torch.float32 and k.startswith("edge_attr") or k == "mask_edge_label"
                                          and v.dtype == torch.float32) else v
    return out


class Recorder:
    """wraps a module-level callable of a reference script (criterion, compute_accuracy) and keeps what it
    returned -- the script's source is untouched, only its module attribute is rebound while train() runs"""

    def __init__(self, mod, name):
        self.mod, self.name, self.inner, self.values, self.inputs = mod, name, getattr(mod, name), [], []

    def __call__(self, *a, **k):
        r = self.inner(*a, **k)
        self.values.append(r.detach().clone() if torch.is_tensor(r) else r)
        self.inputs.append(a[0].detach().clone() if torch.is_tensor(a[0]) else None)
        return r

    def __enter__(self):
        setattr(self.mod, self.name, self)
        return self

    def __exit__(self, *exc):
        setattr(self.mod, self.name, self.inner)


def to_ref_data(ref, g, extra=()):
    d = ref.batch.Data(x=g.x.clone(), edge_index=g.edge_index.clone(), edge_attr=g.edge_attr.clone())
    for k in extra:
        setattr(d, k, getattr(g, k).clone())
    return d


def adam(params):
    return torch.optim.Adam(params, lr=0.001, weight_decay=0)


# ----------------------------------------------------------------------------- chem: masking
def chem_raw(num_graphs, seed):
    rng = np.random.default_rng(seed)
    return [synthetic.zinc_like_graph(rng) for _ in range(num_graphs)]


def chem_masked_graphs(ref, raw, seed, mask_edge):
    """util.MaskAtom.__call__ (chem/util.py:207-277) on every graph, python `random` seeded"""
    random.seed(seed)
    tf = ref.util.MaskAtom(num_atom_type=119, num_edge_type=5, mask_rate=0.15, mask_edge=mask_edge)
    return [tf(to_ref_data(ref, g)) for g in raw]


def chem_models(ref, gnn_type, seed=0, num_layer=5):
    torch.manual_seed(seed)
    model = ref.model.GNN(num_layer, 300, JK="last", drop_ratio=0, gnn_type=gnn_type)
    atoms = torch.nn.Linear(300, 119)
    bonds = torch.nn.Linear(300, 4)
    return [model, atoms, bonds]


def chem_forward_backward(ref, batch, gnn_type, full_rows):
    model, atoms, _ = chem_models(ref, gnn_type)
    model.train()
    h = model(batch.x, batch.edge_index, batch.edge_attr)
    logits = atoms(h[batch.masked_atom_indices])
    loss = ref.pretrain_masking.criterion(logits.double(), batch.mask_node_label[:, 0])
    loss.backward()
    acc = ref.pretrain_masking.compute_accuracy(logits, batch.mask_node_label[:, 0])
    model.eval()
    with torch.no_grad():
        h_eval = model(batch.x, batch.edge_index, batch.edge_attr)
    named = list(model.named_parameters()) + [("head." + n, p) for n, p in atoms.named_parameters()]
    out = {"loss": np.float64(loss.item()), "acc": np.float64(acc), "logits": logits,
           "grads": pack_params(named, lambda p: p.grad),
           "bn_running_mean_4": model.batch_norms[4].running_mean, "bn_running_var_4": model.batch_norms[4].running_var}
    if full_rows and gnn_type == "gin":
        out["out_train"], out["out_eval"] = h, rows_sample(h_eval, "eval" + gnn_type)
    else:
        out["out_train"], out["out_eval"] = rows_sample(h, "train" + gnn_type), rows_sample(h_eval, "eval" + gnn_type)
    # the same reference code in float64 (model.double()): what an fp32 implementation should be measured against when
    # the bar is tighter than fp32-vs-fp32 rounding (a ReLU input within rounding of zero flips between any two fp32 runs)
    model64, atoms64, _ = chem_models(ref, gnn_type)
    model64.double(), atoms64.double()
    model64.train()
    h64 = model64(batch.x, batch.edge_index, batch.edge_attr)
    logits64 = atoms64(h64[batch.masked_atom_indices])
    ref.pretrain_masking.criterion(logits64, batch.mask_node_label[:, 0]).backward()
    named64 = list(model64.named_parameters()) + [("head." + n, p) for n, p in atoms64.named_parameters()]
    out["f64"] = {"logits": logits64.float(), "grads": pack_params(named64, lambda p: p.grad.float()),
                  "out_train": rows_sample(h64.float(), "train" + gnn_type)}
    return out


def chem_train_sequence(ref, graphs, batch_size, gnn_type, mask_edge, steps):
    """pretrain_masking.train (chem/pretrain_masking.py:34-78) over `steps` batches"""
    pm = ref.pretrain_masking
    loader = ref.dataloader.DataLoaderMasking(graphs[:steps * batch_size], batch_size=batch_size, shuffle=False, num_workers=0)
    models = chem_models(ref, gnn_type)
    opts = [adam(m.parameters()) for m in models]
    args = argparse.Namespace(mask_edge=mask_edge)
    with Recorder(pm, "criterion") as crit, Recorder(pm, "compute_accuracy") as accs:
        ret = pm.train(args, models, loader, opts, torch.device("cpu"))
    per = 2 if mask_edge else 1
    loss_terms = np.array([float(v) for v in crit.values]).reshape(steps, per)
    acc_terms = np.array(accs.values, dtype=np.float64).reshape(steps, per)
    named = list(models[0].named_parameters())
    return {"returned": np.array(ret, dtype=np.float64), "loss_terms": loss_terms, "acc_terms": acc_terms,
            "loss": loss_terms.sum(1), "final_params": pack_params(named, lambda p: p),
            "final_head_weight": models[1].weight, "bn_running_mean_0": models[0].batch_norms[0].running_mean,
            "bn_running_var_4": models[0].batch_norms[4].running_var}


def make_chem_masking(ref):
    for name, num_graphs, full in (("ref_chem_masking_b32", 32, True), ("ref_chem_masking_b256", 256, False)):
        raw = chem_raw(num_graphs, seed=0)
        fx = {"raw": raw_pack(raw)}
        for mask_edge in (0, 1):
            graphs = chem_masked_graphs(ref, raw, seed=1, mask_edge=mask_edge)
            batch = ref.batch.BatchMasking.from_data_list(graphs)
            tag = "mask_edge%d" % mask_edge
            fx[tag] = {"batch": batch_pack(batch),
                       "mask_counts": np.array([g.masked_atom_indices.numel() for g in graphs]),
                       "mask_local": torch.cat([g.masked_atom_indices for g in graphs])}
            if mask_edge == 0:
                for gt in ("gin", "gcn"):
                    fx[tag][gt] = chem_forward_backward(ref, batch, gt, full)
        save(name, fx)

    # train() sequences: 5 x 32 graphs (gin, gin + bond masking, gcn) and 10 x 256 graphs (gin)
    raw = chem_raw(5 * 32, seed=2)
    fx = {"raw": raw_pack(raw), "batch_size": 32}
    for tag, gt, me in (("gin", "gin", 0), ("gin_mask_edge", "gin", 1), ("gcn", "gcn", 0)):
        graphs = chem_masked_graphs(ref, raw, seed=3, mask_edge=me)
        fx[tag] = chem_train_sequence(ref, graphs, 32, gt, me, steps=5)
        fx[tag]["mask_counts"] = np.array([g.masked_atom_indices.numel() for g in graphs])
        fx[tag]["mask_local"] = torch.cat([g.masked_atom_indices for g in graphs])
    save("ref_chem_masking_train_b32", fx)
    raw = chem_raw(10 * 256, seed=4)  # ten steps: the +-0.1 % epoch-accuracy bar of BASELINE.json is checked on this run
    graphs = chem_masked_graphs(ref, raw, seed=5, mask_edge=0)
    fx = {"raw": raw_pack(raw), "batch_size": 256, "gin": chem_train_sequence(ref, graphs, 256, "gin", 0, steps=10)}
    fx["gin"]["mask_counts"] = np.array([g.masked_atom_indices.numel() for g in graphs])
    fx["gin"]["mask_local"] = torch.cat([g.masked_atom_indices for g in graphs])
    save("ref_chem_masking_train_b256", fx)


# ----------------------------------------------------------------------------- chem: the [13, 12] molecule of chem/util.py:365-419
def spec_molecule(ref):
    """'C#Cc1c(O)c(Cl)cc(/C=C/N)c1S' hand-encoded the way loader.mol_to_graph_data_obj_simple (chem/loader.py:53-100)
    lays a molecule out (rdkit is absent): atoms in SMILES order, [atomic number - 1, chirality 0]; every bond as two
    adjacent directed edges with identical [bond type, bond direction]; bonds in SMILES order with the ring closure
    last (the order does not enter any of the reference's assertions)."""
    z = [6, 6, 6, 6, 8, 6, 17, 6, 6, 6, 6, 7, 6, 16]
    S, D, T, A = 0, 1, 2, 3
    bonds = [(0, 1, T, 0), (1, 2, S, 0), (2, 3, A, 0), (3, 4, S, 0), (3, 5, A, 0), (5, 6, S, 0), (5, 7, A, 0), (7, 8, A, 0),
             (8, 9, S, 1), (9, 10, D, 0), (10, 11, S, 1), (8, 12, A, 0), (12, 13, S, 0), (12, 2, A, 0)]
    x = torch.tensor([[a - 1, 0] for a in z], dtype=torch.long)
    ei, ea = [], []
    for u, v, t, d in bonds:
        ei += [(u, v), (v, u)]
        ea += [[t, d], [t, d]]
    return ref.batch.Data(x=x, edge_index=torch.tensor(ei, dtype=torch.long).t().contiguous(),
                          edge_attr=torch.tensor(ea, dtype=torch.long))


def make_chem_spec(ref):
    fx = {}
    mol = spec_molecule(ref)
    fx["molecule"] = {"x": mol.x, "edge_index": mol.edge_index, "edge_attr": mol.edge_attr}
    for me in (False, True):
        d = spec_molecule(ref)
        ref.util.MaskAtom(118, 5, 0.1, mask_edge=me)(d, [13, 12])
        fx["mask_edge%d" % me] = {k: d[k] for k in d.keys}
    for tag, (k, l1, l2) in {"k2_l1_1_l2_3": (2, 1, 3), "k1_l1_1_l2_10000": (1, 1, 10000), "k5_l1_4_l2_7": (5, 4, 7)}.items():
        d = spec_molecule(ref)
        ref.util.ExtractSubstructureContextPair(k, l1, l2)(d, 13)
        fx["context_" + tag] = {key: d[key] for key in d.keys}
    save("ref_chem_spec_molecule", fx)


# ----------------------------------------------------------------------------- chem: context prediction
def context_orders(ref, data, root, k, l1, l2):
    """the node orders networkx gives the induced substructure / context graphs inside
    ExtractSubstructureContextPair (chem/util.py:96-149): re-derived with the same calls so that a fixture
    reader can map the renumbered graphs back to the molecule's atoms"""
    import networkx as nx
    G = ref.loader.graph_data_obj_to_nx_simple(data)
    sub = nx.single_source_shortest_path_length(G, root, k).keys()
    c1 = nx.single_source_shortest_path_length(G, root, l1).keys()
    c2 = nx.single_source_shortest_path_length(G, root, l2).keys()
    ctx = set(c1).symmetric_difference(set(c2))
    return list(G.subgraph(sub).nodes()), (list(G.subgraph(ctx).nodes()) if len(ctx) else [])


def chem_context_graphs(ref, raw, seed, k=5, l1=4, l2=7):
    rng = np.random.default_rng(seed)
    tf = ref.util.ExtractSubstructureContextPair(k, l1, l2)
    graphs, roots, sub_order, ctx_order = [], [], [], []
    for g in raw:
        root = int(rng.integers(0, g.x.size(0)))
        d = to_ref_data(ref, g)
        so, co = context_orders(ref, d, root, k, l1, l2)
        graphs.append(tf(d, root))
        roots.append(root)
        sub_order.append(so)
        ctx_order.append(co)
    return graphs, np.array(roots), sub_order, ctx_order


def ragged(lists):
    return {"slices": np.cumsum([0] + [len(x) for x in lists]), "values": np.array([v for x in lists for v in x], dtype=np.int64)}


def context_train_sequence(ref, graphs, batch_size, steps, mode, num_layer=5, csize=3, domain="chem"):
    pc = ref.pretrain_contextpred
    torch.manual_seed(0)
    model_substruct = ref.model.GNN(num_layer, 300, JK="last", drop_ratio=0, gnn_type="gin")
    # chem: GNN(int(l2 - l1)) with l2 = l1 + csize (chem/pretrain_contextpred.py:145-157); bio: GNN(3) (bio/pretrain_contextpred.py:131)
    model_context = ref.model.GNN(csize, 300, JK="last", drop_ratio=0, gnn_type="gin")
    opt_s, opt_c = adam(model_substruct.parameters()), adam(model_context.parameters())
    loader = ref.dataloader.DataLoaderSubstructContext(graphs[:steps * batch_size], batch_size=batch_size, shuffle=False, num_workers=0)
    args = argparse.Namespace(mode=mode, context_pooling="mean", neg_samples=1)
    with Recorder(pc, "criterion") as crit:
        ret = pc.train(args, model_substruct, model_context, loader, opt_s, opt_c, torch.device("cpu"))
    terms = np.array([float(v) for v in crit.values]).reshape(steps, 2)
    out = {"returned": np.array(ret, dtype=np.float64), "loss_pos": terms[:, 0], "loss_neg": terms[:, 1],
           "pred_pos_step0": crit.inputs[0].float(), "pred_neg_step0": crit.inputs[1].float(),
           "final_params_substruct": pack_params(list(model_substruct.named_parameters()), lambda p: p),
           "final_params_context": pack_params(list(model_context.named_parameters()), lambda p: p)}
    return out


def make_chem_contextpred(ref):
    for name, num_graphs, bs, steps in (("ref_chem_contextpred_b32", 4 * 32, 32, 4), ("ref_chem_contextpred_b256", 3 * 256, 256, 3)):
        raw = chem_raw(num_graphs, seed=6)
        graphs, roots, sub_order, ctx_order = chem_context_graphs(ref, raw, seed=7)
        fx = {"raw": raw_pack(raw), "roots": roots, "batch_size": bs, "steps": steps,
              "sub_order": ragged(sub_order), "ctx_order": ragged(ctx_order),
              "has_context": np.array([hasattr(g, "x_context") for g in graphs]),
              "has_overlap": np.array([hasattr(g, "overlap_context_substruct_idx") for g in graphs])}
        fx["batches"] = {str(i): batch_pack(ref.batch.BatchSubstructContext.from_data_list(graphs[i * bs:(i + 1) * bs]))
                         for i in range(steps)}
        fx["cbow"] = context_train_sequence(ref, graphs, bs, steps, "cbow")
        if bs == 32:
            fx["skipgram"] = context_train_sequence(ref, graphs, bs, steps, "skipgram")
        save(name, fx)


# ----------------------------------------------------------------------------- chem: fine-tuning
def make_chem_finetune(ref):
    ft = ref.finetune
    rng = np.random.default_rng(8)
    raw, ys = [], []
    for _ in range(4 * 32):
        g = synthetic.zinc_like_graph(rng)
        y = rng.choice([-1, 1], size=12)
        y[rng.random(12) < 0.2] = 0
        g.y = torch.from_numpy(y.astype(np.int64))
        raw.append(g)
        ys.append(y)
    graphs = [to_ref_data(ref, g, extra=("y",)) for g in raw]
    fx = {"raw": raw_pack(raw), "y": np.stack(ys), "batch_size": 32}
    for pooling in ("mean", "sum"):
        torch.manual_seed(0)
        model = ref.model.GNN_graphpred(5, 300, 12, JK="last", drop_ratio=0, graph_pooling=pooling, gnn_type="gin")
        opt = adam(model.parameters())
        loader = ft.DataLoader(graphs, batch_size=32, shuffle=False, num_workers=0)  # torch_geometric.data.DataLoader, chem/finetune.py:12
        with Recorder(ft, "criterion") as crit:
            ft.train(argparse.Namespace(), model, torch.device("cpu"), loader, opt)
        first_pred = crit.inputs[0].float()
        losses = []
        for lm, b in zip(crit.values, loader):
            y = b.y.view(lm.shape).double()
            valid = y ** 2 > 0
            losses.append(float((torch.where(valid, lm, torch.zeros_like(lm)).sum() / valid.sum()).item()))
        auc = ft.eval(argparse.Namespace(), model, torch.device("cpu"), loader)
        model.eval()
        with torch.no_grad():
            b0 = next(iter(loader))
            pred_eval = model(b0.x, b0.edge_index, b0.edge_attr, b0.batch)
        fx[pooling] = {"pred_step0": first_pred, "loss": np.array(losses), "roc_auc": np.float64(auc), "pred_eval_batch0": pred_eval,
                       "final_params": pack_params(list(model.named_parameters()), lambda p: p)}
    save("ref_chem_finetune_b32", fx)


# ----------------------------------------------------------------------------- bio
def bio_raw(num_graphs, seed):
    rng = np.random.default_rng(seed)
    return [synthetic.ppi_like_graph(rng) for _ in range(num_graphs)]


def to_ref_bio(ref, g):
    d = ref.batch.Data(x=g.x.clone(), edge_index=g.edge_index.clone(), edge_attr=g.edge_attr.clone())
    d.center_node_idx = g.center_node_idx.clone()
    return d


def bio_models(ref, gnn_type, num_layer=5):
    torch.manual_seed(0)
    model = ref.model.GNN(num_layer, 300, JK="last", drop_ratio=0, gnn_type=gnn_type)
    head = torch.nn.Linear(300, 7)
    return [model, head]


def make_bio_masking(ref):
    pm = ref.pretrain_masking
    for name, num_graphs, bs, steps, types in (("ref_bio_masking_b8", 4 * 8, 8, 4, ("gin", "gcn")),
                                               ("ref_bio_masking_b256", 2 * 256, 256, 2, ("gin",))):
        raw = bio_raw(num_graphs, seed=9)
        random.seed(10)
        tf = ref.util.MaskEdge(mask_rate=0.15)
        graphs = [tf(to_ref_bio(ref, g)) for g in raw]
        fx = {"raw": raw_pack(raw), "batch_size": bs, "steps": steps,
              "mask_counts": np.array([g.masked_edge_idx.numel() for g in graphs]),
              "mask_local": torch.cat([g.masked_edge_idx for g in graphs])}
        batch = ref.batch.BatchMasking.from_data_list(graphs[:bs])
        fx["batch0"] = batch_pack(batch)
        for gt in types:
            model, head = bio_models(ref, gt)
            model.train()
            h = model(batch.x, batch.edge_index, batch.edge_attr)
            mei = batch.edge_index[:, batch.masked_edge_idx]
            logits = head(h[mei[0]] + h[mei[1]])
            label = torch.argmax(batch.mask_edge_label, dim=1)
            loss = pm.criterion(logits, label)
            loss.backward()
            named = list(model.named_parameters()) + [("head." + n, p) for n, p in head.named_parameters()]
            m64, h64m = bio_models(ref, gt)
            m64.double(), h64m.double()
            m64.train()
            hh = m64(batch.x.double(), batch.edge_index, batch.edge_attr.double())
            lg64 = h64m(hh[mei[0]] + hh[mei[1]])
            pm.criterion(lg64, label).backward()
            named64 = list(m64.named_parameters()) + [("head." + n, p) for n, p in h64m.named_parameters()]
            f64 = {"logits": lg64.float() if bs <= 8 else rows_sample(lg64.float(), "biologits", 2048),
                   "out_train": rows_sample(hh.float(), "bio" + gt),
                   "grads": pack_params(named64, lambda p: p.grad.float())}
            one = {"f64": f64, "loss": np.float64(loss.item()), "acc": np.float64(pm.compute_accuracy(logits, label)),
                   "logits": logits if bs <= 8 else rows_sample(logits, "biologits", 2048),
                   "out_train": h if bs <= 8 else rows_sample(h, "bio" + gt), "grads": pack_params(named, lambda p: p.grad)}
            models = bio_models(ref, gt)
            opts = [adam(m.parameters()) for m in models]
            loader = ref.dataloader.DataLoaderMasking(graphs, batch_size=bs, shuffle=False, num_workers=0)
            with Recorder(pm, "criterion") as crit, Recorder(pm, "compute_accuracy") as accs:
                ret = pm.train(argparse.Namespace(), models, loader, opts, torch.device("cpu"))
            one["train"] = {"returned": np.array(ret, dtype=np.float64), "loss": np.array([float(v) for v in crit.values]),
                            "acc": np.array(accs.values, dtype=np.float64),
                            "final_params": pack_params(list(models[0].named_parameters()), lambda p: p)}
            fx[gt] = one
        save(name, fx)


def bio_context_orders(ref, data, root, l1):
    import networkx as nx
    G = ref.loader.graph_data_obj_to_nx(data)
    c1 = nx.single_source_shortest_path_length(G, root, l1).keys()
    ctx = set(c1).symmetric_difference(set(range(data.x.size(0))))
    return list(G.subgraph(ctx).nodes()) if len(ctx) else []


def make_bio_contextpred(ref):
    for name, num_graphs, bs, steps in (("ref_bio_contextpred_b8", 4 * 8, 8, 4), ("ref_bio_contextpred_b64", 2 * 64, 64, 2)):
        raw = bio_raw(num_graphs, seed=11)
        tf = ref.util.ExtractSubstructureContextPair(l1=1, center=True)
        graphs, orders = [], []
        for g in raw:
            d = to_ref_bio(ref, g)
            orders.append(bio_context_orders(ref, d, int(d.center_node_idx.item()), tf.l1))
            graphs.append(tf(d))
        fx = {"raw": raw_pack(raw), "batch_size": bs, "steps": steps, "ctx_order": ragged(orders),
              "has_context": np.array([hasattr(g, "x_context") for g in graphs])}
        fx["batches"] = {str(i): batch_pack(ref.batch.BatchSubstructContext.from_data_list(graphs[i * bs:(i + 1) * bs]))
                         for i in range(steps)}
        fx["cbow"] = context_train_sequence(ref, graphs, bs, steps, "cbow", domain="bio")
        save(name, fx)


# ----------------------------------------------------------------------------- chem: edge prediction, Deep Graph Infomax
def make_chem_edgepred(ref):
    """chem/util.py NegativeEdge -> chem/batch.py BatchAE (chem/dataloader.py DataLoaderAE) -> chem/pretrain_edgepred.py:25-52 train()"""
    pe = ref.pretrain_edgepred
    raw = chem_raw(4 * 32, seed=12)
    torch.manual_seed(13)  # NegativeEdge draws its candidates with torch.randint
    tf = ref.util.NegativeEdge()
    graphs = [tf(to_ref_data(ref, g)) for g in raw]
    fx = {"raw": raw_pack(raw), "batch_size": 32, "neg": ragged([g.negative_edge_index.t().reshape(-1).tolist() for g in graphs])}
    loader = ref.dataloader.DataLoaderAE(graphs, batch_size=32, shuffle=False, num_workers=0)
    fx["batch0"] = batch_pack(next(iter(loader)))
    for gt in ("gin", "gcn"):
        torch.manual_seed(0)
        model = ref.model.GNN(5, 300, JK="last", drop_ratio=0, gnn_type=gt)
        opt = adam(model.parameters())
        with Recorder(pe, "criterion") as crit:
            ret = pe.train(argparse.Namespace(), model, torch.device("cpu"), loader, opt)
        vals = [float(v) for v in crit.values]  # two calls per step: positive pairs, negative pairs
        fx[gt] = {"returned": np.array(ret, dtype=np.float64), "loss": np.array([a + b for a, b in zip(vals[0::2], vals[1::2])]),
                  "final_params": pack_params(list(model.named_parameters()), lambda p: p)}
    save("ref_chem_edgepred_b32", fx)


def make_chem_infomax(ref):
    """torch_geometric DataLoader (chem/pretrain_deepgraphinfomax.py:4) -> its Infomax / Discriminator -> train() :52-90"""
    dgi = ref.pretrain_deepgraphinfomax
    raw = chem_raw(4 * 32, seed=14)
    graphs = [to_ref_data(ref, g) for g in raw]
    loader = dgi.DataLoader(graphs, batch_size=32, shuffle=False, num_workers=0)
    torch.manual_seed(0)
    gnn = ref.model.GNN(5, 300, JK="last", drop_ratio=0, gnn_type="gin")
    model = dgi.Infomax(gnn, dgi.Discriminator(300))
    disc0 = model.discriminator.weight.detach().clone()
    opt = adam(model.parameters())
    vals = []

    class RecordingLoss(torch.nn.Module):  # `loss` is a registered child module: its stand-in has to be one too
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, *a):
            r = self.inner(*a)
            vals.append(float(r))
            return r

    model.loss = RecordingLoss(model.loss)  # an instance attribute: the script's source is untouched
    ret = dgi.train(argparse.Namespace(), model, torch.device("cpu"), loader, opt)
    fx = {"raw": raw_pack(raw), "batch_size": 32, "discriminator_init": disc0, "returned": np.array(ret, dtype=np.float64),
          "loss": np.array([a + b for a, b in zip(vals[0::2], vals[1::2])]),
          "final_params": pack_params(list(model.gnn.named_parameters()) + [("discriminator.weight", model.discriminator.weight)], lambda p: p)}
    save("ref_chem_infomax_b32", fx)


# ----------------------------------------------------------------------------- bio: fine-tuning
def make_bio_finetune(ref):
    """bio/batch.py BatchFinetune (bio/dataloader.py DataLoaderFinetune) -> bio/model.py GNN_graphpred :293-347 ->
    bio/finetune.py:25-65 train() / eval()"""
    ft = ref.finetune
    rng = np.random.default_rng(15)
    raw, ys = [], []
    for _ in range(4 * 32):
        g = synthetic.ppi_like_graph(rng)
        ys.append((rng.random(40) < 0.3).astype(np.int64))
        raw.append(g)
    graphs = []
    for g, y in zip(raw, ys):
        d = to_ref_bio(ref, g)
        d.go_target_downstream = torch.from_numpy(y)
        graphs.append(d)
    fx = {"raw": raw_pack(raw), "y": np.stack(ys), "batch_size": 32}
    loader = ref.dataloader.DataLoaderFinetune(graphs, batch_size=32, shuffle=False, num_workers=0)
    b0 = next(iter(loader))
    fx["batch0"] = {"center_node_idx": b0.center_node_idx, "batch": b0.batch}
    for pooling in ("mean", "sum"):
        torch.manual_seed(0)
        model = ref.model.GNN_graphpred(5, 300, 40, JK="last", drop_ratio=0, graph_pooling=pooling, gnn_type="gin")
        opt = adam(model.parameters())
        with Recorder(ft, "criterion") as crit:
            ft.train(argparse.Namespace(), model, torch.device("cpu"), loader, opt)
        roc = ft.eval(argparse.Namespace(), model, torch.device("cpu"), loader)
        model.eval()
        with torch.no_grad():
            pred_eval = model(b0)
        fx[pooling] = {"pred_step0": crit.inputs[0].float(), "loss": np.array([float(v) for v in crit.values]), "roc": np.asarray(roc, dtype=np.float64),
                       "pred_eval_batch0": pred_eval, "final_params": pack_params(list(model.named_parameters()), lambda p: p)}
    save("ref_bio_finetune_b32", fx)


# ----------------------------------------------------------------------------- bio: edge prediction, Deep Graph Infomax
def make_bio_edgepred(ref):
    """bio/util.py:16-44 NegativeEdge -> bio/batch.py:123-172 BatchAE (bio/dataloader.py:45 DataLoaderAE) ->
    bio/pretrain_edgepred.py:20-43 train() (sums divided by the step COUNT, unlike the chem script)"""
    pe = ref.pretrain_edgepred
    raw = bio_raw(4 * 16, seed=16)
    torch.manual_seed(17)  # NegativeEdge draws its candidates with torch.randint
    tf = ref.util.NegativeEdge()
    graphs = [tf(to_ref_bio(ref, g)) for g in raw]
    fx = {"raw": raw_pack(raw), "batch_size": 16, "neg": ragged([g.negative_edge_index.t().reshape(-1).tolist() for g in graphs])}
    loader = ref.dataloader.DataLoaderAE(graphs, batch_size=16, shuffle=False, num_workers=0)
    fx["batch0"] = batch_pack(next(iter(loader)))
    for gt in ("gin", "gcn"):
        torch.manual_seed(0)
        model = ref.model.GNN(5, 300, JK="last", drop_ratio=0, gnn_type=gt)
        opt = adam(model.parameters())
        with Recorder(pe, "criterion") as crit:
            ret = pe.train(argparse.Namespace(), model, torch.device("cpu"), loader, opt)
        vals = [float(v) for v in crit.values]  # two calls per step: positive pairs, negative pairs
        fx[gt] = {"returned": np.array(ret, dtype=np.float64), "loss": np.array([a + b for a, b in zip(vals[0::2], vals[1::2])]),
                  "final_params": pack_params(list(model.named_parameters()), lambda p: p)}
    save("ref_bio_edgepred_b16", fx)


def make_bio_infomax(ref):
    """torch_geometric DataLoader (bio/pretrain_deepgraphinfomax.py:4) -> its Infomax / Discriminator :27-50 -> train() :52-84"""
    dgi = ref.pretrain_deepgraphinfomax
    raw = bio_raw(4 * 16, seed=18)
    graphs = [to_ref_bio(ref, g) for g in raw]
    loader = dgi.DataLoader(graphs, batch_size=16, shuffle=False, num_workers=0)
    torch.manual_seed(0)
    gnn = ref.model.GNN(5, 300, JK="last", drop_ratio=0, gnn_type="gin")
    model = dgi.Infomax(gnn, dgi.Discriminator(300))
    disc0 = model.discriminator.weight.detach().clone()
    opt = adam(model.parameters())
    vals = []

    class RecordingLoss(torch.nn.Module):  # `loss` is a registered child module: its stand-in has to be one too
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, *a):
            r = self.inner(*a)
            vals.append(float(r))
            return r

    model.loss = RecordingLoss(model.loss)
    ret = dgi.train(argparse.Namespace(), model, torch.device("cpu"), loader, opt)
    fx = {"raw": raw_pack(raw), "batch_size": 16, "discriminator_init": disc0, "returned": np.array(ret, dtype=np.float64),
          "loss": np.array([a + b for a, b in zip(vals[0::2], vals[1::2])]),
          "final_params": pack_params(list(model.gnn.named_parameters()) + [("discriminator.weight", model.discriminator.weight)], lambda p: p)}
    save("ref_bio_infomax_b16", fx)


# ----------------------------------------------------------------------------- main
def main():
    if not refshim.available():
        sys.exit("needs the reference sources under %s" % refshim.REFERENCE_ROOT)
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)  # multi-threaded torch-CPU reductions are not run-to-run reproducible; Adam amplifies that to 1e-4
    only = set(sys.argv[1:])
    chem, bio = refshim.load("chem"), refshim.load("bio")
    jobs = [("chem_spec", make_chem_spec, chem), ("chem_masking", make_chem_masking, chem),
            ("chem_contextpred", make_chem_contextpred, chem), ("chem_finetune", make_chem_finetune, chem),
            ("chem_edgepred", make_chem_edgepred, chem), ("chem_infomax", make_chem_infomax, chem),
            ("bio_masking", make_bio_masking, bio), ("bio_contextpred", make_bio_contextpred, bio),
            ("bio_finetune", make_bio_finetune, bio), ("bio_edgepred", make_bio_edgepred, bio),
            ("bio_infomax", make_bio_infomax, bio)]
    for tag, fn, ref in jobs:
        if not only or tag in only:
            fn(ref)


if __name__ == "__main__":
    main()

"""-m gpu: model-level parity of the HIP-backed GNN / GNN_graphpred classes against the CPU oracle.

Bar (BASELINE.json north_star): node embeddings and masked-atom logits within 1e-4 (fp32) of the
reference CPU path; here |a-b| <= 1e-4 + 1e-4*|b|.  Gradients and multi-step training are checked
with the tolerances written next to each assert.
"""
import os

import pytest
import torch

from oracle import bio as obio
from oracle import chem as ochem
from oracle import pyg_semantics as pyg
from oracle import steps
from pretrain_gnns_amd.data import synthetic
from oracle import hostdata

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = dict(rtol=1e-4, atol=1e-4)


def _hip():
    from pretrain_gnns_amd.bio import model as hbio
    from pretrain_gnns_amd.chem import model as hchem
    return hchem, hbio


def _pair(ocls, hcls, *args, seed=0, **kw):
    torch.manual_seed(seed)
    ref = ocls(*args, **kw)
    hip = hcls(*args, **kw)
    hip.load_state_dict(ref.state_dict())
    return ref, hip.to(DEV)


def _grads_close(ref, hip, batch_args, weight, l2_tol=2e-2):
    """Gradient parity against the FLOAT64 oracle in relative L2 norm, robust to ReLU mask flips.

    A pre-activation within rounding distance of zero lands on different sides in two fp32
    implementations; one flipped unit perturbs the gradients of everything upstream by ~1e-3
    (dense, because weight gradients sum over all nodes).  Measured on the MI355X host with
    tests/tools/debug_layers.py: torch-CPU fp32 vs fp64 deviates by 1e-3..5e-2 (max-norm) on the very
    batch where this stack sits at 1e-6.  Hence the model-level bar is a relative L2 error of
    ``l2_tol`` per parameter (real kernel bugs are O(10%) and the op-level tests in
    test_gpu_ops.py hold each kernel to 1e-5); analytically-zero gradients (biases feeding a
    BatchNorm) are normalised by the model's largest gradient instead of their own.
    """
    import copy
    ref64 = copy.deepcopy(ref).double()
    ref64.zero_grad()
    out = ref64(*batch_args)
    (out * weight.double()).sum().backward()
    g64 = {n: p.grad for n, p in ref64.named_parameters()}
    gscale = max(float(g.abs().max()) for g in g64.values() if g is not None)
    bad = []
    for name, ph in hip.named_parameters():
        g = g64[name]
        if g is None:
            assert ph.grad is None or float(ph.grad.abs().max()) == 0.0, name
            continue
        err = (ph.grad.detach().cpu().double() - g)
        l2 = float(err.norm() / (g.norm() + 1e-3 * gscale * g.numel() ** 0.5))
        if not l2 <= l2_tol:
            bad.append((name, l2))
    assert not bad, bad


@pytest.mark.parametrize("gnn_type", ["gin", "gcn", "graphsage", "gat"])
@pytest.mark.parametrize("graphs", [1, 32])
def test_chem_gnn_forward_backward(gnn_type, graphs):
    hchem, _ = _hip()
    ref, hip = _pair(ochem.GNN, hchem.GNN, 5, 300, gnn_type=gnn_type)
    b = hostdata.chem_masking_batch(graphs, seed=graphs)
    d = b.clone().to(DEV)
    out_ref = ref(b.x, b.edge_index, b.edge_attr)
    out_hip = hip(d.x, d.edge_index, d.edge_attr)
    torch.testing.assert_close(out_hip.detach().cpu(), out_ref.detach(), **TOL)
    w = torch.randn_like(out_ref)
    (out_hip * w.to(DEV)).sum().backward()
    _grads_close(ref, hip, (b.x, b.edge_index, b.edge_attr), w)
    # data-object overload and eval mode
    ref.eval(), hip.eval()
    with torch.no_grad():
        torch.testing.assert_close(hip(d).cpu(), ref(b), **TOL)


def test_chem_masked_atom_logits_match():
    """the quantity BASELINE.json names: masked-atom logits within 1e-4."""
    hchem, _ = _hip()
    ref, hip = _pair(ochem.GNN, hchem.GNN, 5, 300)
    torch.manual_seed(5)
    head = torch.nn.Linear(300, 119)
    head_d = torch.nn.Linear(300, 119)
    head_d.load_state_dict(head.state_dict())
    head_d = head_d.to(DEV)
    b = hostdata.chem_masking_batch(256, seed=0)
    d = b.clone().to(DEV)
    lr = head(ref(b.x, b.edge_index, b.edge_attr)[b.masked_atom_indices])
    lh = head_d(hip(d.x, d.edge_index, d.edge_attr)[d.masked_atom_indices])
    torch.testing.assert_close(lh.detach().cpu(), lr.detach(), **TOL)
    assert steps.compute_accuracy(lh.cpu(), b.mask_node_label[:, 0]) == steps.compute_accuracy(lr, b.mask_node_label[:, 0])


@pytest.mark.parametrize("jk", ["concat", "max", "sum"])
def test_chem_jk_modes(jk):
    hchem, _ = _hip()
    ref, hip = _pair(ochem.GNN, hchem.GNN, 3, 64, JK=jk)
    b = hostdata.chem_plain_batch(4, seed=2)
    d = b.clone().to(DEV)
    torch.testing.assert_close(hip(d.x, d.edge_index, d.edge_attr).detach().cpu(),
                               ref(b.x, b.edge_index, b.edge_attr).detach(), **TOL)


@pytest.mark.parametrize("pool", ["mean", "sum", "max", "attention", "set2set2"])
def test_chem_graphpred(pool):
    hchem, _ = _hip()
    ref, hip = _pair(ochem.GNN_graphpred, hchem.GNN_graphpred, 5, 300, 12, graph_pooling=pool)
    b = hostdata.chem_plain_batch(16, seed=3)
    d = b.clone().to(DEV)
    out_ref = ref(b.x, b.edge_index, b.edge_attr, b.batch)
    out_hip = hip(d.x, d.edge_index, d.edge_attr, d.batch)
    torch.testing.assert_close(out_hip.detach().cpu(), out_ref.detach(), **TOL)
    out_hip.sum().backward()
    _grads_close(ref, hip, (b.x, b.edge_index, b.edge_attr, b.batch), torch.ones_like(out_ref))
    torch.testing.assert_close(hip(d).detach().cpu(), ref(b).detach(), **TOL)


@pytest.mark.parametrize("gnn_type", ["gin", "gcn", "graphsage", "gat"])
def test_bio_gnn_forward_backward(gnn_type):
    _, hbio = _hip()
    ref, hip = _pair(obio.GNN, hbio.GNN, 5, 300, gnn_type=gnn_type)
    b = hostdata.bio_masking_batch(8, seed=1)
    d = b.clone().to(DEV)
    out_ref = ref(b.x, b.edge_index, b.edge_attr)
    out_hip = hip(d.x, d.edge_index, d.edge_attr)
    torch.testing.assert_close(out_hip.detach().cpu(), out_ref.detach(), **TOL)
    w = torch.randn_like(out_ref)
    (out_hip * w.to(DEV)).sum().backward()
    _grads_close(ref, hip, (b.x.double(), b.edge_index, b.edge_attr.double()), w)


@pytest.mark.parametrize("graphs,layers,training", [(8, 5, True), (64, 3, True), (8, 2, False)])
def test_bio_one_call_network_equals_per_layer_path(graphs, layers, training, monkeypatch):
    """pgnn_bio_gin_stack_fwd/_bwd (the whole bio GIN network in one call per direction: fused ReLU between layers, side
    stream for the weight-gradient products) against the per-layer calls: outputs and BatchNorm running statistics
    bit-identical; every gradient bit-identical with PGNN_BWD_TRANSPOSED=0 and within fp32 rounding of a different
    summation order with backward-data on transposed weights (the default)"""
    import copy
    from pretrain_gnns_amd import ops
    _, hbio = _hip()
    _, a = _pair(obio.GNN, hbio.GNN, layers, 300, seed=4)
    b = copy.deepcopy(a)
    c = copy.deepcopy(a)
    for m in (a, b, c):
        m.train(training)
    d = hostdata.bio_masking_batch(graphs, seed=3).to(DEV)
    w = torch.randn(d.x.size(0), 300, device=DEV)

    def run(m, stack, transposed):
        monkeypatch.setattr(hbio, "_STACK_CALL", stack)
        monkeypatch.setenv("PGNN_BWD_TRANSPOSED", "2" if transposed else "0")
        monkeypatch.setenv("PGNN_GEMM_WP_MIN_TILES", "160")  # weight planes only where the per-layer calls run the same arithmetic
        monkeypatch.setenv("PGNN_BN_STATS_IN_GEMM", "0")     # (the one-call forward takes the mlp's BatchNorm statistics from the GEMM epilogue)
        ops.load().pgnn_reload_env()
        for _ in range(2):  # twice: running statistics advance identically
            m.zero_grad()
            out = m(d.x, d.edge_index, d.edge_attr)
            (out * w).sum().backward()
        return out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}, {k: v.clone() for k, v in m.named_buffers()}

    per_layer, exact, default = run(a, False, False), run(b, True, False), run(c, True, True)
    monkeypatch.delenv("PGNN_BWD_TRANSPOSED")
    monkeypatch.delenv("PGNN_GEMM_WP_MIN_TILES")
    monkeypatch.delenv("PGNN_BN_STATS_IN_GEMM")
    ops.load().pgnn_reload_env()
    assert torch.equal(per_layer[0], exact[0]) and torch.equal(per_layer[0], default[0])
    for k in per_layer[2]:
        assert torch.equal(per_layer[2][k], exact[2][k]) and torch.equal(per_layer[2][k], default[2][k]), k
    top = max(float(g.abs().max()) for g in per_layer[1].values())
    for k, g in per_layer[1].items():
        assert torch.equal(g, exact[1][k]), k
        assert float((g - default[1][k]).abs().max()) <= 2e-5 * float(g.abs().max()) + 1e-5 * top, k


@pytest.mark.parametrize("family,graphs", [("chem", 256), ("chem", 96), ("bio", 64)])
def test_products_on_two_fp16_planes_match_the_three_plane_products(family, graphs, monkeypatch):
    """PGNN_GEMM_2P=1: the one-call networks' forward and backward-data products on TWO fp16 planes with a power-of-two scale per row
    (three MFMA products per accumulator, csrc/linear.hip k_gemm2pw; every workgroup takes the maxima of its own rows) against the
    default three bf16 planes (six products): both are fp32-level approximations of the same product, so outputs, running
    statistics and every gradient agree to fp32 rounding carried through the layers -- two training steps, the second on top of
    the first's running statistics."""
    import copy
    from pretrain_gnns_amd import ops
    hchem, hbio = _hip()
    if family == "chem":
        _, a = _pair(ochem.GNN, hchem.GNN, 5, 300, seed=41)
        d = hostdata.chem_masking_batch(graphs, seed=42).to(DEV)
    else:
        _, a = _pair(obio.GNN, hbio.GNN, 5, 300, seed=43)
        d = hostdata.bio_masking_batch(graphs, seed=44).to(DEV)
    b = copy.deepcopy(a)
    a.train(), b.train()
    assert int(ops.load().pgnn_linear_wp_preferred(d.x.size(0), 600, 300)) == 1  # the products run on planes at this size
    w = torch.randn(d.x.size(0), 300, device=DEV)
    res = []
    for m, flag in ((a, "1"), (b, "0")):
        monkeypatch.setenv("PGNN_GEMM_2P", flag)
        ops.load().pgnn_reload_env()
        for _ in range(2):
            m.zero_grad()
            out = m(d.x, d.edge_index, d.edge_attr)
            (out * w).sum().backward()
        res.append((out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}, {k: v.clone() for k, v in m.named_buffers()}))
    monkeypatch.delenv("PGNN_GEMM_2P")
    ops.load().pgnn_reload_env()
    assert not torch.equal(res[0][0], res[1][0])  # (the knob did switch the arithmetic)
    torch.testing.assert_close(res[0][0], res[1][0], rtol=1e-4, atol=1e-4)
    assert float((res[0][0] - res[1][0]).abs().max()) <= 2e-5 * float(res[1][0].abs().max())
    for k in res[0][2]:
        if res[0][2][k].dtype.is_floating_point:
            torch.testing.assert_close(res[0][2][k], res[1][2][k], rtol=1e-5, atol=1e-6)
    # gradients: behind a forward that differs by a few 1e-6 a handful of ReLU decisions flip, and a weight-gradient entry is a sum of
    # ~n signed terms of which one flipped term is ~n^-1/2 of the total -- percent-level deviations of single entries between two
    # equally accurate forwards (measured: up to 2.3 % of a tensor's largest entry; the deviations from the REFERENCE's fixtures are
    # the same with either form: profiles/r03/parity_metrics_two_planes.jsonl against parity_metrics.jsonl).  So only a gross
    # bound here -- a wrong scale or plane would be an O(1) error: relative l2 over all parameters with a gradient to speak of
    top = max(float(g.abs().max()) for g in res[1][1].values())
    num = den = 0.0
    worst = 0.0
    for k, g in res[1][1].items():
        assert bool(torch.isfinite(res[0][1][k]).all()), k
        if float(g.abs().max()) < 1e-3 * top:
            continue  # (a bias in front of a BatchNorm: its gradient is zero up to rounding)
        d = res[0][1][k] - g
        num += float((d * d).sum())
        den += float((g * g).sum())
        worst = max(worst, float(d.abs().max()) / float(g.abs().max()))
    rel = (num / den) ** 0.5
    print("two planes vs three planes, %s %d graphs: max |d out| / max |out| %.2e, gradients: relative l2 %.2e, worst entry / largest entry %.2e"
          % (family, graphs, float((res[0][0] - res[1][0]).abs().max() / res[1][0].abs().max()), rel, worst))
    assert rel <= 0.1 and worst <= 0.25, (rel, worst)


def test_bio_batchnorm_statistics_from_the_gemm_epilogue_match_the_separate_pass(monkeypatch):
    """bio one-call network: the statistics of the mlp's BatchNorm1d(2D) taken from the 600 -> 600 product's epilogue (the default
    from ~1 500 rows) against the same call with PGNN_BN_STATS_IN_GEMM=0 (a pass over the pre-activation): outputs and running
    statistics agree to fp32 rounding carried through the layers, gradients in relative l2"""
    import copy
    from pretrain_gnns_amd import ops
    _, hbio = _hip()
    _, a = _pair(obio.GNN, hbio.GNN, 5, 300, seed=12)
    b = copy.deepcopy(a)
    a.train(), b.train()
    d = hostdata.bio_masking_batch(64, seed=13).to(DEV)
    assert int(ops.load().pgnn_linear_wp_preferred(d.x.size(0), 600, 300)) == 1
    w = torch.randn(d.x.size(0), 300, device=DEV)
    res = []
    for m, flag in ((a, "1"), (b, "0")):
        monkeypatch.setenv("PGNN_BN_STATS_IN_GEMM", flag)
        ops.load().pgnn_reload_env()
        for _ in range(2):
            m.zero_grad()
            out = m(d.x, d.edge_index, d.edge_attr)
            (out * w).sum().backward()
        res.append((out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}, {k: v.clone() for k, v in m.named_buffers()}))
    monkeypatch.delenv("PGNN_BN_STATS_IN_GEMM")
    ops.load().pgnn_reload_env()
    torch.testing.assert_close(res[0][0], res[1][0], rtol=1e-4, atol=1e-4)
    for k in res[0][2]:
        if res[0][2][k].dtype.is_floating_point:
            torch.testing.assert_close(res[0][2][k], res[1][2][k], rtol=1e-5, atol=1e-6)
        else:
            assert torch.equal(res[0][2][k], res[1][2][k]), k
    top = max(float(g.abs().max()) for g in res[1][1].values())
    for k in res[0][1]:
        g0, g1 = res[0][1][k].double(), res[1][1][k].double()
        assert float((g0 - g1).norm()) <= 1e-2 * float(g1.norm()) + 1e-4 * top, k


@pytest.mark.parametrize("pool", ["mean", "attention"])
def test_bio_graphpred(pool):
    _, hbio = _hip()
    ref, hip = _pair(obio.GNN_graphpred, hbio.GNN_graphpred, 5, 300, 40, graph_pooling=pool)
    b = hostdata.bio_masking_batch(8, seed=2)
    d = b.clone().to(DEV)
    torch.testing.assert_close(hip(d).detach().cpu(), ref(b).detach(), **TOL)


def _opt(*mods):
    return [torch.optim.Adam(m.parameters(), lr=1e-3) for m in mods]


@pytest.mark.parametrize("mask_edge", [False, True])
def test_chem_masking_train_steps(mask_edge):
    """reference train() body (chem/pretrain_masking.py:47-76) driven through both stacks."""
    hchem, _ = _hip()
    ref, hip = _pair(ochem.GNN, hchem.GNN, 5, 300)
    torch.manual_seed(9)
    heads = [torch.nn.Linear(300, 119), torch.nn.Linear(300, 4)]
    heads_d = [torch.nn.Linear(300, 119), torch.nn.Linear(300, 4)]
    for a, c in zip(heads_d, heads):
        a.load_state_dict(c.state_dict())
    heads_d = [h.to(DEV) for h in heads_d]
    opt_r, opt_h = _opt(ref, *heads), _opt(hip, *heads_d)
    for step in range(4):
        b = hostdata.chem_masking_batch(32, seed=100 + step, mask_edge=mask_edge)
        lr, ar, er = steps.chem_masking_step([ref] + heads, opt_r, b, mask_edge)
        lh, ah, eh = steps.chem_masking_step([hip] + heads_d, opt_h, b.clone().to(DEV), mask_edge)
        assert abs(lr - lh) < (1e-4 if step == 0 else 2e-2) * max(1.0, abs(lr)), (step, lr, lh)
        assert abs(ar - ah) <= 0.02 and abs(er - eh) <= 0.02
    for (n, pr), (_, ph) in zip(ref.named_parameters(), hip.named_parameters()):
        # Adam moves every coordinate by ~lr per step whatever the gradient size, so coordinates whose
        # gradient is rounding noise (e.g. biases in front of a BatchNorm) random-walk: bound = 4 steps x 2 lr
        assert float((ph.detach().cpu() - pr.detach()).abs().max()) < 1e-2, n


def test_epoch_accuracy_matches_oracle_within_a_tenth_of_a_percent():
    """SURVEY 8(d) accuracy check: the reference's own metric (compute_accuracy averaged over an epoch,
    chem/pretrain_masking.py:30-31,54-55,78) from identical init and an identical stream of BASELINE-size
    batches (256 graphs, ~1150 masked atoms each) must agree with the CPU oracle to +-0.1 % absolute."""
    from pretrain_gnns_amd import train as ptrain
    hchem, _ = _hip()
    ref, hip = _pair(ochem.GNN, hchem.GNN, 5, 300, seed=21)
    torch.manual_seed(22)
    heads = [torch.nn.Linear(300, 119), torch.nn.Linear(300, 4)]
    heads_d = [torch.nn.Linear(300, 119), torch.nn.Linear(300, 4)]
    for a, b in zip(heads, heads_d):
        b.load_state_dict(a.state_dict())
    heads_d = [h.to(DEV) for h in heads_d]
    stream = [hostdata.chem_masking_batch(256, seed=100 + i) for i in range(9)]
    ref_out = steps.chem_masking_epoch([ref] + heads, _opt(ref, *heads), stream)
    hip_out = ptrain.chem_masking_epoch([hip] + heads_d, _opt(hip, *heads_d), [b.clone() for b in stream], device=DEV)
    assert abs(ref_out[1] - hip_out[1]) <= 1e-3, (ref_out, hip_out)          # epoch accuracy
    assert abs(ref_out[0] - hip_out[0]) <= 2e-2 * abs(ref_out[0]), (ref_out, hip_out)  # epoch loss


@pytest.mark.parametrize("mask_edge", [False, True])
def test_epoch_sums_on_the_device_equal_the_per_step_read_back(mask_edge):
    """readback="epoch" (sums kept on the device, one fetch per epoch; the fused head adds its numbers inside its forward
    launch) returns bit-for-bit what readback="end" (fetch every step, add on the host as chem/pretrain_masking.py:72-76
    does) returns: the same float64 additions in the same order.  Batches of different sizes, so correct / n differs per step.
    (Bit-for-bit where the step itself is deterministic, i.e. without the bond head.)"""
    from pretrain_gnns_amd import train as ptrain
    hchem, _ = _hip()
    stream = [hostdata.chem_masking_batch(8 + 5 * i, seed=40 + i, mask_edge=mask_edge).to(DEV) for i in range(5)]
    outs = []
    for mode in ("end", "epoch"):
        torch.manual_seed(3)
        mods = [hchem.GNN(5, 300).to(DEV), torch.nn.Linear(300, 119).to(DEV), torch.nn.Linear(300, 4).to(DEV)]
        outs.append(ptrain.chem_masking_epoch(mods, _opt(*mods), stream, mask_edge=mask_edge, readback=mode))
    if mask_edge:  # torch's index backward of the bond head accumulates with atomics: two runs differ by rounding, Adam carries it on
        assert all(abs(a - c) <= 1e-3 * max(1.0, abs(a)) for a, c in zip(*outs)), outs
    else:
        assert outs[0] == outs[1], outs
    assert outs[0][0] > 0 and (outs[0][2] > 0) == mask_edge
    with pytest.raises(ValueError):
        ptrain.chem_masking_step(mods, _opt(*mods), stream[0], mask_edge, readback="epoch")  # no accumulator given


def test_contextpred_epoch_sums_on_the_device_equal_the_per_step_read_back():
    from pretrain_gnns_amd import train as ptrain
    hchem, _ = _hip()
    stream = [hostdata.chem_contextpred_batch(6 + 3 * i, seed=70 + i).to(DEV) for i in range(4)]
    outs = []
    for mode in ("end", "epoch"):
        torch.manual_seed(5)
        ms, mc = hchem.GNN(5, 300).to(DEV), hchem.GNN(3, 300).to(DEV)
        o_s, o_c = _opt(ms, mc)
        outs.append(ptrain.chem_contextpred_epoch(ms, mc, o_s, o_c, stream, pool=hchem.global_mean_pool, readback=mode))
    # (torch's index / pooling backward accumulate with atomics: two runs differ by rounding from the second step on)
    assert all(abs(a - c) <= 1e-3 * max(1.0, abs(a)) for a, c in zip(*outs)), outs
    assert outs[0][0] > 0


def test_bio_epoch_sums_on_the_device_equal_the_per_step_read_back():
    from pretrain_gnns_amd import train as ptrain
    _, hbio = _hip()
    stream = [hostdata.bio_masking_batch(4 + i, seed=60 + i).to(DEV) for i in range(4)]
    outs = []
    for mode in ("end", "epoch"):
        torch.manual_seed(4)
        mods = [hbio.GNN(5, 300).to(DEV), torch.nn.Linear(300, 7).to(DEV)]
        outs.append(ptrain.bio_masking_epoch(mods, _opt(*mods), stream, readback=mode))
    assert outs[0] == outs[1], outs


@pytest.mark.parametrize("readback", ["end", "epoch"])
def test_hip_graph_replay_of_the_masking_step_equals_eager_steps(readback):
    """train.GraphedChemMaskingStep (the step captured once into a HIP graph: the library's launches, its side-stream fork / join,
    the device-side Adam step count) replays to the numbers of eager steps from the same state: 3 warm-up steps + 2 replays
    against 5 eager steps, per-step read-back and sums kept on the device"""
    import copy
    from pretrain_gnns_amd import optim, train as ptrain
    hchem, _ = _hip()
    torch.manual_seed(11)
    mods_a = [hchem.GNN(5, 300).to(DEV), torch.nn.Linear(300, 119).to(DEV), torch.nn.Linear(300, 4).to(DEV)]
    mods_b = copy.deepcopy(mods_a)
    b = hostdata.chem_masking_batch(24, seed=12).to(DEV)
    opts_a = optim.Adam.shared([m.parameters() for m in mods_a], lr=1e-3)
    eager = [ptrain.chem_masking_step(mods_a, opts_a, b) for _ in range(5)]
    opts_b = optim.Adam.shared([m.parameters() for m in mods_b], lr=1e-3)
    g = ptrain.GraphedChemMaskingStep(mods_b, opts_b, b, warmup=3, readback=readback)
    if readback == "end":
        got = [g(), g()]
        for want, have in zip(eager[3:], got):
            assert abs(want[0] - have[0]) <= 1e-6 * abs(want[0]) and abs(want[1] - have[1]) <= 1e-12, (eager, got)
    else:
        assert g() is None and g() is None
        sums = g.sums()
        assert sums[3] == 2.0
        assert abs(sums[0] - (eager[3][0] + eager[4][0])) <= 1e-6 * abs(sums[0]), (eager, sums)
        assert abs(sums[1] - (eager[3][1] + eager[4][1])) <= 1e-12
        assert g.sums() == [0.0, 0.0, 0.0, 0.0]
    for pa, pb in zip(mods_a[0].parameters(), mods_b[0].parameters()):
        torch.testing.assert_close(pa, pb, rtol=1e-5, atol=1e-6)
    assert int(opts_b[0].step_count) == 5


def test_hip_graph_replay_at_the_benchmarked_batch_takes_the_same_path_and_is_deterministic():
    """The capture test above runs 24 graphs (~550 rows): below the two-plane products' 48-tile threshold and below
    PGNN_DW_PAIR_MIN_ROWS, so the captured step never met the fork that is a product's own dispatch (hipExtLaunchKernelGGL's stop
    event), the paired weight-gradient launch with the bond columns or the deferred bond-table launch -- and a lost fork edge under
    capture went through the suite (replayed loss 0.99 against 0.37, different from run to run).  256 graphs, the
    benchmarked batch: 3 warm-up + 3 replays against 6 eager steps, parameters equal; and two independent captures from the same
    state replay to the same bits."""
    import copy
    from pretrain_gnns_amd import optim, train as ptrain
    hchem, _ = _hip()
    torch.manual_seed(21)
    mods_a = [hchem.GNN(5, 300).to(DEV), torch.nn.Linear(300, 119).to(DEV), torch.nn.Linear(300, 4).to(DEV)]
    mods_b, mods_c = copy.deepcopy(mods_a), copy.deepcopy(mods_a)
    b = hostdata.chem_masking_batch(256, seed=22).to(DEV)
    opts_a = optim.Adam.shared([m.parameters() for m in mods_a], lr=1e-3)
    eager = [ptrain.chem_masking_step(mods_a, opts_a, b) for _ in range(6)]
    replayed = []
    for mods in (mods_b, mods_c):
        opts = optim.Adam.shared([m.parameters() for m in mods], lr=1e-3)
        g = ptrain.GraphedChemMaskingStep(mods, opts, b, warmup=3, readback="end")
        replayed.append([g(), g(), g()])
        assert int(opts[0].step_count) == 6
        torch.cuda.synchronize()
    for want, have in zip(eager[3:], replayed[0]):
        assert abs(want[0] - have[0]) <= 2e-5 * abs(want[0]) + 1e-7 and abs(want[1] - have[1]) <= 1e-12, (eager, replayed)
    assert replayed[0] == replayed[1], replayed  # bit-stable from capture to capture
    for pa, pb, pc in zip(mods_a[0].parameters(), mods_b[0].parameters(), mods_c[0].parameters()):
        assert torch.equal(pb, pc)
        torch.testing.assert_close(pa, pb, rtol=2e-5, atol=2e-6)


def test_product_train_step_mirrors_oracle_step():
    """pretrain_gnns_amd.train (what bench.py times) == oracle.steps on the same HIP model, for both
    readback placements, including with torch's fused Adam."""
    from pretrain_gnns_amd import train as ptrain
    hchem, _ = _hip()
    b = hostdata.chem_masking_batch(16, seed=5, mask_edge=True).to(DEV)
    results = []
    for mode in ("oracle", "inline", "end"):
        torch.manual_seed(0)
        mods = [hchem.GNN(5, 300).to(DEV), torch.nn.Linear(300, 119).to(DEV), torch.nn.Linear(300, 4).to(DEV)]
        opts = [torch.optim.Adam(m.parameters(), lr=1e-3, fused=(mode == "end")) for m in mods]
        out = []
        for _ in range(3):
            if mode == "oracle":
                out.append(steps.chem_masking_step(mods, opts, b, True))
            else:
                out.append(ptrain.chem_masking_step(mods, opts, b, True, readback=mode))
        results.append(out)
    for a, c in zip(results[0], results[1]):
        assert a == c  # identical code path, identical kernels: bitwise equal
    for a, c in zip(results[0], results[2]):
        assert abs(a[0] - c[0]) < 1e-3 * abs(a[0]) and abs(a[1] - c[1]) < 0.02 and abs(a[2] - c[2]) < 0.02


def test_chem_contextpred_train_steps():
    hchem, _ = _hip()
    ref_s, hip_s = _pair(ochem.GNN, hchem.GNN, 5, 300, seed=1)
    ref_c, hip_c = _pair(ochem.GNN, hchem.GNN, 3, 300, seed=2)
    o_rs, o_rc = _opt(ref_s, ref_c)
    o_hs, o_hc = _opt(hip_s, hip_c)
    for step in range(3):
        b = hostdata.chem_contextpred_batch(32, seed=50 + step)
        lr, ar = steps.chem_contextpred_step(ref_s, ref_c, o_rs, o_rc, b)
        lh, ah = steps.chem_contextpred_step(hip_s, hip_c, o_hs, o_hc, b.clone().to(DEV), pool=hchem.global_mean_pool)
        assert abs(lr - lh) < (1e-4 if step == 0 else 3e-2) * max(1.0, abs(lr)), (step, lr, lh)
        assert abs(ar - ah) <= 0.05


def test_chem_edgepred_and_infomax_train_steps():
    """the two remaining pre-training objectives of the reference (chem/pretrain_edgepred.py:32-46,
    chem/pretrain_deepgraphinfomax.py:61-84) driven through both stacks"""
    from pretrain_gnns_amd import train as ptrain
    hchem, _ = _hip()
    ref, hip = _pair(ochem.GNN, hchem.GNN, 5, 300, seed=31)
    o_r, o_h = _opt(ref)[0], _opt(hip)[0]
    for step in range(3):
        b = hostdata.chem_edgepred_batch(32, seed=90 + step)
        assert b.negative_edge_index.size(1) > 0 and int(b.negative_edge_index.max()) < b.x.size(0)
        lr, ar = steps.chem_edgepred_step(ref, o_r, b)
        lh, ah = ptrain.chem_edgepred_step(hip, o_h, b.clone().to(DEV))
        assert abs(lr - lh) < (1e-4 if step == 0 else 3e-2) * max(1.0, abs(lr)), (step, lr, lh)
        assert abs(ar - ah) <= 0.02
    ref, hip = _pair(ochem.GNN, hchem.GNN, 5, 300, seed=32)
    torch.manual_seed(33)
    d_ref = steps.Discriminator(300)
    d_hip = ptrain.Discriminator(300)
    d_hip.load_state_dict(d_ref.state_dict())
    model = ptrain.Infomax(hip, d_hip.to(DEV))
    o_r = torch.optim.Adam(list(ref.parameters()) + list(d_ref.parameters()), lr=1e-3)
    o_h = torch.optim.Adam(model.parameters(), lr=1e-3)
    for step in range(3):
        b = hostdata.chem_plain_batch(32, seed=95 + step)
        lr, ar = steps.chem_infomax_step(ref, d_ref, o_r, b)
        lh, ah = ptrain.chem_infomax_step(model, o_h, b.clone().to(DEV))
        assert abs(lr - lh) < (1e-4 if step == 0 else 3e-2) * max(1.0, abs(lr)), (step, lr, lh)
        assert abs(ar - ah) <= 0.02


def test_bio_masking_train_steps():
    _, hbio = _hip()
    ref, hip = _pair(obio.GNN, hbio.GNN, 5, 300)
    torch.manual_seed(3)
    head = torch.nn.Linear(300, 7)
    head_d = torch.nn.Linear(300, 7)
    head_d.load_state_dict(head.state_dict())
    head_d = head_d.to(DEV)
    opt_r, opt_h = _opt(ref, head), _opt(hip, head_d)
    for step in range(3):
        b = hostdata.bio_masking_batch(8, seed=70 + step)
        lr, ar = steps.bio_masking_step([ref, head], opt_r, b)
        lh, ah = steps.bio_masking_step([hip, head_d], opt_h, b.clone().to(DEV))
        assert abs(lr - lh) < (1e-4 if step == 0 else 3e-2) * max(1.0, abs(lr)), (step, lr, lh)
        assert abs(ar - ah) <= 0.02


@pytest.mark.parametrize("gnn_type", ["gin", "gcn"])
def test_chem_finetune_steps_and_eval(gnn_type):
    """chem/finetune.py train() body and eval(): GNN_graphpred (mean pooling) + masked multi-task BCE in
    float64, then eval-mode scores -> ROC-AUC.  drop_ratio = 0 so both sides are deterministic."""
    from pretrain_gnns_amd import train
    hchem, _ = _hip()
    ref, hip = _pair(ochem.GNN_graphpred, hchem.GNN_graphpred, 5, 300, 12, gnn_type=gnn_type, seed=11)
    b = hostdata.chem_finetune_batch(48, num_tasks=12, seed=3)
    bd = hostdata.chem_finetune_batch(48, num_tasks=12, seed=3).to(DEV)
    o_ref, o_hip = torch.optim.Adam(ref.parameters(), lr=1e-3), torch.optim.Adam(hip.parameters(), lr=1e-3)
    for step in range(3):
        l_ref, l_hip = steps.chem_finetune_step(ref, o_ref, b), train.chem_finetune_step(hip, o_hip, bd)
        assert abs(l_ref - l_hip) <= (1e-4 if step == 0 else 3e-2) * max(1.0, abs(l_ref)), (step, l_ref, l_hip)
    hip.load_state_dict(ref.state_dict())  # same weights again: eval parity is then a pure forward check
    val = hostdata.chem_finetune_batch(64, num_tasks=12, seed=4)
    auc_ref = steps.chem_eval(ref, [b, val])
    auc_hip = train.chem_eval(hip, [bd, hostdata.chem_finetune_batch(64, num_tasks=12, seed=4).to(DEV)])
    assert abs(auc_ref - auc_hip) <= 1e-3, (auc_ref, auc_hip)  # "within +-0.1 % absolute" 


@pytest.mark.parametrize("stack", [True, False])
def test_dropout_is_reproducible_under_manual_seed(stack, monkeypatch):
    """fused dropout draws its seeds from torch's CPU generator: torch.manual_seed pins the whole
    forward/backward, on the one-call path and on the per-layer path"""
    hchem, _ = _hip()
    monkeypatch.setattr(hchem, "_STACK_CALL", stack)
    torch.manual_seed(0)
    m = hchem.GNN(3, 300, drop_ratio=0.5).to(DEV)
    d = hostdata.chem_plain_batch(24, seed=2).to(DEV)
    runs = []
    for seed in (7, 7, 8):
        torch.manual_seed(seed)
        m.zero_grad()
        out = m(d.x, d.edge_index, d.edge_attr)
        out.square().sum().backward()
        runs.append((out.detach().clone(), torch.cat([p.grad.flatten() for p in m.parameters()]).clone()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    assert not torch.equal(runs[0][0], runs[2][0])
    zero_frac = float((runs[0][0] == 0).float().mean())
    assert 0.45 < zero_frac < 0.55  # last layer: no ReLU, so exactly the dropped half is zero


def test_chem_finetune_with_dropout_runs_and_regularises():
    """drop_ratio = 0.5 (chem/finetune.py default): the training forward differs run to run and from
    the eval forward, eval is deterministic, the loss goes down over a few steps"""
    from pretrain_gnns_amd import train
    hchem, _ = _hip()
    torch.manual_seed(0)
    hip = hchem.GNN_graphpred(5, 300, 12, drop_ratio=0.5).to(DEV)
    bd = hostdata.chem_finetune_batch(64, num_tasks=12, seed=5).to(DEV)
    hip.train()
    a = hip(bd.x, bd.edge_index, bd.edge_attr, bd.batch).detach()
    b = hip(bd.x, bd.edge_index, bd.edge_attr, bd.batch).detach()
    assert not torch.equal(a, b)
    hip.eval()
    with torch.no_grad():
        c = hip(bd).clone()
        assert torch.equal(c, hip(bd))
    hip.train()
    opt = torch.optim.Adam(hip.parameters(), lr=1e-3)
    losses = [train.chem_finetune_step(hip, opt, bd) for _ in range(12)]
    assert all(l == l for l in losses) and sum(losses[-3:]) < sum(losses[:3])


@pytest.mark.parametrize("name", ["chem_gcn_contextpred", "bio_gcn_masking", "chem_graphsage_contextpred",
                                  "bio_graphsage_masking", "chem_gat_contextpred", "bio_gat_masking"])
def test_golden_checkpoint_parity(name):
    """real shipped GCN / GraphSAGE weights + BN running stats: strict load into the HIP classes, eval- and
    train-mode embeddings and one gradient must match the fixture (oracle on the reference blob)."""
    hchem, hbio = _hip()
    fx = torch.load(os.path.join(GOLDEN, name + ".pt"), map_location="cpu")
    cls = hchem.GNN if fx["kind"] == "chem" else hbio.GNN
    m = cls(5, 300, gnn_type=name.split("_")[1])
    res = m.load_state_dict(fx["state_dict"], strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    m = m.to(DEV)
    bt = {k: v.to(DEV) for k, v in fx["batch"].items()}
    m.eval()
    with torch.no_grad():
        out = m(bt["x"], bt["edge_index"], bt["edge_attr"])
    scale = float(fx["out_eval"].abs().max())
    assert float((out.cpu() - fx["out_eval"]).abs().max()) <= 1e-4 * max(1.0, scale)
    m.train()
    out = m(bt["x"], bt["edge_index"], bt["edge_attr"])
    assert float((out.detach().cpu() - fx["out_train"]).abs().max()) <= 1e-4 * max(1.0, float(fx["out_train"].abs().max()))
    out.square().mean().backward()
    g = dict(m.named_parameters())[fx["grad_name"]].grad.cpu()
    assert float((g - fx["grad"]).abs().max()) <= 2e-3 * float(fx["grad"].abs().max()) + 1e-7


@pytest.mark.parametrize("graphs", [64, 256, 2048])
def test_forward_is_bitwise_deterministic(graphs):
    """run-to-run bit equality of outputs and gradients: 64 graphs, the reference's 256 (side stream active in
    the backward) and BASELINE's full 2048 (54k nodes: single-stream regime, two-level reductions)"""
    hchem, _ = _hip()
    _, hip = _pair(ochem.GNN, hchem.GNN, 5, 300)
    d = hostdata.chem_masking_batch(graphs, seed=4).to(DEV)
    outs, grads = [], []
    for _ in range(3):
        hip.zero_grad()
        o = hip(d.x, d.edge_index, d.edge_attr)
        o.square().sum().backward()
        outs.append(o.detach().clone())
        grads.append(torch.cat([p.grad.flatten() for p in hip.parameters()]).clone())
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    assert all(torch.equal(grads[0], g) for g in grads[1:])


@pytest.mark.parametrize("graphs,layers,training", [(48, 5, True), (3, 2, True), (48, 3, False), (1500, 5, True)])
def test_one_call_network_equals_per_layer_path(graphs, layers, training, monkeypatch):
    """pgnn_chem_gin_stack_fwd/_bwd (whole network, one call per direction; side-stream overlap across
    layers) must be BIT-identical to the per-layer calls: outputs, every gradient, BN running stats.
    (With PGNN_BWD_TRANSPOSED=0: by default the one-call backward runs backward-data on transposed weights through the
    split-bf16 kernel, which the per-layer path does not -- that pairing is held to a tolerance in the next test.)"""
    import copy
    from pretrain_gnns_amd import ops
    hchem, _ = _hip()
    monkeypatch.setenv("PGNN_BWD_TRANSPOSED", "0")
    monkeypatch.setenv("PGNN_BN_STATS_IN_GEMM", "0")  # (the one-call forward takes the BatchNorm statistics from the GEMM epilogue)
    # the one-call network's products run on pre-split weight planes, bit-identical to the per-layer calls wherever THOSE take the
    # split-bf16 kernel (from 160 tiles; the 1500-graph case); below, the per-layer calls run the fp32-MFMA kernel, so the planes
    # are held off there -- test_one_call_network_on_weight_planes_below_the_split_threshold holds that pairing to fp32 rounding
    monkeypatch.setenv("PGNN_GEMM_WP_MIN_TILES", "160")
    # ... on THREE bf16 planes: the default since round 4, two fp16 planes + row scales, is a different (equally accurate) arithmetic
    # -- test_one_call_network_on_two_planes_against_the_per_layer_path holds that pairing to fp32 rounding
    monkeypatch.setenv("PGNN_GEMM_2P", "0")
    # ... and with the BatchNorm-backward column sums taken by a pass of their own, as the per-layer path takes them: the one-call
    # backward's default (round 4) takes them in the transposed aggregation's epilogue, in another (fixed) order of the rows --
    # test_batchnorm_backward_sums_from_the_transposed_aggregation holds that pairing to fp32 rounding
    monkeypatch.setenv("PGNN_BN_BWD_IN_AGG", "0")
    monkeypatch.setenv("PGNN_DW_PAIR", "0")  # ... and the two weight gradients of a layer as two launches (one launch: half the splits over the rows)
    ops.load().pgnn_reload_env()
    _, a = _pair(ochem.GNN, hchem.GNN, layers, 300, seed=5)
    b = copy.deepcopy(a)
    a.train(training), b.train(training)
    d = hostdata.chem_masking_batch(graphs, seed=6).to(DEV)
    w = torch.randn(d.x.size(0), 300, device=DEV)
    res = []
    for m, flag in ((a, True), (b, False)):
        monkeypatch.setattr(hchem, "_STACK_CALL", flag)
        for _ in range(2):  # twice: running statistics and counters advance identically
            m.zero_grad()
            out = m(d.x, d.edge_index, d.edge_attr)
            (out * w).sum().backward()
        res.append((out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()},
                    {k: v.clone() for k, v in m.named_buffers()}))
    assert torch.equal(res[0][0], res[1][0])
    for k in res[0][1]:
        if k.startswith("x_embedding"):
            # the one-call path groups the atoms once by (type, chirality) and folds; the per-layer path sums each
            # column on its own: same values, different association order
            scale = float(res[1][1][k].abs().max())
            torch.testing.assert_close(res[0][1][k], res[1][1][k], rtol=1e-4, atol=1e-4 * scale)
        else:
            assert torch.equal(res[0][1][k], res[1][1][k]), k
    for k in res[0][2]:
        assert torch.equal(res[0][2][k], res[1][2][k]), k
    monkeypatch.delenv("PGNN_BWD_TRANSPOSED")
    monkeypatch.delenv("PGNN_BN_STATS_IN_GEMM")
    monkeypatch.delenv("PGNN_GEMM_WP_MIN_TILES")
    ops.load().pgnn_reload_env()


@pytest.mark.parametrize("graphs,two_planes", [(64, "0"), (90, "0"), (64, "1"), (90, "1"), (256, "1"), (1500, "1")])
def test_one_call_network_on_weight_planes_below_the_split_threshold(graphs, two_planes, monkeypatch):
    """between 48 and 160 tiles (~1 500 .. 2 600 rows) the one-call network multiplies on weight planes (split-bf16 arithmetic)
    while the per-layer calls still take the fp32-MFMA kernel: equal to fp32 rounding carried through five BatchNorm'ed layers,
    not bit for bit.  two_planes = "1" (the default since round 4, at every size where the network runs on planes): the one-call
    network's products on two fp16 planes + row scales against the per-layer calls' fp32-MFMA / split-bf16 kernels, same bar."""
    import copy
    from pretrain_gnns_amd import ops
    hchem, _ = _hip()
    monkeypatch.setenv("PGNN_GEMM_2P", two_planes)
    ops.load().pgnn_reload_env()
    _, a = _pair(ochem.GNN, hchem.GNN, 5, 300, seed=5)
    b = copy.deepcopy(a)
    a.train(), b.train()
    d = hostdata.chem_masking_batch(graphs, seed=6).to(DEV)
    assert int(ops.load().pgnn_linear_wp_preferred(d.x.size(0), 600, 300)) == 1
    w = torch.randn(d.x.size(0), 300, device=DEV)
    res = []
    for m, flag in ((a, True), (b, False)):
        monkeypatch.setattr(hchem, "_STACK_CALL", flag)
        m.zero_grad()
        out = m(d.x, d.edge_index, d.edge_attr)
        (out * w).sum().backward()
        res.append((out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    torch.testing.assert_close(res[0][0], res[1][0], rtol=1e-4, atol=1e-4)  # (measured 5e-5: two fp32-accurate products, five layers)
    top = max(float(g.abs().max()) for g in res[1][1].values())
    for k in res[0][1]:  # relative L2 per tensor: a ReLU input within rounding of zero may land on the other side
        d = float((res[0][1][k] - res[1][1][k]).double().norm())
        assert d <= 2e-2 * float(res[1][1][k].double().norm()) + 1e-4 * top, k


@pytest.mark.parametrize("graphs,training,pol", [(256, True, None), (64, True, None), (3, True, None), (200, False, None), (1400, True, None),
                                                 (256, True, "3"), (1400, True, "3")])
def test_batchnorm_backward_sums_from_the_transposed_aggregation(graphs, training, pol, monkeypatch):
    """one-call chem GIN backward (chem/model.py:269-275 under autograd): below the top layer the column sums of the BatchNorm
    backward -- sum of dyr and of dyr * xhat over the rows -- come out of the transposed aggregation that writes dy
    (k_aggregate_dma's TAIL, csrc/aggregate.hip), folded in the same launch; PGNN_BN_BWD_IN_AGG=0 takes them by the pass of their
    own (k_bn_bwd_partial).  Same forward bit for bit; every gradient -- dgamma / dbeta are those sums themselves -- equal to fp32
    rounding of sums taken in another order; and the fused form is deterministic (three runs, bit-equal).  pol "3": the kernel
    instance of the large batches (non-temporal row loads and stores; what a batch beyond 128 MB of rows runs) forced onto these."""
    import copy
    from pretrain_gnns_amd import ops
    if pol is not None:
        monkeypatch.setenv("PGNN_DMA_POL", pol)
    hchem, _ = _hip()
    _, a = _pair(ochem.GNN, hchem.GNN, 5, 300, seed=15)
    b = copy.deepcopy(a)
    a.train(training), b.train(training)
    d = hostdata.chem_masking_batch(graphs, seed=16).to(DEV)
    w = torch.randn(d.x.size(0), 300, device=DEV)
    res = []
    for m, flag in ((a, "1"), (b, "0"), (a, "1"), (a, "1")):
        monkeypatch.setenv("PGNN_BN_BWD_IN_AGG", flag)
        ops.load().pgnn_reload_env()
        m.zero_grad()
        out = m(d.x, d.edge_index, d.edge_attr)
        (out * w).sum().backward()
        res.append((out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}))
        if training and len(res) == 2:
            a.load_state_dict(b.state_dict())  # (same running statistics before the repeats; they do not enter a training-mode pass anyway)
    if not training:
        assert torch.equal(res[0][0], res[1][0])
    else:
        assert torch.equal(res[0][0], res[1][0])  # (the first pass of two identical copies)
    top = max(float(g.abs().max()) for g in res[1][1].values())
    for k, g in res[1][1].items():
        assert bool(torch.isfinite(res[0][1][k]).all()), k
        if float(g.abs().max()) < 1e-3 * top:
            continue  # (a bias in front of a BatchNorm: its gradient is zero up to rounding -- pure summation-order noise)
        dnorm = float((res[0][1][k] - g).double().norm())
        assert dnorm <= 1e-4 * float(g.double().norm()) + 1e-6 * top, (k, dnorm, float(g.double().norm()))
    for k in res[0][1]:
        assert torch.equal(res[2][1][k], res[3][1][k]), k  # deterministic


@pytest.mark.parametrize("graphs,layers", [(256, 5), (96, 3), (700, 2)])
def test_bond_table_gradients_out_of_the_weight_gradient_product(graphs, layers, monkeypatch):
    """one-call chem GIN backward (chem/model.py:37-52 under autograd): the gradients of edge_embedding1 / 2,
    demb = cfeat^T dagg with dagg = dhid W1, are taken as (cfeat^T dhid) W1 -- cfeat^T dhid rides as twelve columns in the tile padding
    of the dW1 product that runs anyway (linear.hip, linear_bwd_weight_pair_ext), the fold of its split-K partials folds it, one small
    launch per backward does G^T W1 for every layer (float64 accumulators).  PGNN_BOND_IN_DW=0: the pass over dagg per layer
    (k_rowfeat_bwd_partial / _final).  Every OTHER gradient bit-identical (the extra columns touch no other accumulator); the bond
    tables equal to fp32 rounding of a re-associated product; deterministic."""
    import copy
    from pretrain_gnns_amd import ops
    hchem, _ = _hip()
    _, a = _pair(ochem.GNN, hchem.GNN, layers, 300, seed=31)
    a.train()
    d = hostdata.chem_masking_batch(graphs, seed=32).to(DEV)
    w = torch.randn(d.x.size(0), 300, device=DEV)
    res = []
    for flag in ("1", "0", "1"):
        monkeypatch.setenv("PGNN_BOND_IN_DW", flag)
        ops.load().pgnn_reload_env()
        m = copy.deepcopy(a)
        m.zero_grad()
        out = m(d.x, d.edge_index, d.edge_attr)
        (out * w).sum().backward()
        torch.cuda.synchronize()
        res.append(dict({k: p.grad.clone() for k, p in m.named_parameters()}, __out=out.detach().clone()))
    monkeypatch.delenv("PGNN_BOND_IN_DW")
    ops.load().pgnn_reload_env()
    bonds = [k for k in res[0] if "edge_embedding" in k]
    assert len(bonds) == 2 * layers
    for k in res[0]:
        assert torch.equal(res[0][k], res[2][k]), k  # deterministic
        if k in bonds:
            # rows of the tables no edge of the batch uses have a zero gradient either way
            ref = res[1][k].double()
            err = float((res[0][k].double() - ref).abs().max())
            assert err <= 2e-5 * float(ref.abs().max()) + 1e-30, (k, err, float(ref.abs().max()))
            assert torch.equal(res[0][k] == 0, res[1][k] == 0) or err <= 1e-6 * float(ref.abs().max()), k
        else:
            assert torch.equal(res[0][k], res[1][k]), k


@pytest.mark.parametrize("graphs", [64, 256])
def test_side_stream_schedules_of_the_backward_give_the_same_bits(graphs, monkeypatch):
    """one-call chem GIN backward: with the weight gradients and bond tables on the side stream (default) or everything on the
    caller's stream (PGNN_SIDE_STREAM=0) -- schedules of the SAME kernels on the same buffers: every gradient bit-identical, run
    after run"""
    import copy
    from pretrain_gnns_amd import ops
    hchem, _ = _hip()
    _, a = _pair(ochem.GNN, hchem.GNN, 5, 300, seed=21)
    d = hostdata.chem_masking_batch(graphs, seed=22).to(DEV)
    w = torch.randn(d.x.size(0), 300, device=DEV)
    res = []
    # ... and PGNN_PRODUCER_AMAX=0: the row maxima of agg / dz taken by the two-plane products themselves instead of by the
    # aggregation / the BatchNorm backward that write those rows (round 4): the same maxima, hence the same scales and bits
    for env in ({}, {"PGNN_SIDE_STREAM": "0"}, {"PGNN_PRODUCER_AMAX": "0"}, {}, {"PGNN_SIDE_STREAM": "0", "PGNN_PRODUCER_AMAX": "0"}):
        for k in ("PGNN_SIDE_STREAM", "PGNN_PRODUCER_AMAX"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ops.load().pgnn_reload_env()
        m = copy.deepcopy(a).train()
        for _ in range(2):
            m.zero_grad()
            out = m(d.x, d.edge_index, d.edge_attr)
            (out * w).sum().backward()
        torch.cuda.synchronize()
        res.append(dict({k: p.grad.clone() for k, p in m.named_parameters()}, __out=out.detach().clone()))
    for other in res[1:]:
        for k in res[0]:
            assert torch.equal(res[0][k], other[k]), k


@pytest.mark.parametrize("graphs", [4, 256])
def test_num_batches_tracked_is_counted_inside_the_stack_call(graphs):
    """nn.BatchNorm1d.forward adds one to num_batches_tracked per training-mode forward; the one-call GIN networks do it inside
    a launch they make anyway (the weight-plane split at 256 graphs, a one-block launch at 4) instead of a torch launch:
    +1 per training forward for every layer, nothing in eval mode, chem and bio"""
    hchem, hbio = _hip()
    torch.manual_seed(0)
    m = hchem.GNN(5, 300, gnn_type="gin").to(DEV)
    d = hostdata.chem_masking_batch(graphs, seed=2).to(DEV)
    mb = hbio.GNN(3, 300, gnn_type="gin").to(DEV)
    db = hostdata.bio_masking_batch(max(2, graphs // 8), seed=2).to(DEV)
    for net, args, bns in ((m, (d.x, d.edge_index, d.edge_attr), list(m.batch_norms)),
                           (mb, (db.x, db.edge_index, db.edge_attr), [c.mlp[1] for c in mb.gnns])):
        net.train()
        for k in range(1, 4):
            net(*args)
            assert [int(bn.num_batches_tracked) for bn in bns] == [k] * len(bns)
        net.eval()
        with torch.no_grad():
            net(*args)
        assert [int(bn.num_batches_tracked) for bn in bns] == [3] * len(bns)


@pytest.mark.parametrize("graphs,layers", [(48, 5), (256, 5), (3, 2)])
def test_batchnorm_statistics_from_the_gemm_epilogue_match_the_separate_pass(graphs, layers, monkeypatch):
    """one-call network with the training-mode BatchNorm statistics taken from the second product's epilogue (per-16-row
    blocks merged in float64; the default) against the same call with PGNN_BN_STATS_IN_GEMM=0 (shifted sums over z in a pass of
    its own): outputs and running statistics agree to fp32 rounding carried through the layers, gradients up to ReLU flips"""
    import copy
    from pretrain_gnns_amd import ops
    hchem, _ = _hip()
    _, a = _pair(ochem.GNN, hchem.GNN, layers, 300, seed=8)
    b = copy.deepcopy(a)
    d = hostdata.chem_masking_batch(graphs, seed=9).to(DEV)
    w = torch.randn(d.x.size(0), 300, device=DEV)
    res = []
    for m, flag in ((a, "1"), (b, "0")):
        monkeypatch.setenv("PGNN_BN_STATS_IN_GEMM", flag)
        ops.load().pgnn_reload_env()
        for _ in range(2):
            m.zero_grad()
            out = m(d.x, d.edge_index, d.edge_attr)
            (out * w).sum().backward()
        res.append((out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()},
                    {k: v.clone() for k, v in m.named_buffers()}))
    monkeypatch.delenv("PGNN_BN_STATS_IN_GEMM")
    ops.load().pgnn_reload_env()
    torch.testing.assert_close(res[0][0], res[1][0], rtol=1e-4, atol=1e-4)
    top = max(float(g.abs().max()) for g in res[1][1].values())
    for k in res[0][1]:
        # Gradients: in the l2 norm, 1e-2.  The two paths differ by rounding in the statistics, which is enough to flip the few ReLU
        # units (of 20 million at 256 graphs) whose pre-activation sits within rounding of zero; each flip is a step in the
        # gradients below it.  tools/bn_stats_ab.py: 3e-3 between the two paths in the bottom layers, 7e-7 in the top layer (no
        # flip above it), and EITHER path is 1e-3 .. 3e-3 from the float64 oracle for the same reason, with equal forward error.
        g0, g1 = res[0][1][k].double(), res[1][1][k].double()
        assert float((g0 - g1).norm()) <= 1e-2 * float(g1.norm()) + 1e-5 * top * g1.numel() ** 0.5, k
    for k in res[0][2]:
        if res[0][2][k].is_floating_point():
            torch.testing.assert_close(res[0][2][k], res[1][2][k], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("graphs", [64, 256, 700, 1200])
def test_batchnorm_statistics_folded_inside_the_product_launch(graphs, monkeypatch):
    """one-call chem network, training mode: the per-16-row-block statistics of z = the second product's result are merged inside
    that product's launch (tile through LDS, groups of 16 row tiles by the last tile to arrive, the column panel by the last group:
    bn_fold.h; round 4, the default on two planes) against PGNN_BN_STATS_FOLD=0, the launch of its own that merges the same blocks
    (k_bn_stats_final_blocks).  The same blocks merged in another association order, in float64: save_mean / invstd and the
    running statistics to a few fp32 ulps, the forward output to fp32 rounding carried through five layers; deterministic."""
    import copy
    from pretrain_gnns_amd import ops
    hchem, _ = _hip()
    _, a = _pair(ochem.GNN, hchem.GNN, 5, 300, seed=18)
    b = copy.deepcopy(a)
    d = hostdata.chem_masking_batch(graphs, seed=19).to(DEV)
    assert int(ops.load().pgnn_linear_wp_preferred(d.x.size(0), 600, 300)) == 1
    res = []
    for m, flag in ((a, "1"), (b, "0"), (copy.deepcopy(b), "1")):
        monkeypatch.setenv("PGNN_BN_STATS_FOLD", flag)
        ops.load().pgnn_reload_env()
        if len(res) == 2:
            m.load_state_dict(res[0][2])  # the third run repeats the first from the first's initial state
        state0 = copy.deepcopy(m.state_dict())
        m.train()
        with torch.no_grad():
            out = m(d.x, d.edge_index, d.edge_attr)
        res.append((out.clone(), {k: v.clone() for k, v in m.named_buffers()}, state0))
    torch.testing.assert_close(res[0][0], res[1][0], rtol=2e-5, atol=2e-5)
    for k in res[0][1]:
        if res[0][1][k].is_floating_point():
            torch.testing.assert_close(res[0][1][k], res[1][1][k], rtol=2e-6, atol=1e-7)
        else:
            assert torch.equal(res[0][1][k], res[1][1][k]), k  # num_batches_tracked
    assert torch.equal(res[0][0], res[2][0])
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[2][1][k]), k


@pytest.mark.parametrize("graphs,layers", [(48, 5), (1500, 5), (3, 2)])
def test_transposed_backward_data_matches_the_fp32_mfma_backward(graphs, layers, monkeypatch):
    """one-call backward with backward-data on pre-transposed weights (forward split-bf16 kernel, pgnn_linear_bwd_data_t)
    against the same call with PGNN_BWD_TRANSPOSED=0 (fp32 MFMA, exact FMA chains): every gradient within 2e-5 of its
    tensor's largest entry (+ 1e-5 of the largest gradient of the network: biases in front of a BatchNorm have a
    mathematically zero gradient, what is left of them is the cancellation noise of 6747 terms) -- fp32 rounding of a different summation order carried through 5 BatchNorm'ed
    layers, nothing coarser"""
    from pretrain_gnns_amd import ops
    hchem, _ = _hip()
    _, m = _pair(ochem.GNN, hchem.GNN, layers, 300, seed=8)
    m.train()
    d = hostdata.chem_masking_batch(graphs, seed=9).to(DEV)
    w = torch.randn(d.x.size(0), 300, device=DEV)
    grads = []
    for flag in ("2", "0"):  # 2 = transposed weights at every size (the default switches at 16 384 rows)
        monkeypatch.setenv("PGNN_BWD_TRANSPOSED", flag)
        ops.load().pgnn_reload_env()
        m.zero_grad()
        (m(d.x, d.edge_index, d.edge_attr) * w).sum().backward()
        grads.append({k: p.grad.clone() for k, p in m.named_parameters()})
    monkeypatch.delenv("PGNN_BWD_TRANSPOSED")
    ops.load().pgnn_reload_env()
    top = max(float(g.abs().max()) for g in grads[1].values())
    for k in grads[0]:
        scale = float(grads[1][k].abs().max())
        assert float((grads[0][k] - grads[1][k]).abs().max()) <= 2e-5 * scale + 1e-5 * top, (k, scale, top)


@pytest.mark.parametrize("gnn_type", ["gcn", "graphsage"])
@pytest.mark.parametrize("graphs,layers,training,drop", [(48, 5, True, 0.0), (3, 2, True, 0.0), (48, 3, False, 0.0), (400, 4, True, 0.0)])
def test_one_call_gcn_and_graphsage_equal_per_layer_path(gnn_type, graphs, layers, training, drop, monkeypatch):
    """pgnn_chem_lin_stack_fwd/_bwd must be BIT-identical to the per-layer GCN / GraphSAGE calls"""
    import copy
    hchem, _ = _hip()
    _, a = _pair(ochem.GNN, hchem.GNN, layers, 300, seed=6, gnn_type=gnn_type)
    b = copy.deepcopy(a)
    a.train(training), b.train(training)
    d = hostdata.chem_masking_batch(graphs, seed=7).to(DEV)
    w = torch.randn(d.x.size(0), 300, device=DEV)
    res = []
    for m, flag in ((a, True), (b, False)):
        monkeypatch.setattr(hchem, "_STACK_CALL", flag)
        for _ in range(2):
            m.zero_grad()
            out = m(d.x, d.edge_index, d.edge_attr)
            (out * w).sum().backward()
        res.append((out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()},
                    {k: v.clone() for k, v in m.named_buffers()}))
    assert torch.equal(res[0][0], res[1][0])
    for k in res[0][1]:
        if k.startswith("x_embedding"):
            # the one-call path groups the atoms once by (type, chirality) and folds; the per-layer path sums each
            # column on its own: same values, different association order
            scale = float(res[1][1][k].abs().max())
            torch.testing.assert_close(res[0][1][k], res[1][1][k], rtol=1e-4, atol=1e-4 * scale)
        else:
            assert torch.equal(res[0][1][k], res[1][1][k]), k
    for k in res[0][2]:
        assert torch.equal(res[0][2][k], res[1][2][k]), k


@pytest.mark.parametrize("graphs,forced", [(1400, False), (300, True), (37, True)])
def test_fused_mlp_network_equals_the_two_products(graphs, forced, monkeypatch):
    """one-call chem GIN network with both products of every mlp in ONE launch per direction (k_mlp2p_fused, csrc/mlp_fused.hip: from
    32 768 rows on by default, PGNN_MLP_FUSED=2 everywhere) against PGNN_MLP_FUSED=0, the two products on planes: the hidden
    activations have the same bits and so has every product that consumes them, so outputs, every gradient and the running statistics
    are EQUAL where the BatchNorm statistics take the same route (above 32 768 rows: a pass of their own either way); below, the
    fused launch hands per-16-row blocks to the merge launch where the tiled product folds them in its own launch -- the same
    blocks in another association order: fp32 rounding carried through five layers (the bar of the fold's own test)."""
    import copy
    from pretrain_gnns_amd import ops
    hchem, _ = _hip()
    _, a = _pair(ochem.GNN, hchem.GNN, 5, 300, seed=28)
    b = copy.deepcopy(a)
    d = hostdata.chem_masking_batch(graphs, seed=29).to(DEV)
    n = d.x.size(0)
    assert (n >= 32768) != forced
    w = torch.randn(n, 300, device=DEV)
    res = []
    for m, flag in ((a, "2" if forced else "1"), (b, "0")):
        monkeypatch.setenv("PGNN_MLP_FUSED", flag)
        ops.load().pgnn_reload_env()
        m.train()
        for _ in range(2):
            m.zero_grad()
            out = m(d.x, d.edge_index, d.edge_attr)
            (out * w).sum().backward()
        res.append((out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}, {k: v.clone() for k, v in m.named_buffers()}))
    monkeypatch.delenv("PGNN_MLP_FUSED")
    ops.load().pgnn_reload_env()
    if not forced:
        assert torch.equal(res[0][0], res[1][0])
        for k in res[0][1]:
            assert torch.equal(res[0][1][k], res[1][1][k]), k
        for k in res[0][2]:
            assert torch.equal(res[0][2][k], res[1][2][k]), k
    else:
        torch.testing.assert_close(res[0][0], res[1][0], rtol=2e-5, atol=2e-5)
        top = max(float(g.abs().max()) for g in res[1][1].values())
        for k in res[0][1]:
            scale = float(res[1][1][k].abs().max())
            assert float((res[0][1][k] - res[1][1][k]).abs().max()) <= 2e-5 * scale + 1e-5 * top, (k, scale, top)
        for k in res[0][2]:
            if res[0][2][k].is_floating_point():
                torch.testing.assert_close(res[0][2][k], res[1][2][k], rtol=2e-6, atol=1e-7)
            else:
                assert torch.equal(res[0][2][k], res[1][2][k]), k



@pytest.mark.parametrize("gnn_type", ["gin", "gcn"])
def test_direct_gradient_deposit_equals_autograd_accumulation(gnn_type, monkeypatch):
    """the one-call networks write parameter gradients into .grad themselves (ops._DIRECT_GRADS): same bits as
    routing them through autograd, AccumulateGrad's assign-or-add semantics, frozen parameters untouched, and
    an in-place update between forward and backward is refused"""
    import copy
    from pretrain_gnns_amd import ops
    hchem, _ = _hip()
    _, a = _pair(ochem.GNN, hchem.GNN, 3, 300, seed=8, gnn_type=gnn_type)
    b = copy.deepcopy(a)
    d = hostdata.chem_masking_batch(24, seed=9).to(DEV)
    w = torch.randn(d.x.size(0), 300, device=DEV)
    outs = {}
    for m, direct in ((a, True), (b, False)):
        monkeypatch.setattr(ops, "_DIRECT_GRADS", direct)
        out = m(d.x, d.edge_index, d.edge_attr)
        (out * w).sum().backward()
        outs[direct] = (out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()})
    assert torch.equal(outs[True][0], outs[False][0])
    for k in outs[True][1]:
        assert torch.equal(outs[True][1][k], outs[False][1][k]), k
    monkeypatch.setattr(ops, "_DIRECT_GRADS", True)
    (a(d.x, d.edge_index, d.edge_attr) * w).sum().backward()  # second backward without zeroing: accumulate
    a.eval()  # (BatchNorm batch statistics do not depend on the running buffers: same gradients both times)
    for k, p in a.named_parameters():
        assert torch.equal(p.grad, 2 * outs[True][1][k]), k
    a.train()
    a.zero_grad()
    frozen = next(p for n_, p in a.named_parameters() if n_.endswith("gnns.1.edge_embedding1.weight"))
    frozen.requires_grad_(False)
    (a(d.x, d.edge_index, d.edge_attr) * w).sum().backward()
    assert frozen.grad is None and all(p.grad is not None for p in a.parameters() if p.requires_grad)
    frozen.requires_grad_(True)
    out = a(d.x, d.edge_index, d.edge_attr)
    with torch.no_grad():
        a.batch_norms[0].weight.mul_(1.5)
    with pytest.raises(RuntimeError, match="modified between forward and backward"):
        (out * w).sum().backward()
    a.zero_grad()
    out = a(d.x, d.edge_index, d.edge_attr)
    a.batch_norms[0].weight.data = a.batch_norms[0].weight.data.clone()  # re-assigned storage: no version bump
    with pytest.raises(RuntimeError, match="modified between forward and backward"):
        (out * w).sum().backward()
    # a parameter with a hook sends the whole network through autograd, and the hook fires
    a.zero_grad()
    seen = []
    handle = a.batch_norms[1].bias.register_hook(lambda g: seen.append(g.shape))
    (a(d.x, d.edge_index, d.edge_attr) * w).sum().backward()
    handle.remove()
    assert seen == [a.batch_norms[1].bias.shape]
    assert ops.set_direct_grads(False) is True and ops.direct_grads_enabled() is False


def test_large_batch_properties():
    """BASELINE full size (2048 graphs): size-independent checks instead of a slow oracle run --
    linearity of the aggregation in x and agreement of the aggregation with a torch index_add on GPU."""
    from pretrain_gnns_amd import ops
    b = hostdata.chem_masking_batch(2048, seed=8).to(DEV)
    n = b.x.size(0)
    g = ops.build_chem_graph(b.edge_index, b.edge_attr, n)
    g.check()
    torch.manual_seed(0)
    e1 = torch.randn(6, 300, device=DEV)
    e2 = torch.randn(3, 300, device=DEV)
    x = torch.randn(n, 300, device=DEV)
    y = torch.randn(n, 300, device=DEV)
    z = torch.zeros(6, 300, device=DEV), torch.zeros(3, 300, device=DEV)
    ax = ops.ChemAggregate.apply(x, *z, g)
    ay = ops.ChemAggregate.apply(y, *z, g)
    axy = ops.ChemAggregate.apply(x + y, *z, g)
    torch.testing.assert_close(axy, ax + ay, rtol=1e-5, atol=1e-5)
    full = ops.ChemAggregate.apply(x, e1, e2, g)
    ei = torch.cat([b.edge_index, torch.arange(n, device=DEV).repeat(2, 1)], 1)
    ea = torch.cat([b.edge_attr, torch.tensor([[4, 0]], device=DEV).repeat(n, 1)], 0)
    msg = x[ei[1]] + (e1[ea[:, 0]] + e2[ea[:, 1]])
    want = torch.zeros(n, 300, device=DEV).index_add_(0, ei[0], msg)
    torch.testing.assert_close(full, want, rtol=1e-5, atol=1e-5)


def test_single_tiny_graph_and_eval_mode():
    """one 6-atom molecule (N = 6 < one aggregation step), train and eval"""
    hchem, _ = _hip()
    ref, hip = _pair(ochem.GNN, hchem.GNN, 5, 300)
    import numpy as np
    rng = np.random.default_rng(3)
    g = synthetic.zinc_like_graph(rng)
    keep = 6
    sel = (g.edge_index[0] < keep) & (g.edge_index[1] < keep)
    x, ei, ea = g.x[:keep], g.edge_index[:, sel], g.edge_attr[sel]
    out_ref = ref(x, ei, ea)
    out_hip = hip(x.to(DEV), ei.to(DEV), ea.to(DEV))
    torch.testing.assert_close(out_hip.detach().cpu(), out_ref.detach(), rtol=1e-3, atol=1e-3)  # BN over 6 rows
    ref.eval(), hip.eval()
    with torch.no_grad():
        torch.testing.assert_close(hip(x.to(DEV), ei.to(DEV), ea.to(DEV)).cpu(), ref(x, ei, ea), **TOL)


@pytest.mark.parametrize("gnn_type", ["gin", "gcn", "graphsage"])
def test_batch_without_any_edge(gnn_type):
    """empty edge set (a batch of single-atom molecules): only the self loops exist; forward, backward and
    the pooling head must behave like the oracle"""
    hchem, _ = _hip()
    ref, hip = _pair(ochem.GNN_graphpred, hchem.GNN_graphpred, 3, 300, 5, gnn_type=gnn_type, seed=4)
    x = torch.tensor([[5, 0], [7, 1], [6, 0], [5, 2]])
    ei = torch.zeros(2, 0, dtype=torch.int64)
    ea = torch.zeros(0, 2, dtype=torch.int64)
    batch = torch.tensor([0, 1, 2, 3])
    out_ref = ref(x, ei, ea, batch)
    out_hip = hip(x.to(DEV), ei.to(DEV), ea.to(DEV), batch.to(DEV))
    torch.testing.assert_close(out_hip.detach().cpu(), out_ref.detach(), rtol=1e-3, atol=1e-3)  # BatchNorm over 4 rows
    out_hip.sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in hip.parameters() if p.grad is not None)


@pytest.mark.parametrize("emb_dim", [64, 128, 256, 304, 512])
@pytest.mark.parametrize("gnn_type", ["gin", "gcn", "graphsage"])
def test_other_embedding_widths(emb_dim, gnn_type):
    """the reference's --emb_dim is free: every kernel family must work off 300 (generic DMA instantiation,
    the wide-row fallback above 320, other tile counts in the GEMMs), forward and backward, through the
    one-call path"""
    hchem, _ = _hip()
    ref, hip = _pair(ochem.GNN, hchem.GNN, 3, emb_dim, seed=emb_dim, gnn_type=gnn_type)
    b = hostdata.chem_plain_batch(24, seed=emb_dim)
    d = b.clone().to(DEV)
    out_ref = ref(b.x, b.edge_index, b.edge_attr)
    out_hip = hip(d.x, d.edge_index, d.edge_attr)
    torch.testing.assert_close(out_hip.detach().cpu(), out_ref.detach(), **TOL)
    w = torch.randn_like(out_ref)
    (out_hip * w.to(DEV)).sum().backward()
    _grads_close(ref, hip, (b.x, b.edge_index, b.edge_attr), w)
    ref.eval(), hip.eval()
    with torch.no_grad():
        torch.testing.assert_close(hip(d.x, d.edge_index, d.edge_attr).cpu(), ref(b.x, b.edge_index, b.edge_attr), **TOL)


@pytest.mark.parametrize("emb_dim", [64, 256, 312, 512])
@pytest.mark.parametrize("gnn_type", ["gin", "gcn", "graphsage"])
def test_bio_other_embedding_widths(emb_dim, gnn_type):
    """(312 and 512: wider than the 112-row LDS tile of the graph-resident aggregation holds -- pgnn_neighbor_sum_tiled falls
    back to the row-streaming kernel there)"""
    _, hbio = _hip()
    ref, hip = _pair(obio.GNN, hbio.GNN, 3, emb_dim, seed=emb_dim, gnn_type=gnn_type)
    b = hostdata.bio_masking_batch(6, seed=emb_dim)
    d = b.clone().to(DEV)
    out_ref = ref(b.x, b.edge_index, b.edge_attr)
    out_hip = hip(d.x, d.edge_index, d.edge_attr)
    torch.testing.assert_close(out_hip.detach().cpu(), out_ref.detach(), **TOL)
    w = torch.randn_like(out_ref)
    (out_hip * w.to(DEV)).sum().backward()
    _grads_close(ref, hip, (b.x.double(), b.edge_index, b.edge_attr.double()), w)


@pytest.mark.parametrize("num_layer", [2, 7])
def test_other_depths(num_layer):
    hchem, _ = _hip()
    ref, hip = _pair(ochem.GNN, hchem.GNN, num_layer, 300, seed=num_layer)
    b = hostdata.chem_plain_batch(16, seed=num_layer)
    d = b.clone().to(DEV)
    out_ref = ref(b.x, b.edge_index, b.edge_attr)
    out_hip = hip(d.x, d.edge_index, d.edge_attr)
    torch.testing.assert_close(out_hip.detach().cpu(), out_ref.detach(), rtol=3e-4, atol=3e-4)
    w = torch.randn_like(out_ref)
    (out_hip * w.to(DEV)).sum().backward()
    _grads_close(ref, hip, (b.x, b.edge_index, b.edge_attr), w)


def test_class_surface_errors():
    hchem, hbio = _hip()
    with pytest.raises(ValueError):
        hchem.GNN(1, 300)
    with pytest.raises(ValueError):
        hchem.GNN_graphpred(5, 300, 1, graph_pooling="bogus")
    m = hchem.GNN(2, 32).to(DEV)
    with pytest.raises(ValueError):
        m(torch.zeros(1), torch.zeros(1))


def test_top_layer_batchnorm_sums_over_the_masked_rows_only(monkeypatch):
    """pgnn_stack_bwd_dy_rows (round 6): the masking head's gradient is zero outside node_rep[masked_atom_indices]
    (chem/pretrain_masking.py:51-52), so the one-call backward sums its top BatchNorm's column sums over those rows only.  Same
    sums up to the order of the additions: every parameter gradient of a masking step equals the dense pass's (PGNN_SPARSE_TOP_GRAD=0)
    to fp32 rounding, and the hint is dropped when another gradient reaches the backward."""
    import copy
    from pretrain_gnns_amd import ops, train as ptrain
    hchem, _ = _hip()
    torch.manual_seed(31)
    mods = [hchem.GNN(5, 300).to(DEV), torch.nn.Linear(300, 119).to(DEV), torch.nn.Linear(300, 4).to(DEV)]
    b = hostdata.chem_masking_batch(64, seed=32).to(DEV)
    grads = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("PGNN_SPARSE_TOP_GRAD", flag)
        ops.load().pgnn_reload_env()
        ms = copy.deepcopy(mods)
        h = ms[0](b.x, b.edge_index, b.edge_attr)
        loss, _ = ops.masked_head(h, b.masked_atom_indices, ms[1], b.mask_node_label[:, 0])
        loss.backward()
        assert getattr(ops._row_support_tls, "rec", None) is None  # consumed by the backward (same thread: the autograd engine's device thread sets and takes it)
        grads[flag] = {k: p.grad.clone() for k, p in ms[0].named_parameters()}
    monkeypatch.delenv("PGNN_SPARSE_TOP_GRAD")
    ops.load().pgnn_reload_env()
    top = max(float(g.abs().max()) for g in grads["0"].values())
    for k, g1 in grads["1"].items():  # (a bias in front of a BatchNorm has a gradient of pure rounding noise: the floor is the network's scale)
        g0 = grads["0"][k]
        torch.testing.assert_close(g1, g0, rtol=2e-5, atol=2e-6 * float(g0.abs().max()) + 1e-6 * top)
    # a dense gradient (not the head's tensor): the hint of an earlier head backward must not leak into this backward
    ms = copy.deepcopy(mods)
    h = ms[0](b.x, b.edge_index, b.edge_attr)
    loss, _ = ops.masked_head(h.detach().requires_grad_(True), b.masked_atom_indices, ms[1], b.mask_node_label[:, 0])
    loss.backward()  # leaves a hint behind (its dnode went to a leaf, not to a network)
    w = torch.randn_like(h)
    (h * w).sum().backward()
    ref = copy.deepcopy(mods)
    monkeypatch.setenv("PGNN_SPARSE_TOP_GRAD", "0")
    ops.load().pgnn_reload_env()
    (ref[0](b.x, b.edge_index, b.edge_attr) * w).sum().backward()
    monkeypatch.delenv("PGNN_SPARSE_TOP_GRAD")
    ops.load().pgnn_reload_env()
    for (k, p), (_, q) in zip(ms[0].named_parameters(), ref[0].named_parameters()):
        assert torch.equal(p.grad, q.grad), k

"""CPU (-m "not gpu"): the data-parallel layer under gloo, world_size 2.

The DP layer is model-agnostic, so it is exercised here with the CPU oracle as the model: sharding
by graph, one flat-bucket all-reduce per step, identical optimizer updates on every rank.  With
BatchNorm in eval mode (no cross-sample coupling) the weighted all-reduced gradient must equal the
single-process gradient of the whole batch.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from pretrain_gnns_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _loss(model, head, batch):
    rep = model(batch.x, batch.edge_index, batch.edge_attr)
    return F.cross_entropy(head(rep[batch.masked_atom_indices]).double(), batch.mask_node_label[:, 0], reduction="sum")


def _worker(rank, world, port, out_dir):
    import numpy as np
    from oracle import chem as ochem
    from pretrain_gnns_amd.data import synthetic
    from oracle import hostdata

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    r, _, w = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)

    torch.manual_seed(100 + rank)  # deliberately different initial weights per rank
    model, head = ochem.GNN(3, 32), torch.nn.Linear(32, 119)
    parallel.broadcast_parameters([model, head])
    model.eval()

    rng = np.random.default_rng(7)
    graphs = [hostdata.mask_atoms(synthetic.zinc_like_graph(rng), rng) for _ in range(9)]
    mine = [graphs[i] for i in parallel.shard_graphs(len(graphs), rank, world)]
    local = hostdata.collate(mine)
    whole = hostdata.collate(graphs)

    opts = [torch.optim.SGD(model.parameters(), lr=0.1), torch.optim.SGD(head.parameters(), lr=0.1)]
    dp = parallel.AllReduceOptimizers(opts, weight_fn=lambda: 1.0)  # losses are sums -> plain sum of grads
    before = [p.detach().clone() for p in list(model.parameters()) + list(head.parameters())]
    for o in dp:
        o.zero_grad()
    _loss(model, head, local).backward()
    for o in dp:
        o.step()
    after = [p.detach().clone() for p in list(model.parameters()) + list(head.parameters())]
    # the reduced gradients were not copied back: p.grad IS the bucket's view
    in_bucket = all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(dp.bucket.params, dp.bucket.views))
    # second step without dropping the gradients: zero in place (the views), backward accumulates into the bucket directly
    for o in dp:
        o.zero_grad(set_to_none=False)
    zeroed = float(dp.bucket.flat.abs().sum()) == 0.0
    _loss(model, head, local).backward()
    still_in_bucket = all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(dp.bucket.params, dp.bucket.views))
    for o in dp:
        o.step()
    after2 = [p.detach().clone() for p in list(model.parameters()) + list(head.parameters())]

    # single-process reference on the whole batch, from the same (broadcast) weights
    ref_model, ref_head = ochem.GNN(3, 32), torch.nn.Linear(32, 119)
    for p, b in zip(list(ref_model.parameters()) + list(ref_head.parameters()), before):
        p.data.copy_(b)
    ref_model.load_state_dict(model.state_dict(), strict=True)  # buffers too (weights get overwritten next)
    for p, b in zip(list(ref_model.parameters()) + list(ref_head.parameters()), before):
        p.data.copy_(b)
    ref_model.eval()
    _loss(ref_model, ref_head, whole).backward()
    torch.save({"before": before, "after": after, "after2": after2, "flags": (in_bucket, zeroed, still_in_bucket),
                "ref_grads": [p.grad if p.grad is not None else torch.zeros_like(p)
                              for p in list(ref_model.parameters()) + list(ref_head.parameters())],
                "bucket_bytes": dp.bucket.nbytes}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    # broadcast made the ranks identical, and they stay identical after the step
    for a, b in zip(r0["before"], r1["before"]):
        assert torch.equal(a, b)
    for a, b in zip(r0["after"], r1["after"]):
        assert torch.equal(a, b)
    # gradients stay in the bucket; a second step on gradients zeroed in place keeps the ranks identical and moves the weights
    assert r0["flags"] == (True, True, True) and r1["flags"] == (True, True, True)
    for a, b in zip(r0["after2"], r1["after2"]):
        assert torch.equal(a, b)
    assert any(not torch.equal(a, b) for a, b in zip(r0["after"], r0["after2"]))
    # SGD step == lr * (sum over ranks of local grads) == lr * whole-batch grad
    for before, after, g in zip(r0["before"], r0["after"], r0["ref_grads"]):
        torch.testing.assert_close((before - after) / 0.1, g, rtol=2e-4, atol=2e-5)
    n_params = sum(p.numel() for p in r0["before"])
    assert r0["bucket_bytes"] == 4 * n_params  # ONE flat fp32 bucket holds every gradient


def _exact_worker(rank, world, port, out_dir):
    """training-mode BatchNorm: exact statistics + loss weights local_M / global_M == the single-process step"""
    import numpy as np
    from oracle import chem as ochem
    from pretrain_gnns_amd.data import synthetic
    from oracle import hostdata

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    parallel.init_from_env(backend="gloo")
    torch.manual_seed(3)
    model, head = ochem.GNN(3, 32), torch.nn.Linear(32, 119)
    ref_model, ref_head = ochem.GNN(3, 32), torch.nn.Linear(32, 119)
    ref_model.load_state_dict(model.state_dict())
    ref_head.load_state_dict(head.state_dict())
    parallel.use_exact_batchnorm(model)
    assert list(model.state_dict()) == list(ref_model.state_dict())
    rng = np.random.default_rng(11)
    graphs = [hostdata.mask_atoms(synthetic.zinc_like_graph(rng), rng) for _ in range(7)]
    local = hostdata.collate([graphs[i] for i in parallel.shard_graphs(len(graphs), rank, world)])
    whole = hostdata.collate(graphs)
    m_local, m_global = local.masked_atom_indices.numel(), whole.masked_atom_indices.numel()
    opts = [torch.optim.SGD(model.parameters(), lr=0.1), torch.optim.SGD(head.parameters(), lr=0.1)]
    # with statistics shared across ranks the backward of one rank's rows carries terms of EVERY rank's loss, so the
    # share local_M / global_M must multiply the loss itself (not the finished gradient): the bucket then plainly sums
    dp = parallel.AllReduceOptimizers(opts, weight_fn=lambda: 1.0)
    model.train(), ref_model.train()

    def mean_loss(m, h, b):
        rep = m(b.x, b.edge_index, b.edge_attr)
        return F.cross_entropy(h(rep[b.masked_atom_indices]).double(), b.mask_node_label[:, 0])

    for o in dp:
        o.zero_grad()
    (mean_loss(model, head, local) * (m_local / m_global)).backward()
    dp._before_step()  # the all-reduce only: keep the gradients for the comparison
    mean_loss(ref_model, ref_head, whole).backward()
    torch.save({"grads": [p.grad.clone() for p in list(model.parameters()) + list(head.parameters())],
                "ref_grads": [p.grad.clone() for p in list(ref_model.parameters()) + list(ref_head.parameters())],
                "running_mean": model.batch_norms[0].running_mean.clone(), "ref_running_mean": ref_model.batch_norms[0].running_mean.clone(),
                "running_var": model.batch_norms[2].running_var.clone(), "ref_running_var": ref_model.batch_norms[2].running_var.clone()},
               os.path.join(out_dir, "exact%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_exact_batchnorm_and_loss_weights_reproduce_the_single_process_step(tmp_path):
    port = _free_port()
    mp.spawn(_exact_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        d = torch.load(tmp_path / ("exact%d.pt" % r))
        scale = max(float(g.abs().max()) for g in d["ref_grads"])
        for g, ref in zip(d["grads"], d["ref_grads"]):
            assert float((g - ref).abs().max()) <= 2e-5 * scale
        torch.testing.assert_close(d["running_mean"], d["ref_running_mean"], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(d["running_var"], d["ref_running_var"], rtol=1e-5, atol=1e-6)


def test_shard_graphs_partition():
    for n, w in [(9, 2), (256, 8), (5, 8), (2048, 8)]:
        parts = [list(parallel.shard_graphs(n, r, w)) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1


def test_single_process_is_a_no_op():
    lin = torch.nn.Linear(4, 2)
    opt = parallel.AllReduceOptimizers([torch.optim.SGD(lin.parameters(), lr=1.0)])
    lin(torch.ones(1, 4)).sum().backward()
    g = lin.weight.grad.clone()
    opt[0].step()
    assert torch.equal(lin.weight.grad, g)


def test_grad_bucket_lays_late_parameters_behind_the_rest():
    """parallel.GradBucket(params, late=...): the parameters whose gradients a backward finishes last (layers below the milestone,
    the atom embeddings) form the tail of the flat buffer, everything else its head up to ``split`` -- what the overlapped
    all-reduce reduces first; on CPU parameters AllReduceOptimizers ignores ``overlap`` (no communication stream to put it on)"""
    import torch
    from oracle import chem as ochem
    from pretrain_gnns_amd import parallel

    torch.manual_seed(0)
    gnn = ochem.GNN(5, 16, JK="last", drop_ratio=0, gnn_type="gin")
    head = torch.nn.Linear(16, 7)
    params = list(gnn.parameters()) + list(head.parameters())
    top = parallel.top_of_network(gnn, 2)
    assert {id(p) for p in top} == {id(p) for l in (2, 3, 4) for p in list(gnn.gnns[l].parameters()) + list(gnn.batch_norms[l].parameters())}
    late = [p for p in gnn.parameters() if id(p) not in {id(q) for q in top}]
    b = parallel.GradBucket(params, late=late)
    assert [id(p) for p in b.params[b.n_early:]] == [id(p) for p in late]  # (original order kept inside each part)
    assert {id(p) for p in b.params[:b.n_early]} == {id(p) for p in top} | {id(p) for p in head.parameters()}
    assert b.split == sum(p.numel() for p in b.params[:b.n_early]) and b.flat.numel() == sum(p.numel() for p in params)
    off = 0
    for p, v in zip(b.params, b.views):
        assert v.shape == p.shape and v.data_ptr() == b.flat.data_ptr() + 4 * off
        off += p.numel()
    opts = parallel.AllReduceOptimizers([torch.optim.Adam(params)], overlap=(gnn, 2))
    assert opts.overlap_layer is None and opts.comm_stream is None and opts.bucket.n_early == len(opts.bucket.params)


def test_bench_gpus_n_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it (WORLD_SIZE unset) must start its two ranks itself
    instead of exiting.  Without a GPU the ranks end in bench.py's own "needs an MI355X" assertion -- which proves both were
    started with RANK / WORLD_SIZE set, and that the launcher's failure status comes back as bench.py's."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available():
        pytest.skip("the GPU suite runs the literal command to completion (test_gpu_parallel.py)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PGNN_BENCH_WATCHDOG"] = "120"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "--gpus 2 without a launcher: starting" in p.stderr and "--nproc-per-node 2" in p.stderr
    assert p.stderr.count("bench.py needs an MI355X") >= 2, p.stderr[-3000:]  # both ranks got as far as the device check
    assert "needs a torchrun launch" not in p.stderr


def test_bench_ranks_that_never_return_are_dumped_and_ended(tmp_path):
    """A rank that does not come back must not hang the job: every rank of `python bench.py --gpus 2` arms a
    faulthandler watchdog (PGNN_BENCH_WATCHDOG seconds: all thread stacks to stderr, exit 1) and the self-launcher ends its process
    group 60 s after that.  Here both ranks sleep forever (PGNN_BENCH_TEST_HANG_RANK=all, a hook in front of everything else): the command
    comes back non-zero within the watchdog's time with the stack dumps on stderr."""
    import subprocess
    import sys
    import time as _time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PGNN_BENCH_WATCHDOG="6", PGNN_BENCH_TEST_HANG_RANK="all")
    t0 = _time.time()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode != 0
    assert _time.time() - t0 < 120
    assert "Timeout (0:00:06)" in p.stderr, p.stderr[-2000:]  # faulthandler's dump header (the launcher ends the other rank when the first one exits)
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]

#!/usr/bin/env python
"""Benchmark of the hot path: edges/sec through the 5-layer GIN (emb_dim 300) masking pre-train step.

  python bench.py --gpus N --steps K --warmup W
  N > 1 either way: under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
  --master-port P bench.py --gpus N ...: WORLD_SIZE is set, this process is one rank), or bare (`python bench.py --gpus 8`:
  WORLD_SIZE is unset, bench.py starts the N ranks itself through the same launcher on a free port -- self_launch()).

One "step" = the loop body of the reference train() (chem/pretrain_masking.py:47-76): GNN forward
(CSR build + 5 x [aggregation, mlp, BatchNorm]) -> masked-atom head -> float64 CE -> backward ->
3 x Adam, on ONE synthetic ZINC-2M-shaped BatchMasking batch of 256 graphs per GPU (BASELINE.json
configs[1]; weak scaling: every rank has its own 256 graphs, one flat-bucket gradient all-reduce per
step).  Inputs are resident in HBM before the timed region.  edges = edge_index.size(1) (directed,
self loops excluded), counted once per step.  The step's loss / accuracy are summed on the device (the
float64 additions train() does on the host) and fetched once after the K steps, inside the timed region
(--readback epoch, what train.chem_masking_epoch does); `per_step_readback` times the same steps with a
fetch after every optimizer step, `reference_loop` with the reference's own two syncs per step.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      : the aggregation kernel (pgnn_chem_aggregate_fwd) on a roofline-sized batch
                  (>= 16384 graphs, working set > the 256 MB Infinity Cache), algorithmic bytes =
                  2400 N + 6 E + 4 (N+1) per launch / HIP-event time per launch, vs 8 TB/s;
  roofline_mlp  : the forward GEMM of the mlp (two fp16 planes, three MFMA products per accumulator) vs its own ceiling
                  (2500 / 3 TFLOP/s) and vs the fp32 MFMA peak 157.3 TFLOP/s, with the fp32-MFMA kernel on the same shape next to it;
  roofline_mlp_step : the two forward products of one mlp at the TIMED batch's row count on pre-split weight planes (what the
                  one-call networks run there), against the same ceiling;
  cpu_baseline  : the CPU oracle's identical train step on the host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0  # dense, MI355X_MICROARCH.md
MFMA_F32_PEAK_TF = 157.3   # MI355X_MICROARCH.md: fp32 MFMA dense peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--settle-steps", type=int, default=200,
                    help="untimed steps of the same train step in FRONT of the W warm-up steps (same count on every rank): ~0.2 s that take the "
                         "chip out of its idle state -- coming out of an idle gap it runs a few launches at boost clocks, then the power "
                         "controller undershoots for some milliseconds, and a K = 20 window is ~20 ms; 0 = none.  Reported in config")
    ap.add_argument("--graphs-per-gpu", type=int, default=256)
    ap.add_argument("--roofline-graphs", type=int, default=16384)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-hipgraph", action="store_true")
    ap.add_argument("--no-loader", action="store_true", help="skip the device-side loader leg")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the context-prediction (BASELINE configs[2]) and bio masking (configs[4]) legs")
    ap.add_argument("--roofline-only", action="store_true",
                    help="run only the two roofline kernels (for `rocprofv3 --kernel-trace --stats`: the profile then "
                         "holds exactly the launches `roofline.achieved` is computed from)")
    ap.add_argument("--sweep-graphs", default="2048,16384",
                    help="extra per-GPU batch sizes reported under `large_batch` (same step, same code); '' disables")
    ap.add_argument("--adam", default="pgnn", choices=["pgnn", "fused", "foreach"],
                    help="pgnn = one launch for the three optimizers (default); fused / foreach = torch.optim.Adam variants")
    ap.add_argument("--readback", default="epoch", choices=["epoch", "end", "inline"],
                    help="where loss/accuracy reach the host: epoch = summed on the device, one fetch after the timed steps (default); "
                         "end = one fetch per step; inline = the reference's two syncs per step")
    return ap.parse_args()


def make_models(dev, seed=0):
    from pretrain_gnns_amd import ops
    from pretrain_gnns_amd.chem import model as hmodel

    ops.set_direct_grads(True)  # opt-in (off by default): parameter gradients are deposited into .grad by the library

    torch.manual_seed(seed)
    model = hmodel.GNN(5, 300, JK="last", drop_ratio=0, gnn_type="gin").to(dev)
    atoms = torch.nn.Linear(300, 119).to(dev)
    bonds = torch.nn.Linear(300, 4).to(dev)
    return [model, atoms, bonds]


def make_optimizers(mods, kind="pgnn", capturable=False):
    """the three Adam optimizers of chem/pretrain_masking.py:134-136 (lr 1e-3, decay 0).  kind: "pgnn" = one launch for the
    three (pretrain_gnns_amd.optim.Adam.shared: torch's update formula, device-side step count), "fused" / "foreach" =
    torch.optim.Adam's single-kernel-per-optimizer / default multi-kernel implementations"""
    if kind == "pgnn":
        from pretrain_gnns_amd import optim
        return optim.Adam.shared([m.parameters() for m in mods], lr=1e-3, weight_decay=0)
    kw = {"fused": True} if kind == "fused" else {}
    if capturable:
        kw["capturable"] = True
    return [torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=0, **kw) for m in mods]


READBACK_NOTE = {"epoch": "epoch: loss / accuracy summed on the device by the step, ONE fetch after the K timed steps (inside the timed region)",
                 "end": "end: one fetch after every optimizer step", "inline": "inline: the reference's two syncs per step"}
ADAM_NOTE = {"pgnn": "pretrain_gnns_amd.optim.Adam.shared: the three optimizers' update in one launch",
             "fused": "torch.optim.Adam(fused=True)", "foreach": "torch.optim.Adam (foreach)"}


def event_time_ms(fn, iters, warmup=3):
    """average ms per call measured with HIP events on torch's current stream (the stream the
    C-ABI kernels are launched on)."""
    for _ in range(warmup):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def steady_state_ms(launch, warm_s=0.1, iters=50):
    """average ms per launch in the chip's steady state + the per-launch samples.  Coming out of an idle gap the chip
    runs ~6 launches at boost clocks (aggregation: ~210 us), then the power controller undershoots for ~3 ms (~250 us)
    before it settles (~215 us) -- a 20-launch sample taken cold straddles that transient (round 1: 194..267 us inside
    one sample; the MFMA GEMM shows the mirror image, 1184 -> 965 us).  So: `warm_s` seconds of the kernel itself,
    then `iters` launches, each bracketed by its own event pair so that the spread is reported, not hidden."""
    t_end = time.perf_counter() + warm_s
    while time.perf_counter() < t_end:
        for _ in range(20):
            launch()
        torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        launch()
        ev[i + 1].record()
    torch.cuda.synchronize()
    per = torch.tensor([ev[i].elapsed_time(ev[i + 1]) for i in range(iters)], dtype=torch.float64)
    return float(ev[0].elapsed_time(ev[iters]) / iters), per, iters


def _time_aggregation(dev, big, which="plain"):
    """steady-state time of one aggregation instance on ``big`` -> (ms, per-launch ms, launches, nodes, edges, algorithmic bytes).
    which: "plain" = pgnn_chem_aggregate_fwd (layer 0 of the forward); "bn_on_read" = pgnn_chem_aggregate_bn_fwd (layers 1-4: the
    BatchNorm + ReLU of the layer below applied to the rows as they land); "transposed" = pgnn_neighbor_sum on the CSR by source (the
    backward of layer 0's aggregation); "transposed_tail" = pgnn_neighbor_sum_bn_bwd (the backward of layers 1-4: the same sum whose
    launch also reads z [n, 300] of the layer below and leaves its BatchNorm-backward column sums -- bytes: + 1200 n)."""
    from pretrain_gnns_amd import ops

    n, e = big.x.size(0), big.edge_index.size(1)
    g = ops.build_chem_graph(big.edge_index, big.edge_attr, n)
    torch.manual_seed(0)
    x = torch.randn(n, 300, device=dev)
    e1, e2 = torch.randn(6, 300, device=dev), torch.randn(3, 300, device=dev)
    out = torch.empty(n, 300, device=dev)
    lib = ops.load()
    sp = ops.stream_ptr()
    alg = 2400.0 * n + 6.0 * e + 4.0 * (n + 1)
    if which == "plain":
        def launch():
            ops.check(lib.pgnn_chem_aggregate_fwd(x.data_ptr(), 300, g.in_ptr.data_ptr(), g.in_src.data_ptr(),
                                                  g.in_code.data_ptr(), e1.data_ptr(), e2.data_ptr(), None,
                                                  out.data_ptr(), 300, n, 300, sp), "aggregate")
    elif which == "bn_on_read":
        coef = torch.stack([torch.rand(300, device=dev) + 0.5, torch.randn(300, device=dev) * 0.2]).contiguous()

        def launch():
            ops.check(lib.pgnn_chem_aggregate_bn_fwd(x.data_ptr(), 300, coef.data_ptr(), 1, g.in_ptr.data_ptr(), g.in_src.data_ptr(),
                                                     g.in_code.data_ptr(), e1.data_ptr(), e2.data_ptr(), out.data_ptr(), 300, n, 300, sp),
                      "aggregate_bn")
    elif which == "transposed":
        alg = 2400.0 * n + 4.0 * e + 4.0 * (n + 1)  # (no bond codes on the way back)

        def launch():
            ops.check(lib.pgnn_neighbor_sum(x.data_ptr(), 300, g.out_ptr.data_ptr(), g.out_dst.data_ptr(), None, out.data_ptr(), 300, n, 300, sp),
                      "neighbor_sum")
    elif which == "transposed_tail":
        import ctypes
        alg = 3600.0 * n + 4.0 * e + 4.0 * (n + 1)  # + z, read once
        z = torch.randn(n, 300, device=dev)
        gamma, beta = torch.rand(300, device=dev) + 0.5, torch.randn(300, device=dev) * 0.2
        mean, invstd = z.mean(0).contiguous(), (1.0 / torch.sqrt(z.var(0, unbiased=False) + 1e-5)).contiguous()
        dgamma, dbeta = torch.empty(300, device=dev), torch.empty(300, device=dev)
        ws = torch.empty(int(lib.pgnn_bn_workspace_bytes(n, 300)), dtype=torch.uint8, device=dev)
        fused, coef_out = ctypes.c_int(0), ctypes.c_void_p()

        def launch():
            ops.check(lib.pgnn_neighbor_sum_bn_bwd(x.data_ptr(), 300, g.out_ptr.data_ptr(), g.out_dst.data_ptr(), out.data_ptr(), 300,
                                                   z.data_ptr(), 300, gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), invstd.data_ptr(), 1, 1,
                                                   dgamma.data_ptr(), dbeta.data_ptr(), n, 300, ws.data_ptr(), ws.numel(), ctypes.byref(coef_out),
                                                   ctypes.byref(fused), sp), "neighbor_sum_bn_bwd")
        launch()
        assert fused.value == 1, "the BatchNorm-backward tail did not run fused"
    else:
        raise ValueError(which)
    ms, per, iters = steady_state_ms(launch)
    return ms, per, iters, n, e, alg


AGG_INSTANCES = (("plain", 1, "k_aggregate_dma<true,2,10,false,19>: forward, layer 0"),
                 ("bn_on_read", 4, "k_aggregate_dma<true,2,10,true,19>: forward, layers 1-4 (BatchNorm + ReLU on read)"),
                 ("transposed", 1, "k_aggregate_dma<false,2,10,false,19>: backward, layer 0"),
                 ("transposed_tail", 4, "k_aggregate_dma<false,2,10,false,19,false,true>: backward, layers 1-4 (+ z, BatchNorm-backward sums)"))


def roofline_aggregation(dev, graphs):
    """time the GIN aggregation kernel alone at a batch whose working set exceeds the Infinity Cache.  `frac` (round 6, VERDICT r05
    item 1): the plain instance on the batch of SURVEY 8d's generator -- the binding definition of the synthetic input (a tree, parent
    within 3 rows).  Beside it: `smiles_relabelled` (molecules in SMILES parse order -- what chem/loader.py:53-100 feeds -- through
    the loader's once-per-dataset renumbering: the headline of rounds 4-5), `as_fed` (the same molecules without it: 6 % of the edges
    outside the kernel's LDS window), and `in_step`: EVERY aggregation instance a 5-layer train step launches, each on its own bytes."""
    from pretrain_gnns_amd.data import synthetic

    big, info = synthetic.chem_aggregation_batch(graphs, "survey", False, device=dev)
    ms, per, iters, n, e, alg_bytes = _time_aggregation(dev, big)
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic(n, e)
    res = {"bound": "hbm", "kernel": "k_aggregate_dma<true,2,10,false,19> (pgnn_chem_aggregate_fwd)",
           "kernel_note": "rows loaded and stored non-temporally (x + out exceed the Infinity Cache); far source rows fetched a step ahead",
           "achieved": round(gbs, 1),
           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
           "traffic_source": traffic_src, "ms_per_launch": round(ms, 4), "launches_timed": iters,
           "ms_per_launch_std": round(float(per.std()), 4), "ms_per_launch_min": round(float(per.min()), 4),
           "ms_per_launch_max": round(float(per.max()), 4), "warmup": "0.1 s of the same kernel",
           "algorithmic_bytes_per_launch": int(alg_bytes),
           "bytes_per_edge_per_layer": round(alg_bytes / e, 1), "graphs": int(big.batch[-1].item()) + 1,
           "nodes": n, "edges": e, "batch": info, "batch_note": "SURVEY 8d's generator (the binding synthetic input), as fed"}
    # every aggregation launch of one 5-layer train step on this batch, each instance on the bytes IT moves
    in_step, t_sum, b_sum = {}, 0.0, 0.0
    for which, count, kernel in AGG_INSTANCES:
        if which == "plain":
            ms_i, per_i, alg_i = ms, per, alg_bytes
        else:
            ms_i, per_i, _, _, _, alg_i = _time_aggregation(dev, big, which)
        in_step[which] = {"kernel": kernel, "launches_per_step": count, "ms": round(ms_i, 4), "ms_std": round(float(per_i.std()), 4),
                          "bytes": int(alg_i), "frac": round(alg_i / (ms_i * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          "traffic": pmc_traffic_instance(n, e, which)}
        t_sum += count * ms_i
        b_sum += count * alg_i
    in_step["all_ten_launches"] = {"ms": round(t_sum, 4), "bytes": int(b_sum), "frac": round(b_sum / (t_sum * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    in_step["in_a_profiled_step"] = "profiles/r06/step_b16384_kernel_stats.csv (rocprofv3, the same instances inside a 16 384-graph step)"
    res["in_step"] = in_step
    del big
    for tag, order, relabel in (("smiles_relabelled", "smiles", True), ("as_fed", "smiles", False)):
        b2, i2 = synthetic.chem_aggregation_batch(graphs, order, relabel, device=dev)
        ms2, per2, _, n2, e2, alg2 = _time_aggregation(dev, b2)
        res[tag] = {"frac": round(alg2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "ms_per_launch": round(ms2, 4),
                    "ms_per_launch_std": round(float(per2.std()), 4), "nodes": n2, "edges": e2, "batch": i2}
        if tag == "smiles_relabelled":
            msp, perp, _, _, _, algp = _time_aggregation(dev, b2, "bn_on_read")
            res[tag]["bn_on_read"] = {"frac": round(algp / (msp * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "ms_per_launch": round(msp, 4)}
        del b2
    # the same kernel on the SMILES-ordered molecules WITHOUT the loader's renumbering: what the class surface
    # gets under the reference's own DataLoader
    res["frac_as_fed"] = res["as_fed"]["frac"]
    return res


def aggregation_robustness(dev, graphs):
    """the same roofline launch on atom orders less local than SURVEY 8d's generator -- parent of atom i uniform in [i - 8 / 12 / 16,
    i), SMILES parse order, atoms relabelled at random within each molecule -- each AS FED and through the loader's once-per-dataset
    renumbering (``ResidentDataset(relabel=True)``, data/relabel.py), with the fraction of edges whose source row lies outside the
    kernel's LDS window for its destination's 8-node step ([base - 8, base + 16), base = 8 floor(i / 8): those take the far-row
    path).  Same algorithmic bytes formula, same 8 TB/s denominator."""
    import numpy as np
    from pretrain_gnns_amd.data import relabel, resident, synthetic

    out = {}
    makers = (("survey_order", lambda r: synthetic.zinc_like_graph(r)),
              ("parent_within_8", lambda r: synthetic.zinc_like_graph(r, parent_window=8)),
              ("parent_within_12", lambda r: synthetic.zinc_like_graph(r, parent_window=12)),
              ("parent_within_16", lambda r: synthetic.zinc_like_graph(r, parent_window=16)),
              ("smiles_order", synthetic.zinc_like_graph_smiles),
              ("atoms_permuted", lambda r: synthetic.zinc_like_graph(r, permute=True)))
    for tag, make in makers:
        rng = np.random.default_rng(777)
        gl = [make(rng) for _ in range(2048)]
        for suffix, rl in (("", False), ("_relabelled", True)):
            if rl and tag == "survey_order":
                continue
            base = resident.ResidentDataset.from_graphs(gl, dev, relabel=rl).collate(np.arange(len(gl)))
            big = synthetic.tile_batch(base, max(1, graphs // 2048)).to(dev)
            dst, src = big.edge_index[0], big.edge_index[1]
            lo = (dst // 8) * 8 - 8
            miss = float(((src < lo) | (src >= lo + 24)).float().mean())
            ms, per, _, n, e, alg = _time_aggregation(dev, big)
            out[tag + suffix] = {"out_of_window_edge_fraction": round(miss, 4), "ms_per_launch": round(ms, 4), "ms_per_launch_std": round(float(per.std()), 4),
                                 "achieved_GBps": round(alg / (ms * 1e-3) / 1e9, 1), "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "nodes": n, "edges": e}
            del big, base
    return out


def pmc_traffic(n, e, name="agg_pmc_traffic.json"):
    """HBM bytes per launch of an aggregation kernel from the committed rocprofv3 PMC pass
    (profiles/r03/agg_pmc_traffic.json, bio_agg_pmc_traffic.json: FETCH_SIZE x2 (gfx950 half-count) + WRITE_SIZE, separate
    --pmc runs of tools/agg_bench.py / tools/bio_tile_pmc.py on this same batch).  Counters cannot be read from inside this
    process, so the figure is only quoted when the recorded batch shape matches; otherwise null."""
    for rnd in ("r06", "r05", "r04", "r03", "r02"):  # (the newest pass whose batch AND kernel this run reproduces)
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", rnd, name)
        try:
            rec = json.load(open(path))
            if rec["nodes"] == n and rec["edges"] == e:
                return int(rec["hbm_bytes_per_launch"]), "profiles/%s/%s" % (rnd, name)
        except (OSError, KeyError, ValueError):
            pass
    return None, None


def knob_leg(dev, args, batch, steps_n, knobs, note):
    """the SAME steps (fresh models, same batch, same stepper) with environment knobs of the library set -- reported beside `value` so
    that alternative arithmetics / paths stay comparable.  Any failure of such a leg is recorded, not raised."""
    from pretrain_gnns_amd import ops

    out = {"knob": " ".join("%s=%s" % kv for kv in knobs.items()), "note": note}
    os.environ.update(knobs)
    try:
        ops.load().pgnn_reload_env()
        mods = make_models(dev)
        opts = make_optimizers(mods, args.adam)
        step, finish = masking_stepper(mods, list(opts), args.readback, dev)
        for _ in range(5):
            step(batch)
        finish()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps_n):
            step(batch)
        loss = finish()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out.update({"ms_per_step": round(1e3 * dt / steps_n, 4), "edges_per_s": round(batch.edge_index.size(1) * steps_n / dt, 1),
                    "mean_loss": round(float(loss), 5)})
    except Exception as exc:  # (an experimental leg must not take the bench line with it)
        out["error"] = "%s: %s" % (type(exc).__name__, exc)
    finally:
        for k in knobs:
            os.environ.pop(k, None)
        try:
            ops.load().pgnn_reload_env()
        except Exception:
            pass
    return out


def pmc_traffic_instance(n, e, which, name="agg_pmc_traffic_instances.json"):
    """HBM bytes per launch of one aggregation instance (profiles/rNN/agg_pmc_traffic_instances.json: the same separate FETCH_SIZE /
    WRITE_SIZE passes over tools/agg_instances.py), quoted only when the recorded batch is this run's; else null"""
    for rnd in ("r06",):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", rnd, name)
        try:
            rec = json.load(open(path))
            if rec["nodes"] == n and rec["edges"] == e:
                return int(rec["instances"][which]["hbm_bytes_per_launch"])
        except (OSError, KeyError, ValueError):
            pass
    return None


def three_plane_products_leg(dev, args, batch, steps_n):
    """PGNN_GEMM_2P=0: the one-call network's forward and backward-data products on three bf16 planes (six MFMA products per
    accumulator, csrc/linear.hip k_gemm3w: round 3's default) instead of two fp16 planes + a power-of-two scale per row"""
    return knob_leg(dev, args, batch, steps_n, {"PGNN_GEMM_2P": "0"}, "round 3's arithmetic: products on three bf16 planes; everything else as in `value`")


def fp32_mfma_step_leg(dev, args, batch, steps_n):
    """PGNN_GEMM_SPLIT=0 PGNN_GEMM_2P=0: EVERY product of the step on v_mfma_f32_16x16x4_f32 (k_gemm, true fp32 operands, no planes) --
    the no-caveat fp32 number beside `value`"""
    return knob_leg(dev, args, batch, steps_n, {"PGNN_GEMM_SPLIT": "0", "PGNN_GEMM_2P": "0"},
                    "every product on v_mfma_f32_16x16x4_f32 (fp32 operands as they are); everything else as in `value`")


def coo_structure_leg(dev, args, batch, steps_n):
    """the same steps on CLONES of the batch's tensors: nothing is attached to them, so GNN.forward builds both CSRs from the int64 COO
    every step (pgnn_chem_graph_build: 6 launches) -- what a batch that did not come from the resident loader costs"""
    from pretrain_gnns_amd.data import Data
    b2 = Data(**{k: (getattr(batch, k).clone() if torch.is_tensor(getattr(batch, k)) else getattr(batch, k)) for k in batch.keys})
    return knob_leg(dev, args, b2, steps_n, {}, "structure rebuilt from edge_index / edge_attr every step (a foreign batch)")


def reference_loop_leg(dev, args, batch, steps_n):
    """the same step with the reference script's loop taken literally: torch's default (foreach, multi-kernel) Adam as
    `optim.Adam(...)` builds it (chem/pretrain_masking.py:134-136), accuracy and loss read back where the script reads them
    (two device->host syncs per step, :54,76), ordinary autograd for every parameter (no direct gradient deposit).
    `value` differs from this by three declared choices: fused Adam, one readback at the end of the step, direct deposit."""
    from pretrain_gnns_amd import ops
    from pretrain_gnns_amd import train as steps

    mods = make_models(dev)
    prev = ops.set_direct_grads(False)
    try:
        opts = [torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=0) for m in mods]
        for _ in range(5):
            steps.chem_masking_step(mods, opts, batch, mask_edge=False, readback="inline")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps_n):
            steps.chem_masking_step(mods, opts, batch, mask_edge=False, readback="inline")
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps_n
    finally:
        ops.set_direct_grads(prev)
    return {"edges_per_s": round(batch.edge_index.size(1) / dt, 1), "ms_per_step": round(dt * 1e3, 4),
            "adam": "foreach (torch default)", "metrics_readback": "inline", "direct_grads": False}


def unchanged_script_leg(dev, args, batch, steps_n):
    """what an UNCHANGED chem/pretrain_masking.py gets: its train() body taken statement by statement
    (:47-78) -- the script's own torch ops for the prediction head (`linear_pred_atoms(node_rep[masked_atom_indices])`, float64
    cross-entropy, compute_accuracy with its .item()), its three torch.optim.Adam, `float(loss.cpu().item())` per step -- around
    the drop-in GNN.  Nothing of train.py / optim.py is used.  Two runs: as imported, and with PGNN_DIRECT_GRADS=1 in the
    environment (no change to the script; the one-call network then writes `.grad` itself -- assign when None, add otherwise,
    what AccumulateGrad does -- which torch.optim.Adam cannot tell from autograd's: tests/test_gpu_models.py)."""
    import torch.nn.functional as F
    from pretrain_gnns_amd import ops

    def compute_accuracy(pred, target):  # chem/pretrain_masking.py:30-31
        return float(torch.sum(torch.max(pred.detach(), dim=1)[1] == target).cpu().item()) / len(pred)

    out = {}
    for tag, direct in (("as_imported", False), ("PGNN_DIRECT_GRADS=1", True)):
        model, linear_pred_atoms, linear_pred_bonds = make_models(dev)
        opts = [torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=0) for m in (model, linear_pred_atoms, linear_pred_bonds)]
        criterion = torch.nn.CrossEntropyLoss()
        prev = ops.set_direct_grads(direct)
        try:
            def one_step():
                node_rep = model(batch.x, batch.edge_index, batch.edge_attr)
                pred_node = linear_pred_atoms(node_rep[batch.masked_atom_indices])
                loss = criterion(pred_node.double(), batch.mask_node_label[:, 0])
                acc = compute_accuracy(pred_node, batch.mask_node_label[:, 0])
                for o in opts:
                    o.zero_grad()
                loss.backward()
                for o in opts:
                    o.step()
                return float(loss.cpu().item()), acc

            model.train()
            for _ in range(5):
                one_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps_n):
                loss, acc = one_step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps_n
            out[tag] = {"ms_per_step": round(dt * 1e3, 4), "edges_per_s": round(batch.edge_index.size(1) / dt, 1), "last_loss": round(loss, 5)}
        except Exception as exc:
            out[tag] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        finally:
            ops.set_direct_grads(prev)
    out["note"] = "torch head ops + three torch.optim.Adam (foreach) + two device->host syncs per step, as the script has them"
    # where an unchanged script's step goes: the same statements with a device sync behind every phase, so the
    # phases are serialised and sum to MORE than the step above; the two library calls are `forward` and `backward`, the rest is
    # the script's own torch code (head ops, accuracy .item(), three foreach Adams) and not reachable from inside the drop-in module
    try:
        model, linear_pred_atoms, linear_pred_bonds = make_models(dev)
        opts = [torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=0) for m in (model, linear_pred_atoms, linear_pred_bonds)]
        criterion = torch.nn.CrossEntropyLoss()
        ph = {k: 0.0 for k in ("forward", "head+loss", "accuracy(.item)", "zero_grad", "backward", "adam x3", "loss.item")}
        reps = 25

        def timed(key, fn):
            t0 = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            ph[key] += time.perf_counter() - t0
            return r

        for it in range(reps + 5):
            if it == 5:
                ph = {k: 0.0 for k in ph}
            node_rep = timed("forward", lambda: model(batch.x, batch.edge_index, batch.edge_attr))
            pred_node, loss = timed("head+loss", lambda: (lambda p: (p, criterion(p.double(), batch.mask_node_label[:, 0])))(
                linear_pred_atoms(node_rep[batch.masked_atom_indices])))
            timed("accuracy(.item)", lambda: compute_accuracy(pred_node, batch.mask_node_label[:, 0]))
            timed("zero_grad", lambda: [o.zero_grad() for o in opts])
            timed("backward", loss.backward)
            timed("adam x3", lambda: [o.step() for o in opts])
            timed("loss.item", lambda: float(loss.cpu().item()))
        out["script_phases_us"] = {k: round(1e6 * v / reps, 1) for k, v in ph.items()}
        out["script_phases_note"] = "as imported, one device sync per phase (serialised); forward + backward are the library, the rest the script's torch code"
    except Exception as exc:
        out["script_phases_us"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    return out


def forward_only(dev, mods, batch, iters):
    """edges/s of the GNN forward alone (training-mode BatchNorm, no autograd tape): SURVEY 8(d)(ii)."""
    model = mods[0]

    def fwd():
        with torch.no_grad():
            model(batch.x, batch.edge_index, batch.edge_attr)

    ms = event_time_ms(fwd, iters=iters, warmup=5)
    return {"edges_per_s": round(batch.edge_index.size(1) / (ms * 1e-3), 1), "ms_per_pass": round(ms, 4)}


def masking_stepper(mods, opts, readback, dev, mask_edge=False):
    """(step(batch), finish()) for a run of chem masking train steps under one read-back mode.  "epoch": the three epoch
    sums stay on the device (train.epoch_accumulator) and finish() fetches them -- inside the timed region -- as
    train.chem_masking_epoch does; "end" / "inline" fetch every step.  finish() returns the mean loss of the steps run."""
    from pretrain_gnns_amd import train as steps

    state = {"accum": steps.epoch_accumulator(dev) if readback == "epoch" else None, "loss": 0.0, "n": 0}

    def step(batch):
        out = steps.chem_masking_step(mods, opts, batch, mask_edge=mask_edge, readback=readback, accum=state["accum"])
        if out is not None:
            state["loss"] += out[0]
            state["n"] += 1

    def finish():
        if state["accum"] is not None:
            vals = state["accum"].cpu().tolist()
            state["accum"].zero_()
            return vals[0] / max(vals[3], 1.0)
        mean = state["loss"] / max(state["n"], 1)
        state["loss"], state["n"] = 0.0, 0
        return mean

    return step, finish


def resident_loader_leg(dev, args, steps_n):
    """every step draws a NEW 256-graph batch from a dataset resident in HBM
    (device-side collate + MaskAtom, csrc/loader.hip) and trains on it -- the end-to-end rate with the
    loader inside the timed region; beside it the host cost of the reference-style collate (MaskAtom per
    graph + BatchMasking.from_data_list + H2D) for the same batches."""
    import numpy as np
    from pretrain_gnns_amd import train as steps
    from pretrain_gnns_amd.data import resident, synthetic

    rng = np.random.default_rng(1234)
    graphs = [synthetic.zinc_like_graph(rng) for _ in range(4096)]
    ds = resident.ResidentDataset.from_graphs(graphs, dev)
    loader = resident.ResidentLoader(ds, args.graphs_per_gpu, shuffle=True, seed=1, mask_rate=0.15, drop_last=True)
    mods = make_models(dev)
    opts = make_optimizers(mods)
    edges, done, t0 = 0, 0, None
    step, finish = masking_stepper(mods, opts, args.readback, dev)
    while done < steps_n + 5:
        for batch in loader:
            if done == 5:  # warm-up done
                finish()
                torch.cuda.synchronize()
                t0, edges = time.perf_counter(), 0
            step(batch)
            edges += batch.edge_index.size(1)
            done += 1
            if done >= steps_n + 5:
                break
    finish()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ids = loader.batch_ids(0)[:8]
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for i in ids:
        ds.collate(i, mask_rate=0.15, seed=3)
    e.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    from oracle import hostdata  # (the CPU side of this comparison: the host restatement of the reference's transform + collate)
    for i in ids[:4]:
        hostdata.collate([hostdata.mask_atoms(graphs[j], rng) for j in i]).to(dev)
    torch.cuda.synchronize()
    host_ms = (time.perf_counter() - t1) / 4 * 1e3
    return {"edges_per_s": round(edges / dt, 1), "ms_per_step": round(dt / steps_n * 1e3, 4),
            "device_collate_mask_ms": round(s.elapsed_time(e) / len(ids), 4), "host_collate_mask_h2d_ms": round(host_ms, 3),
            "dataset_graphs": len(graphs)}


def roofline_mlp(dev, rows):
    """time the forward product of the GIN mlp (first Linear 300->600 + bias + ReLU) alone at a large row count: the kernel the
    one-call networks run there since round 4 (two fp16 planes + row scales; from 16 384 rows on k_gemm2pr, the weight planes resident
    in LDS), with the tiled two-plane kernel (PGNN_GEMM2P_RES=0: k_gemm2pw, what smaller batches run), the three-bf16-plane kernel of
    pgnn_linear_fwd (three bf16 planes, k_gemm3: rounds 2-3) and the fp32-MFMA kernel (PGNN_GEMM_SPLIT=0, which keeps the
    smallest shapes) beside it.  Fractions against the planes' own ceiling (dense fp16 MFMA peak / 3) AND the fp32 MFMA peak."""
    import os
    from pretrain_gnns_amd import ops

    torch.manual_seed(0)
    x = torch.randn(rows, 300, device=dev)
    w = torch.randn(600, 300, device=dev) * 0.05
    b = torch.randn(600, device=dev)
    y = torch.empty(rows, 600, device=dev)
    lib, sp = ops.load(), ops.stream_ptr()
    (p2,) = ops.weight_planes_2p([w])

    def launch2p():
        ops.linear_fwd_2p(x, p2, b, 600, relu=True, out=y)

    def launch():
        ops.check(lib.pgnn_linear_fwd(x.data_ptr(), 300, w.data_ptr(), b.data_ptr(), y.data_ptr(), 600, rows, 300, 600,
                                      1, sp), "linear")

    flops = 2.0 * rows * 300 * 600
    ms2, per2, iters2 = steady_state_ms(launch2p, iters=30)
    tf2 = flops / (ms2 * 1e-3) / 1e12
    ms, per, iters = steady_state_ms(launch, iters=30)
    tf = flops / (ms * 1e-3) / 1e12
    os.environ["PGNN_GEMM2P_RES"] = "0"
    lib.pgnn_reload_env()
    try:
        mst, pert, _ = steady_state_ms(launch2p, iters=30)
    finally:
        del os.environ["PGNN_GEMM2P_RES"]
        lib.pgnn_reload_env()
    tft = flops / (mst * 1e-3) / 1e12
    prev = os.environ.get("PGNN_GEMM_SPLIT")
    os.environ["PGNN_GEMM_SPLIT"] = "0"
    lib.pgnn_reload_env()
    try:
        ms32, per32, _ = steady_state_ms(launch, iters=30)
    finally:
        if prev is None:
            del os.environ["PGNN_GEMM_SPLIT"]
        else:
            os.environ["PGNN_GEMM_SPLIT"] = prev
        lib.pgnn_reload_env()
    tf32 = flops / (ms32 * 1e-3) / 1e12
    planes_peak, split_peak = MFMA_BF16_PEAK_TF / 3.0, MFMA_BF16_PEAK_TF / 6.0
    fused = fused_mlp_pair(dev, rows, planes_peak)
    return {"bound": "mfma", "fused_pair": fused, "mfma_util": MFMA_UTIL_PMC["k_gemm2pr<10,120>"], "mfma_util_source": MFMA_UTIL_SOURCE,
            "kernel": "k_gemm2pr<10,120,8,EPI_BIAS> (pgnn_linear_fwd_2p 300->600 from 16 384 rows on: fp32 values as two fp16 planes under a "
                      "power-of-two scale per row, three v_mfma_f32_16x16x32_f16 products per k-step, fp32 accumulate; the planes of 120 weight "
                      "rows resident in LDS per persistent workgroup, the activations streamed through registers a 16-row block ahead, no "
                      "barrier in the loop; the row maxima folded out of the fragments the wave holds; bit-identical to the tiled k_gemm2pw; "
                      "error vs float64 at the fp32-MFMA kernel's, tests/test_gpu_ops.py)",
            "achieved": round(tf2, 2), "peak": round(planes_peak, 1), "unit": "TFLOP/s (fp32-equivalent)",
            "frac": round(tf2 / planes_peak, 4), "frac_of_fp32_mfma_peak": round(tf2 / MFMA_F32_PEAK_TF, 4),
            "peak_note": "dense fp16 MFMA peak 2500 TFLOP/s / 3 products per fp32 product = 833; the fp32 MFMA peak is %.1f" % MFMA_F32_PEAK_TF,
            "ms_per_launch": round(ms2, 4), "ms_per_launch_std": round(float(per2.std()), 4), "launches_timed": iters2,
            "rows": rows,
            "tiled_two_plane_kernel": {"kernel": "k_gemm2pw<128,160,8,1,3,EPI_BIAS> (PGNN_GEMM2P_RES=0: one tile per workgroup, both operands through an LDS "
                                                 "ring; what this leg timed before the resident-plane kernel)",
                                       "achieved": round(tft, 2), "frac": round(tft / planes_peak, 4), "ms_per_launch": round(mst, 4),
                                       "ms_per_launch_std": round(float(pert.std()), 4)},
            "three_plane_kernel": {"kernel": "k_gemm3<128,160,4,2,true,true,EPI_BIAS,false> (pgnn_linear_fwd: three bf16 terms, six products; "
                                             "the per-op entry point's kernel and round 3's roofline_mlp)",
                                   "achieved": round(tf, 2), "peak": round(split_peak, 1), "frac": round(tf / split_peak, 4),
                                   "frac_of_fp32_mfma_peak": round(tf / MFMA_F32_PEAK_TF, 4),
                                   "ms_per_launch": round(ms, 4), "ms_per_launch_std": round(float(per.std()), 4), "launches_timed": iters},
            "fp32_mfma_kernel": {"kernel": "k_gemm<64,160,4,2,true,true,EPI_BIAS> (v_mfma_f32_16x16x4_f32; PGNN_GEMM_SPLIT=0; products "
                                           "below ~160 tiles still run this template)",
                                 "achieved": round(tf32, 2), "peak": MFMA_F32_PEAK_TF, "frac": round(tf32 / MFMA_F32_PEAK_TF, 4),
                                 "ms_per_launch": round(ms32, 4), "ms_per_launch_std": round(float(per32.std()), 4)}}


# MFMA utilisation from hardware counters: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), rocprofv3 --pmc,
# one pass per counter pair (tools/gpu_r05b.sh .. gpu_r05g.sh); recorded figures of this round's kernels -- bench.py itself cannot
# collect them (the counters need the profiler around the process).  Wave counters are 4-cycle quanta; the clock under these
# kernels is 1.5-1.85 GHz (SQ_BUSY_CU_CYCLES / 256 over the event time), not the 2.4 GHz behind the 2 500 TFLOP/s peak.
MFMA_UTIL_PMC = {"k_mlp2p_fused fwd (262144 rows)": 0.517, "k_mlp2p_fused bwd (262144 rows)": 0.430, "k_gemm2pr<10,120>": 0.417,
                 "k_gemm2pr<19,64>": 0.340, "k_gemm2pw<112,160> (6740 rows)": 0.235, "k_gemm2pw<64,160> (6740 rows)": 0.238,
                 "k_gemm3_pair<64,160> (6740 rows)": 0.358}  # (the last three re-taken in round 6: profiles/r06/step_b256_gemm_pmc_summary.txt)
MFMA_UTIL_SOURCE = "profiles/r05/mlp_fused_pmc_summary.txt (large M), profiles/r06/step_b256_gemm_pmc_summary.txt: MFMA_BUSY / (4 CU_BUSY)"
MFMA_UTIL_HOW = "SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES) per launch, rocprofv3 --pmc, one counter pair per pass"



def fused_mlp_pair(dev, rows, planes_peak):
    """both products of the GIN mlp in ONE launch per direction (round 5, csrc/mlp_fused.hip: k_mlp2p_fused -- a wave owns 16 rows
    for both products, the [rows, 600] hidden activation is written once and never re-read, the weights stream from L2 through an LDS
    ring) against the two products on planes it replaces, forward (300 -> 600 + ReLU -> 300) and backward-data, HIP events, steady
    state.  What the one-call chem network runs from 32 768 rows on."""
    import ctypes
    from pretrain_gnns_amd import ops

    lib, sp = ops.load(), ops.stream_ptr()
    k1, n1, n2 = 300, 600, 300
    if not lib.pgnn_mlp_2p_fused_supported(rows, k1, n1, n2):
        return None
    torch.manual_seed(0)
    x = torch.randn(rows, k1, device=dev)
    w1, b1 = torch.randn(n1, k1, device=dev) * 0.05, torch.randn(n1, device=dev) * 0.1
    w2, b2 = torch.randn(n2, n1, device=dev) * 0.05, torch.randn(n2, device=dev)
    p1, p2 = ops.weight_planes_2p([w1, w2])
    p2t, p1t = ops.weight_planes_2p([w2, w1], transpose=[True, True])
    hid, y = torch.empty(rows, n1, device=dev), torch.empty(rows, n2, device=dev)
    ham = torch.zeros(rows, dtype=torch.int32, device=dev)
    dy = torch.randn(rows, n2, device=dev) * 1e-3
    dhid, dx = torch.empty(rows, n1, device=dev), torch.empty(rows, k1, device=dev)

    def fused_fwd():
        ops.check(lib.pgnn_mlp_fwd_2p_fused(x.data_ptr(), k1, p1.data_ptr(), b1.data_ptr(), p2.data_ptr(), b2.data_ptr(), hid.data_ptr(), n1,
                                            y.data_ptr(), n2, rows, k1, n1, n2, None, sp), "mlp fused fwd")

    def two_fwd():
        ham.zero_()
        ops.linear_fwd_2p(x, p1, b1, n1, relu=True, out=hid, y_amax=ham)
        ops.linear_fwd_2p(hid, p2, b2, n2, out=y, x_amax=ham)

    def fused_bwd():
        ops.check(lib.pgnn_mlp_bwd_data_2p_fused(dy.data_ptr(), n2, p2t.data_ptr(), hid.data_ptr(), n1, p1t.data_ptr(), dhid.data_ptr(), n1,
                                                 dx.data_ptr(), k1, rows, n2, n1, k1, sp), "mlp fused bwd")

    def two_bwd():
        ham.zero_()
        ops.check(lib.pgnn_linear_bwd_data_2p(dy.data_ptr(), n2, None, p2t.data_ptr(), hid.data_ptr(), n1, dhid.data_ptr(), n1, rows, n1, n2,
                                              ham.data_ptr(), sp), "bwd 1")
        ops.check(lib.pgnn_linear_bwd_data_2p(dhid.data_ptr(), n1, ham.data_ptr(), p1t.data_ptr(), None, 0, dx.data_ptr(), k1, rows, k1, n1, None,
                                              sp), "bwd 2")

    two_fwd()
    flops = 2.0 * rows * (k1 * n1 + n1 * n2)
    out = {"kernel": "k_mlp2p_fused<10,19> (pgnn_mlp_fwd_2p_fused / pgnn_mlp_bwd_data_2p_fused; H has the bits of the two products', "
                     "tests/test_gpu_ops.py::test_fused_mlp_against_float64_and_the_two_products)", "rows": rows,
           "unit": "TFLOP/s (fp32-equivalent, both products of the pair)", "peak": round(planes_peak, 1)}
    for tag, f_fused, f_two in (("forward", fused_fwd, two_fwd), ("backward_data", fused_bwd, two_bwd)):
        msf, perf, it = steady_state_ms(f_fused, iters=20)
        mst, pert, _ = steady_state_ms(f_two, iters=20)
        tf = flops / (msf * 1e-3) / 1e12
        out[tag] = {"ms_per_launch": round(msf, 4), "ms_per_launch_std": round(float(perf.std()), 4), "launches_timed": it,
                    "achieved": round(tf, 2), "frac": round(tf / planes_peak, 4), "two_products_ms": round(mst, 4),
                    "speedup_vs_two_products": round(mst / msf, 3)}
    out["forward"]["mfma_util"] = MFMA_UTIL_PMC["k_mlp2p_fused fwd (262144 rows)"]
    out["backward_data"]["mfma_util"] = MFMA_UTIL_PMC["k_mlp2p_fused bwd (262144 rows)"]
    out["mfma_util_source"] = MFMA_UTIL_SOURCE
    out["ablation"] = "profiles/r05/mlp_fused_ablation.txt"
    return out


def roofline_mlp_planes(dev, rows, what):
    """the two forward products of one GIN mlp (300->600 + ReLU, 600->300) at `rows` rows as the one-call networks run them since
    round 4 -- weights pre-split into two fp16 planes under a power-of-two scale per row, activations DMA'd as fp32 and split by the
    consuming wave under their own row scales (the second product takes the row maxima the first one's epilogue left), three
    v_mfma_f32_16x16x32_f16 per accumulator (csrc/linear.hip k_gemm2pw) -- each timed alone in the steady state.  Fractions
    against BOTH ceilings: the fp32 MFMA peak (157.3: what an fp32 product may cost at best without the
    planes) and the planes' own ceiling, dense fp16 MFMA peak / 3 products per fp32 product (833)."""
    from pretrain_gnns_amd import ops

    torch.manual_seed(0)
    x = torch.randn(rows, 300, device=dev)
    w1, b1 = torch.randn(600, 300, device=dev) * 0.05, torch.randn(600, device=dev)
    w2, b2 = torch.randn(300, 600, device=dev) * 0.05, torch.randn(300, device=dev)
    p1, p2 = ops.weight_planes_2p([w1, w2])
    hid = torch.empty(rows, 600, device=dev)
    z = torch.empty(rows, 300, device=dev)
    amax = torch.zeros(rows, dtype=torch.int32, device=dev)
    ops.linear_fwd_2p(x, p1, b1, 600, relu=True, out=hid, y_amax=amax)
    planes_peak = MFMA_BF16_PEAK_TF / 3.0
    flops = 2.0 * rows * 300 * 600
    out = {"bound": "mfma", "rows": rows, "what": what, "peak": round(planes_peak, 1), "peak_fp32_mfma": MFMA_F32_PEAK_TF,
           "unit": "TFLOP/s (fp32-equivalent)",
           "kernel": "k_gemm2pw (pgnn_linear_fwd_2p: two fp16 planes + power-of-two row scales per operand, three "
                     "v_mfma_f32_16x16x32_f16 per 16x16x32 block, fp32 accumulate; error against float64 at the fp32-MFMA kernel's, "
                     "tests/test_gpu_ops.py::test_products_on_two_fp16_planes_against_float64)"}
    for tag, fn in (("300_to_600", lambda: ops.linear_fwd_2p(x, p1, b1, 600, relu=True, out=hid)),
                    ("600_to_300", lambda: ops.linear_fwd_2p(hid, p2, b2, 300, out=z, x_amax=amax))):
        ms, per, iters = steady_state_ms(fn, warm_s=0.05, iters=50)
        tf = flops / (ms * 1e-3) / 1e12
        out[tag] = {"achieved": round(tf, 2), "frac": round(tf / planes_peak, 4), "frac_of_fp32_mfma_peak": round(tf / MFMA_F32_PEAK_TF, 4),
                    "ms_per_launch": round(ms, 5), "ms_per_launch_std": round(float(per.std()), 5), "launches_timed": iters}
    if rows < 16384:  # (the tiled kernel's counters were taken at the 256-graph batch)
        out["300_to_600"]["mfma_util"] = MFMA_UTIL_PMC["k_gemm2pw<112,160> (6740 rows)"]
        out["600_to_300"]["mfma_util"] = MFMA_UTIL_PMC["k_gemm2pw<64,160> (6740 rows)"]
        out["weight_gradient_pair_mfma_util"] = MFMA_UTIL_PMC["k_gemm3_pair<64,160> (6740 rows)"]
        out["mfma_util_source"] = MFMA_UTIL_SOURCE
    return out


def host_cpu_model():
    """the `model name` line of /proc/cpuinfo (BASELINE.md asks for the CPU next to the core count)"""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_sample(step_fn, edges, seconds, max_steps=200):
    step_fn()
    n, t0 = 0, time.perf_counter()
    while True:
        step_fn()
        n += 1
        el = time.perf_counter() - t0
        if el > seconds or n >= max_steps:
            return edges * n / el, n, el


def contextpred_leg(dev, args, steps_n, with_cpu):
    """BASELINE configs[2]: chem/pretrain_contextpred.py train step (cbow, mean context pooling, 1 negative) at
    batch_size 256 on one GPU -- 5-layer substructure GNN + 3-layer context GNN, negative-sampling dot-product loss,
    two Adam -- with the transform IN the timed loop: every step extracts the substructure / context pair of 256 new
    molecules on the device (ExtractSubstructureContextPair(5, 4, 7) + BatchSubstructContext, csrc/loader.hip).
    edges = directed edges of the 256 source molecules per step (the unit of the headline metric) next to the edges the
    two GNNs actually traverse."""
    import numpy as np
    from pretrain_gnns_amd import train as steps
    from pretrain_gnns_amd.chem import model as hmodel
    from pretrain_gnns_amd.data import resident, synthetic

    rng = np.random.default_rng(4321)
    graphs = [synthetic.zinc_like_graph(rng) for _ in range(4096)]
    ds = resident.ResidentDataset.from_graphs(graphs, dev)
    loader = resident.ResidentLoader(ds, args.graphs_per_gpu, shuffle=True, seed=2, drop_last=True, substruct_context=(5, 4, 7))
    torch.manual_seed(0)
    ms_, mc_ = hmodel.GNN(5, 300, gnn_type="gin").to(dev), hmodel.GNN(3, 300, gnn_type="gin").to(dev)
    os_, oc_ = make_optimizers((ms_, mc_))
    ms_.train(), mc_.train()
    mol_edges = ds._edges  # directed edges per source molecule (host copy of the slice differences)
    src_edges = gnn_edges = done = 0
    t0 = None
    readback = "epoch" if args.readback == "epoch" else "end"
    accum = steps.epoch_accumulator(dev) if readback == "epoch" else None
    while done < steps_n + 5:
        for ids, batch in zip(loader.batch_ids(loader.epoch), loader):
            if done == 5:
                torch.cuda.synchronize()
                t0, src_edges, gnn_edges = time.perf_counter(), 0, 0
            out = steps.chem_contextpred_step(ms_, mc_, os_, oc_, batch, readback=readback, accum=accum)
            if out is not None:
                loss = out[0]
            src_edges += int(mol_edges[ids].sum())
            gnn_edges += batch.edge_index_substruct.size(1) + batch.edge_index_context.size(1)
            done += 1
            if done >= steps_n + 5:
                break
    if accum is not None:
        loss = accum.cpu().tolist()[0] / max(done, 1)  # the one fetch, inside the timed region (mean over all steps run)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"workload": "chem/pretrain_contextpred.py train step (cbow, mean pooling, 1 negative), substructure GNN 5 layers + context "
                       "GNN 3 layers, emb_dim 300, batch_size %d, device-side ExtractSubstructureContextPair(5,4,7) in the loop "
                       "(BASELINE configs[2])" % args.graphs_per_gpu,
           "ms_per_step": round(dt / steps_n * 1e3, 4), "edges_per_s": round(src_edges / dt, 1),
           "gnn_edges_per_s": round(gnn_edges / dt, 1), "graphs_per_s": round(args.graphs_per_gpu * steps_n / dt, 1),
           "mean_loss" if accum is not None else "last_loss": round(float(loss), 5),
           # the step's dominant kernels (profiles/r03/ctx_step_kernel_stats.csv): the forward / backward-data products of its
           # eight GIN layers on weight planes; timed alone at the row count of this run's last substructure batch
           "roofline": roofline_mlp_planes(dev, int(batch.x_substruct.size(0)), "substructure batch of this run's last step")}
    if with_cpu:
        from oracle import chem as ochem
        from oracle import steps as osteps
        cores = usable_cores()
        torch.set_num_threads(cores)
        from oracle import hostdata
        hb = hostdata.chem_contextpred_batch(args.graphs_per_gpu, seed=0)
        torch.manual_seed(0)
        a, b = ochem.GNN(5, 300), ochem.GNN(3, 300)
        oa, ob = torch.optim.Adam(a.parameters(), lr=1e-3), torch.optim.Adam(b.parameters(), lr=1e-3)
        rng0 = np.random.default_rng(0)  # chem_contextpred_batch(seed=0) draws its molecules from this same stream
        e_src = int(round(float(np.mean(mol_edges)) * args.graphs_per_gpu))
        rate, n, el = _cpu_sample(lambda: osteps.chem_contextpred_step(a, b, oa, ob, hb), e_src, max(2.0, args.cpu_seconds / 3))
        out["cpu_baseline"] = {"value": round(rate, 1), "unit": "edges/s", "cores": cores, "cpu": host_cpu_model(), "kind": "port",
                               "sample": "%d context-prediction train steps of the torch-CPU oracle on one %d-molecule batch "
                                         "(host-built transform outside the timed region), %.1f s" % (n, args.graphs_per_gpu, el)}
    return out


def bio_leg(dev, args, steps_n, with_cpu):
    """BASELINE configs[4] shape on one GPU: bio/pretrain_masking.py train step, 5-layer bio GIN (9-dim edge attributes,
    E' ~ 19 N), 256 PPI-ego-shaped graphs per GPU, device-side collate + MaskEdge in the loop; and the bio aggregation
    (neighbour sum + edge-feature product = one GINConv message/aggregate) alone on a batch that exceeds the Infinity
    Cache, against SURVEY 8d's algorithmic bytes 3604 N + 40 E."""
    import numpy as np
    from pretrain_gnns_amd import ops
    from pretrain_gnns_amd import train as steps
    from pretrain_gnns_amd.bio import model as hbio
    from pretrain_gnns_amd.data import resident, synthetic

    rng = np.random.default_rng(99)
    graphs = [synthetic.ppi_like_graph(rng) for _ in range(1024)]
    ds = resident.ResidentDataset.from_graphs(graphs, dev)
    loader = resident.ResidentLoader(ds, args.graphs_per_gpu, shuffle=True, seed=3, mask_rate=0.15, drop_last=True)
    torch.manual_seed(0)
    mods = [hbio.GNN(5, 300, gnn_type="gin").to(dev), torch.nn.Linear(300, 7).to(dev)]
    opts = make_optimizers(mods)
    for m in mods:
        m.train()
    edges = done = 0
    t0 = None
    readback = "epoch" if args.readback == "epoch" else "end"
    accum = steps.epoch_accumulator(dev) if readback == "epoch" else None
    warm = 8  # (every batch has its own node / edge counts: the caching allocator needs a few of them to settle)
    while done < steps_n + warm:
        for batch in loader:
            if done == warm:
                torch.cuda.synchronize()
                t0, edges = time.perf_counter(), 0
            out = steps.bio_masking_step(mods, opts, batch, readback=readback, accum=accum)
            if out is not None:
                loss = out[0]
            edges += batch.edge_index.size(1)
            done += 1
            if done >= steps_n + warm:
                break
    if accum is not None:
        loss = accum.cpu().tolist()[0] / max(done, 1)  # the epoch's one fetch, inside the timed region (mean over all steps run)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"workload": "bio/pretrain_masking.py train step, 5-layer bio GIN emb_dim=300, batch_size %d PPI-ego-shaped graphs per GPU, "
                       "device-side collate + MaskEdge in the loop (BASELINE configs[4] shape)" % args.graphs_per_gpu,
           "ms_per_step": round(dt / steps_n * 1e3, 4), "edges_per_s": round(edges / dt, 1),
           "edges_per_step": int(edges / steps_n), "mean_loss" if accum is not None else "last_loss": round(float(loss), 5)}
    out["roofline"] = bio_roofline(dev, ds, len(graphs))
    if with_cpu:
        from oracle import bio as obio
        from oracle import steps as osteps
        cores = usable_cores()
        torch.set_num_threads(cores)
        from oracle import hostdata
        hb = hostdata.bio_masking_batch(256, seed=0)  # the GPU leg's batch size (64 graphs before)
        torch.manual_seed(0)
        om = [obio.GNN(5, 300), torch.nn.Linear(300, 7)]
        oo = [torch.optim.Adam(m.parameters(), lr=1e-3) for m in om]
        rate, k, el = _cpu_sample(lambda: osteps.bio_masking_step(om, oo, hb), hb.edge_index.size(1), max(2.0, args.cpu_seconds / 3), 50)
        out["cpu_baseline"] = {"value": round(rate, 1), "unit": "edges/s", "cores": cores, "cpu": host_cpu_model(), "kind": "port",
                               "sample": "%d bio masking train steps of the torch-CPU oracle on one 256-graph batch (%d edges), %.1f s"
                                         % (k, hb.edge_index.size(1), el)}
    return out


def bio_roofline(dev, ds=None, num_graphs=1024):
    """the bio GINConv aggregate alone at a cache-exceeding batch (4 096 PPI-ego-shaped graphs): x [N,300] -> [N,600] =
    [sum_j x_j + x_i | sum_e enc(e) + enc(loop)], against SURVEY 8d's algorithmic bytes 3604 N + 40 E"""
    import numpy as np
    from pretrain_gnns_amd import ops
    from pretrain_gnns_amd.data import resident, synthetic

    if ds is None:
        rng = np.random.default_rng(99)
        ds = resident.ResidentDataset.from_graphs([synthetic.ppi_like_graph(rng) for _ in range(num_graphs)], dev)
    big = ds.collate(np.arange(4096) % num_graphs)
    n, e = big.x.size(0), big.edge_index.size(1)
    graph = ops.build_bio_graph(big.edge_index, big.edge_attr, n, gcn=False)
    x = torch.randn(n, 300, device=dev)
    enc_w, enc_b = torch.randn(300, 9, device=dev), torch.randn(300, device=dev)

    def launch():
        with torch.no_grad():
            ops.BioAggregate.apply(x, enc_w, enc_b, graph)

    ms, per, iters = steady_state_ms(launch, iters=30)
    alg = 3604.0 * n + 40.0 * e
    gbs = alg / (ms * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic(n, e, "bio_agg_pmc_traffic.json")
    return {"bound": "hbm", "kernel": "bio GINConv aggregate = k_neighbor_sum_tile_pipe<false,true> (graph-resident: neighbour sum + edge-feature product in one launch; a loader wave DMAs the next ego net into LDS under the gathers of this one, tiles handed out by ticket; csrc/tile.hip)", "achieved": round(gbs, 1),
                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                       "ms_per_launch": round(ms, 4), "ms_per_launch_std": round(float(per.std()), 4),
                       "algorithmic_bytes_per_launch": int(alg), "nodes": n, "edges": e,
                       # the same launch priced on what this implementation must move (3644 N + 4 E) and on what it did move (PMC)
                       "own_compulsory_bytes_per_launch": int(3644.0 * n + 4.0 * e),
                       "frac_on_own_bytes": round((3644.0 * n + 4.0 * e) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                       "frac_on_traffic": round(traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                       "note": "3604 N + 40 E (SURVEY 8d) counts the fp32 [E,9] attributes a layer would read; this implementation reads "
                               "them once per batch (per-node feature sums), so a layer moves 3600 N + 4 E + 40 N bytes"}


def usable_cores():
    """host cores this process may really use: min(affinity, cgroup cpu quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def hipgraph_replay(dev, args, batch):
    """the identical step captured into one HIP graph and replayed (fixed shapes only; informational --
    `value` above is the eager, shape-agnostic path)."""
    from pretrain_gnns_amd import train as steps

    try:
        mods = make_models(dev)
        opts = make_optimizers(mods)
        epoch = args.readback == "epoch"
        g = steps.GraphedChemMaskingStep(mods, opts, batch, readback="epoch" if epoch else "end")
        for _ in range(5):
            g()
        if epoch:
            g.sums()
        torch.cuda.synchronize()
        t0, n = time.perf_counter(), max(args.steps, 20)
        for _ in range(n):
            out = g()
        loss = g.sums()[0] / n if epoch else out[0]  # (epoch: the one fetch, inside the timed region)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        e = batch.edge_index.size(1)
        return {"edges_per_s": round(e / dt, 1), "ms_per_step": round(dt * 1e3, 4), "mean_loss" if epoch else "last_loss": round(loss, 5),
                "metrics_readback": "epoch" if epoch else "end"}
    except Exception as ex:  # capture support differs between ROCm/torch builds; never break the headline run
        return {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}


def large_batch_sweep(dev, sizes, args):
    """the same train step at larger per-GPU batches (informational: the 256-graph step of the reference is
    launch/tile-quantisation bound on an MI355X; these show what the kernels sustain once the chip is filled)."""
    from pretrain_gnns_amd import train as steps
    from pretrain_gnns_amd.data import synthetic

    out = {}
    base = synthetic.chem_masking_batch(2048, seed=7, device=dev)
    for g in sizes:
        batch = (synthetic.tile_batch(base, g // 2048) if g >= 2048 else synthetic.chem_masking_batch(g, seed=7, device=dev)).to(dev)
        mods = make_models(dev)
        opts = make_optimizers(mods, args.adam)
        step, finish = masking_stepper(mods, opts, args.readback, dev)
        for _ in range(2):
            step(batch)
        finish()
        torch.cuda.synchronize()
        t0, n = time.perf_counter(), 5
        for _ in range(n):
            step(batch)
        finish()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        e = batch.edge_index.size(1)
        out[str(g)] = {"edges_per_s": round(e / dt, 1), "ms_per_step": round(dt * 1e3, 3), "edges": int(e),
                       "mlp_tflops": round(3 * 5 * 720000.0 * batch.x.size(0) / dt / 1e12, 1)}
        del batch, mods, opts
        torch.cuda.empty_cache()
    return out


def _cpu_rate(graphs, threads, seconds):
    from oracle import chem as ochem
    from oracle import hostdata, steps

    torch.set_num_threads(threads)
    torch.manual_seed(0)
    batch = hostdata.chem_masking_batch(graphs, seed=0)
    mods = [ochem.GNN(5, 300), torch.nn.Linear(300, 119), torch.nn.Linear(300, 4)]
    opts = [torch.optim.Adam(m.parameters(), lr=1e-3) for m in mods]
    for _ in range(2):
        steps.chem_masking_step(mods, opts, batch)
    n, t0 = 0, time.perf_counter()
    while True:
        steps.chem_masking_step(mods, opts, batch)
        n += 1
        el = time.perf_counter() - t0
        if el > seconds or n >= 200:
            break
    e = batch.edge_index.size(1)
    return e * n / el, n, e, el


def cpu_baseline(graphs, seconds):
    """the oracle's train step (same synthetic batch shape) on the host cores; SURVEY 8(d) also asks for
    the reference's CPU-runnable config (batch 32) and a single-thread figure -- short samples of both
    ride along under `also`."""
    cores = usable_cores()
    rate, n, e, el = _cpu_rate(graphs, cores, seconds)
    also = {}
    for name, g, t in (("batch32_%dthreads" % cores, 32, cores), ("batch%d_1thread" % graphs, graphs, 1)):
        r, n2, _, el2 = _cpu_rate(g, t, seconds / 4)
        also[name] = {"edges_per_s": round(r, 1), "steps": n2, "seconds": round(el2, 1)}
    torch.set_num_threads(cores)
    return {"value": round(rate, 1), "unit": "edges/s", "cores": cores, "cpu": host_cpu_model(), "kind": "port",
            "sample": "%d train steps (fwd+bwd+3xAdam) of the torch-CPU oracle on one %d-graph batch (%d edges), %.1f s"
                      % (n, graphs, e, el), "also": also}


class _StdoutToStderr:
    """Everything a run prints besides its ONE JSON line goes to stderr -- RCCL writes a version banner to fd 1 when the first
    communicator is created, which would otherwise precede the line the driver parses."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start the N ranks ourselves -- the same
    command line under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>`
    (one process per GPU, RCCL), pass rank 0's ONE JSON line through and return the launcher's exit status.  A rank that does not
    come back: every rank dumps its Python stacks after PGNN_BENCH_WATCHDOG seconds (main() below) and exits 1, torchrun then
    ends the others; the launcher itself is killed (whole process group) 60 s after that."""
    import signal
    import subprocess
    wd = float(os.environ.get("PGNN_BENCH_WATCHDOG", "900"))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", str(max(1, usable_cores() // n))), PGNN_BENCH_SELF_LAUNCHED="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without a launcher: starting %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return proc.wait(timeout=wd + 60 if wd > 0 else None)
    except subprocess.TimeoutExpired:
        print("bench.py: the %d-rank launch did not end within %.0f s: killing its process group" % (n, wd + 60), file=sys.stderr, flush=True)
        os.killpg(proc.pid, signal.SIGKILL)
        proc.wait()
        return 1
    except KeyboardInterrupt:
        os.killpg(proc.pid, signal.SIGTERM)
        raise


def main():
    if "WORLD_SIZE" not in os.environ:
        n = parse().gpus
        if n > 1:
            sys.exit(self_launch(n))
    # a run that does not come back is worse than one that fails: after PGNN_BENCH_WATCHDOG seconds (default 900, 0 = off; a default
    # run takes ~70 s) every thread's Python stack goes to stderr and the process exits with status 1
    wd = float(os.environ.get("PGNN_BENCH_WATCHDOG", "900"))
    if wd > 0:
        import faulthandler
        faulthandler.dump_traceback_later(wd, exit=True)
    hang = os.environ.get("PGNN_BENCH_TEST_HANG_RANK")  # tests only: a rank that never comes back ("all" or a rank number)
    if hang is not None and hang in ("all", os.environ.get("RANK", "0")):
        time.sleep(10 ** 6)
    with _StdoutToStderr():
        line = _run()
    if line is not None:
        print(line, flush=True)


def _run():
    args = parse()
    from pretrain_gnns_amd import parallel
    from pretrain_gnns_amd.data import synthetic
    from pretrain_gnns_amd import train as steps

    rank, local, world = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d inside a launch of WORLD_SIZE=%d ranks" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    if args.roofline_only:  # exactly the launches behind every roofline object of the JSON line (profiles/rNN/roofline_only_kernel_stats.csv)
        return json.dumps({"roofline": roofline_aggregation(dev, args.roofline_graphs), "roofline_mlp": roofline_mlp(dev, 262144),
                           "roofline_mlp_step": roofline_mlp_planes(dev, 6747, "one 256-graph batch (BASELINE configs[1])"),
                           "contextpred_roofline": roofline_mlp_planes(dev, 5100, "substructure batch of a 256-molecule context-prediction step"),
                           "bio_masking_roofline": bio_roofline(dev)})
    mods = make_models(dev)
    parallel.broadcast_parameters(mods)
    opts = make_optimizers(mods, args.adam)
    if world > 1 or dist.is_initialized():
        # PGNN_DP_OVERLAP=1 (opt-in: only one-GPU functional runs of it exist): the all-reduce of the heads and layers 2-4 on a
        # communication stream behind the backward's milestone events, under the backward of layers 1-0
        overlap = (mods[0], 2) if os.environ.get("PGNN_DP_OVERLAP") == "1" else None
        opts = parallel.AllReduceOptimizers(opts, overlap=overlap)
    batch = synthetic.chem_masking_batch(args.graphs_per_gpu, seed=rank, device=dev)  # collated and MaskAtom'ed on the device
    edges_local = batch.edge_index.size(1)

    step, finish = masking_stepper(mods, list(opts), args.readback, dev)
    import gc
    gc.collect()  # BEFORE the warm-up: a collection between warm-up and timed steps leaves the GPU idle for tens of milliseconds, and
    gc.disable()  # the K = 20 window then starts on a chip that has clocked down (1.04 against 1.00 ms per step, profiles/r04/bench_window_ab.txt)
    for _ in range(max(args.settle_steps, 0)):  # (untimed, see --settle-steps; the W warm-up steps follow as the contract has them)
        step(batch)
    finish()
    for _ in range(args.warmup):
        step(batch)
    finish()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sync()  # (the collector is off since before the warm-up: no pauses of the host interpreter inside the timed region)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(batch)
    loss = finish()  # (readback="epoch": the one device->host fetch of the K steps, inside the timed region)
    sync()
    elapsed = time.perf_counter() - t0
    # the same K-step window, repeated: a K = 20 window is ~20 ms, too short to judge to 5 % by itself.  After a
    # 0.2 s clock warm-up of the same steps: at least five consecutive windows, each bracketed like `value`'s.  Reported beside
    # `value`, which stays the driver's window.
    value_windows = None
    if world == 1:
        tw = time.perf_counter()
        while time.perf_counter() - tw < 0.2:
            step(batch)
        finish()
        sync()
        wins = []
        for _ in range(7):
            sync()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step(batch)
            finish()
            sync()
            wins.append(1e3 * (time.perf_counter() - t1) / args.steps)
        sw = sorted(wins)
        value_windows = {"windows": len(wins), "steps_per_window": args.steps, "clock_warmup_s": 0.2,
                         "ms_per_step_min": round(sw[0], 4), "ms_per_step_median": round(sw[len(sw) // 2], 4), "ms_per_step_max": round(sw[-1], 4),
                         "edges_per_s_median": round(edges_local / (sw[len(sw) // 2] * 1e-3), 1), "ms_per_step_each": [round(w, 4) for w in wins]}
    per_step_readback = None
    if world == 1 and args.readback == "epoch":  # the same K steps fetching (loss, correct) after every step, for comparison
        step2, finish2 = masking_stepper(mods, list(opts), "end", dev)
        for _ in range(5):
            step2(batch)
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step2(batch)
        finish2()
        sync()
        dt2 = time.perf_counter() - t1
        per_step_readback = {"ms_per_step": round(1e3 * dt2 / args.steps, 4), "edges_per_s": round(edges_local * args.steps / dt2, 1),
                             "note": "readback='end': one device->host fetch after every optimizer step (round-1/early round-2 headline mode)"}
    gc.enable()

    comm = parallel.comm_report(opts) if dist.is_initialized() else {"initialized": False, "world": 1}
    if dist.is_initialized():
        comm["launched_by"] = "bench.py itself (self_launch)" if os.environ.get("PGNN_BENCH_SELF_LAUNCHED") == "1" else "the caller's launcher"
        comm["overlap_default"] = "off: the all-reduce under the backward is opt-in (PGNN_DP_OVERLAP=1) because no two-GPU RCCL run of it exists"
    if isinstance(opts, parallel.AllReduceOptimizers) and opts.overlap_layer is not None:
        comm["overlap"] = {"from_layer": opts.overlap_layer, "steps_behind_the_milestone": opts.overlapped_steps,
                           "head_bytes": opts.bucket.split * 4}
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    etot = torch.tensor([float(edges_local)], dtype=torch.float64, device=dev)
    if world > 1:
        per_rank = [torch.zeros_like(tmax) for _ in range(world)]
        dist.all_gather(per_rank, tmax)  # every rank's own wall time for the K steps: a straggler shows in `comm`
        comm["ms_per_step_by_rank"] = [round(1e3 * float(t.item()) / args.steps, 4) for t in per_rank]
        comm["ms_per_step_min"], comm["ms_per_step_max"] = min(comm["ms_per_step_by_rank"]), max(comm["ms_per_step_by_rank"])
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(etot, op=dist.ReduceOp.SUM)
    elapsed, edges_total = float(tmax.item()), float(etot.item())
    if world > 1:
        # the collective part of the run is over: the process group goes away HERE, so that no rank sits in a barrier while rank 0
        # times its single-GPU roofline launches below (ranks > 0 simply exit)
        dist.barrier()
        dist.destroy_process_group()

    line = None
    if rank == 0:
        res = {
            "metric": "edges/sec through 5-layer GIN (emb_dim=300) masking pre-train step (fwd+bwd+Adam)",
            "value": round(edges_total * args.steps / elapsed, 1), "unit": "edges/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (mlp operands 2xfp16 planes, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": "chem/pretrain_masking.py train step: 5-layer GIN, emb_dim 300, %d ZINC-shaped graphs/GPU (BASELINE configs[1])"
                                   % args.graphs_per_gpu,
                       "graphs_per_gpu": args.graphs_per_gpu, "global_batch": args.graphs_per_gpu * world,
                       "nodes_per_gpu": int(batch.x.size(0)), "edges_per_gpu": int(edges_local),
                       "parallelism": "dp%d" % world, "mean_loss": round(float(loss), 5),
                       "adam": ADAM_NOTE[args.adam], "metrics_readback": READBACK_NOTE[args.readback],
                       "direct_grads": True, "settle_steps_before_warmup": max(args.settle_steps, 0),
                       # (the driver's record keeps 128 characters of a string: one fact per key)
                       "mlp_products": "fwd / bwd-data: fp32 operands as two fp16 planes, power-of-two scale per row (22 bits), 3 MFMA f16 per accumulator",
                       "mlp_weight_gradients": "three bf16 planes (24 bits), six MFMA bf16 per accumulator, fp32 accumulate",
                       "mlp_accuracy": "error vs float64 <= the fp32-MFMA kernel's (tests/test_gpu_ops.py); a pure fp32-MFMA step: `fp32_mfma_step`",
                       "graph_structure": "int32 CSRs attached to the batch by the resident loader (offset-add at collate time); `coo_structure`: rebuilt per step",
                       "parity": "node embeddings within 1e-4 of the reference run in fp64; vs its fp32 run up to 2.4e-4 where that run is 2.4e-4 off"},
            "comm": comm,
        }
        if value_windows is not None:
            res["value_windows"] = value_windows
        if per_step_readback is not None:
            res["per_step_readback"] = per_step_readback
        if world == 1:
            res["three_plane_products"] = three_plane_products_leg(dev, args, batch, args.steps)
            res["fp32_mfma_step"] = fp32_mfma_step_leg(dev, args, batch, args.steps)
            res["coo_structure"] = coo_structure_leg(dev, args, batch, args.steps)
            res["reference_loop"] = reference_loop_leg(dev, args, batch, max(args.steps // 2, 20))
            res["unchanged_script"] = unchanged_script_leg(dev, args, batch, max(args.steps // 2, 20))
            res["forward_only"] = forward_only(dev, mods, batch, max(args.steps, 20))
        if world == 1 and not args.no_loader:
            res["resident_loader"] = resident_loader_leg(dev, args, max(args.steps, 20))
        if world == 1 and not args.no_hipgraph:
            res["hipgraph_replay"] = hipgraph_replay(dev, args, batch)
        if args.sweep_graphs and world == 1:
            res["large_batch"] = large_batch_sweep(dev, [int(g) for g in args.sweep_graphs.split(",") if g], args)
        if not args.no_roofline:
            res["roofline"] = roofline_aggregation(dev, args.roofline_graphs)
            res["roofline_mlp"] = roofline_mlp(dev, 262144)
            res["roofline_mlp_step"] = roofline_mlp_planes(dev, int(batch.x.size(0)), "the timed %d-graph batch" % args.graphs_per_gpu)
            if world == 1:
                res["aggregation_robustness"] = aggregation_robustness(dev, args.roofline_graphs)
        if world == 1 and not args.no_extra_configs:
            res["contextpred"] = contextpred_leg(dev, args, max(args.steps // 2, 20), not args.no_cpu_baseline)
            res["bio_masking"] = bio_leg(dev, args, max(args.steps // 3, 10), not args.no_cpu_baseline)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.graphs_per_gpu, args.cpu_seconds)
        line = json.dumps(res)
    if dist.is_initialized():  # (a single-rank group, PGNN_DP_FORCE_INIT=1)
        dist.destroy_process_group()
    return line


if __name__ == "__main__":
    main()

"""Context-prediction pre-training (cbow, mean context pooling, one negative per graph), the way chem/pretrain_contextpred.py does it,
on the MI355X stack: the substructure network (5-layer GIN) and the context network (l2 - l1 layers), dataset resident in HBM with
ExtractSubstructureContextPair(k, l1, l2) + BatchSubstructContext on the device -- the loader plans the next batch's extraction in
front of the current train step --, the whole loss as three launches, optional data parallelism (one process per GPU).

    python examples/pretrain_contextpred.py --epochs 2                        # one GPU, synthetic ZINC-shaped corpus
    torchrun --nproc-per-node 8 examples/pretrain_contextpred.py --epochs 2   # 8 GPUs, RCCL all-reduce

With the reference's data on disk, replace ``synthetic_corpus`` by
``ResidentDataset.from_inmemory(dataset.data, dataset.slices, device)`` on its ``MoleculeDataset``.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pretrain_gnns_amd import optim, parallel, train  # noqa: E402
from pretrain_gnns_amd.chem.model import GNN  # noqa: E402
from pretrain_gnns_amd.data import resident, synthetic  # noqa: E402


def synthetic_corpus(num_graphs, device):
    rng = np.random.default_rng(0)
    return resident.ResidentDataset.from_graphs([synthetic.zinc_like_graph(rng) for _ in range(num_graphs)], device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch_size", type=int, default=256, help="graphs per GPU and step")
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--num_layer", type=int, default=5)
    ap.add_argument("--csize", type=int, default=3, help="context size: l2 = l1 + csize (chem/pretrain_contextpred.py:128-130)")
    ap.add_argument("--emb_dim", type=int, default=300)
    ap.add_argument("--neg_samples", type=int, default=1)
    ap.add_argument("--mode", default="cbow", choices=("cbow", "skipgram"))
    ap.add_argument("--gnn_type", default="gin")
    ap.add_argument("--graphs", type=int, default=8192, help="size of the synthetic corpus")
    ap.add_argument("--output_model_file", default="")
    args = ap.parse_args()

    rank, local, world = parallel.init_from_env()
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    torch.manual_seed(0)

    l1 = args.num_layer - 1
    l2 = l1 + args.csize
    dataset = synthetic_corpus(args.graphs, device)
    loader = resident.ResidentLoader(dataset, args.batch_size * world, shuffle=True, seed=0, rank=rank, world_size=world, drop_last=True,
                                     substruct_context=(args.num_layer, l1, l2))
    model_substruct = GNN(args.num_layer, args.emb_dim, JK="last", drop_ratio=0, gnn_type=args.gnn_type).to(device)
    model_context = GNN(int(l2 - l1), args.emb_dim, JK="last", drop_ratio=0, gnn_type=args.gnn_type).to(device)
    parallel.broadcast_parameters([model_substruct, model_context])
    # the reference's two optim.Adam (chem/pretrain_contextpred.py:157-158): same update, one launch for both
    optimizer_substruct, optimizer_context = optim.Adam.shared([model_substruct.parameters(), model_context.parameters()], lr=args.lr)
    if world > 1:
        optimizer_substruct, optimizer_context = parallel.AllReduceOptimizers([optimizer_substruct, optimizer_context])

    for epoch in range(1, args.epochs + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss, acc = train.chem_contextpred_epoch(model_substruct, model_context, optimizer_substruct, optimizer_context, loader,
                                                 neg_samples=args.neg_samples, mode=args.mode)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rank == 0:
            print("epoch %d  balanced loss %.4f  acc %.4f  %.1f steps/s (%d graphs per step)"
                  % (epoch, loss, acc, len(loader) / dt, args.batch_size * world), flush=True)
    if rank == 0 and args.output_model_file:
        torch.save(model_substruct.state_dict(), args.output_model_file + ".pth")  # loads into the reference's GNN_graphpred.from_pretrained


if __name__ == "__main__":
    main()

"""Masked-atom pre-training of the 5-layer GIN, the way chem/pretrain_masking.py does it, on the MI355X
stack: drop-in ``GNN`` classes, dataset resident in HBM with device-side collate + MaskAtom, optional data
parallelism (one process per GPU, launch with torchrun).

    python examples/pretrain_masking.py --epochs 2                        # one GPU, synthetic ZINC-shaped corpus
    torchrun --nproc-per-node 8 examples/pretrain_masking.py --epochs 2   # 8 GPUs, RCCL all-reduce

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
    ap.add_argument("--emb_dim", type=int, default=300)
    ap.add_argument("--mask_rate", type=float, default=0.15)
    ap.add_argument("--gnn_type", default="gin")
    ap.add_argument("--graphs", type=int, default=8192, help="size of the synthetic corpus")
    ap.add_argument("--output_model_file", default="")
    args = ap.parse_args()

    rank, local, world = parallel.init_from_env()
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    torch.manual_seed(0)

    dataset = synthetic_corpus(args.graphs, device)
    # the global batch is batch_size * world graphs; every rank takes its share of each global batch
    loader = resident.ResidentLoader(dataset, args.batch_size * world, shuffle=True, seed=0, mask_rate=args.mask_rate,
                                     rank=rank, world_size=world, drop_last=True)
    model = GNN(args.num_layer, args.emb_dim, JK="last", drop_ratio=0, gnn_type=args.gnn_type).to(device)
    linear_pred_atoms = torch.nn.Linear(args.emb_dim, 119).to(device)
    linear_pred_bonds = torch.nn.Linear(args.emb_dim, 4).to(device)
    model_list = [model, linear_pred_atoms, linear_pred_bonds]
    parallel.broadcast_parameters(model_list)
    # the reference's three optim.Adam (chem/pretrain_masking.py:134-136): same update, one launch for all of them
    optimizer_list = optim.Adam.shared([m.parameters() for m in model_list], lr=args.lr)
    if world > 1:
        optimizer_list = list(parallel.AllReduceOptimizers(optimizer_list))

    for epoch in range(1, args.epochs + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss, acc_node, _ = train.chem_masking_epoch(model_list, optimizer_list, loader)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rank == 0:
            print("epoch %d  loss %.4f  acc %.4f  %.1f steps/s (%d graphs per step)"
                  % (epoch, loss, acc_node, len(loader) / dt, args.batch_size * world), flush=True)
    if rank == 0 and args.output_model_file:
        torch.save(model.state_dict(), args.output_model_file + ".pth")  # loads into the reference's GNN_graphpred.from_pretrained


if __name__ == "__main__":
    main()

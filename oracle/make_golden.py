"""Generate tests/golden/*.pt from the reference's shipped checkpoints.  Test infrastructure only.

Run in the BUILD container (needs /root/reference):  python -m oracle.make_golden

What gets pinned: the reference ships real GCN weights + BatchNorm running
statistics (chem/model_architecture/gcn_contextpred.pth, bio/model_architecture/gcn_masking.pth;
the GIN blobs are absent from the repo).  Each fixture stores
  * the checkpoint's state dict exactly as shipped (the key/shape contract),
  * a small seeded synthetic batch,
  * the eval-mode node embeddings (and train-mode embeddings + one gradient) that the REFERENCE'S OWN
    `model.GNN` (unmodified /root/reference/{chem,bio}/model.py, imported through oracle/refshim)
    computes on it after strict-loading the checkpoint.
tests/test_cpu_oracle.py checks the oracle against them, the -m gpu tests load the same dict into the
HIP-backed classes and must reproduce the stored outputs to 1e-4.
/root/reference does not exist on the GPU box, hence the committed fixtures.
"""
import os

import torch

from oracle import refshim
from oracle import hostdata

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _fixture(kind, ckpt, model, batch):
    sd = torch.load(os.path.join(REF, ckpt), map_location="cpu")
    missing = model.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    model.eval()
    with torch.no_grad():
        out_eval = model(batch.x, batch.edge_index, batch.edge_attr)
    model.train()
    out_train = model(batch.x, batch.edge_index, batch.edge_attr)
    out_train.square().mean().backward()
    first = next(n for n, _ in model.named_parameters() if n.endswith("linear.weight"))  # also GAT's weight_linear
    grad = dict(model.named_parameters())[first].grad.clone()
    return {
        "kind": kind, "checkpoint": ckpt, "state_dict": {k: v.clone() for k, v in sd.items()},
        "batch": {k: getattr(batch, k) for k in ("x", "edge_index", "edge_attr")},
        "out_eval": out_eval, "out_train": out_train.detach(), "grad_name": first, "grad": grad,
    }


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)
    torch.manual_seed(0)
    ochem, obio = refshim.load("chem").model, refshim.load("bio").model  # the reference's classes
    fx = _fixture("chem", "chem/model_architecture/gcn_contextpred.pth", ochem.GNN(5, 300, gnn_type="gcn"),
                  hostdata.chem_plain_batch(6, seed=11))
    torch.save(fx, os.path.join(OUT, "chem_gcn_contextpred.pt"))
    fx = _fixture("bio", "bio/model_architecture/gcn_masking.pth", obio.GNN(5, 300, gnn_type="gcn"),
                  hostdata.bio_masking_batch(3, seed=12))
    torch.save(fx, os.path.join(OUT, "bio_gcn_masking.pt"))
    fx = _fixture("chem", "chem/model_architecture/graphsage_contextpred.pth", ochem.GNN(5, 300, gnn_type="graphsage"),
                  hostdata.chem_plain_batch(6, seed=13))
    torch.save(fx, os.path.join(OUT, "chem_graphsage_contextpred.pt"))
    fx = _fixture("bio", "bio/model_architecture/graphsage_masking.pth", obio.GNN(5, 300, gnn_type="graphsage"),
                  hostdata.bio_masking_batch(3, seed=14))
    torch.save(fx, os.path.join(OUT, "bio_graphsage_masking.pt"))
    fx = _fixture("chem", "chem/model_architecture/gat_contextpred.pth", ochem.GNN(5, 300, gnn_type="gat"),
                  hostdata.chem_plain_batch(6, seed=15))
    torch.save(fx, os.path.join(OUT, "chem_gat_contextpred.pt"))
    fx = _fixture("bio", "bio/model_architecture/gat_masking.pth", obio.GNN(5, 300, gnn_type="gat"),
                  hostdata.bio_masking_batch(3, seed=16))
    torch.save(fx, os.path.join(OUT, "bio_gat_masking.pt"))
    # vocabulary / layout constants the reference fixes (chem/model.py:9-13,43; chem/util.py:212-213)
    torch.save({"num_atom_type": 120, "num_chirality_tag": 3, "num_bond_type": 6, "num_bond_direction": 3,
                "self_loop_bond_type": 4, "atom_mask_token": 119, "bond_mask_token": 5,
                "edge_order_example": torch.tensor([[0, 1, 1, 2, 2, 3], [1, 0, 2, 1, 3, 2]])},
               os.path.join(OUT, "constants.pt"))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()

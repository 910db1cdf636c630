"""soak of the two-rank data-parallel flow on ONE GPU (gloo; RCCL refuses two ranks on a device): `iters` masking train steps per
rank over variable-shape loader batches -- HIP GNN + heads, direct gradient deposit, AllReduceOptimizers over the shared Adam,
alternating the single collective and the overlapped form (gradient milestone) every 50 steps -- with faulthandler armed in every
rank: a rank that stops returning dumps its Python stacks and exits instead of hanging (one unexplained
non-return in round 3, one explained in round 4).  Prints a progress line every 25 steps and the ranks' parameter checksums.
usage: python tools/soak_two_ranks.py [iters=200]"""
import faulthandler
import os
import socket
import sys
import time

import numpy as np
import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def worker(rank, world, port, iters):
    faulthandler.dump_traceback_later(240, exit=True, file=sys.stderr)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      PGNN_DP_BACKEND="gloo")
    import torch.distributed as dist
    from pretrain_gnns_amd import ops, optim, parallel
    from pretrain_gnns_amd import train as steps
    from pretrain_gnns_amd.chem import model as hmodel
    from pretrain_gnns_amd.data import resident, synthetic

    r, local, w = parallel.init_from_env()
    dev = torch.device("cuda", local)
    ops.set_direct_grads(True)
    rng = np.random.default_rng(11)
    ds = resident.ResidentDataset.from_graphs([synthetic.zinc_like_graph(rng) for _ in range(2048)], dev)
    loader = resident.ResidentLoader(ds, 64, shuffle=True, seed=3, mask_rate=0.15, drop_last=True, rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    mods = [hmodel.GNN(5, 300).to(dev), torch.nn.Linear(300, 119).to(dev), torch.nn.Linear(300, 4).to(dev)]
    parallel.broadcast_parameters(mods)
    t0, done, epoch = time.perf_counter(), 0, 0
    while done < iters:
        overlap = (mods[0], 2) if (done // 50) % 2 else None
        opts = parallel.AllReduceOptimizers(optim.Adam.shared([m.parameters() for m in mods], lr=1e-3), overlap=overlap)
        for batch in loader:
            out = steps.chem_masking_step(mods, list(opts), batch, readback="end")
            done += 1
            if done % 25 == 0:
                faulthandler.cancel_dump_traceback_later()
                faulthandler.dump_traceback_later(240, exit=True, file=sys.stderr)
                if rank == 0:
                    print("step %4d  loss %.4f  overlap %s  overlapped_steps %d  %.1f s" % (done, out[0], overlap is not None, opts.overlapped_steps,
                                                                                         time.perf_counter() - t0), flush=True)
            if done >= iters or done % 50 == 0:
                break
        epoch += 1
    torch.cuda.synchronize()
    chk = torch.stack([p.detach().double().sum() for m in mods for p in m.parameters()]).sum()
    both = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(both, chk)
    if rank == 0:
        print("parameter checksums by rank:", [float(b) for b in both], "identical" if float(both[0]) == float(both[1]) else "DIFFERENT", flush=True)
        print("soak ok: %d steps per rank in %.1f s" % (done, time.perf_counter() - t0), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(worker, args=(2, port, iters), nprocs=2, join=True)

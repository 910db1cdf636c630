"""micro-benchmark of the aggregation kernels on a roofline-sized batch (HIP-event timing).
usage: [PGNN_AGG_VARIANT=0|1] [PGNN_AGG_BLOCKS_PER_CU=k] python tools/agg_bench.py [graphs]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pretrain_gnns_amd import ops
from pretrain_gnns_amd.data import synthetic

graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = "cuda"
# bench.py's roofline batch: SMILES-ordered molecules through the loader's renumbering (ORDER=survey|smiles|permuted, RELABEL=0|1 override)
big, info = synthetic.chem_aggregation_batch(graphs, os.environ.get("ORDER", "smiles"), os.environ.get("RELABEL", "1") != "0", device=dev)
print("batch:", info)
n, e = big.x.size(0), big.edge_index.size(1)
g = ops.build_chem_graph(big.edge_index, big.edge_attr, n)
torch.manual_seed(0)
x = torch.randn(n, 300, device=dev); out = torch.empty_like(x)
e1, e2 = torch.randn(6, 300, device=dev), torch.randn(3, 300, device=dev)
lib, sp = ops.load(), ops.stream_ptr()

def timeit(fn, iters=20):
    for _ in range(3): fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / iters

alg = 2400.0 * n + 6.0 * e + 4.0 * (n + 1)
print("nodes %d edges %d algorithmic bytes %d" % (n, e, int(alg)))
def agg():
    ops.check(lib.pgnn_chem_aggregate_fwd(x.data_ptr(), 300, g.in_ptr.data_ptr(), g.in_src.data_ptr(), g.in_code.data_ptr(),
              e1.data_ptr(), e2.data_ptr(), None, out.data_ptr(), 300, n, 300, sp), "agg")
def nsum():
    ops.check(lib.pgnn_neighbor_sum(x.data_ptr(), 300, g.out_ptr.data_ptr(), g.out_dst.data_ptr(), None, out.data_ptr(), 300, n, 300, sp), "ns")
ms = timeit(agg); print("variant", os.environ.get("PGNN_AGG_VARIANT", "1"), "bpc", os.environ.get("PGNN_AGG_BLOCKS_PER_CU", "-"),
                        "aggregate_fwd %.1f us  %.0f GB/s (%.1f%% of 8 TB/s)" % (ms * 1e3, alg / ms / 1e6, alg / ms / 1e6 / 80))
ms = timeit(nsum); print("   neighbor_sum  %.1f us  %.0f GB/s" % (ms * 1e3, alg / ms / 1e6))
ms = timeit(lambda: out.copy_(x)); print("   torch copy    %.1f us  %.0f GB/s (2*N*D*4 bytes)" % (ms * 1e3, 2.0 * n * 1200 / ms / 1e6))
for blocks in (1024, 2048, 4096, 8192, 16384):
    ms = timeit(lambda: ops.check(lib.pgnn_debug_stream_copy(x.data_ptr(), out.data_ptr(), n * 300, blocks, sp), "copy"))
    print("   float4 copy, %5d blocks  %.1f us  %.0f GB/s" % (blocks, ms * 1e3, 2.0 * n * 1200 / ms / 1e6))
ref = torch.zeros_like(x)
if os.environ.get("CHECK"):
    agg(); a = out.clone()
    os.environ["PGNN_AGG_VARIANT"] = "0"; lib.pgnn_reload_env(); agg(); print("   variants bit-equal:", torch.equal(a, out))

"""micro-benchmark of the bio aggregation pieces: structure build (with / without the tile pass), neighbour sum plain vs
graph-resident tiles, edge-feature product; 256 graphs (one training batch) and 4096 graphs (cache-exceeding).
usage: python tools/bio_agg_bench.py"""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pretrain_gnns_amd import ops
from pretrain_gnns_amd.data import synthetic, resident
dev = "cuda"
rng = np.random.default_rng(99)
graphs = [synthetic.ppi_like_graph(rng) for _ in range(1024)]
ds = resident.ResidentDataset.from_graphs(graphs, dev)
def timeit(fn, iters=30, warm=0.05):
    t_end = time.perf_counter() + warm
    while time.perf_counter() < t_end:
        for _ in range(5): fn()
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for ng in (256, 4096):
    b = ds.collate(np.arange(ng) % len(graphs))
    n, e = b.x.size(0), b.edge_index.size(1)
    ops._BIO_TILES = False
    t_build0 = timeit(lambda: ops.build_bio_graph(b.edge_index, b.edge_attr, n))
    ops._BIO_TILES = True
    t_build1 = timeit(lambda: ops.build_bio_graph(b.edge_index, b.edge_attr, n))
    g = ops.build_bio_graph(b.edge_index, b.edge_attr, n)
    x = torch.randn(n, 300, device=dev); out = torch.empty(n, 600, device=dev)
    t_plain = timeit(lambda: ops._neighbor_sum(x, g.in_ptr, g.in_src, None, n, 300, out=out[:, :300]))
    t_tiled = timeit(lambda: ops._neighbor_sum(x, g.in_ptr, g.in_src, None, n, 300, out=out[:, :300], tiles=g.tiles))
    dbg = []
    for m in (1, 2):
        os.environ["PGNN_TILE_DEBUG"] = str(m); ops.load().pgnn_reload_env()
        dbg.append(timeit(lambda: ops._neighbor_sum(x, g.in_ptr, g.in_src, None, n, 300, out=out[:, :300], tiles=g.tiles)))
    os.environ.pop("PGNN_TILE_DEBUG"); ops.load().pgnn_reload_env()
    print("   tiled, no gather loop %.1f us; tiled, no row DMA %.1f us" % tuple(dbg))
    table = torch.randn(10, 300, device=dev)
    t_feat = timeit(lambda: ops._rowfeat_fwd(g.cfeat, table, out[:, 300:], 300, False))
    t_fused = timeit(lambda: ops._neighbor_sum(x, g.in_ptr, g.in_src, None, n, 300, out=out[:, :300], tiles=g.tiles,
                                              feat=(g.cfeat, table, out[:, 300:])))
    print("   fused neighbour sum + edge-feature product (one launch) %.1f us = %.0f GB/s of 3604 N + 40 E" % (t_fused, (3604.0 * n + 40.0 * e) / t_fused / 1e3))
    t_copy = timeit(lambda: out[:, :300].copy_(x))
    print("graphs %d nodes %d edges %d | build %.1f us (+tiles %.1f) | neighbour sum plain %.1f us, tiled %.1f us (%.0f GB/s of 2*N*1200+4E) | "
          "edge-feature product %.1f us | strided copy of x %.1f us" % (ng, n, e, t_build0, t_build1, t_plain, t_tiled,
          (2400.0 * n + 4 * e) / t_tiled / 1e3, t_feat, t_copy), flush=True)

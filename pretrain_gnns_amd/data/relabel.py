"""Per-graph node relabelling for locality (round 4).

The aggregation kernel (csrc/aggregate.hip, ``k_aggregate_dma``) streams node rows through an LDS ring and serves a
neighbour row from LDS when it lies within 8 rows below / 16 rows above its destination's 8-row step; any other source is a
round trip to L2 / HBM.  The reference's atom order is the SMILES parse order (chem/loader.py:53-100): a substituent sits
behind the atom that carries it and the REST of a ring behind the substituent, so 5-6 % of a ZINC batch's edges span more than
that window (``synthetic.zinc_like_graph_smiles``), and an arbitrary order up to a third.

``bandwidth_order`` computes, once per dataset, a Cuthill-McKee order of every graph (breadth-first levels from a
pseudo-peripheral node, nodes of a level ordered by their first parent's position, then by degree): neighbours end up at most
one level apart, and a molecule's levels are 1-4 atoms wide.  Only the node LABELS change: ``edge_index`` keeps its column
order, so every destination still meets its incoming edges in the reference's order and the sums are bit-identical up to the
permutation of the rows (tests/test_gpu_loader.py).  Host-side numpy, level-synchronous over the whole dataset at once (a few
array passes per level; levels <= the longest shortest path of any graph).
"""
import numpy as np


def _bfs_levels(n_nodes, src, dst, graph_of, roots):
    """level of every node from its graph's root (-1: not reached)"""
    level = np.full(n_nodes, -1, dtype=np.int64)
    level[roots] = 0
    cur = 0
    while True:
        on = level[src] == cur
        if not on.any():
            break
        cand = dst[on]
        cand = cand[level[cand] < 0]
        if cand.size == 0:
            break
        level[cand] = cur + 1
        cur += 1
    return level


def bandwidth_order(edge_index, node_slice, edge_slice):
    """edge_index [2, E] int64 with GRAPH-LOCAL node ids (the concatenated ``(data, slices)`` layout), node_slice / edge_slice
    [G + 1].  Returns ``new_of_old`` [sum n] int64: the new graph-local label of every node (a permutation within each graph)."""
    ei = np.asarray(edge_index, dtype=np.int64)
    ns = np.asarray(node_slice, dtype=np.int64)
    es = np.asarray(edge_slice, dtype=np.int64)
    n, g = int(ns[-1]), ns.size - 1
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    graph_of = np.repeat(np.arange(g), np.diff(ns))
    e_graph = np.repeat(np.arange(g), np.diff(es))
    src = ei[0] + ns[e_graph]  # global ids; the graph is symmetric in the reference's data (both directions stored), but nothing
    dst = ei[1] + ns[e_graph]  # below needs that: edges are walked in both directions
    a = np.concatenate([src, dst])
    b = np.concatenate([dst, src])
    deg = np.bincount(a, minlength=n)
    first = ns[:-1]
    has = np.diff(ns) > 0

    def argmin_per_graph(key, mask=None):
        """first node of every graph with the smallest key (among mask), -1 if none"""
        k = key.astype(np.int64) * 2
        if mask is not None:
            k = np.where(mask, k, np.iinfo(np.int64).max // 4)
        order = np.lexsort((np.arange(n), k, graph_of))
        pick = order[np.searchsorted(graph_of[order], np.arange(g))[has.nonzero()[0]]] if has.any() else np.zeros(0, dtype=np.int64)
        out = np.full(g, -1, dtype=np.int64)
        gi = has.nonzero()[0]
        ok = np.ones(gi.size, dtype=bool) if mask is None else mask[pick]
        out[gi[ok]] = pick[ok]
        return out

    rank = np.full(n, -1, dtype=np.int64)  # position inside the graph
    placed = np.zeros(g, dtype=np.int64)   # nodes placed per graph so far
    todo = np.ones(n, dtype=bool)
    while todo.any():  # one round per connected component of the most fragmented graph
        # root: minimum degree among the unplaced nodes, then one George-Liu step -- the farthest node of minimum degree from it
        r0 = argmin_per_graph(deg, todo)
        live = r0 >= 0
        keep = todo[a] & todo[b]
        aa, bb = a[keep], b[keep]
        lv = _bfs_levels(n, aa, bb, graph_of, r0[live])
        far = np.where(lv >= 0, -lv, 1)  # most negative = deepest level
        r1 = argmin_per_graph(far * (deg.max() + 1) + deg, todo & (lv >= 0))
        lv = _bfs_levels(n, aa, bb, graph_of, r1[live])
        comp = lv >= 0
        # Cuthill-McKee inside the component, level by level: key of a node = position of its earliest-placed neighbour in the level above
        roots = r1[live]
        rank[roots] = placed[graph_of[roots]]
        placed[graph_of[roots]] += 1
        cur = 0
        while True:
            nxt = np.nonzero(lv == cur + 1)[0]
            if nxt.size == 0:
                break
            on = (lv[aa] == cur) & (lv[bb] == cur + 1)
            parent_pos = np.full(n, np.iinfo(np.int64).max, dtype=np.int64)
            np.minimum.at(parent_pos, bb[on], rank[aa[on]])
            order = nxt[np.lexsort((nxt, deg[nxt], parent_pos[nxt], graph_of[nxt]))]
            go = graph_of[order]
            start = np.searchsorted(go, go)  # index of the first node of the same graph in this level
            rank[order] = placed[go] + (np.arange(order.size) - start)
            placed += np.bincount(go, minlength=g)
            cur += 1
        todo &= ~comp
    return rank


def apply_order(new_of_old, node_slice):
    """old_of_new [sum n] as GLOBAL row indices: row ``j`` of a relabelled node tensor is row ``old_of_new[j]`` of the original"""
    ns = np.asarray(node_slice, dtype=np.int64)
    graph_of = np.repeat(np.arange(ns.size - 1), np.diff(ns))
    new_global = new_of_old + ns[graph_of]
    old_of_new = np.empty_like(new_global)
    old_of_new[new_global] = np.arange(new_global.size)
    return old_of_new


def window_miss_fraction(edge_index, node_slice, edge_slice, new_of_old=None):
    """fraction of directed edges whose source row lies outside k_aggregate_dma's LDS window of its destination
    ([base - 8, base + 16), base = 8 floor(dst / 8), global row numbers of the concatenated dataset)"""
    ei = np.asarray(edge_index, dtype=np.int64)
    ns = np.asarray(node_slice, dtype=np.int64)
    es = np.asarray(edge_slice, dtype=np.int64)
    e_graph = np.repeat(np.arange(ns.size - 1), np.diff(es))
    d, s = ei[0] + ns[e_graph], ei[1] + ns[e_graph]
    if new_of_old is not None:
        graph_of = np.repeat(np.arange(ns.size - 1), np.diff(ns))
        glob = new_of_old + ns[graph_of]
        d, s = glob[d], glob[s]
    lo = (d // 8) * 8 - 8
    return float(((s < lo) | (s >= lo + 24)).mean()) if d.size else 0.0

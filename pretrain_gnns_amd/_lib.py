"""ctypes binding of ``libpgnn.so`` (the C ABI declared in include/pgnn.h).

This is the same stub a maintainer of the reference would add to bind the library (see
INTEGRATION.md).  There is no fallback: if the shared object is missing or a symbol is absent the
import of any op raises -- the product path never silently degrades to PyTorch/CPU code.
torch must be imported first so that the process-wide HIP runtime (libamdhip64.so.7) is the one
PyTorch-ROCm ships; the library then shares torch's streams and device memory.
"""
import ctypes
import os

import torch  # noqa: F401  (loads libamdhip64 before libpgnn)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpgnn.so")

_p = ctypes.c_void_p
_i64 = ctypes.c_int64
_i = ctypes.c_int
_f = ctypes.c_float
_sz = ctypes.c_size_t
_u64 = ctypes.c_uint64

# name -> (restype, argtypes); mirrors include/pgnn.h one to one
PROTOTYPES = {
    "pgnn_abi_version": (_i, []),
    "pgnn_last_error": (ctypes.c_char_p, []),
    "pgnn_reload_env": (None, []),
    "pgnn_graph_workspace_bytes": (_sz, [_i64, _i64]),
    "pgnn_chem_graph_build": (_i, [_p, _p, _i64, _i64, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "pgnn_bio_graph_build": (_i, [_p, _p, _i64, _i64, _i, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "pgnn_group_workspace_bytes": (_sz, [_i64, _i64]),
    "pgnn_group_by_key": (_i, [_p, _i64, _i64, _i64, _p, _p, _p, _p, _sz, _p]),
    "pgnn_group_by_key_pair": (_i, [_p, _p, _i64, _i64, _i64, _i64, _p, _p, _p, _p, _sz, _p]),
    "pgnn_pair_fold": (_i, [_p, _i64, _i64, _p, _i64, _p, _i64, _i64, _p]),
    "pgnn_chem_aggregate_fwd": (_i, [_p, _i64, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p]),
    "pgnn_chem_aggregate_bn_fwd": (_i, [_p, _i64, _p, _i, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p]),
    "pgnn_neighbor_sum": (_i, [_p, _i64, _p, _p, _p, _p, _i64, _i64, _i64, _p]),
    "pgnn_rowfeat_matmul_fwd": (_i, [_p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i, _p]),
    "pgnn_rowfeat_matmul_bwd_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "pgnn_rowfeat_matmul_bwd": (_i, [_p, _i64, _p, _i64, _p, _i64, _i64, _i64, _p, _sz, _p]),
    "pgnn_embed_fwd": (_i, [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _p, _p]),
    "pgnn_segment_sum_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "pgnn_segment_sum": (_i, [_p, _i64, _p, _p, _i64, _i64, _i, _p, _i64, _i64, _p, _sz, _p]),
    "pgnn_segment_broadcast": (_i, [_p, _i64, _p, _p, _i, _p, _i64, _i64, _i64, _p]),
    "pgnn_bn_workspace_bytes": (_sz, [_i64, _i64]),
    "pgnn_bn_fwd": (_i, [_p, _i64, _p, _p, _p, _p, _f, _f, _i, _i, _p, _i64, _p, _p, _f, _u64, _i64, _i64, _p, _sz, _p]),
    "pgnn_bn_stats_fwd": (_i, [_p, _i64, _p, _p, _p, _p, _f, _f, _i, _p, _p, _p, _i64, _i64, _p, _sz, _p]),
    "pgnn_bn_bwd": (_i, [_p, _i64, _p, _i64, _p, _p, _p, _p, _i, _i, _p, _i64, _p, _p, _f, _u64, _i64, _i64, _p, _sz,
                         _p]),
    "pgnn_mean_l2norm_fwd": (_i, [_p, _i64, _p, _p, _i64, _p, _i64, _i64, _p]),
    "pgnn_mean_l2norm_bwd": (_i, [_p, _i64, _p, _i64, _p, _p, _p, _i64, _i64, _i64, _p]),
    "pgnn_linear_fwd": (_i, [_p, _i64, _p, _p, _p, _i64, _i64, _i64, _i64, _i, _p]),
    "pgnn_linear_fwd_colstats": (_i, [_p, _i64, _p, _p, _p, _i64, _i64, _i64, _i64, _i, _p, _p]),
    "pgnn_bn_stats_fwd_blocks": (_i, [_p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _i64, _i64, _p]),
    "pgnn_bn_apply_fwd": (_i, [_p, _i64, _p, _i, _p, _i64, _f, _u64, _i64, _i64, _p]),
    "pgnn_debug_gemm3w_profile": (_i, [_p, _i64, _p, _p, _p, _i64, _i64, _i64, _i64, _i, _p, _p]),
    "pgnn_contextpred_loss_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "pgnn_contextpred_loss_fwd": (_i, [_p, _i64, _i64, _p, _p, _i64, _i64, _p, _p, _i64, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "pgnn_contextpred_loss_bwd": (_i, [_p, _i64, _i64, _p, _i64, _p, _p, _i64, _i64, _i64, _i64, _p, _p, _i64, _p, _i64, _p, _sz, _p]),
    "pgnn_weight_planes_bytes": (_sz, [_i64, _i64]),
    "pgnn_linear_wp_preferred": (_i, [_i64, _i64, _i64]),
    "pgnn_split_weights": (_i, [_p, _p, _p, _p, _p, _i64, _p]),
    "pgnn_linear_fwd_wp": (_i, [_p, _i64, _p, _p, _p, _i64, _i64, _i64, _i64, _i, _p, _p]),
    "pgnn_linear_bwd_data_wp": (_i, [_p, _i64, _p, _p, _i64, _p, _i64, _i64, _i64, _i64, _p]),
    "pgnn_split_weights_2p": (_i, [_p, _p, _p, _p, _p, _i64, _p]),
    "pgnn_stack_bwd_milestone_arm": (_i, [_i, _p]),
    "pgnn_stack_bwd_milestone_wait": (_i, [_p]),
    "pgnn_linear_fwd_2p": (_i, [_p, _i64, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i, _p, _p, _p]),
    "pgnn_linear_bwd_data_2p": (_i, [_p, _i64, _p, _p, _p, _i64, _p, _i64, _i64, _i64, _i64, _p, _p]),
    "pgnn_mlp_2p_fused_supported": (_i, [_i64, _i64, _i64, _i64]),
    "pgnn_mlp_fwd_2p_fused": (_i, [_p, _i64, _p, _p, _p, _p, _p, _i64, _p, _i64, _i64, _i64, _i64, _i64, _p, _p]),
    "pgnn_mlp_bwd_data_2p_fused": (_i, [_p, _i64, _p, _p, _i64, _p, _p, _i64, _p, _i64, _i64, _i64, _i64, _i64, _p]),
    "pgnn_linear_bwd_data": (_i, [_p, _i64, _p, _p, _i64, _p, _i64, _i64, _i64, _i64, _p]),
    "pgnn_bio_gin_stack_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "pgnn_bio_gin_stack_fwd": (_i, [_p, _i64, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _i64, _i64, _p, _sz, _p]),
    "pgnn_bio_gin_stack_bwd": (_i, [_p, _i64, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _i64, _i64, _p, _sz, _p]),
    "pgnn_adam_max_tensors": (_i, []),
    "pgnn_adam_step": (_i, [_p, _p, _p, _p, _i64, _p, _p, _p, _f, _f, _f, _f, _f, _p]),
    "pgnn_masked_head_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "pgnn_masked_head_fwd": (_i, [_p, _i64, _i64, _p, _i64, _p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "pgnn_masked_head_bwd": (_i, [_p, _i64, _i64, _p, _i64, _p, _p, _i64, _p, _p, _i64, _i64, _p, _i64, _p, _p, _p, _sz, _p]),
    "pgnn_edge_head_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "pgnn_edge_head_fwd": (_i, [_p, _i64, _i64, _p, _i64, _p, _p, _p, _i64, _p, _i64, _i64, _i64, _i64, _i, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p,
                                _p, _sz, _p]),
    "pgnn_edge_head_bwd": (_i, [_p, _i64, _i64, _p, _i64, _p, _p, _i64, _p, _i64, _i64, _p, _p, _p, _i64, _i64, _i, _p, _i64, _p, _p, _p, _p, _sz,
                                _p]),
    "pgnn_linear_bwd_data_t": (_i, [_p, _i64, _p, _p, _i64, _p, _i64, _i64, _i64, _i64, _p]),
    "pgnn_transpose_batch": (_i, [_p, _p, _p, _p, _i64, _p]),
    "pgnn_linear_bwd_weight_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "pgnn_linear_bwd_weight": (_i, [_p, _i64, _p, _i64, _p, _p, _i64, _i64, _i64, _p, _sz, _p]),
    "pgnn_linear_bwd_weight_pair": (_i, [_p, _i64, _p, _i64, _p, _p, _i64, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _i64, _i64, _p, _sz, _p]),
    "pgnn_chem_gin_layer_workspace_bytes": (_sz, [_i64, _i64]),
    "pgnn_chem_gin_layer_fwd": (_i, [_p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _f, _i, _i,
                                     _p, _p, _p, _p, _p, _p, _f, _u64, _i64, _i64, _p, _sz, _p]),
    "pgnn_chem_gin_layer_bwd": (_i, [_p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i,
                                     _p, _p, _p, _p, _p, _p, _p, _p, _f, _u64, _i64, _i64, _p, _sz, _p]),
    "pgnn_chem_gin_stack_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _i64]),
    "pgnn_chem_gin_stack_fwd": (_i, [_p, _p, _i64, _p, _i64, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _f, _u64,
                                     _i64, _i64, _p, _sz, _p]),
    "pgnn_chem_gin_stack_bwd": (_i, [_p, _i64, _p, _i64, _i64, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _f, _u64,
                                     _i64, _i64, _p, _sz, _p]),
    "pgnn_chem_lin_stack_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "pgnn_chem_lin_stack_fwd": (_i, [_i, _p, _p, _i64, _p, _i64, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _f, _u64,
                                     _i64, _i64, _p, _sz, _p]),
    "pgnn_chem_lin_stack_bwd": (_i, [_i, _p, _i64, _p, _i64, _i64, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p,
                                     _f, _u64, _i64, _i64, _p, _sz, _p]),
    "pgnn_batch_offsets": (_i, [_p, _i64, _i64, _p, _p, ctypes.c_double, _i, _p, _p, _p, _i64, _i64, _i64, _p, _p]),
    "pgnn_collate_graphs": (_i, [_p, _i64, _i64, _p, _p, _p, _p, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _p, _p, _p,
                                 _p, _p]),
    "pgnn_stack_bwd_dy_rows": (_i, [_p, _p, _i64]),
    "pgnn_neighbor_sum_bn_bwd": (_i, [_p, _i64, _p, _p, _p, _i64, _p, _i64, _p, _p, _p, _p, _i, _i, _p, _p, _i64, _i64, _p, _sz, _p, _p, _p]),
    "pgnn_collate_structure": (_i, [_p, _i64, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _p,
                                    _p, _p, _p]),
    "pgnn_mask_select": (_i, [_p, _i64, _p, _i, _p, _i64, _u64, _p, _p]),
    "pgnn_mask_edges_apply": (_i, [_p, _i64, _p, _i64, _i64, _p, _p, _p]),
    "pgnn_mask_atoms_apply": (_i, [_p, _i64, _p, _i64, _i64, _i64, _p, _p, _p]),
    "pgnn_substruct_context_plan": (_i, [_p, _i64, _i64, _p, _p, _p, _p, _p, _i64, _p, _u64, _i, _i, _i, _p, _p, _p, _p,
                                         _p, _p, _p, _p, _p]),
    "pgnn_substruct_context_fill": (_i, [_p, _i64, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _i64,
                                         _p, _i64, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "pgnn_graph_tiles_workspace_bytes": (_sz, [_i64]),
    "pgnn_graph_tiles": (_i, [_p, _p, _p, _p, _i64, _p, _p, _p, _sz, _p]),
    "pgnn_neighbor_sum_tiled": (_i, [_p, _i64, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _p]),
    "pgnn_gat_fwd": (_i, [_p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _p, _f, _p, _p, _p, _p, _p, _i64, _i64, _i64,
                          _p]),
    "pgnn_gat_bwd": (_i, [_p, _i64, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _f, _p, _p, _p, _p, _p, _p, _p,
                          _i64, _i64, _i64, _p]),
    "pgnn_segment_softmax_fwd": (_i, [_p, _p, _p, _p, _i64, _i64, _p]),
    "pgnn_segment_softmax_bwd": (_i, [_p, _p, _p, _p, _p, _i64, _i64, _p]),
    "pgnn_segment_max_fwd": (_i, [_p, _i64, _p, _p, _p, _i64, _p, _i64, _i64, _p]),
    "pgnn_segment_max_bwd": (_i, [_p, _i64, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "pgnn_debug_stream_copy": (_i, [_p, _p, _i64, _i64, _p]),
    "pgnn_debug_aggregate_profile": (_i, [_p, _i64]),
}

ABI_VERSION = 11


class GinLayer(ctypes.Structure):
    """``pgnn_gin_layer`` of include/pgnn.h (per-layer parameter / gradient pointers of the stack calls)."""

    _fields_ = [(k, _p) for k in ("emb1", "emb2", "w1", "b1", "w2", "b2", "gamma", "beta", "running_mean",
                                  "running_var")] + [("momentum", _f), ("eps", _f)] + \
               [(k, _p) for k in ("demb", "dw1", "db1", "dw2", "db2", "dgamma", "dbeta", "num_batches_tracked")]

_lib = None


class PgnnError(RuntimeError):
    pass


def load():
    """Load libpgnn.so once and bind every prototype; raise loudly if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PgnnError(
            "%s not found: build it with `python -m pretrain_gnns_amd.build` (needs hipcc, gfx950). "
            "There is no PyTorch/CPU fallback for the HIP hot path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise PgnnError("libpgnn.so lacks symbol %s (stale build?)" % name) from e
        fn.restype = res
        fn.argtypes = args
    if lib.pgnn_abi_version() != ABI_VERSION:
        raise PgnnError("libpgnn.so ABI %d != binding ABI %d" % (lib.pgnn_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().pgnn_last_error()
        raise PgnnError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr(device_index=None):
    """the current HIP stream of the current (or given) device as an integer.  The two torch._C calls are what
    ``torch.cuda.current_stream().cuda_stream`` ends in; going there directly skips ~8 us of Python per call (a train step
    asks six times)"""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device() if device_index is None else device_index)
    return (torch.cuda.current_stream() if device_index is None else torch.cuda.current_stream(device_index)).cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise PgnnError(
                "pretrain_gnns_amd runs the message-passing path on an MI355X only; got a %s tensor. "
                "Move the model and the batch to the GPU (there is no CPU fallback)." % t.device)


_status_pools = {}
_STATUS_POOL_WORDS = 1024


def status_word(dev):
    """one zeroed int32 on ``dev`` for a call's ``status`` argument.  Handed out from a pre-zeroed pool (a fresh
    ``torch.zeros(1)`` per call is a fill launch each: a dozen per train step); every word is used by one call only, so a
    batch's ``GraphStruct.check()`` still reports that batch's own count."""
    if torch.cuda.is_current_stream_capturing():  # a captured launch keeps its word for every replay: give it its own
        return torch.zeros(1, dtype=torch.int32, device=dev)
    key = (dev.type, dev.index, stream_ptr(dev.index))  # per stream: a pool is zero-filled on the stream that is current when it is made
    pool = _status_pools.get(key)
    if pool is None or pool[1] >= _STATUS_POOL_WORDS:  # (views keep an exhausted pool alive for as long as they are held)
        pool = _status_pools[key] = [torch.zeros(_STATUS_POOL_WORDS, dtype=torch.int32, device=dev), 0]
    i = pool[1]
    pool[1] = i + 1
    return pool[0][i:i + 1]

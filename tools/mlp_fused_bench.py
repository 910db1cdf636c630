"""k_mlp2p_fused (csrc/mlp_fused.hip) against the two products on planes it replaces: HIP-event time per call, forward and
backward-data, over row counts; results as JSON lines (profiles/r05/mlp_fused_ab.jsonl is a run of this).
usage: python tools/mlp_fused_bench.py [rows ...] [--iters N] [--out FILE]"""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pretrain_gnns_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")


def planes(lib, sp, mats, transpose):
    cnt = len(mats)
    bufs = []
    for w, tr in zip(mats, transpose):
        r, c = (w.size(1), w.size(0)) if tr else (w.size(0), w.size(1))
        bufs.append(torch.zeros(int(lib.pgnn_weight_planes_bytes(r, c)) // 2, dtype=torch.int16, device=DEV))
    arr = lambda vals, ty: (ty * cnt)(*vals)
    ops.check(lib.pgnn_split_weights_2p(arr([w.data_ptr() for w in mats], ctypes.c_void_p), arr([b.data_ptr() for b in bufs], ctypes.c_void_p),
                                        arr([w.size(0) for w in mats], ctypes.c_int64), arr([w.size(1) for w in mats], ctypes.c_int64),
                                        arr([int(t) for t in transpose], ctypes.c_int32), cnt, sp), "split")
    return bufs


def timed(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("rows", nargs="*", type=int, default=[262144, 65536, 32768, 16384, 6747])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default=None)
    ap.add_argument("--fused-only", action="store_true", help="time the fused launches only (ablation builds: their results are wrong)")
    a = ap.parse_args()
    lib, sp = ops.load(), ops.stream_ptr()
    k1, n1, n2 = 300, 600, 300
    torch.manual_seed(0)
    w1 = (torch.randn(n1, k1) * 0.05).to(DEV)
    w2 = (torch.randn(n2, n1) * 0.05).to(DEV)
    b1, b2 = torch.randn(n1, device=DEV) * 0.1, torch.randn(n2, device=DEV)
    p1, p2, p2t, p1t = planes(lib, sp, [w1, w2, w2, w1], [False, False, True, True])
    out = open(a.out, "a") if a.out else None
    for m in a.rows:
        x = torch.randn(m, k1, device=DEV)
        hid, y = torch.empty(m, n1, device=DEV), torch.empty(m, n2, device=DEV)
        hid_u, y_u = torch.empty(m, n1, device=DEV), torch.empty(m, n2, device=DEV)
        ham = torch.zeros(m, dtype=torch.int32, device=DEV)

        def fused_fwd():
            ops.check(lib.pgnn_mlp_fwd_2p_fused(x.data_ptr(), k1, p1.data_ptr(), b1.data_ptr(), p2.data_ptr(), b2.data_ptr(), hid.data_ptr(), n1,
                                                y.data_ptr(), n2, m, k1, n1, n2, None, sp), "fused fwd")

        def two_fwd():
            ham.zero_()
            ops.check(lib.pgnn_linear_fwd_2p(x.data_ptr(), k1, None, p1.data_ptr(), b1.data_ptr(), hid_u.data_ptr(), n1, m, k1, n1, 1, None,
                                             ham.data_ptr(), sp), "fwd 1")
            ops.check(lib.pgnn_linear_fwd_2p(hid_u.data_ptr(), n1, ham.data_ptr(), p2.data_ptr(), b2.data_ptr(), y_u.data_ptr(), n2, m, n1, n2, 0, None,
                                             None, sp), "fwd 2")

        if a.fused_only:
            dy = torch.randn(m, n2, device=DEV) * 1e-3
            dhid, dx = torch.empty(m, n1, device=DEV), torch.empty(m, k1, device=DEV)

            def fused_bwd0():
                ops.check(lib.pgnn_mlp_bwd_data_2p_fused(dy.data_ptr(), n2, p2t.data_ptr(), hid_u.data_ptr(), n1, p1t.data_ptr(), dhid.data_ptr(), n1,
                                                         dx.data_ptr(), k1, m, n2, n1, k1, sp), "fused bwd")

            two_fwd()
            rec = {"rows": m, "abl": os.environ.get("PGNN_FUSED_ABL", "0"), "fwd_fused_us": round(timed(fused_fwd, a.iters), 1),
                   "bwd_fused_us": round(timed(fused_bwd0, a.iters), 1)}
            print(json.dumps(rec), flush=True)
            if out:
                out.write(json.dumps(rec) + "\n")
                out.flush()
            continue
        t_f, t_u = timed(fused_fwd, a.iters), timed(two_fwd, a.iters)
        same_hid = bool(torch.equal(hid, hid_u))
        dy_rel = ((y - y_u).abs().max() / y_u.abs().max()).item()
        dy = torch.randn(m, n2, device=DEV) * 1e-3
        dhid, dx = torch.empty(m, n1, device=DEV), torch.empty(m, k1, device=DEV)
        dhid_u, dx_u = torch.empty(m, n1, device=DEV), torch.empty(m, k1, device=DEV)
        dam = torch.zeros(m, dtype=torch.int32, device=DEV)

        def fused_bwd():
            ops.check(lib.pgnn_mlp_bwd_data_2p_fused(dy.data_ptr(), n2, p2t.data_ptr(), hid.data_ptr(), n1, p1t.data_ptr(), dhid.data_ptr(), n1,
                                                     dx.data_ptr(), k1, m, n2, n1, k1, sp), "fused bwd")

        def two_bwd():
            dam.zero_()
            ops.check(lib.pgnn_linear_bwd_data_2p(dy.data_ptr(), n2, None, p2t.data_ptr(), hid.data_ptr(), n1, dhid_u.data_ptr(), n1, m, n1, n2,
                                                  dam.data_ptr(), sp), "bwd 1")
            ops.check(lib.pgnn_linear_bwd_data_2p(dhid_u.data_ptr(), n1, dam.data_ptr(), p1t.data_ptr(), None, 0, dx_u.data_ptr(), k1, m, k1, n1, None,
                                                  sp), "bwd 2")

        tb_f, tb_u = timed(fused_bwd, a.iters), timed(two_bwd, a.iters)
        flop = 2.0 * m * (k1 * n1 + n1 * n2)
        rec = {"rows": m, "fwd_fused_us": round(t_f, 1), "fwd_two_products_us": round(t_u, 1), "bwd_fused_us": round(tb_f, 1),
               "bwd_two_products_us": round(tb_u, 1), "fwd_fused_tflops_fp32_equiv": round(flop / t_f * 1e-6, 1),
               "bwd_fused_tflops_fp32_equiv": round(flop / tb_f * 1e-6, 1), "hid_bit_identical": same_hid,
               "dhid_bit_identical": bool(torch.equal(dhid, dhid_u)), "y_max_rel_diff": dy_rel,
               "dx_max_rel_diff": ((dx - dx_u).abs().max() / dx_u.abs().max()).item()}
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + "\n")
            out.flush()


if __name__ == "__main__":
    main()

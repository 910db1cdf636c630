"""accuracy (against float64) and speed of the two fp32 GEMM paths: PGNN_GEMM_SPLIT=0 (v_mfma_f32_16x16x4_f32) and
=1 (three-term bf16 split, six v_mfma_f32_16x16x32_bf16 products).
usage: python tools/gemm_split_check.py [rows ...]   (env PGNN_GEMM3_CFG passes through: 0 = 128x160 tiles, 1 = 64x160)"""
import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pretrain_gnns_amd import ops
rows = [int(a) for a in sys.argv[1:]] or [6747, 262144]
lib, sp, dev = ops.load(), ops.stream_ptr(), "cuda"

def timeit(fn, iters=20):
    for _ in range(3): fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / iters

def relerr(got, want):  # max over entries of |err| / (|a|.|b| row-col bound is overkill: use the rms of the exact result)
    d = (got.double() - want).abs()
    return (d.max() / want.abs().mean()).item(), (d.pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()

out = []
for m in rows:
    for (k, n) in ((300, 600), (600, 300)):
        torch.manual_seed(m + k)
        x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) * 0.05; b = torch.randn(n, device=dev)
        dy = torch.randn(m, n, device=dev)
        mask = torch.relu(torch.randn(m, k, device=dev))
        small = m <= 20000
        if small:
            x64, w64, dy64 = x.double(), w.double(), dy.double()
            y_ref = torch.relu(x64 @ w64.t() + b.double())
            dx_ref = (dy64 @ w64) * (mask > 0)
            dw_ref = dy64.t() @ x64
            db_ref = dy64.sum(0)
        rec = {"M": m, "K": k, "N": n}
        for mode in (0, 1):
            os.environ["PGNN_GEMM_SPLIT"] = str(mode)
            lib.pgnn_reload_env()
            y = torch.empty(m, n, device=dev); dx = torch.empty(m, k, device=dev)
            dw = torch.empty(n, k, device=dev); db = torch.empty(n, device=dev)
            ws = torch.empty(int(lib.pgnn_linear_bwd_weight_workspace_bytes(m, k, n)), dtype=torch.uint8, device=dev)
            fwd = lambda: ops.check(lib.pgnn_linear_fwd(x.data_ptr(), k, w.data_ptr(), b.data_ptr(), y.data_ptr(), n, m, k, n, 1, sp), "f")
            bwd = lambda: ops.check(lib.pgnn_linear_bwd_data(dy.data_ptr(), n, w.data_ptr(), mask.data_ptr(), k, dx.data_ptr(), k, m, k, n, sp), "d")
            wgt = lambda: ops.check(lib.pgnn_linear_bwd_weight(dy.data_ptr(), n, x.data_ptr(), k, dw.data_ptr(), db.data_ptr(), m, k, n, ws.data_ptr(), ws.numel(), sp), "w")
            fl = 2.0 * m * k * n
            r = {}
            for name, fn in (("fwd", fwd), ("bwd_data", bwd), ("bwd_weight", wgt)):
                t = timeit(fn)
                r[name] = {"us": round(t * 1e3, 1), "tflops": round(fl / t / 1e9, 1)}
            if small:
                fwd(); bwd(); wgt(); torch.cuda.synchronize()
                for name, got, want in (("fwd", y, y_ref), ("bwd_data", dx, dx_ref), ("bwd_weight", dw, dw_ref), ("bias_grad", db, db_ref)):
                    mx, rms = relerr(got, want)
                    r.setdefault(name, {}).update({"max_err_over_mean": mx, "rms_rel_err": rms})
            rec["split" if mode else "fp32_mfma"] = r
        out.append(rec)
        print(json.dumps(rec))
os.environ.pop("PGNN_GEMM_SPLIT", None)

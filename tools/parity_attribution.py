"""ref_bio_masking_b8/gin/grads under different arithmetics of the products: runs the reference test for that
fixture with each knob set and prints the gradient statistics it logs.  usage: python tools/parity_attribution.py"""
import json, os, subprocess, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
log = os.path.join(root, "gpurun_out", "parity_metrics.jsonl")
for tag, env in (("default (two fp16 planes fwd / bwd-data, three bf16 planes dW)", {}),
                 ("PGNN_GEMM_2P=0 (three bf16 planes everywhere)", {"PGNN_GEMM_2P": "0"}),
                 ("PGNN_GEMM_SPLIT=0 PGNN_GEMM_2P=0 (fp32 MFMA everywhere)", {"PGNN_GEMM_SPLIT": "0", "PGNN_GEMM_2P": "0"}),
                 ("PGNN_BIO_STACK=0 style: per-layer path (PGNN_STACK_CALL=0)", {"PGNN_STACK_CALL": "0"})):
    if os.path.exists(log):
        os.remove(log)
    p = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_reference.py", "-q", "-x", "-k", "bio_masking_vs_reference and b8"],
                       cwd=root, env=dict(os.environ, **env), capture_output=True, text=True)
    print("==", tag, "|", p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-300:])
    if os.path.exists(log):
        for line in open(log):
            r = json.loads(line)
            if r["test"] == "ref_bio_masking_b8/gin/grads":
                print("   hip", {k: "%.2e" % v for k, v in r["hip_vs_f64"].items()}, "| ref32", {k: "%.2e" % v for k, v in r["ref32_vs_f64"].items()})
            if r["test"] == "ref_bio_masking_b8/gin/grads/per_tensor":
                for t in r["worst_median"]:
                    print("      %-28s hip median %.2e max %.2e | ref32 median %.2e max %.2e" % (t["tensor"], t["hip"]["median"], t["hip"]["max"], t["ref32"]["median"], t["ref32"]["max"]))

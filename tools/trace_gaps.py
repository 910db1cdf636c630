"""analyse a rocprofv3 kernel trace: per-step span, GPU-busy time, idle gaps and what precedes them."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# use the Adam multi_tensor kernels as step delimiters: find repeating period via the embed kernel
marks = [i for i, e in enumerate(ev) if "k_embed_fwd" in e[2]]
print("steps found", len(marks))
if len(marks) < 6: sys.exit()
a, b = marks[-6], marks[-1]
seg = ev[a:b]
span = (seg[-1][1] - seg[0][0]) / 5 / 1e3
busy = 0; cur_end = seg[0][0]; gaps = collections.Counter(); gapn = collections.Counter()
for s, e, k in seg:
    if s > cur_end:
        gaps[prev[:50]] += s - cur_end; gapn[prev[:50]] += 1
        busy += e - s; cur_end = e
    else:
        busy += max(0, e - cur_end); cur_end = max(cur_end, e)
    prev = k
print("per step: span %.1f us, busy(union) %.1f us, idle %.1f us, launches %.0f" % (span, busy / 5 / 1e3, span - busy / 5 / 1e3, len(seg) / 5))
print("largest idle gaps by preceding kernel (us per step, count per step):")
for k, v in gaps.most_common(14):
    print("  %7.1f  %5.1f  after %s" % (v / 5 / 1e3, gapn[k] / 5, k))

"""start / duration / stream / gap timeline of ONE train step from a rocprofv3 kernel trace of tools/step_profile.py
(profiles/rNN/step_b256_timeline_*.txt).  usage: python tools/step_timeline.py <kernel_trace.csv> [step_from_end=2]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows)
marks = [i for i, e in enumerate(ev) if "k_embed_fwd" in e[3]]
a, b = marks[-back - 1], marks[-back]
seg = ev[a:b]
queues = {}
for e in ev:
    queues.setdefault(e[2], "q%d" % (len(queues) + 1))
t0 = seg[0][0]
print("one eager train step (step %d of %d in the trace), %d kernels, %.1f us from its first kernel to the next step's first (profiler attached)"
      % (len(marks) - back, len(marks), len(seg), (ev[b][0] - t0) / 1e3))
print("columns: start [us since the step's first kernel], duration [us], queue (q1 = caller's stream, others = side streams), gap to the end of the latest earlier kernel [us], kernel")
end = t0
for s, e, q, k in seg:
    k = k.replace("pgnn::(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")
    print("%8.1f %7.1f %s gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, queues[q], (s - end) / 1e3, k[:100]))
    end = max(end, e)

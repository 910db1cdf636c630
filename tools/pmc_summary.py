"""summarise rocprofv3 --pmc counter_collection CSVs: per kernel, mean counter value per dispatch."""
import csv, glob, sys, collections
pat = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(pat, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if flt and flt not in k: continue
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        v = sorted(v)
        print("   %-28s n=%-4d median %.6g  max %.6g" % (c, len(v), v[len(v)//2], v[-1]))

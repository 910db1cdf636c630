"""FETCH_SIZE / WRITE_SIZE counter_collection CSVs of one kernel (separate rocprofv3 --pmc passes) -> the traffic record bench.py
quotes as `roofline.traffic`.  gfx950: FETCH_SIZE counts 128-byte requests of wide coalesced reads at 64 B (MI355X_MICROARCH.md,
HBM section) -> x2; both counters are in KB.
usage: python tools/pmc_traffic_json.py <fetch.csv> <write.csv> <kernel substring> <nodes> <edges> <algorithmic bytes> <command> > out.json"""
import csv, json, sys


def median(path, counter, flt):
    v = sorted(float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and flt in r.get("Kernel_Name", ""))
    # counters are reported per dispatch and per XCD/SE instance: sum the instances of a dispatch
    by = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and flt in r.get("Kernel_Name", ""):
            by[r["Dispatch_Id"]] = by.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    s = sorted(by.values())
    return s[len(s) // 2], len(s)


fetch_csv, write_csv, flt, n, e, alg, cmd = sys.argv[1:8]
f, nf = median(fetch_csv, "FETCH_SIZE", flt)
w, nw = median(write_csv, "WRITE_SIZE", flt)
fb, wb = int(f * 1024 * 2), int(w * 1024)
print(json.dumps({"kernel": flt, "command": cmd, "nodes": int(n), "edges": int(e), "dispatches": [nf, nw], "FETCH_SIZE_KB_median": f, "WRITE_SIZE_KB_median": w,
                  "gfx950_fetch_correction": "x2 (MI355X_MICROARCH.md HBM: FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads)",
                  "fetch_bytes": fb, "write_bytes": wb, "hbm_bytes_per_launch": fb + wb, "algorithmic_bytes_per_launch": int(alg),
                  "ratio_to_algorithmic": round((fb + wb) / float(alg), 3)}, indent=1))

"""Is the idle gap at the step boundary (last kernel of step t -> k_embed_fwd of step t + 1) host- or device-side?  From a
`rocprofv3 --hip-trace --kernel-trace` run of tools/step_profile.py: for the last steps, when the HOST called the launch of k_embed_fwd
relative to when the device finished the previous step's last kernel and started k_embed_fwd.
usage: python tools/boundary_gap.py <kernel_trace.csv> <hip_api_trace.csv>"""
import csv, sys
k = list(csv.DictReader(open(sys.argv[1])))
a = list(csv.DictReader(open(sys.argv[2])))
kern = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Correlation_Id"])) for r in k))
api = {int(r["Correlation_Id"]): (int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in a}
embeds = [i for i, e in enumerate(kern) if "k_embed_fwd" in e[2]]
for i in embeds[-8:]:
    s, e, name, cid = kern[i]
    prev_end = max(x[1] for x in kern[max(0, i - 6):i]) if i else s
    call = api.get(cid)
    if call is None:
        continue
    print("embed starts %.1f us after the previous step's last kernel ended; its launch call was issued %.1f us BEFORE that end (negative = after), call took %.1f us"
          % ((s - prev_end) / 1e3, (prev_end - call[0]) / 1e3, (call[1] - call[0]) / 1e3))
# the API calls between the previous step's last launch call and the embed launch call, for the last boundary
i = embeds[-2]
cid = kern[i][3]
t_embed = api[cid][0]
prev_cid = kern[i - 1][3]
t_prev = api[prev_cid][0]
print("host API calls between the launch of the previous kernel and the launch of k_embed_fwd (last boundary but one):")
for r in sorted(a, key=lambda r: int(r["Start_Timestamp"])):
    t = int(r["Start_Timestamp"])
    if t_prev <= t <= t_embed:
        print("  +%.1f us  %s (%.1f us)" % ((t - t_prev) / 1e3, r["Function"], (int(r["End_Timestamp"]) - t) / 1e3))

"""Print a rocprofv3 kernel_stats.csv compactly: calls, average us, share, short kernel name."""
import csv
import sys

for row in list(csv.DictReader(open(sys.argv[1])))[: int(sys.argv[2]) if len(sys.argv) > 2 else 50]:
    name = row["Name"].replace("pgnn::(anonymous namespace)::", "").replace("void ", "")
    print("%6s %9.2f %6s  %s" % (row["Calls"], float(row["AverageNs"]) / 1e3, row["Percentage"], name[:90]))

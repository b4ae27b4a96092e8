"""Merge the per-process rocprofv3 kernel_stats CSVs of one bench run (rank process + helper processes)
into one summary: python tools/merge_kstats.py out.csv in1.csv in2.csv ..."""
import csv
import sys

out, ins = sys.argv[1], sys.argv[2:]
acc = {}
for path in ins:
    for r in csv.DictReader(open(path)):
        a = acc.setdefault(r["Name"], dict(Calls=0, Total=0.0, Min=float("inf"), Max=0.0))
        a["Calls"] += int(r["Calls"])
        a["Total"] += float(r["TotalDurationNs"])
        a["Min"] = min(a["Min"], float(r["MinNs"]))
        a["Max"] = max(a["Max"], float(r["MaxNs"]))
tot = sum(a["Total"] for a in acc.values())
with open(out, "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for name, a in sorted(acc.items(), key=lambda kv: -kv[1]["Total"]):
        w.writerow([name, a["Calls"], int(a["Total"]), a["Total"] / a["Calls"], 100.0 * a["Total"] / tot,
                    int(a["Min"]), int(a["Max"]), 0.0])
print(f"merged {len(ins)} files, {len(acc)} kernels -> {out}")

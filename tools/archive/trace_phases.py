"""Average kernel durations of an early and a late part of a rocprofv3 kernel trace (CSV)."""
import csv
import collections
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:28] for r in rows]
joins = [i for i, n in enumerate(names) if n.startswith("pp3_join")]
print("scans (pp3_join launches):", len(joins))


def part(lo, hi, tag):
    a, b = joins[lo], joins[hi]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r, n in zip(rows[a:b], names[a:b]):
        acc[n][0] += 1
        acc[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    span = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / (hi - lo) / 1e3
    busy = sum(v[1] for v in acc.values()) / (hi - lo) / 1e3
    print(f"== {tag}: {span:.1f} us of wall per scan, {busy:.1f} us of kernels per scan")
    for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"   {n:28s} {c / (hi - lo):5.2f} per scan, avg {t / c / 1e3:8.1f} us")


part(50, 150, "scans 50..150")
part(len(joins) - 150, len(joins) - 50, "late scans")

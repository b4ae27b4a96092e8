"""Per-scan GPU time by kernel from a rocprofv3 --kernel-trace --stats CSV (scan count = ppb_join launches x scans per
chain + pp3_join launches)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
import os
batched = any('ppb_join' in r['Name'] for r in rows)
key = sys.argv[3] if len(sys.argv) > 3 else ('ppb_join' if batched else 'pp3_join')
# a launch of the batched chain holds PP_BATCH scans (bench.py --pp-batch, default 4)
ns = sum(int(r['Calls']) for r in rows if key in r['Name']) * (int(os.environ.get("PP_BATCH", "4")) if batched and key == 'ppb_join' else 1)
ns += sum(int(r['Calls']) for r in rows if batched and key == 'ppb_join' and 'pp3_join' in r['Name'])
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:top]:
    print(f"{r['Name'][:58]:58s} calls/scan {int(r['Calls']) / ns:5.1f}  avg {float(r['AverageNs']) / 1e3:7.1f} us"
          f"  per-scan {float(r['TotalDurationNs']) / ns / 1e3:7.1f} us")
print(f"total {tot / ns / 1e3:.1f} us/scan over {ns} scans, {sum(int(r['Calls']) for r in rows) / ns:.1f} launches/scan")

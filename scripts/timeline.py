"""Timeline of the last render step in a rocprofv3 --kernel-trace CSV dir: per launch the start
offset (us), the duration and the idle gap since the previous kernel ended, then totals per kernel.

usage: timeline.py <dir> [max_rows]
"""
import csv
import glob
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last step ends at the last k_quantize and begins after the k_quantize before it (or at the first kernel)
ends = [i for i, r in enumerate(rows) if "k_quantize" in r["Kernel_Name"]]
last_end = ends[-1] if ends else len(rows) - 1
first = ends[-2] + 1 if len(ends) > 1 else 0
# (skip what the harness runs between two steps: torch's fills and copies up to the first kernel of the library)
while first < last_end and "rtd::" not in rows[first]["Kernel_Name"] and "fill" not in rows[first]["Kernel_Name"].lower():
    first += 1
sel = rows[first:last_end + 1]
t0 = int(sel[0]["Start_Timestamp"])
prev_end = t0
busy = defaultdict(float)
count = defaultdict(int)
gap_total = 0.0
for k, r in enumerate(sel):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void rtd::", "")[:40]
    gap = (s - prev_end) / 1e3
    if gap > 0:
        gap_total += gap
    busy[name] += (e - s) / 1e3
    count[name] += 1
    if k < limit:
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {gap:7.1f}  grid {r['Grid_Size_X']:>8}  {name}")
    prev_end = max(prev_end, e)
print(f"step: {(prev_end - t0) / 1e3:.1f} us, idle gaps {gap_total:.1f} us, {len(sel)} launches")
for name in sorted(busy, key=busy.get, reverse=True):
    print(f"  {busy[name]:9.1f} us  {count[name]:4d}x  {name}")

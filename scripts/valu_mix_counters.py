"""Which SQ counter a wave64 instruction of each class lands in: per kernel of bin/valu_mix (one class each), every counter of the
rocprofv3 --pmc passes divided by the instructions of that class the kernel's waves executed (4 launches x 8192 waves x 8192 x 16).
usage: valu_mix_counters.py <pass dir>...   (scripts/valu_mix_pmc.sh)"""
import collections
import csv
import glob
import sys

EXPECTED = 4 * 8192 * 8192 * 16  # launches x waves x iterations x instructions per iteration (cmp_cndmask2_f64 runs three per slot)
agg = collections.defaultdict(dict)
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        part = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            part[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, cs in part.items():
            agg[k].update(cs)
print(f"# counter value / instructions of the class executed ({EXPECTED} per kernel: 4 launches); values below 0.005 are left out")
for k, cs in agg.items():
    shown = {c: round(v / EXPECTED, 3) for c, v in sorted(cs.items()) if v / EXPECTED >= 0.005}
    print(f"{k:16s} " + "  ".join(f"{c.replace('SQ_INSTS_VALU_', '').replace('SQ_', '')}={v}" for c, v in shown.items()))

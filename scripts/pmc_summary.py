"""Summarise a rocprofv3 --pmc CSV: per kernel name, sum of each counter over dispatches."""
import csv, sys, glob, collections
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
        for k, cs in agg.items():
            print(k, {c: (v, n[(k, c)]) for c, v in cs.items()})

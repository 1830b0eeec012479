"""traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/profile_round.sh.
usage: make_traffic.py <round dir with pmc_fetch_<scene>/ and pmc_write_<scene>/> > traffic.json
HBM bytes per launch of the dominant kernel (k_bounce where the fused bounce runs, else k_extend /
k_extend_refill).  rocprofv3 reports FETCH_SIZE and WRITE_SIZE in kilobytes; FETCH_SIZE is doubled on
gfx950 as MI355X_MICROARCH.md prescribes (it counts 64 B per 128 B request)."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
out = {}
for scene in ("cornell", "knight", "blocks_ball"):
    per = {}
    for what, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        files = glob.glob(os.path.join(root, f"pmc_{what}_{scene}", "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        agg = collections.defaultdict(float)
        n = collections.Counter()
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            name = "k_bounce" if "k_bounce" in k else "k_extend_refill" if "k_extend_refill" in k else "k_extend" if "k_extend" in k else None
            if name:
                agg[name] += float(r["Counter_Value"])
                n[name] += 1
        per[what] = (agg, n)
    if len(per) < 2:
        continue
    # the kernel that runs every bounce (a single k_bounce next to many k_extend launches is the tail launch)
    kernel = max(("k_bounce", "k_extend", "k_extend_refill"), key=lambda name: per["fetch"][1][name])
    launches = per["fetch"][1][kernel]
    fetch = per["fetch"][0][kernel] * 1024.0 * 2.0 / launches
    write = per["write"][0][kernel] * 1024.0 / per["write"][1][kernel]
    out[scene] = {
        "kernel": kernel, "precision": "f64", "launches": launches,
        "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write,
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of `bench.py --steps 1 --warmup 0`; "
                  "counters in KB; FETCH_SIZE doubled (gfx950 counts 64 B per 128 B request); mean over the kernel's launches of the step",
    }
print(json.dumps(out, indent=1))

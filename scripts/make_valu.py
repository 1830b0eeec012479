"""valu.json from the SQ pass of scripts/profile_round.sh and the bench line of the same scene.
usage: make_valu.py <round dir with pmc_sq_<scene>/ and bench_<scene>.json> > valu.json
VALU wave-instructions per ray of the dominant kernel = SQ_INSTS_VALU summed over the kernel's launches of one step
(`bench.py --steps 1 --warmup 0`, rocprofv3 --pmc in its own run) / the rays those launches traced (the bench line's
rays per step minus the tail launch's).  bench.py multiplies it by the rays of a launch and divides by the launch's
duration, measured live, to price the kernel against the VALU issue rate (1024 SIMDs x 2.4 GHz / 4 cycles)."""
import collections
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from make_pmc import kernel_of  # noqa: E402
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
out = {}
for scene in ("cornell", "knight", "blocks_ball"):
    files = glob.glob(os.path.join(root, f"pmc_sq_{scene}", "**", "*counter_collection.csv"), recursive=True)
    bench = os.path.join(root, f"bench_{scene}.json")
    if not files or not os.path.exists(bench):
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"]
        name = kernel_of(k)  # (scripts/make_pmc.py: the tail launch is k_bounce's CAN_TAIL = true instantiation)
        if name in ("k_bounce", "k_bounce_tail", "k_extend_refill", "k_extend", "k_shade"):
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_INSTS_VALU":
                n[name] += 1
    b = json.load(open(bench))
    kernel = b["roofline"]["kernel"]
    rays = b["roofline"]["rays_per_launch"] * b["roofline"]["launches"]
    c = agg[kernel]
    if kernel == "k_bounce" and "k_bounce_tail" in agg:  # the fused kernel's tail launch is a k_bounce launch too: its rays are in `rays`
        for name, v in agg.pop("k_bounce_tail").items():
            c[name] += v
        n[kernel] += n.pop("k_bounce_tail")
    if not c or rays <= 0:
        continue
    out[scene] = {
        "kernel": kernel, "precision": "f64", "launches": n[kernel], "rays": rays,
        "valu_insts_per_ray": c["SQ_INSTS_VALU"] / rays,
        "salu_insts_per_ray": c["SQ_INSTS_SALU"] / rays,
        "wait_any_frac_of_wave_cycles": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] if c["SQ_WAVE_CYCLES"] else None,
        "other_kernels": {k: {"SQ_INSTS_VALU": v["SQ_INSTS_VALU"], "launches": n[k]} for k, v in agg.items() if k != kernel},
        "source": os.path.basename(os.path.normpath(root)) + "/pmc_summary.txt (rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU "
                  "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY, one step)",
    }
print(json.dumps(out, indent=1))

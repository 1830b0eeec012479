"""pmc.json from the rocprofv3 --pmc passes of scripts/profile_round.sh: per scene and kernel, every counter summed over the
kernel's launches of ONE step (`bench.py --steps 1 --warmup 0 --no-launch-timing`, one pass per run), the launches, the sum of the
dispatch durations where rocprofv3 reports them, and the rays each kernel traced (from the bench line the same run printed).

usage: make_pmc.py <round dir with pmc_<pass>_<scene>/ and pmc_<pass>_<scene>.json> > pmc.json

Units as rocprofv3 reports them: FETCH_SIZE / WRITE_SIZE in KiB (bench.py doubles FETCH_SIZE on gfx950 as MI355X_MICROARCH.md
prescribes), SQ_*_CYCLES and SQ_ACTIVE_INST_* / SQ_WAIT_* in quad-cycles summed over waves, SQ_THREAD_CYCLES_VALU in
lane x quad-cycles, GRBM_GUI_ACTIVE in cycles summed over the 8 XCDs."""
import collections
import csv
import glob
import json
import os
import re
import sys

SCENES = ("cornell", "knight", "blocks_ball", "globe", "two_spheres", "blue_ball")


def kernel_of(name: str):
    if "k_bounce" in name:  # k_bounce<T, STACK, MODE, CAN_TAIL, FIRST>: the tail launch is the CAN_TAIL = true instantiation
        m = re.search(r"k_bounce<[^,]+, *\d+, *\d+, *(true|false)", name) or re.search(r"k_bounceI\w+?Li\d+ELi\d+ELb([01])E", name)
        return "k_bounce_tail" if m and m.group(1) in ("true", "1") else "k_bounce"
    for k in ("k_extend_refill", "k_extend", "k_shade", "k_raygen", "k_accumulate", "k_scan_tiles", "k_quantize"):
        if k in name:
            return k
    return None


def main() -> None:
    root = sys.argv[1]
    out = {}
    for scene in SCENES:
        dirs = sorted(glob.glob(os.path.join(root, f"pmc_*_{scene}")))
        dirs = [d for d in dirs if os.path.isdir(d)]
        if not dirs:
            continue
        kernels = collections.defaultdict(lambda: {"launches": 0, "counters": collections.defaultdict(float), "duration_ns": 0.0})
        line = None
        for d in dirs:
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                continue
            seen = collections.defaultdict(set)
            this_pass = collections.defaultdict(lambda: collections.defaultdict(float))
            for r in csv.DictReader(open(files[0])):
                k = kernel_of(r["Kernel_Name"])
                if not k:
                    continue
                e = kernels[k]
                this_pass[k][r["Counter_Name"]] += float(r["Counter_Value"])
                seen[k].add(r.get("Dispatch_Id"))
                e["vgpr"] = int(r.get("VGPR_Count") or r.get("Arch_VGPR_Count") or 0)
                e["sgpr"] = int(r.get("SGPR_Count") or 0)
                e["scratch"] = int(r.get("Scratch_Size") or 0)
                e["lds"] = int(r.get("LDS_Block_Size") or 0)
            for k, cs in this_pass.items():   # a counter that several passes collect is taken from the first
                for c, v in cs.items():
                    kernels[k]["counters"].setdefault(c, v)
            for k, ids in seen.items():
                kernels[k]["launches"] = max(kernels[k]["launches"], len(ids))
            # dispatch durations (one pass is enough): Start_Timestamp / End_Timestamp per row, one row per counter
            if all(v["duration_ns"] == 0.0 for v in kernels.values()):
                done = set()
                for r in csv.DictReader(open(files[0])):
                    k = kernel_of(r["Kernel_Name"])
                    if not k or "Start_Timestamp" not in r or (k, r.get("Dispatch_Id")) in done:
                        continue
                    done.add((k, r.get("Dispatch_Id")))
                    kernels[k]["duration_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            j = d + ".json"
            if os.path.exists(j):
                for text in open(j):
                    if text.startswith("{"):
                        line = json.loads(text)
        if line is None:
            continue
        st = line["stage_counts"]
        rays_nontail = st["rays"] - st["rays_tail"]
        rays = {"k_bounce": st["rays"] if st["extend_kernel"] == 2 else 0, "k_bounce_tail": st["rays_tail"], "k_extend": rays_nontail,
                "k_extend_refill": rays_nontail, "k_shade": rays_nontail, "k_raygen": st["paths"], "k_accumulate": st["paths"]}
        if st["extend_kernel"] == 2:  # the fused kernel's own tail launch is a k_bounce launch too: priced together
            if "k_bounce_tail" in kernels:
                t = kernels.pop("k_bounce_tail")
                b = kernels["k_bounce"]
                b["launches"] += t["launches"]
                b["duration_ns"] += t["duration_ns"]
                for c, v in t["counters"].items():
                    b["counters"][c] += v
        out[scene] = {
            "precision": line["dtype"], "workload": line["config"]["workload"], "rays_per_step": st["rays"], "paths_per_step": st["paths"],
            "rays_tail": st["rays_tail"], "paths_tail": st["paths_tail"],
            "kernels": {k: {"launches": v["launches"], "rays": rays.get(k, 0), "duration_ms": v["duration_ns"] * 1e-6,
                            "vgpr": v.get("vgpr"), "sgpr": v.get("sgpr"), "scratch_bytes_per_lane": v.get("scratch"), "lds_bytes": v.get("lds"),
                            "counters": dict(sorted(v["counters"].items()))} for k, v in sorted(kernels.items())},
            "method": "rocprofv3 --pmc, three runs of `RT_ONE_LANE=1 bench.py --steps 1 --warmup 0 --no-launch-timing --scene S` (one pass per run, no "
                      "trace in the same run); counters summed over the launches of the step",
        }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

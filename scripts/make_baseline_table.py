"""The measured table of BASELINE.md from the bench lines under profiles/<round>/ (bench_<scene>.json: `bench.py --scene S`).
usage: make_baseline_table.py profiles/r02"""
import json
import os
import sys

root = sys.argv[1]
rows = [("two_spheres", "two_spheres.json5 (configs[0])"), ("cornell", "cornell.json5 (configs[1], bench default)"), ("globe", "globe.json5 (configs[2])"),
        ("blocks_ball", "blocks_ball.json5 (configs[3])"), ("knight", "knight.json5, synthetic 100 k-triangle knight.obj"),
        ("knight_1m", "knight.json5, synthetic 1.0 M-triangle knight.obj (the size configs[4] names)")]
print("| Config | image / spp / bounces | rays per image | 1x MI355X ms per image | 1x MI355X Mrays/s | CPU oracle Mrays/s (threads) | CPU s per image (from the sample) | GPU / CPU | dominant kernel: VALU-issue frac / HBM frac |")
print("|---|---|---|---|---|---|---|---|---|")
for key, label in rows:
    path = os.path.join(root, f"bench_{key}.json")
    if not os.path.exists(path):
        continue
    d = json.load(open(path))
    w = d["config"]["workload"].split(",")[0].split(" ")[1]
    rays = d["config"]["rays_per_step"]
    cpu = d.get("cpu_baseline")
    r = d["roofline"]
    issue = f"{r['frac']:.2f}" if r["bound"] == "valu_issue" else "-"
    print(f"| {label} | {w} / {d['config']['spp_image']} / 50 | {rays / 1e6:.1f} M | {d['ms_per_step']:.2f} | {d['value']:.0f} | "
          + (f"{cpu['value']:.1f} ({cpu['cores']})" if cpu else "-") + " | " + (f"{rays / cpu['value'] / 1e6:.1f}" if cpu else "-") + " | "
          + (f"{d['value'] / cpu['value']:.0f}x" if cpu else "-") + f" | {r['kernel']}: {issue} / {r['hbm']['frac']:.3f} |")

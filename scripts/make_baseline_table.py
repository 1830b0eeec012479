"""The measured tables of BASELINE.md / DESIGN.md from the bench lines and measurement files under profiles/<round>/
(bench_<scene>.json: `bench.py --scene S`; bench_<scene>_<n>spp.json: a rank's share; bench_<scene>_f32.json, f32_mode.json).

usage: make_baseline_table.py profiles/r03                      print every table
       make_baseline_table.py profiles/r03 --write FILE...      replace the blocks between
           <!-- generated:NAME --> and <!-- end generated:NAME --> in the files (NAME = main, kernels, scaling, f32)
Numbers in the documents come from here, not from typing."""
import json
import os
import re
import sys

root = sys.argv[1]


def load(name):
    path = os.path.join(root, name)
    return json.load(open(path)) if os.path.exists(path) else None


SCENES = [("two_spheres", "two_spheres.json5 (configs[0])"), ("cornell", "cornell.json5 (configs[1], bench default)"), ("globe", "globe.json5 (configs[2])"),
          ("blocks_ball", "blocks_ball.json5 (configs[3])"), ("knight", "knight.json5, synthetic 100 k-triangle knight.obj"),
          ("knight_1m", "knight.json5, synthetic 1.0 M-triangle knight.obj (the size configs[4] names)")]


def frac_text(r):
    if r["bound"] != "valu_issue":
        return "-"
    lo, hi = r.get("frac_range", [r["frac"], r["frac"]])
    return f"{lo:.2f}-{hi:.2f}"


def table_main():
    out = ["| Config | image / spp / bounces | rays per image | 1x MI355X ms per image | 1x MI355X Mrays/s | CPU oracle Mrays/s (threads) | CPU s per image (from the sample) | GPU / CPU | dominant kernel: VALU-issue frac (range) / HBM frac |",
           "|---|---|---|---|---|---|---|---|---|"]
    for key, label in SCENES:
        d = load(f"bench_{key}.json")
        if not d:
            continue
        w = d["config"]["workload"].split(",")[0].split(" ")[1]
        rays = d["config"]["rays_per_step"]
        cpu = d.get("cpu_baseline")
        r = d["roofline"]
        out.append(f"| {label} | {w} / {d['config']['spp_image']} / 50 | {rays / 1e6:.1f} M | {d['ms_per_step']:.2f} | {d['value']:.0f} | "
                   + (f"{cpu['value']:.1f} ({cpu['cores']})" if cpu else "-") + " | " + (f"{rays / cpu['value'] / 1e6:.1f}" if cpu else "-") + " | "
                   + (f"{d['value'] / cpu['value']:.0f}x" if cpu else "-") + f" | {r['kernel']}: {frac_text(r)} / {r['hbm']['frac']:.3f} |")
    return "\n".join(out)


def table_kernels():
    out = ["| scene | kernel | launches per image | ms per image (share of the step) | VALU issue frac (unnamed instructions at 2 - at 4 cycles) | lanes active in VALU instructions | wave cycles waiting (any) | waves per SIMD | L2 hit rate | HBM: compulsory bytes / 8 TB/s | PMC traffic / compulsory |",
           "|---|---|---|---|---|---|---|---|---|---|---|"]
    for key, _ in SCENES:
        d = load(f"bench_{key}.json")
        if not d:
            continue
        for k in d["roofline"].get("kernels", []):
            if "issue_frac" not in k:
                continue
            g = lambda name, fmt: (fmt % k[name]) if k.get(name) is not None else "-"
            out.append(f"| {key} | `{k['kernel']}` | {k['launches']} | {k['ms']:.2f} ({100 * k.get('share_of_instrumented_step', 0):.0f} %) | "
                       f"{k['issue_frac'][0]:.2f} - {k['issue_frac'][1]:.2f} | {g('valu_lanes_active', '%.2f')} | {g('wait_any_frac_of_wave_cycles', '%.2f')} | "
                       f"{g('waves_per_simd', '%.1f')} | {g('l2_hit_rate', '%.2f')} | {g('hbm_frac', '%.3f')} | {g('traffic_over_compulsory', '%.2f')} |")
    return "\n".join(out)


# One ncclReduce(sum, f64) of a 600 x 600 x 3 accumulation buffer to rank 0, modelled: every peer's 8.64 MB crosses its own
# xGMI link (153 GB/s, MI355X_MICROARCH.md) at once, rank 0 adds 7 buffers (60 MB at ~4 TB/s), + 30 us of launch / protocol latency.
def reduce_model_ms(width, height, n_gpus):
    if n_gpus <= 1:
        return 0.0
    nbytes = width * height * 3 * 8
    return (nbytes / 153e9 + (n_gpus - 1) * nbytes / 4e12 + 30e-6) * 1e3


def table_scaling():
    scenes = [("cornell", 600, 600), ("knight", 600, 600), ("blocks_ball", 600, 338)]
    head = "| spp of a rank (GPUs) | " + " | ".join(s for s, _, _ in scenes) + " |"
    out = [head, "|---|" + "---|" * len(scenes)]
    full = {s: load(f"bench_{s}.json") for s, _, _ in scenes}
    for spp, gpus in ((100, 1), (50, 2), (25, 4), (13, 8)):
        cells = []
        for s, w, h in scenes:
            d = full[s] if spp == 100 else load(f"bench_{s}_{spp}spp.json")
            if not d or not full[s]:
                cells.append("-")
                continue
            if spp == 100:
                cells.append(f"{d['ms_per_step']:.2f} ms")
                continue
            ideal = full[s]["ms_per_step"] / gpus
            red = reduce_model_ms(w, h, gpus)
            cells.append(f"{d['ms_per_step']:.2f} ms ({100 * ideal / d['ms_per_step']:.0f} %; with the modelled reduce of {red:.2f} ms: {100 * ideal / (d['ms_per_step'] + red):.0f} %)")
        out.append(f"| {spp} ({gpus}) | " + " | ".join(cells) + " |")
    cells = []
    for s, _, _ in scenes:
        d = load(f"bench_{s}_1spp.json")
        cells.append(f"{d['ms_per_step']:.2f} ms" if d else "-")
    out.append("| 1 (the fixed cost: one sample of every pixel) | " + " | ".join(cells) + " |")
    return "\n".join(out)


def table_f32():
    f = load("f32_mode.json") or {}
    out = ["| scene | f64 ms per image | f32 ms per image | f32 Mrays/s | golden rays: hit / miss flips, other primitive hit (rate), max relative t error | primary rays at 160 px: the same |",
           "|---|---|---|---|---|---|"]

    def cell(r):
        if not r:
            return "-"
        return f"{r['hit_miss_flips']} of {r['rays']}, {r['id_mismatches']} of {r['judged']} ({r['id_mismatch_rate']:.1e}), {r['max_rel_t_error']:.1e}"

    for s in ("cornell", "knight", "blocks_ball"):
        a, b = load(f"bench_{s}.json"), load(f"bench_{s}_f32.json")
        row = f.get(s, {})
        if a and b:
            out.append(f"| {s} | {a['ms_per_step']:.2f} | {b['ms_per_step']:.2f} | {b['value']:.0f} | {cell(row.get('golden_rays'))} | {cell(row.get('primary_160px'))} |")
    return "\n".join(out)


TABLES = {"main": table_main, "kernels": table_kernels, "scaling": table_scaling, "f32": table_f32}

if "--write" in sys.argv:
    for path in sys.argv[sys.argv.index("--write") + 1:]:
        text = open(path).read()
        for name, make in TABLES.items():
            pattern = re.compile(r"(<!-- generated:%s -->\n).*?(<!-- end generated:%s -->)" % (name, name), re.S)
            if pattern.search(text):
                text = pattern.sub(lambda m: m.group(1) + make() + "\n" + m.group(2), text)
        open(path, "w").write(text)
else:
    for name, make in TABLES.items():
        print(f"## {name}\n")
        print(make())
        print()

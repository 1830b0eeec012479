"""valu_mix.json from the table bin/valu_mix prints (scripts/micro/valu_mix.hip, run by scripts/valu_mix_pmc.sh): per instruction
class the SIMD cycles one wave64 instruction costs (kernel time by HIP events x the clock measured inside the kernel) and that clock.
usage: make_valu_mix.py <valu_mix.txt> > valu_mix.json"""
import json
import re
import sys

out = {"classes": {}, "source": "scripts/micro/valu_mix.hip on the GPU box: 16 instructions of one class per iteration, 8 waves per SIMD, "
                                "every CU; cycles = kernel time (HIP events) x in-kernel clock (s_memtime / s_memrealtime)"}
for line in open(sys.argv[1]):
    m = re.match(r"(\w+)\s+([\d.]+) ms\s+([\d.]+) ns/inst/SIMD\s+clock ([\d.]+) GHz\s+([\d.]+) cycles", line)
    if m:
        out["classes"][m.group(1)] = {"cycles": float(m.group(5)), "ns": float(m.group(3)), "clock_GHz": float(m.group(4))}
c = out["classes"]
# the clock the chip holds under dense f64 arithmetic (what the path tracer's kernels are made of)
out["f64_clock_GHz"] = round(sum(c[k]["clock_GHz"] for k in ("add_f64", "mul_f64", "fma_f64")) / 3, 3)
print(json.dumps(out, indent=1))

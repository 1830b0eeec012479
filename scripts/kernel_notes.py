#!/usr/bin/env python3
"""Prints vgpr / spill / scratch / LDS figures of every kernel of one RT_PART of kernels.hip (device-only compile).
usage: scripts/kernel_notes.py <part 0-5> [extra hipcc flags]"""
import re, subprocess, sys
part = sys.argv[1]
out = f"/tmp/kernel_notes_{part}.co"
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-parameter", "-Wno-pass-failed",
       "-DRT_EXTEND_WAVES=4", f"-DRT_PART={part}", *sys.argv[2:], "--cuda-device-only", "-c", "raytracer_amd/csrc/device/kernels.hip", "-o", out]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
co = out
head = open(out, "rb").read(24)
if head.startswith(b"__CLANG_OFFLOAD_BUNDLE__"):
    co = out + ".elf"
    subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={out}", f"--output={co}", "--unbundle"], check=True)
txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
for blk in txt.split("  - .agpr_count")[1:]:
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
    dem = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem).replace("void rtd::", "")
    print(f"{dem:60s} vgpr {g('vgpr_count'):>4s} vspill {g('vgpr_spill_count'):>4s} sspill {g('sgpr_spill_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")

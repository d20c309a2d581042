#!/usr/bin/env python3
"""Registers, LDS and occupancy of every kernel of the engine, from hipcc's -Rpass-analysis=kernel-resource-usage (no GPU needed):
python tools/kernel_resources.py  ->  one line per kernel (rocPRIM's left out)"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "longqc_amd", "csrc")
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-c", "engine.cpp", "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"], cwd=csrc, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True)
rows, cur = [], {}
for l in r.stderr.splitlines():
    m = re.search(r"remark: +([\w \[\]/]+): (\S+)", l)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k in ("Function Name", "Name"):
        if cur:
            rows.append(cur)
        cur = {"name": v}
    else:
        cur[k] = v
if cur:
    rows.append(cur)
names = [x["name"] for x in rows]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
print("%-60s %5s %5s %8s %4s %7s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "LDS"))
for x, d in zip(rows, dem):
    if "rocprim" in d:
        continue
    d = re.sub(r"^void ", "", d).split("(")[0]
    print("%-60s %5s %5s %8s %4s %7s" % (d[:60], x.get("VGPRs"), x.get("AGPRs"), x.get("ScratchSize [bytes/lane]"), x.get("Occupancy [waves/SIMD]"), x.get("LDS Size [bytes/block]")))

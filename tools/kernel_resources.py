#!/usr/bin/env python3
"""Per-kernel register / occupancy table of libgaq's device code (hipcc -Rpass-analysis=kernel-resource-usage), one line
per kernel instantiation.  No GPU needed.  python3 tools/kernel_resources.py [> profiles/rNN_kernel_resources.txt]"""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# `make report` compiles every translation unit (gaq.hip + the eight parts of gaq_inst.hip) with the resource-usage remarks on
out = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "gym_art_amd", "csrc"), "report"], stdout=subprocess.PIPE,
                     stderr=subprocess.STDOUT, text=True).stdout
rows, cur = [], None
KEYS = ("VGPRs", "AGPRs", "SGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "VGPRs Spill", "SGPRs Spill",
        "LDS Size [bytes/block]")
for ln in out.splitlines():
    if "Function Name:" in ln:
        mangled = ln.split("Function Name:")[1].split("[")[0].strip()
        name = subprocess.run(["c++filt", mangled], stdout=subprocess.PIPE, text=True).stdout.strip()
        name = name.replace("(anonymous namespace)::", "").replace("gaqk::", "").split("(")[0].replace("void ", "")
        cur = {"name": name}
        rows.append(cur)
        continue
    for key in KEYS:
        m = re.search(r"remark:\s+" + re.escape(key) + r":\s+(\d+)", ln)
        if m and cur is not None:
            cur[key] = int(m.group(1))
print("%-28s %6s %6s %6s %8s %8s %10s" % ("kernel", "VGPRs", "AGPRs", "SGPRs", "spillV", "scratch", "waves/SIMD"))
for r in rows:
    print("%-28s %6d %6d %6d %8d %8d %10d" % (r["name"], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("SGPRs", -1),
                                           r.get("VGPRs Spill", -1), r.get("ScratchSize [bytes/lane]", -1),
                                           r.get("Occupancy [waves/SIMD]", -1)))

#!/usr/bin/env python3
"""Which step / rollout kernel instantiations a GPU test run launched, against the instantiation lists of gaq_kernels.hpp.
  KERNEL_COVERAGE_OUT=gpurun_out/coverage.json python -m pytest tests -m gpu -q     (on the GPU box; tests/conftest.py writes the file)
  python3 tools/kernel_coverage.py gpurun_out/coverage.json > profiles/rNN_kernel_coverage.txt
Child processes of the tests (bench.py runs, the multi-rank rehearsals) are not in the report."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "gym_art_amd", "csrc", "gaq_kernels.hpp")).read()
BITS = [(1, "PER_ENV"), (2, "LAG"), (4, "NOISE"), (8, "GENERIC"), (16, "ALIAS"), (32, "FP32"), (64, "LITE"), (128, "PREDRAW"), (256, "NT"),
        (512, "DIAG"), (1024, "PACK"), (2048, "RZ"), (4096, "ROWS"), (8192, "CTR"), (16384, "MELL"), (32768, "SWARM"), (65536, "AUXP"),
        (131072, "ENVX"), (262144, "BIAS")]


def masks(prefix):
    out = set()
    for m in re.finditer(r"#define %s_PART\d\(X\)(.*)" % prefix, src):
        out |= {int(x) for x in re.findall(r"X\((\d+)u\)", m.group(1))}
    return out


def name(f):
    return "|".join(n for b, n in BITS if f & b) or "plain"


seen = json.load(open(sys.argv[1]))
for kind, prefix in (("step", "GAQ_STEP"), ("rollout", "GAQ_ROLL")):
    inst, got = masks(prefix), set(seen.get(kind, []))
    missing = sorted(inst - got)
    print("%s_kernel: %d instantiations, %d launched by the test run, %d not" % (kind, len(inst), len(inst & got), len(missing)))
    for f in missing:
        print("  never launched: <%d>  %s" % (f, name(f)))
    extra = sorted(got - inst)
    if extra:
        print("  launched but not in the lists (?): %s" % extra)
calls = seen.get("abi_calls")
if calls:
    never = sorted(k for k, v in calls.items() if v == 0)
    print("C ABI: %d entry points bound, %d called through the Python binding by the test run, %d not" % (len(calls), len(calls) - len(never), len(never)))
    for k in never:
        print("  never called: %s" % k)

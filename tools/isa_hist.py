#!/usr/bin/env python3
"""Static instruction mix of ONE step / rollout kernel instantiation (no GPU needed).

  python tools/isa_hist.py 1046 [more masks ...] [--roll] [--extra "-DGAQ_X=1"] [--keep DIR] [--top 25]

Compiles a one-instantiation translation unit of gaq_kernels.hpp for gfx950 to assembly (hipcc -S --cuda-device-only) and
prints, per kernel: registers / scratch / occupancy from the code object's metadata and a histogram of the instructions by class
(fp64 VALU, fp32 VALU, integer VALU, transcendental, readlane / writelane, LDS, VMEM, SALU, SMEM, waits / nops).  The counts are
STATIC (a loop body counts once; the sub-step loop is what the compiler made of it), which is what the A/B of a source change
needs; dynamic counts come from tools/pmc_case.sh on the GPU box.
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gym_art_amd", "csrc")

TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_log_", "v_exp_", "v_sin_", "v_cos_")


def classify(mn):
    if mn.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane<->sgpr"
    if mn.startswith("s_nop"):
        return "s_nop"
    if mn.startswith("s_waitcnt"):
        return "s_waitcnt"
    if mn.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if mn.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if mn.startswith(TRANS):
        return "trans_f64" if mn.endswith("f64") else "trans_f32"
    if mn.startswith("v_"):
        base = mn
        for suf in ("_e32", "_e64", "_dpp", "_sdwa"):
            if base.endswith(suf):
                base = base[: -len(suf)]
        if "f64" in base:
            return "valu_f64"
        if "f32" in base or "f16" in base:
            return "valu_f32" if "cvt" not in base else "valu_cvt"
        return "valu_int"
    return "other"


def one(mask, roll, extra, keep, top, csrc=CSRC):
    kind = "rollout_kernel" if roll else "step_kernel"
    sig = "GAQ_ROLL_SIG" if roll else "GAQ_STEP_SIG"
    src = '#include "%s/gaq_kernels.hpp"\ntemplate __global__ %s(%uu)\n' % (csrc, sig, mask)
    d = keep or tempfile.mkdtemp(prefix="isa_")
    os.makedirs(d, exist_ok=True)
    hip = os.path.join(d, "k%u.hip" % mask)
    asm = os.path.join(d, "k%u.s" % mask)
    open(hip, "w").write(src)
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", asm, hip] + extra
    subprocess.run(cmd, check=True)
    text = open(asm).read()
    # the kernel body: from its label to s_endpgm
    m = re.search(r"^(_ZN4gaqk\d+%s[^\n:]*):[^\n]*\n(.*?)\n\.Lfunc_end" % kind, text, re.S | re.M)
    if not m:
        raise SystemExit("kernel body not found in " + asm)
    body = m.group(2)
    hist = collections.Counter()
    cls = collections.Counter()
    for line in body.splitlines():
        line = line.split(";")[0].strip()
        if not line or line.endswith(":") or line.startswith("."):
            continue
        mn = line.split()[0]
        hist[mn] += 1
        cls[classify(mn)] += 1
    meta = {}
    for key in (".vgpr_count", ".sgpr_count", ".agpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size",
                ".group_segment_fixed_size"):
        mm = re.search(r"%s:\s+(\d+)" % re.escape(key), text)
        meta[key] = int(mm.group(1)) if mm else -1
    total = sum(cls.values())
    valu = sum(v for k, v in cls.items() if k.startswith(("valu", "trans", "lane")))
    print("%s<%u>  %s" % (kind, mask, " ".join(extra)))
    print("  vgpr %d agpr %d sgpr %d  sgpr_spill %d vgpr_spill %d scratch %d B" % (
        meta[".vgpr_count"], meta[".agpr_count"], meta[".sgpr_count"], meta[".sgpr_spill_count"], meta[".vgpr_spill_count"],
        meta[".private_segment_fixed_size"]))
    print("  instructions %d, VALU-issue %d" % (total, valu))
    for k, v in sorted(cls.items(), key=lambda kv: -kv[1]):
        print("    %-12s %5d" % (k, v))
    if top:
        print("  top mnemonics:")
        for k, v in hist.most_common(top):
            print("    %-28s %5d" % (k, v))
    return asm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("masks", nargs="+", type=int)
    ap.add_argument("--roll", action="store_true")
    ap.add_argument("--extra", default="")
    ap.add_argument("--keep", default=None)
    ap.add_argument("--top", type=int, default=0)
    ap.add_argument("--csrc", default=CSRC, help="another copy of gym_art_amd/csrc (what-if edits of the headers)")
    ap.add_argument("--hot", action="store_true", help="compile the in-kernel reset out (-DGAQ_PROBE_HOT=1): static counts ~ an ordinary step")
    a = ap.parse_args()
    for mk in a.masks:
        one(mk, a.roll, a.extra.split() + (["-DGAQ_PROBE_HOT=1"] if a.hot else []), a.keep, a.top, os.path.abspath(a.csrc))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Summarise rocprofv3 PC-sampling output directories: where do the waves of a kernel spend their time?

  python tools/pcs_summary.py DIR [DIR ...]

For every *pc_sampling*.csv found below a directory: the header (the format is a beta feature: printed so that a changed layout is
seen), the number of samples, the share of samples per instruction class (tools/isa_hist.py's classes) and per mnemonic, and the
hottest individual instructions.  A compact per-instruction count table is written next to the CSV (<name>.counts.txt) so that the
raw sample file need not be kept.
"""
import collections
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_hist import classify  # noqa: E402


def summarise(path):
    print("==", path, "%.1f MB" % (os.path.getsize(path) / 1e6))
    with open(path, newline="") as f:
        rd = csv.reader(f)
        header = next(rd, None)
        if not header:
            print("  empty")
            return
        print("  header:", header)
        low = [h.lower() for h in header]
        icol = next((k for k, h in enumerate(low) if h == "instruction"), None)
        ccol = next((k for k, h in enumerate(low) if "comment" in h), None)
        if icol is None:
            icol = next((k for k, h in enumerate(low) if "inst" in h), None)
        per_inst = collections.Counter()
        n = 0
        first = []
        for row in rd:
            if len(first) < 3:
                first.append(row)
            n += 1
            if icol is not None and icol < len(row):
                key = row[icol].strip()
                if ccol is not None and ccol < len(row) and row[ccol].strip():
                    key += "   ; " + row[ccol].strip()
                per_inst[key] += 1
        for r in first:
            print("  row:", r)
        print("  samples:", n)
        if not per_inst:
            return
        cls = collections.Counter()
        mn = collections.Counter()
        for key, c in per_inst.items():
            m = key.split()[0] if key.split() else "?"
            cls[classify(m)] += c
            mn[m] += c
        print("  by class:")
        for k, c in cls.most_common():
            print("    %-12s %6.2f %%" % (k, 100.0 * c / n))
        print("  by mnemonic (top 40):")
        for k, c in mn.most_common(40):
            print("    %-28s %6.2f %%" % (k, 100.0 * c / n))
        print("  hottest instructions (top 40):")
        for k, c in per_inst.most_common(40):
            print("    %6.2f %%  %s" % (100.0 * c / n, k[:150]))
        with open(path + ".counts.txt", "w") as out:
            for k, c in per_inst.most_common():
                out.write("%d\t%s\n" % (c, k))


def main():
    for d in sys.argv[1:]:
        files = glob.glob(os.path.join(d, "**", "*pc_sampling*.csv"), recursive=True)
        if not files:
            print("==", d, ": no pc_sampling csv")
        for p in sorted(files):
            summarise(p)


if __name__ == "__main__":
    main()

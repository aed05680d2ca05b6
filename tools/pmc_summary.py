#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes into the JSON that bench.py reads for `roofline.traffic`.

Usage (on the GPU box; separate passes per counter group, as MI355X_MICROARCH.md's HBM section prescribes):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT ... (optional)
    python3 tools/pmc_summary.py --kernel step_kernel --envs 1048576 --out profiles/rNN_pmc.json gpurun_out/pmc_fetch gpurun_out/pmc_write [...]

Corrections: both counters are in KiB, and on gfx950 FETCH_SIZE under-reports by 2x (guide), so fetch bytes =
FETCH_SIZE * 1024 * 2 and write bytes = WRITE_SIZE * 1024; the calibration against a known 1-GiB copy is in
profiles/r01_hbm_calib_copy_bandwidth.jsonl.
"""
import argparse
import csv
import glob
import hashlib
import json
import os
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha256():
    """hash of the kernel sources (the same function as bench.py's): bench.py reports a profile's traffic only for the
    sources it was taken from"""
    h = hashlib.sha256()
    for rel in ("gym_art_amd/csrc/gaq_kernels.hpp", "gym_art_amd/csrc/quad_core.hpp"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--kernel", default="step_kernel")
    ap.add_argument("--envs", type=int, default=1 << 20)
    ap.add_argument("--alg-bytes", type=float, default=352.0)
    ap.add_argument("--label", default="")
    ap.add_argument("--out", required=True)
    ap.add_argument("--index-key", default="",
                    help="register the summary in profiles/pmc_index.json under this key (bench.py: default_alias, default_shadow, "
                         "default_plain, c3_alias ...) together with the hash of the kernel sources")
    args = ap.parse_args()
    vals = defaultdict(list)
    for d in args.dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = defaultdict(float)
            with open(path) as f:
                for row in csv.DictReader(f):
                    if args.kernel not in row.get("Kernel_Name", ""):
                        continue
                    per_dispatch[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
            for (_, name), v in per_dispatch.items():
                vals[name].append(v)
    out = {k: {"launches": len(v), "mean": sum(v) / len(v)} for k, v in sorted(vals.items())}
    der = {"kernel": args.label or args.kernel, "N": args.envs}
    if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
        fb = out["FETCH_SIZE"]["mean"] * 1024 * 2
        wb = out["WRITE_SIZE"]["mean"] * 1024
        der.update(fetch_bytes_per_launch=fb, write_bytes_per_launch=wb, traffic_bytes_per_launch=fb + wb,
                   fetch_bytes_per_env_step=fb / args.envs, write_bytes_per_env_step=wb / args.envs,
                   traffic_bytes_per_env_step=(fb + wb) / args.envs, algorithmic_bytes_per_env_step=args.alg_bytes)
    if "SQ_WAVES" in out and out["SQ_WAVES"]["mean"]:
        w = out["SQ_WAVES"]["mean"]
        if "SQ_INSTS_VALU" in out:
            der["valu_insts_per_wave"] = out["SQ_INSTS_VALU"]["mean"] / w
        if "SQ_WAVE_CYCLES" in out:
            der["wave_cycles_per_wave(quad-cycles x4)"] = out["SQ_WAVE_CYCLES"]["mean"] * 4 / w
            for k, name in (("SQ_WAIT_ANY", "wait_any_fraction"), ("SQ_WAIT_INST_ANY", "wait_inst_any_fraction")):
                if k in out:
                    der[name] = out[k]["mean"] / out["SQ_WAVE_CYCLES"]["mean"]
        if "SQ_LDS_BANK_CONFLICT" in out:
            der["lds_bank_conflict"] = out["SQ_LDS_BANK_CONFLICT"]["mean"]
    der["note"] = "FETCH_SIZE x2 (gfx950), WRITE_SIZE exact, both in KiB; separate --pmc passes (tools/pmc_summary.py)"
    der["source_sha256"] = source_sha256()
    out["_derived"] = der
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    if args.index_key:
        idx_path = os.path.join(os.path.dirname(os.path.abspath(args.out)), "pmc_index.json")   # copy both into profiles/
        try:
            with open(idx_path) as f:
                idx = json.load(f)
        except Exception:
            idx = {}
        idx[args.index_key] = {"file": os.path.basename(args.out), "source_sha256": der["source_sha256"]}
        with open(idx_path, "w") as f:
            json.dump(idx, f, indent=1, sort_keys=True)
    print(json.dumps(der, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc passes (counter_collection.csv files) into one JSON: per kernel, the mean of every
counter per dispatch plus the derived figures DESIGN.md quotes (MFMA-busy share, parked share, LDS bank-conflict
ratio, L2->CU request bytes).  Counters that rocprofv3 reports per XCD / per instance are summed per dispatch.

    python tools/pmc_summary.py OUT.json DIR [DIR ...]
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import pretty  # noqa: E402


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    # kernel -> counter -> dispatch id -> summed value
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = pretty(row["Kernel_Name"])
                acc[k][row["Counter_Name"]][(f, row["Dispatch_Id"])] += float(row["Counter_Value"])
    res = {}
    for k, counters in acc.items():
        if "qvc" not in k and "<" not in k and "kernel" not in k:
            continue
        c = {name: sum(v.values()) / len(v) for name, v in counters.items()}
        c["dispatches"] = max(len(v) for v in counters.values())
        d = {}
        if c.get("SQ_BUSY_CYCLES") and c.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None and c.get("GRBM_GUI_ACTIVE"):
            # MFMA_BUSY counts cycles summed over the 1024 SIMDs; GUI_ACTIVE is summed over the 8 XCDs
            d["mfma_busy_share_of_wall"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        if c.get("SQ_WAVE_CYCLES"):
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if n in c:
                    d[n.lower() + "_share_of_wave_cycles"] = c[n] / c["SQ_WAVE_CYCLES"]
        if c.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_ratio"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
        if "TCP_TCC_READ_REQ" in c:
            d["l2_to_cu_read_bytes_at_64B_per_req"] = c["TCP_TCC_READ_REQ"] * 64.0
        if c.get("TCP_TCC_READ_REQ") and c.get("TCP_TCC_READ_REQ_LATENCY"):
            d["mean_l2_read_latency_cycles"] = c["TCP_TCC_READ_REQ_LATENCY"] / c["TCP_TCC_READ_REQ"]
        if c.get("TCC_HIT") is not None and c.get("TCC_MISS") is not None and (c["TCC_HIT"] + c["TCC_MISS"]) > 0:
            d["l2_hit_rate"] = c["TCC_HIT"] / (c["TCC_HIT"] + c["TCC_MISS"])
        res[k] = {"counters_per_dispatch": c, "derived": d}
    try:
        head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        head = ""
    json.dump({"git_head": head, "kernels": res}, open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(res.items()):
        print(k, json.dumps(v["derived"]))


if __name__ == "__main__":
    main()

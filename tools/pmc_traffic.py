#!/usr/bin/env python3
"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/r03_traffic.json.

HBM bytes per launch, per kernel, as MI355X_MICROARCH.md (HBM section) prescribes: separate passes for
FETCH_SIZE and WRITE_SIZE, both in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced
reads, so it is doubled; WRITE_SIZE is taken as is.  (Narrower access widths are uncalibrated.)

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r03_traffic.json
"""
import collections
import csv
import glob
import json
import re
import sys


def load(d, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(d + "/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                tot[row["Kernel_Name"]] += float(row["Counter_Value"])
                n[row["Kernel_Name"]] += 1
    return tot, n


EPI = {'1': 'gau', '3': 'smp'}   # EpiKind (qvc_plan.h)


def pretty(name):
    m = re.search(r"rbchain_kernelI(DF16_|DF16b)Li(\d+)ELi(\d+)ELi(\d+)E(DF16_)?", name)
    if m:
        t = "f16" if m.group(1) == "DF16_" else ("bf16x" if m.group(5) else "bf16")
        return f"rbchain<{t},MF{m.group(2)},NF{m.group(3)},WM{m.group(4)}>"
    m = re.search(r"wn_stack2_kernelI(DF16_|DF16b)Li(\d+)ELi(\d+)ELi(\d+)E", name)
    if m:
        return f"wn_stack2<{'f16' if m.group(1) == 'DF16_' else 'bf16'},W{2 * int(m.group(2))},L4{',pre+post' if m.group(4) != '0' else ''}>"
    m = re.search(r"rbpair_kernelI(DF16_|DF16b)Li(\d+)ELi(\d+)ELi(\d+)ELi\d+E(DF16_)?", name)
    if m:   # 6th template argument = stream type: spelled out (DF16_) only when it differs from the operand type
        t = "f16" if m.group(1) == "DF16_" else ("bf16x" if m.group(5) else "bf16")
        return f"rbpair<{t},MF{m.group(2)},NF{m.group(3)},WM{m.group(4)}>"
    m = re.search(r"conv_mfma_kernelI(DF16_|DF16b)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name)
    if m:
        return f"conv<{'f16' if m.group(1) == 'DF16_' else 'bf16'},MF{m.group(2)},NF{m.group(3)},WM{m.group(4)},{EPI.get(m.group(5), 'std')}>"
    m = re.search(r"post_tail_kernelI(DF16_|DF16b)Li(\d+)E", name)
    if m:
        return f"post_tail<{'f16' if m.group(1) == 'DF16_' else 'bf16'}>"
    m = re.search(r"wn_stack_kernelI(DF16_|DF16b)Li(\d+)ELi(\d+)ELi(\d+)E", name)
    if m:
        return f"wn_stack<{'f16' if m.group(1) == 'DF16_' else 'bf16'},W{m.group(4)},L4{',pre+post' if m.group(3) != '0' else ''}>"
    m = re.search(r"wn_layer_kernelI(DF16_|DF16b)Li(\d+)ELb(\d)ELi(\d+)E", name)
    if m:
        return f"wn_layer<{'f16' if m.group(1) == 'DF16_' else 'bf16'},W{m.group(4)},NF{m.group(2)}{',last' if m.group(3) == '1' else ''}>"
    return re.sub(r"\(.*", "", name).replace("qvc::", "")


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    f, nf = load(fetch_dir, "FETCH_SIZE")
    w, nw = load(write_dir, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        if "qvc" not in k:
            continue
        launches = max(nf.get(k, 0), nw.get(k, 0), 1)
        fetch = f.get(k, 0.0) / max(nf.get(k, 1), 1) * 1024.0
        write = w.get(k, 0.0) / max(nw.get(k, 1), 1) * 1024.0
        res[pretty(k)] = {"launches_profiled": launches, "fetch_bytes_raw_per_launch": fetch,
                          "write_bytes_per_launch": write, "hbm_bytes_per_launch": 2.0 * fetch + write,
                          "note": "FETCH_SIZE doubled (gfx950 wide-read correction), WRITE_SIZE as reported"}
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from bench import kernel_source_sha1
    json.dump({"kernel_source_sha1": kernel_source_sha1(), "kernels": res}, open(out, "w"), indent=1, sort_keys=True)
    for k, v in res.items():
        print(f"{k:40s} {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch")


if __name__ == "__main__":
    main()

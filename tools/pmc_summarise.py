#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv output into profiles/traffic.json
(bytes per launch for the fused kernels).  Units per MI355X_MICROARCH.md (HBM section):
counters are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide (16 B/lane)
coalesced streaming read, so the corrected read figure doubles it (upper bound for kernels that
also issue 4 B/lane loads, which are uncalibrated)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fetch_dir, write_dir, tag = sys.argv[1], sys.argv[2], sys.argv[3]


def per_kernel(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"\b(k_[A-Za-z0-9_]+)", row["Kernel_Name"])
            if m and row.get("Counter_Name") == counter:
                name = m.group(1)
                if name == "k_step_rows":       # template <FP, HP, H2P, FUNC, ...>: functional state copy?
                    t = re.search(r"k_step_rows<(\d+), \d+, \d+, (true|false)", row["Kernel_Name"])
                    if t and t.group(2) == "true":
                        name = "k_step_rows_functional"
                    if t and t.group(1) != "32":      # cfg3 (F = 64)
                        name += "_f" + t.group(1)
                if name == "k_euclid_mfma2":     # template <FT, TAIL>: 0 selector alone, 1 + cached step, 2 + steady-state step
                    t = re.search(r"k_euclid_mfma2<\d+, (\d+|true|false)>", row["Kernel_Name"])
                    tail = {"false": "0", "true": "1"}.get(t.group(1), t.group(1)) if t else "0"
                    if tail != "0":
                        name += "_tail" + tail
                if name == "k_learned_select":   # template <MODE, TAIL, EX>: TAIL 2 = the steady-state step
                    t = re.search(r"k_learned_select<\d+, (\d+|true|false)", row["Kernel_Name"])
                    if t and t.group(1) == "2":
                        name += "_steady"
                if name == "k_bptt_rows":        # template <FP, HP, H2P, MODE>: 2 = pass A of the LearnedEdge backward
                    t = re.search(r"k_bptt_rows<(\d+), \d+, \d+, (\d+)", row["Kernel_Name"])
                    if t and t.group(2) == "2":
                        name = "k_bptt_rows_learned"
                    elif t and t.group(1) != "32":
                        name += "_f" + t.group(1)
                acc[name].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


fetch, nf = per_kernel(fetch_dir, "FETCH_SIZE")
write, _ = per_kernel(write_dir, "WRITE_SIZE")
out = {}
for k in sorted(fetch):
    short = k
    f_kib, w_kib = fetch[k], write.get(k, 0.0)
    out.setdefault(short, {
        "kernel": k, "launches": nf[k],
        "FETCH_SIZE_KiB": f_kib, "WRITE_SIZE_KiB": w_kib,
        "bytes_raw": (f_kib + w_kib) * 1024,
        "bytes_corrected": (2 * f_kib + w_kib) * 1024,
        "note": "corrected = 2*FETCH_SIZE + WRITE_SIZE (gfx950 counts 128-B read requests as 64 B)"})
path = os.path.join(ROOT, "profiles", f"{tag}_traffic_detail.json")
json.dump(out, open(path, "w"), indent=1)
if "--keep-traffic-json" not in sys.argv:      # (a side profile: profiles/traffic.json stays the bench's)
    flat = {k: v["bytes_corrected"] for k, v in out.items()}
    json.dump(flat, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
for k, v in out.items():
    print(f"{k:24s} launches={v['launches']:5d} fetch={v['FETCH_SIZE_KiB'] / 1024:8.2f} MiB "
          f"write={v['WRITE_SIZE_KiB'] / 1024:8.2f} MiB corrected={v['bytes_corrected'] / 1e6:8.2f} MB")

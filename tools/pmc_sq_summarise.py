#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc SQ_* passes (MFMA utilisation, LDS conflicts, issue stalls) per kernel into
profiles/<tag>_mfma_util.json.

  python3 tools/pmc_sq_summarise.py <tag> <pmc dir> [<pmc dir> ...]

Every directory is one rocprofv3 pass (its *counter_collection.csv rows: one per dispatch and counter, values summed
over the chip's shader engines / SIMDs as the tool reports them; *kernel_trace.csv of the same pass gives each
dispatch's begin / end).  Per kernel (template arguments kept for the MFMA kernels): mean counter values per launch,
the mean duration, and the derived figures
  mfma_busy_frac  = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)  - the matrix pipes' busy cycles over the cycles
                    the CUs that held waves were busy (gfx94x's MfmaUtil formula with the kernel's own busy CUs instead of
                    all of them; checked on k_euclid_mfma2: SQ_INSTS_MFMA = 262 144 = 256 workgroups x 16 waves x 64, busy
                    cycles = 64 per v_mfma_f32_32x32x2_f32, SQ_BUSY_CU_CYCLES / 256 = the kernel's duration in cycles)
  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE   (MI355X_MICROARCH.md: extra cycles / all LDS-array cycles)
  wait_inst_lds_frac, wait_inst_any_frac, wait_any_frac = SQ_WAIT_* / SQ_WAVE_CYCLES (quad-cycles both)
Units: MI355X_MICROARCH.md ("s_memtime tick vs SQ PMC units"): SQ_VALU_MFMA_BUSY_CYCLES counts cycles,
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* quad-cycles."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, dirs = sys.argv[1], sys.argv[2:]
N_CU, N_SIMD = 256, 4


def short(name):
    m = re.search(r"\b(k_[A-Za-z0-9_]+)(<[^>(]*>)?", name)
    if not m:
        return None
    return m.group(1) + (m.group(2) or "")


vals = defaultdict(lambda: defaultdict(list))      # kernel -> counter -> [per-dispatch values]
durs = defaultdict(list)
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if k is None:
                continue
            key = (row.get("Dispatch_Id"), row["Counter_Name"])
            per_dispatch[key] += float(row["Counter_Value"])     # (one row per dimension instance on some versions)
            names[row.get("Dispatch_Id")] = k
        for (disp, ctr), v in per_dispatch.items():
            vals[names[disp]][ctr].append(v)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if k is not None:
                durs[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)

out = {}
for k in sorted(vals):
    c = {ctr: sum(v) / len(v) for ctr, v in vals[k].items()}
    ent = {"launches": max(len(v) for v in vals[k].values()), "counters_per_launch": {a: round(b, 1) for a, b in sorted(c.items())}}
    if durs.get(k):
        ent["avg_us_under_pmc"] = round(sum(durs[k]) / len(durs[k]), 3)
    mf, busy_cu, busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES"), c.get("SQ_BUSY_CU_CYCLES"), c.get("SQ_BUSY_CYCLES")
    if mf is not None and busy_cu:
        ent["mfma_busy_frac"] = round(mf / (N_SIMD * busy_cu), 4)
    if mf is not None and c.get("SQ_INSTS_MFMA"):
        ent["mfma_cycles_per_inst"] = round(mf / c["SQ_INSTS_MFMA"], 2)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        ent["lds_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for ctr, key in (("SQ_WAIT_INST_LDS", "wait_inst_lds_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_any_frac"),
                         ("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_ACTIVE_INST_ANY", "active_inst_any_frac")):
            if ctr in c:
                ent[key] = round(c[ctr] / wc, 4)
    out[k] = ent
path = os.path.join(ROOT, "profiles", f"{tag}_mfma_util.json")
json.dump(out, open(path, "w"), indent=1)
for k, e in out.items():
    if "mfma_busy_frac" in e:
        print(f"{k:56s} n={e['launches']:5d} us={e.get('avg_us_under_pmc', 0):9.2f} mfma_busy={e.get('mfma_busy_frac')} "
              f"cyc/mfma={e.get('mfma_cycles_per_inst')} lds_conf={e.get('lds_conflict_frac')} wait_lds={e.get('wait_inst_lds_frac')}")

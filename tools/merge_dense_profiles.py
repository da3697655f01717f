#!/usr/bin/env python3
"""Fold the dense-regime side profile (tools/collect_profiles.sh part `dense`: profiles/<tag>_dense_traffic_detail.json,
profiles/<tag>_dense_mfma_util.json) into the two lookup files bench.py reads - profiles/traffic.json (bytes per launch:
k_step_colcache, k_step_rows_dense = the general live-row kernel with every row live, k_bptt_rows_dense) and
profiles/mfma_util.json (per kernel name).   python3 tools/merge_dense_profiles.py r06"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
P = lambda n: os.path.join(ROOT, "profiles", n)
tr = json.load(open(P("traffic.json")))
det = json.load(open(P(f"{tag}_dense_traffic_detail.json")))
for src, dst in (("k_step_colcache", "k_step_colcache"), ("k_step_colcache8", "k_step_colcache"),
                 ("k_step_rows", "k_step_rows_dense"), ("k_bptt_rows", "k_bptt_rows_dense")):
    if src in det:
        tr[dst] = det[src]["bytes_corrected"]
json.dump(dict(sorted(tr.items())), open(P("traffic.json"), "w"), indent=1)
mu = json.load(open(P("mfma_util.json")))
for k, v in json.load(open(P(f"{tag}_dense_mfma_util.json"))).items():
    if k.startswith(("k_step_colcache", "k_step_rows<", "k_bptt_rows<32, 32, 32, 4>")):
        mu[k] = v
json.dump(mu, open(P("mfma_util.json"), "w"), indent=1)
print("merged", [k for k in tr if "dense" in k or "colcache" in k])

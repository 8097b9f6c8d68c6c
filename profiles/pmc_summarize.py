"""Aggregates rocprofv3 --pmc counter_collection.csv files (one directory per pass) into a per-kernel summary.

    python profiles/pmc_summarize.py gpurun_out/pmc1 gpurun_out/pmc2 ... > profiles/rNN_pmc_summary.json

Counter passes are collected separately (`rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python bench.py
--steps 3 --warmup 1 --no-cpu-baseline`), as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE cannot share a pass).
gfx950 corrections applied here: FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE under-reports wide coalesced reads by 2x
(doubled in `hbm_bytes_per_launch`); GRBM_GUI_ACTIVE and the SQ_* counters are summed over the 8 XCDs / 1024 SIMDs.
"""
import collections
import csv
import glob
import json
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in agg.items():
    if not k.startswith("k_"):
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    e = {"launches_sampled": max(len(v) for v in cs.values()), "mean_per_launch": m}
    if m.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0                       # per-XCD active cycles
        e["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)   # 256 CUs x 4 SIMDs
        e["kernel_cycles"] = cyc
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e["hbm_bytes_per_launch"] = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
    if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
        e["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    out[k] = e
# provenance: bench.py nulls `traffic` / `mfma_busy_frac` when the kernels have changed since these counters were collected
import os
import subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench          # the fingerprint bench.py compares against: csrc/ with comments and blank lines excluded
try:
    head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or os.environ.get("GCDM_GIT_HEAD")
except Exception:
    head = os.environ.get("GCDM_GIT_HEAD")
out["_meta"] = {"csrc_sha16": bench.csrc_sha16(), "git_head": head, "note": "git_head = the commit checked out when the counters were summarised "
                "(on the GPU box there is no .git: the summary is then re-stamped in the build container from GCDM_GIT_HEAD)"}
json.dump(out, sys.stdout, indent=1)

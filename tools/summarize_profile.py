"""Condense a gpurun_out/prof_<tag>/ directory (tools/profile_bench.sh) into profiles/<tag>_*:
the rocprofv3 kernel-stats CSV verbatim plus a JSON with per-kernel averages of the PMC counters and the
HBM traffic per launch, corrected as /opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes
(FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950 -> doubled; both counters are in KiB)."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out_stem = sys.argv[2] if len(sys.argv) > 2 else f"{tag}_bench"  # profiles/<out_stem>_{kernel_stats.csv,pmc_summary.json}
src = os.path.join("gpurun_out", f"prof_{tag}")
os.makedirs("profiles", exist_ok=True)
shutil.copy(os.path.join(src, "trace", "ac_kernel_stats.csv"), os.path.join("profiles", f"{out_stem}_kernel_stats.csv"))
summary = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_mfma", "pmc_wait"):
    path = os.path.join(src, sub, "ac_counter_collection.csv")
    if not os.path.exists(path):
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if "at::native" in k or "rocclr" in k:
            continue
        e = summary.setdefault(k, {})
        for c, v in d.items():
            e[c] = sum(v) / len(v)
            e["launches_" + sub] = len(v)
for k, e in summary.items():
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0
json.dump(summary, open(os.path.join("profiles", f"{out_stem}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: v.get("hbm_bytes_per_launch") for k, v in summary.items()}, indent=1))

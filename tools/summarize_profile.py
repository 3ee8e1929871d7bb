#!/usr/bin/env python3
"""Summaries of one profiles/collect.sh run: python tools/summarize_profile.py <gpurun_out dir> <tag>
-> <tag>_kernel_stats.csv (names shortened) and <tag>_pmc_hbm.json (FETCH_SIZE / WRITE_SIZE per launch, matrix-core utilisation per kernel)."""
import collections
import csv
import glob
import json
import re
import sys

out, tag = sys.argv[1], sys.argv[2]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); return re.sub(r"\(.*", "", n)
stats = glob.glob(f"{out}/{tag}_stats/**/*kernel_stats.csv", recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(f"{out}/{tag}_kernel_stats.csv", "w") as f:
        w = csv.writer(f); w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct"])
        for r in rows: w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
    trace_avg = {short(r["Name"]): float(r["AverageNs"]) * 1e-6 for r in rows}      # kernel -> average ms under the tracer (the running step)
else:
    trace_avg = {}
res = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of 'python bench.py --steps 3 --warmup 2 "
               "--no-cpu-baseline --no-variants --no-profile'; KiB per launch as reported by the counters; gfx950: FETCH_SIZE reports half of a wide "
               "coalesced read stream (MI355X_MICROARCH.md, HBM section) -> corrected read bytes = 2 * FETCH_SIZE KiB * 1024", "kernels": {}}
for key, pat in (("fetch", "fetch"), ("write", "write")):
    fs = glob.glob(f"{out}/{tag}_{pat}/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        k = short(r["Kernel_Name"]); agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        res["kernels"].setdefault(k, {})[f"{key}_kib_avg"] = v / n; res["kernels"][k]["launches"] = n
fs = glob.glob(f"{out}/{tag}_mfma/**/*counter_collection.csv", recursive=True)
if fs:
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = short(r["Kernel_Name"]); agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k, v in agg.items():
        e = res["kernels"].setdefault(k, {}); e.setdefault("launches", len(disp[k]))
        e["mfma_busy_cycles_avg"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / len(disp[k])
        e["gui_active_cycles_avg"] = v.get("GRBM_GUI_ACTIVE", 0.0) / len(disp[k])
        e["mfma_mops_f16_avg"] = v.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) / len(disp[k])
        if v.get("GRBM_GUI_ACTIVE"):
            e["mfma_util"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (128.0 * v["GRBM_GUI_ACTIVE"])
    res["note"] += "; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES (summed over all SIMDs) / (128 SIMDs per XCD x GRBM_GUI_ACTIVE, which is reported summed over the 8 XCDs: checked on creff_rr_kernel, 43.25 M MFMAs x 16 cycles over a 2.9 ms launch), separate --pmc pass, dispatches serialised by the profiler"
conv = [v for k, v in res["kernels"].items() if k.startswith("conv_igemm_kernel") or k.startswith("gemm_x3_kernel") or k.startswith("conv3x3_patch_kernel") or k.startswith("conv16")]
if conv:
    n = sum(v["launches"] for v in conv)
    fetch = sum(v.get("fetch_kib_avg", 0) * v["launches"] for v in conv) / n; write = sum(v.get("write_kib_avg", 0) * v["launches"] for v in conv) / n
    res["conv_all_tiles"] = {"launches": n, "fetch_kib_avg": fetch, "write_kib_avg": write, "hbm_bytes_per_launch": (2 * fetch + write) * 1024}
    busy = sum(v.get("mfma_busy_cycles_avg", 0) * v["launches"] for v in conv); act = sum(v.get("gui_active_cycles_avg", 0) * v["launches"] for v in conv)
    if act:
        res["conv_all_tiles"]["mfma_util"] = busy / (128.0 * act)
for k, ms in trace_avg.items():
    if k in res["kernels"]:
        res["kernels"][k]["avg_ms"] = ms      # rocprofv3 --kernel-trace --stats average of the kernel in the same bench command
json.dump(res, open(f"{out}/{tag}_pmc_hbm.json", "w"), indent=1)
print(open(f"{out}/{tag}_bench.json").read()[:600])

#!/bin/bash
# Collects the round's evidence on the MI355X box (run from the repo root through gpurun):
#   bash profiles/collect.sh <tag> [config]   -> gpurun_out/<tag>_*   (copy the summaries you keep into profiles/)
# 1. bench.py with the CPU baseline (the JSON line)           2. rocprofv3 --kernel-trace --stats of the same command
# 3./4. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (no trace domains besides --kernel-trace)
# 5. rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE: matrix-core utilisation per kernel
#    = MFMA busy cycles (summed over the SIMDs) / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE of the dispatch)
# The summary <tag>_pmc_hbm.json is what bench.py reads as profiles/traffic_<config>.json.
# The conv plans tuned in step 1 are reloaded (ARSEG_CONV_PLAN_FILE) so that the profiled runs contain no trial launches.
TAG=${1:-r01}
CFG=${2:-psp}          # bench.py --config (psp = the headline workload)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export ARSEG_CONV_PLAN_FILE=$OUT/${TAG}_plans.json
rm -f $ARSEG_CONV_PLAN_FILE
cd $R
python bench.py --config $CFG --steps 9 --warmup 3 --conv-layers $OUT/${TAG}_conv_layers.json 2> $OUT/${TAG}_bench.err | tail -1 > $OUT/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --config $CFG --steps 3 --warmup 2 --no-cpu-baseline --no-variants"
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o s --output-format csv -- $BENCH > $OUT/${TAG}_bench_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_fetch -o f --output-format csv -- $BENCH --no-profile > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_write -o w --output-format csv -- $BENCH --no-profile > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE -d $OUT/${TAG}_mfma -o m --output-format csv -- $BENCH --no-profile > /dev/null 2>&1
python - <<PY
import csv, glob, json, collections, re
out = "$OUT"; tag = "$TAG"
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); return re.sub(r"\(.*", "", n)
stats = glob.glob(f"{out}/{tag}_stats/**/*kernel_stats.csv", recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(f"{out}/{tag}_kernel_stats.csv", "w") as f:
        w = csv.writer(f); w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct"])
        for r in rows: w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
res = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 3 --warmup 2 "
               "--no-cpu-baseline --no-profile`; KiB per launch as reported by the counters; gfx950: FETCH_SIZE reports half of a wide "
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
            e["mfma_util"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * v["GRBM_GUI_ACTIVE"])
    res["note"] += "; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE), both summed over the kernel's dispatches (separate --pmc pass)"
conv = [v for k, v in res["kernels"].items() if k.startswith("conv_igemm_kernel") or k.startswith("conv3x3_patch_kernel") or k.startswith("conv16")]
if conv:
    n = sum(v["launches"] for v in conv)
    fetch = sum(v.get("fetch_kib_avg", 0) * v["launches"] for v in conv) / n; write = sum(v.get("write_kib_avg", 0) * v["launches"] for v in conv) / n
    res["conv_all_tiles"] = {"launches": n, "fetch_kib_avg": fetch, "write_kib_avg": write, "hbm_bytes_per_launch": (2 * fetch + write) * 1024}
    busy = sum(v.get("mfma_busy_cycles_avg", 0) * v["launches"] for v in conv); act = sum(v.get("gui_active_cycles_avg", 0) * v["launches"] for v in conv)
    if act:
        res["conv_all_tiles"]["mfma_util"] = busy / (1024.0 * act)
json.dump(res, open(f"{out}/{tag}_pmc_hbm.json", "w"), indent=1)
print(open(f"{out}/{tag}_bench.json").read()[:600])
PY

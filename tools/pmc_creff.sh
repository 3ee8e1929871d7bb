#!/bin/bash
# SQ counters of the CReFF kernels alone (tools/bench_creff.py --skip-old): three --pmc passes (8 SQ slots each), kernel trace only.
#   bash tools/pmc_creff.sh <tag> [extra bench_creff.py args]  -> gpurun_out/<tag>_pmc_creff.json
# MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES); VALU issue share = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (quad-cycles).
TAG=${1:-r03}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
  case $i in
    1) C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES";;
    2) C="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU";;
    3) C="SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES";;
  esac
  rocprofv3 --pmc $C --kernel-trace -d $OUT/${TAG}_pmc_creff/p$i -o out --output-format csv -- python $R/tools/bench_creff.py --skip-old --iters 3 "$@" > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections, json
res = {}
for i in (1, 2, 3):
    fs = glob.glob("$OUT/${TAG}_pmc_creff/p%d/**/*counter_collection.csv" % i, recursive=True)
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); seen[k].add(r["Dispatch_Id"])
    for k, v in agg.items():
        if "creff" in k:
            n = len(seen[k])
            res.setdefault(k, {"launches": n}).update({a: round(b / n) for a, b in v.items()})
for k, v in res.items():
    if "SQ_BUSY_CU_CYCLES" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        v["mfma_util_of_busy_cu_cycles"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / max(v["SQ_BUSY_CU_CYCLES"], 1)
    if "SQ_WAVE_CYCLES" in v:
        for a in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if a in v:
                v[a + "/WAVE_CYCLES"] = v[a] / v["SQ_WAVE_CYCLES"]
json.dump({"note": "rocprofv3 --pmc (3 passes) --kernel-trace of tools/bench_creff.py --skip-old; per launch averages", "kernels": res},
          open("$OUT/${TAG}_pmc_creff.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY

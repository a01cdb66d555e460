#!/bin/bash
# SQ / GRBM / TCC counters of one command's kernels, summed per kernel name:  bash tools/pmc_kernel.sh <outdir under gpurun_out> <kernel substring> -- <python args...>
set -u
out=$1; kern=$2; shift 3
repo=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$repo/$out"; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$repo/$out/time" -o r -- python3 "$@" > "$repo/$out/time.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$repo/$out/sq" -o r -- python3 "$@" > "$repo/$out/sq.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$repo/$out/fetch" -o r -- python3 "$@" > "$repo/$out/fetch.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$repo/$out/write" -o r -- python3 "$@" > "$repo/$out/write.log" 2>&1
python3 - "$repo/$out" "$kern" <<'PY'
import csv, glob, os, sys, json
src, kern = sys.argv[1], sys.argv[2]
def rows(d, suffix):
    out = []
    for f in glob.glob(os.path.join(src, d, "**", "*" + suffix), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out
t = [x for x in rows("time", "kernel_trace.csv") if kern in x["Kernel_Name"]]
t.sort(key=lambda x: int(x["Start_Timestamp"]))
ms = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e6 for x in t]
res = {"kernel": kern, "dispatches": len(ms), "ms_last": ms[-1] if ms else None}
for d in ("sq", "fetch", "write"):
    r = [x for x in rows(d, "counter_collection.csv") if kern in x["Kernel_Name"]]
    by = {}
    for x in r:
        by.setdefault(x["Counter_Name"], {}).setdefault(int(x["Dispatch_Id"]), 0.0)
        by[x["Counter_Name"]][int(x["Dispatch_Id"])] += float(x["Counter_Value"])
    for name, dd in by.items():
        last = dd[max(dd)]
        res[name] = last
if "SQ_WAVE_CYCLES" in res:
    w = res["SQ_WAVE_CYCLES"]
    res["frac_wait_any"] = res["SQ_WAIT_ANY"] / w; res["frac_wait_inst"] = res["SQ_WAIT_INST_ANY"] / w; res["frac_active"] = res["SQ_ACTIVE_INST_ANY"] / w
    res["mfma_busy_of_cu_busy"] = res["SQ_VALU_MFMA_BUSY_CYCLES"] / res["SQ_BUSY_CU_CYCLES"] / 4
if "FETCH_SIZE" in res: res["fetch_bytes_corrected"] = res["FETCH_SIZE"] * 2 * 1024
if "WRITE_SIZE" in res: res["write_bytes"] = res["WRITE_SIZE"] * 1024
json.dump(res, open(os.path.join(src, "summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY

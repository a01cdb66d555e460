#!/bin/bash
# Round profile set for one precision: bench line, rocprofv3 kernel stats, HBM traffic PMC passes.
# usage: tools/profile_bench.sh <f32|f32s|bf16|bf16s> <outdir under gpurun_out>
prec=${1:-f32}; out=${2:-gpurun_out/prof_$prec}
repo=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$repo/$out"; cd /tmp; export TMPDIR=/tmp
python3 "$repo/bench.py" --precision $prec --no-other-modes > "$repo/$out/bench.json" 2> "$repo/$out/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/stats" -o r -- python3 "$repo/bench.py" --precision $prec --no-cpu-baseline --no-other-modes > "$repo/$out/stats.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$repo/$out/pmc_$n" -o r -- python3 "$repo/bench.py" --precision $prec --steps 1 --warmup 0 --no-cpu-baseline --no-other-modes > "$repo/$out/pmc_$n.log" 2>&1
done
python3 - "$repo/$out" <<'PY'
import sys, glob, csv, collections, json
d = sys.argv[1]
vals = collections.defaultdict(list)
for f in glob.glob(d + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "resblock" in r["Kernel_Name"]: vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
o = {k + "_avg": sum(v) / len(v) for k, v in vals.items()}
o.update({k + "_min": min(v) for k, v in vals.items() if k == "FETCH_SIZE"})
o.update({k + "_max": max(v) for k, v in vals.items() if k == "FETCH_SIZE"})
o["launches"] = len(vals.get("FETCH_SIZE", []))
json.dump(o, open(d + "/pmc_traffic_raw.json", "w"), indent=1)
print(json.dumps(o))
PY
tail -1 "$repo/$out/bench.json"

#!/bin/bash
# L2-miss read bytes (FETCH_SIZE x 2, the gfx950 calibration) of one bf16 residual-block launch at B = 512, per layer of a dilation
# cycle: bash tools/fetch_by_layer.sh   (on the GPU box; every profiler pass bounded by timeout)
set -u
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp; export TMPDIR=/tmp
for lay in 0 1 2 3 4 5 6 7 8 9 10 11; do
  rm -rf "$repo/gpurun_out/ft_$lay"
  timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$repo/gpurun_out/ft_$lay" -o r -- python3 "$repo/tools/run_resblock.py" 512 bf16 2 $lay > "$repo/gpurun_out/ft_$lay.log" 2>&1 || echo "layer $lay: profiler pass failed (see gpurun_out/ft_$lay.log)" >&2
done
python3 - "$repo" <<'PY'
import glob, csv, sys
repo = sys.argv[1]
for lay in range(12):
    v = []
    for f in glob.glob(f"{repo}/gpurun_out/ft_{lay}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "resblock" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE": v.append(float(r["Counter_Value"]))
    if not v:
        print("layer", lay, "no counter rows (profiler pass failed)")
        continue
    rd = sum(v) / len(v) * 2048
    print("layer", lay, "read GB %.2f  traffic/algorithmic %.3f" % (rd / 1e9, (rd + 16.78e9) / 33.554e9))
PY

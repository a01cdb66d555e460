#!/bin/bash
# per-kernel split of one white-box gradient step (B = 10, t* = 5): rocprofv3 --kernel-trace --stats around tools/bench_whitebox.py
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in ${MODES:-bf16 f32}; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/wb_$m -o r -- python3 $R/tools/bench_whitebox.py 10 5 $m > $R/gpurun_out/wb_$m.log 2>&1
grep white-box $R/gpurun_out/wb_$m.log
python3 - $R/gpurun_out/wb_$m <<'PY'
import csv,glob,sys
rows=[]
for f in glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True):
    rows+=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:12]: print(f"  {r['Name'][:70]:70s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:9.1f} us  share {float(r['TotalDurationNs'])/tot:.3f}")
PY
done

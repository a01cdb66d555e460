"""Kernel durations and launch gaps of a small-batch purification (graph replay): python tools/trace_small_batch.py <trace.csv>
   producer: rocprofv3 --kernel-trace --output-format csv -d DIR -o r -- python3 tools/run_small_batch.py B mode"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last replay: take the last N kernels where N = kernels per replay (found by the repeating name pattern of the final 1/3)
n = len(rows)
tail = rows[-(n // 3):]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tail]
gap = [(int(tail[i + 1]["Start_Timestamp"]) - int(tail[i]["End_Timestamp"])) / 1e3 for i in range(len(tail) - 1)]
by = {}
for r, d in zip(tail, dur):
    k = r["Kernel_Name"][:60]
    by.setdefault(k, []).append(d)
print(f"{len(tail)} kernels: busy {sum(dur):.0f} us, gaps {sum(g for g in gap if g < 1000):.0f} us (median gap {sorted(gap)[len(gap)//2]:.1f} us)")
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print(f"  {k:60s} n={len(v):4d} avg {sum(v)/len(v):7.1f} us  total {sum(v):8.0f}")

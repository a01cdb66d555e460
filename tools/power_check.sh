#!/bin/bash
# Board power and clocks while one kernel family runs back to back (rocm-smi / amd-smi sampled from the side, ordinary user):
# is the clock a kernel holds a power cap at work?   bash tools/power_check.sh [outfile]
out=${1:-gpurun_out/power_check.txt}
repo=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$(dirname "$repo/$out")"
{
echo "== idle"; rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -E "Power|sclk|mclk|Max" | head -8
for spec in "512 bf16 150 9" "512 f32s 40 9" "256 f32 40 9"; do
  echo "== run_resblock_layers $spec"
  python3 "$repo/tools/run_resblock_layers.py" $spec > /dev/null 2>&1 &
  pid=$!
  sleep 14                                   # model build + weight packing
  for i in 1 2 3 4 5 6; do
    kill -0 $pid 2>/dev/null || break
    rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk" | tr '\n' ' '; echo
    sleep 0.4
  done
  wait $pid
done
} > "$repo/$out" 2>&1
cat "$repo/$out"

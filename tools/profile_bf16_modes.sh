#!/bin/bash
# Round-6 evidence for the two bf16 modes (AP_PREC_BF16 and AP_PREC_BF16_STORE): per-kernel times and HBM-side traffic of one eps
# evaluation of the shipped net at B = 512.  One plain --kernel-trace pass per mode, then SEPARATE --pmc passes (FETCH_SIZE,
# WRITE_SIZE, SQ + GRBM) -- never combined with other trace domains (MI355X_MICROARCH.md, HBM / rocprofv3).
# usage: tools/profile_bf16_modes.sh <outdir under gpurun_out> ; then python tools/summarize_bf16_modes.py gpurun_out/<outdir> r6
out=${1:-gpurun_out/r6_bf16}
repo=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$repo/$out"; cd /tmp; export TMPDIR=/tmp
B=${B:-512}
for mode in bf16 bf16s; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/time_$mode" -o r -- python3 "$repo/tools/run_eps_bf16.py" $B 2 -1 $mode > "$repo/$out/time_$mode.log" 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$repo/$out/fetch_$mode" -o r -- python3 "$repo/tools/run_eps_bf16.py" $B 2 -1 $mode > "$repo/$out/fetch_$mode.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$repo/$out/write_$mode" -o r -- python3 "$repo/tools/run_eps_bf16.py" $B 2 -1 $mode > "$repo/$out/write_$mode.log" 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$repo/$out/sq_$mode" -o r -- python3 "$repo/tools/run_eps_bf16.py" $B 2 -1 $mode > "$repo/$out/sq_$mode.log" 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$repo/$out/lds_$mode" -o r -- python3 "$repo/tools/run_eps_bf16.py" $B 2 -1 $mode > "$repo/$out/lds_$mode.log" 2>&1
done
# keep what travels back small: the per-dispatch csv files only
find "$repo/$out" -name '*.csv' -size +20M -delete
python3 "$repo/tools/summarize_bf16_modes.py" "$repo/$out" r6 --no-copy

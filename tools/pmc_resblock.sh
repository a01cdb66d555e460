#!/bin/bash
# PMC passes over the residual-block kernel alone: tools/pmc_resblock.sh <precision> <outdir>
prec=${1:-bf16}; out=${2:-gpurun_out/pmc_$prec}
repo=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$repo/$out"; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$repo/$out/p$i" -o r -- python3 "$repo/tools/run_resblock.py" ${3:-256} $prec 2 > "$repo/$out/p$i.log" 2>&1
done
python3 - "$repo/$out" <<'PY'
import sys, glob, csv, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "resblock" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print(f"{k:36s} {tot[k] / max(n[k], 1):16.0f}   (per dispatch, {n[k]} dispatches)")
PY

#!/bin/bash
# Round-2 evidence set (run on the GPU box): bench line, rocprofv3 kernel stats of the same command, HBM-traffic and SQ PMC
# passes over the bf16 / fp32 residual-block kernels alone.  Every profiler pass is bounded by `timeout`.
#   bash tools/profile_round2.sh [outdir under gpurun_out]
out=${1:-gpurun_out/r2}
repo=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$repo/$out"; cd /tmp; export TMPDIR=/tmp
timeout 900 python3 "$repo/bench.py" --steps 2 --warmup 1 > "$repo/$out/bench.json" 2> "$repo/$out/bench.err"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/stats" -o r -- python3 "$repo/bench.py" --steps 1 --warmup 1 --no-cpu-baseline > "$repo/$out/stats.log" 2>&1
for prec in bf16 f32; do
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | tr ' ' '_')
    timeout 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$repo/$out/pmc_${prec}_$n" -o r -- python3 "$repo/tools/run_resblock.py" 512 $prec 2 > "$repo/$out/pmc_${prec}_$n.log" 2>&1
  done
done
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$repo/$out/sq_bf16_p$i" -o r -- python3 "$repo/tools/run_resblock.py" 256 bf16 2 > "$repo/$out/sq_bf16_p$i.log" 2>&1
done
python3 - "$repo/$out" <<'PY'
import sys, glob, csv, collections, json
d = sys.argv[1]
res = {}
for prec in ("bf16", "f32"):
    vals = collections.defaultdict(list)
    for f in glob.glob(d + f"/pmc_{prec}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "resblock" in r["Kernel_Name"]: vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[prec] = {k: sum(v) / len(v) for k, v in vals.items()}
sq = collections.defaultdict(list)
for f in glob.glob(d + "/sq_bf16_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "resblock" in r["Kernel_Name"]: sq[r["Counter_Name"]].append(float(r["Counter_Value"]))
res["sq_bf16_B256"] = {k: sum(v) / len(v) for k, v in sq.items()}
json.dump(res, open(d + "/pmc_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
cp "$repo/$out"/stats/*kernel_stats.csv "$repo/$out/kernel_stats.csv" 2>/dev/null
tail -c 400 "$repo/$out/bench.json"
# tools-build evidence: phase stamps of one steady-state tile, timing-only ablations, persistent vs per-tile bit identity
timeout 200 python3 "$repo/tools/trace_resblock_bf16p.py" 256 9 > "$repo/$out/phase_trace.txt" 2>&1
timeout 400 python3 "$repo/tools/dbg_resblock_bf16.py" 256 0 4096 1 2 3 4 8 128 256 384 512 1024 8192 > "$repo/$out/ablation.txt" 2>&1
timeout 200 python3 "$repo/tools/cmp_bf16_kernels.py" 4 > "$repo/$out/cmp_kernels.txt" 2>&1
timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 "$repo/bench.py" --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-other-modes --no-other-configs > "$repo/$out/bench_torchrun_n1.json" 2> "$repo/$out/bench_torchrun_n1.err"
bash "$repo/tools/fetch_by_layer.sh" > "$repo/$out/fetch_by_layer.txt" 2>&1
